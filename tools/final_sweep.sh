#!/bin/bash
# GPU box: the sweeps of DESIGN.md 2's table on the library as it stands (the round's last build): ~12 minutes.  Every sweep's last lines
# are shown AND its exit status is kept: the script ends non-zero when any sweep crashed, timed out or found a mismatch.
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
B=${1:-260000}
FAIL=0
run() {  # label, lines to show, command...
  local label=$1 n=$2; shift; shift
  echo "== $label"
  "$@" > gpurun_out/sweep_step.log 2>&1
  local rc=$?
  tail -"$n" gpurun_out/sweep_step.log
  if [ $rc -ne 0 ] || grep -qi "traceback" gpurun_out/sweep_step.log; then echo "   -> FAILED (exit $rc)"; FAIL=1; fi
}
run "tile $B 256"              1 timeout 1500 python tools/cross_check.py --set tile $B 256
run "hot $((B+2000)) 64"       1 timeout 1500 python tools/cross_check.py --set hot $((B+2000)) 64
run "schedules $((B+1000)) 128" 1 timeout 900 python tools/cross_check.py --set schedules $((B+1000)) 128
run "sizes 600"                1 timeout 900 python tools/cross_check.py --set sizes 600
run "fuzz_frames"              2 timeout 900 python tools/fuzz_frames.py
run "fuzz_config"              2 timeout 900 python tools/fuzz_config.py
for a in "1 300" "32 150" "256 100"; do run "soak $a" 1 timeout 900 python tools/soak_determinism.py $a; done
run "rounds synthetic"         2 timeout 900 python tools/rounds_sweep.py 3
run "rounds photographs"       2 timeout 900 python tools/rounds_sweep.py --real 2
rm -f gpurun_out/sweep_step.log
[ $FAIL -eq 0 ] && echo "ALL SWEEPS CLEAN" || echo "SOME SWEEP FAILED"
exit $FAIL
