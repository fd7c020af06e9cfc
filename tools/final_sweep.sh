#!/bin/bash
# GPU box: the sweeps of DESIGN.md 2's table on the library as it stands (the round's last build): ~10 minutes
cd $GRAFT_REPO_ROOT
B=${1:-260000}
echo "== tile $B 256";        timeout 1500 python tools/cross_check.py --set tile $B 256 2>&1 | tail -1
echo "== schedules $((B+1000)) 128"; timeout 900 python tools/cross_check.py --set schedules $((B+1000)) 128 2>&1 | tail -1
echo "== sizes 600";          timeout 900 python tools/cross_check.py --set sizes 600 2>&1 | tail -1
echo "== fuzz_frames";        timeout 900 python tools/fuzz_frames.py 2>&1 | tail -2
echo "== fuzz_config";        timeout 900 python tools/fuzz_config.py 2>&1 | grep -c -i "mismatch\|error\|traceback"
echo "== soak";               for a in "1 300" "32 150" "256 100"; do timeout 900 python tools/soak_determinism.py $a 2>&1 | tail -1; done
echo "== rounds";             timeout 900 python tools/rounds_sweep.py 3 2>&1 | tail -2; timeout 900 python tools/rounds_sweep.py --real 2 2>&1 | tail -2
