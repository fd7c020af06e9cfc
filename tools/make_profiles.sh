#!/bin/bash
# GPU box: regenerates the rocprofv3 summaries kept under profiles/ for ONE round (written to gpurun_out/prof_rNN/; copy them
# into profiles/ with tools/install_profiles.sh --round N).  rocprofv3 is run from /tmp with the program directly after "--";
# counters in their own passes (--kernel-trace --pmc only, never with a trace domain: gpurun refuses that combination).
#   tools/make_profiles.sh --round N [stats] [pmc] [sq] [plain] [sweep4k] [real] [markers] [extras]      (default: all groups)
#     stats    rocprofv3 --kernel-trace --stats of the headline batch, F = 32, configs 2 / 3 / 5, the 2048-frame batch
#     pmc      FETCH_SIZE and WRITE_SIZE passes (separate runs) of the headline batch and configs 3 / 5 / the 2048-frame batch
#     sq       SQ / TCC counter passes of the headline batch (instruction counts and issue cycles of every kernel)
#     plain    the same workloads without the profiler attached (bench_<name>_plain.json) and the default bench line
#     sweep4k  config 5 at F = 16 / 32 / 64 (is 4K's rate the plateau?)
#     real     the headline batch on frames cut from real photographs (bench.py --real-images): rate, relaxation rounds, fallbacks
#     markers  rocprofv3 --kernel-trace --marker-trace with PLI_ROCTX=1 (F = 32): the library's roctx ranges — entry point > stage > launch
#     extras   the headline batch with every kernel alone (per-round times), and the real-image batch with round 4's 64 distinct windows
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
ROUND=6
if [[ $1 == --round ]]; then ROUND=$2; shift; shift; fi
RN=$(printf "r%02d" $ROUND)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$RN
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WHAT="${*:-stats pmc sq plain sweep4k real markers extras}"
run_stats() {  # name, bench args
  local name=$1; shift
  rm -rf $R/gpurun_out/ps_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ps_$name -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/$name.log 2>&1
  python3 $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/ps_$name -name "*.db" | head -1) > $O/kernel_stats_$name.txt
  grep '^{"metric"' $O/$name.log | tail -1 > $O/bench_$name.json
  rm -rf $R/gpurun_out/ps_$name
}
run_pmc() {  # name, "counters", bench args -> appends to pmc_<name>.txt
  local name=$1; local grp=$2; shift; shift
  rm -rf $R/gpurun_out/pp
  timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pp -o s -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O/pmc.log 2>&1
  echo "## --pmc $grp   (bench.py --steps 1 --warmup 0 --no-cpu-baseline $*)" >> $O/pmc_$name.txt
  python3 $R/tools/rocprof_summary.py pmc $(find $R/gpurun_out/pp -name "*.db" | head -1) >> $O/pmc_$name.txt
  rm -rf $R/gpurun_out/pp
}
if [[ $WHAT == *stats* ]]; then
  run_stats default_f256
  run_stats f32 --frames-per-gpu 32
  run_stats config2_single_pair --config 2
  run_stats f2048_sequential --frames-per-gpu 2048
  run_stats config3_720p --config 3
  run_stats config5_4k --config 5
fi
if [[ $WHAT == *pmc* ]]; then
  for w in "default_f256" "f2048_sequential --frames-per-gpu 2048" "config5_4k --config 5" "config3_720p --config 3"; do
    set -- $w; name=$1; shift
    : > $O/pmc_$name.txt
    run_pmc $name FETCH_SIZE "$@"
    run_pmc $name WRITE_SIZE "$@"
  done
fi
if [[ $WHAT == *sq* ]]; then
  for w in "default_f256"; do
    set -- $w; name=sq_$1; shift
    : > $O/pmc_$name.txt
    for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
      run_pmc $name "$grp" "$@"
    done
  done
fi
if [[ $WHAT == *markers* ]]; then
  rm -rf $R/gpurun_out/pm
  PLI_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --marker-trace --stats -d $R/gpurun_out/pm -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-leg --no-large-batch-leg --frames-per-gpu 32 > $O/markers.log 2>&1
  python3 $R/tools/rocprof_summary.py markers $(find $R/gpurun_out/pm -name "*.db" | head -1) > $O/marker_ranges_f32.txt
  rm -rf $R/gpurun_out/pm
fi
cd $R
if [[ $WHAT == *plain* ]]; then
  timeout 900 python3 bench.py --steps 5 --warmup 1 > $O/bench_default.log 2>&1
  grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
  # the other workloads without the profiler attached (the bench_<name>.json beside the kernel stats are runs under rocprofv3)
  for w in "config5_4k --config 5" "config3_720p --config 3" "config2_single_pair --config 2" "f32 --frames-per-gpu 32" "f2048_sequential --frames-per-gpu 2048"; do
    set -- $w; name=$1; shift
    timeout 900 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_${name}_plain.json
  done
fi
if [[ $WHAT == *sweep4k* ]]; then
  : > $O/sweep_config5_4k.jsonl
  for f in 16 32 64; do
    timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --config 5 --frames-per-gpu $f 2>/dev/null | grep '^{"metric"' | tail -1 >> $O/sweep_config5_4k.jsonl
  done
fi
if [[ $WHAT == *real* ]]; then
  timeout 900 python3 bench.py --steps 5 --warmup 1 --real-images --no-host-leg --no-large-batch-leg 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_real_images_f256.json
fi
if [[ $WHAT == *extras* ]]; then
  # the headline batch with every kernel alone and in a row, per-round times (name@round); round 4's real-image workload (64 distinct windows)
  PLI_USE_DEV_LIB=1 PLI_SIDE_MAX=0 PLI_RX_PROFROUNDS=1 timeout 900 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-leg --no-large-batch-leg 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_default_f256_alone_per_round.json
  timeout 900 python3 bench.py --steps 5 --warmup 1 --real-images --unique-frames 64 --no-cpu-baseline --no-host-leg --no-large-batch-leg 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_real_images_64windows_f256.json
fi
ls -la $O
