#!/bin/bash
# GPU box: regenerates the rocprofv3 summaries kept under profiles/ (written to gpurun_out/prof_out/).
# rocprofv3 is run from /tmp with the program directly after "--"; counters in their own passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_stats() {  # name, bench args
  local name=$1; shift
  rm -rf $R/gpurun_out/ps_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ps_$name -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/$name.log 2>&1
  python3 $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/ps_$name -name "*.db" | head -1) > $O/kernel_stats_$name.txt
  grep '^{"metric"' $O/$name.log | tail -1 > $O/bench_$name.json
}
run_stats f2048_sequential
run_stats f32_relaxation --frames-per-gpu 32
run_stats f1_relaxation --frames-per-gpu 1
: > $O/pmc_relaxation_f32.txt
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_EA_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pp
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pp -o s -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames-per-gpu 32 > $O/pmc.log 2>&1
  echo "## --pmc $grp   (bench.py --steps 1 --warmup 0 --frames-per-gpu 32)" >> $O/pmc_relaxation_f32.txt
  python3 $R/tools/rocprof_summary.py pmc $(find $R/gpurun_out/pp -name "*.db" | head -1) | grep "^#\|k_rx_\|k_lsd_grad" >> $O/pmc_relaxation_f32.txt
done
: > $O/pmc_fetch_write_f2048.txt
for grp in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pp
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pp -o s -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc.log 2>&1
  echo "## --pmc $grp   (bench.py --steps 1 --warmup 0, 2048 frames)" >> $O/pmc_fetch_write_f2048.txt
  python3 $R/tools/rocprof_summary.py pmc $(find $R/gpurun_out/pp -name "*.db" | head -1) >> $O/pmc_fetch_write_f2048.txt
done
python3 $R/tools/make_traffic_json.py $O 2048 > $O/traffic.json
cp $O/traffic.json $R/profiles/r01_traffic.json      # bench.py reads the per-launch traffic of the dominant kernel from here
cd $R && timeout 600 python3 bench.py --steps 3 --warmup 1 > $O/bench_default.log 2>&1
grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
rm -rf $R/gpurun_out/ps_* $R/gpurun_out/pp
ls -la $O
