#!/bin/bash
# GPU box: SQ counters of the sequential LSD grower at 1024 frames (written to gpurun_out/pmc_grow.txt).
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
R=$GRAFT_REPO_ROOT
export PLI_USE_DEV_LIB=${PLI_USE_DEV_LIB-1}      # (environment switches are read by the development build of the library only)
O=$R/gpurun_out/pmc_grow.txt
cd /tmp && export TMPDIR=/tmp
: > $O
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH"; do
  rm -rf $R/gpurun_out/pp
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pp -o s -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${FRAMES:+--frames-per-gpu $FRAMES} ${MODE:+--lsd-mode $MODE} > $R/gpurun_out/pmc_grow.log 2>&1
  echo "## --pmc $grp" >> $O
  python3 $R/tools/rocprof_summary.py pmc $(find $R/gpurun_out/pp -name "*.db" | head -1) | grep "^#\|k_lsd_grow" >> $O
done
rm -rf $R/gpurun_out/pp
cat $O
