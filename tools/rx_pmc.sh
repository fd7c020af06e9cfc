#!/bin/bash
# GPU box: PMC counters of the relaxation kernels at F=32 (one pass per counter group, kernel-trace only)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_EA_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/rxpmc$i -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --frames-per-gpu 32 --lsd-mode 1 > $R/gpurun_out/rxpmc$i.log 2>&1
  db=$(find $R/gpurun_out/rxpmc$i -name "*.db" | head -1)
  python3 $R/tools/rocprof_summary.py pmc $db | grep "k_rx_grow\|k_rx_seed\|k_rx_classify\|k_lsd_grad\|^#"
done
