#!/bin/bash
# GPU box (dev tool): as tools/ab_libs.sh, listing the kernels whose name starts with $KERNELS (default: the front pass and the sort);
# CHECK=1 runs the LSD parity subset against every variant first.   KERNELS="k_lsd k_tx_sort" tools/ab_kernels.sh base build/variant
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
cd "$GRAFT_REPO_ROOT" || exit 1
export PLI_USE_DEV_LIB=1      # (environment switches are read by the development build of the library only)
for v in "$@"; do
  if [ "$v" != base ]; then export PLI_LIB_PATH=$(ls $GRAFT_REPO_ROOT/$v/libpli_frontend_dev.so 2>/dev/null || echo $GRAFT_REPO_ROOT/$v/libpli_frontend.so); else unset PLI_LIB_PATH; fi
  if [ -n "$CHECK" ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lsd or config2 or real or hostile or tile or key_mode" 2>&1 | tail -2; fi
  for i in 1 2; do
    PLI_SIDE_MAX=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-large-batch-leg > gpurun_out/lib.json 2>/dev/null
    echo "[$v] $(python tools/round_times.py gpurun_out/lib.json ${KERNELS:-k_lsd_front k_tx_sort} | tr '\n' ' ')"
  done
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-large-batch-leg > gpurun_out/lib.json 2>/dev/null
  echo "[$v] line: $(python tools/round_times.py gpurun_out/lib.json k_lsd_front | tr '\n' ' ')"
done
