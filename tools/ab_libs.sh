#!/bin/bash
# GPU box (dev tool): the headline batch (kernels alone) with variant builds of the library; args: build dirs ("base" = the shipped one).
# With CHECK=1 the LSD parity subset runs against every variant first.
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
cd "$GRAFT_REPO_ROOT" || exit 1
export PLI_USE_DEV_LIB=1      # (environment switches are read by the development build of the library only)
for v in "$@"; do
  if [ "$v" != base ]; then export PLI_LIB_PATH=$(ls $GRAFT_REPO_ROOT/$v/libpli_frontend_dev.so 2>/dev/null || echo $GRAFT_REPO_ROOT/$v/libpli_frontend.so); else unset PLI_LIB_PATH; fi
  if [ -n "$CHECK" ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lsd or config2 or real or hostile or tile or key_mode" 2>&1 | tail -2; fi
  for i in 1 2; do
    PLI_SIDE_MAX=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-large-batch-leg > gpurun_out/lib.json 2>/dev/null
    echo "[$v] $(python tools/round_times.py gpurun_out/lib.json k_tx_grow k_tx_round2 k_tx_diffmark k_tx_prep k_rx_rect k_tx_tail | tr '\n' ' ')"
  done
done
