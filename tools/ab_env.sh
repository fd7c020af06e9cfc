#!/bin/bash
# GPU box (dev tool): the headline batch under a list of environment settings, "VAR=val[,VAR2=val2] ..." each, kernels alone (PLI_SIDE_MAX=0)
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
cd "$GRAFT_REPO_ROOT" || exit 1
export PLI_USE_DEV_LIB=${PLI_USE_DEV_LIB-1}      # (environment switches are read by the development build of the library only)
for setting in "$@"; do
  (
    IFS=, ; for kv in $setting; do [ "$kv" != "none" ] && export "$kv"; done; unset IFS
    PLI_SIDE_MAX=${PLI_SIDE_MAX:-0} python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-large-batch-leg $BENCH_ARGS > gpurun_out/env.json 2>/dev/null
    echo "[$setting] $(python tools/round_times.py gpurun_out/env.json k_tx_grow k_tx_round2 k_tx_diffmark k_tx_prep k_rx_rect k_tx_tail k_tx_sort | tr '\n' ' ')"
  )
done
