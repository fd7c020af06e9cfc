#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hot_records or lsd or config2 or tile or key_mode or hostile or real" 2>&1 | tail -4
echo "== cross_check hot"; timeout 900 python tools/cross_check.py --set hot 310000 32 2>&1 | tail -3
echo "== timing"; tools/ab_env.sh PLI_TX_HOT=0 PLI_TX_HOT=1 PLI_TX_HOT=0 PLI_TX_HOT=1
for h in 0 1; do echo "== HOT=$h"; PLI_TX_HOT=$h PLI_SIDE_MAX=0 tools/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "k_tx_grow|k_rx_rect" --no-host-leg --no-large-batch-leg; done
echo "== fast cells A/B (r05 library against this one)"
KERNELS="k_fast_cells k_octree k_blur k_describe" tools/ab_kernels.sh build/r05 base
