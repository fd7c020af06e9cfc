#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== orb parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "orb or fast or config2 or config5 or fuzz" 2>&1 | tail -5
echo "== timing nofold variant (HOT=1) vs shipped"
export PLI_TX_HOT=1
tools/ab_libs.sh base build/nofold base build/nofold
for v in base build/nofold; do
  if [ "$v" != base ]; then export PLI_LIB_PATH=$GRAFT_REPO_ROOT/$v/libpli_frontend.so; else unset PLI_LIB_PATH; fi
  echo "== $v"; PLI_SIDE_MAX=0 tools/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "k_tx_grow_h2|k_fast_cells" --no-host-leg --no-large-batch-leg
done
unset PLI_LIB_PATH; unset PLI_TX_HOT
echo "== fast cells A/B (r05 library against this one)"
KERNELS="k_fast_cells k_octree k_blur k_describe" tools/ab_kernels.sh build/r05 base
