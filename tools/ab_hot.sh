#!/bin/bash
# GPU box (dev tool, round 6): round 1 of the tile relaxation on the 16-byte records (PLI_TX_HOT=0) against the 8-byte hot records
# (PLI_TX_HOT=1) — kernel times alone, then the memory-side and instruction counters of the growers, one rocprofv3 pass each.
: "${GRAFT_REPO_ROOT:?}"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export PLI_USE_DEV_LIB=${PLI_USE_DEV_LIB-1}      # (environment switches are read by the development build of the library only)
PAT=${PAT:-"k_tx_grow|k_tx_round2|k_lsd_front|k_tx_sort"}
tools/ab_env.sh PLI_TX_HOT=0 PLI_TX_HOT=1 PLI_TX_HOT=0 PLI_TX_HOT=1
for h in 0 1; do
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    echo "== PLI_TX_HOT=$h --pmc $grp"
    PLI_TX_HOT=$h PLI_SIDE_MAX=0 tools/pmc_quick.sh "$grp" "$PAT" --no-host-leg --no-large-batch-leg
  done
done
