#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== full gpu suite"; timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
echo "== cross_check hot"; timeout 900 python tools/cross_check.py --set hot 330000 32 2>&1 | tail -3
echo "== timing alone"; tools/ab_env.sh PLI_TX_HOT=0 PLI_TX_HOT=1 PLI_TX_HOT=2 PLI_TX_HOT=0 PLI_TX_HOT=2
echo "== line: hot0 / hot1 / hot2"
tools/ab_full.sh base:PLI_TX_HOT=0 base:PLI_TX_HOT=1 base:PLI_TX_HOT=2 base:PLI_TX_HOT=0 base:PLI_TX_HOT=2
echo "== real images"
BENCH_ARGS="--real-images" tools/ab_full.sh base:PLI_TX_HOT=0 base:PLI_TX_HOT=2 base:PLI_TX_HOT=0 base:PLI_TX_HOT=2
