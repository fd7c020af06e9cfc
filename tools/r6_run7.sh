#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== hot tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hot or lsd or config2 or key_mode or product" 2>&1 | tail -3
echo "== timing alone"; tools/ab_env.sh PLI_TX_HOT=0 PLI_TX_HOT=2 PLI_TX_HOT=0 PLI_TX_HOT=2
echo "== line"
tools/ab_full.sh build/r05 base:PLI_TX_HOT=0 base:PLI_TX_HOT=2 build/r05 base:PLI_TX_HOT=2
echo "== real images"
BENCH_ARGS="--real-images" tools/ab_full.sh build/r05 base:PLI_TX_HOT=2 build/r05 base:PLI_TX_HOT=2
echo "== 4K"
BENCH_ARGS="--config 5" tools/ab_full.sh build/r05 base:PLI_TX_HOT=2
echo "== 720p"
BENCH_ARGS="--config 3" tools/ab_full.sh build/r05 base:PLI_TX_HOT=2
echo "== single pair / F=32"
BENCH_ARGS="--config 2" tools/ab_full.sh build/r05 base:PLI_TX_HOT=2
BENCH_ARGS="--frames-per-gpu 32" tools/ab_full.sh build/r05 base:PLI_TX_HOT=2
