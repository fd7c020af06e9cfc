#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_real_images.py -q -m gpu -x -k "hot or lsd or config2 or config3 or config5 or key_mode or owner_word or tile or real or photographs or hostile" 2>&1 | tail -4
echo "== cross_check hot"; timeout 900 python tools/cross_check.py --set hot 500000 32 2>&1 | tail -2
echo "== alone: lazy (PACK1=2) vs sort-written whole records (default)"
tools/ab_env.sh PLI_TX_PACK1=2 none PLI_TX_PACK1=2 none
echo "== line"
tools/ab_full.sh build/r05 base:PLI_TX_PACK1=2 base build/r05 base
echo "== photographs"
BENCH_ARGS="--real-images" tools/ab_full.sh build/r05 base:PLI_TX_PACK1=2 base
echo "== 4K / 720p / F32 / pair"
BENCH_ARGS="--config 5" tools/ab_full.sh build/r05 base
BENCH_ARGS="--config 3" tools/ab_full.sh build/r05 base:PLI_TX_PACK1=2 base
BENCH_ARGS="--frames-per-gpu 32" tools/ab_full.sh build/r05 base:PLI_TX_PACK1=2 base
BENCH_ARGS="--config 2" tools/ab_full.sh build/r05 base:PLI_TX_PACK1=2 base
