#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== full gpu suite"; timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
echo "== cross_check hot"; timeout 900 python tools/cross_check.py --set hot 510000 32 2>&1 | tail -2
echo "== cross_check sizes"; timeout 900 python tools/cross_check.py --set sizes 200 2>&1 | tail -2
echo "== alone: pitch 902 (NOPAD) vs 912 (default)"
tools/ab_env.sh PLI_LSD_NOPAD=1 none PLI_LSD_NOPAD=1 none
echo "== line"
tools/ab_full.sh base:PLI_LSD_NOPAD=1 base base:PLI_LSD_NOPAD=1 base
echo "== photographs"
BENCH_ARGS="--real-images" tools/ab_full.sh base:PLI_LSD_NOPAD=1 base base:PLI_LSD_NOPAD=1 base
echo "== 720p / F32 / 4K"
BENCH_ARGS="--config 3" tools/ab_full.sh base:PLI_LSD_NOPAD=1 base
BENCH_ARGS="--frames-per-gpu 32" tools/ab_full.sh base:PLI_LSD_NOPAD=1 base
BENCH_ARGS="--config 5" tools/ab_full.sh base:PLI_LSD_NOPAD=1 base
