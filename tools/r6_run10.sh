#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== width 752 (LSD rows of 902 pixels: misaligned): lazy vs sort-written, alone"
tools/ab_env.sh PLI_TX_PACK1=2 none PLI_TX_PACK1=2 none
echo "== width 760 (LSD rows of 912 pixels = 57 x 16: 64-byte aligned), alone"
BENCH_ARGS="--width 760" tools/ab_env.sh PLI_TX_PACK1=2 none PLI_TX_PACK1=2 none
echo "== width 760, line"
BENCH_ARGS="--width 760" tools/ab_full.sh base:PLI_TX_PACK1=2 base base:PLI_TX_PACK1=2 base
