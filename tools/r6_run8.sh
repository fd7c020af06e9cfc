#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== sort-written ids on the hot plane (PLI_TX_PACK1=1) against lazy ids, alone"
tools/ab_env.sh none PLI_TX_PACK1=1 none PLI_TX_PACK1=1
echo "== line"
tools/ab_full.sh base base:PLI_TX_PACK1=1 base base:PLI_TX_PACK1=1
echo "== photographs"
BENCH_ARGS="--real-images" tools/ab_full.sh base base:PLI_TX_PACK1=1
echo "== 720p"
BENCH_ARGS="--config 3" tools/ab_full.sh base base:PLI_TX_PACK1=1
tools/make_profiles.sh --round 6 plain real extras > gpurun_out/r6_make_profiles2.log 2>&1
