cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hot_records or config3_f2f or projection_searches_dense or local_map" 2>&1 | tail -15
echo "== cross_check hot"; timeout 900 python tools/cross_check.py --set hot 300000 32 2>&1 | tail -6
echo "== timing"; tools/ab_env.sh PLI_TX_HOT=0 PLI_TX_HOT=1 PLI_TX_HOT=0 PLI_TX_HOT=1
