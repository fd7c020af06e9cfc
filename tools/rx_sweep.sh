#!/bin/bash
# run on the GPU box: builds variants of the group grower and times F=32
cd $GRAFT_REPO_ROOT/pli_slam_amd/csrc
for cfg in "16 512" "16 256" "16 128" "8 256" "8 128"; do
  set -- $cfg
  touch lsd_relax.hip
  make EXTRA="-DRX_GL_=$1 -DRX_GQ_=$2" >/dev/null 2>&1
  echo "GL=$1 GQ=$2"
  (cd ../.. && PLI_RX_PROFROUNDS=1 timeout 100 python bench.py --no-cpu-baseline --frames-per-gpu 32 --lsd-mode 1 --steps 2 2>&1 | tail -1 | python3 tools/rx_rounds.py | grep "value\|totals")
done
