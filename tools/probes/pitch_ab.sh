#!/bin/bash
# GPU box: the headline batch at image widths whose SCALED rows do (760 -> 912 = 57 x 16) and do not (752 -> 903) start on
# 64-byte boundaries, every kernel alone (PLI_SIDE_MAX=0): which passes over the LSD planes pay for the misaligned pitch?
cd $GRAFT_REPO_ROOT
for w in 752 760 752 760; do
  PLI_SIDE_MAX=0 python bench.py --width $w --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg --no-large-batch-leg > gpurun_out/pitch.json 2>gpurun_out/pitch.err || tail -3 gpurun_out/pitch.err
  echo "== width $w: $(python tools/round_times.py gpurun_out/pitch.json k_ | tr '\n' ' ')"
done
