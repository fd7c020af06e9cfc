#!/bin/bash
# copies the summaries written by tools/make_profiles.sh (gpurun_out/prof_out) into profiles/ under the round's names
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/prof_out; P=$R/profiles; N=${1:-r01}
for f in kernel_stats_f2048_sequential.txt kernel_stats_f32_relaxation.txt kernel_stats_f1_relaxation.txt pmc_fetch_write_f2048.txt \
         pmc_relaxation_f32.txt bench_default.json bench_f32_relaxation.json bench_f1_relaxation.json; do cp $O/$f $P/${N}_$f; done
cp $O/traffic.json $P/${N}_traffic.json
