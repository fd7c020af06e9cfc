#!/bin/bash
# copies the summaries written by tools/make_profiles_r02.sh (gpurun_out/prof_r02) into profiles/ as r02_* and builds r02_traffic.json
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/prof_r02; P=$R/profiles
for f in $O/kernel_stats_*.txt $O/pmc_*.txt $O/bench_*.json; do [ -s "$f" ] && cp $f $P/r02_$(basename $f); done
python3 $R/tools/make_traffic_json_r02.py $O > $P/r02_traffic.json
ls $P | grep r02
