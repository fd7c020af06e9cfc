#!/bin/bash
# copies the summaries written by tools/make_profiles_r03.sh (gpurun_out/prof_r03) into profiles/ as r03_* and builds r03_traffic.json
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/prof_r03; P=$R/profiles
for f in $O/kernel_stats_*.txt $O/pmc_*.txt $O/bench_*.json; do [ -s "$f" ] && cp $f $P/r03_$(basename $f); done
python3 $R/tools/make_traffic_json_r03.py $O > $P/r03_traffic.json
ls $P | grep r03
