#!/bin/bash
# copies the summaries written by tools/make_profiles.sh --round N (gpurun_out/prof_rNN) into profiles/ as rNN_* and builds
# rNN_traffic.json (read back by bench.py for roofline.traffic / roofline.issue):   tools/install_profiles.sh --round N
ROUND=6
if [[ $1 == --round ]]; then ROUND=$2; fi
RN=$(printf "r%02d" $ROUND)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/prof_$RN; P=$R/profiles
for f in $O/kernel_stats_*.txt $O/pmc_*.txt $O/bench_*.json $O/sweep_*.jsonl $O/marker_*.txt $O/rounds_*.json; do [ -s "$f" ] && cp $f $P/${RN}_$(basename $f); done
python3 $R/tools/make_traffic_json.py $O > $P/${RN}_traffic.json
ls $P | grep $RN
