#!/bin/bash
export PLI_USE_DEV_LIB=${PLI_USE_DEV_LIB-1}      # (environment switches are read by the development build of the library only)
# GPU box dev tool: bench.py under several values of one environment variable.
#   tools/sweep_env.sh VAR "v1 v2 ..." kernel1,kernel2 [bench args]
var=$1; vals=$2; kern=$3; shift; shift; shift
for v in $vals; do
  env $var=$v python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | VAL=$v KERN=$kern VAR=$var python3 -c '
import json,sys,os
d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_step"]
print(os.environ["VAR"], os.environ["VAL"], round(d["value"]), "f/s", round(d["ms_per_step"],1), "ms", {n:k.get(n) for n in os.environ["KERN"].split(",")})'
done
