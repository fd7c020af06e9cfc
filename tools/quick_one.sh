#!/bin/bash
export PLI_USE_DEV_LIB=${PLI_USE_DEV_LIB-1}      # (environment switches are read by the development build of the library only)
# GPU box dev tool: one bench.py line, condensed.  tools/quick_one.sh [bench args]   (environment passes through)
timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
top=sorted(k.items(), key=lambda kv:-kv[1])[:8]
print(' '.join(sys.argv[1:]), round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', [(n,round(v,1)) for n,v in top])" "$@"
