for a in "--frames-per-gpu 32" "--config 2" "--config 3" "--config 5" "--frames-per-gpu 768" "--frames-per-gpu 768 --lsd-mode 2" "--frames-per-gpu 1024 --lsd-mode 3" "--frames-per-gpu 1024 --lsd-mode 2"; do
export PLI_USE_DEV_LIB=${PLI_USE_DEV_LIB-1}      # (environment switches are read by the development build of the library only)
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
top=sorted(k.items(), key=lambda kv:-kv[1])[:6]
print('$a', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', d.get('single_pair'), [(n,round(v,1)) for n,v in top])"
done
