#!/bin/bash
# GPU box (dev tool): full per-kernel listing (main + side stream) of one bench line per "lib[:ENV=val,...]" argument ("base" = the shipped library);
# BENCH_ARGS adds bench.py arguments.   tools/ab_full.sh build/r04 base base:PLI_SIDE_DEFER_MAX=64
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
cd "$GRAFT_REPO_ROOT" || exit 1
export PLI_USE_DEV_LIB=1      # (environment switches are read by the development build of the library only)
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#*:}
  (
    if [ "$lib" != base ]; then export PLI_LIB_PATH=$(ls $GRAFT_REPO_ROOT/$lib/libpli_frontend_dev.so 2>/dev/null || echo $GRAFT_REPO_ROOT/$lib/libpli_frontend.so); fi
    IFS=, ; for kv in $envs; do export "$kv"; done; unset IFS
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-large-batch-leg $BENCH_ARGS > gpurun_out/full.json 2>/dev/null
    python - "$spec" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/full.json").read().strip().splitlines()[-1])
k = dict(d["roofline"]["kernel_ms_per_step"]); s = d["roofline"].get("side_stream_kernel_ms_per_step", {})
print("[%s] %.1f f/s %.3f ms rounds %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["lsd_rounds"]["needed_by_slowest_image"]))
print("   main:", " ".join("%s %.2f" % (n.replace("k_", ""), v) for n, v in sorted(k.items(), key=lambda kv: -kv[1])[:16]), "| sum %.2f" % sum(k.values()))
print("   side:", " ".join("%s %.2f" % (n.replace("k_", ""), v) for n, v in sorted(s.items(), key=lambda kv: -kv[1])[:8]))
PY
  )
done
