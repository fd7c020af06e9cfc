#!/bin/bash
# GPU box (dev tool): one bench line per "config[:ENV=val,...]" argument, kernels of the line chain + rate; e.g. tools/quick_cfg.sh 3 3:PLI_TX_KEYS=0 5
: "${GRAFT_REPO_ROOT:?}"      # (GPU box: gpurun exports it)
cd "$GRAFT_REPO_ROOT" || exit 1
export PLI_USE_DEV_LIB=${PLI_USE_DEV_LIB-1}      # (environment switches are read by the development build of the library only)
for spec in "$@"; do
  cfg=${spec%%:*}; envs=""; [ "$spec" != "$cfg" ] && envs=${spec#*:}
  (
    IFS=, ; for kv in $envs; do export "$kv"; done; unset IFS
    python bench.py --config $cfg --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-host-leg --no-large-batch-leg $BENCH_ARGS > gpurun_out/cfg.json 2>gpurun_out/cfg.err || tail -3 gpurun_out/cfg.err
    echo "[$spec] $(python tools/round_times.py gpurun_out/cfg.json k_tx k_lsd k_rx | tr '\n' ' ')"
    python - <<'PY'
import json
d=json.loads(open("gpurun_out/cfg.json").read().strip().splitlines()[-1])
print("   parity:", d.get("parity"), "rounds:", d.get("lsd_rounds", d.get("config", {}).get("lsd_rounds")))
PY
  )
done
