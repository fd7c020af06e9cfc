#!/bin/bash
# GPU box: WRITE_SIZE of the LSD kernels per LAUNCH over two steps of the headline batch, with the launch geometry
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ws
timeout 900 rocprofv3 --kernel-trace --pmc ${1:-WRITE_SIZE} -d /tmp/ws -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-leg --no-large-batch-leg > /tmp/ws.log 2>&1
tail -1 /tmp/ws.log | cut -c1-300
python3 - <<PY
import sqlite3, glob
db = glob.glob('/tmp/ws/**/*.db', recursive=True)
cur = sqlite3.connect(db[0]).cursor()
rows = {}
for name, cn, v, d, gx, gy, gz, wx, lds, scr in cur.execute("select kernel_name, counter_name, value, duration, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, lds_block_size, scratch_size from counters_collection order by start"):
    n = name.split('(')[0].replace('void ', '').replace('pli::', '')
    rows.setdefault((n, cn), []).append((v, d, gx, gy, gz, wx, lds, scr))
for (n, cn), r in sorted(rows.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    if sum(x[0] for x in r) > 1e5: print("%-24s %-22s %3d launches  %s  grid %s wg %d lds %d scratch %d" % (n[:24], cn, len(r), " ".join("%.0f" % x[0] for x in r[:4]), r[0][2:5], r[0][5], r[0][6], r[0][7]))
PY
