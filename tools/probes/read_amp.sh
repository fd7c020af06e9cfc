#!/bin/bash
# GPU box: the L2's memory-side read counters on tools/probes/read_amp's patterns of known size
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_MISS_sum"; do
  rm -rf /tmp/ra
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/ra -o s -- $R/tools/probes/read_amp > /tmp/ra.log 2>&1
  tail -2 /tmp/ra.log
  python3 - <<PY
import sqlite3, glob
db = glob.glob('/tmp/ra/**/*.db', recursive=True)
cur = sqlite3.connect(db[0]).cursor()
for name, cn, v, d in sorted(cur.execute("select kernel_name, counter_name, value, duration from counters_collection")):
    print("%-50s %-28s %16.1f %10.0f ns" % (name[:50], cn, v, d))
PY
done
