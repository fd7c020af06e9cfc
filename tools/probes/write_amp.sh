#!/bin/bash
# GPU box: WRITE_SIZE (and the raw TCC_EA0_WRREQ counters) of tools/probes/write_amp's store patterns
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for grp in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  rm -rf /tmp/wa
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/wa -o s -- $R/tools/probes/write_amp > /tmp/wa.log 2>&1
  tail -4 /tmp/wa.log
  python3 - <<PY
import sqlite3, glob
db = glob.glob('/tmp/wa/**/*.db', recursive=True)
cur = sqlite3.connect(db[0]).cursor()
for name, cn, v, d in sorted(cur.execute("select kernel_name, counter_name, value, duration from counters_collection")):
    print("%-60s %-24s %16.1f %10.0f ns" % (name[:60], cn, v, d))
PY
done
