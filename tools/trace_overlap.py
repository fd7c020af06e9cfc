#!/usr/bin/env python3
"""Dev tool: lists the kernels of a rocprofv3 --kernel-trace database with start/end (ms) and queue, to see whether
kernels of different HIP streams overlap.   python tools/trace_overlap.py <results.db> [min_ms]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
minms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if not view:
    print(tabs); sys.exit(1)
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
print(cols)
rows = list(cur.execute("select name, start, end, queue_id, stream_id from %s order by start" % view))
t0 = rows[0][1]
for name, s, e, q, st in rows:
    if (e - s) * 1e-6 >= minms:
        print("%-28s q%-3s s%-3s %10.2f -> %10.2f  (%8.2f ms)" % (name.split("(")[0][-28:], q, st, (s - t0) * 1e-6, (e - t0) * 1e-6, (e - s) * 1e-6))
