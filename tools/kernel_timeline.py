#!/usr/bin/env python3
"""GPU box (dev tool): the kernels of the LAST step of a rocprofv3 --kernel-trace run as a timeline — per stream: busy time, gaps, and the
list of kernels with start (ms from the step's first kernel), duration and how many other kernels ran at the same time.
   python tools/kernel_timeline.py <results.db> [first kernel of a step, default k_ingest]"""
import re, sqlite3, sys
db = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_ingest"
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
view = "kernels" if "kernels" in tabs else kd[0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
name_c = "name" if "name" in cols else "kernel_name"
q = "select %s, start, end, %s from %s order by start" % (name_c, "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0"), view)
rows = [(re.sub(r"^(void )?(pli::)?", "", n).split("(")[0], s, e, st) for n, s, e, st in cur.execute(q)]
starts = [i for i, r in enumerate(rows) if r[0].startswith(first)]
if not starts:
    print("no kernel named", first, "; tables:", tabs, "columns:", cols); sys.exit(1)
i0 = starts[-1]
step = rows[i0:]
t0 = step[0][1]
streams = {}
for n, s, e, st in step:
    streams.setdefault(st, []).append((n, s, e))
print("# last step: %d kernels on %d streams, %.3f ms from the first start to the last end" % (len(step), len(streams), (max(r[2] for r in step) - t0) / 1e6))
ev = sorted([(s, 1) for _, s, e, _ in step] + [(e, -1) for _, s, e, _ in step])
busy1 = busy2 = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
print("# some kernel running %.3f ms, two or more %.3f ms" % (busy1 / 1e6, busy2 / 1e6))
for st, ks in streams.items():
    b = sum(e - s for _, s, e in ks)
    gaps = [(ks[i + 1][1] - ks[i][2]) for i in range(len(ks) - 1)]
    print("# stream %s: %d kernels, busy %.3f ms, first start %.3f, last end %.3f, gaps > 20 us: %d (sum %.3f ms)" %
          (st, len(ks), b / 1e6, (ks[0][1] - t0) / 1e6, (ks[-1][2] - t0) / 1e6, sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
print("%-9s %-9s %-8s %-28s %s" % ("start_ms", "dur_ms", "stream", "kernel", "others running at its midpoint"))
for n, s, e, st in step:
    mid = (s + e) / 2
    others = sorted(set(m for m, s2, e2, st2 in step if s2 <= mid <= e2 and (m, s2) != (n, s)))
    if e - s >= 50000:
        print("%-9.3f %-9.3f %-8s %-28s %s" % ((s - t0) / 1e6, (e - s) / 1e6, st, n[:28], " ".join(o[:18] for o in others[:4])))
