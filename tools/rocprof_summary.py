#!/usr/bin/env python3
"""Turns rocprofv3's rocpd sqlite output into the text summaries committed under profiles/.

  python tools/rocprof_summary.py stats  <results.db>            kernel-trace --stats table
  python tools/rocprof_summary.py pmc    <results.db> [counter]  per-kernel counter sums (KB for FETCH/WRITE_SIZE)
  python tools/rocprof_summary.py markers <results.db>           roctx ranges of a --marker-trace run (PLI_ROCTX=1): totals per range name
                                                                 and the tree of the last pli_batch_run call
"""
import json
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"(?:void )?(?:pli::)?([A-Za-z0-9_:]+)", name)
    n = m.group(1) if m else name
    return n if len(n) < 60 else n[:57] + "..."


def stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats  (durations in microseconds)")
    print("%-34s %8s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        print("%-34s %8d %14.3f %14.3f %8.3f" % (short(name), calls, tot, avg, pct))


def pmc(path, counter=None):
    cur = sqlite3.connect(path).cursor()
    q = "select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection"
    if counter:
        q += " where counter_name = '%s'" % counter
    q += " group by kernel_name, counter_name order by sum(value) desc"
    print("# rocprofv3 --pmc  (FETCH_SIZE / WRITE_SIZE are in KB as reported; uncorrected)")
    print("%-34s %-12s %8s %16s %16s %12s" % ("kernel", "counter", "launches", "sum", "avg_per_launch", "avg_ns"))
    for name, cn, n, s, a, d in cur.execute(q):
        print("%-34s %-12s %8d %16.3f %16.3f %12.0f" % (short(name), cn, n, s, a, d))


def markers(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select stack_id, parent_stack_id, start, end, extdata from regions where category like 'MARKER%' order by start"))
    rs = [(sid, par, st, en, json.loads(ext).get("message", "?")) for sid, par, st, en, ext in rows]
    print("# rocprofv3 --kernel-trace --marker-trace with PLI_ROCTX=1: host-side roctx ranges (entry point > stage > kernel launch)")
    tot = {}
    for sid, par, st, en, msg in rs:
        c = tot.setdefault(msg, [0, 0])
        c[0] += 1; c[1] += en - st
    print("%-28s %8s %14s %12s" % ("range", "count", "total_us", "avg_us"))
    for msg, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("%-28s %8d %14.1f %12.1f" % (msg, n, d / 1e3, d / 1e3 / n))
    tops = [r for r in rs if r[4] == "pli_batch_run"]
    if not tops:
        return
    top = tops[-1]
    kids = {}
    for r in rs:
        kids.setdefault(r[1], []).append(r)
    print("\n# the last pli_batch_run call, as nested (host time of the launch calls, not kernel time; kernels of the side stream are")
    print("# enqueued inside the line chain's range — the fork is in runLines)")

    def walk(r, depth):
        print("%s%-*s %10.1f us" % ("  " * depth, 34 - 2 * depth, r[4], (r[3] - r[2]) / 1e3))
        for k in kids.get(r[0], []):
            walk(k, depth + 1)
    walk(top, 0)


if __name__ == "__main__":
    if sys.argv[1] == "markers":
        markers(sys.argv[2])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
