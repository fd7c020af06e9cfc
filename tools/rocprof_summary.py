#!/usr/bin/env python3
"""Turns rocprofv3's rocpd sqlite output into the text summaries committed under profiles/.

  python tools/rocprof_summary.py stats  <results.db>            kernel-trace --stats table
  python tools/rocprof_summary.py pmc    <results.db> [counter]  per-kernel counter sums (KB for FETCH/WRITE_SIZE)
"""
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
