#!/usr/bin/env python3
"""Reads one bench.py JSON line (run with PLI_RX_PROFROUNDS=1) on stdin and prints the per-round
kernel times of the LSD relaxation."""
import collections, json, sys
d = json.loads(sys.stdin.read())
k = d["roofline"]["kernel_ms_per_step"]
rounds = collections.defaultdict(dict)
other = {}
for n, v in k.items():
    if "@" in n:
        a, b = n.split("@"); rounds[int(b)][a[5:]] = v
    else:
        other[n] = v
print("value %.1f %s, %.2f ms/step" % (d["value"], d["unit"], d["ms_per_step"]))
tot = collections.Counter()
for t in sorted(rounds):
    print(t, rounds[t], round(sum(rounds[t].values()), 3))
    tot.update(rounds[t])
print("relaxation totals", {a: round(b, 3) for a, b in tot.most_common()}, round(sum(tot.values()), 3))
print("other", {a: b for a, b in sorted(other.items(), key=lambda x: -x[1])[:8]}, round(sum(other.values()), 3))
