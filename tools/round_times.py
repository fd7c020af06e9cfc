"""Per-round kernel times of one bench line (PLI_RX_PROFROUNDS=1): python tools/round_times.py <bench json> [kernel name prefix ...]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_step"]
want = sys.argv[2:]
print(round(d["value"], 1) if d["value"] is not None else "PARITY-FAIL", "f/s", round(d["ms_per_step"], 3), "ms")
for n in sorted(k, key=lambda n: (n.split("@")[-1] if "@" in n else "00", n)):
    if not want or any(n.startswith(w) for w in want):
        print(" ", n, k[n])
