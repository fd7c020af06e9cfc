import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernel_ms_per_step"]
print(sys.argv[1], round(d["value"],1), "f/s", round(d["ms_per_step"],3), "ms", d.get("lsd_rounds",{}).get("needed_by_slowest_image"), {n:round(v,3) for n,v in list(k.items())[:9]}, "single", d.get("single_pair"), "host", (d.get("host_inclusive") or {}).get("value"))
