#!/usr/bin/env python3
"""profiles/r03_traffic.json (read back by bench.py for roofline.traffic) from the FETCH_SIZE / WRITE_SIZE passes of
tools/make_profiles_r03.sh:   python tools/make_traffic_json_r03.py gpurun_out/prof_r03 > profiles/r03_traffic.json"""
import json, os, sys

WORKLOADS = {"default_f256": "752x480_F256", "f2048_sequential": "752x480_F2048", "config5_4k": "3840x2160_F16", "config3_720p": "1280x720_F64"}


def rows(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        p = line.split()
        if len(p) >= 6 and p[1] in ("FETCH_SIZE", "WRITE_SIZE") and p[0].startswith("k_"):
            out.setdefault(p[0], {})["fetch_kb" if p[1] == "FETCH_SIZE" else "write_kb"] = float(p[4])
    return {k: v for k, v in out.items() if len(v) == 2}


d = sys.argv[1]
doc = {"note": "HBM traffic per launch from rocprofv3 PMC passes (separate runs for FETCH_SIZE and WRITE_SIZE, bench.py --steps 1 "
               "--warmup 0), per-launch averages per kernel, keyed by workload WxH_F<frames per GPU>.  bytes = (2*FETCH_SIZE_KB + "
               "WRITE_SIZE_KB)*1024: FETCH_SIZE is halved on gfx950 for coalesced streams (MI355X_MICROARCH.md, checked on k_lsd_hist); "
               "for gather kernels the truth lies between 1x and 2x FETCH_SIZE.  Kernel names as rocprofv3 reports them "
               "(k_lsd_grow2_spec is bench.py's k_lsd_grow2; k_lsd_grad64 / k_lsd_blur64 / k_lsd_resize64 are k_lsd_grad / k_blur_lsd / k_resize_lsd).",
       "workloads": {key: rows(os.path.join(d, "pmc_%s.txt" % name)) for name, key in WORKLOADS.items()}}
ALIAS = {"k_lsd_grow2_spec": "k_lsd_grow2", "k_lsd_grow_spec": "k_lsd_grow", "k_lsd_grad64": "k_lsd_grad", "k_lsd_blur64": "k_blur_lsd",
         "k_lsd_resize64": "k_resize_lsd"}
for w in doc["workloads"].values():
    for a, b in ALIAS.items():
        if a in w and b not in w:
            w[b] = w[a]
print(json.dumps(doc, indent=1))
