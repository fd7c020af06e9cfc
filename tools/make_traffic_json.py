#!/usr/bin/env python3
"""Builds profiles' traffic json (read back by bench.py for roofline.traffic) from the PMC summaries written by
tools/make_profiles.sh:   python tools/make_traffic_json.py <prof_out dir> <frames of the default run> > r01_traffic.json"""
import json, os, sys


def rows(path, kernels):
    out = {}
    for line in open(path):
        p = line.split()
        if len(p) >= 6 and p[0] in kernels and p[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            out.setdefault(p[0], {})["fetch_kb" if p[1] == "FETCH_SIZE" else "write_kb"] = float(p[4])
    return out


d, frames = sys.argv[1], sys.argv[2]
doc = {"note": "HBM traffic per launch from rocprofv3 PMC passes (separate runs for FETCH_SIZE and WRITE_SIZE) of bench.py, keyed by "
               "frames per GPU and kernel; per-launch averages. bytes = (2*FETCH_SIZE_KB + WRITE_SIZE_KB)*1024: FETCH_SIZE is "
               "halved on gfx950 per MI355X_MICROARCH.md, checked on k_lsd_hist.",
       "workloads": {frames: rows(os.path.join(d, "pmc_fetch_write_f%s.txt" % frames), ("k_lsd_grow2",)),
                     "32": rows(os.path.join(d, "pmc_relaxation_f32.txt"), ("k_rx_grow_big", "k_rx_grow"))}}
print(json.dumps(doc, indent=1))
