#!/usr/bin/env python3
"""profiles/rNN_traffic.json (read back by bench.py for roofline.traffic and roofline.issue) from the counter passes of
tools/make_profiles.sh --round N:   python tools/make_traffic_json.py gpurun_out/prof_rNN > profiles/rNN_traffic.json

  workloads.<WxH_Fn>.<kernel>  = {fetch_kb, write_kb}   per-launch averages of the FETCH_SIZE / WRITE_SIZE passes
  issue.<WxH_Fn>.<kernel>      = {valu, salu, active_valu_quad_cycles, avg_ns, launches}  from the SQ passes (per launch)"""
import json, os, sys

WORKLOADS = {"default_f256": "752x480_F256", "f2048_sequential": "752x480_F2048", "config5_4k": "3840x2160_F16", "config3_720p": "1280x720_F64"}
ALIAS = {"k_lsd_grow2_spec": "k_lsd_grow2", "k_lsd_grow_spec": "k_lsd_grow", "k_lsd_grad64": "k_lsd_grad", "k_lsd_blur64": "k_blur_lsd",
         "k_lsd_resize64": "k_resize_lsd", "k_lsd_front64": "k_lsd_front",
         # round 5: round 1 of the tile relaxation has three instances (lazy ids / sort-written ids / owner plane) and the later
         # rounds' bookkeeping two forms; bench.py's profile names are the left-hand kernels' roles
         "k_tx_grow_p2": "k_tx_grow", "k_tx_grow_p1": "k_tx_grow", "k_tx_diffmark_cells": "k_tx_diffmark", "k_tx_prep_cells": "k_tx_prep",
         # round 6: round 1 on the 8-byte hot records (key / rank mode), the later rounds on the cold pair
         "k_tx_grow_h2": "k_tx_grow", "k_tx_grow_h1": "k_tx_grow", "k_tx_grow_sparse_h": "k_tx_grow_sparse"}


def rows(path, counters):
    """kernel -> counter -> (launches, avg per launch, avg ns) of a rocprof_summary.py pmc file"""
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        p = line.split()
        if len(p) >= 6 and p[1] in counters and p[0].startswith("k_"):
            out.setdefault(p[0], {})[p[1]] = (int(p[2]), float(p[4]), float(p[5]))
    return out


def alias(w):
    for a, b in ALIAS.items():
        if a in w and b not in w:
            w[b] = dict(w[a], alias_of=a)           # (bench.py looks kernels up by its own profile names; sums skip the aliases)
    return w


d = sys.argv[1]
doc = {"note": "HBM traffic per launch from rocprofv3 PMC passes (separate runs for FETCH_SIZE and WRITE_SIZE, bench.py --steps 1 "
               "--warmup 0), per-launch averages per kernel, keyed by workload WxH_F<frames per GPU>.  bytes = (2*FETCH_SIZE_KB + "
               "WRITE_SIZE_KB)*1024 for EVERY kernel: FETCH_SIZE tallies a read request as 64 bytes on gfx950 (MI355X_MICROARCH.md) and every "
               "request of these kernels is a 128-byte line, the growers' random 16-byte gathers included (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ; "
               "profiles/r05_counter_calibration.txt; the files of rounds 2-4 said x 1 for the growers, which understated them).  "
               "`issue`: SQ passes of the same workload — wave instructions per launch (SQ_INSTS_VALU / _SALU) and SQ_ACTIVE_INST_VALU "
               "(quad-cycles in which a SIMD's VALU was executing): valu_issue_frac = active_valu_quad_cycles * 4 / (1024 SIMDs x cycles "
               "of the launch at 2.4 GHz).  Kernel names as rocprofv3 reports them.",
       "workloads": {}, "issue": {}}
for name, key in WORKLOADS.items():
    r = rows(os.path.join(d, "pmc_%s.txt" % name), ("FETCH_SIZE", "WRITE_SIZE"))
    doc["workloads"][key] = alias({k: {"fetch_kb": v["FETCH_SIZE"][1], "write_kb": v["WRITE_SIZE"][1], "launches": v["FETCH_SIZE"][0]}
                                   for k, v in r.items() if len(v) == 2})
    s = rows(os.path.join(d, "pmc_sq_%s.txt" % name), ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU"))
    if s:
        doc["issue"][key] = alias({k: {"valu": v["SQ_INSTS_VALU"][1], "salu": v["SQ_INSTS_SALU"][1],
                                       "active_valu_quad_cycles": v["SQ_ACTIVE_INST_VALU"][1], "avg_ns": v["SQ_ACTIVE_INST_VALU"][2],
                                       "launches": v["SQ_ACTIVE_INST_VALU"][0]}
                                   for k, v in s.items() if len(v) == 3})
print(json.dumps(doc, indent=1))
