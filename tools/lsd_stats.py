#!/usr/bin/env python3
"""Dev tool (GPU box, library built with EXTRA=-DLSD_STATS): per-image counts of the sequential LSD grower."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = capi.default_config(752, 480, max_frames=F, lsd_mode=2)
fe = Frontend(cfg)
uniq = np.stack([np.stack(synth.make_stereo_pair(s_, 752, 480)) for s_ in range(min(F, 32))])
frames = uniq[np.arange(F) % len(uniq)]
fe.batch_run_host(frames)
out = (C.c_ulonglong * 24)()
fe.L.pli_lsd_stats(out)
fe.L.pli_lsd_stats_max.restype = C.c_ulonglong
fe.L.pli_lsd_stats_max()
fe.batch_run_host(frames)
fe.L.pli_lsd_stats(out)
mx = fe.L.pli_lsd_stats_max()
n = 2 * F
names = ["regions", "batches", "px in kept regions", "px in all regions", "kept regions", "regions of 1", "regions <= 4", "defined px"]
for k, v in zip(names, out):
    print("%-22s %10.1f per image" % (k, v / n))
print("%-36s %10.1f per image" % ("single-entry steps (queue near full)", out[13] / n))
print("%-36s %10.1f per image in %.1f regions" % ("px in regions past the batch limit", out[14] / n, out[15] / n))
print("%-36s %10.3f Mcycles (mean %.3f)" % ("slowest image wave", mx / 1e6, out[8] / n / 1e6))
tn = ["kernel", "seed batch fetch", "batch: queue read + neighbour load", "batch: accept loop", "region2rect"]
for k, v in zip(tn, list(out)[8:13]):
    print("%-36s %10.3f Mcycles per image (%.1f %%)" % (k, v / n / 1e6, 100.0 * v / max(out[8], 1)))
