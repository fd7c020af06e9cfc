#!/usr/bin/env python3
"""Dev tool (GPU box; library built with EXTRA=-DLSD_STATS, see tools/build_stats.sh; PLI_LIB_PATH=build/stats/libpli_frontend.so
PLI_LSD_SPEC=1): phase cycles and counts of the speculative sequential LSD grower, per image."""
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
names = {0: "super-rows", 1: "wave-wide regions", 2: "  of them taken over from a lane", 3: "clean lanes", 4: "lanes that lost a tag", 5: "lanes that hit the cap",
         6: "todo lanes whose seed was taken", 7: "px accepted wave-wide", 14: "speculative steps", 15: "wave-wide batches"}
for k in sorted(names):
    print("%-36s %10.1f per image" % (names[k], out[k] / n))
print("%-36s %10.3f Mcycles (mean %.3f)" % ("slowest image wave", mx / 1e6, out[8] / n / 1e6))
for k, nme in ((9, "fill"), (10, "speculation"), (11, "validation"), (12, "resolution (incl. region2rect)"), (13, "region2rect")):
    print("%-36s %10.3f Mcycles per image (%.1f %%)" % (nme, out[k] / n / 1e6, 100.0 * out[k] / max(out[8], 1)))
for k, nme in ((16, "batch fetch (issue -> data)"), (17, "batch accept loop")):
    print("%-36s %10.3f Mcycles per image (%.1f %%)" % (nme, out[k] / n / 1e6, 100.0 * out[k] / max(out[8], 1)))
print("%-36s %10.1f per image" % ("entries popped by batches", out[18] / n))
print("%-36s %10.1f per image" % ("accept-loop iterations", out[19] / n))
