#!/usr/bin/env python3
"""Dev tool (GPU box): wall time of the tracking matchers on a 1280x720 frame pair (config 3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
W, H = 1280, 720
cfg = capi.default_config(W, H, orb_nfeatures=2000, lsd_nfeatures=200, max_frames=2)
fe = Frontend(cfg)
f0 = synth.make_stereo_pair(8, W, H, t=0); f1 = synth.make_stereo_pair(8, W, H, t=1)
recs = fe.batch_run_host(np.stack([np.stack(f0), np.stack(f1)]))
last, cur = recs
kpl = last["kpL"]
q = np.zeros(len(kpl), capi.PROJ_QUERY_DT)
q["u"] = kpl["x"] - 3.0; q["v"] = kpl["y"] - 1.0
q["radius"] = 7.0 * (np.float32(1.2) ** kpl["octave"]).astype(np.float32)
q["ur"] = q["u"]; q["min_level"] = kpl["octave"] - 1; q["max_level"] = kpl["octave"] + 1
q["angle"] = kpl["angle"]; q["valid"] = 1
b = (0.0, float(W), 0.0, float(H))
def tm(f, n=5):
    f(); t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n * 1e3
print("queries %d keypoints %d" % (len(q), len(cur["kpL"])))
print("search_by_projection  %.2f ms" % tm(lambda: fe.search_by_projection(q, last["descL"], cur["kpL"], cur["descL"], cur["uright"], b, True)))
print("search_local_map      %.2f ms" % tm(lambda: fe.search_local_map(q, last["descL"], cur["kpL"], cur["descL"], cur["uright"], b, 0.8)))
print("match lines (mutual)  %.2f ms" % tm(lambda: fe.match(last["ldescL"], cur["ldescL"], 0.9)))
print("knn2 2000x2000        %.2f ms" % tm(lambda: fe.knn2(last["descL"], cur["descL"])))
