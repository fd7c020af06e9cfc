#!/usr/bin/env python3
"""Dev tool (GPU box): line extraction of many seeds / sizes in both LSD schedules against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

bad = 0
for (W, H) in ((752, 480), (640, 480), (1280, 720), (320, 200)):
    fes = {m: Frontend(capi.default_config(W, H, lsd_nfeatures=0, max_frames=1, lsd_mode=m)) for m in (1, 2)}
    fr = po.Frame(po.Config.from_buffer_copy(bytes(fes[1].cfg)))
    base = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    for seed in range(base, base + (6 if W < 1000 else 3)):
        L, R = synth.make_stereo_pair(seed, W, H)
        for img in (L, R):
            m, okl, old = fr.line_extract(0, img)
            for mode, fe in fes.items():
                n, kl, ld = fe.line_extract(0, img)
                ok = n == m and kl.tobytes() == okl.tobytes() and np.array_equal(ld, old)
                if not ok:
                    bad += 1
                    print("MISMATCH", W, H, seed, "mode", mode, n, m, flush=True)
    print("size %dx%d done" % (W, H), flush=True)
print("mismatches:", bad)
