#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer boundary (pli_batch_run_host): H2D images + all kernels + D2H tables."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
imgs = synth.make_batch(min(F, 16), 752, 480)
imgs = imgs[np.arange(F) % imgs.shape[0]]
fe = Frontend(capi.default_config(752, 480, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=F))
fe.batch_run_host(imgs)
t = time.perf_counter(); n = 3
for _ in range(n): fe.batch_run_host(imgs)
dt = (time.perf_counter() - t) / n
print("F=%d host-buffer boundary: %.1f ms/step, %.1f stereo frames/s (pageable host memory, includes parsing)" % (F, dt * 1e3, F / dt))
