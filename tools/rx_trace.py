#!/usr/bin/env python3
"""Dev tool for the GPU box: one line_extract in relaxation mode with the per-round trace
(PLI_RX_TRACE=1) and a comparison with the oracle.  python tools/rx_trace.py [W H seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PLI_RX_TRACE", "1")
import numpy as np
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

W = int(sys.argv[1]) if len(sys.argv) > 1 else 752
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L, R = synth.make_stereo_pair(SEED, W, H)
cfg = capi.default_config(W, H, lsd_nfeatures=0, max_frames=1, lsd_mode=int(os.environ.get("RX_MODE", "1")))
fe = Frontend(cfg)
fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
t0 = time.time()
n, kl, ld = fe.line_extract(0, L)
print("gpu lines", n, "%.1f ms" % ((time.time() - t0) * 1e3), flush=True)
t0 = time.time()
n, kl, ld = fe.line_extract(0, L)
print("gpu lines (2nd call)", n, "%.1f ms" % ((time.time() - t0) * 1e3), flush=True)
m, okl, old = fr.line_extract(0, L)
print("oracle lines", m, "equal:", n == m and kl.tobytes() == okl.tobytes() and np.array_equal(ld, old))
