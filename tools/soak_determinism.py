#!/usr/bin/env python3
"""Dev tool (GPU box): the same batch many times; every result table must be byte-identical to the first one (the
relaxation's claims race by design, its result must not).   python tools/soak_determinism.py [frames] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H = 752, 480
uniq = np.stack([np.stack(synth.make_stereo_pair(200 + s, W, H)) for s in range(min(F, 16))])
imgs = uniq[np.arange(F) % len(uniq)]
left, right = np.ascontiguousarray(imgs[:, 0]), np.ascontiguousarray(imgs[:, 1])
fe = Frontend(capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=F))
ref = None
bad = 0
for it in range(N):
    table = np.zeros(fe.table_bytes(F), np.uint8)
    capi.check(fe.L.pli_batch_run_host(fe.h, F, capi.ptr(left), capi.ptr(right), W, W * H, capi.RUN_ALL, capi.ptr(table)))
    if ref is None:
        ref = table
    elif not np.array_equal(ref, table):
        bad += 1
        print("run %d differs in %d bytes" % (it, int((ref != table).sum())), flush=True)
print("F=%d: %d runs, %d differ from the first" % (F, N, bad))
sys.exit(1 if bad else 0)
