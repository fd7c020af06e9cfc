import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
for (W, H) in ((96, 64), (160, 120), (128, 96)):
    L4, R4 = synth.make_stereo_pair(3, 4 * W, 4 * H)
    L = np.ascontiguousarray(L4[::4, ::4])
    cfg = capi.default_config(W, H, orb_nfeatures=200, lsd_nfeatures=0, lsd_mode=2)
    fe = Frontend(cfg)
    n, kl, ld = fe.line_extract(0, L)
    raw = fe.debug_fetch(0, capi.DBG_LSD_SEGMENTS)
    ns = int(raw[:4].view(np.int32)[0]); segs = raw[4:4 + 16 * ns].view(np.float32).reshape(-1, 4)
    fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
    fr.line_extract(0, L)
    osegs = fr.lsd_segments(0)
    print(W, H, "gpu segs", len(segs), "oracle", len(osegs))
    k = 0
    while k < min(len(segs), len(osegs)) and np.array_equal(segs[k], osegs[k]): k += 1
    print("  first difference at", k, segs[k] if k < len(segs) else None, osegs[k] if k < len(osegs) else None)
