#!/usr/bin/env python3
"""Dev tool (GPU box): full stereo frames of unusual content / sizes / feature budgets against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

def compare(name, cfg, L, R):
    fe = Frontend(cfg)
    rec = fe.batch_run_host(np.stack([L, R])[None])[0]
    fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
    bad = []
    for eye, img, k in ((0, L, "L"), (1, R, "R")):
        n, kp, desc = fr.orb_extract(eye, img)
        if n != len(rec["kp" + k]) or kp.tobytes() != rec["kp" + k].tobytes() or not np.array_equal(desc, rec["desc" + k]):
            bad.append("orb%s(%d vs %d)" % (k, len(rec["kp" + k]), n))
        m, kl, ld = fr.line_extract(eye, img)
        if m != len(rec["kl" + k]) or kl.tobytes() != rec["kl" + k].tobytes() or not np.array_equal(ld, rec["ldesc" + k]):
            bad.append("line%s(%d vs %d)" % (k, len(rec["kl" + k]), m))
    ur, dp, _, _ = fr.stereo_points()
    if ur.tobytes() != rec["uright"].tobytes() or dp.tobytes() != rec["depth"].tobytes():
        bad.append("stereoP")
    disp, le, _ = fr.stereo_lines()
    if disp.tobytes() != rec["disp"].tobytes() or le.tobytes() != rec["le"].tobytes():
        bad.append("stereoL")
    print("%-34s kp %4d/%4d lines %4d/%4d  %s" % (name, len(rec["kpL"]), len(rec["kpR"]), len(rec["klL"]), len(rec["klR"]),
                                                  "OK" if not bad else "MISMATCH " + " ".join(bad)), flush=True)
    return not bad

ok = True
rng = np.random.default_rng(0)
for (W, H) in ((96, 64), (128, 96), (160, 120), (257, 131)):
    L4, R4 = synth.make_stereo_pair(3, 4 * W, 4 * H)
    L, R = np.ascontiguousarray(L4[::4, ::4]), np.ascontiguousarray(R4[::4, ::4])
    ok &= compare("small %dx%d" % (W, H), capi.default_config(W, H, orb_nfeatures=200, lsd_nfeatures=0), L, R)
W, H = 376, 240
L, R = synth.make_stereo_pair(4, W, H)
noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
ok &= compare("noise both eyes", capi.default_config(W, H, orb_nfeatures=1000, lsd_nfeatures=50), noise, noise)
ok &= compare("noise left / scene right", capi.default_config(W, H, orb_nfeatures=1000, lsd_nfeatures=50), noise, R)
ok &= compare("identical eyes (zero disparity)", capi.default_config(W, H, orb_nfeatures=500, lsd_nfeatures=50), L, L)
ok &= compare("tiny feature budget", capi.default_config(W, H, orb_nfeatures=12, lsd_nfeatures=3), L, R)
ok &= compare("huge feature budget", capi.default_config(W, H, orb_nfeatures=4500, lsd_nfeatures=1000), L, R)
sat = np.clip(L.astype(int) * 3 - 200, 0, 255).astype(np.uint8)
ok &= compare("saturated contrast", capi.default_config(W, H, orb_nfeatures=800, lsd_nfeatures=0), sat, sat)
grad = (np.add.outer(np.arange(H), np.arange(W)) % 256).astype(np.uint8)
ok &= compare("sawtooth ramp", capi.default_config(W, H, orb_nfeatures=800, lsd_nfeatures=0), grad, grad)
print("ALL OK" if ok else "SOME MISMATCH")
sys.exit(0 if ok else 1)
