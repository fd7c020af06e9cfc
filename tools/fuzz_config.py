#!/usr/bin/env python3
"""Dev tool (GPU box): extreme configurations — each must either be refused with an error status or agree with the
oracle; never crash."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

W, H = 376, 240
L, R = synth.make_stereo_pair(4, W, H)

FAILED = []


def run(name, **over):
    try:
        base = dict(orb_nfeatures=400, lsd_nfeatures=0); base.update(over)
        cfg = capi.default_config(W, H, **base)
        fe = Frontend(cfg)
    except capi.PliError as e:
        print("%-44s refused: %s" % (name, e), flush=True)
        return
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
    if ur.tobytes() != rec["uright"].tobytes():
        bad.append("stereoP")
    disp, le, _ = fr.stereo_lines()
    if disp.tobytes() != rec["disp"].tobytes():
        bad.append("stereoL")
    print("%-44s kp %4d lines %4d  %s" % (name, len(rec["kpL"]), len(rec["klL"]), "OK" if not bad else "MISMATCH " + " ".join(bad)), flush=True)
    if bad:
        FAILED.append(name)

run("one pyramid level", orb_nlevels=1)
run("two levels, factor 2.0", orb_nlevels=2, orb_scale_factor=2.0)
run("12 levels, factor 1.1", orb_nlevels=12, orb_scale_factor=1.1)
run("FAST thresholds 5 / 2", orb_ini_th_fast=5, orb_min_th_fast=2)
run("FAST thresholds 80 / 40", orb_ini_th_fast=80, orb_min_th_fast=40)
run("LSD 64 bins", lsd_n_bins=64)
run("LSD 4096 bins", lsd_n_bins=4096)
run("LSD scale 0.5", lsd_scale=0.5)
run("LSD scale 2.0", lsd_scale=2.0)
run("LSD angle tolerance 5 deg", lsd_ang_th=5.0)
run("LSD angle tolerance 60 deg", lsd_ang_th=60.0)
for ang in (1.0, 5.0, 22.5, 45.0, 60.0, 85.0, 86.5, 120.0):       # sequential waves: vector-form alignment test; >= 85.9 deg: exact path only
    run("sequential waves, angle tolerance %g deg" % ang, lsd_ang_th=ang, lsd_mode=2)
run("LSD quant 0.5", lsd_quant=0.5)
run("LSD quant 8", lsd_quant=8.0)
run("min line length 0.3", min_line_length=0.3)
run("matching window 0", matching_s_ws=0)
run("matching window 40", matching_s_ws=40)
run("tiny bf", bf=1.0)
run("refine = 1 (unsupported)", lsd_refine=1)
run("zero levels", orb_nlevels=0)
run("negative features", orb_nfeatures=-5)
print("done: %d configurations mismatch%s" % (len(FAILED), (" " + repr(FAILED)) if FAILED else ""))
sys.exit(1 if FAILED else 0)
