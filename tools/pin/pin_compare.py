#!/usr/bin/env python3
"""Compares what a REAL OpenCV returned (tests/golden/opencv_pin/*.npy, written by tools/pin/pin_against_opencv.cpp on the machine
that has OpenCV) with the oracle's restatement, primitive by primitive, and reports which PLI_PARITY_* flags reproduce the real
library.  Exit code 0 = every primitive pinned (for LSD: some flag set matches).  Also run by tests/test_opencv_pin.py."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle as po


def compare(d):
    res = {}
    man = open(os.path.join(d, "manifest.txt")).read().splitlines()
    res["manifest"] = man[:3]
    ld = lambda n: np.load(os.path.join(d, n))
    # libm: is this machine's cosf / sinf the glibc >= 2.28 algorithm the TRIG_F32 switches restate?
    xs, c, s = ld("libm_x.npy"), ld("libm_cosf.npy"), ld("libm_sinf.npy")
    mine_c = np.array([po.glibc_cosf(float(x)) for x in xs], np.float32)
    mine_s = np.array([po.glibc_sinf(float(x)) for x in xs], np.float32)
    new = np.array_equal(mine_c, c) and np.array_equal(mine_s, s)
    cr = (np.array_equal(np.cos(xs.astype(np.float64)).astype(np.float32), c) and
          np.array_equal(np.sin(xs.astype(np.float64)).astype(np.float32), s))
    # informational: which cosf the float-overload call sites get on that machine
    res["libm cosf/sinf of that machine"] = ("the glibc >= 2.28 algorithm (PLI_PARITY_TRIG_F32_* = 1 reproduces it)" if new else
                                             "correctly rounded on the sample" if cr else
                                             "an older libm: neither setting of PLI_PARITY_TRIG_F32_* reproduces its cosf bit for bit")
    yx, a = ld("fastatan2_yx.npy"), ld("fastatan2.npy")
    res["fastAtan2"] = bool(np.array_equal(np.array([po.fast_atan2(float(y), float(x)) for y, x in yx], np.float32), a))
    for line in man:
        if not line.startswith("image "):
            continue
        _, name, w, h = line.split()
        w, h = int(w), int(h)
        img = np.fromfile(os.path.join(d, "inputs", name + ".raw"), np.uint8).reshape(h, w)
        g = lambda k: ld("%s_%s.npy" % (name, k))
        r1 = g("resize_level1")
        res[name + " resize"] = bool(np.array_equal(po.resize(img, r1.shape[1], r1.shape[0], w / r1.shape[1], h / r1.shape[0]), r1))
        res[name + " GaussianBlur 7x7 s2"] = bool(np.array_equal(po.gaussian_blur(img, 7, 2.0), g("blur7_s2")))
        b5 = po.gaussian_blur(img, 5, 1.0)
        res[name + " GaussianBlur 5x5 s1"] = bool(np.array_equal(b5, g("blur5_s1")))
        dx, dy = po.sobel(g("blur5_s1"))
        res[name + " Sobel"] = bool(np.array_equal(dx, g("sobel_dx")) and np.array_equal(dy, g("sobel_dy")))
        for th in (20, 7):
            res[name + " FAST t=%d" % th] = bool(np.array_equal(po.fast_image(img, th), g("fast_t%d" % th)))
        want = g("lsd_segments")
        hits = []
        for flags in (0, po.PARITY_TRIG_F32_LSD, po.PARITY_LSD_F64, po.PARITY_LSD_F64 | po.PARITY_TRIG_F32_LSD):
            fr = po.Frame(po.default_config(w, h, lsd_nfeatures=0, parity_flags=flags))
            fr.line_extract(0, img)
            seg = fr.lsd_segments(0)
            exact = seg.shape == want.shape and np.array_equal(seg, want)
            close = seg.shape == want.shape and np.abs(seg - want).max(initial=0) <= 0.5
            hits.append((flags, "exact" if exact else "within 0.5 px" if close else "%d vs %d segments" % (len(seg), len(want))))
        res[name + " LSD per parity flags (0 u8, 2 u8+cosf, 8 f64, 10 f64+cosf)"] = hits
        # stage 2 (tools/pin/pin_reference_extractors.cpp): the reference's own ORBextractor / Lineextractor on the same image — pins
        # the quadtree's tie-break, the LSD seed order and the order of equal responses, which no primitive shows
        if os.path.exists(os.path.join(d, name + "_ref_orb_kp_f.npy")):
            rkf, rki, rd = g("ref_orb_kp_f"), g("ref_orb_kp_i"), g("ref_orb_desc")
            orb_hits = []
            for flags in (po.PARITY_LSD_F64, po.PARITY_LSD_F64 | po.PARITY_TRIG_F32_ORB):
                fr = po.Frame(po.default_config(w, h, orb_nfeatures=1200, lsd_nfeatures=0, parity_flags=flags))
                n, kp, desc = fr.orb_extract(0, img)
                kf = np.stack([kp["x"], kp["y"], kp["size"], kp["angle"], kp["response"]], axis=1).astype(np.float32)
                same_set = n == len(rkf) and np.array_equal(kf[:, :3], rkf[:, :3]) and np.array_equal(kp["octave"], rki[:, 0])
                exact = same_set and np.array_equal(kf, rkf) and np.array_equal(desc, rd)
                orb_hits.append((flags, "exact" if exact else "same keypoints, angles / bits differ" if same_set else
                                 "%d vs %d keypoints (quadtree order or FAST differs)" % (n, len(rkf))))
            res[name + " reference ORBextractor per parity flags (8 cos double, 9 cosf)"] = orb_hits
            rlf, rli, rld = g("ref_kl_f"), g("ref_kl_i"), g("ref_kl_desc")
            kl_hits = []
            for flags in (0, po.PARITY_TRIG_F32_LBD, po.PARITY_LSD_F64, po.PARITY_LSD_F64 | po.PARITY_TRIG_F32_LBD, 14):
                fr = po.Frame(po.default_config(w, h, lsd_nfeatures=0, parity_flags=flags))
                m, kl, ld = fr.line_extract(0, img)
                KF = ("angle", "pt_x", "pt_y", "response", "size", "startPointX", "startPointY", "endPointX", "endPointY", "sPointInOctaveX",
                      "sPointInOctaveY", "ePointInOctaveX", "ePointInOctaveY", "lineLength")
                lf = np.stack([kl[k_] for k_ in KF], axis=1).astype(np.float32) if m else np.zeros((0, 14), np.float32)
                exact = m == len(rlf) and np.array_equal(lf, rlf) and np.array_equal(kl["class_id"], rli[:, 0]) and np.array_equal(ld, rld)
                close = m == len(rlf) and np.abs(lf[:, 5:9] - rlf[:, 5:9]).max(initial=0) <= 0.5
                kl_hits.append((flags, "exact" if exact else "end points within 0.5 px, another field differs" if close else
                                "%d vs %d key lines" % (m, len(rlf))))
            res[name + " reference Lineextractor per parity flags"] = kl_hits
    return res


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "opencv_pin")
    out = compare(d)
    bad = 0
    for k, v in out.items():
        print("%-70s %s" % (k, v))
        if v is False:
            bad += 1
        if isinstance(v, list) and v and isinstance(v[0], tuple) and not any(h[1] == "exact" for h in v):
            bad += 1
    print("PINNED" if bad == 0 else "%d primitive(s) NOT reproduced: the oracle's restatement of those differs from this OpenCV" % bad)
    sys.exit(1 if bad else 0)
