#!/usr/bin/env python3
"""Writes the inputs of the OpenCV pin (tools/pin/pin_against_opencv.cpp, pin_reference_extractors.cpp) as raw u8 files + inputs.txt:
seeded synthetic scenes, hostile patterns, and (round 4) the shapes the test-suite gained — a KITTI-sized frame (1241x376, the
reference's second stereo example) and real photographs (tests/golden/real/photos.npz: a Middlebury stereo eye, a natural scene, print)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from pli_slam_amd import synth


def inputs():
    L, R = synth.make_stereo_pair(0, 752, 480)
    yield "euroc_s0_left", L
    L, R = synth.make_stereo_pair(3, 376, 240)
    yield "small_s3_left", L
    rng = np.random.default_rng(5)
    yield "noise", rng.integers(0, 256, (240, 376), dtype=np.uint8)
    yy, xx = np.mgrid[0:240, 0:376]
    yield "checker", (((xx // 16) + (yy // 16)) % 2 * 200 + 20).astype(np.uint8)
    yield "odd_643x481", synth.make_stereo_pair(5, 643, 481)[0]
    yield "kitti_1241x376", synth.make_stereo_pair(900, 1241, 376)[0]
    from pli_slam_amd import realdata
    ph = realdata.photos()
    for name in ("motorcycle_left", "camera", "text"):
        yield "photo_" + name, ph[name]


if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "inputs.txt"), "w") as f:
        for name, img in inputs():
            img.tofile(os.path.join(out, name + ".raw"))
            f.write("%s %d %d\n" % (name, img.shape[1], img.shape[0]))
    print("inputs written to", out)
