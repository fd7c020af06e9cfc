#!/usr/bin/env python3
"""Dump seeded synthetic images as raw u8 files for the C++ simulators in this directory: <out>_<W>x<H>.raw"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pli_slam_amd import synth
seed, W, H, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
L, R = synth.make_stereo_pair(seed, W, H)
L.tofile(out)
print(out, L.shape)
