#!/usr/bin/env python3
"""GPU box: wall time of a Frame constructor through the C++ drop-in (tests/cpp/dropin_harness.cpp): four threads vs four calls."""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cpp_dropin import build_harness, write_input, read_dump
from pli_slam_amd import synth
d = tempfile.mkdtemp()
exe = build_harness(d)
frames = [synth.make_stereo_pair(40 + s, 752, 480, t=t) for s in range(2) for t in range(5)]
for mode in (1, 4, 0):          # four threads; four threads without the pyramid copy-back (ORBextractor::pliCopyPyramidBack(false)); four calls in a row
    write_input(os.path.join(d, "in"), frames, 6, mode)
    r = subprocess.run([exe, os.path.join(d, "in"), os.path.join(d, "out")], capture_output=True, text=True)
    print(r.stdout.strip(), r.stderr.strip()[-300:])
