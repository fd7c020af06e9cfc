#!/bin/bash
# CPU only: the oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer on two full stereo frames.
# (GPU sanitizers are not available on the pool; the host side of the product — adapters, registry — runs under TSAN / ASAN in
#  tests/test_cpp_host.py against tests/cpp/mock_pli.cpp.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -ffp-contract=off -shared -o /tmp/liboracle_asan.so $R/oracle/oracle_capi.cpp
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python3 - <<PY
import sys; sys.path.insert(0, "$R")
from oracle import pyoracle as po
po.build = lambda force=False: "/tmp/liboracle_asan.so"
from pli_slam_amd import synth
for (W, H, seed) in ((376, 240, 3), (752, 480, 0)):
    L, R_ = synth.make_stereo_pair(seed, W, H)
    f = po.Frame(po.default_config(W, H, orb_nfeatures=500, lsd_nfeatures=60))
    f.run(L, R_)
    print(W, H, "clean")
PY
