#!/usr/bin/env python3
"""Dev tool (GPU box): the hot-record levels under a few schedules on a handful of pairs — rounds needed, fallbacks, bytes against level 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
W, H, B = 752, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 4
pairs = [synth.make_stereo_pair(700 + i, W, H) for i in range(B)]
imgs = np.stack([np.stack(p) for p in pairs])
KEYS = ("PLI_TX_HOT", "PLI_TX_TS", "PLI_TX_TAIL", "PLI_TX_CELLS", "PLI_RECT_ASIDE", "PLI_TX_TAIL_T0", "PLI_RX_MAXROUNDS")
ref = None
for env in ({"PLI_TX_HOT": "0", "PLI_TX_TS": "64"}, {"PLI_TX_HOT": "1", "PLI_TX_TS": "64"}, {"PLI_TX_HOT": "2", "PLI_TX_TS": "64"},
            {"PLI_TX_HOT": "2", "PLI_TX_TS": "64", "PLI_TX_TAIL": "0"}, {"PLI_TX_HOT": "2", "PLI_TX_TS": "64", "PLI_TX_CELLS": "1"},
            {"PLI_TX_HOT": "2", "PLI_TX_TS": "64", "PLI_TX_CELLS": "1", "PLI_TX_TAIL": "0"}, {"PLI_TX_HOT": "2", "PLI_TX_TS": "32"},
            {"PLI_TX_HOT": "2", "PLI_TX_TS": "64", "PLI_TX_TAIL_T0": "3"}):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    fe = Frontend(capi.default_config(W, H, lsd_nfeatures=0, orb_nfeatures=200, max_frames=B), dev=True)
    recs = fe.batch_run_host(imgs, stages=capi.RUN_LINES)
    recs = fe.batch_run_host(imgs, stages=capi.RUN_LINES)
    st = fe.lsd_round_stats()
    sig = b"".join(r["klL"].tobytes() + r["klR"].tobytes() for r in recs)
    if ref is None:
        ref = sig
    print(env, "stats", st, "lines", sum(len(r["klL"]) for r in recs), "same as level 0:", sig == ref, flush=True)
