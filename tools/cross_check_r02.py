#!/usr/bin/env python3
"""Dev tool (GPU box): one-off wide seed sweep of the round-2 growers against the oracle — every LSD schedule (1 relaxation,
2 speculative sequential, 3 tile-sequential with 64- and 128-pixel tiles), both detector pipelines, batches of 8 pairs
(lsd_nfeatures = 0: every segment and its LBD bits are compared).   python tools/cross_check_r02.py [first seed] [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

base = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H, B = 752, 480, 8
bad = 0
for flags in (None, 0):
    over = {} if flags is None else {"parity_flags": flags}
    cfg0 = capi.default_config(W, H, lsd_nfeatures=0, orb_nfeatures=200, max_frames=B, **over)
    for b0 in range(0, npairs, B):
        pairs = [synth.make_stereo_pair(base + b0 + i, W, H) for i in range(B)]

        def oracle(i):
            fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg0)))
            return [fr.line_extract(e, pairs[i][e]) for e in (0, 1)]
        with ThreadPoolExecutor(8) as ex:
            want = list(ex.map(oracle, range(B)))
        for mode, ts in ((1, None), (2, None), (3, "64"), (3, "128")):
            if ts:
                os.environ["PLI_TX_TS"] = ts
            else:
                os.environ.pop("PLI_TX_TS", None)
            fe = Frontend(capi.default_config(W, H, lsd_nfeatures=0, orb_nfeatures=200, max_frames=B, lsd_mode=mode, **over))
            recs = fe.batch_run_host(np.stack([np.stack(p) for p in pairs]), stages=capi.RUN_LINES)
            for i, rec in enumerate(recs):
                for e, k in ((0, "L"), (1, "R")):
                    m, kl, ld = want[i][e]
                    if m != len(rec["kl" + k]) or kl.tobytes() != rec["kl" + k].tobytes() or not np.array_equal(ld, rec["ldesc" + k]):
                        bad += 1
                        print("MISMATCH flags", flags, "seed", base + b0 + i, "eye", e, "mode", mode, ts, len(rec["kl" + k]), m, flush=True)
            del fe
        print("flags", flags, "seeds", base + b0, "..", base + b0 + B - 1, "done", flush=True)
print("mismatches:", bad)
