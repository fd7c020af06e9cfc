#!/usr/bin/env python3
"""Dev tool (GPU box): wide seed sweep of the round-3 schedules of the tile relaxation against the oracle — the default (late rounds in
the persistent tail kernel), the tail from round 3 on, no tail (planned look-free rounds; the second call on the context plans from
the first), the speculative round 1 (PLI_TX_SPEC=1), tiles of 16 / 32 / 64, region ids as ranks instead of keys (PLI_TX_KEYS=0) — batches of 8 pairs, lsd_nfeatures = 0 (every segment
and its LBD bits are compared), both detector pipelines.      python tools/cross_check_r03.py [first seed] [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

base = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H, B = 752, 480, 8
VARIANTS = [("default", {}), ("tail_t3", {"PLI_TX_TAIL_T0": "3"}), ("no_tail", {"PLI_TX_TAIL": "0"}), ("spec", {"PLI_TX_SPEC": "1"}),
            ("spec_tail_t3_ts32", {"PLI_TX_SPEC": "1", "PLI_TX_TAIL_T0": "3", "PLI_TX_TS": "32"}), ("ts16", {"PLI_TX_TS": "16"}),
            ("ranks", {"PLI_TX_KEYS": "0"}), ("ranks_no_tail_ts32", {"PLI_TX_KEYS": "0", "PLI_TX_TAIL": "0", "PLI_TX_TS": "32"})]
KEYS = sorted({k for _, e in VARIANTS for k in e})
bad = 0
for flags in (None, 0):
    over = {} if flags is None else {"parity_flags": flags}
    cfg0 = capi.default_config(W, H, lsd_nfeatures=0, orb_nfeatures=200, max_frames=B, **over)
    for b0 in range(0, npairs, B):
        with ThreadPoolExecutor(32) as ex:
            pairs = list(ex.map(lambda i: synth.make_stereo_pair(base + b0 + i, W, H), range(B)))

            def oracle(i):
                fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg0)))
                return [fr.line_extract(e, pairs[i][e]) for e in (0, 1)]
            want = list(ex.map(oracle, range(B)))
        imgs = np.stack([np.stack(p) for p in pairs])
        for name, env in VARIANTS:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            fe = Frontend(capi.default_config(W, H, lsd_nfeatures=0, orb_nfeatures=200, max_frames=B, lsd_mode=3, **over))
            for call in range(2):
                recs = fe.batch_run_host(imgs, stages=capi.RUN_LINES)
                for i, rec in enumerate(recs):
                    for e, k in ((0, "L"), (1, "R")):
                        m, kl, ld = want[i][e]
                        if m != len(rec["kl" + k]) or kl.tobytes() != rec["kl" + k].tobytes() or not np.array_equal(ld, rec["ldesc" + k]):
                            bad += 1
                            print("MISMATCH flags", flags, "seed", base + b0 + i, "eye", e, name, "call", call, len(rec["kl" + k]), m, flush=True)
            st = fe.lsd_round_stats()
            if st[2]:
                print("note: device-side fallback used", name, st, flush=True)
            del fe
        print("flags", flags, "seeds", base + b0, "..", base + b0 + B - 1, "done", flush=True)
print("mismatches:", bad)
