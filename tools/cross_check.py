#!/usr/bin/env python3
"""Dev tool (GPU box): wide seed sweeps of the LSD schedules against the oracle, lsd_nfeatures = 0 (every segment and its LBD
bits are compared), both detector pipelines (parity_flags default and 0), batches of 8 pairs of 752x480.

  python tools/cross_check.py [--set schedules|tile|sizes] [first seed] [pairs]
    schedules  lsd_mode 1 / 2 / 3 with 64- and 128-pixel tiles                                   (the round-2 sweep)
    tile       the tile relaxation's variants: tail kernel from round 8 / 3 / none, speculative round 1, tiles of 16 / 32,
               region ids as ranks instead of keys, round 1's owner word in the owner plane / written by the sort / lazy with every
               unclaimed pixel through the exact test; two calls per context (the second plans from the first)  (the round-3 sweep)
    sizes      four image sizes, lsd_mode 1 and 2, one image per call                              (the round-1 sweep)
    hot        round 1 of the tile relaxation on the 8-byte hot records (round 6): key / rank mode, wide filter margin, exact rect sums"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

argv = sys.argv[1:]
which = "tile"
if argv and argv[0] == "--set":
    which, argv = argv[1], argv[2:]
base = int(argv[0]) if argv else 70000
npairs = int(argv[1]) if len(argv) > 1 else 64
SETS = {
    "schedules": [("mode1", {"lsd_mode": 1}, {}), ("mode2", {"lsd_mode": 2}, {}), ("mode3_ts64", {"lsd_mode": 3}, {"PLI_TX_TS": "64"}),
                  ("mode3_ts128", {"lsd_mode": 3}, {"PLI_TX_TS": "128"})],
    "tile": [(n, {"lsd_mode": 3}, e) for n, e in (
        ("default", {}), ("tail_t3", {"PLI_TX_TAIL_T0": "3"}), ("no_tail", {"PLI_TX_TAIL": "0"}), ("spec", {"PLI_TX_SPEC": "1"}),
        ("spec_tail_t3_ts32", {"PLI_TX_SPEC": "1", "PLI_TX_TAIL_T0": "3", "PLI_TX_TS": "32"}), ("ts16", {"PLI_TX_TS": "16"}),
        ("ranks", {"PLI_TX_KEYS": "0"}), ("ranks_no_tail_ts32", {"PLI_TX_KEYS": "0", "PLI_TX_TAIL": "0", "PLI_TX_TS": "32"}),
        ("owner_plane", {"PLI_TX_PACK1": "0"}), ("sort_written", {"PLI_TX_PACK1": "1"}), ("lazy_all_exact", {"PLI_TX_PACK1": "2", "PLI_TX_LAZY_MARGIN": "2000000"}),
        ("lazy_all_exact_rec16", {"PLI_TX_HOT": "0", "PLI_TX_LAZY_MARGIN": "2000000"}),
        ("block_rounds", {"PLI_TX_CELLS": "0"}), ("cells_no_tail", {"PLI_TX_CELLS": "1", "PLI_TX_TAIL": "0"}))],
    # round 6: round 1 on the 8-byte hot records against the 16-byte ones, key and rank mode, a wide filter margin (the exact sums
    # folded in mid-growth), region2rect's exact sums for every region
    "hot": [(n, {"lsd_mode": 3}, e) for n, e in (
        ("rec16", {"PLI_TX_HOT": "0"}), ("hot_round1_only", {"PLI_TX_HOT": "1"}), ("hot", {"PLI_TX_HOT": "2"}), ("hot_ranks", {"PLI_TX_HOT": "2", "PLI_TX_KEYS": "0"}),
        ("hot_margin2", {"PLI_TX_HOT": "2", "PLI_ALIGN_MARGIN_DEG": "2"}), ("hot_margin2_all_exact", {"PLI_TX_HOT": "2", "PLI_ALIGN_MARGIN_DEG": "2", "PLI_TX_HOT_BAND2": "10"}),
        ("hot_rect_exact", {"PLI_TX_HOT": "2", "PLI_RECT_APPROX_BAND": "10"}), ("hot_ts32_no_tail", {"PLI_TX_HOT": "2", "PLI_TX_TS": "32", "PLI_TX_TAIL": "0"}),
        ("hot_tail_t3", {"PLI_TX_HOT": "2", "PLI_TX_TAIL_T0": "3"}), ("hot_block_rounds", {"PLI_TX_HOT": "2", "PLI_TX_CELLS": "0"}),
        ("hot_lazy_ids", {"PLI_TX_HOT": "2", "PLI_TX_PACK1": "2"}), ("hot_lazy_ids_margin2", {"PLI_TX_HOT": "2", "PLI_TX_PACK1": "2", "PLI_ALIGN_MARGIN_DEG": "2"}))],
}
bad = 0
if which == "sizes":
    for (W, H) in ((752, 480), (640, 480), (1280, 720), (320, 200)):
        fes = {m: Frontend(capi.default_config(W, H, lsd_nfeatures=0, max_frames=1, lsd_mode=m)) for m in (1, 2)}
        fr = po.Frame(po.Config.from_buffer_copy(bytes(fes[1].cfg)))
        for seed in range(base, base + (6 if W < 1000 else 3)):
            for img in synth.make_stereo_pair(seed, W, H):
                m, okl, old = fr.line_extract(0, img)
                for mode, fe in fes.items():
                    n, kl, ld = fe.line_extract(0, img)
                    if not (n == m and kl.tobytes() == okl.tobytes() and np.array_equal(ld, old)):
                        bad += 1
                        print("MISMATCH", W, H, seed, "mode", mode, n, m, flush=True)
        print("size %dx%d done" % (W, H), flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
VARIANTS = SETS[which]
KEYS = sorted({k for _, _, e in VARIANTS for k in e})
W, H, B = 752, 480, 8
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
        for name, cfgover, env in VARIANTS:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            fe = Frontend(capi.default_config(W, H, lsd_nfeatures=0, orb_nfeatures=200, max_frames=B, **cfgover, **over))
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
sys.exit(1 if bad else 0)
