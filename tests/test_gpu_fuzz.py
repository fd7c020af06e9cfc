"""Seeded fuzzers of the HIP front-end against the oracle (formerly tools/fuzz_frames.py, fuzz_config.py,
soak_determinism.py): seed sweeps at EuRoC size over every LSD schedule, unusual image content / sizes / feature budgets,
extreme configurations (refused with an error status or equal to the oracle, never a crash), and a determinism soak
(the relaxations' claims race by design, their results must not)."""
from concurrent.futures import ThreadPoolExecutor
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from pli_slam_amd import capi, synth
    from pli_slam_amd.frontend import Frontend
    from oracle import pyoracle as po

    class G:
        pass
    g = G()
    g.capi, g.synth, g.Frontend, g.po = capi, synth, Frontend, po
    return g


def ocfg(g, cfg):
    return g.po.Config.from_buffer_copy(bytes(cfg))


def frame_mismatches(g, cfg, rec, L, R, lines_only=False):
    fr = g.po.Frame(ocfg(g, cfg))
    bad = []
    for eye, img, k in ((0, L, "L"), (1, R, "R")):
        if not lines_only:
            n, kp, desc = fr.orb_extract(eye, img)
            if n != len(rec["kp" + k]) or kp.tobytes() != rec["kp" + k].tobytes() or not np.array_equal(desc, rec["desc" + k]):
                bad.append("orb%s(%d vs %d)" % (k, len(rec["kp" + k]), n))
        m, kl, ld = fr.line_extract(eye, img)
        if m != len(rec["kl" + k]) or kl.tobytes() != rec["kl" + k].tobytes() or not np.array_equal(ld, rec["ldesc" + k]):
            bad.append("line%s(%d vs %d)" % (k, len(rec["kl" + k]), m))
    if not lines_only:
        ur, dp, _, _ = fr.stereo_points()
        if ur.tobytes() != rec["uright"].tobytes() or dp.tobytes() != rec["depth"].tobytes():
            bad.append("stereoP")
    disp, le, _ = fr.stereo_lines()
    if disp.tobytes() != rec["disp"].tobytes() or le.tobytes() != rec["le"].tobytes():
        bad.append("stereoL")
    return bad


SWEEP_SEEDS = list(range(300, 316))


@pytest.fixture(scope="module")
def sweep(gpu):
    """16 seeded EuRoC-size pairs and the oracle's line tables for them (lsd_nfeatures = 0: every segment is compared)."""
    g = gpu
    W, H = 752, 480
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        pairs = list(ex.map(lambda s: g.synth.make_stereo_pair(s, W, H), SWEEP_SEEDS))
    return W, H, pairs


@pytest.mark.parametrize("mode,batch", [(2, 16), (3, 16), (3, 1), (1, 16), (0, 16)])
def test_seed_sweep_every_lsd_schedule(gpu, sweep, mode, batch):
    """Sequential waves, tile-sequential relaxation (64-px tiles for the batch, 32-px tiles per pair), lane relaxation and auto:
    16 seeds x 2 eyes, keylines + LBD + stereo lines byte-identical to the oracle."""
    g = gpu
    W, H, pairs = sweep
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=0, max_frames=batch, lsd_mode=mode)
    fe = g.Frontend(cfg)
    stages = g.capi.RUN_LINES | g.capi.RUN_STEREO_LINES
    recs = []
    for b in range(0, len(pairs), batch):
        recs += fe.batch_run_host(np.stack([np.stack(p) for p in pairs[b:b + batch]]), stages=stages)
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        bad = list(ex.map(lambda i: frame_mismatches(g, cfg, recs[i], pairs[i][0], pairs[i][1], lines_only=True), range(len(pairs))))
    assert not any(bad), {SWEEP_SEEDS[i]: b for i, b in enumerate(bad) if b}
    assert min(len(r["klL"]) for r in recs) > 300


def _content_cases(g):
    rng = np.random.default_rng(0)
    cases = []
    for (W, H) in ((96, 64), (128, 96), (160, 120), (257, 131)):
        L4, R4 = g.synth.make_stereo_pair(3, 4 * W, 4 * H)
        cases.append(("small %dx%d" % (W, H), dict(orb_nfeatures=200, lsd_nfeatures=0), np.ascontiguousarray(L4[::4, ::4]),
                      np.ascontiguousarray(R4[::4, ::4])))
    W, H = 376, 240
    L, R = g.synth.make_stereo_pair(4, W, H)
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    sat = np.clip(L.astype(int) * 3 - 200, 0, 255).astype(np.uint8)
    grad = (np.add.outer(np.arange(H), np.arange(W)) % 256).astype(np.uint8)
    cases += [("noise both eyes", dict(orb_nfeatures=1000, lsd_nfeatures=50), noise, noise),
              ("noise left / scene right", dict(orb_nfeatures=1000, lsd_nfeatures=50), noise, R),
              ("identical eyes (zero disparity)", dict(orb_nfeatures=500, lsd_nfeatures=50), L, L),
              ("tiny feature budget", dict(orb_nfeatures=12, lsd_nfeatures=3), L, R),
              ("huge feature budget", dict(orb_nfeatures=4500, lsd_nfeatures=1000), L, R),
              ("saturated contrast", dict(orb_nfeatures=800, lsd_nfeatures=0), sat, sat),
              ("sawtooth ramp", dict(orb_nfeatures=800, lsd_nfeatures=0), grad, grad)]
    return cases


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_unusual_content_sizes_and_budgets(gpu, mode):
    g = gpu
    failures = {}
    for name, over, L, R in _content_cases(g):
        cfg = g.capi.default_config(L.shape[1], L.shape[0], lsd_mode=mode, **over)
        rec = g.Frontend(cfg).batch_run_host(np.stack([L, R])[None])[0]
        bad = frame_mismatches(g, cfg, rec, L, R)
        if bad:
            failures[name] = bad
    assert not failures, failures


EXTREME = [
    ("one pyramid level", dict(orb_nlevels=1)), ("two levels, factor 2.0", dict(orb_nlevels=2, orb_scale_factor=2.0)),
    ("12 levels, factor 1.1", dict(orb_nlevels=12, orb_scale_factor=1.1)),
    ("FAST thresholds 5 / 2", dict(orb_ini_th_fast=5, orb_min_th_fast=2)),
    ("FAST thresholds 80 / 40", dict(orb_ini_th_fast=80, orb_min_th_fast=40)),
    ("LSD 64 bins", dict(lsd_n_bins=64)), ("LSD 4096 bins", dict(lsd_n_bins=4096)), ("LSD scale 0.5", dict(lsd_scale=0.5)),
    ("LSD scale 2.0", dict(lsd_scale=2.0)), ("LSD angle tolerance 5 deg", dict(lsd_ang_th=5.0)),
    ("LSD angle tolerance 60 deg", dict(lsd_ang_th=60.0)),
] + [("sequential waves, angle tolerance %g deg" % a, dict(lsd_ang_th=a, lsd_mode=2)) for a in (1.0, 45.0, 85.0, 86.5, 120.0)] + [
    ("tiles, angle tolerance %g deg" % a, dict(lsd_ang_th=a, lsd_mode=3)) for a in (1.0, 45.0, 86.5, 120.0)] + [
    ("LSD quant 0.5", dict(lsd_quant=0.5)), ("LSD quant 8", dict(lsd_quant=8.0)), ("min line length 0.3", dict(min_line_length=0.3)),
    ("matching window 0", dict(matching_s_ws=0)), ("matching window 40", dict(matching_s_ws=40)), ("tiny bf", dict(bf=1.0)),
    ("refine = 1 (unsupported)", dict(lsd_refine=1)), ("zero levels", dict(orb_nlevels=0)), ("negative features", dict(orb_nfeatures=-5)),
]


def test_extreme_configurations_are_refused_or_exact(gpu):
    g = gpu
    W, H = 376, 240
    L, R = g.synth.make_stereo_pair(4, W, H)
    failures, refused = {}, []
    for name, over in EXTREME:
        base = dict(orb_nfeatures=400, lsd_nfeatures=0)
        base.update(over)
        cfg = g.capi.default_config(W, H, **base)
        try:
            fe = g.Frontend(cfg)
        except g.capi.PliError as e:
            assert e.status in (-1,), (name, e)
            refused.append(name)
            continue
        rec = fe.batch_run_host(np.stack([L, R])[None])[0]
        bad = frame_mismatches(g, cfg, rec, L, R)
        if bad:
            failures[name] = bad
    assert not failures, failures
    assert {"refine = 1 (unsupported)", "zero levels", "negative features", "LSD 4096 bins"} <= set(refused), refused


@pytest.mark.parametrize("mode,F", [(0, 8), (3, 8), (3, 1)])
def test_determinism_soak(gpu, mode, F):
    """The same batch 12 times: byte-identical result tables."""
    g = gpu
    W, H = 752, 480
    uniq = np.stack([np.stack(g.synth.make_stereo_pair(200 + s, W, H)) for s in range(F)])
    left, right = np.ascontiguousarray(uniq[:, 0]), np.ascontiguousarray(uniq[:, 1])
    fe = g.Frontend(g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=F, lsd_mode=mode))
    ref = None
    for it in range(12):
        table = np.zeros(fe.table_bytes(F), np.uint8)
        g.capi.check(fe.L.pli_batch_run_host(fe.h, F, g.capi.ptr(left), g.capi.ptr(right), W, W * H, g.capi.RUN_ALL, g.capi.ptr(table)))
        if ref is None:
            ref = table
        else:
            assert np.array_equal(ref, table), "run %d differs in %d bytes" % (it, int((ref != table).sum()))
