"""GPU parity tests: the HIP front-end, called through the C ABI, against the oracle.

Bit-exact for every integer / byte / index output (FAST scores, keypoint tables, rBRIEF and
LBD bits, match indices) and — because the kernels reproduce the reference's float operation
order — also bit-exact for the float outputs (angles, LSD endpoints, uRight/depth, line
disparities); the contract in BASELINE.json only asks for 0.5 px on LSD endpoints.
"""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def assert_frame_equal(g, rec, fr, L, R, what=""):
    """rec: parsed GPU record; fr: oracle Frame (fresh); runs the oracle and compares every table."""
    for eye, img, k in ((0, L, "L"), (1, R, "R")):
        n, kp, desc = fr.orb_extract(eye, img)
        assert n == len(rec["kp" + k]), "%s eye %d keypoint count %d vs oracle %d" % (what, eye, len(rec["kp" + k]), n)
        for f in kp.dtype.names:
            d = np.flatnonzero(rec["kp" + k][f].view(np.int32) != kp[f].view(np.int32))
            assert d.size == 0, "%s eye %d kp.%s differs at %s" % (what, eye, f, d[:5])
        assert np.array_equal(rec["desc" + k], desc), "%s eye %d ORB descriptors" % (what, eye)
        m, kl, ld = fr.line_extract(eye, img)
        assert m == len(rec["kl" + k]), "%s eye %d keyline count %d vs oracle %d" % (what, eye, len(rec["kl" + k]), m)
        # contract: endpoints within 0.5 px ...
        for f in ("startPointX", "startPointY", "endPointX", "endPointY"):
            assert np.abs(rec["kl" + k][f] - kl[f]).max(initial=0) <= 0.5
        # ... achieved: identical bits
        assert rec["kl" + k].tobytes() == kl.tobytes(), "%s eye %d keylines" % (what, eye)
        assert np.array_equal(rec["ldesc" + k], ld), "%s eye %d LBD descriptors" % (what, eye)
    ur, dp, _, _ = fr.stereo_points()
    assert rec["uright"].tobytes() == ur.tobytes() and rec["depth"].tobytes() == dp.tobytes(), what + " stereo points"
    disp, le, _ = fr.stereo_lines()
    assert rec["disp"].tobytes() == disp.tobytes() and rec["le"].tobytes() == le.tobytes(), what + " stereo lines"
    assert rec["counts"][4] == int((ur >= 0).sum()) and rec["counts"][5] == int((disp[:, 0] >= 0).sum())


# ---------------------------------------------------------------------------------------------
# config 2: 752x480, 1200 ORB kp + LSD/LBD, extract + stereo match — every stage
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg2(gpu):
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=2)
    fe = g.Frontend(cfg)
    fe.debug_enable(True)
    L, R = g.synth.make_stereo_pair(0, W, H)
    fr = g.po.Frame(ocfg(g, cfg))
    return g, cfg, fe, fr, L, R


def test_config2_orb_stages(cfg2):
    g, cfg, fe, fr, L, R = cfg2
    capi = g.capi
    for eye, img in ((0, L), (1, R)):
        n, kp, desc = fe.orb_extract(eye, img)
        on, okp, odesc = fr.orb_extract(eye, img)
        for l in range(cfg.orb_nlevels):
            assert np.array_equal(fe.debug_fetch(eye, capi.DBG_PYRAMID_LEVEL, l), fr.pyramid(eye, l).ravel()), ("pyramid", l)
            assert np.array_equal(fe.pyramid_level(eye, l), fr.pyramid(eye, l)), ("mvImagePyramid", l)
            assert np.array_equal(fe.debug_points(eye, capi.DBG_FAST_CANDIDATES, l), fr.level_points(eye, l)), ("FAST", l)
            assert np.array_equal(fe.debug_points(eye, capi.DBG_LEVEL_KEYPOINTS, l), fr.level_points(eye, l, True)), ("octree", l)
            assert np.array_equal(fe.debug_fetch(eye, capi.DBG_BLUR_LEVEL, l), fr.pyramid(eye, l, True).ravel()), ("blur", l)
        assert n == on and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)
        assert n >= 1200                                   # the quadtree returns at least the quota on this stream


def test_config2_line_stages(cfg2):
    g, cfg, fe, fr, L, R = cfg2
    capi = g.capi
    W, H = cfg.width, cfg.height
    for eye, img in ((0, L), (1, R)):
        n, kl, ld = fe.line_extract(eye, img)
        on, okl, old = fr.line_extract(eye, img)
        if cfg.parity_flags & capi.PARITY_LSD_F64:         # OpenCV 3.x: the scaled image is a CV_64FC1 plane
            assert np.array_equal(fe.debug_fetch(eye, capi.DBG_LSD_SCALED).view(np.float64), fr.lsd_scaled64(eye).ravel())
        else:
            assert np.array_equal(fe.debug_fetch(eye, capi.DBG_LSD_SCALED), fr.lsd_scaled(eye).ravel())
        oang = fr.lsd_angle(eye).ravel()
        assert np.array_equal(fe.debug_fetch(eye, capi.DBG_LSD_ANGLE).view(np.float32), oang)
        raw = fe.debug_fetch(eye, capi.DBG_LSD_ORDER).view(np.int32)
        oo = fr.lsd_order(eye)
        assert np.array_equal(raw[1:1 + raw[0]], oo[oang[oo] != -1024])
        raw = fe.debug_fetch(eye, capi.DBG_LSD_SEGMENTS)
        ns = int(raw[:4].view(np.int32)[0])
        segs = raw[4:4 + 16 * ns].view(np.float32).reshape(-1, 4)
        osegs = fr.lsd_segments(eye)
        assert segs.shape == osegs.shape and np.abs(segs - osegs).max() <= 0.5     # contract
        assert np.array_equal(segs, osegs)                                           # achieved
        raw = fe.debug_fetch(eye, capi.DBG_LBD_DXDY).view(np.int16)
        dx, dy = fr.lbd_dxdy(eye, (H, W))
        assert np.array_equal(raw[:W * H], dx.ravel()) and np.array_equal(raw[W * H:], dy.ravel())
        assert n == on == 100 and kl.tobytes() == okl.tobytes()
        lf = fe.debug_fetch(eye, capi.DBG_LBD_FLOAT).view(np.float32).reshape(-1, 72)[:n]
        assert np.array_equal(lf, fr.lbd_float(eye, on))
        assert np.array_equal(ld, old)


def test_config2_stereo(cfg2):
    g, cfg, fe, fr, L, R = cfg2
    for eye, img in ((0, L), (1, R)):
        fe.orb_extract(eye, img); fe.line_extract(eye, img)
        fr.orb_extract(eye, img); fr.line_extract(eye, img)
    ur, dp = fe.compute_stereo_matches()
    our, odp, obi, osad = fr.stereo_points()
    n = len(our)
    raw = fe.debug_fetch(0, g.capi.DBG_STEREO_SAD).view(np.int32)
    assert np.array_equal(raw[:fe.kp_cap][:n], osad) and np.array_equal(raw[fe.kp_cap:][:n], obi)   # match indices
    assert ur[:n].tobytes() == our.tobytes() and dp[:n].tobytes() == odp.tobytes()
    assert (our >= 0).sum() > 300
    disp, le = fe.compute_stereo_matches_lines()
    odisp, ole, om = fr.stereo_lines()
    m = len(odisp)
    assert disp[:m].tobytes() == odisp.tobytes() and le[:m].tobytes() == ole.tobytes()
    assert (odisp[:, 0] >= 0).sum() > 10


def test_batch_equals_per_call_and_oracle(cfg2):
    g, cfg, fe, fr, L, R = cfg2
    L2, R2 = g.synth.make_stereo_pair(5, cfg.width, cfg.height)
    recs = fe.batch_run_host(np.stack([np.stack([L, R]), np.stack([L2, R2])]))
    assert_frame_equal(g, recs[0], g.po.Frame(ocfg(g, cfg)), L, R, "batch frame 0")
    assert_frame_equal(g, recs[1], g.po.Frame(ocfg(g, cfg)), L2, R2, "batch frame 1")
    # the per-call drop-ins give the same tables as the batch path
    n, kp, desc = fe.orb_extract(0, L2)
    assert kp.tobytes() == recs[1]["kpL"].tobytes() and np.array_equal(desc, recs[1]["descL"])
    m, kl, ld = fe.line_extract(1, R2)
    assert kl.tobytes() == recs[1]["klR"].tobytes() and np.array_equal(ld, recs[1]["ldescR"])


@pytest.mark.parametrize("name,flags", [("oracle_small.npz", 0), ("oracle_small_default.npz", 9), ("oracle_small_all.npz", 15)])
def test_golden_vectors_small(gpu, name, flags):
    """The committed fixtures (one per set of parity flags; oracle_small.npz is round 1's file, flags 0)."""
    g = gpu
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    L, R = gold["left"], gold["right"]
    cfg = g.capi.default_config(L.shape[1], L.shape[0], orb_nfeatures=300, lsd_nfeatures=40, parity_flags=flags)
    rec = g.Frontend(cfg).batch_run_host(np.stack([L, R])[None])[0]
    for s in ("L", "R"):
        assert rec["kp" + s].tobytes() == gold["kp" + s].tobytes() and np.array_equal(rec["desc" + s], gold["d" + s])
        assert rec["kl" + s].tobytes() == gold["kl" + s].tobytes() and np.array_equal(rec["ldesc" + s], gold["ld" + s])
    assert rec["uright"].tobytes() == gold["uright"].tobytes() and rec["depth"].tobytes() == gold["depth"].tobytes()
    assert rec["disp"].tobytes() == gold["disp"].tobytes() and rec["le"].tobytes() == gold["le"].tobytes()


@pytest.mark.parametrize("seed,W,H,nf,nl", [(1, 752, 480, 1000, 500), (2, 640, 480, 800, 0), (3, 376, 240, 500, 60),
                                           (4, 200, 136, 150, 30), (5, 643, 481, 600, 0), (6, 333, 245, 300, 40),
                                           # KITTI00-02.yaml:9,21-22,28,41 (Examples/Stereo/stereo_kitti.cc): 1241x376, fx 718.856, bf 386.1448
                                           (7, 1241, 376, 2000, 500)])
def test_other_shapes_and_parameters(gpu, seed, W, H, nf, nl):
    g = gpu
    rig = dict(bf=386.1448, fx=718.856) if W == 1241 else {}
    cfg = g.capi.default_config(W, H, orb_nfeatures=nf, lsd_nfeatures=nl, max_frames=1, **rig)
    L, R = g.synth.make_stereo_pair(seed, W, H)
    rec = g.Frontend(cfg).batch_run_host(np.stack([L, R])[None])[0]
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, R, "seed %d %dx%d" % (seed, W, H))


def test_tiny_budgets_and_small_images(gpu):
    """A per-level quota of 0-3 keypoints (DistributeOctTree still expands its root nodes once: up to 4 * nIni per
    level, ORBextractor.cc:540-589) and images whose top pyramid levels have no FAST cells at all."""
    g = gpu
    W, H = 376, 240
    L, R = g.synth.make_stereo_pair(4, W, H)
    cfg = g.capi.default_config(W, H, orb_nfeatures=12, lsd_nfeatures=3)
    rec = g.Frontend(cfg).batch_run_host(np.stack([L, R])[None])[0]
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, R, "12 keypoints")
    assert len(rec["kpL"]) > 12                                           # more than asked for, like the reference
    L4, R4 = g.synth.make_stereo_pair(3, 512, 384)
    for step, (w, h) in ((4, (128, 96)), (2, (256, 192))):
        Ls, Rs = np.ascontiguousarray(L4[::step, ::step]), np.ascontiguousarray(R4[::step, ::step])
        cfg = g.capi.default_config(w, h, orb_nfeatures=200, lsd_nfeatures=0)
        rec = g.Frontend(cfg).batch_run_host(np.stack([Ls, Rs])[None])[0]
        assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), Ls, Rs, "%dx%d" % (w, h))


@pytest.mark.parametrize("over", [
    dict(orb_scale_factor=1.5, orb_nlevels=5, orb_ini_th_fast=30, orb_min_th_fast=10, lsd_scale=1.0, lsd_ang_th=15.0),
    dict(orb_scale_factor=1.1, orb_nlevels=10, lsd_scale=0.8, lsd_quant=1.0, lsd_n_bins=512, min_line_length=0.05),
    dict(orb_nlevels=3, lsd_scale=1.5, lsd_sigma_scale=0.8, lsd_ang_th=30.0, lsd_mode=2, matching_s_ws=5, line_sim_th=0.6,
         best_lr_matches=0),
    dict(orb_scale_factor=2.7, orb_nlevels=3, lsd_scale=2.6),      # resize windows too wide for the LDS staging: direct path
    dict(lsd_sigma_scale=0.4),                                     # blur radius 2: the fused front pass with looped taps
    dict(lsd_sigma_scale=0.25, lsd_scale=1.3),                     # blur radius 1
])
def test_non_default_algorithm_parameters(gpu, over):
    """The yaml-tunable parameters of both extractors and the line matcher away from their defaults
    (pyramid shape, FAST thresholds, LSD scale = 1 / < 1 / > 1, angle tolerance, quantisation, bins, length cut)."""
    g = gpu
    W, H = 640, 400
    cfg = g.capi.default_config(W, H, orb_nfeatures=700, lsd_nfeatures=0, max_frames=1, **over)
    L, R = g.synth.make_stereo_pair(9, W, H)
    rec = g.Frontend(cfg).batch_run_host(np.stack([L, R])[None])[0]
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, R, str(over))
    assert len(rec["kpL"]) > 100 and len(rec["klL"]) > 20


# ---------------------------------------------------------------------------------------------
# config 3: 1280x720, 8 levels, 2000 kp + 200 lines, frame-to-frame track match
# ---------------------------------------------------------------------------------------------
def n_all(best, obest, q):
    """(an upper bound for the number of matches of a masked search: every valid query)"""
    return int((q["valid"] != 0).sum()) + 1


def test_config3_f2f(gpu):
    g = gpu
    W, H = 1280, 720
    cfg = g.capi.default_config(W, H, orb_nfeatures=2000, lsd_nfeatures=200, max_frames=2)
    fe = g.Frontend(cfg)
    f0 = g.synth.make_stereo_pair(8, W, H, t=0)
    f1 = g.synth.make_stereo_pair(8, W, H, t=1)
    recs = fe.batch_run_host(np.stack([np.stack(f0), np.stack(f1)]))
    assert_frame_equal(g, recs[0], g.po.Frame(ocfg(g, cfg)), f0[0], f0[1], "config3 t=0")
    assert_frame_equal(g, recs[1], g.po.Frame(ocfg(g, cfg)), f1[0], f1[1], "config3 t=1")
    last, cur = recs[0], recs[1]
    # line f2f: match(last.desc_line, cur.desc_line, minRatio12L), Tracking.cc:3058
    n, m12 = fe.match(last["ldescL"], cur["ldescL"], 0.9)
    on, om12 = g.po.match_lines(last["ldescL"], cur["ldescL"], 0.9, True)
    assert n == on and np.array_equal(m12, om12) and n > 20
    # point f2f: SearchByProjection(cur, last, th=7): queries = last frame's stereo points moved by the known motion
    kpl = last["kpL"]
    q = np.zeros(len(kpl), g.capi.PROJ_QUERY_DT)
    q["u"] = kpl["x"] - 3.0; q["v"] = kpl["y"] - 1.0
    q["radius"] = 7.0 * (np.float32(1.2) ** kpl["octave"]).astype(np.float32)
    disp = np.where(last["uright"] >= 0, kpl["x"] - last["uright"], 0).astype(np.float32)
    q["ur"] = q["u"] - disp
    q["min_level"] = kpl["octave"] - 1; q["max_level"] = kpl["octave"] + 1
    q["angle"] = kpl["angle"]; q["valid"] = (last["depth"] > 0).astype(np.int32)
    bounds = (0.0, float(W), 0.0, float(H))
    for chk in (True, False):
        n, best = fe.search_by_projection(q, last["descL"], cur["kpL"], cur["descL"], cur["uright"], bounds, chk)
        on, obest = g.po.search_by_projection(q, last["descL"], cur["kpL"], cur["descL"], cur["uright"], bounds, chk)
        assert n == on and np.array_equal(best, obest)
    assert on > 100
    # ... with 30 % of the current keypoints already holding an observed map point (ORBmatcher.cc:2255-2257), and with a third of the
    # last frame's map points without observations (their matches do not take the keypoint away: the same keypoint is matched again)
    rng0 = np.random.default_rng(30)
    occ = (rng0.random(len(cur["kpL"])) < 0.3).astype(np.uint8)
    qn = q.copy()
    qn["valid"] = np.where((qn["valid"] != 0) & (rng0.random(len(qn)) < 0.33), 3, qn["valid"])
    qq = np.concatenate([qn, qn])                     # (every map point twice: the second copy meets what the first left behind)
    qqd = np.concatenate([last["descL"], last["descL"]])
    for chk in (True, False):
        n, best, raw = fe.search_by_projection(q, last["descL"], cur["kpL"], cur["descL"], cur["uright"], bounds, chk, occupied=occ, with_raw=True)
        on, obest, oraw = g.po.search_by_projection(q, last["descL"], cur["kpL"], cur["descL"], cur["uright"], bounds, chk, occupied=occ, with_raw=True)
        assert n == on and np.array_equal(best, obest) and np.array_equal(raw, oraw)
        assert not occ[best[best >= 0]].any() and 50 < on < n_all(best, obest, q)
        n, best, raw = fe.search_by_projection(qq, qqd, cur["kpL"], cur["descL"], cur["uright"], bounds, chk, occupied=occ, with_raw=True)
        on, obest, oraw = g.po.search_by_projection(qq, qqd, cur["kpL"], cur["descL"], cur["uright"], bounds, chk, occupied=occ, with_raw=True)
        assert n == on and np.array_equal(best, obest) and np.array_equal(raw, oraw)
        taken = raw[raw >= 0]
        assert len(taken) > len(np.unique(taken))     # some keypoint was matched twice: by a query without observations first
    # local map tracking: SearchByProjection(F, vpMapPoints, th) ORBmatcher.cc:44 with the last frame's
    # points standing in for the local map (each twice, so that later map points meet occupied keypoints), and
    # match(MapLines, Frame) LineMatcher.cpp:161
    rng = np.random.default_rng(3)
    q2 = np.concatenate([q, q])
    q2["min_level"] = np.concatenate([kpl["octave"] - 1, kpl["octave"]]); q2["max_level"] = np.concatenate([kpl["octave"], kpl["octave"] + 1])
    q2["u"][len(q):] += 0.75
    qd2 = np.concatenate([last["descL"], last["descL"]])
    occ = (rng.random(len(cur["kpL"])) < 0.3).astype(np.uint8)
    for nnratio, oc in ((0.8, occ), (0.6, None), (1.0, occ)):
        n, best = fe.search_local_map(q2, qd2, cur["kpL"], cur["descL"], cur["uright"], bounds, nnratio, oc)
        on, obest = g.po.search_local_map(q2, qd2, cur["kpL"], cur["descL"], cur["uright"], oc, bounds, nnratio)
        assert n == on and np.array_equal(best, obest)
        assert on > 100 and (oc is None or not oc[obest[obest >= 0]].any())
        hit = obest[obest >= 0]
        assert len(np.unique(hit)) == len(hit)
    n, m12 = fe.match_nnr(last["ldescL"], cur["ldescL"], 0.9)
    on, om12 = g.po.match_nnr(last["ldescL"], cur["ldescL"], 0.9)
    assert n == on and np.array_equal(m12, om12) and n > 20


def test_local_map_search_ties_and_levels(gpu):
    """Second-best / same-level ratio rule of ORBmatcher.cc:118-138 on engineered tables with many equal distances."""
    g = gpu
    fe = g.Frontend(g.capi.default_config(128, 128))
    rng = np.random.default_rng(11)
    ncur, nq = 700, 500
    kp = np.zeros(ncur, g.capi.KEYPOINT_DT)
    kp["x"] = rng.uniform(0, 640, ncur).astype(np.float32); kp["y"] = rng.uniform(0, 480, ncur).astype(np.float32)
    kp["octave"] = rng.integers(0, 4, ncur)
    desc = (rng.integers(0, 2, (ncur, 32)) * 255).astype(np.uint8)       # distances are multiples of 8
    ur = np.where(rng.random(ncur) < 0.5, kp["x"] - rng.uniform(0, 30, ncur), -1).astype(np.float32)
    q = np.zeros(nq, g.capi.PROJ_QUERY_DT)
    src = rng.integers(0, ncur, nq)
    q["u"] = kp["x"][src] + rng.uniform(-4, 4, nq).astype(np.float32); q["v"] = kp["y"][src] + rng.uniform(-4, 4, nq).astype(np.float32)
    q["radius"] = rng.uniform(10, 60, nq).astype(np.float32)
    q["ur"] = q["u"] - rng.uniform(0, 30, nq).astype(np.float32)
    q["min_level"] = rng.integers(-1, 3, nq); q["max_level"] = q["min_level"] + rng.integers(0, 3, nq)
    q["valid"] = rng.random(nq) < 0.9
    qd = desc[src].copy()
    flip = rng.random((nq, 32)) < 0.25
    qd[flip] ^= 255
    bounds = (0.0, 640.0, 0.0, 480.0)
    tot = 0
    for nnratio in (0.8, 0.5, 1.0):
        n, best = fe.search_local_map(q, qd, kp, desc, ur, bounds, nnratio)
        on, obest = g.po.search_local_map(q, qd, kp, desc, ur, None, bounds, nnratio)
        assert n == on and np.array_equal(best, obest)
        tot += on
    assert tot > 100
    # queries off the grid, no keypoints, no queries
    q["u"] = 5000.0
    n, best = fe.search_local_map(q, qd, kp, desc, ur, bounds, 0.8)
    assert n == 0 and (best == -1).all()
    n, best = fe.search_local_map(q[:0], qd[:0], kp, desc, ur, bounds, 0.8)
    assert n == 0 and len(best) == 0
    n, m = fe.match_nnr(desc[:50], desc[:0], 0.9)
    assert n == 0 and (m == -1).all()
    for nnr in (0.9, 1.0):
        n, m = fe.match_nnr(qd, desc, nnr)
        on, om = g.po.match_nnr(qd, desc, nnr)
        assert n == on and np.array_equal(m, om)


@pytest.mark.gpu
@pytest.mark.parametrize("ncur,nq", [(3000, 1500), (17000, 600)])
def test_projection_searches_dense_windows(gpu, ncur, nq):
    """Both projection searches with windows that hold more candidates than the per-query candidate list keeps
    (rescanned in the ordered phase) and, at 17000 keypoints, the frame too large for the LDS owner table."""
    g = gpu
    fe = g.Frontend(g.capi.default_config(128, 128))
    rng = np.random.default_rng(ncur)
    kp = np.zeros(ncur, g.capi.KEYPOINT_DT)
    kp["x"] = rng.uniform(0, 640, ncur).astype(np.float32); kp["y"] = rng.uniform(0, 480, ncur).astype(np.float32)
    kp["octave"] = rng.integers(0, 8, ncur); kp["angle"] = rng.uniform(0, 360, ncur).astype(np.float32)
    desc = rng.integers(0, 256, (ncur, 32), dtype=np.uint8)
    desc[:, 8:] = desc[0, 8:]                                            # close descriptors: most candidates pass the limit
    ur = np.where(rng.random(ncur) < 0.5, kp["x"] - rng.uniform(0, 30, ncur), -1).astype(np.float32)
    occ = (rng.random(ncur) < 0.2).astype(np.uint8)
    q = np.zeros(nq, g.capi.PROJ_QUERY_DT)
    src = rng.integers(0, ncur, nq)
    q["u"] = kp["x"][src] + rng.uniform(-4, 4, nq).astype(np.float32); q["v"] = kp["y"][src] + rng.uniform(-4, 4, nq).astype(np.float32)
    q["radius"] = np.where(rng.random(nq) < 0.5, rng.uniform(5, 30, nq), rng.uniform(60, 200, nq)).astype(np.float32)
    q["ur"] = q["u"] - rng.uniform(0, 30, nq).astype(np.float32)
    q["min_level"] = rng.integers(-1, 3, nq); q["max_level"] = np.where(rng.random(nq) < 0.3, -1, q["min_level"] + rng.integers(0, 6, nq))
    q["angle"] = (kp["angle"][src] + rng.choice([0.0, 0.0, 0.0, 90.0, 200.0], nq)).astype(np.float32) % 360
    q["valid"] = rng.random(nq) < 0.95
    qd = desc[src].copy()
    qd[:, :3] ^= rng.integers(0, 256, (nq, 3), dtype=np.uint8)
    bounds = (0.0, 640.0, 0.0, 480.0)
    for nnratio in (0.8, 0.3):
        n, best = fe.search_local_map(q, qd, kp, desc, ur, bounds, nnratio, cur_occupied=occ)
        on, obest = g.po.search_local_map(q, qd, kp, desc, ur, occ, bounds, nnratio)
        assert n == on and np.array_equal(best, obest)
    assert on > nq // 20
    for ori in (True, False):
        n, best = fe.search_by_projection(q, qd, kp, desc, ur, bounds, ori)
        on, obest = g.po.search_by_projection(q, qd, kp, desc, ur, bounds, ori)
        assert n == on and np.array_equal(best, obest)
    assert on > nq // 20
    # ... the frame-to-frame search with occupied keypoints and map points without observations (both kernels: the two-phase one
    # and, at 17000 keypoints, the scan)
    qn = q.copy()
    qn["valid"] = np.where((qn["valid"] != 0) & (rng.random(nq) < 0.4), 3, qn["valid"])
    for ori in (True, False):
        n, best, raw = fe.search_by_projection(qn, qd, kp, desc, ur, bounds, ori, occupied=occ, with_raw=True)
        on, obest, oraw = g.po.search_by_projection(qn, qd, kp, desc, ur, bounds, ori, occupied=occ, with_raw=True)
        assert n == on and np.array_equal(best, obest) and np.array_equal(raw, oraw)
        assert not occ[raw[raw >= 0]].any()


@pytest.mark.parametrize("nl,nr,nq", [(1500, 1400, 900), (300, 5000, 2500), (0, 40, 30)])
def test_local_map_search_two_fisheye_cameras(gpu, nl, nr, nq):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) with F.Nleft != -1 (ORBmatcher.cc:44-214) on random tables: both
    cameras, stereo partners both ways, slots taken before the call, windows with more candidates than the candidate list
    keeps, map points seen by one camera only."""
    g = gpu
    fe = g.Frontend(g.capi.default_config(128, 128))
    rng = np.random.default_rng(nl + 7 * nr)

    def table(n):
        kp = np.zeros(n, g.capi.KEYPOINT_DT)
        kp["x"] = rng.uniform(0, 640, n).astype(np.float32); kp["y"] = rng.uniform(0, 480, n).astype(np.float32)
        kp["octave"] = rng.integers(0, 8, n)
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        if n:
            d[:, 8:] = 0x5A                                              # close descriptors: most candidates pass the limit
        return kp, d
    kpl, dl = table(nl)
    kpr, dr = table(nr)
    l2r = np.full(nl, -1, np.int32); r2l = np.full(nr, -1, np.int32)
    npair = min(nl, nr) // 3
    if npair:
        a = rng.choice(nl, npair, replace=False); b = rng.choice(nr, npair, replace=False)
        l2r[a] = b; r2l[b] = a
    occl = (rng.random(nl) < 0.2).astype(np.uint8); occr = (rng.random(nr) < 0.2).astype(np.uint8)

    def queries(kp, n):
        q = np.zeros(nq, g.capi.PROJ_QUERY_DT)
        if n == 0:
            q["u"] = rng.uniform(0, 640, nq); q["v"] = rng.uniform(0, 480, nq); q["radius"] = 10; q["valid"] = 1
            return q, None
        src = rng.integers(0, n, nq)
        q["u"] = kp["x"][src] + rng.uniform(-4, 4, nq).astype(np.float32); q["v"] = kp["y"][src] + rng.uniform(-4, 4, nq).astype(np.float32)
        q["radius"] = np.where(rng.random(nq) < 0.6, rng.uniform(5, 30, nq), rng.uniform(60, 200, nq)).astype(np.float32)
        q["min_level"] = rng.integers(-1, 3, nq); q["max_level"] = np.where(rng.random(nq) < 0.3, -1, q["min_level"] + rng.integers(0, 6, nq))
        q["valid"] = rng.random(nq) < 0.8
        return q, src
    ql, srcl = queries(kpl, nl)
    qr, srcr = queries(kpr, nr)
    qd = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    qd[:, 8:] = 0x5A
    if srcr is not None:
        qd[:, :8] = dr[srcr][:, :8]
    qd[:, :2] ^= rng.integers(0, 256, (nq, 2), dtype=np.uint8)
    bounds = (0.0, 640.0, 0.0, 480.0)
    for nnratio in (0.8, 0.3):
        for ol, orr in ((occl, occr), (None, None)):
            n, mpl, mpr = fe.search_local_map_fisheye(ql, qr, qd, kpl, dl, l2r, kpr, dr, r2l, bounds, nnratio, ol, orr)
            on, ompl, ompr = g.po.search_local_map_fisheye(ql, qr, qd, kpl, dl, ol, l2r, kpr, dr, orr, r2l, bounds, nnratio)
            assert n == on and np.array_equal(mpl, ompl) and np.array_equal(mpr, ompr)
    assert on > nq // 20
    # nothing to search, and partner indices out of range
    n, mpl, mpr = fe.search_local_map_fisheye(ql[:0], qr[:0], qd[:0], kpl, dl, l2r, kpr, dr, r2l, bounds)
    assert n == 0 and (mpl == -1).all() and (mpr == -1).all()
    if nr:
        bad = r2l.copy(); bad[0] = nl
        with pytest.raises(g.capi.PliError):
            fe.search_local_map_fisheye(ql, qr, qd, kpl, dl, l2r, kpr, dr, bad, bounds)


# ---------------------------------------------------------------------------------------------
# matchers on engineered descriptor tables (ties, empties)
# ---------------------------------------------------------------------------------------------
def test_matchers_random_and_ties(gpu):
    g = gpu
    fe = g.Frontend(g.capi.default_config(128, 128))
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (333, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (257, 32), dtype=np.uint8)
    b[100:110] = a[5]                   # exact ties at distance 0
    b[200] = a[6]; b[201] = a[6]; b[201, 0] ^= 1
    assert np.array_equal(fe.descriptor_distance(a[:257], b), g.po.descriptor_distance(a[:257], b))
    idx, dist = fe.knn2(a, b)
    oidx, odist = g.po.knn2(a, b)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    for nnr in (0.9, 0.75, 1.0):
        n, m = fe.match(a, b, nnr)
        on, om = g.po.match_lines(a, b, nnr, True)
        assert n == on and np.array_equal(m, om)
    # low-entropy descriptors: many equal distances
    c = (rng.integers(0, 2, (90, 32)) * 255).astype(np.uint8)
    d = (rng.integers(0, 2, (70, 32)) * 255).astype(np.uint8)
    n, m = fe.match(c, d, 0.9)
    on, om = g.po.match_lines(c, d, 0.9, True)
    assert n == on and np.array_equal(m, om)
    # degenerate sizes
    n, m = fe.match(a[:3], b[:1], 0.9)
    assert n == 0 and m.tolist() == [-1, -1, -1]
    idx, dist = fe.knn2(a[:2], b[:1])
    assert idx.tolist() == [[0, -1], [0, -1]]
    assert fe.descriptor_distance(np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8))[0] == 256


def test_stereo_lines_engineered_tables(gpu):
    """matchGrid prefix-min / mutual rule and the geometric filters on hand-made line tables, through the batch record."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, lsd_nfeatures=100)
    fe = g.Frontend(cfg)
    rng = np.random.default_rng(11)
    L, R = g.synth.make_stereo_pair(9, W, H)
    rec = fe.batch_run_host(np.stack([L, R])[None])[0]
    # cross-check the oracle's table entry point against the frame path on the GPU's own tables
    disp, le, m = g.po.stereo_lines_tables(ocfg(g, cfg), rec["klL"], rec["ldescL"], rec["klR"], rec["ldescR"], W, H)
    assert rec["disp"].tobytes() == disp.tobytes() and rec["le"].tobytes() == le.tobytes()


# ---------------------------------------------------------------------------------------------
# edge cases the reference handles (or crashes on) and the error behaviour of the boundary
# ---------------------------------------------------------------------------------------------
def test_edge_cases(gpu):
    g = gpu
    W, H = 320, 240
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=50)
    fe = g.Frontend(cfg)
    blank = np.full((H, W), 128, np.uint8)
    # blank image: no keypoints, no lines; Frame.cc:146-149 returns early, stereo outputs stay at "none"
    rec = fe.batch_run_host(np.stack([blank, blank])[None])[0]
    assert rec["counts"][:6].tolist() == [0, 0, 0, 0, 0, 0]
    n, kp, desc = fe.orb_extract(0, blank)
    assert n == 0 and len(kp) == 0
    # empty image: ORBextractor::operator() returns -1 (ORBextractor.cc:1072)
    n, _, _ = fe.orb_extract(0, None)
    assert n == -1
    # wrong size / null pointers are errors, not crashes
    with pytest.raises(g.capi.PliError):
        fe.orb_extract(0, np.zeros((H + 1, W), np.uint8))
    assert g.capi.lib().pli_batch_run(fe.h, 1, None, None, W, W * H, 15, None) == -1
    assert g.capi.lib().pli_batch_run_host(fe.h, 5, C.c_void_p(1), C.c_void_p(1), W, W * H, 15, C.c_void_p(1)) == -1   # > max_frames
    # a result table that is not 16-byte aligned is refused (the records hold 8-byte fields)
    import torch
    dimg = torch.zeros(2 * W * H, dtype=torch.uint8, device="cuda")
    dtab = torch.zeros(fe.table_bytes(1) + 16, dtype=torch.uint8, device="cuda")
    assert g.capi.lib().pli_batch_run(fe.h, 1, C.c_void_p(dimg.data_ptr()), C.c_void_p(dimg.data_ptr() + W * H), W, W * H, 15,
                                      C.c_void_p(dtab.data_ptr() + 1)) == -1
    assert g.capi.lib().pli_batch_run(fe.h, 1, C.c_void_p(dimg.data_ptr()), C.c_void_p(dimg.data_ptr() + W * H), W, W * H, 15,
                                      C.c_void_p(dtab.data_ptr() + 16)) == 0
    fe.sync()
    # one eye featureless: left has features, right is blank -> no stereo, still consistent with the oracle
    L, _ = g.synth.make_stereo_pair(12, W, H)
    rec = fe.batch_run_host(np.stack([L, blank])[None])[0]
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, blank, "right blank")
    # a single bright rectangle: few keypoints (corners only), 4 long lines
    rect = np.full((H, W), 30, np.uint8); rect[60:180, 80:240] = 220
    rec = fe.batch_run_host(np.stack([rect, rect])[None])[0]
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), rect, rect, "rectangle")
    assert len(rec["klL"]) >= 4
    # strided input rows
    big = np.zeros((H, W + 40), np.uint8); big[:, :W] = L
    n1, kp1, d1 = fe.orb_extract(0, L)
    view = big[:, :W]
    kp = np.zeros(fe.kp_cap, g.capi.KEYPOINT_DT); desc = np.zeros((fe.kp_cap, 32), np.uint8); n = C.c_int32()
    g.capi.check(g.capi.lib().pli_orb_extract(fe.h, 0, C.c_void_p(big.ctypes.data), W, H, big.strides[0],
                                              g.capi.ptr(kp), fe.kp_cap, g.capi.ptr(desc), C.byref(n)))
    assert n.value == n1 and kp[:n1].tobytes() == kp1.tobytes()
    # stereo match before extraction on a fresh context is a state error
    fe2 = g.Frontend(cfg)
    with pytest.raises(g.capi.PliError) as e:
        fe2.compute_stereo_matches()
    assert e.value.status == -6


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_lsd_execution_modes_give_identical_results(gpu, mode):
    """lsd_mode 1 (rank-ordered relaxation), 2 (sequential waves) and 3 (tile-sequential relaxation) are three schedules of
    the same algorithm."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=0, max_frames=2, lsd_mode=mode)
    fe = g.Frontend(cfg)
    frames = [g.synth.make_stereo_pair(s, W, H) for s in (31, 32)]
    recs = fe.batch_run_host(np.stack([np.stack(f) for f in frames]), stages=g.capi.RUN_LINES | g.capi.RUN_STEREO_LINES)
    for rec, (L, R) in zip(recs, frames):
        fr = g.po.Frame(ocfg(g, cfg))
        for eye, img, k in ((0, L, "L"), (1, R, "R")):
            m, kl, ld = fr.line_extract(eye, img)
            assert m == len(rec["kl" + k]) and m > 500          # lsd_nfeatures = 0 keeps every segment above the length cut
            assert rec["kl" + k].tobytes() == kl.tobytes() and np.array_equal(rec["ldesc" + k], ld)
        disp, le, _ = fr.stereo_lines()
        assert rec["disp"].tobytes() == disp.tobytes() and rec["le"].tobytes() == le.tobytes()


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_lsd_hostile_images(gpu, mode):
    """Images that stress the relaxation's work lists (noise: hundreds of thousands of tiny regions; checkerboard and
    stripes: many long regions of equal gradient, i.e. long chains of equal-bin seeds) — same answer as the oracle."""
    g = gpu
    W, H = 376, 240
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    checker = (((xx // 16) + (yy // 16)) % 2 * 200 + 20).astype(np.uint8)
    stripes = ((np.sin((xx + 2 * yy) / 5.0) * 0.5 + 0.5) * 255).astype(np.uint8)
    ramp = ((xx * 255) // (W - 1)).astype(np.uint8)                       # constant gradient: one bin, raster-ordered seeds
    mixed = np.where(xx < W // 2, noise, stripes).astype(np.uint8)
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, lsd_mode=mode)
    fe = g.Frontend(cfg)
    for name, img in (("noise", noise), ("checker", checker), ("stripes", stripes), ("ramp", ramp), ("mixed", mixed)):
        fr = g.po.Frame(ocfg(g, cfg))
        n, kl, ld = fe.line_extract(0, img)
        m, okl, old = fr.line_extract(0, img)
        assert n == m, "%s: %d lines vs oracle %d" % (name, n, m)
        assert kl.tobytes() == okl.tobytes() and np.array_equal(ld, old), name


def test_long_parallel_structures_settle_without_the_fallback(gpu, monkeypatch):
    """Full-size diagonal stripes: every flank is one region of thousands of pixels that has seeds in a dozen tiles, so round 1 of
    the tile relaxation grows it a dozen times at once, each copy with its own queue overflow blocks.  With the arena a context
    gets (16 words per scaled pixel) the image settles in a few rounds; with the 3 words of rounds 1-2 (dev switch PLI_RX_ARENA)
    it runs out of blocks and is redone by the sequential grower — the same lines either way, the oracle's."""
    g = gpu
    W, H = 752, 480
    yy, xx = np.mgrid[0:H, 0:W]
    stripes = ((np.sin((xx + 2 * yy) / 5.0) * 0.5 + 0.5) * 255).astype(np.uint8)
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, lsd_mode=3)
    want = g.po.Frame(ocfg(g, cfg)).line_extract(0, stripes)
    assert want[0] > 50
    for arena, falls_back in ((None, False), ("3", True)):
        if arena:
            monkeypatch.setenv("PLI_RX_ARENA", arena)
        fe = g.Frontend(cfg, dev=True)
        for _ in range(2):                                   # (the second call: look-free rounds)
            n, kl, ld = fe.line_extract(0, stripes)
            assert n == want[0] and kl.tobytes() == want[1].tobytes() and np.array_equal(ld, want[2]), arena
        assert (fe.lsd_round_stats()[2] > 0) == falls_back, (arena, fe.lsd_round_stats())
        monkeypatch.delenv("PLI_RX_ARENA", raising=False)


def test_stripes_in_a_256_frame_context_do_not_take_the_fallback(gpu):
    """The arena a context gets per image depends on how many images it is sized for.  The bench's context (256 frames) must keep
    the 16 words per pixel long parallel structures need: with 8 (round 3's last commit) every stripes image overflowed and was
    redone by the sequential grower — exact, ten times slower, and only tools/rounds_sweep.py noticed."""
    g = gpu
    W, H = 752, 480
    yy, xx = np.mgrid[0:H, 0:W]
    stripes = ((np.sin((xx + 2 * yy) / 5.0) * 0.5 + 0.5) * 255).astype(np.uint8)
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=256)
    fe = g.Frontend(cfg)
    batch = np.broadcast_to(stripes, (2, 2, H, W)).copy()
    for _ in range(2):
        recs = fe.batch_run_host(batch, stages=g.capi.RUN_LINES)
    st = fe.lsd_round_stats()
    assert st[2] == 0 and 0 < st[1] <= 8, "stripes in the 256-frame context: round stats %s" % (st,)
    m, kl, ld = g.po.Frame(ocfg(g, cfg)).line_extract(0, stripes)
    assert len(recs[0]["klL"]) == m and recs[0]["klL"].tobytes() == kl.tobytes()


def test_full_size_batch_properties(gpu):
    """BASELINE config at batch scale (96 EuRoC-size frames, 8 distinct pairs cycled), through size-independent
    properties: the two LSD schedules write byte-identical tables; a frame's record does not depend on its position
    in the batch or on its neighbours (duplicates are identical); one record of each distinct pair equals the oracle."""
    g = gpu
    W, H, F, U = 752, 480, 96, 8
    pairs = [g.synth.make_stereo_pair(40 + s, W, H) for s in range(U)]
    images = np.stack([np.stack(pairs[i % U]) for i in range(F)])
    tables = {}
    for mode in (1, 2, 3):
        cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=F, lsd_mode=mode)
        fe = g.Frontend(cfg)
        left, right = np.ascontiguousarray(images[:, 0]), np.ascontiguousarray(images[:, 1])
        table = np.zeros(fe.table_bytes(F), np.uint8)
        g.capi.check(fe.L.pli_batch_run_host(fe.h, F, g.capi.ptr(left), g.capi.ptr(right), W, W * H, g.capi.RUN_ALL,
                                             g.capi.ptr(table)))
        tables[mode] = (fe, cfg, table)
    fe, cfg, t1 = tables[1]
    assert np.array_equal(t1, tables[2][2]), "relaxation and sequential schedules differ"
    assert np.array_equal(t1, tables[3][2]), "tile-sequential relaxation and sequential schedules differ"
    rb = int(fe.layout.record_bytes)
    recs = t1.reshape(F, rb)
    for i in range(U, F):
        assert np.array_equal(recs[i], recs[i % U]), "record %d differs from its duplicate %d" % (i, i % U)
    for i in (0, U - 1):
        assert_frame_equal(g, fe.parse_record(t1, F - U + i), g.po.Frame(ocfg(g, cfg)), pairs[i][0], pairs[i][1], "pair %d" % i)


@pytest.mark.gpu
def test_mono_stream_through_batch_entry(gpu):
    """SURVEY §8f-4, monocular / RGB-D colour streams: consecutive frames take the two eye slots of a record
    (pointer arithmetic on the existing batch entry point, stereo stages off); every frame equals the oracle's
    single-image extraction (what Frame.cc:334 / :231 run)."""
    g = gpu
    W, H, NF = 376, 240, 6
    frames = np.stack([g.synth.make_stereo_pair(120 + i, W, H)[i & 1] for i in range(NF)])
    cfg = g.capi.default_config(W, H, orb_nfeatures=500, lsd_nfeatures=50, max_frames=NF // 2)
    fe = g.Frontend(cfg)
    out = fe.batch_run_mono_host(frames)
    fr = g.po.Frame(ocfg(g, cfg))
    for f in range(NF):
        n, kp, desc = fr.orb_extract(0, frames[f])
        assert n == len(out[f]["kp"]) and kp.tobytes() == out[f]["kp"].tobytes() and np.array_equal(desc, out[f]["desc"]), "orb frame %d" % f
        m, kl, ld = fr.line_extract(0, frames[f])
        assert m == len(out[f]["kl"]) and kl.tobytes() == out[f]["kl"].tobytes() and np.array_equal(ld, out[f]["ldesc"]), "lines frame %d" % f


@pytest.mark.gpu
@pytest.mark.parametrize("ang", [2.0, 45.0, 80.0, 86.5, 130.0])
def test_sequential_grower_angle_tolerances(gpu, ang):
    """The sequential grower decides alignment in vector form (dot^2 against cos^2(prec +- margin) |sum|^2) and takes the
    reference's expression inside the margin; from prec + margin >= 1.5 rad (85.9 deg) on only the reference's expression
    is used.  Both regimes and the narrow end against the oracle."""
    g = gpu
    W, H = 376, 240
    cfg = g.capi.default_config(W, H, orb_nfeatures=200, lsd_nfeatures=0, max_frames=1, lsd_mode=2, lsd_ang_th=ang)
    fe = g.Frontend(cfg)
    fr = g.po.Frame(ocfg(g, cfg))
    for seed in (90, 91):
        L, _ = g.synth.make_stereo_pair(seed, W, H)
        n, kl, ld = fe.line_extract(0, L)
        m, okl, old = fr.line_extract(0, L)
        assert n == m and kl.tobytes() == okl.tobytes() and np.array_equal(ld, old), "ang_th %g seed %d" % (ang, seed)


@pytest.mark.gpu
def test_large_batch_schedule(gpu):
    """The large-batch schedule of lsd_mode auto (>= 640 frames: sequential image waves, two per block) on 660 small
    frames: duplicates identical, distinct pairs equal to the oracle."""
    g = gpu
    W, H, F, U = 240, 180, 660, 6
    pairs = [g.synth.make_stereo_pair(70 + s, W, H) for s in range(U)]
    images = np.stack([np.stack(pairs[i % U]) for i in range(F)])
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=40, max_frames=F)
    fe = g.Frontend(cfg)
    left, right = np.ascontiguousarray(images[:, 0]), np.ascontiguousarray(images[:, 1])
    table = np.zeros(fe.table_bytes(F), np.uint8)
    g.capi.check(fe.L.pli_batch_run_host(fe.h, F, g.capi.ptr(left), g.capi.ptr(right), W, W * H, g.capi.RUN_ALL,
                                         g.capi.ptr(table)))
    recs = table.reshape(F, int(fe.layout.record_bytes))
    for i in range(U, F):
        assert np.array_equal(recs[i], recs[i % U]), "record %d differs from its duplicate %d" % (i, i % U)
    for i in range(U):
        assert_frame_equal(g, fe.parse_record(table, F - U + i), g.po.Frame(ocfg(g, cfg)), pairs[i][0], pairs[i][1], "pair %d" % i)


def test_bow_vocabulary_descent(gpu):
    """the per-feature part of DBoW2's transform (Frame::ComputeBoW, Frame.cc:858) on synthetic
    vocabularies of ORBvoc.txt shape (k = 10; ragged trees, stopped words), ORB and LBD descriptors of a real frame."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=0)
    fe = g.Frontend(cfg)
    L_, _ = g.synth.make_stereo_pair(5, W, H)
    _, _, orb = fe.orb_extract(0, L_)
    _, _, lbd = fe.line_extract(0, L_)
    for (k, depth, seed, ragged) in ((10, 4, 1, True), (10, 3, 2, False), (3, 6, 3, True), (64, 2, 4, False)):
        vk, vL, parent, is_leaf, vdesc, weight = g.synth.make_vocabulary(k, depth, seed, ragged)
        ov = g.po.Vocabulary(vk, vL, parent, is_leaf, vdesc, weight)
        gv = fe.vocab_create(vk, vL, parent, is_leaf, vdesc, weight)
        rng = np.random.default_rng(seed)
        near = vdesc[rng.integers(0, len(vdesc), 300)].copy()            # descriptors near tree nodes: deep, tie-rich descents
        near[:, rng.integers(0, 32)] ^= 0x11
        for feats in (orb, lbd, near, vdesc[:50]):
            for levelsup in (4, 1, 0, 9):
                word, wt, node = fe.bow_transform(gv, feats, levelsup)
                oword, owt, onode = ov.descend(feats, levelsup)
                assert np.array_equal(word, oword) and np.array_equal(node, onode) and wt.tobytes() == owt.tobytes()
            # the caller's accumulation (BowVector::addWeight in feature order, L1 norm in word order) on the device result
            word, wt, node = fe.bow_transform(gv, feats, 4)
            bow = {}
            for w_, v_ in zip(word.tolist(), wt.tolist()):
                if v_ > 0:
                    bow[w_] = bow.get(w_, 0.0) + v_
            norm = 0.0
            for w_ in sorted(bow):
                norm += abs(bow[w_])
            ws, vals = ov.bow_vector(feats, 4)
            assert sorted(bow) == ws.tolist()
            assert np.array([bow[w_] / norm for w_ in sorted(bow)], np.float64).tobytes() == vals.tobytes()
        fe.vocab_destroy(gv)
    w0, _, _ = fe.bow_transform(fe.vocab_create(*g.synth.make_vocabulary(10, 3, 7)), orb[:0], 4)
    assert len(w0) == 0


def test_rgbd_depth_association(gpu):
    """Frame::ComputeStereoFromRGBD (Frame.cc:1309) on the left keypoints."""
    g = gpu
    W, H = 640, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=1000, lsd_nfeatures=50)
    fe = g.Frontend(cfg)
    L, _ = g.synth.make_stereo_pair(17, W, H)
    n, kp, _ = fe.orb_extract(0, L)
    rng = np.random.default_rng(1)
    depth = rng.uniform(0.3, 8.0, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.2] = 0                                  # holes of the sensor
    depth[rng.random((H, W)) < 0.02] = -1
    ur, dp = fe.stereo_from_depth(depth)
    our, odp = g.po.stereo_from_depth(kp, depth, cfg.bf)
    assert n > 500 and ur[:n].tobytes() == our.tobytes() and dp[:n].tobytes() == odp.tobytes()
    assert 0.1 < (dp[:n] < 0).mean() < 0.4


def test_rectification_fused_into_ingest(gpu):
    """cv::remap(im, imRect, M1, M2, INTER_LINEAR) of the stereo driver (stereo_euroc.cc:166-167) applied
    by the ingest kernel: level 0 equals the oracle's remap of the raw image and everything downstream equals the
    oracle pipeline on the rectified pair."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=1000, lsd_nfeatures=100, max_frames=2)
    fe = g.Frontend(cfg)
    fe.debug_enable(True)
    maps = [g.synth.rectify_maps(e, W, H) for e in (0, 1)]
    for e in (0, 1):
        fe.set_rectify_maps(e, *maps[e])
    frames = [g.synth.make_stereo_pair(s, W, H) for s in (21, 22)]
    recs = fe.batch_run_host(np.stack([np.stack(f) for f in frames]))
    for i, (rec, (L, R)) in enumerate(zip(recs, frames)):
        Lr, Rr = g.po.remap_linear(L, *maps[0]), g.po.remap_linear(R, *maps[1])
        assert (Lr != L).mean() > 0.5                                    # the maps really move pixels
        for eye, ref in ((0, Lr), (1, Rr)):
            lvl0 = fe.debug_fetch(2 * i + eye, g.capi.DBG_PYRAMID_LEVEL, 0)
            assert np.array_equal(lvl0, ref.ravel()), "frame %d eye %d level 0" % (i, eye)
        assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), Lr, Rr, "rectified frame %d" % i)
    # per-call path, and maps with out-of-image coordinates (constant-0 border)
    mx, my = maps[0]
    n, kp, desc = fe.orb_extract(0, frames[0][0])
    fr = g.po.Frame(ocfg(g, cfg))
    on, okp, odesc = fr.orb_extract(0, g.po.remap_linear(frames[0][0], mx, my))
    assert n == on and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)
    fe.set_rectify_maps(0, mx * np.float32(1.3) - 40, my * np.float32(1.3) - 60)
    n, kp, desc = fe.orb_extract(0, frames[0][0])
    ref = g.po.remap_linear(frames[0][0], mx * np.float32(1.3) - 40, my * np.float32(1.3) - 60)
    assert (ref == 0).mean() > 0.05
    assert np.array_equal(fe.debug_fetch(0, g.capi.DBG_PYRAMID_LEVEL, 0), ref.ravel())
    # removing the maps restores the plain copy
    fe.set_rectify_maps(0, None, None)
    fe.orb_extract(0, frames[0][0])
    assert np.array_equal(fe.debug_fetch(0, g.capi.DBG_PYRAMID_LEVEL, 0), frames[0][0].ravel())


def test_stereo_maxd_inf_switch(gpu):
    g = gpu
    W, H = 376, 240
    L, R = g.synth.make_stereo_pair(13, W, H)
    cfg = g.capi.default_config(W, H, orb_nfeatures=500, lsd_nfeatures=40, stereo_maxd_inf=1)
    rec = g.Frontend(cfg).batch_run_host(np.stack([L, R])[None])[0]
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, R, "maxD=inf")


# ---------------------------------------------------------------------------------------------
# config 5 size: 3840x2160, 4000 kp + 500 lines (ORB compared directly; lines through properties)
# ---------------------------------------------------------------------------------------------
def test_config5_4k(gpu):
    g = gpu
    W, H = 3840, 2160
    cfg = g.capi.default_config(W, H, orb_nfeatures=4000, lsd_nfeatures=500, max_frames=1)
    fe = g.Frontend(cfg)
    L, R = g.synth.make_stereo_pair(21, W, H)
    fr = g.po.Frame(ocfg(g, cfg))
    for eye, img in ((0, L), (1, R)):
        n, kp, desc = fe.orb_extract(eye, img)
        on, okp, odesc = fr.orb_extract(eye, img)
        assert n == on and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)
    ur, dp = fe.compute_stereo_matches()
    our, odp, _, _ = fr.stereo_points()
    assert ur[:len(our)].tobytes() == our.tobytes() and dp[:len(our)].tobytes() == odp.tobytes()
    # idempotence at full size: the same image gives the same tables on a second pass
    n2, kp2, desc2 = fe.orb_extract(1, R)
    assert kp2.tobytes() == kp.tobytes() and np.array_equal(desc2, desc)
    # LSD/LBD at 4K (4608 x 2592 scaled image, ~10 M seeds) against the oracle
    for eye, img in ((0, L), (1, R)):
        m, kl, ld = fe.line_extract(eye, img)
        om, okl, old = fr.line_extract(eye, img)
        assert m == om == 500 and kl.tobytes() == okl.tobytes() and np.array_equal(ld, old)
    disp, le = fe.compute_stereo_matches_lines()
    odisp, ole, _ = fr.stereo_lines()
    assert disp[:len(odisp)].tobytes() == odisp.tobytes() and le[:len(ole)].tobytes() == ole.tobytes()
    assert (odisp[:, 0] >= 0).sum() > 20


def test_pipelined_host_entry_point(gpu):
    """pli_batch_submit_host / pli_batch_wait (pinned staging, copy streams): three batches in flight through two slots give
    the tables of the synchronous entry point, in order."""
    g = gpu
    W, H, F = 376, 240, 3
    cfg = g.capi.default_config(W, H, orb_nfeatures=400, lsd_nfeatures=40, max_frames=F)
    fe = g.Frontend(cfg)
    batches = [np.stack([np.stack(g.synth.make_stereo_pair(60 + 3 * b + i, W, H)) for i in range(F)]) for b in range(3)]
    want = []
    for imgs in batches:
        left, right = np.ascontiguousarray(imgs[:, 0]), np.ascontiguousarray(imgs[:, 1])
        t = np.zeros(fe.table_bytes(F), np.uint8)
        g.capi.check(fe.L.pli_batch_run_host(fe.h, F, g.capi.ptr(left), g.capi.ptr(right), W, W * H, g.capi.RUN_ALL, g.capi.ptr(t)))
        want.append(t)
    n = W * H
    lefts = [fe.pinned(F * n).reshape(F, n) for _ in range(3)]
    rights = [fe.pinned(F * n).reshape(F, n) for _ in range(3)]
    tabs = [fe.pinned(fe.table_bytes(F)) for _ in range(3)]
    for b, imgs in enumerate(batches):
        lefts[b][:] = imgs[:, 0].reshape(F, n)
        rights[b][:] = imgs[:, 1].reshape(F, n)
        tabs[b][:] = 0xEE
    for b in range(3):
        fe.host_submit(F, lefts[b], rights[b], tabs[b])          # the third submit waits for the first slot
    fe.host_wait_all()
    for b in range(3):
        assert np.array_equal(tabs[b], want[b]), "batch %d" % b
    # strided host images (row stride > width) take the 2D-copy path
    wide = np.zeros((F, 2, H, W + 24), np.uint8)
    wide[:, :, :, :W] = batches[0]
    tp = fe.pinned(fe.table_bytes(F))
    g.capi.check(fe.L.pli_batch_submit_host(fe.h, F, g.capi.ptr(wide[:, 0]), C.c_void_p(wide.ctypes.data + H * (W + 24)), W + 24,
                                            2 * H * (W + 24), g.capi.RUN_ALL, g.capi.ptr(tp)))
    fe.host_wait()
    assert np.array_equal(tp, want[0])


def test_batch_track_config3(gpu):
    """pli_batch_track on the device tables of a batch of consecutive 1280x720 frames (BASELINE config 3): frame i against
    frame i-1 — SearchByProjection(CurrentFrame, LastFrame) with the projection done on the device, and match() of the line
    descriptors — equals the oracle's restatement, for the three motion cases (neutral window, forward, backward)."""
    import torch
    g = gpu
    W, H, F = 1280, 720, 4
    cfg = g.capi.default_config(W, H, orb_nfeatures=2000, lsd_nfeatures=200, max_frames=F)
    fe = g.Frontend(cfg)
    frames = [g.synth.make_stereo_pair(8, W, H, t=t) for t in range(F)]
    imgs = np.stack([np.stack(f) for f in frames])
    left, right = np.ascontiguousarray(imgs[:, 0]), np.ascontiguousarray(imgs[:, 1])
    table = np.zeros(fe.table_bytes(F), np.uint8)
    g.capi.check(fe.L.pli_batch_run_host(fe.h, F, g.capi.ptr(left), g.capi.ptr(right), W, W * H, g.capi.RUN_ALL, g.capi.ptr(table)))
    recs = [fe.parse_record(table, f) for f in range(F)]

    def pose(t, tz):
        a = np.deg2rad(0.5 * t)
        T = np.eye(4, dtype=np.float64)
        T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        T[:3, 3] = [-0.02 * t, -0.007 * t, tz]
        return T[:3].astype(np.float32)
    poses = np.stack([pose(0, 0.0), pose(1, 0.01), pose(2, -0.35), pose(3, 0.05)])      # neutral, forward (tlc.z > mb), backward
    tp = fe.track_params(th=15.0, mono=False, check_orientation=True, nnr_lines=0.9, cx=W / 2.0 - 3.5, cy=H / 2.0 + 2.25)
    tl = fe.track_layout()
    d_table = torch.from_numpy(table).cuda()
    d_poses = torch.from_numpy(poses.reshape(-1)).cuda()
    d_track = torch.zeros(F * tl.record_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    fe.batch_track_device(F, d_table.data_ptr(), d_poses.data_ptr(), tp, d_track.data_ptr())
    fe.sync()
    track = d_track.cpu().numpy()
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)
    bounds = (0.0, float(W), 0.0, float(H))
    seen = set()
    for f in range(1, F):
        last, cur = recs[f - 1], recs[f]
        tr = fe.parse_track(track, f)
        q = g.po.track_queries(last["kpL"], last["depth"], poses[f - 1], poses[f], tp.fx, tp.fy, tp.cx, tp.cy, tp.bf, tp.th, False, sf)
        v = q["valid"] > 0
        oc = last["kpL"]["octave"][v]
        if np.array_equal(q["min_level"][v], oc - 1) and np.array_equal(q["max_level"][v], oc + 1):
            seen.add("neutral")
        elif np.array_equal(q["min_level"][v], oc) and (q["max_level"][v] < 0).all():
            seen.add("forward")
        elif (q["min_level"][v] == 0).all() and np.array_equal(q["max_level"][v], oc):
            seen.add("backward")
        on, obest = g.po.search_by_projection(q, last["descL"], cur["kpL"], cur["descL"], cur["uright"], bounds, True)
        assert tr["counts"][0] == len(last["kpL"]) and tr["counts"][1] == on, (f, tr["counts"], on)
        assert np.array_equal(tr["best"], obest), "frame %d point track matches" % f
        ln, lm = g.po.match_lines(last["ldescL"], cur["ldescL"], 0.9, True)
        assert tr["counts"][2] == len(last["ldescL"]) and tr["counts"][3] == ln and np.array_equal(tr["lines"], lm), "frame %d lines" % f
        assert on > 50 and ln > 10
    assert seen == {"neutral", "forward", "backward"}, seen
    # a shard of the stream with its 1-frame halo (frame-to-frame matching across shard borders): the table
    # [record of frame 1 | records of frames 2, 3] with the poses of frames 1..3 gives frames 2 and 3 the tracks of the full run
    rb = int(fe.layout.record_bytes)
    d_halo = torch.from_numpy(np.concatenate([table[1 * rb:2 * rb], table[2 * rb:4 * rb]])).cuda()
    d_poses_h = torch.from_numpy(poses[1:4].reshape(-1)).cuda()
    d_track_h = torch.zeros(3 * tl.record_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    fe.batch_track_device(3, d_halo.data_ptr(), d_poses_h.data_ptr(), tp, d_track_h.data_ptr())
    fe.sync()
    th = d_track_h.cpu().numpy()
    for f in (2, 3):
        a, b = fe.parse_track(track, f), fe.parse_track(th, f - 1)
        assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["best"], b["best"]) and np.array_equal(a["lines"], b["lines"])


@pytest.mark.parametrize("flags", [0, 1, 2, 4, 8, 6, 15])
def test_parity_flags_every_variant(gpu, flags):
    """The arithmetic the reference leaves to its toolchain (PLI_PARITY_*: cosf vs correctly rounded cos per call site, CV_8UC1 vs
    CV_64FC1 LineSegmentDetector): oracle and kernels take the same flags and agree bit for bit in every variant, in all three
    LSD schedules; the scaled image / angle map are compared stage by stage for the LSD pipelines."""
    g = gpu
    W, H = 376, 240
    L, R = g.synth.make_stereo_pair(21, W, H)
    for mode in (2, 3, 1):
        cfg = g.capi.default_config(W, H, orb_nfeatures=500, lsd_nfeatures=0, max_frames=1, lsd_mode=mode, parity_flags=flags)
        fe = g.Frontend(cfg)
        fe.debug_enable(True)
        rec = fe.batch_run_host(np.stack([L, R])[None])[0]
        fr = g.po.Frame(ocfg(g, cfg))
        assert_frame_equal(g, rec, fr, L, R, "flags %d mode %d" % (flags, mode))
        if mode == 2:
            n, kl, ld = fe.line_extract(0, L)
            if flags & g.capi.PARITY_LSD_F64:
                assert np.array_equal(fe.debug_fetch(0, g.capi.DBG_LSD_SCALED).view(np.float64), fr.lsd_scaled64(0).ravel())
            else:
                assert np.array_equal(fe.debug_fetch(0, g.capi.DBG_LSD_SCALED), fr.lsd_scaled(0).ravel())
            assert np.array_equal(fe.debug_fetch(0, g.capi.DBG_LSD_ANGLE).view(np.float32), fr.lsd_angle(0).ravel())
    assert len(rec["klL"]) > 100 and len(rec["kpL"]) > 300


def test_parity_flags_change_results(gpu):
    """The switches are not no-ops: the two LSD pipelines give different segments on the same image."""
    g = gpu
    W, H = 376, 240
    L, R = g.synth.make_stereo_pair(21, W, H)
    out = {}
    for flags in (0, 8):
        cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, parity_flags=flags)
        out[flags] = g.Frontend(cfg).batch_run_host(np.stack([L, R])[None])[0]["klL"]
    assert out[0].tobytes() != out[8].tobytes()


# ---------------------------------------------------------------------------------------------
# fisheye stereo — lapping-area ordering of the extractor + ComputeStereoFishEyeMatches
# ---------------------------------------------------------------------------------------------
# a KannalaBrandt8 camera whose r(theta) is the Taylor polynomial of tan(theta): pinhole-like, so that the synthetic
# stream's horizontal disparities are consistent with the epipolar geometry of a pure x-baseline
PINHOLE_KB8 = [435.2, 435.2, 367.2, 252.2, 1 / 3, 2 / 15, 17 / 315, 62 / 2835]
TUMVI_KB8 = ([190.978477, 190.973307, 254.931706, 256.897442, 0.00348238940, 0.000715034845, -0.00205323614, 0.000202936736],
             [190.442369, 190.434438, 252.597254, 254.917230, 0.00340031805, 0.00176627874, -0.00266312161, 0.000329951911])


def test_fisheye_lapping_order_and_stereo(gpu):
    """ORBextractor.cc:1135-1144 + Frame.cc:1577-1618 through pli_orb_extract_lapping / pli_stereo_fisheye: table order, mono
    counts, mvLeftToRightMatch / mvRightToLeftMatch / mvDepth / mvStereo3Dpoints equal to the oracle's, bit for bit."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=0)
    fe = g.Frontend(cfg)
    fr = g.po.Frame(ocfg(g, cfg))
    sigma2 = fr.level_sigma2()
    a = np.deg2rad(0.05)
    Rlr = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    tlr = np.array([0.11, 0.0004, -0.0003], np.float32)
    total = 0
    for seed, lapL, lapR in ((5, (90, 700), (30, 640)), (6, (0, 751), (0, 751)), (7, (300, 420), (280, 400)), (8, (0, 0), (0, 0))):
        L, R = g.synth.make_stereo_pair(seed, W, H)
        tabs = []
        for eye, img, lap in ((0, L, lapL), (1, R, lapR)):
            n, mono, kp, desc = fe.orb_extract_lapping(eye, img, lap)
            on, okp, odesc = fr.orb_extract(eye, img)
            order, omono = g.po.lapping_order(okp, lap[0], lap[1])
            assert n == on and mono == omono
            assert kp.tobytes() == okp[order].tobytes() and np.array_equal(desc, odesc[order])
            tabs.append((kp, desc, mono))
        (kpL, dL, mL), (kpR, dR, mR) = tabs
        for cams in ((PINHOLE_KB8, PINHOLE_KB8), TUMVI_KB8):
            nm, l2r, r2l, depth, p3d = fe.stereo_fisheye(cams[0], cams[1], Rlr, tlr, len(kpL), len(kpR))
            onm, ol2r, or2l, odepth, op3d = g.po.stereo_fisheye(kpL, dL, mL, kpR, dR, mR, cams[0], cams[1], Rlr, tlr, sigma2)
            assert nm == onm and np.array_equal(l2r, ol2r) and np.array_equal(r2l, or2l)
            assert depth.tobytes() == odepth.tobytes() and p3d.tobytes() == op3d.tobytes()
            assert (l2r[:mL] == -1).all() and (depth[l2r < 0] == -1).all() and (depth[l2r >= 0] > 0.0001).all()
            if cams[0] is PINHOLE_KB8:
                total += nm
    assert total > 100                                  # the pinhole-like pair triangulates a real share of the lapping area


def test_fisheye_stereo_on_constructed_tables(gpu):
    """pli_stereo_fisheye_tables against the oracle on constructed two-view tables (known 3-D points through two TUM-VI-like
    fisheye cameras, pixel noise, shuffled right table, a right keypoint wanted by two left ones), several seeds."""
    g = gpu
    from helpers_fisheye import fisheye_tables as _fisheye_tables
    cfg = g.capi.default_config(512, 512, orb_nfeatures=500, lsd_nfeatures=0)
    fe = g.Frontend(cfg)
    sigma2 = g.po.Frame(ocfg(g, cfg)).level_sigma2()
    a = np.deg2rad(2.0)
    R = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
    t = [0.101, 0.002, -0.001]
    for seed, noise, n in ((1, 0.0, 300), (2, 0.7, 300), (3, 1.5, 1000), (4, 3.0, 64)):
        P1, kpL, dL, kpR, dR, mono, perm = _fisheye_tables(g.po, TUMVI_KB8[0], TUMVI_KB8[1], R, t, n=n, seed=seed, noise=noise)
        # two left keypoints with the same descriptor: both take the same right keypoint, the later one keeps mvRightToLeftMatch
        kpL = np.concatenate([kpL, kpL[mono + 3:mono + 4]]); dL = np.concatenate([dL, dL[mono + 3:mono + 4]])
        got = fe.stereo_fisheye_tables(kpL, dL, mono, kpR, dR, mono, TUMVI_KB8[0], TUMVI_KB8[1], R, t)
        exp = g.po.stereo_fisheye(kpL, dL, mono, kpR, dR, mono, TUMVI_KB8[0], TUMVI_KB8[1], R, t, sigma2)
        assert got[0] == exp[0] and np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2])
        assert got[3].tobytes() == exp[3].tobytes() and got[4].tobytes() == exp[4].tobytes()
        if noise == 0.0:
            assert got[0] == n + 1 and got[2][got[1][mono + 3]] == len(kpL) - 1
    # empty lapping areas and a single right candidate
    got = fe.stereo_fisheye_tables(kpL, dL, len(kpL), kpR, dR, mono, TUMVI_KB8[0], TUMVI_KB8[1], R, t)
    assert got[0] == 0 and (got[1] == -1).all() and (got[3] == -1).all()
    got = fe.stereo_fisheye_tables(kpL, dL, mono, kpR, dR, len(kpR) - 1, TUMVI_KB8[0], TUMVI_KB8[1], R, t)
    assert got[0] == 0


@pytest.mark.parametrize("flags", [None, 0])
def test_sequential_grower_dev_switches(gpu, flags, monkeypatch):
    """The measured-and-shelved variants of the sequential LSD grower stay exact: region2rect off the wave (k_lsd_rect,
    PLI_LSD_RECT_OFFLOAD), the plain (non-speculative) grower (PLI_LSD_SPEC=0) and the unfused CV_64F front (PLI_LSD_NOFUSE) —
    in both detector pipelines (flags None = default CV_64F, 0 = CV_8U)."""
    g = gpu
    W, H = 752, 480
    over = {} if flags is None else {"parity_flags": flags}
    L, R = g.synth.make_stereo_pair(41, W, H)
    ocfg_ = None
    want = None
    for env in ({}, {"PLI_LSD_RECT_OFFLOAD": "1"}, {"PLI_LSD_SPEC": "0"}, {"PLI_LSD_NOFUSE": "1"}):
        for k in ("PLI_LSD_RECT_OFFLOAD", "PLI_LSD_SPEC", "PLI_LSD_NOFUSE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, lsd_mode=2, **over)
        fe = g.Frontend(cfg, dev=True)
        m, kl, ld = fe.line_extract(0, L)
        if want is None:
            fr = g.po.Frame(ocfg(g, cfg))
            want = fr.line_extract(0, L)
            assert want[0] > 500
        assert m == want[0] and kl.tobytes() == want[1].tobytes() and np.array_equal(ld, want[2]), env


@pytest.mark.parametrize("W,H", [(752, 480), (1600, 1200)])
def test_tile_relaxation_dev_switches(gpu, W, H, monkeypatch):
    """The tile-sequential relaxation with each of its round-2 shortcuts switched off gives the same lines as with them:
    the walk over all seeds instead of the per-tile dirty lists (PLI_TX_NODIRTYLIST), k_tx_diff2 + k_tx_prep instead of the
    fused round 2 (PLI_TX_NOFUSE2), k_rx_diff + k_tx_mark instead of k_tx_diffmark (PLI_TX_NOFUSEDM), k_rx_mark instead of k_tx_mark
    (PLI_TX_OLDMARK), the conservative regrowth rules
    (PLI_TX_BOXRULE, PLI_TX_CELLRULE), every cell compared every round (PLI_TX_FULLDIFF), and the one-block ordered-list scan
    (PLI_LSD_SCAN1; the larger shape has enough chunks for the grouped scan)."""
    g = gpu
    L, R = g.synth.make_stereo_pair(77, W, H)
    switches = ("PLI_TX_NODIRTYLIST", "PLI_TX_NOFUSE2", "PLI_TX_NOFUSEDM", "PLI_TX_OLDMARK", "PLI_TX_BOXRULE", "PLI_TX_CELLRULE",
                "PLI_TX_FULLDIFF", "PLI_LSD_SCAN1")
    want = None
    for on in (None,) + switches:
        for k in switches:
            monkeypatch.delenv(k, raising=False)
        if on:
            monkeypatch.setenv(on, "1")
        cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, lsd_mode=3)
        fe = g.Frontend(cfg, dev=True)
        m, kl, ld = fe.line_extract(0, L)
        if want is None:
            fr = g.po.Frame(ocfg(g, cfg))
            want = fr.line_extract(0, L)
            assert want[0] > 500
        assert m == want[0] and kl.tobytes() == want[1].tobytes() and np.array_equal(ld, want[2]), on


def test_truncation_is_flagged_not_silent(gpu):
    """A max_lines smaller than the number of segments that pass the length cut: the record carries the truncation flag and
    the per-call entry point returns PLI_ERR_CAPACITY; with room enough the flag is clear and the result is the oracle's."""
    g = gpu
    W, H = 752, 480
    L, R = g.synth.make_stereo_pair(9, W, H)
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=50, max_lines=64)
    fe = g.Frontend(cfg)
    rec = fe.batch_run_host(np.stack([L, R])[None])[0]
    assert rec["truncated"][:2].tolist() == [1, 1] and rec["truncated"][2:].tolist() == [0, 0]
    with pytest.raises(g.capi.PliError) as ei:
        fe.line_extract(0, L)
    assert ei.value.status == -3                          # PLI_ERR_CAPACITY
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=50)
    fe = g.Frontend(cfg)
    rec = fe.batch_run_host(np.stack([L, R])[None])[0]
    assert rec["truncated"].tolist() == [0, 0, 0, 0]
    m, kl, ld = fe.line_extract(0, L)
    om, okl, old = g.po.Frame(ocfg(g, cfg)).line_extract(0, L)
    assert m == om == 50 and kl.tobytes() == okl.tobytes()


def test_look_free_rounds_and_device_side_fallback(gpu, monkeypatch):
    """After its first call a context launches the relaxation rounds without a host look (pli_lsd_round_stats).  With a plan that
    is deliberately too short (dev switch PLI_RX_PLAN) the images that have not settled are redone on the device by the sequential
    grower: slow, and still the oracle's result, byte for byte."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=4)
    fe = g.Frontend(cfg, dev=True)
    pairs = [g.synth.make_stereo_pair(60 + i, W, H) for i in range(4)]
    imgs = np.stack([np.stack(p) for p in pairs])
    first = fe.batch_run_host(imgs)                          # the late rounds in the persistent tail kernel: no look, no plan
    st = fe.lsd_round_stats()
    assert st[0] == -1 and st[1] >= 4 and st[2] == 0, st
    monkeypatch.setenv("PLI_TX_TAIL", "0")
    again = fe.batch_run_host(imgs)                          # without the tail: planned rounds, look-free
    st = fe.lsd_round_stats()
    assert st[0] >= st[1] > 0 and st[2] == 0, st             # planned >= needed, nobody took the slow path
    monkeypatch.delenv("PLI_TX_TAIL")
    monkeypatch.setenv("PLI_RX_PLAN", "3")
    short = fe.batch_run_host(imgs)                          # three rounds are not enough: the device-side fallback
    st = fe.lsd_round_stats()
    assert st[0] == 3 and st[1] == -1 and st[2] >= 1, st
    monkeypatch.delenv("PLI_RX_PLAN")
    later = fe.batch_run_host(imgs)
    for f, (L, R) in enumerate(pairs):
        for recs, what in ((first, "tail"), (again, "planned rounds"), (short, "fallback"), (later, "after the fallback")):
            if f == 0 or recs is short:
                assert_frame_equal(g, recs[f], g.po.Frame(ocfg(g, cfg)), L, R, "%s frame %d" % (what, f))
            else:
                for k in ("klL", "klR", "ldescL", "ldescR", "disp", "le"):
                    assert recs[f][k].tobytes() == first[f][k].tobytes(), (what, f, k)


def test_key_mode_and_rank_mode_and_the_segment_list_overflow(gpu, monkeypatch):
    """The tile relaxation names its regions by KEYS (gradient bin | seed pixel: no ordered list is built, lsd_tile.hip k_tx_sort)
    for images of up to 2^20 scaled pixels, by their ranks in the ordered list otherwise (dev switch PLI_TX_KEYS=0).  Both give the
    oracle's lines; and so does an image whose candidate segments do not fit the list k_tx_emit_sorted can order (forced with the
    test switch PLI_TX_EMITCAP): it is counted as a slow-path image and redone by the sequential grower from an ordered list
    built for it alone."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=0, max_frames=3)
    pairs = [g.synth.make_stereo_pair(90 + i, W, H) for i in range(3)]
    imgs = np.stack([np.stack(p) for p in pairs])
    out = {}
    for what, env in (("keys", {}), ("ranks", {"PLI_TX_KEYS": "0"}), ("overflow", {"PLI_TX_EMITCAP": "100"})):
        for k in ("PLI_TX_KEYS", "PLI_TX_EMITCAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        fe = g.Frontend(cfg, dev=True)
        fe.batch_run_host(imgs)
        out[what] = fe.batch_run_host(imgs)                  # (the second call: look-free rounds)
        st = fe.lsd_round_stats()
        assert (st[2] >= 1) == (what == "overflow"), (what, st)
    assert len(out["keys"][0]["klL"]) > 100
    for f, (L, R) in enumerate(pairs):
        assert_frame_equal(g, out["keys"][f], g.po.Frame(ocfg(g, cfg)), L, R, "key mode frame %d" % f)
        for what in ("ranks", "overflow"):
            for k in ("klL", "klR", "ldescL", "ldescR", "disp", "le"):
                assert out[what][f][k].tobytes() == out["keys"][f][k].tobytes(), (what, f, k)


def test_round1_owner_word_forms(gpu, monkeypatch):
    """Round 1 of the tile relaxation keeps owner_1 in the fourth word of the pixel records (lsd_tile.hip, "PACKED ROUND 1"): written by
    the front pass as the gradient norm in fixed point (LAZY ids, key mode: the default), or by k_tx_sort as the pixel's id (rank mode;
    PLI_TX_PACK1=1), or not at all (PLI_TX_PACK1=0: the owner plane, rounds 1-4's form).  All of them give the oracle's lines; and so
    does the LAZY form when EVERY unclaimed pixel is sent through the exact double-plane test (PLI_TX_LAZY_MARGIN wide) or when the
    margin around a bin boundary is at its floor.  Also here: the later rounds' bookkeeping on cell lists instead of block walks."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=2)
    pairs = [g.synth.make_stereo_pair(190 + i, W, H) for i in range(2)]
    imgs = np.stack([np.stack(p) for p in pairs])
    keys = ("PLI_TX_PACK1", "PLI_TX_LAZY_MARGIN", "PLI_TX_KEYS", "PLI_TX_TAIL", "PLI_TX_CELLS", "PLI_TX_HOT")
    out = {}
    # (round 6: on the hot records the default is sort-written ids; "lazy" names the default of the run, whatever it is)
    for what, env in (("lazy", {}), ("lazy_on_hot", {"PLI_TX_PACK1": "2"}), ("lazy_rec16", {"PLI_TX_HOT": "0"}),
                      ("lazy_all_exact", {"PLI_TX_PACK1": "2", "PLI_TX_LAZY_MARGIN": "2000000"}), ("lazy_margin4", {"PLI_TX_PACK1": "2", "PLI_TX_LAZY_MARGIN": "4"}),
                      ("lazy_all_exact_rec16", {"PLI_TX_HOT": "0", "PLI_TX_LAZY_MARGIN": "2000000"}),
                      ("sort_written_keys", {"PLI_TX_PACK1": "1"}), ("sort_written_keys_rec16", {"PLI_TX_PACK1": "1", "PLI_TX_HOT": "0"}), ("sort_written_ranks", {"PLI_TX_KEYS": "0"}),
                      ("owner_plane", {"PLI_TX_PACK1": "0"}), ("owner_plane_ranks_no_tail", {"PLI_TX_PACK1": "0", "PLI_TX_KEYS": "0", "PLI_TX_TAIL": "0"}),
                      # rounds >= 3 on cell lists (k_tx_cells + a wave per listed cell; the default from a million cells per call up), up to
                      # the tail kernel's round 8 and without the tail kernel
                      ("cell_lists", {"PLI_TX_CELLS": "1"}), ("cell_lists_no_tail_ranks", {"PLI_TX_CELLS": "1", "PLI_TX_TAIL": "0", "PLI_TX_KEYS": "0"})):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        fe = g.Frontend(cfg, dev=True)
        fe.batch_run_host(imgs)
        out[what] = fe.batch_run_host(imgs)
        assert fe.lsd_round_stats()[2] == 0, (what, fe.lsd_round_stats())      # nobody on the slow path
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    assert len(out["lazy"][0]["klL"]) > 100
    for f, (L, R) in enumerate(pairs):
        assert_frame_equal(g, out["lazy"][f], g.po.Frame(ocfg(g, cfg)), L, R, "lazy ids frame %d" % f)
        for what in out:
            for k in ("klL", "klR", "ldescL", "ldescR", "disp", "le"):
                assert out[what][f][k].tobytes() == out["lazy"][f][k].tobytes(), (what, f, k)


def test_the_product_library_has_no_lane_relaxation_and_ignores_dev_switches(gpu, monkeypatch):
    """libpli_frontend.so (what an integrator links) refuses lsd_mode 1 — round 1's schedule lives in the development build — and does not
    read the development switches: with PLI_RX_PLAN=1 in the environment (a plan far too short) the product library still settles every
    image in its persistent kernel, where the development build does what the switch says."""
    g = gpu
    W, H = 376, 240
    with pytest.raises(g.capi.PliError) as e:
        g.Frontend(g.capi.default_config(W, H, lsd_mode=1), dev=False)
    assert e.value.status == -1
    L, R = g.synth.make_stereo_pair(5, W, H)
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1)
    monkeypatch.setenv("PLI_RX_PLAN", "1")
    stats = {}
    for dev in (False, True):
        fe = g.Frontend(cfg, dev=dev)
        fe.batch_run_host(np.stack([L, R])[None])
        rec = fe.batch_run_host(np.stack([L, R])[None])[0]
        stats[dev] = fe.lsd_round_stats()
        assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, R, "dev=%s" % dev)
    assert stats[False][0] == -1 and stats[False][2] == 0, stats      # product: the tail kernel, nobody on the slow path
    assert stats[True][0] == 1 and stats[True][2] >= 1, stats          # development build: one planned round, the device-side fallback


@pytest.mark.parametrize("W", [640, 641, 666, 752])          # scaled widths 768 / 769 / 799 / 902: pads of 0 / 15 / 1 / 10 columns
def test_the_row_pitch_of_the_lsd_planes_does_not_show(gpu, W, monkeypatch):
    """The detector's per-pixel planes have a row pitch of the scaled width rounded up to 16 pixels (the pad columns are undefined
    pixels the front pass writes).  Nothing may depend on it: the oracle's lines in the default schedule (key mode), the sequential
    grower, the CV_8U pipeline and a debug context (rank mode, the unfused front pass, every debug plane in the caller's layout) — and
    the same bytes from a development build that runs on the true width (PLI_LSD_NOPAD)."""
    g = gpu
    capi = g.capi
    H = 200
    L, R = g.synth.make_stereo_pair(31 + W, W, H)
    for mode, flags, debug in ((0, None, False), (2, None, False), (0, capi.default_config(64, 64).parity_flags & ~capi.PARITY_LSD_F64, False), (0, None, True)):
        kw = dict(orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, lsd_mode=mode)
        if flags is not None:
            kw["parity_flags"] = flags
        cfg = capi.default_config(W, H, **kw)
        fr = g.po.Frame(ocfg(g, cfg))
        want = fr.line_extract(0, L)
        assert want[0] > 50
        got = {}
        for nopad in (False, True):
            if nopad:
                monkeypatch.setenv("PLI_LSD_NOPAD", "1")
            else:
                monkeypatch.delenv("PLI_LSD_NOPAD", raising=False)
            fe = g.Frontend(cfg, dev=True)
            if debug:
                fe.debug_enable(True)
            m, kl, ld = fe.line_extract(0, L)
            assert m == want[0] and kl.tobytes() == want[1].tobytes() and np.array_equal(ld, want[2]), (mode, flags, debug, nopad)
            if debug:
                oang = fr.lsd_angle(0).ravel()
                assert np.array_equal(fe.debug_fetch(0, capi.DBG_LSD_ANGLE).view(np.float32), oang), nopad
                assert np.array_equal(fe.debug_fetch(0, capi.DBG_LSD_SCALED).view(np.float64), fr.lsd_scaled64(0).ravel()), nopad
                raw = fe.debug_fetch(0, capi.DBG_LSD_ORDER).view(np.int32)
                oo = fr.lsd_order(0)
                assert np.array_equal(raw[1:1 + raw[0]], oo[oang[oo] != -1024]), nopad
                got[nopad] = fe.debug_fetch(0, capi.DBG_LSD_OWNER).tobytes()
        if debug:
            assert got[False] == got[True]            # the owner map of the fixed point, in the caller's layout
        monkeypatch.delenv("PLI_LSD_NOPAD", raising=False)


def test_hot_records_hardware_trig_error_is_inside_the_budget(gpu):
    """Round 1 on the 8-byte hot records runs its vector filter on v_cos_f32 / v_sin_f32 (lsd_tile.hip "HOT RECORDS"); the error
    budget that keeps the filter's decisions inside the margin of the exact expression assumes |error| < 4e-6 for every float angle."""
    g = gpu
    fe = g.Frontend(g.capi.default_config(128, 128))
    err = fe.selftest_hot_trig()
    assert 0.0 < err < 4e-6, err


def test_hot_records(gpu, monkeypatch):
    """The tile relaxation on 8-byte hot records {level-line angle, round 1's owner word} — every round (the default: PLI_TX_HOT=2, the
    16-byte records are not even written) or round 1 only (1), lazy ids / key mode and sort-written ids / rank mode: the oracle's lines,
    byte for byte — also when a wide filter margin sends a third of the candidates
    through the exact expression (and so through the exact sums folded from the 16-byte records in the middle of a region's growth),
    when region2rect recomputes the exact sums for EVERY region, with 32-pixel tiles, without the tail kernel, on the stripes image
    (regions of thousands of pixels: the queue's overflow blocks and the resynchronisation of the filter's sums), and against the
    16-byte records of the same build."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=3)
    pairs = [g.synth.make_stereo_pair(290 + i, W, H) for i in range(2)]
    yy, xx = np.mgrid[0:H, 0:W]
    stripes = (((xx + yy // 3) // 7) % 2 * 170 + 40).astype(np.uint8)
    pairs.append((stripes, np.ascontiguousarray(stripes[:, ::-1])))
    imgs = np.stack([np.stack(p) for p in pairs])
    keys = ("PLI_TX_HOT", "PLI_TX_KEYS", "PLI_ALIGN_MARGIN_DEG", "PLI_RECT_APPROX_BAND", "PLI_TX_TS", "PLI_TX_TAIL", "PLI_RX_ARENA", "PLI_TX_HOT_BAND2", "PLI_TX_PACK1")
    out = {}
    for what, env in (("rec16", {"PLI_TX_HOT": "0"}), ("hot", {}), ("hot_round1_only", {"PLI_TX_HOT": "1"}), ("hot_ranks", {"PLI_TX_KEYS": "0"}),
                      ("hot_lazy_ids", {"PLI_TX_PACK1": "2"}), ("hot_lazy_ids_wide_margin", {"PLI_TX_PACK1": "2", "PLI_ALIGN_MARGIN_DEG": "4", "PLI_TX_HOT_BAND2": "10"}),
                      ("hot_round1_only_ranks_no_tail", {"PLI_TX_HOT": "1", "PLI_TX_KEYS": "0", "PLI_TX_TAIL": "0"}),
                      ("hot_wide_margin", {"PLI_ALIGN_MARGIN_DEG": "4"}),
                      ("hot_wide_margin_ranks", {"PLI_ALIGN_MARGIN_DEG": "8", "PLI_TX_KEYS": "0", "PLI_TX_TAIL": "0"}),
                      ("hot_rect_exact", {"PLI_RECT_APPROX_BAND": "10"}),
                      # (the second band off: every candidate inside the margin takes the exact sums)
                      ("hot_every_event_exact", {"PLI_TX_HOT_BAND2": "10", "PLI_ALIGN_MARGIN_DEG": "4"}),
                      ("hot_ts32_arena3", {"PLI_TX_TS": "32", "PLI_RX_ARENA": "3", "PLI_ALIGN_MARGIN_DEG": "1"})):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        fe = g.Frontend(cfg, dev=True)
        fe.batch_run_host(imgs)
        out[what] = fe.batch_run_host(imgs)
        if "arena3" not in what:
            assert fe.lsd_round_stats()[2] == 0, (what, fe.lsd_round_stats())      # nobody on the slow path
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    assert len(out["hot"][0]["klL"]) > 100
    for f, (L, R) in enumerate(pairs):
        assert_frame_equal(g, out["hot"][f], g.po.Frame(ocfg(g, cfg)), L, R, "hot records frame %d" % f)
        for what in out:
            for k in ("klL", "klR", "ldescL", "ldescR", "disp", "le"):
                assert out[what][f][k].tobytes() == out["hot"][f][k].tobytes(), (what, f, k)


def test_frame_extract_equals_the_per_call_entry_points(cfg2):
    """pli_frame_extract (what the adapters fuse the four extractor threads of a Frame into) against the four per-call entry points
    and the two stereo matchers: the same record, and the per-call state it leaves (pyramid levels, stereo matchers without a rerun)."""
    g, cfg, fe, fr, L, R = cfg2
    rec = fe.frame_extract(L, R)
    ur, dp = fe.compute_stereo_matches()                  # handed out from the frame's record
    disp, le = fe.compute_stereo_matches_lines()
    N, NL = len(rec["kpL"]), len(rec["klL"])
    assert rec["uright"].tobytes() == ur[:N].tobytes() and rec["depth"].tobytes() == dp[:N].tobytes()
    assert rec["disp"].tobytes() == disp[:NL].tobytes() and rec["le"].tobytes() == le[:NL].tobytes()
    p0 = fe.pyramid_level(0, 0)
    assert np.array_equal(p0, L) and fe.pyramid_level(1, 7).shape == fr.pyramid(1, 7).shape
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, R, "pli_frame_extract")
    for eye, img, k in ((0, L, "L"), (1, R, "R")):       # the per-call path afterwards: same tables, and the stereo matchers run again
        n, kp, desc = fe.orb_extract(eye, img)
        assert kp.tobytes() == rec["kp" + k].tobytes() and np.array_equal(desc, rec["desc" + k])
        m, kl, ld = fe.line_extract(eye, img)
        assert kl.tobytes() == rec["kl" + k].tobytes() and np.array_equal(ld, rec["ldesc" + k])
    ur2, dp2 = fe.compute_stereo_matches()
    assert ur2[:N].tobytes() == ur[:N].tobytes() and dp2[:N].tobytes() == dp[:N].tobytes()


def test_roctx_ranges_are_pushed_when_asked_for(gpu):
    """with PLI_ROCTX=1 the library brackets its entry points, stages and kernel launches with roctx ranges (the
    marker library is found at run time); without the switch nothing is pushed.  Run in child processes: the switch is read when
    the library is loaded."""
    import subprocess, sys
    code = ("import numpy as np\n"
            "from pli_slam_amd import capi, synth\n"
            "from pli_slam_amd.frontend import Frontend\n"
            "L, R = synth.make_stereo_pair(3, 376, 240)\n"
            "fe = Frontend(capi.default_config(376, 240, orb_nfeatures=500, lsd_nfeatures=60, max_frames=1))\n"
            "rec = fe.batch_run_host(np.stack([L, R])[None])[0]\n"
            "print('RANGES', capi.lib().pli_trace_ranges(), len(rec['kpL']))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for flag in ("0", "1"):
        env = dict(os.environ, PLI_ROCTX=flag, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RANGES")][-1].split()
        out[flag] = (int(line[1]), int(line[2]))
    assert out["0"][0] == 0, "ranges pushed without the switch: %s" % (out,)
    assert out["1"][0] > 40, "PLI_ROCTX=1: only %d ranges for a whole frame (entry point + stages + ~60 kernel launches)" % out["1"][0]
    assert out["0"][1] == out["1"][1] > 100


def test_an_aborted_tail_kernel_switches_the_process_to_planned_rounds(gpu, tmp_path):
    """several PROCESSES on one device can keep the persistent relaxation kernel (k_tx_tail) from becoming resident; its grid
    barrier then times out, the kernel gives up, and the unsettled images take the exact fallback.  A process that sees this once must
    not stall in every later call: it latches the planned-rounds schedule for that device.  Run in a process of its own (the latch is
    per process), with the test switch that raises the abort word before the launch."""
    import subprocess, sys, textwrap
    script = tmp_path / "abort.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np
        from oracle import pyoracle as po
        from pli_slam_amd import capi, synth
        from pli_slam_amd.frontend import Frontend
        W, H = 752, 480
        cfg = capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1)
        L, R = synth.make_stereo_pair(321, W, H)
        imgs = np.stack([L, R])[None]
        fe = Frontend(cfg, dev=True)           # (PLI_TX_TAIL_FORCE_ABORT is a test switch of the development build)
        fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
        want = [fr.line_extract(e, im) for e, im in ((0, L), (1, R))]
        def same(rec):
            return all(rec["kl" + k].tobytes() == want[e][1].tobytes() and np.array_equal(rec["ldesc" + k], want[e][2]) for e, k in ((0, "L"), (1, "R")))
        r0 = fe.batch_run_host(imgs)[0]; s0 = fe.lsd_round_stats()
        os.environ["PLI_TX_TAIL_FORCE_ABORT"] = "1"
        r1 = fe.batch_run_host(imgs)[0]; s1 = fe.lsd_round_stats()
        del os.environ["PLI_TX_TAIL_FORCE_ABORT"]
        r2 = fe.batch_run_host(imgs)[0]; s2 = fe.lsd_round_stats()
        r3 = fe.batch_run_host(imgs)[0]; s3 = fe.lsd_round_stats()
        import json
        print("RESULT" + json.dumps([bool(same(r0)), bool(same(r1)), bool(same(r2)), bool(same(r3)), [int(v) for v in s0], [int(v) for v in s1],
                                     [int(v) for v in s2], [int(v) for v in s3]]))
    """ % ROOT))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1]
    import json
    ok0, ok1, ok2, ok3, s0, s1, s2, s3 = json.loads(line[len("RESULT"):])
    assert ok0 and ok1 and ok2 and ok3, line                  # exact on every path
    assert s0[0] == -1 and s0[2] == 0, line                   # call 1: the tail kernel, nobody on the slow path
    assert s1[2] >= 2, line                                   # the aborted tail: both images redone by the sequential grower
    assert s2[0] >= 1 and s3[0] >= 1 and s3[2] == s2[2] == s1[2], line   # latched: planned rounds from then on, no further slow-path images
    assert "planned-rounds schedule" in out.stderr


def test_where_the_orb_chain_forks_does_not_change_a_byte(gpu, monkeypatch):
    """pli_batch_run puts the ORB chain on a side stream that forks from the line chain before it starts, behind round 1's growth
    (batches of up to 64 images), or behind round 2's (larger batches; DESIGN.md 5: worth +-3 % on the synthetic stream and +-8 % on
    photographs).  Wherever it forks (dev switches PLI_SIDE_DEFER_MAX, PLI_SIDE_FORK_ROUND) the tables are the same bytes."""
    g = gpu
    W, H = 376, 240
    F = 40                                                    # 80 images: above the small-batch line
    cfg = g.capi.default_config(W, H, orb_nfeatures=400, lsd_nfeatures=40, max_frames=F)
    pairs = [g.synth.make_stereo_pair(700 + i, W, H) for i in range(8)]
    imgs = np.stack([np.stack(pairs[i % 8]) for i in range(F)])
    out = {}
    for what, env in (("default", {}), ("before_the_line_chain", {"PLI_SIDE_DEFER_MAX": "0"}), ("round1", {"PLI_SIDE_DEFER_MAX": "4096", "PLI_SIDE_FORK_ROUND": "1"}),
                      ("round2_pass", {"PLI_SIDE_DEFER_MAX": "4096", "PLI_SIDE_FORK_ROUND": "-2"}), ("round3", {"PLI_SIDE_DEFER_MAX": "4096", "PLI_SIDE_FORK_ROUND": "3"}),
                      ("never_in_the_rounds", {"PLI_SIDE_DEFER_MAX": "4096", "PLI_SIDE_FORK_ROUND": "90"}), ("no_side_stream", {"PLI_SIDE_MAX": "0"})):
        for k in ("PLI_SIDE_DEFER_MAX", "PLI_SIDE_FORK_ROUND"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            if k != "PLI_SIDE_MAX":
                monkeypatch.setenv(k, v)
        if what == "no_side_stream":
            continue                                          # (PLI_SIDE_MAX is read once per process: covered by tools/ab_env.sh runs, not here)
        fe = g.Frontend(cfg, dev=True)
        fe.batch_run_host(imgs)
        out[what] = fe.batch_run_host(imgs)
    for k in ("PLI_SIDE_DEFER_MAX", "PLI_SIDE_FORK_ROUND"):
        monkeypatch.delenv(k, raising=False)
    fr = g.po.Frame(ocfg(g, cfg))
    assert_frame_equal(g, out["default"][3], fr, pairs[3][0], pairs[3][1], "default fork, frame 3")
    for what in out:
        for f in range(F):
            for k in out["default"][f]:
                a, b = out[what][f][k], out["default"][f][k]
                assert (a.tobytes() == b.tobytes()) if hasattr(a, "tobytes") else a == b, (what, f, k)


@pytest.mark.parametrize("nbins", [64, 129, 200, 1024])
def test_lsd_bin_counts_through_the_packed_round_one(gpu, nbins):
    """`lsd_n_bins` (Lineextractor's quantisation of the gradient norm into seed bins) through round 1's packed owner word: the lazy form's
    fixed-point bin width needs more than 128 bins (tests/test_lazy_ids_cpu.py found 17 bins overflowing it), below that the ids are
    written into the records by the sort.  Every count gives the oracle's lines."""
    g = gpu
    W, H = 752, 480
    cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, lsd_n_bins=nbins)
    L, R = g.synth.make_stereo_pair(420 + nbins, W, H)
    fe = g.Frontend(cfg)
    rec = fe.batch_run_host(np.stack([L, R])[None])[0]
    assert fe.lsd_round_stats()[2] == 0
    assert len(rec["klL"]) > 50
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, R, "lsd_n_bins = %d" % nbins)
