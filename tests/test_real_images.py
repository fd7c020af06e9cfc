"""Real imagery: photographs instead of pli_slam_amd/synth.py scenes.

tests/golden/real/photos.npz holds 8-bit grayscale copies of photographs the build image ships as data (scikit-image's sample
set; tools/make_real_fixtures.py made the file): natural textures (brick, gravel, grass), scenes (camera, astronaut, coffee,
rocket, chelsea, coins, moon), documents (text, page) and a rectified Middlebury stereo pair (motorcycle, 741x500) WITH
ground-truth disparity — the first evidence for Frame::ComputeStereoMatches (Frame.cc:976-1154) in this repository that does not
come from the oracle: mvuRight must agree with a laser-scanned disparity map.

CPU (not gpu): the fixture, the oracle on the stereo pair against the ground truth.
GPU: HIP == oracle byte for byte on every photograph (both eyes, every LSD schedule), the stereo pair through the C ABI against
the ground truth, round counts / fallbacks of the relaxation on natural images.
"""
import numpy as np
import pytest

from pli_slam_amd import realdata

GT_TOL_PX = 1.5            # |disparity - ground truth| at the keypoint; the reference refines to sub-pixel by a parabola (Frame.cc:1110-1125)
GT_SHARE_FLOOR = 0.85      # measured: 0.917 of 397 matched keypoints with known ground truth (median error 0.35 px)


def stereo_vs_ground_truth(kp, ur, gt):
    ok = ur >= 0
    x, y = kp["x"][ok], kp["y"][ok]
    g = gt[np.rint(y).astype(int), np.rint(x).astype(int)]
    known = g > 0
    err = np.abs((x - ur[ok])[known] - g[known])
    return int(ok.sum()), int(known.sum()), float((err <= GT_TOL_PX).mean()), float(np.median(err))


def motorcycle_cfg(capi, **over):
    L, R, gt = realdata.motorcycle()
    H, W = L.shape
    # (any rig whose maxD = fx covers the scene's disparities, 7..60 px)
    return capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=1, bf=100.0, fx=500.0, **over)


def test_fixture_is_what_the_script_says():
    ph = realdata.photos()
    assert len(ph) == 14 and ph["motorcycle_left"].shape == (500, 741) and ph["camera"].shape == (512, 512)
    L, R, gt = realdata.motorcycle()
    assert (gt > 0).mean() > 0.9 and 7.0 < gt[gt > 0].min() and gt.max() < 60.0
    fr = realdata.frames_752x480(3, seed=1)
    assert all(l.shape == (480, 752) and r.shape == (480, 752) and l.dtype == np.uint8 for l, r in fr)
    assert realdata.frames_752x480(3, seed=1)[2][0].tobytes() == fr[2][0].tobytes()


def test_oracle_stereo_points_against_middlebury_ground_truth(oracle):
    """The CPU restatement of ComputeStereoMatches on a real rectified pair: 9 of 10 matched keypoints land within 1.5 px of
    the scanner's disparity.  (Not a pin of OpenCV's bits — a check that the restated algorithm measures the scene.)"""
    from pli_slam_amd import capi
    po = oracle
    L, R, gt = realdata.motorcycle()
    fr = po.Frame(po.Config.from_buffer_copy(bytes(motorcycle_cfg(capi))))
    n, kp, _ = fr.orb_extract(0, L)
    fr.orb_extract(1, R)
    ur, dp, _, _ = fr.stereo_points()
    matched, known, share, med = stereo_vs_ground_truth(kp, ur, gt)
    print("oracle: %d keypoints, %d stereo matches, %d with ground truth, %.3f within %.1f px, median error %.2f px" % (n, matched, known, share, GT_TOL_PX, med))
    assert n > 1000 and matched > 300 and known > 250
    assert share >= GT_SHARE_FLOOR and med < 0.6
    # depth = bf / disparity (Frame.cc:1131)
    ok = ur >= 0
    assert np.allclose(dp[ok], np.float32(100.0) / (kp["x"][ok] - ur[ok]), rtol=1e-6)
    # the stereo LINE matcher (Frame.cc:1156-1307) against the same ground truth, at the matched lines' end points
    _, kl, _ = fr.line_extract(0, L)
    fr.line_extract(1, R)
    disp, _, _ = fr.stereo_lines()
    errs = []
    for i in np.flatnonzero(disp[:, 0] >= 0):
        for (x, y), d in (((kl["startPointX"][i], kl["startPointY"][i]), disp[i, 0]), ((kl["endPointX"][i], kl["endPointY"][i]), disp[i, 1])):
            gv = gt[min(int(y), gt.shape[0] - 1), min(int(x), gt.shape[1] - 1)]
            if gv > 0:
                errs.append(abs(d - gv))
    print("oracle: %d stereo lines, %d end points with ground truth, median |error| %.2f px" % (int((disp[:, 0] >= 0).sum()), len(errs), np.median(errs)))
    assert len(errs) > 40 and np.median(errs) < 3.0


# ---- GPU ---------------------------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from pli_slam_amd import capi
    from pli_slam_amd.frontend import Frontend
    from oracle import pyoracle as po

    class G:
        pass
    g = G()
    g.capi, g.Frontend, g.po = capi, Frontend, po
    return g


PHOTO_NAMES = ["astronaut", "brick", "camera", "chelsea", "coffee", "coins", "grass", "gravel", "moon", "page", "rocket", "text"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", PHOTO_NAMES + ["motorcycle"])
def test_photograph_whole_frame_equals_the_oracle(gpu, name):
    """Both eyes (the right eye of a single photograph: the left one displaced by 9 px + noise; the motorcycle pair: its real
    right eye), ORB + LSD/LBD + both stereo matchers, default schedule, every table byte for byte."""
    from test_gpu_parity import assert_frame_equal, ocfg
    g = gpu
    if name == "motorcycle":
        L, R, _ = realdata.motorcycle()
    else:
        L = realdata.photos()[name]
        R = realdata.shifted_right(L, 9, seed=len(name))
    H, W = L.shape
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=1, bf=100.0, fx=500.0)
    fe = g.Frontend(cfg)
    rec = fe.batch_run_host(np.stack([L, R])[None])[0]
    assert_frame_equal(g, rec, g.po.Frame(ocfg(g, cfg)), L, R, name)
    st = fe.lsd_round_stats()
    print("%s %dx%d: %d/%d keypoints, %d/%d lines, %d stereo points, %d stereo lines, LSD rounds %s" % (
        name, W, H, len(rec["kpL"]), len(rec["kpR"]), len(rec["klL"]), len(rec["klR"]), rec["counts"][4], rec["counts"][5], st))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_photographs_every_lsd_schedule(gpu, mode, monkeypatch):
    """Every segment above the length cut (lsd_nfeatures = 0) of every photograph under the three schedules of the region
    growing (rank-ordered relaxation, sequential waves, tile relaxation; the tile relaxation also in rank mode): natural
    gradients — long soft edges, texture without edges, print — are what cv::LineSegmentDetector's growth order is sensitive to."""
    from test_gpu_parity import ocfg
    g = gpu
    ph = realdata.photos()
    for keys in (("1", "0") if mode == 3 else ("1",)):
        monkeypatch.setenv("PLI_TX_KEYS", keys)
        for name in PHOTO_NAMES + ["motorcycle_left", "motorcycle_right"]:
            img = ph[name]
            H, W = img.shape
            cfg = g.capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, lsd_mode=mode)
            fe = g.Frontend(cfg, dev=True)
            n, kl, ld = fe.line_extract(0, img)
            m, okl, old = g.po.Frame(ocfg(g, cfg)).line_extract(0, img)
            assert n == m, "%s mode %d: %d lines vs oracle %d" % (name, mode, n, m)
            assert kl.tobytes() == okl.tobytes() and np.array_equal(ld, old), "%s mode %d keys %s" % (name, mode, keys)
    monkeypatch.delenv("PLI_TX_KEYS")


@pytest.mark.gpu
def test_stereo_points_on_the_gpu_against_middlebury_ground_truth(gpu):
    g = gpu
    L, R, gt = realdata.motorcycle()
    fe = g.Frontend(motorcycle_cfg(g.capi))
    rec = fe.frame_extract(L, R)
    matched, known, share, med = stereo_vs_ground_truth(rec["kpL"], rec["uright"], gt)
    print("GPU: %d stereo matches, %d with ground truth, %.3f within %.1f px, median error %.2f px" % (matched, known, share, GT_TOL_PX, med))
    assert matched > 300 and share >= GT_SHARE_FLOOR and med < 0.6
    # stereo lines: endpoints' disparities (mvDisparity_l, Frame.cc:1231-1247) against the ground truth at the endpoints, where known
    kl, disp = rec["klL"], rec["disp"]
    errs = []
    for i in np.flatnonzero(disp[:, 0] >= 0):
        for (x, y), d in (((kl["startPointX"][i], kl["startPointY"][i]), disp[i, 0]), ((kl["endPointX"][i], kl["endPointY"][i]), disp[i, 1])):
            gv = gt[min(int(y), gt.shape[0] - 1), min(int(x), gt.shape[1] - 1)]
            if gv > 0:
                errs.append(abs(d - gv))
    errs = np.array(errs)
    print("GPU: %d stereo lines, %d endpoints with ground truth, median |error| %.2f px, %.2f within 4 px" % (
        int((disp[:, 0] >= 0).sum()), len(errs), float(np.median(errs)), float((errs <= 4).mean())))
    assert len(errs) > 40 and np.median(errs) < 3.0       # (line endpoints sit ON depth edges: looser than the keypoints)


@pytest.mark.gpu
def test_real_frames_batch_rounds_and_fallbacks(gpu):
    """32 frames of 752x480 cut from the photographs, through the batch entry point: records equal the oracle on a sample, the
    relaxation settles in a bounded number of rounds and no image takes the device-side fallback."""
    from test_gpu_parity import assert_frame_equal, ocfg
    g = gpu
    frames = realdata.frames_752x480(32, seed=3)
    cfg = g.capi.default_config(752, 480, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=32)
    fe = g.Frontend(cfg)
    batch = np.stack([np.stack(f) for f in frames])
    recs = fe.batch_run_host(batch)
    recs2 = fe.batch_run_host(batch)           # (the second call plans its rounds from the first)
    st = fe.lsd_round_stats()
    print("real frames: LSD round stats (planned, rounds, images redone by the device fallback, last) =", st)
    assert st[2] == 0, "images took the device-side fallback on natural images: %s" % (st,)
    assert 0 < st[3] <= 40
    for i in (0, 5, 13, 14, 27):               # (13: a motorcycle frame with its real right eye)
        assert_frame_equal(g, recs[i], g.po.Frame(ocfg(g, cfg)), frames[i][0], frames[i][1], "real frame %d" % i)
        assert recs2[i]["klL"].tobytes() == recs[i]["klL"].tobytes() and recs2[i]["uright"].tobytes() == recs[i]["uright"].tobytes()


@pytest.mark.gpu
def test_frame_to_frame_tracking_on_the_real_stereo_pair(gpu):
    """pli_batch_track (SearchByProjection(CurrentFrame, LastFrame) + match() of the line descriptors, Tracking.cc:3046-3058) on four
    instants of the Middlebury pair — a 640x440 window drifting by (3, 1) px per instant over the real left / right images, so depths
    come from real stereo matches — equals the oracle's restatement; and the tracks are real: most stereo points of an instant are
    found again in the next one a few pixels away."""
    import torch
    g = gpu
    L0, R0, _ = realdata.motorcycle()
    W, H, F = 640, 440, 4
    frames = [(np.ascontiguousarray(L0[10 + t:10 + t + H, 20 + 3 * t:20 + 3 * t + W]),
               np.ascontiguousarray(R0[10 + t:10 + t + H, 20 + 3 * t:20 + 3 * t + W])) for t in range(F)]
    cfg = g.capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=F, bf=100.0, fx=500.0)
    fe = g.Frontend(cfg)
    imgs = np.stack([np.stack(f) for f in frames])
    left, right = np.ascontiguousarray(imgs[:, 0]), np.ascontiguousarray(imgs[:, 1])
    table = np.zeros(fe.table_bytes(F), np.uint8)
    g.capi.check(fe.L.pli_batch_run_host(fe.h, F, g.capi.ptr(left), g.capi.ptr(right), W, W * H, g.capi.RUN_ALL, g.capi.ptr(table)))
    recs = [fe.parse_record(table, f) for f in range(F)]
    poses = np.stack([np.eye(4, dtype=np.float32)[:3] for _ in range(F)])          # the motion model's prediction: no motion
    tp = fe.track_params(th=15.0, mono=False, check_orientation=True, nnr_lines=0.9)
    tl = fe.track_layout()
    d_table = torch.from_numpy(table).cuda()
    d_poses = torch.from_numpy(poses.reshape(-1)).cuda()
    d_track = torch.zeros(F * tl.record_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    fe.batch_track_device(F, d_table.data_ptr(), d_poses.data_ptr(), tp, d_track.data_ptr())
    fe.sync()
    track = d_track.cpu().numpy()
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)
    for f in range(1, F):
        last, cur = recs[f - 1], recs[f]
        tr = fe.parse_track(track, f)
        q = g.po.track_queries(last["kpL"], last["depth"], poses[f - 1], poses[f], tp.fx, tp.fy, tp.cx, tp.cy, tp.bf, tp.th, False, sf)
        on, obest = g.po.search_by_projection(q, last["descL"], cur["kpL"], cur["descL"], cur["uright"], (0.0, float(W), 0.0, float(H)), True)
        assert tr["counts"][1] == on and np.array_equal(tr["best"], obest), "instant %d point tracks" % f
        ln, lm = g.po.match_lines(last["ldescL"], cur["ldescL"], 0.9, True)
        assert tr["counts"][3] == ln and np.array_equal(tr["lines"], lm), "instant %d line tracks" % f
        # the tracks are physical: a tracked keypoint sits where the window's drift puts it, (-3, -1) px from its last position
        m = np.flatnonzero(obest >= 0)
        dx = cur["kpL"]["x"][obest[m]] - last["kpL"]["x"][m]
        dy = cur["kpL"]["y"][obest[m]] - last["kpL"]["y"][m]
        good = (np.abs(dx + 3) <= 2.5) & (np.abs(dy + 1) <= 2.5)
        nstereo = int((last["depth"] > 0).sum())
        print("instant %d: %d stereo points, %d tracked, %.2f of them at the drift; %d of %d lines matched" % (f, nstereo, on, good.mean(), ln, len(last["ldescL"])))
        assert on > 0.5 * nstereo and good.mean() > 0.8 and ln > 0.3 * len(last["ldescL"])
