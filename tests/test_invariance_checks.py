"""Symmetries the restated ORB front-end must have, checked on real photographs (CPU; the HIP path equals the oracle bit for bit,
tests/test_real_images.py): evidence for SURVEY §8 rows a3/a4 (per-cell FAST-9 + NMS), a6 (IC_Angle) and a8 (steered rBRIEF) that
does not come from reading OpenCV or the reference — a restatement with a wrong ring order, a wrong moment sign, a lopsided rounding
or a transposed pattern would not survive a 90-degree turn of the image.

Level 0 of a SQUARE image turned counter-clockwise by 90 degrees (an exact permutation of the pixels; the 16-px border, the 7x7
fixed-point Gaussian without intermediate rounding and the circular patch are symmetric under it):
  * the FAST corner at (x, y) reappears at (y, W-1-x) with the SAME score (cornerScore is a property of the ring); the two candidate
    sets differ only where the 30-px cell grid, which is anchored at the top-left, cuts differently (NMS and the per-cell threshold
    fallback look at a cell's own interior only, ORBextractor.cc:790-841);
  * a keypoint selected in both images has its orientation turned by exactly -90 degrees (y points down) and the SAME 256 bits:
    the pattern is steered by the angle, so the steered test points land on the turned pixels.
(The levels above 0 are not exactly symmetric: cv::resize rounds after the horizontal pass, ORBextractor.cc:1165.)"""
import numpy as np
import pytest

from pli_slam_amd import realdata

SQUARE = ["camera", "brick", "astronaut", "moon", "grass"]


@pytest.mark.parametrize("name", SQUARE)
def test_orb_level0_is_equivariant_under_a_quarter_turn(oracle, name):
    from pli_slam_amd import capi
    po = oracle
    img = realdata.photos()[name]
    assert img.shape[0] == img.shape[1]
    W = img.shape[1]
    rot = np.ascontiguousarray(np.rot90(img))                  # counter-clockwise: rot[i, j] = img[j, W-1-i]
    cfg = capi.default_config(W, W, orb_nfeatures=1200, lsd_nfeatures=0, max_frames=1)
    f0, f1 = (po.Frame(po.Config.from_buffer_copy(bytes(cfg))) for _ in range(2))
    n0, kp0, d0 = f0.orb_extract(0, img)
    n1, kp1, d1 = f1.orb_extract(0, rot)
    # FAST candidates of level 0 (coordinates relative to the 16-px border): (x, y) -> (y, W-33-x)
    c0, c1 = f0.level_points(0, 0, False), f1.level_points(0, 0, False)
    s0 = {(int(y), int(W - 33 - x)): int(s) for x, y, s in c0}
    s1 = {(int(x), int(y)): int(s) for x, y, s in c1}
    common = set(s0) & set(s1)
    assert len(common) >= 0.8 * min(len(s0), len(s1)), "%s: only %d of %d / %d corners reappear after the turn" % (name, len(common), len(s0), len(s1))
    assert all(s0[k] == s1[k] for k in common), "a corner's score changed under the turn"
    # selected keypoints of level 0 present in both: angle - 90 degrees, identical descriptor
    k0 = {(float(k["y"]), float(W - 1 - k["x"])): i for i, k in enumerate(kp0) if k["octave"] == 0}
    k1 = {(float(k["x"]), float(k["y"])): i for i, k in enumerate(kp1) if k["octave"] == 0}
    both = sorted(set(k0) & set(k1))
    assert len(both) >= 0.7 * min(len(k0), len(k1)) and len(both) > 50, (name, len(k0), len(k1), len(both))
    da = np.array([((kp1["angle"][k1[k]] - kp0["angle"][k0[k]] + 540.0) % 360.0) - 180.0 for k in both])
    hd = np.array([int(np.unpackbits(d0[k0[k]] ^ d1[k1[k]]).sum()) for k in both])
    print("%s: %d / %d corners reappear with equal scores; %d keypoints in both: angle turned by %.5f +- %.5f deg, descriptor distance max %d" % (
        name, len(common), min(len(s0), len(s1)), len(both), da.mean(), da.std(), hd.max()))
    assert np.abs(da + 90.0).max() < 0.01, "orientation is not turned by -90 degrees: %s" % da[np.abs(da + 90.0) >= 0.01][:5]
    assert (hd <= 2).mean() >= 0.99 and hd.max() <= 16, "steered descriptors differ after the turn: %s" % np.sort(hd)[-5:]


@pytest.mark.parametrize("name", ["camera", "brick", "coffee", "text", "motorcycle_left"])
def test_lines_and_their_lbd_descriptors_survive_a_half_turn(oracle, name):
    """Rows a10-a14 (LSD segments, KeyLine fields, LBD): the photograph turned by 180 degrees (an exact permutation of the pixels).
    The detector is not exactly symmetric (seeds of a gradient bin are visited in raster order), so the two line sets differ in
    places — but a line found in both (end points within 1.5 px after turning the coordinates back) keeps its start / end ORDER (the
    order encodes which side is dark, cv::LineSegmentDetector region2rect) and its 256 LBD bits to within a few: the band descriptor
    is built in the line's own frame (dL, dO; binary_descriptor_custom.cpp:1130-1160), so a wrong sign or a swapped band would show
    up as ~100 differing bits — the distance between unrelated lines."""
    from pli_slam_amd import capi
    po = oracle
    img = realdata.photos()[name]
    H, W = img.shape
    rot = np.ascontiguousarray(np.rot90(img, 2))
    cfg = capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1)
    f0, f1 = (po.Frame(po.Config.from_buffer_copy(bytes(cfg))) for _ in range(2))
    m0, k0, d0 = f0.line_extract(0, img)
    m1, k1, d1 = f1.line_extract(0, rot)
    P0 = np.stack([W - k0["startPointX"], H - k0["startPointY"], W - k0["endPointX"], H - k0["endPointY"]], 1)
    P1 = np.stack([k1["startPointX"], k1["startPointY"], k1["endPointX"], k1["endPointY"]], 1)
    same_order, swapped, hd, unrelated = 0, 0, [], []
    for i in range(m0):
        ds = np.maximum(np.hypot(P1[:, 0] - P0[i, 0], P1[:, 1] - P0[i, 1]), np.hypot(P1[:, 2] - P0[i, 2], P1[:, 3] - P0[i, 3]))
        dw = np.maximum(np.hypot(P1[:, 0] - P0[i, 2], P1[:, 1] - P0[i, 3]), np.hypot(P1[:, 2] - P0[i, 0], P1[:, 3] - P0[i, 1]))
        j = int(np.argmin(np.minimum(ds, dw)))
        if min(ds[j], dw[j]) <= 1.5:
            if dw[j] < ds[j]:
                swapped += 1
                continue
            same_order += 1
            hd.append(int(np.unpackbits(d0[i] ^ d1[j]).sum()))
            unrelated.append(int(np.unpackbits(d0[i] ^ d1[(j + 7) % m1]).sum()))
    hd = np.array(hd)
    print("%s: %d / %d lines, %d found in both with the same end-point order (%d reversed), LBD distance median %d, 90th percentile %d; unrelated lines %d" % (
        name, m0, m1, same_order, swapped, np.median(hd), np.percentile(hd, 90), np.median(unrelated)))
    assert same_order >= 0.3 * min(m0, m1) and swapped <= 0.05 * same_order + 2
    assert np.median(hd) <= 16 and np.percentile(hd, 90) <= 40 and np.median(unrelated) >= 60
