"""Seeded synthetic EuRoC-shaped stereo stream.

u8 grayscale frames: band-limited noise background (mean ~110, sigma ~40) so FAST
finds many more candidates than the per-level quota, plus random dark/bright
convex quadrilaterals and straight high-contrast strips so LSD yields >= 100
segments; the right image is the left canvas re-sampled with a per-row
disparity in [2, 60] px plus independent sensor noise; consecutive frames are
the same scene shifted by (3, 1) px and rotated by 0.5 degrees.

There is no dataset access on the build or GPU machines, so this stream stands
in for EuRoC MH_01 (752x480) in tests and in bench.py.
"""
import numpy as np
from scipy import ndimage


def _fill_poly(canvas, pts, value):
    """Fill a convex polygon (vertices in order) with `value`."""
    h, w = canvas.shape
    x0 = max(int(np.floor(pts[:, 0].min())), 0)
    x1 = min(int(np.ceil(pts[:, 0].max())) + 1, w)
    y0 = max(int(np.floor(pts[:, 1].min())), 0)
    y1 = min(int(np.ceil(pts[:, 1].max())) + 1, h)
    if x1 <= x0 or y1 <= y0:
        return
    ys, xs = np.mgrid[y0:y1, x0:x1]
    inside_pos = np.ones(xs.shape, bool)
    inside_neg = np.ones(xs.shape, bool)
    n = len(pts)
    for i in range(n):
        ax, ay = pts[i]
        bx, by = pts[(i + 1) % n]
        cross = (bx - ax) * (ys - ay) - (by - ay) * (xs - ax)
        inside_pos &= cross >= 0
        inside_neg &= cross <= 0
    m = inside_pos | inside_neg
    canvas[y0:y1, x0:x1][m] = value


def make_scene(seed, w=752, h=480, margin=80, n_quads=14, n_strips=150):
    """Float canvas of (h+2*margin) x (w+2*margin) the frames are cut from."""
    rng = np.random.default_rng(seed)
    H, W = h + 2 * margin, w + 2 * margin
    bg = ndimage.gaussian_filter(rng.random((H, W)), 3.0)
    bg = (bg - bg.mean()) / (bg.std() + 1e-12)
    fine = ndimage.gaussian_filter(rng.random((H, W)), 1.2)
    fine = (fine - fine.mean()) / (fine.std() + 1e-12)
    canvas = 110.0 + 32.0 * bg + 24.0 * fine
    for _ in range(n_quads):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        r = rng.uniform(25, 0.18 * max(w, h))
        ang = np.sort(rng.uniform(0, 2 * np.pi, 4))
        pts = np.stack([cx + r * np.cos(ang) * rng.uniform(0.6, 1.0, 4),
                        cy + r * np.sin(ang) * rng.uniform(0.6, 1.0, 4)], 1)
        _fill_poly(canvas, pts, rng.choice([rng.uniform(15, 60), rng.uniform(180, 240)]))
    for _ in range(n_strips):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        length = rng.uniform(20, 0.4 * max(w, h))
        th = rng.uniform(0, np.pi)
        half_w = rng.uniform(1.0, 3.5)
        d = np.array([np.cos(th), np.sin(th)])
        nrm = np.array([-d[1], d[0]])
        c = np.array([cx, cy])
        pts = np.stack([c - d * length / 2 - nrm * half_w, c + d * length / 2 - nrm * half_w,
                        c + d * length / 2 + nrm * half_w, c - d * length / 2 + nrm * half_w])
        _fill_poly(canvas, pts, rng.choice([rng.uniform(5, 50), rng.uniform(190, 250)]))
    return ndimage.gaussian_filter(canvas, 0.7)


def _row_disparity(rng, h):
    """Piecewise-linear disparity per image row in [2, 60] px."""
    knots_y = np.concatenate([[0], np.sort(rng.uniform(0, h, 3)), [h]])
    knots_d = rng.uniform(2.0, 60.0, 5)
    return np.interp(np.arange(h), knots_y, knots_d)


def _sample(canvas, xs, ys):
    return ndimage.map_coordinates(canvas, [ys, xs], order=1, mode="reflect")


def make_stereo_pair(seed, w=752, h=480, t=0):
    """Left/right u8 images of frame `t` of the scene with this seed."""
    margin = 80
    canvas = make_scene(seed, w, h, margin)
    rng = np.random.default_rng([seed, 7919])
    disp = _row_disparity(rng, h)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    # camera motion: frame t = scene shifted by t*(3,1) px and rotated by t*0.5 deg about the centre
    a = np.deg2rad(0.5 * t)
    cx, cy = w / 2.0, h / 2.0
    xr = np.cos(a) * (xs - cx) - np.sin(a) * (ys - cy) + cx + 3.0 * t + margin
    yr = np.sin(a) * (xs - cx) + np.cos(a) * (ys - cy) + cy + 1.0 * t + margin
    left = _sample(canvas, xr, yr)
    # right camera: a point seen at uL appears at uR = uL - d(v)
    right = _sample(canvas, xr + disp[:, None], yr)
    nrng = np.random.default_rng([seed, t, 104729])
    left = left + nrng.normal(0, 2.0, left.shape)
    right = right + nrng.normal(0, 2.0, right.shape)
    to8 = lambda a_: np.clip(np.rint(a_), 0, 255).astype(np.uint8)
    return to8(left), to8(right)


def make_batch(n_frames, w=752, h=480, seed0=0):
    """(n_frames, 2, h, w) u8: seeds seed0..seed0+n-1, eye 0 = left."""
    out = np.empty((n_frames, 2, h, w), np.uint8)
    for i in range(n_frames):
        out[i, 0], out[i, 1] = make_stereo_pair(seed0 + i, w, h)
    return out


# EuRoC calibration of the reference's Examples/Stereo/Config/EuRoC.yaml (data): K, D (k1 k2 p1 p2 k3), R, P[:3,:3]
EUROC_CALIB = {
    0: dict(K=[458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0],
            D=[-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0],
            R=[0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127,
               0.9999741760894847, 0.007055629199258132, -0.008089410156878961, -0.007044357138835809,
               0.9999424675829176],
            P=[435.2046959714599, 0, 367.4517211914062, 0, 435.2046959714599, 252.2008514404297, 0, 0, 1]),
    1: dict(K=[457.587, 0.0, 379.999, 0.0, 456.134, 255.238, 0.0, 0.0, 1],
            D=[-0.28368365, 0.07451284, -0.00010473, -3.555907e-05, 0.0],
            R=[0.9999633526194376, -0.003625811871560086, 0.007755443660172947, 0.003680398547259526,
               0.9999684752771629, -0.007035845251224894, -0.007729688520722713, 0.007064130529506649,
               0.999945173484644],
            P=[435.2046959714599, 0, 367.4517211914062, 0, 435.2046959714599, 252.2008514404297, 0, 0, 1]),
}


def rectify_maps(eye, width=752, height=480, calib=None):
    """cv::initUndistortRectifyMap(K, D, R, P[:3,:3], size, CV_32F, M1, M2) (stereo_euroc.cc:117) in float64 numpy:
    the (mapx, mapy) float32 planes the stereo driver feeds to cv::remap.  Host-side set-up, done once per camera."""
    c = (calib or EUROC_CALIB)[eye]
    K = np.array(c["K"], np.float64).reshape(3, 3)
    k1, k2, p1, p2, k3 = c["D"]
    iR = np.linalg.inv(np.array(c["P"], np.float64).reshape(3, 3) @ np.array(c["R"], np.float64).reshape(3, 3))
    j, i = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    X = iR[0, 0] * j + iR[0, 1] * i + iR[0, 2]
    Y = iR[1, 0] * j + iR[1, 1] * i + iR[1, 2]
    Wd = iR[2, 0] * j + iR[2, 1] * i + iR[2, 2]
    x, y = X / Wd, Y / Wd
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
    return (K[0, 0] * xd + K[0, 2]).astype(np.float32), (K[1, 1] * yd + K[1, 2]).astype(np.float32)


def make_vocabulary(k=10, L=3, seed=0, ragged=True, stop_fraction=0.05):
    """A random DBoW2-style vocabulary tree in ORBvoc.txt node order (breadth-first, like DBoW2's HKmeansStep output):
    returns (k, L, parent, is_leaf, desc, weight).  `ragged`: some inner nodes get fewer than k children and some
    branches end early (the real file is not a complete tree either); a fraction of the words is stopped (weight 0)."""
    rng = np.random.default_rng(seed)
    parent, is_leaf, desc, weight = [], [], [], []
    frontier = [(0, 0, rng.integers(0, 256, 32, dtype=np.uint8))]            # (node id, level, descriptor)
    next_id = 1
    while frontier:
        nid, lvl, d = frontier.pop(0)
        nchild = k if not ragged else int(rng.integers(2, k + 1))
        for _ in range(nchild):
            cd = d.copy()
            flips = rng.integers(0, 256, size=max(1, 48 >> lvl))                # children are perturbed copies of the parent
            for b in flips:
                cd[b >> 3] ^= np.uint8(1 << (b & 7))
            leaf = (lvl + 1 == L) or (ragged and lvl + 1 >= 2 and rng.random() < 0.1)
            parent.append(nid); is_leaf.append(1 if leaf else 0); desc.append(cd)
            weight.append(0.0 if (leaf and rng.random() < stop_fraction) else (float(rng.uniform(0.5, 9.0)) if leaf else 0.0))
            if not leaf:
                frontier.append((next_id, lvl + 1, cd))
            next_id += 1
    return (k, L, np.array(parent, np.int32), np.array(is_leaf, np.uint8), np.stack(desc).astype(np.uint8),
            np.array(weight, np.float64))
