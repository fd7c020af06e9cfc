"""A second opinion for cv::LineSegmentDetector, written in numpy from the algorithm's description in SURVEY.md
Appendix A (items 1-8 of `createLineSegmentDetector(...)->detect`) — NOT from oracle/line_oracle.hpp: Gaussian blur and bilinear
enlargement by their textbook definitions in float64, the 2x2 level-line gradient, the 1024-bin pseudo-ordering, greedy 8-connected
region growing with the running mean angle, the inertia-axis rectangle.  Two variants of the one library primitive the description
leaves to OpenCV, the angle function: (A) the 7th-order polynomial of `fastAtan2` as Appendix A gives it, (B) the true arctangent.

On small rendered scenes the oracle's segments must agree with (A) to a hundredth of a pixel and with (B)
within the contract's 0.5 px for (nearly) every segment: whatever the restatement in C++ got from reading OpenCV, an
implementation that never saw it finds the same lines.  CPU only (64x64 images; pure-Python loops)."""
import math

import numpy as np
import pytest

PI = math.pi


def fast_atan2_deg(y, x):
    """Appendix A: fastAtan2(y, x) in degrees, float32 arithmetic."""
    f = np.float32
    p1, p3, p5, p7 = (f(0.9997878412794807 * (180 / PI)), f(-0.3258083974640975 * (180 / PI)), f(0.1555786518463281 * (180 / PI)),
                      f(-0.04432655554792128 * (180 / PI)))
    x, y = f(x), f(y)
    ax, ay = abs(x), abs(y)
    eps = f(2.220446049250313e-16)
    if ax >= ay:
        c = f(ay / f(ax + eps)); c2 = f(c * c)
        a = f(f(f(f(f(f(p7 * c2) + p5) * c2) + p3) * c2 + p1) * c)
    else:
        c = f(ax / f(ay + eps)); c2 = f(c * c)
        a = f(f(90.0) - f(f(f(f(f(f(p7 * c2) + p5) * c2) + p3) * c2 + p1) * c))
    if x < 0:
        a = f(f(180.0) - a)
    if y < 0:
        a = f(f(360.0) - a)
    return float(a)


def true_atan2_deg(y, x):
    a = math.degrees(math.atan2(y, x))
    return a + 360.0 if a < 0 else a


def reflect101(i, n):
    if i < 0:
        return -i
    if i >= n:
        return 2 * n - 2 - i
    return i


def lsd_numpy(img, atan2_deg, scale=1.2, sigma_scale=0.6, quant=2.0, ang_th=22.5, n_bins=1024):
    """Appendix A, `createLineSegmentDetector(refine = NONE, ...)->detect`, items 1-8.  Returns [(x1, y1, x2, y2)] in detection order."""
    prec = PI * ang_th / 180.0
    p = ang_th / 180.0
    rho = quant / math.sin(prec)
    src = img.astype(np.float64)
    h0, w0 = src.shape
    # 2. Gaussian blur (sigma = sigma_scale for scale >= 1), then enlargement by `scale` with pixel-centre-aligned bilinear weights
    sigma = sigma_scale / scale if scale < 1 else sigma_scale
    half = int(math.ceil(sigma * math.sqrt(2 * 3 * math.log(10.0))))
    k = np.array([math.exp(-(i * i) / (2 * sigma * sigma)) for i in range(-half, half + 1)])
    k /= k.sum()
    rows = np.zeros_like(src)
    for y in range(h0):
        for x in range(w0):
            rows[y, x] = sum(k[j + half] * src[y, reflect101(x + j, w0)] for j in range(-half, half + 1))
    blur = np.zeros_like(src)
    for y in range(h0):
        for x in range(w0):
            blur[y, x] = sum(k[j + half] * rows[reflect101(y + j, h0), x] for j in range(-half, half + 1))
    W, H = int(round(w0 * scale)), int(round(h0 * scale))
    I = np.zeros((H, W))
    for dy in range(H):
        fy = (dy + 0.5) / scale - 0.5
        sy = math.floor(fy); wy = fy - sy
        y0, y1 = min(max(sy, 0), h0 - 1), min(max(sy + 1, 0), h0 - 1)
        for dx in range(W):
            fx = (dx + 0.5) / scale - 0.5
            sx = math.floor(fx); wx = fx - sx
            if sx < 0:
                sx, wx = 0, 0.0
            if sx >= w0 - 1:
                sx, wx = w0 - 1, 0.0
            x1 = min(sx + 1, w0 - 1)
            top = blur[y0, sx] * (1 - wx) + blur[y0, x1] * wx
            bot = blur[y1, sx] * (1 - wx) + blur[y1, x1] * wx
            I[dy, dx] = top * (1 - wy) + bot * wy
    # 3. level-line angles
    NOTDEF = None
    ang = [[NOTDEF] * W for _ in range(H)]
    mod = np.zeros((H, W))
    for y in range(H - 1):
        for x in range(W - 1):
            DA = I[y + 1, x + 1] - I[y, x]
            BC = I[y, x + 1] - I[y + 1, x]
            gx, gy = DA + BC, DA - BC
            n = math.sqrt((gx * gx + gy * gy) / 4.0)
            mod[y, x] = n
            if n > rho:
                ang[y][x] = atan2_deg(gx, -gy) * PI / 180.0
    max_grad = mod[:H - 1, :W - 1].max()
    # 4. seeds: bin descending, raster order inside a bin
    seeds = []
    coef = (n_bins - 1) / max_grad if max_grad > 0 else 0.0
    for y in range(H - 1):
        for x in range(W - 1):
            b = min(int(mod[y, x] * coef), n_bins - 1)
            seeds.append((-b, y * W + x))
    seeds.sort()
    # 5. smallest region that can be a segment
    log_nt = 5 * (math.log10(W) + math.log10(H)) / 2 + math.log10(11.0)
    min_reg = int(-log_nt / math.log10(p))

    def adiff(a, b):
        d = abs(a - b)
        if d > 1.5 * PI:
            d = abs(d - 2 * PI)
        return d

    used = [[False] * W for _ in range(H)]
    out = []
    for _, pix in seeds:
        sy, sx = divmod(pix, W)
        if used[sy][sx] or ang[sy][sx] is NOTDEF:
            continue
        # 6. region growing
        reg = [(sx, sy)]
        used[sy][sx] = True
        reg_angle = ang[sy][sx]
        sumdx, sumdy = np.float32(math.cos(reg_angle)), np.float32(math.sin(reg_angle))
        i = 0
        while i < len(reg):
            px, py = reg[i]
            for yy in range(max(py - 1, 0), min(py + 1, H - 1) + 1):
                for xx in range(max(px - 1, 0), min(px + 1, W - 1) + 1):
                    a = ang[yy][xx]
                    if used[yy][xx] or a is NOTDEF or adiff(reg_angle, a) > prec:
                        continue
                    used[yy][xx] = True
                    reg.append((xx, yy))
                    af = float(np.float32(a))
                    sumdx = np.float32(sumdx + np.float32(math.cos(af)))
                    sumdy = np.float32(sumdy + np.float32(math.sin(af)))
                    reg_angle = atan2_deg(float(sumdy), float(sumdx)) * PI / 180.0
            i += 1
        if len(reg) < min_reg:
            continue
        # 7. rectangle: weighted centroid, main inertia axis, extent along it
        wsum = sum(mod[y, x] for x, y in reg)
        cx = sum(x * mod[y, x] for x, y in reg) / wsum
        cy = sum(y * mod[y, x] for x, y in reg) / wsum
        Ixx = sum((y - cy) ** 2 * mod[y, x] for x, y in reg)
        Iyy = sum((x - cx) ** 2 * mod[y, x] for x, y in reg)
        Ixy = -sum((x - cx) * (y - cy) * mod[y, x] for x, y in reg)
        lam = 0.5 * (Ixx + Iyy - math.sqrt((Ixx - Iyy) ** 2 + 4 * Ixy * Ixy))
        theta = (atan2_deg(lam - Ixx, Ixy) if abs(Ixx) > abs(Iyy) else atan2_deg(Ixy, lam - Iyy)) * PI / 180.0
        d = theta - reg_angle
        while d <= -PI:
            d += 2 * PI
        while d > PI:
            d -= 2 * PI
        if abs(d) > prec:
            theta += PI
        dx, dy = math.cos(theta), math.sin(theta)
        ls = [(x - cx) * dx + (y - cy) * dy for x, y in reg]
        lmin, lmax = min(ls), max(ls)
        x1, y1, x2, y2 = cx + lmin * dx, cy + lmin * dy, cx + lmax * dx, cy + lmax * dy
        # 8. back to the input image's frame
        out.append(((x1 + 0.5) / scale, (y1 + 0.5) / scale, (x2 + 0.5) / scale, (y2 + 0.5) / scale))
    return out


def scenes():
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:64, 0:64]
    a = np.full((64, 64), 40.0)
    a[14:50, 10:54] = 200.0                                           # a bright rectangle: four edges
    b = np.full((64, 64), 60.0)
    b[(xx + yy > 50) & (xx + yy < 78)] = 190.0                        # a diagonal band: two long edges
    b[30:34, :] = 120.0
    c = np.full((64, 64), 220.0)
    c[(yy > 8) & (yy < 56) & (xx > 6 + (yy - 8) * 0.45) & (xx < 58 - (yy - 8) * 0.45)] = 50.0      # a dark trapezoid
    out = []
    for im in (a, b, c):
        im = im + rng.normal(0.0, 1.5, im.shape)
        out.append(np.clip(np.rint(im), 0, 255).astype(np.uint8))
    # ... and three 64x64 windows of real photographs (tests/golden/real): 55, 18 and 30 segments
    from pli_slam_amd import realdata
    ph = realdata.photos()
    for name, y, x in (("camera", 100, 200), ("text", 40, 100), ("motorcycle_left", 200, 300)):
        out.append(np.ascontiguousarray(ph[name][y:y + 64, x:x + 64]))
    return out


def nearest_endpoint_error(seg, others):
    """max over the two endpoints of the distance to the best matching segment's endpoints (either orientation)"""
    best = 1e9
    for o in others:
        d1 = max(math.hypot(seg[0] - o[0], seg[1] - o[1]), math.hypot(seg[2] - o[2], seg[3] - o[3]))
        d2 = max(math.hypot(seg[0] - o[2], seg[1] - o[3]), math.hypot(seg[2] - o[0], seg[3] - o[1]))
        best = min(best, d1, d2)
    return best


@pytest.mark.parametrize("k", [0, 1, 2, 3, 4, 5])
def test_oracle_lsd_against_an_implementation_written_from_the_description(oracle, k):
    from pli_slam_amd import capi
    po = oracle
    img = scenes()[k]
    cfg = capi.default_config(64, 64, orb_nfeatures=100, lsd_nfeatures=0, max_frames=1)
    fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
    fr.line_extract(0, img)
    want = [tuple(float(v) for v in s) for s in np.asarray(fr.lsd_segments(0)).reshape(-1, 4)]
    assert len(want) >= 2, "the scene was meant to hold line segments"
    # (A) the same angle primitive: the same growth decisions, so the same segments up to the rounding of differently ordered sums
    got_a = lsd_numpy(img, fast_atan2_deg)
    assert len(got_a) == len(want), "variant A finds %d segments, the oracle %d" % (len(got_a), len(want))
    err_a = [nearest_endpoint_error(s, got_a) for s in want]
    assert max(err_a) < 0.01, "endpoints differ by up to %.4f px with the same angle function" % max(err_a)
    # (B) the true arctangent: a pixel within 0.3 degrees of the tolerance may change sides, the lines must not move
    got_b = lsd_numpy(img, true_atan2_deg)
    err_b = [nearest_endpoint_error(s, got_b) for s in want]
    within = sum(e <= 0.5 for e in err_b)
    print("scene %d: %d segments; variant A max endpoint error %.5f px; variant B: %d/%d within 0.5 px (errors %s)" % (
        k, len(want), max(err_a), within, len(want), ["%.2f" % e for e in err_b]))
    assert abs(len(got_b) - len(want)) <= 1
    assert within >= len(want) - 1, "more than one segment moved by over 0.5 px under the true arctangent: %s" % err_b
