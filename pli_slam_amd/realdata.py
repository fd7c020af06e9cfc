"""Real photographs as test / bench input (tests/golden/real/photos.npz, made by tools/make_real_fixtures.py from the
photographs the build image ships with scikit-image: data, not reference code).

  photos()            name -> u8 grayscale array; `motorcycle_left` / `motorcycle_right` are a rectified Middlebury stereo pair
  motorcycle()        (left, right, ground-truth disparity of the left image in px, 0 = unknown)
  shifted_right(img)  a right eye for a single photograph: the left image displaced by a constant disparity + sensor noise
  frames_752x480(n)   n stereo frames of 752x480 cut from the photographs (bilinear enlargement by 1.5-2.2, random windows); the
                      motorcycle frames carry their real right eye, the others a displaced copy

Host-side input preparation only (numpy); nothing here is on the measured path.
"""
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHOTOS_NPZ = os.path.join(_ROOT, "tests", "golden", "real", "photos.npz")
_cache = {}


def photos():
    if "z" not in _cache:
        with np.load(PHOTOS_NPZ) as z:
            _cache["z"] = {k: np.ascontiguousarray(z[k]) for k in z.files}
    return {k: v for k, v in _cache["z"].items() if v.dtype == np.uint8}


def motorcycle():
    photos()
    z = _cache["z"]
    return z["motorcycle_left"], z["motorcycle_right"], z["motorcycle_disp_q64"].astype(np.float32) / np.float32(64.0)


def shifted_right(img, disparity=9, seed=0, sigma=2.0):
    """The view of a fronto-parallel plane from the right camera: columns move left by `disparity`; independent noise."""
    rng = np.random.default_rng(seed)
    h, w = img.shape
    xs = np.clip(np.arange(w) + int(disparity), 0, w - 1)
    r = img[:, xs].astype(np.float32) + rng.normal(0.0, sigma, (h, w)).astype(np.float32)
    return np.clip(np.rint(r), 0, 255).astype(np.uint8)


def _enlarge(img, s, x0, y0, w, h):
    """Window (x0, y0, w, h) of `img` enlarged by s (bilinear, pixel centres aligned)."""
    H, W = img.shape
    fx = (np.arange(w, dtype=np.float64) + x0 + 0.5) / s - 0.5
    fy = (np.arange(h, dtype=np.float64) + y0 + 0.5) / s - 0.5
    ix = np.clip(np.floor(fx).astype(np.int64), 0, W - 2); ax = np.clip(fx - ix, 0.0, 1.0).astype(np.float32)
    iy = np.clip(np.floor(fy).astype(np.int64), 0, H - 2); ay = np.clip(fy - iy, 0.0, 1.0).astype(np.float32)
    f = img.astype(np.float32)
    top = f[iy][:, ix] * (1 - ax) + f[iy][:, ix + 1] * ax
    bot = f[iy + 1][:, ix] * (1 - ax) + f[iy + 1][:, ix + 1] * ax
    return np.clip(np.rint(top * (1 - ay[:, None]) + bot * ay[:, None]), 0, 255).astype(np.uint8)


def frames_752x480(n, seed=0, w=752, h=480):
    """n stereo frames (left, right) of w x h from the photographs; deterministic in (n, seed)."""
    ph = photos()
    names = sorted(k for k in ph if k != "motorcycle_right")
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        name = names[i % len(names)]
        img = ph[name]
        H, W = img.shape
        smin = max(w / W, h / H) * 1.02
        s = float(max(smin, rng.uniform(1.5, 2.2)))
        x0 = int(rng.integers(0, int(W * s) - w + 1))
        y0 = int(rng.integers(0, int(H * s) - h + 1))
        L = _enlarge(img, s, x0, y0, w, h)
        if name == "motorcycle_left":
            R = _enlarge(ph["motorcycle_right"], s, x0, y0, w, h)     # (same window: disparities scale with s)
        else:
            R = shifted_right(L, disparity=int(rng.integers(4, 40)), seed=seed * 100003 + i)
        out.append((L, R))
    return out
