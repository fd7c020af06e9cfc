"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes wrapper over oracle/liboracle.so (the CPU restatement of the reference
front-end) and oracle/_ref/libpli_ref.so (the reference's own LineIterator.cpp +
gridStructure.cpp).  Import only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KEYPOINT_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                        ("response", "<f4"), ("octave", "<i4")])
KEYLINE_DT = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"),
                       ("pt_y", "<f4"), ("response", "<f4"), ("size", "<f4"),
                       ("startPointX", "<f4"), ("startPointY", "<f4"), ("endPointX", "<f4"),
                       ("endPointY", "<f4"), ("sPointInOctaveX", "<f4"), ("sPointInOctaveY", "<f4"),
                       ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"), ("lineLength", "<f4"),
                       ("numOfPixels", "<i4")])
PROJ_QUERY_DT = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("ur", "<f4"),
                          ("min_level", "<i4"), ("max_level", "<i4"), ("angle", "<f4"), ("valid", "<i4")])
assert KEYPOINT_DT.itemsize == 24 and KEYLINE_DT.itemsize == 68 and PROJ_QUERY_DT.itemsize == 32


PARITY_TRIG_F32_ORB, PARITY_TRIG_F32_LSD, PARITY_TRIG_F32_LBD, PARITY_LSD_F64 = 1, 2, 4, 8


class Config(C.Structure):
    """Mirror of pli_frontend_config (include/pli_frontend.h)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("max_frames", C.c_int32),
        ("orb_nfeatures", C.c_int32), ("orb_scale_factor", C.c_float), ("orb_nlevels", C.c_int32),
        ("orb_ini_th_fast", C.c_int32), ("orb_min_th_fast", C.c_int32),
        ("lsd_nfeatures", C.c_int32), ("lsd_refine", C.c_int32), ("lsd_n_bins", C.c_int32),
        ("max_lines", C.c_int32),
        ("min_line_length", C.c_double), ("lsd_scale", C.c_double), ("lsd_sigma_scale", C.c_double),
        ("lsd_quant", C.c_double), ("lsd_ang_th", C.c_double), ("lsd_log_eps", C.c_double),
        ("lsd_density_th", C.c_double),
        ("bf", C.c_float), ("fx", C.c_float), ("stereo_maxd_inf", C.c_int32),
        ("matching_s_ws", C.c_int32), ("best_lr_matches", C.c_int32),
        ("line_sim_th", C.c_double), ("stereo_overlap_th", C.c_double), ("min_ratio_12_l", C.c_double),
        ("ls_min_disp_ratio", C.c_double), ("min_disp", C.c_double), ("line_horiz_th", C.c_double),
        ("lsd_mode", C.c_int32), ("parity_flags", C.c_int32),
    ]


def default_config(width, height, **over):
    """Examples/Stereo/Config/EuRoC.yaml values (same as pli_config_default)."""
    c = Config()
    c.width, c.height, c.max_frames = width, height, 1
    c.orb_nfeatures, c.orb_scale_factor, c.orb_nlevels = 1200, 1.2, 8
    c.orb_ini_th_fast, c.orb_min_th_fast = 20, 7
    c.lsd_nfeatures, c.lsd_refine, c.lsd_n_bins, c.max_lines = 500, 0, 1024, max(4096, width * height // 64)
    c.min_line_length, c.lsd_scale, c.lsd_sigma_scale, c.lsd_quant = 0.025, 1.2, 0.6, 2.0
    c.lsd_ang_th, c.lsd_log_eps, c.lsd_density_th = 22.5, 1.0, 0.6
    c.bf, c.fx, c.stereo_maxd_inf = 47.90639384423901, 435.2046959714599, 0
    c.matching_s_ws, c.best_lr_matches = 10, 1
    c.line_sim_th, c.stereo_overlap_th, c.min_ratio_12_l = 0.75, 0.75, 0.9
    c.ls_min_disp_ratio, c.min_disp, c.line_horiz_th = 0.7, 1.0, 0.1
    c.lsd_mode, c.parity_flags = 0, PARITY_TRIG_F32_ORB | PARITY_LSD_F64
    for k, v in over.items():
        setattr(c, k, v)
    return c


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return so


_lib = None
_ref = None


def use_shipped_build():
    """Back to oracle/liboracle.so (after use_native_build)."""
    global _lib
    _lib = None
    lib()


def use_native_build():
    """cpu_baseline leg of bench.py: rebuild the oracle ON THIS MACHINE with the reference's own flags (-O3 -march=native,
    CMakeLists.txt:12-13) into oracle/_native/ and use that library from here on; the shipped liboracle.so is built -march=x86-64-v3
    in the build container, whose CPU is not the GPU box's.  Returns the flags in use (the shipped library's when g++ is missing)."""
    global _lib
    d = os.path.join(_HERE, "_native")
    so = os.path.join(d, "liboracle_native.so")
    try:
        os.makedirs(d, exist_ok=True)
        subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared",
                               "-o", so, os.path.join(_HERE, "oracle_capi.cpp")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        lib()
        return "-O3 -march=x86-64-v3 (shipped build; no compiler on this machine)"
    _lib = None
    real_build = globals()["build"]
    globals()["build"] = lambda force=False: so
    try:
        lib()
    finally:
        globals()["build"] = real_build
    return "-O3 -march=native (built on this machine)"


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_frame_create.restype = C.c_void_p
        _lib.orc_frame_create.argtypes = [C.POINTER(Config)]
        _lib.orc_frame_destroy.argtypes = [C.c_void_p]
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.orc_cv_round.argtypes = [C.c_double]
    return _lib


def ref():
    """The reference's own LineIterator/gridStructure build, or None when absent."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libpli_ref.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


class Frame:
    """One stereo frame processed by the oracle, with every intermediate kept."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_frame_create(C.byref(cfg)))

    def __del__(self):
        try:
            self.L.orc_frame_destroy(self.h)
        except Exception:
            pass

    def features_per_level(self):
        out = np.zeros(self.cfg.orb_nlevels, np.int32)
        self.L.orc_features_per_level(self.h, _p(out))
        return out

    def umax(self):
        out = np.zeros(16, np.int32)
        self.L.orc_umax(self.h, _p(out))
        return out

    def level_sigma2(self):
        out = np.zeros(self.cfg.orb_nlevels, np.float32)
        self.L.orc_level_sigma2(self.h, _p(out))
        return out

    # ---- ORB ----
    def orb_extract(self, eye, img):
        if img is None or img.size == 0:
            return self.L.orc_orb_extract(self.h, eye, None, 0, 0, C.c_int64(0)), None, None
        img = _u8(img)
        n = self.L.orc_orb_extract(self.h, eye, _p(img), img.shape[1], img.shape[0], C.c_int64(img.strides[0]))
        kp = np.zeros(max(n, 0), KEYPOINT_DT)
        desc = np.zeros((max(n, 0), 32), np.uint8)
        if n > 0:
            self.L.orc_get_keypoints(self.h, eye, _p(kp), _p(desc))
        return n, kp, desc

    def pyramid(self, eye, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        self.L.orc_get_pyramid(self.h, eye, level, int(blurred), None, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        if out.size:
            self.L.orc_get_pyramid(self.h, eye, level, int(blurred), _p(out), C.byref(w), C.byref(h))
        return out

    def level_points(self, eye, level, selected=False):
        n = self.L.orc_get_level_points(self.h, eye, level, int(selected), None, 0)
        out = np.zeros((n, 3), np.int32)
        if n:
            self.L.orc_get_level_points(self.h, eye, level, int(selected), _p(out), n)
        return out

    # ---- lines ----
    def line_extract(self, eye, img):
        img = _u8(img)
        n = self.L.orc_line_extract(self.h, eye, _p(img), img.shape[1], img.shape[0], C.c_int64(img.strides[0]))
        kl = np.zeros(n, KEYLINE_DT)
        desc = np.zeros((n, 32), np.uint8)
        if n:
            self.L.orc_get_keylines(self.h, eye, _p(kl), _p(desc))
        return n, kl, desc

    def lsd_dims(self, eye):
        w, h = C.c_int(), C.c_int()
        self.L.orc_get_lsd_dims(self.h, eye, C.byref(w), C.byref(h))
        return w.value, h.value

    def lsd_scaled(self, eye):
        w, h = self.lsd_dims(eye)
        out = np.zeros((h, w), np.uint8)
        self.L.orc_get_lsd_scaled(self.h, eye, _p(out))
        return out

    def lsd_scaled64(self, eye):
        """The scaled image of the CV_64FC1 pipeline (PARITY_LSD_F64), float64 (h, w)."""
        w, h = self.lsd_dims(eye)
        out = np.zeros((h, w), np.float64)
        self.L.orc_get_lsd_scaled64(self.h, eye, _p(out))
        return out

    def lsd_angle(self, eye):
        w, h = self.lsd_dims(eye)
        out = np.zeros((h, w), np.float32)
        self.L.orc_get_lsd_angle(self.h, eye, _p(out))
        return out

    def lsd_order(self, eye):
        n = self.L.orc_get_lsd_order(self.h, eye, None, 0)
        out = np.zeros(n, np.int32)
        self.L.orc_get_lsd_order(self.h, eye, _p(out), n)
        return out

    def lsd_segments(self, eye):
        n = self.L.orc_get_lsd_segments(self.h, eye, None, 0)
        out = np.zeros((n, 4), np.float32)
        if n:
            self.L.orc_get_lsd_segments(self.h, eye, _p(out), n)
        return out

    def lbd_dxdy(self, eye, shape):
        dx = np.zeros(shape, np.int16)
        dy = np.zeros(shape, np.int16)
        self.L.orc_get_lbd_dxdy(self.h, eye, _p(dx), _p(dy))
        return dx, dy

    def lbd_float(self, eye, n):
        out = np.zeros((n, 72), np.float32)
        if n:
            self.L.orc_get_lbd_float(self.h, eye, _p(out))
        return out

    # ---- stereo ----
    def stereo_points(self):
        n = self.L.orc_stereo_points(self.h, None, None, None, None)
        ur, dp = np.zeros(n, np.float32), np.zeros(n, np.float32)
        bi, sad = np.zeros(n, np.int32), np.zeros(n, np.int32)
        if n:
            self.L.orc_stereo_points(self.h, _p(ur), _p(dp), _p(bi), _p(sad))
        return ur, dp, bi, sad

    def stereo_lines(self):
        n = self.L.orc_stereo_lines(self.h, None, None, None)
        disp, le, m = np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float64), np.zeros(n, np.int32)
        if n:
            self.L.orc_stereo_lines(self.h, _p(disp), _p(le), _p(m))
        return disp, le, m

    def run(self, left, right):
        left, right = _u8(left), _u8(right)
        return self.L.orc_frame_run(self.h, _p(left), _p(right), left.shape[1], left.shape[0],
                                    C.c_int64(left.strides[0]))


# ---- stateless helpers ----
def descriptor_distance(a, b):
    a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
    out = np.zeros(a.shape[0], np.int32)
    lib().orc_descriptor_distance(_p(a), _p(b), a.shape[0], _p(out))
    return out


def knn2(q, t):
    q, t = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((q.shape[0], 2), np.int32)
    dist = np.zeros((q.shape[0], 2), np.int32)
    lib().orc_knn2(_p(q), q.shape[0], _p(t), t.shape[0], _p(idx), _p(dist))
    return idx, dist


def match_lines(d1, d2, nnr, best_lr=True):
    d1, d2 = np.ascontiguousarray(d1, np.uint8), np.ascontiguousarray(d2, np.uint8)
    m = np.full(d1.shape[0], -1, np.int32)
    n = lib().orc_match_lines(_p(d1), d1.shape[0], _p(d2), d2.shape[0], C.c_float(nnr), int(best_lr), _p(m))
    return n, m


def search_by_projection(q, qdesc, kp, desc, uright, bounds, check_ori=True, occupied=None, with_raw=False):
    q = np.ascontiguousarray(q, PROJ_QUERY_DT)
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    kp = np.ascontiguousarray(kp, KEYPOINT_DT)
    desc = np.ascontiguousarray(desc, np.uint8)
    uright = np.ascontiguousarray(uright, np.float32)
    best = np.full(q.shape[0], -1, np.int32)
    raw = np.full(q.shape[0], -1, np.int32)
    oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    n = lib().orc_search_by_projection(_p(q), _p(qdesc), q.shape[0], _p(kp), _p(desc), _p(uright), kp.shape[0],
                                       C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]),
                                       C.c_float(bounds[3]), int(check_ori), _p(best), None if oc is None else _p(oc), _p(raw))
    return (n, best, raw) if with_raw else (n, best)


def track_queries(last_kp, last_depth, Tlw, Tcw, fx, fy, cx, cy, bf, th, mono, scale_factors):
    """Projection part of SearchByProjection(CurrentFrame, LastFrame) (ORBmatcher.cc:2190-2244): the pli_proj_query table."""
    kp = np.ascontiguousarray(last_kp, KEYPOINT_DT)
    dp = np.ascontiguousarray(last_depth, np.float32)
    Tl = np.ascontiguousarray(Tlw, np.float32).reshape(12)
    Tc = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    q = np.zeros(kp.shape[0], PROJ_QUERY_DT)
    lib().orc_track_queries(_p(kp), _p(dp), kp.shape[0], _p(Tl), _p(Tc), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                            C.c_float(cy), C.c_float(bf), C.c_float(th), int(mono), _p(sf), _p(q))
    return q


class Vocabulary:
    """DBoW2 vocabulary tree (node list as in ORBvoc.txt: parent, is-word flag, 32-byte descriptor, weight)."""

    def __init__(self, k, L, parent, is_leaf, desc, weight):
        parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
        L_ = lib()
        L_.orc_vocab_create.restype = C.c_void_p
        self.h = C.c_void_p(L_.orc_vocab_create(k, L, parent.shape[0], _p(parent), _p(is_leaf), _p(desc), _p(weight)))

    def __del__(self):
        try:
            lib().orc_vocab_destroy(self.h)
        except Exception:
            pass

    def descend(self, feat, levelsup=4):
        feat = np.ascontiguousarray(feat, np.uint8)
        n = feat.shape[0]
        word = np.zeros(n, np.int32); weight = np.zeros(n, np.float64); node = np.zeros(n, np.int32)
        lib().orc_bow_descend(self.h, _p(feat), n, levelsup, _p(word), _p(weight), _p(node))
        return word, weight, node

    def bow_vector(self, feat, levelsup=4):
        feat = np.ascontiguousarray(feat, np.uint8)
        n = feat.shape[0]
        words = np.zeros(max(n, 1), np.uint32); vals = np.zeros(max(n, 1), np.float64)
        k = lib().orc_bow_vector(self.h, _p(feat), n, levelsup, _p(words), _p(vals), max(n, 1))
        return words[:k].copy(), vals[:k].copy()


def stereo_from_depth(kp, depth, bf):
    kp = np.ascontiguousarray(kp, KEYPOINT_DT)
    d = np.ascontiguousarray(depth, np.float32)
    ur = np.zeros(kp.shape[0], np.float32)
    dp = np.zeros(kp.shape[0], np.float32)
    lib().orc_stereo_from_depth(_p(kp), kp.shape[0], _p(d), C.c_int64(d.shape[1]), d.shape[1], d.shape[0], C.c_float(bf), _p(ur), _p(dp))
    return ur, dp


def remap_linear(img, mapx, mapy):
    """cv::remap(img, M1, M2, INTER_LINEAR) on an 8U image (constant-0 border)."""
    img = np.ascontiguousarray(img, np.uint8)
    mx, my = np.ascontiguousarray(mapx, np.float32), np.ascontiguousarray(mapy, np.float32)
    dst = np.zeros_like(img)
    lib().orc_remap_linear(_p(img), img.shape[1], img.shape[0], C.c_int64(img.strides[0]), _p(mx), _p(my), _p(dst))
    return dst


def remap_weights(fx, fy):
    w = np.zeros(4, np.int32)
    lib().orc_remap_weights(fx, fy, _p(w))
    return w


def search_local_map(q, qdesc, kp, desc, uright, occupied, bounds, nnratio):
    q = np.ascontiguousarray(q, PROJ_QUERY_DT)
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    kp = np.ascontiguousarray(kp, KEYPOINT_DT)
    desc = np.ascontiguousarray(desc, np.uint8)
    uright = np.ascontiguousarray(uright, np.float32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    best = np.full(q.shape[0], -1, np.int32)
    n = lib().orc_search_local_map(_p(q), _p(qdesc), q.shape[0], _p(kp), _p(desc), _p(uright), _p(occ), kp.shape[0],
                                   C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]),
                                   C.c_float(nnratio), _p(best))
    return n, best


def search_local_map_fisheye(ql, qr, qdesc, kpl, dl, occl, l2r, kpr, dr, occr, r2l, bounds, nnratio):
    ql = np.ascontiguousarray(ql, PROJ_QUERY_DT)
    qr = np.ascontiguousarray(qr, PROJ_QUERY_DT)
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    kpl, kpr = np.ascontiguousarray(kpl, KEYPOINT_DT), np.ascontiguousarray(kpr, KEYPOINT_DT)
    dl, dr = np.ascontiguousarray(dl, np.uint8), np.ascontiguousarray(dr, np.uint8)
    ol = None if occl is None else np.ascontiguousarray(occl, np.uint8)
    orr = None if occr is None else np.ascontiguousarray(occr, np.uint8)
    l2r, r2l = np.ascontiguousarray(l2r, np.int32), np.ascontiguousarray(r2l, np.int32)
    mpl = np.full(kpl.shape[0], -1, np.int32)
    mpr = np.full(kpr.shape[0], -1, np.int32)
    n = lib().orc_search_local_map_fisheye(_p(ql), _p(qr), _p(qdesc), ql.shape[0], _p(kpl), _p(dl), _p(ol), _p(l2r), kpl.shape[0],
                                           _p(kpr), _p(dr), _p(orr), _p(r2l), kpr.shape[0],
                                           C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]),
                                           C.c_float(nnratio), _p(mpl), _p(mpr))
    return n, mpl, mpr


def match_nnr(d1, d2, nnr):
    d1, d2 = np.ascontiguousarray(d1, np.uint8), np.ascontiguousarray(d2, np.uint8)
    m = np.full(d1.shape[0], -1, np.int32)
    n = lib().orc_match_nnr(_p(d1), d1.shape[0], _p(d2), d2.shape[0], C.c_float(nnr), _p(m))
    return n, m


def lapping_order(kp, lap0, lap1):
    """ORBextractor.cc:1135-1144: (order with order[dst] = src, mono count)."""
    kp = np.ascontiguousarray(kp, KEYPOINT_DT)
    order = np.zeros(kp.shape[0], np.int32)
    mono = lib().orc_lapping_order(_p(kp), kp.shape[0], int(lap0), int(lap1), _p(order))
    return order, mono


def stereo_fisheye(kpL, descL, mono_left, kpR, descR, mono_right, cam1, cam2, Rlr, tlr, sigma2):
    """Frame::ComputeStereoFishEyeMatches (Frame.cc:1577-1618) on tables in lapping order."""
    kpL, kpR = np.ascontiguousarray(kpL, KEYPOINT_DT), np.ascontiguousarray(kpR, KEYPOINT_DT)
    descL, descR = np.ascontiguousarray(descL, np.uint8), np.ascontiguousarray(descR, np.uint8)
    c1, c2 = np.ascontiguousarray(cam1, np.float32), np.ascontiguousarray(cam2, np.float32)
    R, t = np.ascontiguousarray(Rlr, np.float32).reshape(9), np.ascontiguousarray(tlr, np.float32).reshape(3)
    s2 = np.ascontiguousarray(sigma2, np.float32)
    nl, nr = kpL.shape[0], kpR.shape[0]
    l2r, r2l = np.zeros(nl, np.int32), np.zeros(nr, np.int32)
    depth, p3d = np.zeros(nl, np.float32), np.zeros((nl, 3), np.float32)
    n = lib().orc_stereo_fisheye(_p(kpL), _p(descL), nl, int(mono_left), _p(kpR), _p(descR), nr, int(mono_right), _p(c1), _p(c2),
                                 _p(R), _p(t), _p(s2), _p(l2r), _p(r2l), _p(depth), _p(p3d))
    return n, l2r, r2l, depth, p3d


def kb8_unproject(cam, u, v):
    c = np.ascontiguousarray(cam, np.float32)
    r = np.zeros(3, np.float32)
    lib().orc_kb8_unproject(_p(c), C.c_float(u), C.c_float(v), _p(r))
    return r


def kb8_project(cam, p):
    c, p = np.ascontiguousarray(cam, np.float32), np.ascontiguousarray(p, np.float32)
    uv = np.zeros(2, np.float32)
    lib().orc_kb8_project(_p(c), _p(p), _p(uv))
    return uv


def stereo_lines_tables(cfg, kl, dl, kr, dr, w, h):
    kl, kr = np.ascontiguousarray(kl, KEYLINE_DT), np.ascontiguousarray(kr, KEYLINE_DT)
    dl, dr = np.ascontiguousarray(dl, np.uint8), np.ascontiguousarray(dr, np.uint8)
    n1 = kl.shape[0]
    disp, le, m = np.zeros((n1, 2), np.float32), np.zeros((n1, 3), np.float64), np.zeros(n1, np.int32)
    lib().orc_stereo_lines_tables(C.byref(cfg), _p(kl), _p(dl), n1, _p(kr), _p(dr), kr.shape[0], w, h,
                                  _p(disp), _p(le), _p(m))
    return disp, le, m


def _coords(fn, x1, y1, x2, y2):
    buf = np.zeros((4096, 2), np.int32)
    n = fn(C.c_double(x1), C.c_double(y1), C.c_double(x2), C.c_double(y2), _p(buf), 4096)
    return buf[:n].copy()


def line_coords(x1, y1, x2, y2):
    return _coords(lib().orc_line_coords, x1, y1, x2, y2)


def ref_line_coords(x1, y1, x2, y2):
    return _coords(ref().ref_line_coords, x1, y1, x2, y2)


def _gridq(fn, segs, rows, cols, qx, qy, win):
    segs = np.ascontiguousarray(segs, np.float64)
    buf = np.zeros(max(1, segs.shape[0]), np.int32)
    n = fn(_p(segs), segs.shape[0], rows, cols, qx, qy, win[0], win[1], win[2], win[3], _p(buf), buf.shape[0])
    return buf[:n].copy()


def grid_query(segs, rows, cols, qx, qy, win):
    return _gridq(lib().orc_grid_query, segs, rows, cols, qx, qy, win)


def ref_grid_query(segs, rows, cols, qx, qy, win):
    return _gridq(ref().ref_grid_query, segs, rows, cols, qx, qy, win)


def fast_image(img, th):
    """cv::FAST(img, kps, th, nonmaxSuppression=true) on the whole image: (n, 3) float32 x, y, response, raster order."""
    img = _u8(img)
    cap = img.size // 4
    out = np.zeros((cap, 3), np.float32)
    n = lib().orc_fast_image(_p(img), img.shape[1], img.shape[0], int(th), _p(out), cap)
    return out[:n].copy()


def glibc_cosf(x):
    L_ = lib(); L_.orc_glibc_cosf.restype = C.c_float; L_.orc_glibc_cosf.argtypes = [C.c_float]
    return float(L_.orc_glibc_cosf(C.c_float(x)))


def glibc_sinf(x):
    L_ = lib(); L_.orc_glibc_sinf.restype = C.c_float; L_.orc_glibc_sinf.argtypes = [C.c_float]
    return float(L_.orc_glibc_sinf(C.c_float(x)))


def sincosf_selfcheck(first, last, step):
    """Floats (by bit pattern) in [first, last) on which the sincosf restatement differs from this machine's cosf/sinf."""
    L_ = lib(); L_.orc_sincosf_selfcheck.restype = C.c_long; L_.orc_sincosf_selfcheck.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    return int(L_.orc_sincosf_selfcheck(first, last, step))


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def cv_round(v):
    return int(lib().orc_cv_round(C.c_double(v)))


def gauss_kernel(n, sigma):
    out = np.zeros(n, np.int32)
    lib().orc_gauss_kernel(n, C.c_double(sigma), _p(out))
    return out


def gaussian_blur(img, n, sigma):
    img = _u8(img)
    out = np.zeros_like(img)
    lib().orc_gaussian_blur(_p(img), img.shape[1], img.shape[0], n, C.c_double(sigma), _p(out))
    return out


def resize(img, dw, dh, sx, sy):
    img = _u8(img)
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize(_p(img), img.shape[1], img.shape[0], dw, dh, C.c_double(sx), C.c_double(sy), _p(out))
    return out


def fast_arc(img, x, y):
    img = _u8(img)
    return int(lib().orc_fast_arc(_p(img), img.shape[1], img.shape[0], x, y))


def sobel(img):
    img = _u8(img)
    dx, dy = np.zeros(img.shape, np.int16), np.zeros(img.shape, np.int16)
    lib().orc_sobel(_p(img), img.shape[1], img.shape[0], _p(dx), _p(dy))
    return dx, dy


def lbd_weights():
    L, G = np.zeros(21, np.float32), np.zeros(63, np.float32)
    lib().orc_lbd_weights(_p(L), _p(G))
    return L, G


def orb_descriptor(img, x, y, angle, trig_f32=True):
    img = _u8(img)
    d = np.zeros(32, np.uint8)
    lib().orc_orb_descriptor(_p(img), img.shape[1], img.shape[0], x, y, C.c_float(angle), _p(d), int(trig_f32))
    return d
