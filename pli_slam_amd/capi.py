"""ctypes binding of libpli_frontend.so (the C ABI of include/pli_frontend.h).

This is the host-side plumbing used by the tests, bench.py and the Python
mirror of the reference interface (pli_slam_amd/frontend.py).  There is no CPU
fallback: if the HIP library is missing or no gfx950 device is visible every
entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library and its development build (-DPLI_DEV: the shelved schedules and the test / tuning switches of tools/README.md,
# read from the environment — the product library reads four documented variables and nothing else).  PLI_LIB_PATH (Python side):
# another build in place of both (tools/build_variant.sh); PLI_USE_DEV_LIB=1: the development build wherever the product one is asked
# for (the tools that run bench.py under environment switches).
LIB_PATH = os.environ.get("PLI_LIB_PATH") or os.path.join(_HERE, "csrc", "libpli_frontend.so")
DEV_LIB_PATH = os.environ.get("PLI_LIB_PATH") or os.path.join(_HERE, "csrc", "libpli_frontend_dev.so")

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

PLI_OK = 0
ERRORS = {-1: "PLI_ERR_INVALID", -2: "PLI_ERR_EMPTY_IMAGE", -3: "PLI_ERR_CAPACITY", -4: "PLI_ERR_HIP",
          -5: "PLI_ERR_NO_DEVICE", -6: "PLI_ERR_STATE"}
PARITY_TRIG_F32_ORB, PARITY_TRIG_F32_LSD, PARITY_TRIG_F32_LBD, PARITY_LSD_F64 = 1, 2, 4, 8
RUN_ORB, RUN_LINES, RUN_STEREO_POINTS, RUN_STEREO_LINES, RUN_ALL = 1, 2, 4, 8, 15
(DBG_PYRAMID_LEVEL, DBG_BLUR_LEVEL, DBG_FAST_CANDIDATES, DBG_LEVEL_KEYPOINTS, DBG_LSD_SCALED, DBG_LSD_ANGLE,
 DBG_LSD_SEGMENTS, DBG_LBD_DXDY, DBG_LSD_ORDER, DBG_LBD_FLOAT, DBG_STEREO_SAD, DBG_LSD_OWNER, DBG_LSD_SIZES) = range(1, 14)


class Config(C.Structure):
    """pli_frontend_config (include/pli_frontend.h)."""
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


class TableLayout(C.Structure):
    """pli_table_layout."""
    _fields_ = [
        ("record_bytes", C.c_int64), ("kp_cap", C.c_int32), ("kl_cap", C.c_int32),
        ("off_counts", C.c_int64), ("off_kp", C.c_int64 * 2), ("off_desc", C.c_int64 * 2),
        ("off_uright", C.c_int64), ("off_depth", C.c_int64), ("off_kl", C.c_int64 * 2),
        ("off_ldesc", C.c_int64 * 2), ("off_disp", C.c_int64), ("off_le", C.c_int64),
    ]


class TrackParams(C.Structure):
    """pli_track_params."""
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("th", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("mono", C.c_int32), ("check_orientation", C.c_int32), ("nnr_lines", C.c_float), ("reserved", C.c_int32)]


class TrackLayout(C.Structure):
    """pli_track_layout."""
    _fields_ = [("record_bytes", C.c_int64), ("off_counts", C.c_int64), ("off_best", C.c_int64), ("off_lines", C.c_int64),
                ("kp_cap", C.c_int32), ("kl_cap", C.c_int32)]


class PliError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(status, "PLI_ERR"), status, msg))
        self.status = status


_lib = None
_dev_lib = None

_PROTOS = {
    "pli_config_default": (None, [C.POINTER(Config), C.c_int32, C.c_int32]),
    "pli_kp_capacity": (C.c_int32, [C.POINTER(Config)]),
    "pli_kl_capacity": (C.c_int32, [C.POINTER(Config)]),
    "pli_ctx_create": (C.c_int32, [C.POINTER(Config), C.c_int32, C.POINTER(C.c_void_p)]),
    "pli_ctx_destroy": (None, [C.c_void_p]),
    "pli_last_error": (C.c_char_p, []),
    "pli_ctx_layout": (C.c_int32, [C.c_void_p, C.POINTER(TableLayout)]),
    "pli_ctx_set_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "pli_ctx_sync": (C.c_int32, [C.c_void_p]),
    "pli_set_rectify_maps": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "pli_batch_run": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint32,
                                  C.c_void_p]),
    "pli_batch_run_host": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                       C.c_uint32, C.c_void_p]),
    "pli_host_alloc": (C.c_int32, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "pli_host_free": (None, [C.c_void_p]),
    "pli_batch_submit_host": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                          C.c_uint32, C.c_void_p]),
    "pli_batch_wait": (C.c_int32, [C.c_void_p, C.c_int32]),
    "pli_track_layout_get": (C.c_int32, [C.c_void_p, C.POINTER(TrackLayout)]),
    "pli_batch_track": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(TrackParams), C.c_void_p]),
    "pli_frame_extract": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "pli_orb_extract": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                    C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]),
    "pli_orb_pyramid_level": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pli_line_extract": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]),
    "pli_lsd_round_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pli_selftest_hot_trig": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double)]),
    "pli_lsd_arena_words": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pli_set_stereo_camera": (C.c_int32, [C.c_void_p, C.c_float, C.c_float]),
    "pli_last_counts": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pli_stereo_match_points": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "pli_orb_extract_lapping": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pli_stereo_fisheye": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pli_stereo_fisheye_tables": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "pli_stereo_from_depth": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]),
    "pli_stereo_match_lines": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "pli_descriptor_distance": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pli_hamming_knn2": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p]),
    "pli_match_lines": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                    C.POINTER(C.c_int32)]),
    "pli_search_by_projection": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                             C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "pli_search_local_map": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_void_p, C.POINTER(C.c_int32)]),
    "pli_search_local_map_fisheye": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                                 C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "pli_match_nnr": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                  C.POINTER(C.c_int32)]),
    "pli_vocab_create": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_void_p)]),
    "pli_vocab_destroy": (None, [C.c_void_p]),
    "pli_bow_transform": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "pli_prof_enable": (C.c_int32, [C.c_void_p, C.c_int32]),
    "pli_prof_reset": (C.c_int32, [C.c_void_p]),
    "pli_prof_report": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    "pli_debug_enable": (C.c_int32, [C.c_void_p, C.c_int32]),
    "pli_debug_fetch": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                    C.POINTER(C.c_int64)]),
    "pli_version": (C.c_char_p, []),
    "pli_trace_ranges": (C.c_int64, []),
}


def exported_symbols():
    """Names include/pli_frontend.h declares (used by the CPU-side ABI test)."""
    return sorted(_PROTOS)


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError("%s is not built (python -c 'import __graft_entry__ as g; g.build()'); the front-end has no CPU fallback"
                           % os.path.basename(path))
    L = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        if os.environ.get("PLI_LIB_PATH") and not hasattr(L, name):
            continue        # (an A/B build of an earlier round, tools/ab_libs.sh: it lacks the entry points added since)
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def lib(dev=False):
    """Load libpli_frontend.so (dev: libpli_frontend_dev.so); raise if it has not been built."""
    global _lib, _dev_lib
    if dev or os.environ.get("PLI_USE_DEV_LIB", "0") not in ("", "0"):
        if _dev_lib is None:
            _dev_lib = _load(DEV_LIB_PATH)
        return _dev_lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def check(status):
    if status != PLI_OK:
        # (the message is thread-local state of the library that failed: with both builds loaded, the one that has a message)
        msgs = [L.pli_last_error().decode("utf-8", "replace") for L in (_lib, _dev_lib) if L is not None]
        raise PliError(status, " | ".join(m for m in msgs if m) or "?")


def default_config(width, height, **over):
    c = Config()
    lib().pli_config_default(C.byref(c), width, height)
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
