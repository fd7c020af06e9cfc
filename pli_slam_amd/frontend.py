"""Host-side mirror of the reference's front-end interface over the C ABI.

Names and argument meaning follow the reference classes so that parity tests
read like the reference's call sites:

  ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)(image, mask, vLappingArea)
      reference include/ORBextractor.h:53-63, src/ORBextractor.cc:1068
  Lineextractor(lsd_nfeatures, min_line_length, lsd_refine, lsd_scale, ...)(image, mask)
      reference include/LineExtractor.h:44-51, src/LineExtractor.cc:31
  Frontend.compute_stereo_matches() / compute_stereo_matches_lines()
      reference Frame::ComputeStereoMatches / _Lines, src/Frame.cc:976,1156
  DescriptorDistance, match, SearchByProjection
      reference src/ORBmatcher.cc:2495,2179 and src/LineMatcher.cpp:201

All arithmetic happens in the HIP library; this file only moves buffers.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import KEYLINE_DT, KEYPOINT_DT, PROJ_QUERY_DT, check, ptr


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim != 2:
        raise ValueError("grayscale u8 image expected")
    return img


PRODUCT_ENV = ("PLI_ROCTX", "PLI_SYNC_DEBUG", "PLI_TX_TAIL", "PLI_LSD_MODE")      # what libpli_frontend.so itself reads
PYTHON_ENV = ("PLI_LIB_PATH", "PLI_USE_DEV_LIB", "PLI_FUSION_WAIT_MS")             # read by this package / the C++ adapters


def needs_dev_library(cfg):
    """The development build is the one that knows lsd_mode 1 (the lane relaxation) and the environment switches of tools/README.md."""
    return cfg.lsd_mode == 1 or any(k.startswith("PLI_") and k not in PRODUCT_ENV + PYTHON_ENV for k in os.environ)


class Frontend:
    """One pli_ctx: device buffers + stream for up to `max_frames` stereo frames."""

    def __init__(self, cfg, device=0, dev=None):
        """dev: the development build of the library (libpli_frontend_dev.so) instead of the product one; None = the product build
        unless the configuration or the environment asks for something only the development build has (the dev-switch tests and
        tools).  bench.py passes False: its line is the product library's."""
        self.cfg = cfg
        self.dev = needs_dev_library(cfg) if dev is None else bool(dev)
        self.L = capi.lib(dev=self.dev)
        self.h = C.c_void_p()
        check(self.L.pli_ctx_create(C.byref(cfg), device, C.byref(self.h)))
        self.layout = capi.TableLayout()
        check(self.L.pli_ctx_layout(self.h, C.byref(self.layout)))
        self.kp_cap = self.layout.kp_cap
        self.kl_cap = self.layout.kl_cap

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.pli_ctx_destroy(self.h)
            self.h = C.c_void_p()
        for p in getattr(self, "_pinned", []):
            self.L.pli_host_free(p)
        self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- per-call drop-ins -------------------------------------------------
    def orb_extract(self, eye, image):
        """ORBextractor::operator(); returns (n, keypoints, descriptors). n = -1 for an empty image."""
        if image is None or getattr(image, "size", 0) == 0:
            n = C.c_int32()
            st = self.L.pli_orb_extract(self.h, eye, None, 0, 0, 0, None, 0, None, C.byref(n))
            if st != -2:
                check(st)
            return -1, np.zeros(0, KEYPOINT_DT), np.zeros((0, 32), np.uint8)
        image = _u8(image)
        kp = np.zeros(self.kp_cap, KEYPOINT_DT)
        desc = np.zeros((self.kp_cap, 32), np.uint8)
        n = C.c_int32()
        check(self.L.pli_orb_extract(self.h, eye, ptr(image), image.shape[1], image.shape[0], image.strides[0],
                                     ptr(kp), self.kp_cap, ptr(desc), C.byref(n)))
        return n.value, kp[:n.value].copy(), desc[:n.value].copy()

    def pyramid_level(self, eye, level):
        w, h = C.c_int32(), C.c_int32()
        check(self.L.pli_orb_pyramid_level(self.h, eye, level, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        check(self.L.pli_orb_pyramid_level(self.h, eye, level, ptr(out), out.size, C.byref(w), C.byref(h)))
        return out

    def line_extract(self, eye, image):
        """Lineextractor::operator(); returns (n, keylines, descriptors)."""
        image = _u8(image)
        kl = np.zeros(self.kl_cap, KEYLINE_DT)
        desc = np.zeros((self.kl_cap, 32), np.uint8)
        n = C.c_int32()
        check(self.L.pli_line_extract(self.h, eye, ptr(image), image.shape[1], image.shape[0], image.strides[0],
                                      ptr(kl), self.kl_cap, ptr(desc), C.byref(n)))
        return n.value, kl[:n.value].copy(), desc[:n.value].copy()

    def compute_stereo_matches(self):
        """Frame::ComputeStereoMatches: (mvuRight, mvDepth) for the left keypoints."""
        ur = np.zeros(self.kp_cap, np.float32)
        dp = np.zeros(self.kp_cap, np.float32)
        check(self.L.pli_stereo_match_points(self.h, ptr(ur), ptr(dp), self.kp_cap))
        return ur, dp

    def compute_stereo_matches_lines(self):
        """Frame::ComputeStereoMatches_Lines: (mvDisparity_l [n,2], mvle_l [n,3])."""
        disp = np.zeros((self.kl_cap, 2), np.float32)
        le = np.zeros((self.kl_cap, 3), np.float64)
        check(self.L.pli_stereo_match_lines(self.h, ptr(disp), ptr(le), self.kl_cap))
        return disp, le

    # ---- stateless matchers --------------------------------------------------
    def descriptor_distance(self, a, b):
        a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
        out = np.zeros(a.shape[0], np.int32)
        check(self.L.pli_descriptor_distance(self.h, ptr(a), ptr(b), a.shape[0], ptr(out)))
        return out

    def knn2(self, q, t):
        q, t = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)
        idx = np.zeros((q.shape[0], 2), np.int32)
        dist = np.zeros((q.shape[0], 2), np.int32)
        check(self.L.pli_hamming_knn2(self.h, ptr(q), q.shape[0], ptr(t), t.shape[0], ptr(idx), ptr(dist)))
        return idx, dist

    def match(self, desc1, desc2, nnr):
        """int match(desc1, desc2, nnr, matches_12): returns (nmatches, matches_12)."""
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        m = np.full(d1.shape[0], -1, np.int32)
        n = C.c_int32()
        check(self.L.pli_match_lines(self.h, ptr(d1), d1.shape[0], ptr(d2), d2.shape[0], nnr, ptr(m), C.byref(n)))
        return n.value, m

    def search_by_projection(self, queries, qdesc, cur_kp, cur_desc, cur_uright, bounds, check_orientation=True, occupied=None,
                             with_raw=False):
        """pli_search_by_projection; `occupied`: per current keypoint, holds a map point with observations before the call;
        with_raw: also the matches before the rotation filter (n, best, raw)."""
        q = np.ascontiguousarray(queries, PROJ_QUERY_DT)
        qd = np.ascontiguousarray(qdesc, np.uint8)
        kp = np.ascontiguousarray(cur_kp, KEYPOINT_DT)
        de = np.ascontiguousarray(cur_desc, np.uint8)
        ur = np.ascontiguousarray(cur_uright, np.float32)
        best = np.full(q.shape[0], -1, np.int32)
        raw = np.full(q.shape[0], -1, np.int32) if with_raw else None
        oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        n = C.c_int32()
        check(self.L.pli_search_by_projection(self.h, ptr(q), ptr(qd), q.shape[0], ptr(kp), ptr(de), ptr(ur),
                                              None if oc is None else ptr(oc), kp.shape[0], bounds[0], bounds[1], bounds[2],
                                              bounds[3], int(check_orientation), ptr(best), None if raw is None else ptr(raw),
                                              C.byref(n)))
        return (n.value, best, raw) if with_raw else (n.value, best)

    def vocab_create(self, k, L, parent, is_leaf, desc, weight):
        """DBoW2 vocabulary (node list as in ORBvoc.txt); returns an opaque handle for bow_transform."""
        parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
        h = C.c_void_p()
        check(self.L.pli_vocab_create(self.h, k, L, parent.shape[0], ptr(parent), ptr(is_leaf), ptr(desc), ptr(weight), C.byref(h)))
        return h

    def vocab_destroy(self, h):
        self.L.pli_vocab_destroy(h)

    def bow_transform(self, vocab, desc, levelsup=4):
        """Per-feature descent of DBoW2's transform(): (word_id, weight, node_id) arrays."""
        d = np.ascontiguousarray(desc, np.uint8)
        n = d.shape[0]
        word = np.zeros(n, np.int32); weight = np.zeros(n, np.float64); node = np.zeros(n, np.int32)
        check(self.L.pli_bow_transform(self.h, vocab, ptr(d), n, levelsup, ptr(word), ptr(weight), ptr(node)))
        return word, weight, node

    def orb_extract_lapping(self, eye, image, lapping):
        """ORBextractor::operator() with vLappingArea = lapping (ORBextractor.cc:1135-1144): (n, mono count, keypoints, descriptors);
        the table keeps the mono-first / lapping-from-the-back order for stereo_fisheye()."""
        image = _u8(image)
        kp = np.zeros(self.kp_cap, KEYPOINT_DT)
        desc = np.zeros((self.kp_cap, 32), np.uint8)
        n, mono = C.c_int32(), C.c_int32()
        check(self.L.pli_orb_extract_lapping(self.h, eye, ptr(image), image.shape[1], image.shape[0], image.strides[0],
                                             int(lapping[0]), int(lapping[1]), ptr(kp), self.kp_cap, ptr(desc), C.byref(n),
                                             C.byref(mono)))
        return n.value, mono.value, kp[:n.value].copy(), desc[:n.value].copy()

    def stereo_fisheye(self, cam1, cam2, Rlr, tlr, nleft, nright):
        """Frame::ComputeStereoFishEyeMatches (Frame.cc:1577-1618): (nMatches, mvLeftToRightMatch, mvRightToLeftMatch, mvDepth,
        mvStereo3Dpoints) on the tables of the last orb_extract_lapping of both eyes.  cam: the 8 KannalaBrandt8 parameters."""
        c1, c2 = np.ascontiguousarray(cam1, np.float32), np.ascontiguousarray(cam2, np.float32)
        R, t = np.ascontiguousarray(Rlr, np.float32).reshape(9), np.ascontiguousarray(tlr, np.float32).reshape(3)
        l2r, r2l = np.zeros(self.kp_cap, np.int32), np.zeros(self.kp_cap, np.int32)
        depth, p3d = np.zeros(self.kp_cap, np.float32), np.zeros((self.kp_cap, 3), np.float32)
        nm = C.c_int32()
        check(self.L.pli_stereo_fisheye(self.h, ptr(c1), ptr(c2), ptr(R), ptr(t), ptr(l2r), self.kp_cap, ptr(r2l), self.kp_cap,
                                        ptr(depth), ptr(p3d), C.byref(nm)))
        return nm.value, l2r[:nleft].copy(), r2l[:nright].copy(), depth[:nleft].copy(), p3d[:nleft].copy()

    def stereo_fisheye_tables(self, kpL, descL, mono_left, kpR, descR, mono_right, cam1, cam2, Rlr, tlr):
        """Frame::ComputeStereoFishEyeMatches on caller tables (mvKeys / mDescriptors / monoLeft, ... of the Frame)."""
        kpL, kpR = np.ascontiguousarray(kpL, KEYPOINT_DT), np.ascontiguousarray(kpR, KEYPOINT_DT)
        descL, descR = np.ascontiguousarray(descL, np.uint8), np.ascontiguousarray(descR, np.uint8)
        c1, c2 = np.ascontiguousarray(cam1, np.float32), np.ascontiguousarray(cam2, np.float32)
        R, t = np.ascontiguousarray(Rlr, np.float32).reshape(9), np.ascontiguousarray(tlr, np.float32).reshape(3)
        nl, nr = kpL.shape[0], kpR.shape[0]
        l2r, r2l = np.zeros(max(nl, 1), np.int32), np.zeros(max(nr, 1), np.int32)
        depth, p3d = np.zeros(max(nl, 1), np.float32), np.zeros((max(nl, 1), 3), np.float32)
        nm = C.c_int32()
        check(self.L.pli_stereo_fisheye_tables(self.h, ptr(kpL), ptr(descL), nl, int(mono_left), ptr(kpR), ptr(descR), nr, int(mono_right),
                                               ptr(c1), ptr(c2), ptr(R), ptr(t), ptr(l2r), ptr(r2l), ptr(depth), ptr(p3d), C.byref(nm)))
        return nm.value, l2r[:nl], r2l[:nr], depth[:nl], p3d[:nl]

    def stereo_from_depth(self, depth):
        """Frame::ComputeStereoFromRGBD (Frame.cc:1309): (mvuRight, mvDepth) of the left keypoints from a float depth image."""
        d = np.ascontiguousarray(depth, np.float32)
        assert d.shape == (self.cfg.height, self.cfg.width)
        ur = np.zeros(self.kp_cap, np.float32)
        dp = np.zeros(self.kp_cap, np.float32)
        check(self.L.pli_stereo_from_depth(self.h, ptr(d), d.shape[1], ptr(ur), ptr(dp), self.kp_cap))
        return ur, dp

    def set_rectify_maps(self, eye, mapx, mapy):
        """cv::remap(im, imRect, M1, M2, INTER_LINEAR) of the stereo driver (stereo_euroc.cc:166) fused into the ingest;
        (None, None) removes the maps."""
        if mapx is None:
            check(self.L.pli_set_rectify_maps(self.h, eye, None, None))
            return
        mx = np.ascontiguousarray(mapx, np.float32)
        my = np.ascontiguousarray(mapy, np.float32)
        assert mx.shape == my.shape == (self.cfg.height, self.cfg.width)
        check(self.L.pli_set_rectify_maps(self.h, eye, ptr(mx), ptr(my)))

    def search_local_map(self, queries, qdesc, cur_kp, cur_desc, cur_uright, bounds, nnratio=0.8, cur_occupied=None):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th) ORBmatcher.cc:44 — see pli_search_local_map."""
        q = np.ascontiguousarray(queries, PROJ_QUERY_DT)
        qd = np.ascontiguousarray(qdesc, np.uint8)
        kp = np.ascontiguousarray(cur_kp, KEYPOINT_DT)
        de = np.ascontiguousarray(cur_desc, np.uint8)
        ur = np.ascontiguousarray(cur_uright, np.float32)
        occ = None if cur_occupied is None else np.ascontiguousarray(cur_occupied, np.uint8)
        best = np.full(q.shape[0], -1, np.int32)
        n = C.c_int32()
        check(self.L.pli_search_local_map(self.h, ptr(q), ptr(qd), q.shape[0], ptr(kp), ptr(de), ptr(ur), ptr(occ),
                                          kp.shape[0], bounds[0], bounds[1], bounds[2], bounds[3], nnratio, ptr(best),
                                          C.byref(n)))
        return n.value, best

    def search_local_map_fisheye(self, q_left, q_right, qdesc, kp_left, desc_left, left_to_right, kp_right, desc_right,
                                 right_to_left, bounds, nnratio=0.8, occ_left=None, occ_right=None):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th) for two fisheye cameras, ORBmatcher.cc:44-214 — see
        pli_search_local_map_fisheye.  Returns (nmatches, mp_left, mp_right)."""
        ql = np.ascontiguousarray(q_left, PROJ_QUERY_DT)
        qr = np.ascontiguousarray(q_right, PROJ_QUERY_DT)
        qd = np.ascontiguousarray(qdesc, np.uint8)
        kl, kr = np.ascontiguousarray(kp_left, KEYPOINT_DT), np.ascontiguousarray(kp_right, KEYPOINT_DT)
        dl, dr = np.ascontiguousarray(desc_left, np.uint8), np.ascontiguousarray(desc_right, np.uint8)
        l2r, r2l = np.ascontiguousarray(left_to_right, np.int32), np.ascontiguousarray(right_to_left, np.int32)
        ol = None if occ_left is None else np.ascontiguousarray(occ_left, np.uint8)
        orr = None if occ_right is None else np.ascontiguousarray(occ_right, np.uint8)
        mpl = np.full(kl.shape[0], -1, np.int32)
        mpr = np.full(kr.shape[0], -1, np.int32)
        n = C.c_int32()
        check(self.L.pli_search_local_map_fisheye(self.h, ptr(ql), ptr(qr), ptr(qd), ql.shape[0], ptr(kl), ptr(dl), ptr(ol), ptr(l2r),
                                                  kl.shape[0], ptr(kr), ptr(dr), ptr(orr), ptr(r2l), kr.shape[0], bounds[0], bounds[1],
                                                  bounds[2], bounds[3], nnratio, ptr(mpl), ptr(mpr), C.byref(n)))
        return n.value, mpl, mpr

    def match_nnr(self, desc1, desc2, nnr):
        """match(vpLocalMapLines, CurrentFrame, nnr, matches_12) LineMatcher.cpp:161 (one-way matchNNR)."""
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        m = np.full(d1.shape[0], -1, np.int32)
        n = C.c_int32()
        check(self.L.pli_match_nnr(self.h, ptr(d1), d1.shape[0], ptr(d2), d2.shape[0], nnr, ptr(m), C.byref(n)))
        return n.value, m

    # ---- batch path ----------------------------------------------------------
    def table_bytes(self, nframes):
        return int(self.layout.record_bytes) * nframes

    def batch_run_host(self, images, stages=capi.RUN_ALL):
        """images: (nframes, 2, H, W) u8.  Returns the list of parsed frame records."""
        images = np.ascontiguousarray(images, np.uint8)
        nf, two, H, W = images.shape
        assert two == 2
        left = np.ascontiguousarray(images[:, 0])
        right = np.ascontiguousarray(images[:, 1])
        table = np.zeros(self.table_bytes(nf), np.uint8)
        check(self.L.pli_batch_run_host(self.h, nf, ptr(left), ptr(right), W, W * H, stages, ptr(table)))
        return [self.parse_record(table, f) for f in range(nf)]

    def batch_run_mono_host(self, images, stages=capi.RUN_ORB | capi.RUN_LINES):
        """A monocular (or RGB-D colour) stream through the batch entry point: the frames 2i and 2i+1 take the two eye
        slots of record i (left pointer = frame 0, right pointer = frame 1, frame stride = two frames), the stereo stages
        stay off.  images: (nframes, H, W) u8, nframes even.  Returns one dict per frame: kp, desc, kl, ldesc — what the
        monocular Frame constructor (Frame.cc:334) extracts."""
        images = np.ascontiguousarray(images, np.uint8)
        nf, H, W = images.shape
        assert nf % 2 == 0 and not (stages & (capi.RUN_STEREO_POINTS | capi.RUN_STEREO_LINES))
        table = np.zeros(self.table_bytes(nf // 2), np.uint8)
        base = images.ctypes.data
        check(self.L.pli_batch_run_host(self.h, nf // 2, C.c_void_p(base), C.c_void_p(base + W * H), W, 2 * W * H, stages,
                                        ptr(table)))
        out = []
        for f in range(nf):
            r = self.parse_record(table, f // 2)
            e = "LR"[f & 1]
            out.append({"kp": r["kp" + e], "desc": r["desc" + e], "kl": r["kl" + e], "ldesc": r["ldesc" + e]})
        return out

    def batch_run_device(self, nframes, dev_left, dev_right, stride, frame_stride, dev_table, stages=capi.RUN_ALL):
        """Asynchronous: device pointers (ints) in, device table out; sync() to wait."""
        check(self.L.pli_batch_run(self.h, nframes, C.c_void_p(dev_left), C.c_void_p(dev_right), stride, frame_stride,
                                   stages, C.c_void_p(dev_table)))

    def frame_extract(self, left, right):
        """pli_frame_extract: the whole front-end of one Frame (both eyes, points and lines, the two stereo matchers) in one
        submission; returns the parsed record and leaves the per-call state behind (pyramid_level, compute_stereo_matches)."""
        left, right = _u8(left), _u8(right)
        rec = np.zeros(int(self.layout.record_bytes), np.uint8)
        check(self.L.pli_frame_extract(self.h, ptr(left), ptr(right), left.shape[1], left.shape[0], left.strides[0],
                                       right.strides[0], ptr(rec)))
        return self.parse_record(rec, 0)

    def lsd_round_stats(self):
        """(rounds launched without a look by the last call, rounds its slowest image needed or -1, images that took the
        device-side fallback so far, rounds the next call plans from) — pli_lsd_round_stats; synchronises."""
        out = (C.c_int32 * 4)()
        check(self.L.pli_lsd_round_stats(self.h, out))
        return tuple(int(v) for v in out)

    def lsd_arena_words(self):
        """Arena words per scaled LSD pixel this context got (16 unless it is large or the device was short of memory) — pli_lsd_arena_words."""
        out = C.c_int32()
        check(self.L.pli_lsd_arena_words(self.h, C.byref(out)))
        return out.value

    def selftest_hot_trig(self):
        """Largest |v_cos / v_sin - cos / sin| over every float angle in [0, 360] degrees on this device (pli_selftest_hot_trig)."""
        out = C.c_double()
        check(self.L.pli_selftest_hot_trig(self.h, C.byref(out)))
        return out.value

    def sync(self):
        check(self.L.pli_ctx_sync(self.h))

    # ---- frame-to-frame track matching of a batch (BASELINE config 3) ------------------------------------------
    def track_layout(self):
        tl = capi.TrackLayout()
        check(self.L.pli_track_layout_get(self.h, C.byref(tl)))
        return tl

    def track_params(self, th=15.0, mono=False, check_orientation=True, nnr_lines=0.9, cx=None, cy=None, fy=None):
        """pli_track_params with the context's camera (fx, bf) and the image rectangle as the frame grid bounds."""
        W, H = self.cfg.width, self.cfg.height
        return capi.TrackParams(fx=self.cfg.fx, fy=self.cfg.fx if fy is None else fy, cx=W / 2.0 if cx is None else cx,
                                cy=H / 2.0 if cy is None else cy, bf=self.cfg.bf, th=th, min_x=0.0, max_x=float(W), min_y=0.0,
                                max_y=float(H), mono=int(mono), check_orientation=int(check_orientation), nnr_lines=nnr_lines,
                                reserved=0)

    def batch_track_device(self, nframes, dev_table, dev_poses, params, dev_track):
        """Asynchronous: frame i against frame i-1 on the device tables (pointers as ints)."""
        check(self.L.pli_batch_track(self.h, nframes, C.c_void_p(dev_table), C.c_void_p(dev_poses), C.byref(params),
                                     C.c_void_p(dev_track)))

    def parse_track(self, track, frame):
        tl = self.track_layout()
        rec = track[frame * tl.record_bytes:(frame + 1) * tl.record_bytes]
        counts = rec[tl.off_counts:tl.off_counts + 16].view(np.int32).copy()
        return {"counts": counts, "best": rec[tl.off_best:tl.off_best + 4 * int(counts[0])].view(np.int32).copy(),
                "lines": rec[tl.off_lines:tl.off_lines + 4 * int(counts[2])].view(np.int32).copy()}

    # ---- pipelined host entry point (pinned buffers, copies overlap the kernels) ---------------------------
    def pinned(self, nbytes):
        """numpy u8 view of `nbytes` of pinned host memory (pli_host_alloc); freed with the Frontend."""
        p = C.c_void_p()
        check(self.L.pli_host_alloc(nbytes, C.byref(p)))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def host_buffers(self, nframes):
        """(left, right, tables): pinned (nframes, H*W) image planes and two pinned result tables."""
        n = self.cfg.width * self.cfg.height
        left = self.pinned(nframes * n).reshape(nframes, n)
        right = self.pinned(nframes * n).reshape(nframes, n)
        tables = [self.pinned(self.table_bytes(nframes)) for _ in range(2)]
        return left, right, tables

    def host_submit(self, nframes, left, right, table, stages=capi.RUN_ALL):
        W, H = self.cfg.width, self.cfg.height
        check(self.L.pli_batch_submit_host(self.h, nframes, ptr(left), ptr(right), W, W * H, stages, ptr(table)))

    def host_wait(self):
        check(self.L.pli_batch_wait(self.h, 0))

    def host_wait_all(self):
        check(self.L.pli_batch_wait(self.h, 1))

    def set_stream(self, hip_stream):
        check(self.L.pli_ctx_set_stream(self.h, C.c_void_p(hip_stream)))

    def parse_record(self, table, frame):
        """Slice one frame record (numpy u8 array of the whole table) into named arrays."""
        Y = self.layout
        rec = table[frame * Y.record_bytes:(frame + 1) * Y.record_bytes]
        counts = rec[Y.off_counts:Y.off_counts + 32].view(np.int32)
        out = {"counts": counts.copy(), "truncated": rec[Y.off_counts + 24:Y.off_counts + 28].copy()}   # lines L/R, keypoints L/R
        for e, name in ((0, "L"), (1, "R")):
            n = int(counts[e])
            out["kp" + name] = rec[Y.off_kp[e]:Y.off_kp[e] + 24 * n].view(KEYPOINT_DT).copy()
            out["desc" + name] = rec[Y.off_desc[e]:Y.off_desc[e] + 32 * n].reshape(n, 32).copy()
            m = int(counts[2 + e])
            out["kl" + name] = rec[Y.off_kl[e]:Y.off_kl[e] + 68 * m].view(KEYLINE_DT).copy()
            out["ldesc" + name] = rec[Y.off_ldesc[e]:Y.off_ldesc[e] + 32 * m].reshape(m, 32).copy()
        n, m = int(counts[0]), int(counts[2])
        out["uright"] = rec[Y.off_uright:Y.off_uright + 4 * n].view(np.float32).copy()
        out["depth"] = rec[Y.off_depth:Y.off_depth + 4 * n].view(np.float32).copy()
        out["disp"] = rec[Y.off_disp:Y.off_disp + 8 * m].view(np.float32).reshape(m, 2).copy()
        out["le"] = rec[Y.off_le:Y.off_le + 24 * m].view(np.float64).reshape(m, 3).copy()
        return out

    # ---- measurement / debug ---------------------------------------------------
    def prof_enable(self, on=True):
        check(self.L.pli_prof_enable(self.h, int(on)))

    def prof_reset(self):
        check(self.L.pli_prof_reset(self.h))

    def prof_report(self):
        """{kernel name: (calls, total_ms)} measured with HIP events on the context stream."""
        buf = C.create_string_buffer(1 << 16)
        check(self.L.pli_prof_report(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.split()
            out[name] = (int(calls), float(ms))
        return out

    def debug_enable(self, on=True):
        check(self.L.pli_debug_enable(self.h, int(on)))

    def debug_fetch(self, image, what, arg=0):
        n = C.c_int64()
        check(self.L.pli_debug_fetch(self.h, image, what, arg, None, 0, C.byref(n)))
        buf = np.zeros(max(int(n.value), 1), np.uint8)
        check(self.L.pli_debug_fetch(self.h, image, what, arg, ptr(buf), buf.size, C.byref(n)))
        return buf[:n.value]

    def debug_points(self, image, what, level):
        raw = self.debug_fetch(image, what, level).view(np.int32)
        return raw[1:1 + 3 * raw[0]].reshape(-1, 3).copy()


class ORBextractor:
    """Mirror of ORB_SLAM3::ORBextractor (include/ORBextractor.h:46-115)."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, image_size, eye=0, frontend=None):
        w, h = image_size
        self.eye = eye
        if frontend is None:
            cfg = capi.default_config(w, h, orb_nfeatures=nfeatures, orb_scale_factor=scaleFactor,
                                      orb_nlevels=nlevels, orb_ini_th_fast=iniThFAST, orb_min_th_fast=minThFAST)
            frontend = Frontend(cfg)
        self.fe = frontend
        self.nlevels = nlevels
        self.scaleFactor = np.float32(scaleFactor)

    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """Returns (monoCount, keypoints, descriptors); -1 for an empty image like the reference."""
        n, kp, desc = self.fe.orb_extract(self.eye, image)
        return n, kp, desc

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return float(self.scaleFactor)

    def GetScaleFactors(self):
        s = [np.float32(1.0)]
        for _ in range(1, self.nlevels):
            s.append(np.float32(s[-1] * self.scaleFactor))
        return np.array(s, np.float32)

    def GetInverseScaleFactors(self):
        return (np.float32(1.0) / self.GetScaleFactors()).astype(np.float32)

    @property
    def mvImagePyramid(self):
        return [self.fe.pyramid_level(self.eye, l) for l in range(self.nlevels)]


class Lineextractor:
    """Mirror of ORB_SLAM3::Lineextractor (include/LineExtractor.h:41-74)."""

    def __init__(self, lsd_nfeatures, min_line_length, lsd_refine, lsd_scale, lsd_sigma_scale, lsd_quant, lsd_ang_th,
                 lsd_log_eps, lsd_density_th, lsd_n_bins, image_size, eye=0, frontend=None):
        w, h = image_size
        self.eye = eye
        if frontend is None:
            cfg = capi.default_config(w, h, lsd_nfeatures=lsd_nfeatures, min_line_length=min_line_length,
                                      lsd_refine=lsd_refine, lsd_scale=lsd_scale, lsd_sigma_scale=lsd_sigma_scale,
                                      lsd_quant=lsd_quant, lsd_ang_th=lsd_ang_th, lsd_log_eps=lsd_log_eps,
                                      lsd_density_th=lsd_density_th, lsd_n_bins=lsd_n_bins)
            frontend = Frontend(cfg)
        self.fe = frontend

    def __call__(self, image, mask=None):
        """Returns (keylines, descriptors_line)."""
        n, kl, desc = self.fe.line_extract(self.eye, image)
        return kl, desc
