"""The C++ drop-in, executed (-m gpu): tests/cpp/dropin_harness.cpp drives pli_slam_amd/adapters/ the way the reference
drives the classes they replace — extractors built as Tracking.cc:743-746 / :87-94, every Frame on four std::threads as
Frame.cc:128-135, ComputeStereoMatches_Lines / ComputeStereoMatches through adapters/frame_stereo.hpp, then
match(...) (LineMatcher.h:63) and ORBmatcher::SearchByProjection(cur, last, th, bMono, match12) (ORBmatcher.cc:2179) on
consecutive frames — and dumps every container.  Compared here, byte for byte:

  * with the oracle (every frame, every container),
  * with the ctypes path through the same library (pli_batch_run_host),
  * the four-thread run with the run of four sequential calls,
  * 20 repetitions of the whole 50-frame sequence with each other (hash of all containers).
"""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pli_slam_amd", "csrc", "libpli_frontend.so")
W, H = 752, 480
NFEATURES, NLINES = 1200, 100
FX, FY, CX, CY, BF = 435.2046959714599, 435.2046959714599, 367.4517211914062, 252.2008514404297, 47.90639384423901


def build_harness(outdir, sanitize=None):
    exe = os.path.join(outdir, "dropin_harness" + ("_" + sanitize if sanitize else ""))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-pthread", "-I", ROOT, "-I", os.path.join(ROOT, "tests", "stubs"),
           os.path.join(ROOT, "tests", "cpp", "dropin_harness.cpp"), LIB, "-Wl,-rpath," + os.path.dirname(LIB),
           "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    if sanitize:
        cmd.insert(1, "-fsanitize=" + sanitize)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


def write_input(path, frames, reps, mode, nfeatures=NFEATURES, nlines=NLINES):
    h, w = frames[0][0].shape
    with open(path, "wb") as f:
        f.write(b"PLIH" + struct.pack("<7i", w, h, len(frames), reps, mode, nfeatures, nlines))
        for L, R in frames:
            f.write(np.ascontiguousarray(L, np.uint8).tobytes())
            f.write(np.ascontiguousarray(R, np.uint8).tobytes())


def read_dump(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"PLID"
    out, o = {}, 4
    dts = {ord("B"): np.uint8, ord("i"): np.int32, ord("f"): np.float32, ord("d"): np.float64, ord("Q"): np.uint64}
    while o < len(raw):
        nlen, dt, rows, cols = struct.unpack_from("<4i", raw, o)
        o += 16
        name = raw[o:o + nlen].decode()
        o += nlen
        dtype = np.dtype(dts[dt])
        n = rows * cols if rows > 0 and cols > 0 else 0
        out[name] = np.frombuffer(raw, dtype, n, o).reshape(max(rows, 0), cols)
        o += n * dtype.itemsize
    return out


def kp_arrays(kp):
    """oracle / ctypes keypoint table -> the harness's (float rows, int rows) of cv::KeyPoint"""
    f = np.stack([kp["x"], kp["y"], kp["size"], kp["angle"], kp["response"]], axis=1).astype(np.float32) if len(kp) else np.zeros((0, 5), np.float32)
    i = np.stack([kp["octave"], np.full(len(kp), -1, np.int32)], axis=1).astype(np.int32) if len(kp) else np.zeros((0, 2), np.int32)
    return f, i


KL_F = ("angle", "pt_x", "pt_y", "response", "size", "startPointX", "startPointY", "endPointX", "endPointY", "sPointInOctaveX",
        "sPointInOctaveY", "ePointInOctaveX", "ePointInOctaveY", "lineLength")
KL_I = ("class_id", "octave", "numOfPixels")


def kl_arrays(kl):
    f = np.stack([kl[n] for n in KL_F], axis=1).astype(np.float32) if len(kl) else np.zeros((0, 14), np.float32)
    i = np.stack([kl[n] for n in KL_I], axis=1).astype(np.int32) if len(kl) else np.zeros((0, 3), np.int32)
    return f, i


def same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, "%s: %s %s vs %s %s" % (what, a.shape, a.dtype, b.shape, b.dtype)
    assert a.tobytes() == b.tobytes(), "%s differs" % what


def frame_containers(kpL, dL, kpR, dR, ur, dp, klL, ldL, klR, ldR, disp, le):
    c = {}
    c["mvKeys.f"], c["mvKeys.i"] = kp_arrays(kpL)
    c["mvKeysRight.f"], c["mvKeysRight.i"] = kp_arrays(kpR)
    c["mDescriptors"], c["mDescriptorsRight"] = dL, dR
    c["mvuRight"], c["mvDepth"] = ur.reshape(-1, 1), dp.reshape(-1, 1)
    c["mvKeys_Line.f"], c["mvKeys_Line.i"] = kl_arrays(klL)
    c["mvKeysRight_Line.f"], c["mvKeysRight_Line.i"] = kl_arrays(klR)
    c["mDescriptors_Line"], c["mDescriptorsRight_Line"] = ldL, ldR
    c["mvDisparity_l"], c["mvle_l"] = disp.reshape(-1, 2), le.reshape(-1, 3)
    return c


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from pli_slam_amd import synth
    d = str(tmp_path_factory.mktemp("dropin"))
    exe = build_harness(d)
    frames = [synth.make_stereo_pair(40 + s, W, H, t=t) for s in range(10) for t in range(5)]      # 50 frames, 10 scenes x 5 instants
    res = {"frames": frames}
    for name, mode, reps in (("threads", 1, 20), ("sequential", 0, 1)):
        inp, outp = os.path.join(d, name + ".in"), os.path.join(d, name + ".out")
        write_input(inp, frames, reps, mode)
        r = subprocess.run([exe, inp, outp], capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        res[name] = read_dump(outp)
        os.unlink(inp)
    return res


@pytest.mark.gpu
def test_four_threads_equal_sequential_calls_and_repeat_identically(runs):
    t, s = runs["threads"], runs["sequential"]
    assert len(t["hashes"]) == 20 and len(set(t["hashes"].ravel().tolist())) == 1, "the 20 four-thread repetitions differ: %s" % t["hashes"].ravel()
    assert t["hashes"][0, 0] == s["hashes"][0, 0]
    assert set(t) == set(s)
    for k in t:
        if k not in ("hashes", "frame_ms"):
            same(t[k], s[k], k)
    # the four threads of a Frame are fused into one submission (FrameFusion); four calls in a row take the per-call path
    print("Frame constructor: %.2f ms on four threads (fused), %.2f ms as four calls" % (t["frame_ms"][0, 0], s["frame_ms"][0, 0]))
    assert t["frame_ms"][0, 0] < s["frame_ms"][0, 0]
    assert int(t["groups_left"][0, 0]) == 0, "extractor destruction left device contexts in the registry"
    n = [len(t["f%d/mvKeys.f" % i]) for i in range(50)]
    assert min(n) > 600 and min(len(t["f%d/mvKeys_Line.f" % i]) for i in range(50)) > 20


@pytest.mark.gpu
def test_every_container_equals_the_oracle_and_the_ctypes_path(runs):
    from oracle import pyoracle as po
    from pli_slam_amd import capi
    from pli_slam_amd.frontend import Frontend
    t, frames = runs["threads"], runs["frames"]
    cfg = capi.default_config(W, H, orb_nfeatures=NFEATURES, lsd_nfeatures=NLINES, max_frames=2)
    fe = Frontend(cfg)
    ocfg = po.Config.from_buffer_copy(bytes(cfg))
    for i, (L, R) in enumerate(frames):
        fr = po.Frame(ocfg)
        nL, kpL, dL = fr.orb_extract(0, L)
        nR, kpR, dR = fr.orb_extract(1, R)
        mL, klL, ldL = fr.line_extract(0, L)
        mR, klR, ldR = fr.line_extract(1, R)
        ur, dp, _, _ = fr.stereo_points()
        disp, le, _ = fr.stereo_lines()
        want = frame_containers(kpL, dL, kpR, dR, ur, dp, klL, ldL, klR, ldR, disp, le)
        for k, v in want.items():
            same(t["f%d/%s" % (i, k)], v, "frame %d %s vs the oracle" % (i, k))
        assert tuple(t["f%d/mono" % i][0]) == (nL, nR)          # operator() returns the mono count = N on the rectified path
        for lvl in (0, 7):
            same(t["f%d/pyrL%d" % (i, lvl)], fr.pyramid(0, lvl), "frame %d mvImagePyramid[%d]" % (i, lvl))
        if i % 5 == 0:      # the ctypes path through the same library (one call per frame)
            rec = fe.batch_run_host(np.stack([L, R])[None])[0]
            got = frame_containers(rec["kpL"], rec["descL"], rec["kpR"], rec["descR"], rec["uright"], rec["depth"], rec["klL"],
                                   rec["ldescL"], rec["klR"], rec["ldescR"], rec["disp"], rec["le"])
            for k, v in got.items():
                same(t["f%d/%s" % (i, k)], v, "frame %d %s vs the ctypes path" % (i, k))


@pytest.mark.gpu
def test_frame_to_frame_matchers_through_the_adapters(runs):
    from oracle import pyoracle as po
    from pli_slam_amd import capi
    t = runs["threads"]
    sf = np.ones(8, np.float32)
    for l in range(1, 8):
        sf[l] = sf[l - 1] * np.float32(1.2)
    total = 0
    for i in range(1, 50):
        pre, last = "f%d/" % i, "f%d/" % (i - 1)
        # match(last.mDescriptors_Line, cur.mDescriptors_Line, 0.9, matches_12): LineMatcher.cpp:201-229
        on, om12 = po.match_lines(t[last + "mDescriptors_Line"], t[pre + "mDescriptors_Line"], 0.9, True)
        assert int(t[pre + "line_nmatches"][0, 0]) == on
        same(t[pre + "line_matches_12"].ravel(), om12, "frame %d line matches_12" % i)
        # SearchByProjection(cur, last, th, bMono, match12): the adapter projects with cv::Mat expressions, the oracle restates them
        lf, li = t[last + "mvKeys.f"], t[last + "mvKeys.i"]
        lkp = np.zeros(len(lf), capi.KEYPOINT_DT)
        for j, n in enumerate(("x", "y", "size", "angle", "response")):
            lkp[n] = lf[:, j]
        lkp["octave"] = li[:, 0]
        cf, ci = t[pre + "mvKeys.f"], t[pre + "mvKeys.i"]
        ckp = np.zeros(len(cf), capi.KEYPOINT_DT)
        for j, n in enumerate(("x", "y", "size", "angle", "response")):
            ckp[n] = cf[:, j]
        ckp["octave"] = ci[:, 0]
        Tlw = t[pre + "Tlw"]
        for c in range(4):
            mono = c == 3
            cs = pre + "sbp%d/" % c
            q = po.track_queries(lkp, t[last + "mvDepth"].ravel(), Tlw, t[cs + "Tcw"], FX, FY, CX, CY, BF, 15.0 if mono else 7.0, mono, sf)
            if c == 0:
                assert (q["max_level"][q["valid"] == 1] == -1).all(), "case 0 is meant to be the forward branch (ORBmatcher.cc:2196)"
            if c == 1:
                assert (q["min_level"][q["valid"] == 1] == 0).all(), "case 1 is meant to be the backward branch"
            on, obest = po.search_by_projection(q, t[last + "mDescriptors"], ckp, t[pre + "mDescriptors"], t[pre + "mvuRight"].ravel(),
                                                (0.0, float(W), 0.0, float(H)), c != 2)
            assert int(t[cs + "nmatches"][0, 0]) == on, "frame %d case %d: nmatches %d vs oracle %d" % (i, c, int(t[cs + "nmatches"][0, 0]), on)
            want = sorted((int(b), int(k)) for k, b in enumerate(obest) if b >= 0)      # match12[bestIdx2] = i, a std::map: sorted by key
            got = [tuple(r) for r in t[cs + "match12"].tolist()]
            assert got == want, "frame %d case %d: match12 differs" % (i, c)
            if i % 5:      # consecutive instants of one scene (i % 5 == 0 pairs two unrelated scenes)
                total += on
    assert total > 40 * 4 * 30, "the projection searches found next to nothing (%d): the test geometry is off" % total


@pytest.mark.gpu
def test_cpp_host_layer_on_the_gpu():
    from test_cpp_host import test_cpp_host_layer_builds_and_links
    test_cpp_host_layer_builds_and_links()
