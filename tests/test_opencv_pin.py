"""The oracle's OpenCV primitives against a REAL OpenCV — runs only when somebody with the toolchain has produced
tests/golden/opencv_pin/ (tools/pin/run_pin.sh: builds tools/pin/pin_against_opencv.cpp against the installed OpenCV and stores what
the real cv:: functions return on seeded inputs).  Without that directory parity stays "unpinned" for the OpenCV-backed part of the
path (DESIGN.md "Oracle") and this test is skipped; the compare logic itself is exercised on a self-made directory."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "pin", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_against_real_opencv_outputs():
    d = os.path.join(ROOT, "tests", "golden", "opencv_pin")
    if not os.path.exists(os.path.join(d, "manifest.txt")):
        pytest.skip("tests/golden/opencv_pin absent: run tools/pin/run_pin.sh on a machine with OpenCV 3.3.1")
    res = _load("pin_compare").compare(d)
    bad = [k for k, v in res.items() if v is False or (isinstance(v, list) and v and isinstance(v[0], tuple) and not any(h[1] == "exact" for h in v))]
    assert not bad, bad


def test_pin_compare_on_a_self_made_directory(tmp_path, oracle):
    """The comparison itself: a directory filled from the ORACLE's own outputs (standing in for OpenCV's) must come out pinned,
    with the LSD flag set that produced it marked exact — and a corrupted primitive must be reported."""
    dump = _load("dump_inputs")
    cmp_ = _load("pin_compare")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "inputs"))
    name, img = next(x for x in dump.inputs() if x[0] == "small_s3_left")
    img.tofile(os.path.join(d, "inputs", name + ".raw"))
    h, w = img.shape
    open(os.path.join(d, "manifest.txt"), "w").write("opencv self\nx\ny\nimage %s %d %d\n" % (name, w, h))
    xs = np.linspace(0.001, 6.28, 50, dtype=np.float32)
    np.save(os.path.join(d, "libm_x.npy"), xs)
    np.save(os.path.join(d, "libm_cosf.npy"), np.array([oracle.glibc_cosf(float(x)) for x in xs], np.float32))
    np.save(os.path.join(d, "libm_sinf.npy"), np.array([oracle.glibc_sinf(float(x)) for x in xs], np.float32))
    yx = np.array([[1.5, -2.0], [-3.0, 0.5], [0.0, 1.0]], np.float32)
    np.save(os.path.join(d, "fastatan2_yx.npy"), yx)
    np.save(os.path.join(d, "fastatan2.npy"), np.array([oracle.fast_atan2(float(y), float(x)) for y, x in yx], np.float32))
    sv = lambda k, a: np.save(os.path.join(d, "%s_%s.npy" % (name, k)), a)
    dw, dh = int(np.rint(np.float32(w) / np.float32(1.2))), int(np.rint(np.float32(h) / np.float32(1.2)))
    sv("resize_level1", oracle.resize(img, dw, dh, w / dw, h / dh))
    sv("blur7_s2", oracle.gaussian_blur(img, 7, 2.0))
    b5 = oracle.gaussian_blur(img, 5, 1.0)
    sv("blur5_s1", b5)
    dx, dy = oracle.sobel(b5)
    sv("sobel_dx", dx); sv("sobel_dy", dy)
    for th in (20, 7):
        sv("fast_t%d" % th, oracle.fast_image(img, th))
    fr = oracle.Frame(oracle.default_config(w, h, lsd_nfeatures=0, parity_flags=oracle.PARITY_LSD_F64))
    fr.line_extract(0, img)
    sv("lsd_segments", fr.lsd_segments(0))
    res = cmp_.compare(d)
    assert all(v is not False for v in res.values()), res
    lsd = [v for k, v in res.items() if "LSD per parity" in k][0]
    assert dict(lsd)[oracle.PARITY_LSD_F64] == "exact" and dict(lsd)[0] != "exact"
    bad = oracle.gaussian_blur(img, 7, 2.0)
    bad[5, 5] ^= 1
    sv("blur7_s2", bad)
    assert cmp_.compare(d)[name + " GaussianBlur 7x7 s2"] is False
