"""The header-only C++ host layer (pli_slam_amd/adapters/pli_cpp.hpp) compiles against the public
header and links the C-ABI library; without a GPU construction fails with PLI_ERR_NO_DEVICE."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "pli_slam_amd/adapters/pli_cpp.hpp"
#include <cstdio>
int main() {
  pli_frontend_config c;
  pli_config_default(&c, 752, 480);
  if (pli_kp_capacity(&c) < 1224) return 2;
  try {
    pli::Frontend fe(c);
    std::vector<pli_keypoint> k; std::vector<uint8_t> d;
    std::vector<uint8_t> img(752 * 480, 128);
    int n = fe.extractORB(0, img.data(), 752, 480, 752, k, d);
    std::printf("gpu present: %d keypoints on a flat image\n", n);
    return n == 0 ? 0 : 3;
  } catch (const pli::Error& e) {
    std::printf("no gpu: %s\n", e.what());
    return e.status == PLI_ERR_NO_DEVICE ? 0 : 4;
  }
}
'''


def test_cpp_host_layer_builds_and_links():
    lib = os.path.join(ROOT, "pli_slam_amd", "csrc", "libpli_frontend.so")
    assert os.path.exists(lib), "build the library first (__graft_entry__.build())"
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++17", "-I", ROOT, src, lib, "-Wl,-rpath," + os.path.dirname(lib),
                               "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


ADAPTER_SRC = r'''
#define PLI_ADAPTER_NO_KEYLINE_HEADER
#define PLI_ADAPTER_KEYLINE_TYPE StubKeyLine
#include <opencv2/core/core.hpp>
struct StubKeyLine {      // cv::line_descriptor::KeyLine, field names as in descriptor_custom.hpp:105-144
  float angle; int class_id; int octave; cv::Point2f pt; float response; float size;
  float startPointX, startPointY, endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength; int numOfPixels;
};
#include "pli_slam_amd/adapters/orbslam_adapters.hpp"
#include <map>
// the members of Frame / MapPoint that SearchByProjection(CurrentFrame, LastFrame, ...) reads (include/Frame.h, MapPoint.h)
struct StubMapPoint { cv::Mat GetWorldPos(); cv::Mat GetDescriptor(); int Observations(); };
struct StubFrame {
  cv::Mat mTcw, mDescriptors; float mb, mbf, fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY; int N;
  std::vector<StubMapPoint*> mvpMapPoints; std::vector<bool> mvbOutlier; std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  std::vector<float> mvScaleFactors, mvuRight;
};
int use(StubFrame& cur, const StubFrame& last, cv::Mat& im, cv::Mat& mask, std::vector<cv::KeyPoint>& kps, cv::Mat& desc,
        std::vector<StubKeyLine>& kl, cv::Mat& ldesc) {
  // the constructor lists of include/ORBextractor.h:53-54 and include/LineExtractor.h:44-46, as Tracking.cc:87-98,743-749 calls them
  ORB_SLAM3::ORBextractor* l = new ORB_SLAM3::ORBextractor(1200, 1.2f, 8, 20, 7);
  ORB_SLAM3::ORBextractor* r = new ORB_SLAM3::ORBextractor(1200, 1.2f, 8, 20, 7);
  ORB_SLAM3::Lineextractor* ll = new ORB_SLAM3::Lineextractor(500, 0.025, 0, 1.2, 0.6, 2.0, 22.5, 1.0, 0.6, 1024, false);
  ORB_SLAM3::Lineextractor* l2 = new ORB_SLAM3::Lineextractor(500, 0.025);
  std::vector<int> lap = {0, 0};
  int n = (*l)(im, mask, kps, desc, lap) + (*r)(im, mask, kps, desc, lap);
  (*ll)(im, mask, kl, ldesc);
  (*l2)(im, mask, kl, ldesc);
  std::vector<int> m12;
  n += ORB_SLAM3::match(ldesc, ldesc, 0.9f, m12);
  typedef ORB_SLAM3::PliORBmatcher<StubFrame, StubMapPoint> ORBmatcher;
  ORBmatcher matcher(0.9f, true);
  std::map<int, int> match12;
  n += matcher.SearchByProjection(cur, last, 15.f, false, match12) + matcher.SearchByProjection(cur, last, 7.f, true);
  n += ORBmatcher::DescriptorDistance(desc, desc) + l->GetLevels() + (int)l->GetScaleFactors().size() + (int)l->mvImagePyramid.size();
  return n;
}
'''


def test_orbslam_adapter_header_is_valid_cpp_against_a_stub_cv():
    """Syntax check (g++ -fsyntax-only against tests/stubs/opencv2, a functional stand-in for the cv:: types, no OpenCV): the
    adapter offers the reference's constructor lists, functors, match(), DescriptorDistance and SearchByProjection signatures.
    The adapters are EXECUTED by tests/test_cpp_dropin.py (-m gpu) and, without a device, by the test below."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "a.cpp")
        open(src, "w").write(ADAPTER_SRC)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", ROOT, "-I", os.path.join(ROOT, "tests", "stubs"), src],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-4000:]


def test_host_descriptor_distance():
    src = r'''
#include "pli_slam_amd/adapters/pli_cpp.hpp"
int main() {
  uint8_t a[32] = {0}, b[32];
  for (int i = 0; i < 32; ++i) b[i] = 0xFF;
  if (pli::descriptorDistance(a, a) != 0 || pli::descriptorDistance(a, b) != 256) return 1;
  b[5] = 0xFE; a[31] = 0x80;
  return pli::descriptorDistance(a, b) == 254 ? 0 : 2;
}
'''
    lib = os.path.join(ROOT, "pli_slam_amd", "csrc", "libpli_frontend.so")
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.cpp")
        open(p, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++17", "-I", ROOT, p, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
        assert subprocess.run([exe]).returncode == 0


def test_dropin_harness_builds_and_fails_loudly_without_a_device(tmp_path):
    """tests/cpp/dropin_harness.cpp (the -m gpu drop-in test's program) compiles against the adapters and links the library here
    too; without a GPU the first operator() call throws pli::Error (PLI_ERR_NO_DEVICE) out of the extractor thread's caller —
    no CPU fallback, no crash.  On the GPU box the same program runs one small frame."""
    import numpy as np
    from test_cpp_dropin import build_harness, write_input, read_dump
    exe = build_harness(str(tmp_path))
    rng = np.random.default_rng(0)
    img = (rng.random((240, 376)) * 255).astype(np.uint8)
    inp, outp = str(tmp_path / "in"), str(tmp_path / "out")
    write_input(inp, [(img, img)], 1, 0, nfeatures=500, nlines=60)       # sequential calls: the exception surfaces in main
    r = subprocess.run([exe, inp, outp], capture_output=True, text=True)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr
        assert int(read_dump(outp)["groups_left"][0, 0]) == 0
    else:
        assert r.returncode == 1 and "no HIP device" in r.stderr, (r.returncode, r.stderr)


import pytest


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_adapters_under_sanitizers_with_a_mock_library(tmp_path, sanitizer):
    """SURVEY §5 host hygiene: the drop-in harness (adapters, registry, Frame-level matchers, four std::threads per Frame) built with
    -fsanitize=thread / address against tests/cpp/mock_pli.cpp — the C entry points with made-up deterministic results and the
    library's threading contract (a lock per context) — and RUN here, without a GPU: no report from the sanitizer, the four-thread
    run equals the sequential one, three repetitions are identical, the registry ends empty.  (The GPU suite runs the same harness on
    the real library: tests/test_cpp_dropin.py.)"""
    import numpy as np
    from test_cpp_dropin import write_input, read_dump
    exe = str(tmp_path / ("harness_" + sanitizer))
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", "-pthread", "-I", ROOT, "-I",
                        os.path.join(ROOT, "tests", "stubs"), os.path.join(ROOT, "tests", "cpp", "dropin_harness.cpp"),
                        os.path.join(ROOT, "tests", "cpp", "mock_pli.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(1)
    frames = [(rng.integers(0, 256, (120, 160), dtype=np.uint8), rng.integers(0, 256, (120, 160), dtype=np.uint8)) for _ in range(6)]
    dumps = {}
    # (PLI_FUSION_WAIT_MS: under a sanitizer a thread start alone can take longer than the adapters' 2 ms rendezvous)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=66", PLI_FUSION_WAIT_MS="60")
    for mode in (1, 0):
        inp, outp = str(tmp_path / "in"), str(tmp_path / ("out%d" % mode))
        write_input(inp, frames, 3, mode, nfeatures=500, nlines=60)
        r = subprocess.run([exe, inp, outp], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr, (r.returncode, r.stderr[-3000:])
        dumps[mode] = read_dump(outp)
    a, b = dumps[1], dumps[0]
    assert set(a) == set(b) and all(a[k].tobytes() == b[k].tobytes() for k in a if k not in ("frame_ms", "fusion_stats"))
    # a thread that comes 200 ms late on Frame 2 (the rendezvous waits 60 ms here) (that Frame goes unfused, the later ones fuse again), line extractors on copies of the
    # images (nothing may fuse), and a rig that changes at Frame 3: the same bytes, no sanitizer report
    for name, mode, kw in (("late", 1, dict(delay_frame=2, delay_ms=200)), ("copies", 2, {})):
        inp, outp = str(tmp_path / "in"), str(tmp_path / ("out_" + name))
        write_input(inp, frames, 3, mode, nfeatures=500, nlines=60, **kw)
        r = subprocess.run([exe, inp, outp], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr, (name, r.returncode, r.stderr[-3000:])
        d = read_dump(outp)
        assert all(d[k].tobytes() == b[k].tobytes() for k in b if k not in ("frame_ms", "fusion_stats")), name
        fused, alone, timeouts, mismatched, sleeps, missed = (int(v) for v in d["fusion_stats"][0])
        if name == "late":
            # (deterministic: Frame 2 of each repetition cannot fuse — its fourth thread comes 200 ms after a 60 ms wait — and its waiters
            # time out; the other Frames have 60 ms to meet)
            assert timeouts >= 3 and 9 <= fused <= 6 * 3 - 3, d["fusion_stats"]
        else:
            assert fused == 0 and mismatched + timeouts > 0, d["fusion_stats"]
    # misses are counted per Frame: two Frames in a row with a late thread are 2 misses, not 8 timeouts — nothing sleeps
    # and the Frames after them fuse; an integrator that calls the extractors one after the other (mode 0) misses once per Frame, sleeps
    # after 8 Frames, and every probe Frame after a sleep that fails sends the fusion straight back to sleep (twice as long)
    inp, outp = str(tmp_path / "in"), str(tmp_path / "out_late2")
    write_input(inp, frames, 3, 1, nfeatures=500, nlines=60, delay_frame=2 + 1000, delay_ms=200)
    r = subprocess.run([exe, inp, outp], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr, (r.returncode, r.stderr[-3000:])
    d = read_dump(outp)
    assert all(d[k].tobytes() == b[k].tobytes() for k in b if k not in ("frame_ms", "fusion_stats"))
    fused, alone, timeouts, mismatched, sleeps, missed = (int(v) for v in d["fusion_stats"][0])
    assert sleeps == 0 and 6 <= missed <= 8 and timeouts >= 6 * 3 and 9 <= fused <= 6 * 3 - 6, d["fusion_stats"]
    env2 = dict(env, PLI_FUSION_WAIT_MS="1")
    inp, outp = str(tmp_path / "in"), str(tmp_path / "out_seq")
    write_input(inp, frames, 30, 0, nfeatures=500, nlines=60)
    r = subprocess.run([exe, inp, outp], capture_output=True, text=True, env=env2)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr, (r.returncode, r.stderr[-3000:])
    fused, alone, timeouts, mismatched, sleeps, missed = (int(v) for v in read_dump(outp)["fusion_stats"][0])
    # 180 Frames: 8 missed Frames, 32 asleep, 1 probe, 64 asleep, 1 probe, 128 asleep ... = 10 missed Frames, 3 sleeps (it was 8 x 4 timeouts per 34 Frames)
    assert fused == 0 and missed == 10 and sleeps == 3 and timeouts == 40 and alone == 180 * 4, (fused, alone, timeouts, mismatched, sleeps, missed)
    assert len(set(a["hashes"].ravel().tolist())) == 1 and int(a["groups_left"][0, 0]) == 0
    assert len(a["f0/mvKeys.f"]) >= 200 and len(a["f1/sbp0/match12"]) > 0 and int(a["f1/line_nmatches"][0, 0]) > 0


def test_extractor_registry_pairing_release_and_plibind(tmp_path):
    """tests/cpp/registry_test.cpp (under AddressSanitizer, against the mock library): construction-order pairing as Tracking.cc
    builds the extractors, eye slots given back on destruction, a System torn down and rebuilt three times leaves no group, the
    monocular main / initial extractors in two groups, pliBind for an order the implicit rule gets wrong, lsd_refine != 0 refused."""
    exe = str(tmp_path / "registry_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-pthread", "-I", ROOT, "-I", os.path.join(ROOT, "tests", "stubs"),
                        os.path.join(ROOT, "tests", "cpp", "registry_test.cpp"), os.path.join(ROOT, "tests", "cpp", "mock_pli.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "registry_test: ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_pin_stage3_source_compiles_and_reports_equal_layouts_against_the_stand_in_headers(tmp_path):
    """tools/pin/adapters_against_opencv.cpp is the one-command check for a machine that HAS OpenCV (tools/pin/run_pin.sh stage 3): the
    adapters against the real headers, cv::KeyPoint / KeyLine offsets against pli_keypoint / pli_keyline.  Here it is kept compiling —
    and its offset table kept honest — against the stand-in headers and the mock library."""
    inc = tmp_path / "inc"
    (inc / "opencv2" / "features2d").mkdir(parents=True)
    (inc / "opencv2" / "features2d" / "features2d.hpp").write_text("#include <opencv2/core/core.hpp>\n")
    (inc / "line_descriptor_custom.hpp").write_text(
        "#pragma once\n#include <opencv2/core/core.hpp>\nnamespace cv { namespace line_descriptor {\n"
        "struct KeyLine { float angle; int class_id; int octave; cv::Point2f pt; float response; float size; float startPointX, startPointY, "
        "endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY; float lineLength; int numOfPixels; };\n}}\n")
    exe = str(tmp_path / "adp")
    r = subprocess.run(["g++", "-std=c++17", "-I", ROOT, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "stubs"), "-I", str(inc),
                        os.path.join(ROOT, "tools", "pin", "adapters_against_opencv.cpp"), os.path.join(ROOT, "tests", "cpp", "mock_pli.cpp"),
                        "-pthread", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "layouts agree" in r.stdout and "differs" not in r.stdout, r.stdout
