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
