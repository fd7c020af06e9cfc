// Stage 3 of tools/pin/run_pin.sh (round 5; cannot run in the build image: no OpenCV there): compiles the C++ drop-in
// (pli_slam_amd/adapters/orbslam_adapters.hpp + frame_stereo.hpp) against the REAL OpenCV headers of the machine and the reference's own
// Thirdparty/line_descriptor header — the GPU suite only ever sees tests/stubs/opencv2 —, and checks that the C ABI's table
// records can be copied into cv::KeyPoint / cv::line_descriptor::KeyLine field for field:
//   pli_keypoint  <->  cv::KeyPoint                       (ORBextractor.cc:868-877,1131 fill pt, size, angle, response, octave)
//   pli_keyline   <->  cv::line_descriptor::KeyLine       (Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:105-144)
// Build (run_pin.sh does it):  g++ -std=c++11 -fsyntax-only ... for the compile check, and a linked run that prints the offsets.
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <line_descriptor_custom.hpp>      // -I$PLI_SLAM_ROOT/Thirdparty/line_descriptor/include
#define PLI_ADAPTER_NO_KEYLINE_HEADER
#define PLI_ADAPTER_KEYLINE_TYPE cv::line_descriptor::KeyLine
#include "pli_slam_amd/adapters/orbslam_adapters.hpp"
#include "pli_slam_amd/adapters/frame_stereo.hpp"
#include <cstddef>
#include <cstdio>

// the reference's call sites, as they appear in Frame.cc:128-163 / Tracking.cc:3046-3058: they must compile unchanged against the adapters
static void call_sites(ORB_SLAM3::ORBextractor* orb, ORB_SLAM3::Lineextractor* lines, const cv::Mat& im) {
  std::vector<cv::KeyPoint> keys;
  cv::Mat desc, ldesc;
  std::vector<int> lapping = {0, 1000};
  (*orb)(im, cv::Mat(), keys, desc, lapping);
  std::vector<cv::line_descriptor::KeyLine> klines;
  (*lines)(im, cv::Mat(), klines, ldesc);
}

#define OFF(T, f) (unsigned)offsetof(T, f)
int main() {
  typedef cv::line_descriptor::KeyLine KL;
  int bad = 0;
  auto chk = [&](const char* what, unsigned a, unsigned b) {
    std::printf("  %-28s %3u  %3u %s\n", what, a, b, a == b ? "" : "  <-- differs");
    bad += a != b;
  };
  std::printf("cv::KeyPoint (sizeof %u) against pli_keypoint (sizeof %u): the adapters copy field by field, only the relative order matters\n",
              (unsigned)sizeof(cv::KeyPoint), (unsigned)sizeof(pli_keypoint));
  chk("pt.x", OFF(cv::KeyPoint, pt), OFF(pli_keypoint, x));
  chk("size", OFF(cv::KeyPoint, size), OFF(pli_keypoint, size));
  chk("angle", OFF(cv::KeyPoint, angle), OFF(pli_keypoint, angle));
  chk("response", OFF(cv::KeyPoint, response), OFF(pli_keypoint, response));
  chk("octave", OFF(cv::KeyPoint, octave), OFF(pli_keypoint, octave));
  std::printf("cv::line_descriptor::KeyLine (sizeof %u) against pli_keyline (sizeof %u): must be identical (memcpy-able)\n", (unsigned)sizeof(KL),
              (unsigned)sizeof(pli_keyline));
  chk("angle", OFF(KL, angle), OFF(pli_keyline, angle));
  chk("class_id", OFF(KL, class_id), OFF(pli_keyline, class_id));
  chk("octave", OFF(KL, octave), OFF(pli_keyline, octave));
  chk("pt", OFF(KL, pt), OFF(pli_keyline, pt_x));
  chk("response", OFF(KL, response), OFF(pli_keyline, response));
  chk("size", OFF(KL, size), OFF(pli_keyline, size));
  chk("startPointX", OFF(KL, startPointX), OFF(pli_keyline, startPointX));
  chk("endPointY", OFF(KL, endPointY), OFF(pli_keyline, endPointY));
  chk("sPointInOctaveX", OFF(KL, sPointInOctaveX), OFF(pli_keyline, sPointInOctaveX));
  chk("ePointInOctaveY", OFF(KL, ePointInOctaveY), OFF(pli_keyline, ePointInOctaveY));
  chk("lineLength", OFF(KL, lineLength), OFF(pli_keyline, lineLength));
  chk("numOfPixels", OFF(KL, numOfPixels), OFF(pli_keyline, numOfPixels));
  chk("sizeof", (unsigned)sizeof(KL), (unsigned)sizeof(pli_keyline));
  (void)&call_sites;
  std::printf(bad ? "LAYOUT MISMATCH: %d field(s)\n" : "layouts agree\n", bad);
  return bad ? 1 : 0;
}
