// Hand-off artefact, stage 2 (needs the PLI-SLAM tree AND its toolchain: OpenCV 3.3.1, Eigen 3): pins the WHOLE extractors.
//
// pin_against_opencv.cpp pins the OpenCV primitives one by one.  What that leaves open is everything the reference builds on top of
// them with implementation-defined order: the quadtree's tie-break between nodes of equal size (std::sort of pair<int, node*>,
// ORBextractor.cc:682), the seed order of equal-gradient pixels inside cv::LineSegmentDetector, the unstable sort by response in
// Lineextractor (LineExtractor.cc:59).  This program links the reference's OWN sources where they lie —
//   src/ORBextractor.cc, src/LineExtractor.cc, src/Config.cpp, Thirdparty/line_descriptor/src/{LSDDetector_custom,
//   binary_descriptor_custom}.cpp — and runs ORB_SLAM3::ORBextractor / Lineextractor exactly as Tracking.cc constructs them
// (EuRoC.yaml values) on the seeded inputs of tools/pin/dump_inputs.py; it writes the keypoint tables, rBRIEF descriptors, KeyLine
// tables and LBD descriptors as .npy next to the primitives' outputs, and tools/pin/pin_compare.py diffs them with the oracle field
// by field (section "reference extractors").  Built by tools/pin/run_pin.sh when PLI_SLAM_ROOT is set:
//
//   g++ -std=c++11 -O3 -march=native pin_reference_extractors.cpp $R/src/ORBextractor.cc $R/src/LineExtractor.cc $R/src/Config.cpp \
//       $R/Thirdparty/line_descriptor/src/LSDDetector_custom.cpp $R/Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp \
//       -I$R -I$R/include -I$R/Thirdparty/line_descriptor/include -I/usr/include/eigen3 `pkg-config --cflags --libs opencv`
//
// It has never been compiled in this repository's build image (no OpenCV, no Eigen): it is written against the reference's headers
// (include/ORBextractor.h:53-87, include/LineExtractor.h:44-51, descriptor_custom.hpp:105-144) and kept deliberately plain.
#include <opencv2/core/core.hpp>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "ORBextractor.h"
#include "LineExtractor.h"

static void writeNpy(const std::string& path, const char* descr, const std::vector<long>& shape, const void* data, size_t bytes) {
  std::ostringstream h;
  h << "{'descr': '" << descr << "', 'fortran_order': False, 'shape': (";
  for (size_t i = 0; i < shape.size(); ++i) h << shape[i] << (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
  h << "), }";
  std::string hs = h.str();
  size_t total = 10 + hs.size() + 1;
  hs.append((64 - total % 64) % 64, ' ');
  hs.push_back('\n');
  std::ofstream f(path.c_str(), std::ios::binary);
  const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  f.write((const char*)magic, 8);
  const uint16_t hl = (uint16_t)hs.size();
  f.write((const char*)&hl, 2);
  f.write(hs.data(), hs.size());
  f.write((const char*)data, bytes);
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: pin_reference_extractors <inputs dir> <output dir>\n"); return 2; }
  const std::string in = argv[1], out = argv[2];
  std::ifstream man((out + "/manifest.txt").c_str());
  std::string line;
  int done = 0;
  while (std::getline(man, line)) {
    std::istringstream ls(line);
    std::string tag, name;
    int w = 0, h = 0;
    ls >> tag >> name >> w >> h;
    if (tag != "image" || w <= 0 || h <= 0) continue;
    cv::Mat img(h, w, CV_8UC1);
    std::ifstream f((in + "/" + name + ".raw").c_str(), std::ios::binary);
    f.read((char*)img.data, (std::streamsize)w * h);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", name.c_str()); return 1; }
    // Tracking.cc:743 with Examples/Stereo/Config/EuRoC.yaml:91-104: 1200 features, 1.2, 8 levels, FAST 20 / 7
    ORB_SLAM3::ORBextractor orb(1200, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    std::vector<int> lap = {0, 0};                      // Frame::ExtractORB(flag, im, 0, 0), Frame.cc:484-491
    orb(img, cv::Mat(), kps, desc, lap);
    std::vector<float> kf(kps.size() * 5);
    std::vector<int32_t> ki(kps.size() * 2);
    for (size_t i = 0; i < kps.size(); ++i) {
      kf[5 * i] = kps[i].pt.x; kf[5 * i + 1] = kps[i].pt.y; kf[5 * i + 2] = kps[i].size; kf[5 * i + 3] = kps[i].angle; kf[5 * i + 4] = kps[i].response;
      ki[2 * i] = kps[i].octave; ki[2 * i + 1] = kps[i].class_id;
    }
    writeNpy(out + "/" + name + "_ref_orb_kp_f.npy", "<f4", {(long)kps.size(), 5}, kf.data(), kf.size() * 4);
    writeNpy(out + "/" + name + "_ref_orb_kp_i.npy", "<i4", {(long)kps.size(), 2}, ki.data(), ki.size() * 4);
    cv::Mat dc = desc.isContinuous() ? desc : desc.clone();
    writeNpy(out + "/" + name + "_ref_orb_desc.npy", "|u1", {(long)dc.rows, 32}, dc.data, (size_t)dc.rows * 32);
    // Tracking.cc:87-90 with EuRoC.yaml:150,156-164: keep all lines (lsd_nfeatures 0 here, so that the top-N cut does not hide a
    // difference) and the yaml's detector options
    ORB_SLAM3::Lineextractor lines(0, 0.025, 0, 1.2, 0.6, 2.0, 22.5, 1.0, 0.6, 1024, false);
    std::vector<cv::line_descriptor::KeyLine> kl;
    cv::Mat ld;
    lines(img, cv::Mat(), kl, ld);
    std::vector<float> lf(kl.size() * 14);
    std::vector<int32_t> li(kl.size() * 3);
    for (size_t i = 0; i < kl.size(); ++i) {
      const cv::line_descriptor::KeyLine& s = kl[i];
      const float v[14] = {s.angle, s.pt.x, s.pt.y, s.response, s.size, s.startPointX, s.startPointY, s.endPointX, s.endPointY,
                           s.sPointInOctaveX, s.sPointInOctaveY, s.ePointInOctaveX, s.ePointInOctaveY, s.lineLength};
      for (int j = 0; j < 14; ++j) lf[14 * i + j] = v[j];
      li[3 * i] = s.class_id; li[3 * i + 1] = s.octave; li[3 * i + 2] = s.numOfPixels;
    }
    writeNpy(out + "/" + name + "_ref_kl_f.npy", "<f4", {(long)kl.size(), 14}, lf.data(), lf.size() * 4);
    writeNpy(out + "/" + name + "_ref_kl_i.npy", "<i4", {(long)kl.size(), 3}, li.data(), li.size() * 4);
    cv::Mat lc = ld.empty() ? cv::Mat(0, 32, CV_8U) : (ld.isContinuous() ? ld : ld.clone());
    writeNpy(out + "/" + name + "_ref_kl_desc.npy", "|u1", {(long)lc.rows, 32}, lc.data, (size_t)lc.rows * 32);
    std::printf("%s: %zu keypoints, %zu key lines from the reference's own extractors\n", name.c_str(), kps.size(), kl.size());
    ++done;
  }
  return done ? 0 : 1;
}
