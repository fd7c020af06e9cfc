// Hand-off artefact: pins the oracle's OpenCV primitives to a REAL OpenCV build.
//
// The build image of this repository has no OpenCV, so oracle/ocv_prims.hpp and oracle/line_oracle.hpp restate OpenCV 3.3.1
// "by intent" (DESIGN.md "Oracle").  Whoever has the toolchain the reference names (OpenCV 3.3.1, README.md:18-20) turns that
// into a pin with one command:
//
//     tools/pin/run_pin.sh  [/path/to/opencv/prefix]
//
// which (1) dumps the seeded synthetic inputs (tools/pin/dump_inputs.py), (2) builds and runs THIS program against the
// installed OpenCV — it writes what the real cv:: functions return on those inputs as .npy files into tests/golden/opencv_pin/
// — and (3) runs tools/pin/pin_compare.py (also executed by tests/test_opencv_pin.py whenever that directory exists), which
// compares every primitive with the oracle and reports which PLI_PARITY_* flag set reproduces the real library.
//
//     g++ -std=c++11 -O2 pin_against_opencv.cpp -o pin_against_opencv `pkg-config --cflags --libs opencv`
//     ./pin_against_opencv <inputs dir> <output dir>
//
// Every call below is the call the reference makes (file:line under the PLI-SLAM tree) with the reference's arguments.
#include <opencv2/core/core.hpp>
#include <opencv2/core/version.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

// ---- minimal .npy writer (version 1.0, C order, little endian) ----------------------------------------------------------
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
static void writeMatU8(const std::string& p, const cv::Mat& m) {
  cv::Mat c = m.isContinuous() ? m : m.clone();
  writeNpy(p, "|u1", {c.rows, c.cols}, c.data, (size_t)c.rows * c.cols);
}
static void writeMatS16(const std::string& p, const cv::Mat& m) {
  cv::Mat c = m.isContinuous() ? m : m.clone();
  writeNpy(p, "<i2", {c.rows, c.cols}, c.data, (size_t)c.rows * c.cols * 2);
}

// which overload does an unqualified cos(float) pick in a translation unit like the reference's?
namespace like_ORBextractor_cc { using namespace std; static const bool cosOfFloatIsFloat = is_same<decltype(cos(1.0f)), float>::value; }
namespace cv { namespace like_lsd_cpp { static const bool cosOfFloatIsFloat = std::is_same<decltype(cos(float(1.0))), float>::value; } }

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <inputs dir> <output dir>\n", argv[0]); return 2; }
  const std::string in = argv[1], out = argv[2];
  std::ifstream list((in + "/inputs.txt").c_str());
  std::string name;
  int w, h;
  std::ofstream manifest((out + "/manifest.txt").c_str());
  manifest << "opencv " << CV_VERSION << "\n";
  manifest << "cos(float) under `using namespace std` (ORBextractor.cc:65,111) is " << (like_ORBextractor_cc::cosOfFloatIsFloat ? "float (cosf)" : "double") << "\n";
  manifest << "cos(float) unqualified inside namespace cv with these headers (lsd.cpp region_grow, binary_descriptor_custom.cpp:1130) is "
           << (cv::like_lsd_cpp::cosOfFloatIsFloat ? "float (cosf)" : "double") << "\n";
  // libm of this machine on a dense set of floats: cosf / sinf bits (decides PLI_PARITY_TRIG_F32_* = glibc >= 2.28 or not)
  {
    std::vector<float> xs, c, s;
    for (uint32_t b = 0x3A000000u; b < 0x40C90FDBu; b += 4099) { float x; std::memcpy(&x, &b, 4); xs.push_back(x); c.push_back(cosf(x)); s.push_back(sinf(x)); }
    writeNpy(out + "/libm_x.npy", "<f4", {(long)xs.size()}, xs.data(), xs.size() * 4);
    writeNpy(out + "/libm_cosf.npy", "<f4", {(long)c.size()}, c.data(), c.size() * 4);
    writeNpy(out + "/libm_sinf.npy", "<f4", {(long)s.size()}, s.data(), s.size() * 4);
  }
  // cv::fastAtan2 (ORBextractor.cc:101, lsd.cpp) on a grid
  {
    std::vector<float> yx, a;
    for (int y = -40; y <= 40; y += 3) for (int x = -40; x <= 40; x += 3) { yx.push_back(y * 1.37f); yx.push_back(x * 0.91f); a.push_back(cv::fastAtan2(y * 1.37f, x * 0.91f)); }
    writeNpy(out + "/fastatan2_yx.npy", "<f4", {(long)a.size(), 2}, yx.data(), yx.size() * 4);
    writeNpy(out + "/fastatan2.npy", "<f4", {(long)a.size()}, a.data(), a.size() * 4);
  }
  while (list >> name >> w >> h) {
    std::vector<unsigned char> buf((size_t)w * h);
    std::ifstream f((in + "/" + name + ".raw").c_str(), std::ios::binary);
    f.read((char*)buf.data(), buf.size());
    cv::Mat img(h, w, CV_8UC1, buf.data());
    const std::string o = out + "/" + name + "_";
    // ORBextractor::ComputePyramid level 1 (ORBextractor.cc:1157-1166): resize(level0, level1, sz, 0, 0, INTER_LINEAR)
    {
      const float inv = 1.0f / 1.2f;
      cv::Size sz(cvRound((float)w * inv), cvRound((float)h * inv));
      cv::Mat d;
      cv::resize(img, d, sz, 0, 0, cv::INTER_LINEAR);
      writeMatU8(o + "resize_level1.npy", d);
    }
    // GaussianBlur(workingMat, workingMat, Size(7, 7), 2, 2, BORDER_REFLECT_101) (ORBextractor.cc:1115) and the 5x5 sigma 1 of
    // BinaryDescriptor::computeSobel (binary_descriptor_custom.cpp:358)
    { cv::Mat d; cv::GaussianBlur(img, d, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101); writeMatU8(o + "blur7_s2.npy", d); }
    cv::Mat b5;
    cv::GaussianBlur(img, b5, cv::Size(5, 5), 1);
    writeMatU8(o + "blur5_s1.npy", b5);
    // Sobel dx / dy CV_16S ksize 3 (binary_descriptor_custom.cpp:395-396)
    { cv::Mat dx, dy; cv::Sobel(b5, dx, CV_16SC1, 1, 0, 3); cv::Sobel(b5, dy, CV_16SC1, 0, 1, 3); writeMatS16(o + "sobel_dx.npy", dx); writeMatS16(o + "sobel_dy.npy", dy); }
    // FAST(cell, kps, iniThFAST / minThFAST, true) (ORBextractor.cc:803-815) on the whole image
    for (int th : {20, 7}) {
      std::vector<cv::KeyPoint> kps;
      cv::FAST(img, kps, th, true);
      std::vector<float> t;
      for (size_t i = 0; i < kps.size(); ++i) { t.push_back(kps[i].pt.x); t.push_back(kps[i].pt.y); t.push_back(kps[i].response); }
      writeNpy(o + "fast_t" + std::to_string(th) + ".npy", "<f4", {(long)kps.size(), 3}, t.data(), t.size() * 4);
    }
    // createLineSegmentDetector(refine 0, 1.2, 0.6, 2.0, 22.5, 0, 0.7, 1024)->detect (LSDDetector_custom.cpp:233-259 with the
    // values of Examples/Stereo/Config/EuRoC.yaml:156-164 / Config.cpp)
    {
      cv::Ptr<cv::LineSegmentDetector> ls = cv::createLineSegmentDetector(cv::LSD_REFINE_NONE, 1.2, 0.6, 2.0, 22.5, 0, 0.7, 1024);
      std::vector<cv::Vec4f> lines;
      ls->detect(img, lines);
      writeNpy(o + "lsd_segments.npy", "<f4", {(long)lines.size(), 4}, lines.empty() ? (const void*)&lines : (const void*)&lines[0], lines.size() * 16);
    }
    manifest << "image " << name << " " << w << " " << h << "\n";
  }
  return 0;
}
