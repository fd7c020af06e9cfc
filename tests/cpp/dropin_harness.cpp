// Drop-in harness (test infrastructure, -m gpu): drives pli_slam_amd/adapters/ exactly the way the reference drives the
// classes they replace, on a machine without OpenCV (tests/stubs/opencv2 is a functional stand-in for the cv:: types):
//
//   * the extractors are constructed as Tracking::Tracking does            (Tracking.cc:743-746, then :87-94),
//   * every Frame runs ExtractORB x2 + ExtractLine x2 on four std::threads (Frame.cc:128-135) — or, with mode 0, as four
//     sequential calls —, then ComputeStereoMatches_Lines / ComputeStereoMatches (Frame.cc:158-163) through
//     adapters/frame_stereo.hpp,
//   * consecutive frames go through match(desc1, desc2, nnr, matches_12)   (Tracking.cc:2717,3058; LineMatcher.h:63) and
//     ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, match12) (Tracking.cc:3046; ORBmatcher.cc:2179).
//
// Every container the reference's Frame would hold afterwards is written to a file of named arrays; the whole sequence
// is repeated `reps` times and a hash of all containers is recorded per repetition.  tests/test_cpp_dropin.py compares
// the arrays byte for byte with the oracle and with the ctypes path, the four-thread run with the sequential one, and
// the repetitions with each other.
//
//   usage: dropin_harness <in> <out>
//   in: "PLIH" i32 W H nframes reps mode(0 = four calls in a row, 1 = four threads, 2 = four threads, line extractors on copies of the images, 3 = four threads, the rig handed to the extractors before the first Frame: pliSetStereoCamera, 4 = four threads, mvImagePyramid not copied back: ORBextractor::pliCopyPyramidBack(false)) nfeatures nlines rigChangeFrame(-1: never) delayFrame(-1: never) delayMs
//       | f32 rig A: fx fy cx cy bf | f32 rig B (Frames from rigChangeFrame on) | images
//   delayFrame: on that Frame (every repetition) the thread of the right line extractor starts delayMs late — a loaded host
//   (VERDICT r3 item 7): that Frame goes unfused, the next ones must fuse again.  delayFrame + 1000 * (n - 1): n Frames in a row.
#define PLI_ADAPTER_NO_KEYLINE_HEADER
#define PLI_ADAPTER_KEYLINE_TYPE cv::line_descriptor::KeyLine
#include <opencv2/core/core.hpp>
namespace cv { namespace line_descriptor {
struct KeyLine {      // field names as Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:105-144
  float angle; int class_id; int octave; cv::Point2f pt; float response; float size;
  float startPointX, startPointY, endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength; int numOfPixels;
};
}}
#include "pli_slam_amd/adapters/frame_stereo.hpp"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <thread>
#include <chrono>

using namespace ORB_SLAM3;
using cv::line_descriptor::KeyLine;

struct Vector3d {                      // Eigen::Vector3d as Frame::mvle_l uses it: three doubles, (x, y, z) constructor
  double v[3];
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double a, double b, double c) : v{a, b, c} {}
};

struct MapPoint {                      // the three MapPoint methods SearchByProjection calls (include/MapPoint.h)
  cv::Mat mWorldPos, mDescriptor;
  int nObs = 0;
  cv::Mat GetWorldPos() { return mWorldPos.clone(); }
  cv::Mat GetDescriptor() { return mDescriptor.clone(); }
  int Observations() { return nObs; }
};

// The members of ORB_SLAM3::Frame on the front-end path, with the constructor flow of Frame.cc:98-228.
class Frame {
 public:
  Frame(const cv::Mat& imLeft, const cv::Mat& imRight, ORBextractor* extractorLeft, ORBextractor* extractorRight,
        Lineextractor* LineextractorLeft, Lineextractor* LineextractorRight, cv::Mat& K, const float& bf, bool fourThreads, int lateMs = 0, bool cloneLineImages = false)
      : mpORBextractorLeft(extractorLeft), mpORBextractorRight(extractorRight), mpLineextractorLeft(LineextractorLeft),
        mpLineextractorRight(LineextractorRight), mK(K.clone()), mbf(bf) {
    mvScaleFactors = mpORBextractorLeft->GetScaleFactors();            // :117-123
    if (fourThreads) {                                                  // :128-135
      std::thread threadLeft(&Frame::ExtractORB, this, 0, imLeft, 0, 0);
      std::thread threadRight(&Frame::ExtractORB, this, 1, imRight, 0, 0);
      // (cloneLineImages: an integrator that hands the line extractors its own copies — the four calls are not ONE Frame's
      // images any more and must not be fused onto the ORB extractors' images)
      const cv::Mat lineLeft = cloneLineImages ? imLeft.clone() : imLeft, lineRight = cloneLineImages ? imRight.clone() : imRight;
      std::thread threadLeft_Line(&Frame::ExtractLine, this, 0, lineLeft);
      std::thread threadRight_Line([this, &lineRight, lateMs] {
        if (lateMs > 0) std::this_thread::sleep_for(std::chrono::milliseconds(lateMs));
        ExtractLine(1, lineRight);
      });
      threadLeft.join();
      threadRight.join();
      threadLeft_Line.join();
      threadRight_Line.join();
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      ExtractORB(0, imLeft, 0, 0);
      ExtractORB(1, imRight, 0, 0);
      const auto t1 = std::chrono::steady_clock::now();
      ExtractLine(0, imLeft);
      ExtractLine(1, imRight);
      const auto t2 = std::chrono::steady_clock::now();
      stageSeconds[0] += std::chrono::duration<double>(t1 - t0).count();
      stageSeconds[1] += std::chrono::duration<double>(t2 - t1).count();
    }
    N = (int)mvKeys.size();                                             // :143
    if (mvKeys.empty()) return;                                         // :146-149
    if (mvKeys_Line.empty()) return;
    mvKeysUn = mvKeys;                                                  // UndistortKeyPoints with mDistCoef = 0 (:151)
    const auto t3 = std::chrono::steady_clock::now();
    ComputeStereoMatches_Lines();                                       // :158-163
    N_l = (int)mvKeys_Line.size();
    ComputeStereoMatches();
    stageSeconds[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t3).count();
    mvpMapPoints = std::vector<MapPoint*>(N, static_cast<MapPoint*>(NULL));     // :171-174
    mvbOutlier = std::vector<bool>(N, false);
    mnMinX = 0.0f; mnMaxX = (float)imLeft.cols; mnMinY = 0.0f; mnMaxY = (float)imLeft.rows;   // ComputeImageBounds, no distortion (:182)
    fx = K.at<float>(0, 0); fy = K.at<float>(1, 1); cx = K.at<float>(0, 2); cy = K.at<float>(1, 2);   // :187-190
    mb = mbf / fx;                                                      // :197
  }
  void ExtractORB(int flag, const cv::Mat& im, const int x0, const int x1) {          // Frame.cc:484-491
    std::vector<int> vLapping = {x0, x1};
    if (flag == 0) monoLeft = (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors, vLapping);
    else monoRight = (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight, vLapping);
  }
  void ExtractLine(int flag, const cv::Mat& im) {                                      // Frame.cc:508-514
    if (flag == 0) (*mpLineextractorLeft)(im, cv::Mat(), mvKeys_Line, mDescriptors_Line);
    else (*mpLineextractorRight)(im, cv::Mat(), mvKeysRight_Line, mDescriptorsRight_Line);
  }
  void ComputeStereoMatches() { pli_frame::ComputeStereoMatches(*this); }                            // Frame.h:152
  void ComputeStereoMatches_Lines(bool initial = true) { pli_frame::ComputeStereoMatches_Lines(*this, initial); }   // Frame.h:154

  static double stageSeconds[3];     // (sequential mode) ORB x2, lines x2, the two stereo matchers
  ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
  Lineextractor *mpLineextractorLeft, *mpLineextractorRight;
  cv::Mat mK;
  float mbf, mb = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  int N = 0, N_l = 0, monoLeft = -1, monoRight = -1;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
  cv::Mat mDescriptors, mDescriptorsRight;
  std::vector<float> mvuRight, mvDepth;
  std::vector<KeyLine> mvKeys_Line, mvKeysRight_Line;
  cv::Mat mDescriptors_Line, mDescriptorsRight_Line;
  std::vector<std::pair<float, float>> mvDisparity_l;
  std::vector<Vector3d> mvle_l;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<float> mvScaleFactors;
  float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
  cv::Mat mTcw;
};

double Frame::stageSeconds[3] = {0, 0, 0};

// ---- output: named arrays -------------------------------------------------------------------------------------------
struct Dump {
  FILE* f;
  explicit Dump(const char* path) : f(std::fopen(path, "wb")) { if (f) std::fwrite("PLID", 1, 4, f); }
  ~Dump() { if (f) std::fclose(f); }
  // dtype: 'B' u8, 'i' i32, 'f' f32, 'd' f64, 'Q' u64
  void put(const std::string& name, char dtype, int rows, int cols, const void* data) {
    const int esz = dtype == 'B' ? 1 : (dtype == 'd' || dtype == 'Q') ? 8 : 4;
    const int32_t hdr[4] = {(int32_t)name.size(), (int32_t)dtype, rows, cols};
    std::fwrite(hdr, 4, 4, f);
    std::fwrite(name.data(), 1, name.size(), f);
    if (rows > 0 && cols > 0) std::fwrite(data, esz, (size_t)rows * cols, f);
  }
};

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

static std::vector<float> keypointRows(const std::vector<cv::KeyPoint>& k, std::vector<int>& ints) {
  std::vector<float> r(k.size() * 5);
  ints.resize(k.size() * 2);
  for (size_t i = 0; i < k.size(); ++i) {
    r[5 * i] = k[i].pt.x; r[5 * i + 1] = k[i].pt.y; r[5 * i + 2] = k[i].size; r[5 * i + 3] = k[i].angle; r[5 * i + 4] = k[i].response;
    ints[2 * i] = k[i].octave; ints[2 * i + 1] = k[i].class_id;
  }
  return r;
}
static std::vector<float> keylineRows(const std::vector<KeyLine>& k, std::vector<int>& ints) {
  std::vector<float> r(k.size() * 14);
  ints.resize(k.size() * 3);
  for (size_t i = 0; i < k.size(); ++i) {
    const KeyLine& s = k[i];
    const float v[14] = {s.angle, s.pt.x, s.pt.y, s.response, s.size, s.startPointX, s.startPointY, s.endPointX, s.endPointY,
                         s.sPointInOctaveX, s.sPointInOctaveY, s.ePointInOctaveX, s.ePointInOctaveY, s.lineLength};
    for (int j = 0; j < 14; ++j) r[14 * i + j] = v[j];
    ints[3 * i] = s.class_id; ints[3 * i + 1] = s.octave; ints[3 * i + 2] = s.numOfPixels;
  }
  return r;
}
static std::vector<uint8_t> descRows(const cv::Mat& d) {
  std::vector<uint8_t> r((size_t)d.rows * 32);
  for (int i = 0; i < d.rows; ++i) std::memcpy(&r[(size_t)i * 32], d.ptr(i), 32);
  return r;
}

// everything a Frame holds after its constructor: hashed always, written when `out` is set
static uint64_t record(const Frame& F, Dump* out, const std::string& pre, uint64_t h) {
  auto emit = [&](const std::string& name, char dt, int rows, int cols, const void* p) {
    const int esz = dt == 'B' ? 1 : dt == 'd' ? 8 : 4;
    h = fnv(h, &rows, 4);
    h = fnv(h, p, (size_t)rows * cols * esz);
    if (out) out->put(pre + name, dt, rows, cols, p);
  };
  std::vector<int> ki;
  std::vector<float> kf;
  kf = keypointRows(F.mvKeys, ki);      emit("mvKeys.f", 'f', (int)F.mvKeys.size(), 5, kf.data());           emit("mvKeys.i", 'i', (int)F.mvKeys.size(), 2, ki.data());
  kf = keypointRows(F.mvKeysRight, ki); emit("mvKeysRight.f", 'f', (int)F.mvKeysRight.size(), 5, kf.data()); emit("mvKeysRight.i", 'i', (int)F.mvKeysRight.size(), 2, ki.data());
  std::vector<uint8_t> d;
  d = descRows(F.mDescriptors);      emit("mDescriptors", 'B', F.mDescriptors.rows, 32, d.data());
  d = descRows(F.mDescriptorsRight); emit("mDescriptorsRight", 'B', F.mDescriptorsRight.rows, 32, d.data());
  emit("mvuRight", 'f', (int)F.mvuRight.size(), 1, F.mvuRight.data());
  emit("mvDepth", 'f', (int)F.mvDepth.size(), 1, F.mvDepth.data());
  kf = keylineRows(F.mvKeys_Line, ki);      emit("mvKeys_Line.f", 'f', (int)F.mvKeys_Line.size(), 14, kf.data());           emit("mvKeys_Line.i", 'i', (int)F.mvKeys_Line.size(), 3, ki.data());
  kf = keylineRows(F.mvKeysRight_Line, ki); emit("mvKeysRight_Line.f", 'f', (int)F.mvKeysRight_Line.size(), 14, kf.data()); emit("mvKeysRight_Line.i", 'i', (int)F.mvKeysRight_Line.size(), 3, ki.data());
  d = descRows(F.mDescriptors_Line);      emit("mDescriptors_Line", 'B', F.mDescriptors_Line.rows, 32, d.data());
  d = descRows(F.mDescriptorsRight_Line); emit("mDescriptorsRight_Line", 'B', F.mDescriptorsRight_Line.rows, 32, d.data());
  std::vector<float> disp(F.mvDisparity_l.size() * 2);
  for (size_t i = 0; i < F.mvDisparity_l.size(); ++i) { disp[2 * i] = F.mvDisparity_l[i].first; disp[2 * i + 1] = F.mvDisparity_l[i].second; }
  emit("mvDisparity_l", 'f', (int)F.mvDisparity_l.size(), 2, disp.data());
  std::vector<double> le(F.mvle_l.size() * 3);
  for (size_t i = 0; i < F.mvle_l.size(); ++i) for (int j = 0; j < 3; ++j) le[3 * i + j] = F.mvle_l[i].v[j];
  emit("mvle_l", 'd', (int)F.mvle_l.size(), 3, le.data());
  const int mono[2] = {F.monoLeft, F.monoRight};
  emit("mono", 'i', 1, 2, mono);
  // public member ORBextractor::mvImagePyramid (ORBextractor.h:87): sizes + a checksum per level, level 0 and the last in full
  const std::vector<cv::Mat>& pyr = F.mpORBextractorLeft->mvImagePyramid;
  for (size_t l = 0; l < pyr.size(); ++l) {
    std::vector<uint8_t> flat((size_t)pyr[l].rows * pyr[l].cols);
    for (int r = 0; r < pyr[l].rows; ++r) std::memcpy(&flat[(size_t)r * pyr[l].cols], pyr[l].ptr(r), pyr[l].cols);
    h = fnv(h, flat.data(), flat.size());
    if (out && (l == 0 || l + 1 == pyr.size())) out->put(pre + "pyrL" + std::to_string(l), 'B', pyr[l].rows, pyr[l].cols, flat.data());
  }
  return h;
}

static cv::Mat pose(float rz, float ry, float tx, float ty, float tz) {      // Tcw = [Ry * Rz | t], angles in radians
  cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
  const float cz = std::cos(rz), sz = std::sin(rz), cy = std::cos(ry), sy = std::sin(ry);
  const float R[9] = {cy * cz, -cy * sz, sy, sz, cz, 0.0f, -sy * cz, sy * sz, cy};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T.at<float>(i, j) = R[3 * i + j];
  T.at<float>(0, 3) = tx; T.at<float>(1, 3) = ty; T.at<float>(2, 3) = tz;
  return T;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: dropin_harness <in> <out>\n"); return 2; }
  FILE* in = std::fopen(argv[1], "rb");
  if (!in) { std::perror(argv[1]); return 2; }
  char magic[4];
  int32_t hd[10];
  float rigs[10];
  if (std::fread(magic, 1, 4, in) != 4 || std::memcmp(magic, "PLIH", 4) || std::fread(hd, 4, 10, in) != 10 || std::fread(rigs, 4, 10, in) != 10) {
    std::fprintf(stderr, "bad input header\n");
    return 2;
  }
  const int W = hd[0], H = hd[1], nframes = hd[2], reps = hd[3], mode = hd[4], nFeatures = hd[5], lsdNFeatures = hd[6];
  // (delayFrame + 1000 * (n - 1): n consecutive Frames from delayFrame on have their late thread)
  const int rigChangeFrame = hd[7], delayFrame = hd[8] < 0 ? -1 : hd[8] % 1000, delayCount = hd[8] < 0 ? 0 : hd[8] / 1000 + 1, delayMs = hd[9];
  std::vector<cv::Mat> imgs((size_t)nframes * 2);
  for (auto& m : imgs) {
    m.create(H, W, CV_8UC1);
    if (std::fread(m.data, 1, (size_t)W * H, in) != (size_t)W * H) { std::fprintf(stderr, "short input\n"); return 2; }
  }
  std::fclose(in);
  try {
    // Tracking::Tracking: ParseORBParamFile builds the two ORB extractors (Tracking.cc:743-746), then the constructor the two
    // line extractors (:87-94); values of Examples/Stereo/Config/EuRoC.yaml except the two budgets the input names
    ORBextractor* mpORBextractorLeft = new ORBextractor(nFeatures, 1.2f, 8, 20, 7);
    ORBextractor* mpORBextractorRight = new ORBextractor(nFeatures, 1.2f, 8, 20, 7);
    Lineextractor* mpLineextractorLeft = new Lineextractor(lsdNFeatures, 0.025, 0, 1.2, 0.6, 2.0, 22.5, 1.0, 0.6, 1024, false);
    Lineextractor* mpLineextractorRight = new Lineextractor(lsdNFeatures, 0.025, 0, 1.2, 0.6, 2.0, 22.5, 1.0, 0.6, 1024, false);
    // the calibration Tracking::ParseCamParamFile reads (Tracking.cc:520-640): K and mbf, one per rig of the input
    cv::Mat Ks[2];
    float bfs[2];
    for (int r = 0; r < 2; ++r) {
      Ks[r] = cv::Mat::eye(3, 3, CV_32F);
      Ks[r].at<float>(0, 0) = rigs[5 * r]; Ks[r].at<float>(1, 1) = rigs[5 * r + 1];
      Ks[r].at<float>(0, 2) = rigs[5 * r + 2]; Ks[r].at<float>(1, 2) = rigs[5 * r + 3];
      bfs[r] = rigs[5 * r + 4];
    }

    // (mode 3: an integrator that tells the extractors the rig once, after reading the calibration, Tracking.cc:520-640 — the first
    // fused Frame is then matched with the right rig in its one submission instead of being matched again)
    if (mode == 3) mpORBextractorLeft->pliSetStereoCamera(bfs[0], Ks[0].at<float>(0, 0));
    ORBextractor::pliCopyPyramidBack(mode != 4);
    Dump out(argv[2]);
    if (!out.f) { std::perror(argv[2]); return 2; }
    std::vector<uint64_t> hashes;
    double frameSeconds = 0.0;      // wall time of the Frame constructors (extraction x4 + the two stereo matchers), the first two frames left out
    long framesTimed = 0;
    // pose of the last frame (a rotated, shifted world) and three motions: forward (tlc.z > mb), backward, sideways
    const cv::Mat Tlw = pose(0.02f, -0.015f, 0.3f, -0.2f, 0.5f);
    for (int rep = 0; rep < reps; ++rep) {
      uint64_t h = 1469598103934665603ull;
      std::unique_ptr<Frame> last;
      std::vector<std::unique_ptr<MapPoint>> lastPoints;
      for (int i = 0; i < nframes; ++i) {
        const auto tFrame0 = std::chrono::steady_clock::now();
        const int rig = rigChangeFrame >= 0 && i >= rigChangeFrame ? 1 : 0;
        cv::Mat& K = Ks[rig];
        std::unique_ptr<Frame> cur(new Frame(imgs[2 * i], imgs[2 * i + 1], mpORBextractorLeft, mpORBextractorRight, mpLineextractorLeft,
                                             mpLineextractorRight, K, bfs[rig], mode >= 1, (i >= delayFrame && i < delayFrame + delayCount) ? delayMs : 0, mode == 2));
        if (rep > 0 || i >= 2) { frameSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tFrame0).count(); ++framesTimed; }
        Dump* o = rep == 0 ? &out : nullptr;
        const std::string pre = "f" + std::to_string(i) + "/";
        h = record(*cur, o, pre, h);
        if (last && last->N > 0 && cur->N > 0) {
          // Tracking::TrackWithMotionModel (Tracking.cc:3046-3058): points by projection, lines by descriptor
          std::vector<int> matches_12;
          const int nl = match(last->mDescriptors_Line, cur->mDescriptors_Line, 0.9f, matches_12);
          h = fnv(h, matches_12.data(), matches_12.size() * 4);
          if (o) { o->put(pre + "line_matches_12", 'i', (int)matches_12.size(), 1, matches_12.data()); o->put(pre + "line_nmatches", 'i', 1, 1, &nl); }
          typedef PliORBmatcher<Frame, MapPoint> ORBmatcher;
          last->mTcw = Tlw.clone();
          const float dz[4] = {0.15f, -0.15f, 0.01f, 0.15f};
          for (int c = 0; c < 4; ++c) {
            const bool bMono = c == 3;
            // Tcw = T_cl * Tlw with T_cl = a small rotation and the translation -(0.01, 0.004, dz): the current camera
            // sits dz ahead of the last one
            cv::Mat Tcl = pose(0.003f, 0.002f, -0.01f, -0.004f, -dz[c]);
            cv::Mat Tcw = cv::Mat::eye(4, 4, CV_32F);
            cv::Mat Rcw = Tcl.rowRange(0, 3).colRange(0, 3) * Tlw.rowRange(0, 3).colRange(0, 3);
            cv::Mat tcw = Tcl.rowRange(0, 3).colRange(0, 3) * Tlw.rowRange(0, 3).col(3) + Tcl.rowRange(0, 3).col(3);
            for (int r = 0; r < 3; ++r) { for (int q = 0; q < 3; ++q) Tcw.at<float>(r, q) = Rcw.at<float>(r, q); Tcw.at<float>(r, 3) = tcw.at<float>(r); }
            cur->mTcw = Tcw;
            std::fill(cur->mvpMapPoints.begin(), cur->mvpMapPoints.end(), static_cast<MapPoint*>(NULL));
            // the sideways case: every fifth keypoint of the current frame already holds a map point with observations, which makes
            // it unavailable (ORBmatcher.cc:2255-2257)
            static MapPoint observedElsewhere;
            observedElsewhere.nObs = 3;
            if (c == 2)
              for (int j = 0; j < cur->N; j += 5) cur->mvpMapPoints[j] = &observedElsewhere;
            ORBmatcher matcher(0.9f, c != 2);          // the sideways case also runs without the orientation check
            std::map<int, int> match12;
            const int nm = matcher.SearchByProjection(*cur, *last, bMono ? 15.f : 7.f, bMono, match12);
            std::vector<int> pairs;
            for (auto& kv : match12) { pairs.push_back(kv.first); pairs.push_back(kv.second); }
            h = fnv(h, pairs.data(), pairs.size() * 4);
            h = fnv(h, &nm, 4);
            if (o) {
              const std::string cs = pre + "sbp" + std::to_string(c) + "/";
              o->put(cs + "match12", 'i', (int)pairs.size() / 2, 2, pairs.data());
              o->put(cs + "nmatches", 'i', 1, 1, &nm);
              std::vector<float> T(12);
              for (int r = 0; r < 3; ++r) for (int q = 0; q < 4; ++q) T[4 * r + q] = Tcw.at<float>(r, q);
              o->put(cs + "Tcw", 'f', 3, 4, T.data());
            }
          }
          if (o) {
            std::vector<float> T(12);
            for (int r = 0; r < 3; ++r) for (int q = 0; q < 4; ++q) T[4 * r + q] = Tlw.at<float>(r, q);
            o->put(pre + "Tlw", 'f', 3, 4, T.data());
          }
        }
        // the map points of this frame for the next one: Frame::UnprojectStereo (Frame.cc:1333-1347): mRwc * x3Dc + mOw, where the
        // frame's pose will be Tlw
        lastPoints.clear();
        std::fill(cur->mvpMapPoints.begin(), cur->mvpMapPoints.end(), static_cast<MapPoint*>(NULL));   // what the searches above assigned
        {
          const cv::Mat Rcw = Tlw.rowRange(0, 3).colRange(0, 3);
          const cv::Mat tcw = Tlw.rowRange(0, 3).col(3);
          const cv::Mat Rwc = Rcw.t();
          const cv::Mat Ow = -Rcw.t() * tcw;
          const float invfx = 1.0f / cur->fx, invfy = 1.0f / cur->fy;
          for (int j = 0; j < cur->N; ++j) {
            const float z = cur->mvDepth[j];
            if (z > 0) {
              const float u = cur->mvKeysUn[j].pt.x, v = cur->mvKeysUn[j].pt.y;
              cv::Mat x3Dc(3, 1, CV_32F);
              x3Dc.at<float>(0) = (u - cur->cx) * z * invfx; x3Dc.at<float>(1) = (v - cur->cy) * z * invfy; x3Dc.at<float>(2) = z;
              std::unique_ptr<MapPoint> mp(new MapPoint());
              mp->mWorldPos = Rwc * x3Dc + Ow;
              mp->mDescriptor = cur->mDescriptors.row(j).clone();
              mp->nObs = (j % 3 == 0) ? 0 : 2;       // (a third of them like Tracking::UpdateLastFrame's temporal points: no observations)
              cur->mvpMapPoints[j] = mp.get();
              lastPoints.push_back(std::move(mp));
            }
          }
        }
        last = std::move(cur);
      }
      hashes.push_back(h);
    }
    out.put("hashes", 'Q', (int)hashes.size(), 1, hashes.data());
    const double msPerFrame = framesTimed ? frameSeconds / framesTimed * 1e3 : 0.0;
    out.put("frame_ms", 'd', 1, 1, &msPerFrame);
    {
      const pli_detail::FrameFusion::Stats fs = mpORBextractorLeft->pliFusionStats();
      const uint64_t v[6] = {fs.fused, fs.unfusedCalls, fs.timeouts, fs.mismatched, fs.sleeps, fs.missedFrames};
      out.put("fusion_stats", 'Q', 1, 6, v);
      std::printf("dropin_harness: fusion: %llu Frames fused, %llu calls alone, %llu timeouts in %llu Frames, %llu mismatched Frames, %llu sleeps\n",
                  (unsigned long long)v[0], (unsigned long long)v[1], (unsigned long long)v[2], (unsigned long long)v[5], (unsigned long long)v[3],
                  (unsigned long long)v[4]);
    }
    // destruction order as a System shutdown; the registry must end empty (contexts are released)
    delete mpORBextractorLeft; delete mpORBextractorRight; delete mpLineextractorLeft; delete mpLineextractorRight;
    const int left = (int)pli_detail::Registry::get().groups.size();
    out.put("groups_left", 'i', 1, 1, &left);
    std::printf("dropin_harness: %d frames x %d repetitions (%s), hash %016llx, %.3f ms per Frame constructor\n", nframes, reps,
                mode >= 1 ? "four threads" : "sequential", (unsigned long long)hashes[0], msPerFrame);
    if (mode == 0)
      std::printf("  per Frame: ORB x2 %.3f ms, lines x2 %.3f ms, stereo matchers %.3f ms\n", Frame::stageSeconds[0] * 1e3 / (nframes * reps),
                  Frame::stageSeconds[1] * 1e3 / (nframes * reps), Frame::stageSeconds[2] * 1e3 / (nframes * reps));
  } catch (const std::exception& e) {
    std::fprintf(stderr, "dropin_harness: %s\n", e.what());
    return 1;
  }
  return 0;
}
