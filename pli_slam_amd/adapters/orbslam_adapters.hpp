// Drop-in shims with the reference's EXACT signatures (build inside the PLI-SLAM tree, where OpenCV 3 exists):
//
//   ORB_SLAM3::ORBextractor   include/ORBextractor.h:46-115   ctor (nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST),
//                                                             functor, scale getters, mvImagePyramid
//   ORB_SLAM3::Lineextractor  include/LineExtractor.h:41-74   both constructors, functor
//   ORB_SLAM3::match          include/LineMatcher.h:63        match(desc1, desc2, nnr, matches_12)
//   ORB_SLAM3::ORBmatcher::DescriptorDistance                 include/ORBmatcher.h:42  (host inline: 32 bytes never go to the GPU)
//   ORB_SLAM3::ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, match12)   include/ORBmatcher.h:51,
//                                                             src/ORBmatcher.cc:2179-2323 (projection on the host with the
//                                                             reference's own cv::Mat expressions, window search on the GPU)
//
// Frame.cc / Tracking.cc keep calling these names unchanged; INTEGRATION.md lists the edits (swap the headers).
//
// How four independent extractor objects end up on ONE device context (the stereo matchers need both eyes' pyramids and
// tables on the device): extractors register in construction order.  Tracking.cc:87-98,743-749 builds
//   mpORBextractorLeft, mpLineextractorLeft, mpORBextractorRight, mpLineextractorRight, mpIniORBextractor, mpIniLineextractor;
// the second ORBextractor / Lineextractor constructed with the SAME parameters as an existing one becomes the right eye of
// that one's group, different parameters (the 2x-feature initial extractors) open a new group; the k-th ORB group and the
// k-th line group share a context, created lazily at the first operator() call for the size of the image it is given.
#pragma once
#if !__has_include(<opencv2/core/core.hpp>)
#error "orbslam_adapters.hpp needs OpenCV 3 (it is meant to be compiled inside the PLI-SLAM tree)"
#endif
#include <opencv2/core/core.hpp>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <vector>
#include "pli_cpp.hpp"
#ifndef PLI_ADAPTER_NO_KEYLINE_HEADER
#include "line_descriptor_custom.hpp"   // cv::line_descriptor::KeyLine (Thirdparty/line_descriptor)
#endif

namespace ORB_SLAM3 {

namespace pli_detail {

struct OrbParams {
  int nfeatures, nlevels, iniThFAST, minThFAST;
  float scaleFactor;
  bool operator==(const OrbParams& o) const {
    return nfeatures == o.nfeatures && nlevels == o.nlevels && iniThFAST == o.iniThFAST && minThFAST == o.minThFAST && scaleFactor == o.scaleFactor;
  }
};
struct LineParams {
  int lsd_nfeatures, lsd_refine, lsd_n_bins;
  double min_line_length, lsd_scale, lsd_sigma_scale, lsd_quant, lsd_ang_th, lsd_log_eps, lsd_density_th;
  bool bFLD;
  bool operator==(const LineParams& o) const {
    return lsd_nfeatures == o.lsd_nfeatures && lsd_refine == o.lsd_refine && lsd_n_bins == o.lsd_n_bins &&
           min_line_length == o.min_line_length && lsd_scale == o.lsd_scale && lsd_sigma_scale == o.lsd_sigma_scale &&
           lsd_quant == o.lsd_quant && lsd_ang_th == o.lsd_ang_th && lsd_log_eps == o.lsd_log_eps &&
           lsd_density_th == o.lsd_density_th && bFLD == o.bFLD;
  }
};

// Frame fusion.  Frame::Frame calls the four extractors of a group from four threads at the same time (Frame.cc:128-135).  When all
// four calls of a frame are in flight together they are funnelled into ONE submission (pli_frame_extract: both eyes, points and
// lines, and the two stereo matchers on top), and every thread takes its part from the frame's record: 2.7 ms per Frame instead of
// 6.5 ms for four calls that queue behind the context's lock.  The results are the same bytes either way
// (tests/test_cpp_dropin.py compares the two).
//
// A call that waits kWaitMs without its three partners withdraws and takes the per-call path — ONE frame is unfused, the next frame
// tries again (a late thread on a host busy with LocalMapping / LoopClosing / the viewer must not cost every later Frame the fused
// path).  Misses are counted per FRAME, not per call: the timed-out calls of one Frame — three waiters and the late-comer, or the
// four calls of an integrator that calls the extractors one after the other — are one miss (a Frame is over when all four kinds
// have timed out, or a kind times out again).  Only kMaxMisses unfused Frames in a row put the fusion to sleep, for kCoolOff calls
// (32 Frames); then ONE Frame probes, and if it does not fuse either the fusion sleeps again at once, twice as long (up to
// kMaxCoolOff calls): a sequential integrator pays the 2 ms waits on 8 Frames once, then on one Frame in 33, 65, 129, 257.  The four calls of a Frame must
// agree: the line extractors must be given the images (pointer, stride, size) the ORB extractors of the same eye were given, as
// Frame.cc:128-135 does; if they differ (a ROI, a preprocessed copy) nobody is fused and every caller extracts from ITS image.
// stats(): how many Frames went which way.
struct FrameFusion {
  static constexpr int kOrbL = 0, kOrbR = 1, kLineL = 2, kLineR = 3;
  static constexpr int kWaitMs = 2, kMaxMisses = 8, kCoolOff = 128, kMaxCoolOff = 1024;
  // (PLI_FUSION_WAIT_MS: the rendezvous wait for test runs under a sanitizer, where a thread start alone takes milliseconds)
  static int waitMs() {
    static const int ms = [] { const char* e = std::getenv("PLI_FUSION_WAIT_MS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : kWaitMs; }();
    return ms;
  }
  struct Stats { uint64_t fused = 0, unfusedCalls = 0, timeouts = 0, mismatched = 0, sleeps = 0, missedFrames = 0; };
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  int misses = 0;                            // Frames that ended in timeouts since the last fused frame
  int missKinds = 0;                         // kinds (bit per extractor) that have timed out in the Frame being counted
  int asleep = 0;                            // calls left to skip before the next probe
  int coolOff = kCoolOff;                    // length of the next sleep (doubles while the probes keep failing)
  uint64_t gen = 0;                          // frames collected so far (fused or released)
  bool lastFused = false;                    // outcome of generation gen - 1
  const uint8_t* img[4] = {nullptr, nullptr, nullptr, nullptr};
  int64_t stride[4] = {0, 0, 0, 0};
  int iw[4] = {0, 0, 0, 0}, ih[4] = {0, 0, 0, 0};
  // table record of the last fused frame (pli_table_layout).  Every caller of that frame takes a reference under the lock and reads
  // its part outside it: a Frame that (against the protocol) starts before the previous one's callers have copied their parts
  // gets a record of its own, it cannot overwrite theirs
  std::shared_ptr<std::vector<uint8_t>> record;
  typedef std::shared_ptr<const std::vector<uint8_t>> RecordRef;
  std::exception_ptr error;                  // what the fused submission threw (every caller of that frame rethrows it)
  Stats st;

  Stats stats() { std::lock_guard<std::mutex> lk(m); return st; }
  // (under the lock) a Frame ended unfused by timeouts
  void missed() {
    ++st.missedFrames;
    if (++misses >= kMaxMisses) {
      // asleep for coolOff calls; the Frame after that is a probe: one more miss and the fusion sleeps again, twice as long
      misses = kMaxMisses - 1;
      asleep = coolOff;
      coolOff = coolOff * 2 > kMaxCoolOff ? kMaxCoolOff : coolOff * 2;
      ++st.sleeps;
    }
  }

  // true: the frame was extracted in one submission and `record` holds it; false: take the per-call path
  bool join(int kind, pli::Frontend& fe, const uint8_t* data, int w, int h, int64_t strideBytes, RecordRef& rec) {
    std::unique_lock<std::mutex> lk(m);
    if (asleep > 0) { --asleep; ++st.unfusedCalls; return false; }
    if (img[kind] != nullptr) { ++st.unfusedCalls; return false; }     // (a second call of the same kind while a frame is collecting: not a Frame)
    img[kind] = data; stride[kind] = strideBytes; iw[kind] = w; ih[kind] = h;
    const uint64_t myGen = gen;
    if (++arrived == 4) {
      error = nullptr;
      const bool agree = img[kLineL] == img[kOrbL] && img[kLineR] == img[kOrbR] && stride[kLineL] == stride[kOrbL] &&
                         stride[kLineR] == stride[kOrbR] && iw[kOrbL] == iw[kOrbR] && ih[kOrbL] == ih[kOrbR] &&
                         iw[kLineL] == iw[kOrbL] && ih[kLineL] == ih[kOrbL] && iw[kLineR] == iw[kOrbL] && ih[kLineR] == ih[kOrbL];
      if (agree) {
        if (!record || record.use_count() > 1) record = std::make_shared<std::vector<uint8_t>>();     // (recycled once its readers are gone)
        try {
          fe.frameExtract(img[kOrbL], img[kOrbR], iw[kOrbL], ih[kOrbL], stride[kOrbL], stride[kOrbR], *record);
        } catch (...) {
          error = std::current_exception();
        }
        ++st.fused;
        misses = 0; missKinds = 0; coolOff = kCoolOff;
      } else {
        ++st.mismatched;
        st.unfusedCalls += 4;
      }
      lastFused = agree;
      arrived = 0;
      for (int k = 0; k < 4; ++k) img[k] = nullptr;
      ++gen;
      cv.notify_all();
    } else if (!cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(waitMs()), [&] { return gen != myGen; })) {
      // (system_clock: pthread_cond_timedwait, which gcc 11's ThreadSanitizer intercepts; the steady-clock wait is pthread_cond_clockwait)
      img[kind] = nullptr;                   // the partners did not come in time: withdraw, this call goes alone
      --arrived;
      ++st.timeouts; ++st.unfusedCalls;
      // one miss per Frame: this kind has timed out already -> that Frame is over and this call opens the next one; all four
      // kinds have timed out -> the Frame is complete
      if (missKinds & (1 << kind)) { missed(); missKinds = 0; }
      missKinds |= 1 << kind;
      if (missKinds == 15) { missed(); missKinds = 0; }
      return false;
    }
    // (one Frame at a time per group: generation myGen's outcome is read before generation myGen + 1 can complete, because that
    // needs this thread's next call)
    if (!lastFused) return false;
    if (error) std::rethrow_exception(error);
    rec = record;
    return true;
  }
};

// One group = the extractors of one Frame constructor: ORB left/right + LSD left/right on one device context per image size.
// orbMask / lineMask: which eye slots (bit 0 = left, bit 1 = right) are held by a living extractor (atomics: operator() reads
// them while pliBind / a destructor may write them under the registry's lock).
struct Group {
  bool hasOrb = false, hasLine = false;
  OrbParams orb{};
  LineParams line{};
  std::atomic<int> orbMask{0}, lineMask{0};
  FrameFusion fusion;
  std::mutex mu;
  std::map<std::pair<int, int>, std::shared_ptr<pli::Frontend>> ctx;     // by image size
  float rigBf = 0.f, rigFx = 0.f;            // the stereo rig (mbf, fx), once a Frame or pliSetStereoCamera has named it

  std::shared_ptr<pli::Frontend> context(int w, int h) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = ctx.find({w, h});
    if (it != ctx.end()) return it->second;
    pli_frontend_config c;
    pli_config_default(&c, w, h);
    if (rigBf > 0 && rigFx > 0) { c.bf = rigBf; c.fx = rigFx; }
    if (hasOrb) {
      c.orb_nfeatures = orb.nfeatures; c.orb_scale_factor = orb.scaleFactor; c.orb_nlevels = orb.nlevels;
      c.orb_ini_th_fast = orb.iniThFAST; c.orb_min_th_fast = orb.minThFAST;
    }
    if (hasLine) {
      c.lsd_nfeatures = line.lsd_nfeatures; c.lsd_refine = line.lsd_refine; c.lsd_n_bins = line.lsd_n_bins;
      c.min_line_length = line.min_line_length; c.lsd_scale = line.lsd_scale; c.lsd_sigma_scale = line.lsd_sigma_scale;
      c.lsd_quant = line.lsd_quant; c.lsd_ang_th = line.lsd_ang_th; c.lsd_log_eps = line.lsd_log_eps;
      c.lsd_density_th = line.lsd_density_th;
      if (c.lsd_nfeatures > c.max_lines) c.max_lines = c.lsd_nfeatures;
    }
    auto fe = std::make_shared<pli::Frontend>(c);      // throws pli::Error (e.g. lsd_refine != 0, no gfx950 device)
    ctx[{w, h}] = fe;
    return fe;
  }
  // The rig Frame::ComputeStereoMatches works with (mbf, fx = mK(0,0); Frame.cc:1005-1008).  The extractors' constructors do not
  // know it (Tracking.cc:743-746), so a context starts with pli_config_default's EuRoC rig until the first Frame names its own:
  // every context of the group — existing and future — takes it; a fused Frame that was matched with the old rig is matched
  // again on its resident tables by the next ComputeStereoMatches (pli_set_stereo_camera drops the cached result).
  void setRig(float bf, float fx) {
    std::lock_guard<std::mutex> lk(mu);
    if (bf == rigBf && fx == rigFx) return;
    rigBf = bf; rigFx = fx;
    for (auto& kv : ctx) kv.second->setStereoCamera(bf, fx);
  }
};

inline int freeEye(int mask) { return (mask & 1) ? 1 : 0; }

struct Registry {
  std::mutex mu;
  std::vector<std::shared_ptr<Group>> groups;
  static Registry& get() { static Registry r; return r; }
  // the group of the k-th distinct parameter set of its kind; eye = the free slot taken (left first)
  std::shared_ptr<Group> joinOrb(const OrbParams& p, int& eye) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& g : groups)
      if (g->hasOrb && g->orb == p && g->orbMask.load() != 3) { eye = freeEye(g->orbMask.load()); g->orbMask |= 1 << eye; return g; }
    for (auto& g : groups)
      if (!g->hasOrb) { g->hasOrb = true; g->orb = p; eye = 0; g->orbMask = 1; return g; }
    groups.push_back(std::make_shared<Group>());
    auto& g = groups.back();
    g->hasOrb = true; g->orb = p; eye = 0; g->orbMask = 1;
    return g;
  }
  std::shared_ptr<Group> joinLine(const LineParams& p, int& eye) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& g : groups)
      if (g->hasLine && g->line == p && g->lineMask.load() != 3) { eye = freeEye(g->lineMask.load()); g->lineMask |= 1 << eye; return g; }
    for (auto& g : groups)
      if (!g->hasLine) { g->hasLine = true; g->line = p; eye = 0; g->lineMask = 1; return g; }
    groups.push_back(std::make_shared<Group>());
    auto& g = groups.back();
    g->hasLine = true; g->line = p; eye = 0; g->lineMask = 1;
    return g;
  }
  // an extractor dies (or is re-bound): its eye slot is free again; a kind without extractors forgets its parameters, and a
  // group without extractors leaves the registry — its device contexts go with the last shared_ptr (a Frame-level matcher
  // still running on one keeps it alive until it returns).
  void leave(const std::shared_ptr<Group>& g, bool isOrb, int eye) {
    if (!g) return;
    std::lock_guard<std::mutex> lk(mu);
    if (isOrb) { g->orbMask &= ~(1 << eye); if (!g->orbMask.load()) g->hasOrb = false; }
    else { g->lineMask &= ~(1 << eye); if (!g->lineMask.load()) g->hasLine = false; }
    if (!g->orbMask.load() && !g->lineMask.load())
      for (size_t i = 0; i < groups.size(); ++i)
        if (groups[i] == g) { groups.erase(groups.begin() + i); break; }
  }
  void adopt(const std::shared_ptr<Group>& g) {
    std::lock_guard<std::mutex> lk(mu);
    groups.push_back(g);
  }
  // a context for the stateless matchers (any group will do)
  std::shared_ptr<pli::Frontend> any() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& g : groups) {
      std::lock_guard<std::mutex> lk2(g->mu);
      if (!g->ctx.empty()) return g->ctx.begin()->second;
    }
    return nullptr;
  }
};

inline void checkGray(const cv::Mat& m, const char* who) {
  if (m.type() != CV_8UC1) throw std::invalid_argument(std::string(who) + ": the image must be CV_8UC1 (the reference asserts the same)");
}

}  // namespace pli_detail

class ORBextractor;
class Lineextractor;
// (not in the reference) Explicit pairing instead of the construction-order rule above: the four extractors of a stereo
// Tracking (Tracking.cc:87-98,743-749) — or the two of a monocular one (right = nullptr) — move onto ONE fresh group /
// device context.  Call it once after construction when the order or the parameters make the implicit rule ambiguous
// (e.g. initial extractors built with the same parameters as the main ones).
inline void pliBind(ORBextractor* orbLeft, ORBextractor* orbRight, Lineextractor* lineLeft, Lineextractor* lineRight);

class ORBextractor {
  friend void pliBind(ORBextractor*, ORBextractor*, Lineextractor*, Lineextractor*);
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : mvImagePyramid(nlevels), nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST) {
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) { mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor; mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i]; }
    mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) { mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]; }
    group_ = pli_detail::Registry::get().joinOrb({nfeatures, nlevels, iniThFAST, minThFAST, scaleFactor}, eye_);
  }
  ~ORBextractor() { pli_detail::Registry::get().leave(group_, true, eye_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Compute the ORB features and descriptors on an image (mask ignored like the reference; vLappingArea: see below).
  int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors,
                 std::vector<int>& vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    pli_detail::checkGray(image, "ORBextractor");
    std::shared_ptr<pli::Frontend> fe = group_->context(image.cols, image.rows);
    lastW_ = image.cols; lastH_ = image.rows;
    std::vector<pli_keypoint> kps;
    std::vector<uint8_t> desc;
    int n;
    pli_detail::FrameFusion::RecordRef fused;
    if (group_->orbMask.load() == 3 && group_->lineMask.load() == 3 &&
        group_->fusion.join(eye_ ? pli_detail::FrameFusion::kOrbR : pli_detail::FrameFusion::kOrbL, *fe, image.data, image.cols, image.rows,
                            (int64_t)image.step, fused)) {
      // the frame's record: counts, then this eye's keypoint table and descriptors
      const pli_table_layout& Y = fe->layout();
      const uint8_t* rec = fused->data();
      n = reinterpret_cast<const int32_t*>(rec + Y.off_counts)[eye_];
      kps.resize(n);
      desc.resize((size_t)n * 32);
      if (n) {
        std::memcpy(kps.data(), rec + Y.off_kp[eye_], (size_t)n * sizeof(pli_keypoint));
        std::memcpy(desc.data(), rec + Y.off_desc[eye_], (size_t)n * 32);
      }
    } else {
      n = fe->extractORB(eye_, image.data, image.cols, image.rows, (int64_t)image.step, kps, desc);
    }
    if (n < 0) return -1;
    // ORBextractor.cc:1135-1144: keypoints inside [vLappingArea[0], vLappingArea[1]] (level-0 x) go to the back of the
    // arrays (filled from the end), the others to the front in order; the return value is the number of front entries.
    // On the rectified stereo path vLappingArea = {0, 0}, where this is the identity except for keypoints at x == 0
    // (none: the extractor keeps a 16-px border).
    std::vector<int> order(n);
    int mono = 0, stereo = n - 1;
    for (int i = 0; i < n; ++i) {
      const bool lapping = vLappingArea.size() >= 2 && kps[i].x >= vLappingArea[0] && kps[i].x <= vLappingArea[1];
      if (lapping) order[stereo--] = i; else order[mono++] = i;
    }
    _keypoints.resize(n);
    if (n == 0) _descriptors.release();
    else _descriptors.create(n, 32, CV_8U);
    cv::Mat D = n ? _descriptors.getMat() : cv::Mat();
    for (int j = 0; j < n; ++j) {
      const pli_keypoint& k = kps[order[j]];
      _keypoints[j] = cv::KeyPoint(k.x, k.y, k.size, k.angle, k.response, k.octave, -1);
      std::memcpy(D.ptr(j), desc.data() + (size_t)order[j] * 32, 32);
    }
    // public member the stereo matcher and drawers read (ORBextractor.h:87); levels come back without the border.  An integrator
    // that uses adapters/frame_stereo.hpp (the stereo matchers run on the device, on the resident pyramids) and no drawer of the
    // pyramid can switch the copy off — pliCopyPyramidBack(false): 1.3 MB per eye and Frame stay on the device, the member is left empty
    if (pliPyramidFlag().load(std::memory_order_relaxed)) {
      for (int l = 0; l < nlevels; ++l) {
        int w = 0, h = 0;
        pli::check(pli_orb_pyramid_level(fe->handle(), eye_, l, nullptr, 0, &w, &h));
        mvImagePyramid[l].create(h, w, CV_8U);
        pli::check(pli_orb_pyramid_level(fe->handle(), eye_, l, mvImagePyramid[l].data, (int64_t)w * h, &w, &h));
      }
    } else {
      for (int l = 0; l < nlevels; ++l) mvImagePyramid[l].release();
    }
    return mono;
  }
  // (not in the reference) whether operator() fills mvImagePyramid (default: yes, as the reference's ComputePyramid does); process-wide
  static void pliCopyPyramidBack(bool on) { pliPyramidFlag().store(on, std::memory_order_relaxed); }

  int inline GetLevels() { return nlevels; }
  float inline GetScaleFactor() { return scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  std::vector<cv::Mat> mvImagePyramid;

  // (not in the reference) the shared device context for an image size: Frame's stereo matchers run on it
  std::shared_ptr<pli::Frontend> pliContext(int w, int h) { return group_->context(w, h); }
  int pliEye() const { return eye_; }
  // (not in the reference) size of the image of the last operator() call (0 x 0 before the first): what mvImagePyramid[0] says when it is copied back
  void pliLastImageSize(int& w, int& h) const { w = lastW_; h = lastH_; }
  // (not in the reference) the rig of the Frames this extractor serves: mbf and fx (Frame.cc:1005-1008).  Frame::ComputeStereoMatches
  // (adapters/frame_stereo.hpp) calls it for every Frame; an integrator may call it once after reading the calibration
  // (Tracking.cc:620-640) so that even the first fused Frame is matched with the right rig in its one submission.
  void pliSetStereoCamera(float bf, float fx) { group_->setRig(bf, fx); }
  // (not in the reference) Frames fused / calls that went alone / timeouts / mismatched frames / sleeps of this extractor's group
  pli_detail::FrameFusion::Stats pliFusionStats() { return group_->fusion.stats(); }

 protected:
  static std::atomic<bool>& pliPyramidFlag() { static std::atomic<bool> on{true}; return on; }
  int lastW_ = 0, lastH_ = 0;
  int nfeatures;
  double scaleFactor;
  int nlevels, iniThFAST, minThFAST;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::shared_ptr<pli_detail::Group> group_;
  int eye_ = 0;
};

#ifdef PLI_ADAPTER_KEYLINE_TYPE
typedef PLI_ADAPTER_KEYLINE_TYPE PliKeyLine;
#else
typedef cv::line_descriptor::KeyLine PliKeyLine;
#endif

class Lineextractor {
  friend void pliBind(ORBextractor*, ORBextractor*, Lineextractor*, Lineextractor*);
 public:
  Lineextractor(int _lsd_nfeatures, double _llength_th, bool _bFLD = false)
      : Lineextractor(_lsd_nfeatures, _llength_th, 0, 0.8, 0.6, 2.0, 22.5, 1.0, 0.7, 1024, _bFLD) {}   // LSDOptions defaults, LineExtractor.cc:31-48
  Lineextractor(int _lsd_nfeatures, double _llength_th, int _lsd_refine, double _lsd_scale, double _lsd_sigma_scale, double _lsd_quant,
                double _lsd_ang_th, double _lsd_log_eps, double _lsd_density_th, int _lsd_n_bins, bool _bFLD = false)
      : lsd_nfeatures(_lsd_nfeatures), min_line_length(_llength_th), lsd_refine(_lsd_refine), lsd_scale(_lsd_scale),
        lsd_sigma_scale(_lsd_sigma_scale), lsd_quant(_lsd_quant), lsd_ang_th(_lsd_ang_th), lsd_log_eps(_lsd_log_eps),
        lsd_density_th(_lsd_density_th), lsd_n_bins(_lsd_n_bins), bFLD(_bFLD) {
    if (bFLD) throw std::invalid_argument("Lineextractor: the FLD detector is not on the reference path (bFLD = false everywhere)");
    group_ = pli_detail::Registry::get().joinLine({lsd_nfeatures, lsd_refine, lsd_n_bins, min_line_length, lsd_scale, lsd_sigma_scale,
                                                   lsd_quant, lsd_ang_th, lsd_log_eps, lsd_density_th, bFLD}, eye_);
  }
  ~Lineextractor() { pli_detail::Registry::get().leave(group_, false, eye_); }
  Lineextractor(const Lineextractor&) = delete;
  Lineextractor& operator=(const Lineextractor&) = delete;

  void operator()(const cv::Mat& image, const cv::Mat& /*mask*/, std::vector<PliKeyLine>& keylines, cv::Mat& descriptors_line) {
    pli_detail::checkGray(image, "Lineextractor");
    std::shared_ptr<pli::Frontend> fe = group_->context(image.cols, image.rows);
    std::vector<pli_keyline> kls;
    std::vector<uint8_t> desc;
    pli_detail::FrameFusion::RecordRef fused;
    if (group_->orbMask.load() == 3 && group_->lineMask.load() == 3 &&
        group_->fusion.join(eye_ ? pli_detail::FrameFusion::kLineR : pli_detail::FrameFusion::kLineL, *fe, image.data, image.cols, image.rows,
                            (int64_t)image.step, fused)) {
      const pli_table_layout& Y = fe->layout();
      const uint8_t* rec = fused->data();
      const int n = reinterpret_cast<const int32_t*>(rec + Y.off_counts)[2 + eye_];
      kls.resize(n);
      desc.resize((size_t)n * 32);
      if (n) {
        std::memcpy(kls.data(), rec + Y.off_kl[eye_], (size_t)n * sizeof(pli_keyline));
        std::memcpy(desc.data(), rec + Y.off_ldesc[eye_], (size_t)n * 32);
      }
    } else {
      fe->extractLines(eye_, image.data, image.cols, image.rows, (int64_t)image.step, kls, desc);
    }
    keylines.resize(kls.size());
    for (size_t i = 0; i < kls.size(); ++i) {
      PliKeyLine& k = keylines[i];
      const pli_keyline& s = kls[i];
      k.angle = s.angle; k.class_id = s.class_id; k.octave = s.octave; k.pt = cv::Point2f(s.pt_x, s.pt_y);
      k.response = s.response; k.size = s.size;
      k.startPointX = s.startPointX; k.startPointY = s.startPointY; k.endPointX = s.endPointX; k.endPointY = s.endPointY;
      k.sPointInOctaveX = s.sPointInOctaveX; k.sPointInOctaveY = s.sPointInOctaveY;
      k.ePointInOctaveX = s.ePointInOctaveX; k.ePointInOctaveY = s.ePointInOctaveY;
      k.lineLength = s.lineLength; k.numOfPixels = s.numOfPixels;
    }
    if (!kls.empty()) {     // the reference leaves descriptors_line untouched when no line survives
      descriptors_line.create((int)kls.size(), 32, CV_8UC1);
      std::memcpy(descriptors_line.data, desc.data(), desc.size());
    }
  }

  std::shared_ptr<pli::Frontend> pliContext(int w, int h) { return group_->context(w, h); }
  int pliEye() const { return eye_; }

 protected:
  int lsd_nfeatures;
  double min_line_length;
  int lsd_refine;
  double lsd_scale, lsd_sigma_scale, lsd_quant, lsd_ang_th, lsd_log_eps, lsd_density_th;
  int lsd_n_bins;
  bool bFLD;
  std::shared_ptr<pli_detail::Group> group_;
  int eye_ = 0;
};

inline void pliBind(ORBextractor* orbLeft, ORBextractor* orbRight, Lineextractor* lineLeft, Lineextractor* lineRight) {
  using namespace pli_detail;
  if (!orbLeft && !lineLeft) throw std::invalid_argument("pliBind: no left extractor");
  if ((orbRight && !orbLeft) || (lineRight && !lineLeft)) throw std::invalid_argument("pliBind: a right extractor without its left one");
  auto orbParams = [](ORBextractor* e) { return OrbParams{e->nfeatures, e->nlevels, e->iniThFAST, e->minThFAST, (float)e->scaleFactor}; };
  auto lineParams = [](Lineextractor* e) {
    return LineParams{e->lsd_nfeatures, e->lsd_refine, e->lsd_n_bins, e->min_line_length, e->lsd_scale, e->lsd_sigma_scale,
                      e->lsd_quant, e->lsd_ang_th, e->lsd_log_eps, e->lsd_density_th, e->bFLD};
  };
  if (orbRight && !(orbParams(orbLeft) == orbParams(orbRight))) throw std::invalid_argument("pliBind: the two ORB extractors differ");
  if (lineRight && !(lineParams(lineLeft) == lineParams(lineRight))) throw std::invalid_argument("pliBind: the two line extractors differ");
  auto g = std::make_shared<Group>();
  Registry& R = Registry::get();
  if (orbLeft) {
    g->hasOrb = true; g->orb = orbParams(orbLeft);
    R.leave(orbLeft->group_, true, orbLeft->eye_);
    orbLeft->group_ = g; orbLeft->eye_ = 0; g->orbMask |= 1;
    if (orbRight) { R.leave(orbRight->group_, true, orbRight->eye_); orbRight->group_ = g; orbRight->eye_ = 1; g->orbMask |= 2; }
  }
  if (lineLeft) {
    g->hasLine = true; g->line = lineParams(lineLeft);
    R.leave(lineLeft->group_, false, lineLeft->eye_);
    lineLeft->group_ = g; lineLeft->eye_ = 0; g->lineMask |= 1;
    if (lineRight) { R.leave(lineRight->group_, false, lineRight->eye_); lineRight->group_ = g; lineRight->eye_ = 1; g->lineMask |= 2; }
  }
  R.adopt(g);
}

// int match(const cv::Mat& desc1, const cv::Mat& desc2, float nnr, std::vector<int>& matches_12), LineMatcher.h:63 /
// LineMatcher.cpp:201-229 (uses the context of the extractors that produced the descriptors; they exist by then)
inline int match(const cv::Mat& desc1, const cv::Mat& desc2, float nnr, std::vector<int>& matches_12) {
  std::shared_ptr<pli::Frontend> fe = pli_detail::Registry::get().any();
  if (!fe) throw std::logic_error("ORB_SLAM3::match: no extractor has run yet (no device context)");
  matches_12.assign(desc1.rows, -1);
  if (desc1.rows == 0) return 0;
  return fe->matchLines(desc1.data, desc1.rows, desc2.data, desc2.rows, nnr, matches_12);
}

// The parts of ORB_SLAM3::ORBmatcher on the hot path.  Template on the tree's Frame / MapPoint so that this header does
// not need Frame.h; inside the PLI-SLAM tree: `using ORBmatcher = ORB_SLAM3::PliORBmatcher<Frame, MapPoint>;`.
template <class FrameT, class MapPointT>
class PliORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
  PliORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  // ORBmatcher.cc:2495-2511 (bit-twiddling popcount over 8 x 32 bits): the same number as 4 x popcount(64)
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return pli::descriptorDistance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }

  int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
    std::map<int, int> m;
    return SearchByProjection(CurrentFrame, LastFrame, th, bMono, m);
  }

  // ORBmatcher.cc:2179-2323.  The projection (lines 2190-2244) is the reference's own cv::Mat arithmetic, run here on the host;
  // the window search, the "already taken" exclusion, TH_HIGH, the rotation histogram and ComputeThreeMaxima run on the GPU.
  int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono, std::map<int, int>& match12) {
    match12.clear();
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const int N = LastFrame.N;
    std::vector<pli_proj_query> q((size_t)N);
    std::vector<uint8_t> qdesc((size_t)N * 32, 0);
    for (int i = 0; i < N; i++) {
      pli_proj_query& Q = q[i];
      std::memset(&Q, 0, sizeof(Q));
      Q.max_level = -1;
      MapPointT* pMP = LastFrame.mvpMapPoints[i];
      if (!pMP || LastFrame.mvbOutlier[i]) continue;
      cv::Mat x3Dw = pMP->GetWorldPos();
      cv::Mat x3Dc = Rcw * x3Dw + tcw;
      const float xc = x3Dc.at<float>(0);
      const float yc = x3Dc.at<float>(1);
      const float invzc = 1.0 / x3Dc.at<float>(2);
      if (invzc < 0) continue;
      Q.u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
      Q.v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
      const int nLastOctave = LastFrame.mvKeys[i].octave;
      Q.radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
      if (bForward) { Q.min_level = nLastOctave; Q.max_level = -1; }
      else if (bBackward) { Q.min_level = 0; Q.max_level = nLastOctave; }
      else { Q.min_level = nLastOctave - 1; Q.max_level = nLastOctave + 1; }
      Q.ur = Q.u - CurrentFrame.mbf * invzc;
      Q.angle = LastFrame.mvKeysUn[i].angle;
      // (the image-bounds test :2225-2228 is done by the library; a map point without observations — UpdateLastFrame's temporal points in
      // localisation mode — does not make the keypoint it is written to unavailable to the queries behind it, :2255-2257)
      Q.valid = pMP->Observations() > 0 ? 1 : (1 | PLI_PROJ_NO_OBSERVATIONS);
      const cv::Mat dMP = pMP->GetDescriptor();
      std::memcpy(&qdesc[(size_t)i * 32], dMP.ptr<uint8_t>(), 32);
    }
    // the current frame: keypoints, descriptors, mvuRight; keypoints that already hold a map point with observations are
    // not available (:2255-2257): handed over as the `cur_occupied` mask
    const int M = CurrentFrame.N;
    std::vector<pli_keypoint> kp((size_t)M);
    for (int j = 0; j < M; ++j) {
      const cv::KeyPoint& k = CurrentFrame.mvKeysUn[j];
      kp[j].x = k.pt.x; kp[j].y = k.pt.y; kp[j].size = k.size; kp[j].angle = k.angle; kp[j].response = k.response; kp[j].octave = k.octave;
    }
    std::vector<uint8_t> occupied((size_t)M, 0);
    bool anyOccupied = false;
    for (int j = 0; j < M; ++j)
      if (CurrentFrame.mvpMapPoints[j] && CurrentFrame.mvpMapPoints[j]->Observations() > 0) { occupied[j] = 1; anyOccupied = true; }
    std::shared_ptr<pli::Frontend> fe = pli_detail::Registry::get().any();
    if (!fe) throw std::logic_error("SearchByProjection: no extractor has run yet (no device context)");
    std::vector<int> best, raw;
    const int nmatches = fe->searchByProjection(q, qdesc.data(), kp, CurrentFrame.mDescriptors.data, CurrentFrame.mvuRight.data(),
                                                CurrentFrame.mnMinX, CurrentFrame.mnMaxX, CurrentFrame.mnMinY, CurrentFrame.mnMaxY,
                                                mbCheckOrientation, best, anyOccupied ? occupied.data() : nullptr, &raw);
    // the reference's writes, replayed in its order: every match as it was made (:2280-2282: the last writer holds the keypoint,
    // std::map::insert keeps the first pair of a key), then the rotation filter's removals (:2315-2317)
    for (int i = 0; i < N; ++i)
      if (raw[i] >= 0) {
        CurrentFrame.mvpMapPoints[raw[i]] = LastFrame.mvpMapPoints[i];
        match12.insert(std::pair<int, int>(raw[i], i));
      }
    for (int i = 0; i < N; ++i)
      if (raw[i] >= 0 && best[i] < 0) {
        CurrentFrame.mvpMapPoints[raw[i]] = static_cast<MapPointT*>(nullptr);
        match12.erase(raw[i]);
      }
    return nmatches;
  }

 protected:
  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3
