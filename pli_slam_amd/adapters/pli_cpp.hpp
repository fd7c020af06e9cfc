// Header-only C++ host layer over the C ABI (include/pli_frontend.h).
// No OpenCV types here, so it builds anywhere the library does; the shims with
// the reference's exact signatures (cv::Mat / cv::KeyPoint / KeyLine) are in
// orbslam_adapters.hpp and forward to these classes.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/pli_frontend.h"

namespace pli {

struct Error : std::runtime_error {
  pli_status status;
  Error(pli_status s, const std::string& what) : std::runtime_error(what), status(s) {}
};

inline void check(pli_status s) {
  if (s != PLI_OK) throw Error(s, pli_last_error());
}

// ORBmatcher::DescriptorDistance (ORBmatcher.cc:2495-2511) / LineMatcher distance() (LineMatcher.cpp:231-247) for ONE pair of
// 32-byte descriptors, on the host: the reference's bit-twiddling popcount over 8 x 32 bits gives the same number.
inline int descriptorDistance(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 4; ++i) {
    uint64_t x, y;
    std::memcpy(&x, a + 8 * i, 8);
    std::memcpy(&y, b + 8 * i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

// One context = the four extractors of Tracking (ORB left/right, LSD left/right) plus the stereo matchers,
// sharing device-resident pyramids and tables (reference Tracking.cc:87-98,743-749 and Frame.cc:98-228).
// Thread safety: every pli_* call on a context holds the context's own lock inside the library
// (include/pli_frontend.h, "Conventions"), so the methods below may be called from several threads at once —
// the four std::threads of Frame.cc:128-135 — and run one after the other; this class adds no state of its own
// that would need a second lock.
class Frontend {
 public:
  explicit Frontend(const pli_frontend_config& cfg, int device = 0) : cfg_(cfg) {
    check(pli_ctx_create(&cfg_, device, &ctx_));
    check(pli_ctx_layout(ctx_, &layout_));
  }
  ~Frontend() { pli_ctx_destroy(ctx_); }
  Frontend(const Frontend&) = delete;
  Frontend& operator=(const Frontend&) = delete;

  const pli_frontend_config& config() const { return cfg_; }
  const pli_table_layout& layout() const { return layout_; }
  pli_ctx* handle() { return ctx_; }

  // ORBextractor::operator(): returns the reference's return value (count, -1 for an empty image)
  int extractORB(int eye, const uint8_t* img, int w, int h, int64_t stride, std::vector<pli_keypoint>& kps,
                 std::vector<uint8_t>& desc /* n x 32 */) {
    kps.resize(layout_.kp_cap);
    desc.resize((size_t)layout_.kp_cap * 32);
    int32_t n = 0;
    pli_status s = pli_orb_extract(ctx_, eye, img, w, h, stride, kps.data(), layout_.kp_cap, desc.data(), &n);
    if (s == PLI_ERR_EMPTY_IMAGE) { kps.clear(); desc.clear(); return -1; }
    check(s);
    kps.resize(n);
    desc.resize((size_t)n * 32);
    return n;
  }

  // Lineextractor::operator()
  void extractLines(int eye, const uint8_t* img, int w, int h, int64_t stride, std::vector<pli_keyline>& kls,
                    std::vector<uint8_t>& desc) {
    kls.resize(layout_.kl_cap);
    desc.resize((size_t)layout_.kl_cap * 32);
    int32_t n = 0;
    check(pli_line_extract(ctx_, eye, img, w, h, stride, kls.data(), layout_.kl_cap, desc.data(), &n));
    kls.resize(n);
    desc.resize((size_t)n * 32);
  }

  // the whole front-end of one Frame in one submission (pli_frame_extract): `record` = the frame's table record (layout())
  void frameExtract(const uint8_t* left, const uint8_t* right, int w, int h, int64_t strideLeft, int64_t strideRight, std::vector<uint8_t>& record) {
    record.resize((size_t)layout_.record_bytes);
    check(pli_frame_extract(ctx_, left, right, w, h, strideLeft, strideRight, record.data()));
  }
  // sizes of the device tables of the last per-call extractions: mvKeys, mvKeysRight, mvKeys_Line, mvKeysRight_Line (-1: not run)
  void lastCounts(int32_t counts[4]) { check(pli_last_counts(ctx_, counts)); }
  // the rig Frame::ComputeStereoMatches works with: mbf and fx = mK(0,0) (Frame.cc:1005-1008)
  void setStereoCamera(float bf, float fx) { check(pli_set_stereo_camera(ctx_, bf, fx)); }
  // Frame::ComputeStereoMatches
  void computeStereoMatches(std::vector<float>& uRight, std::vector<float>& depth) {
    uRight.assign(layout_.kp_cap, -1.f);
    depth.assign(layout_.kp_cap, -1.f);
    check(pli_stereo_match_points(ctx_, uRight.data(), depth.data(), layout_.kp_cap));
  }
  // Frame::ComputeStereoMatches_Lines
  void computeStereoMatchesLines(std::vector<float>& disp /* n x 2 */, std::vector<double>& le /* n x 3 */) {
    disp.assign((size_t)layout_.kl_cap * 2, -1.f);
    le.assign((size_t)layout_.kl_cap * 3, 0.0);
    check(pli_stereo_match_lines(ctx_, disp.data(), le.data(), layout_.kl_cap));
  }
  // match(desc1, desc2, nnr, matches_12), LineMatcher.cpp:201
  int matchLines(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, std::vector<int>& m12) {
    m12.assign(n1, -1);
    int32_t n = 0;
    check(pli_match_lines(ctx_, d1, n1, d2, n2, nnr, m12.data(), &n));
    return n;
  }
  // ORBmatcher::DescriptorDistance for n pairs at once (one pair: pli::descriptorDistance on the host)
  void descriptorDistances(const uint8_t* a, const uint8_t* b, int n, std::vector<int>& dist) {
    dist.assign(n, 0);
    if (n) check(pli_descriptor_distance(ctx_, a, b, n, dist.data()));
  }
  // core of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, ...)
  int searchByProjection(const std::vector<pli_proj_query>& q, const uint8_t* qdesc, const std::vector<pli_keypoint>& cur,
                         const uint8_t* curDesc, const float* curURight, float minX, float maxX, float minY, float maxY,
                         bool checkOrientation, std::vector<int>& bestIdx2, const uint8_t* curOccupied = nullptr,
                         std::vector<int>* rawIdx2 = nullptr) {
    bestIdx2.assign(q.size(), -1);
    if (rawIdx2) rawIdx2->assign(q.size(), -1);
    int32_t n = 0;
    check(pli_search_by_projection(ctx_, q.data(), qdesc, (int)q.size(), cur.data(), curDesc, curURight, curOccupied, (int)cur.size(),
                                   minX, maxX, minY, maxY, checkOrientation ? 1 : 0, bestIdx2.data(),
                                   rawIdx2 ? rawIdx2->data() : nullptr, &n));
    return n;
  }
  // Frame::ComputeStereoFromRGBD(imDepth) Frame.cc:1309 (depth: CV_32F, row stride in floats)
  void computeStereoFromRGBD(const float* depth, int64_t strideFloats, std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
    mvuRight.assign(layout_.kp_cap, -1.f); mvDepth.assign(layout_.kp_cap, -1.f);
    check(pli_stereo_from_depth(ctx_, depth, strideFloats, mvuRight.data(), mvDepth.data(), layout_.kp_cap));
  }
  // cv::remap(im, imRect, M1, M2, INTER_LINEAR) of the stereo driver (stereo_euroc.cc:166), fused into the ingest
  void setRectifyMaps(int eye, const float* mapx, const float* mapy) { check(pli_set_rectify_maps(ctx_, eye, mapx, mapy)); }
  // core of ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) ORBmatcher.cc:44
  int searchLocalMap(const std::vector<pli_proj_query>& q, const uint8_t* qdesc, const std::vector<pli_keypoint>& cur,
                     const uint8_t* curDesc, const float* curURight, const uint8_t* curOccupied, float minX, float maxX,
                     float minY, float maxY, float nnratio, std::vector<int>& bestIdx2) {
    bestIdx2.assign(q.size(), -1);
    int32_t n = 0;
    check(pli_search_local_map(ctx_, q.data(), qdesc, (int)q.size(), cur.data(), curDesc, curURight, curOccupied,
                               (int)cur.size(), minX, maxX, minY, maxY, nnratio, bestIdx2.data(), &n));
    return n;
  }
  // the same for a frame of two fisheye cameras (F.Nleft != -1), ORBmatcher.cc:44-214: both halves of F.mvpMapPoints
  // (mpLeft[k] / mpRight[k] = index into vpMapPoints of the point the call left in slot k / k + Nleft, or -1)
  int searchLocalMapFishEye(const std::vector<pli_proj_query>& qLeft, const std::vector<pli_proj_query>& qRight, const uint8_t* qdesc,
                            const std::vector<pli_keypoint>& kpLeft, const uint8_t* descLeft, const uint8_t* occLeft,
                            const std::vector<int>& leftToRight, const std::vector<pli_keypoint>& kpRight,
                            const uint8_t* descRight, const uint8_t* occRight, const std::vector<int>& rightToLeft, float minX,
                            float maxX, float minY, float maxY, float nnratio, std::vector<int>& mpLeft, std::vector<int>& mpRight) {
    mpLeft.assign(kpLeft.size(), -1); mpRight.assign(kpRight.size(), -1);
    int32_t n = 0;
    check(pli_search_local_map_fisheye(ctx_, qLeft.data(), qRight.data(), qdesc, (int)qLeft.size(), kpLeft.data(), descLeft, occLeft,
                                       leftToRight.data(), (int)kpLeft.size(), kpRight.data(), descRight, occRight,
                                       rightToLeft.data(), (int)kpRight.size(), minX, maxX, minY, maxY, nnratio, mpLeft.data(),
                                       mpRight.data(), &n));
    return n;
  }
  // int match(const vector<MapLine*>&, Frame&, nnr, matches_12) LineMatcher.cpp:161 on the descriptor tables
  int matchNNR(const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float nnr, std::vector<int>& matches12) {
    matches12.assign(n1, -1);
    int32_t n = 0;
    check(pli_match_nnr(ctx_, desc1, n1, desc2, n2, nnr, matches12.data(), &n));
    return n;
  }
  // Frame::ComputeStereoFishEyeMatches Frame.cc:1577-1618 on the Frame's own tables (mvKeys / mDescriptors / monoLeft, ...);
  // fills mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, mvStereo3Dpoints (x, y, z per left keypoint) and returns nMatches
  int stereoFishEye(const std::vector<pli_keypoint>& kpLeft, const uint8_t* descLeft, int monoLeft,
                    const std::vector<pli_keypoint>& kpRight, const uint8_t* descRight, int monoRight, const pli_kb8_camera& cam1,
                    const pli_kb8_camera& cam2, const float* Rlr, const float* tlr, std::vector<int>& leftToRight,
                    std::vector<int>& rightToLeft, std::vector<float>& depth, std::vector<float>& points3d) {
    leftToRight.assign(kpLeft.size(), -1);
    rightToLeft.assign(kpRight.size(), -1);
    depth.assign(kpLeft.size(), -1.0f);
    points3d.assign(kpLeft.size() * 3, 0.0f);
    int32_t n = 0;
    check(pli_stereo_fisheye_tables(ctx_, kpLeft.data(), descLeft, (int)kpLeft.size(), monoLeft, kpRight.data(), descRight,
                                    (int)kpRight.size(), monoRight, &cam1, &cam2, Rlr, tlr, leftToRight.data(), rightToLeft.data(),
                                    depth.data(), points3d.data(), &n));
    return n;
  }

 private:
  pli_frontend_config cfg_;
  pli_table_layout layout_{};
  pli_ctx* ctx_ = nullptr;
};

// DBoW2::TemplatedVocabulary<cv::Mat, FORB> as Frame::ComputeBoW uses it (Frame.cc:858-870): the descents run on the
// device (pli_bow_transform); BowVector::addWeight / FeatureVector::addFeature in feature order and the L1
// normalisation in word order are the reference's own host arithmetic (TemplatedVocabulary.h:1139-1208,
// BowVector.cpp:33-81), so the two maps are identical to DBoW2's.
using BowVector = std::map<unsigned, double>;
using FeatureVector = std::map<unsigned, std::vector<unsigned>>;

class Vocabulary {
 public:
  // node list as loadFromTextFile reads it from ORBvoc.txt: node i+1 has parent[i], isLeaf[i], desc[32*i..], weight[i]
  Vocabulary(Frontend& fe, int k, int L, int nnodes, const int32_t* parent, const uint8_t* isLeaf, const uint8_t* desc,
             const double* weight) : fe_(fe) {
    check(pli_vocab_create(fe.handle(), k, L, nnodes, parent, isLeaf, desc, weight, &v_));
  }
  ~Vocabulary() { pli_vocab_destroy(v_); }
  Vocabulary(const Vocabulary&) = delete;
  Vocabulary& operator=(const Vocabulary&) = delete;

  // transform(features, v, fv, levelsup) for TF-IDF weighting and L1 scoring (the ORB / LBD vocabularies)
  void transform(const uint8_t* desc, int n, BowVector& v, FeatureVector& fv, int levelsup) const {
    v.clear();
    fv.clear();
    std::vector<int32_t> word(n), node(n);
    std::vector<double> w(n);
    check(pli_bow_transform(fe_.handle(), v_, desc, n, levelsup, word.data(), w.data(), node.data()));
    for (int i = 0; i < n; ++i)
      if (w[i] > 0) {                       // not a stopped word
        v[(unsigned)word[i]] += w[i];
        fv[(unsigned)node[i]].push_back((unsigned)i);
      }
    double norm = 0.0;
    for (auto& kv : v) norm += std::fabs(kv.second);
    if (norm > 0.0)
      for (auto& kv : v) kv.second /= norm;
  }

 private:
  Frontend& fe_;
  pli_vocab* v_ = nullptr;
};

}  // namespace pli
