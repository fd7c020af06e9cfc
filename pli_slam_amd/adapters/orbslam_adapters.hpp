// Drop-in shims with the reference's signatures (build inside the PLI-SLAM tree, where OpenCV 3 exists):
//
//   ORB_SLAM3::ORBextractor   include/ORBextractor.h:46-115   (functor + scale getters + mvImagePyramid)
//   ORB_SLAM3::Lineextractor  include/LineExtractor.h:41-74
//   ORB_SLAM3::match          include/LineMatcher.h:63
//   ORB_SLAM3::ORBmatcher::DescriptorDistance  include/ORBmatcher.h:42
//
// Frame.cc / Tracking.cc keep calling these names unchanged; INTEGRATION.md lists the three edits a
// maintainer makes (swap two headers, share one pli::Frontend between the four extractors).
// The arithmetic is in libpli_frontend.so; this file only converts containers.
#pragma once
#if !__has_include(<opencv2/core/core.hpp>)
#error "orbslam_adapters.hpp needs OpenCV 3 (it is meant to be compiled inside the PLI-SLAM tree)"
#endif
#include <opencv2/core/core.hpp>
#include <memory>
#include "pli_cpp.hpp"
#include "line_descriptor_custom.hpp"   // cv::line_descriptor::KeyLine (Thirdparty/line_descriptor)

namespace ORB_SLAM3 {

// All extractors of one Tracking object share this context (device pyramids feed the stereo matcher).
struct PliShared {
  std::shared_ptr<pli::Frontend> fe;
  static pli_frontend_config makeConfig(int w, int h, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                                        int minThFAST) {
    pli_frontend_config c;
    pli_config_default(&c, w, h);
    c.orb_nfeatures = nfeatures; c.orb_scale_factor = scaleFactor; c.orb_nlevels = nlevels;
    c.orb_ini_th_fast = iniThFAST; c.orb_min_th_fast = minThFAST;
    return c;
  }
};

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
  // `eye`: 0 for mpORBextractorLeft, 1 for mpORBextractorRight (Tracking.cc:743-746)
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, PliShared shared, int eye)
      : mvImagePyramid(nlevels), sh_(shared), eye_(eye), nlevels_(nlevels), scaleFactor_(scaleFactor) {
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) { mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor; mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i]; }
    mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) { mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]; }
    (void)nfeatures; (void)iniThFAST; (void)minThFAST;   // already in the shared context's config
  }
  // int operator()(InputArray image, InputArray mask, vector<KeyPoint>&, OutputArray descriptors, vector<int>& vLappingArea)
  int operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                 cv::OutputArray _descriptors, std::vector<int>& /*vLappingArea = {0,0} on the stereo path*/) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    std::vector<pli_keypoint> kps;
    std::vector<uint8_t> desc;
    int n = sh_.fe->extractORB(eye_, image.data, image.cols, image.rows, (int64_t)image.step, kps, desc);
    if (n < 0) return -1;
    _keypoints.resize(n);
    for (int i = 0; i < n; ++i)
      _keypoints[i] = cv::KeyPoint(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, -1);
    if (n == 0) _descriptors.release();
    else {
      _descriptors.create(n, 32, CV_8U);
      std::memcpy(_descriptors.getMat().data, desc.data(), (size_t)n * 32);
    }
    // public member the stereo matcher and drawers read (ORBextractor.h:87); levels come back without the border
    for (int l = 0; l < nlevels_; ++l) {
      int w = 0, h = 0;
      pli::check(pli_orb_pyramid_level(sh_.fe->handle(), eye_, l, nullptr, 0, &w, &h));
      mvImagePyramid[l].create(h, w, CV_8U);
      pli::check(pli_orb_pyramid_level(sh_.fe->handle(), eye_, l, mvImagePyramid[l].data, (int64_t)w * h, &w, &h));
    }
    return n;
  }
  int inline GetLevels() { return nlevels_; }
  float inline GetScaleFactor() { return scaleFactor_; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }
  std::vector<cv::Mat> mvImagePyramid;
  PliShared& shared() { return sh_; }

 protected:
  PliShared sh_;
  int eye_, nlevels_;
  float scaleFactor_;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

class Lineextractor {
 public:
  Lineextractor(int /*lsd_nfeatures*/, double /*llength_th*/, int /*lsd_refine*/, double /*lsd_scale*/,
                double /*lsd_sigma_scale*/, double /*lsd_quant*/, double /*lsd_ang_th*/, double /*lsd_log_eps*/,
                double /*lsd_density_th*/, int /*lsd_n_bins*/, bool /*bFLD*/, PliShared shared, int eye)
      : sh_(shared), eye_(eye) {}
  // void operator()(const cv::Mat& image, const cv::Mat& mask, vector<KeyLine>& keylines, cv::Mat& descriptors_line)
  void operator()(const cv::Mat& img, const cv::Mat& /*mask*/, std::vector<cv::line_descriptor::KeyLine>& keylines,
                  cv::Mat& descriptors_line) {
    std::vector<pli_keyline> kls;
    std::vector<uint8_t> desc;
    sh_.fe->extractLines(eye_, img.data, img.cols, img.rows, (int64_t)img.step, kls, desc);
    keylines.resize(kls.size());
    for (size_t i = 0; i < kls.size(); ++i) {
      cv::line_descriptor::KeyLine& k = keylines[i];
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

 protected:
  PliShared sh_;
  int eye_;
};

// int match(const cv::Mat& desc1, const cv::Mat& desc2, float nnr, std::vector<int>& matches_12), LineMatcher.h:63
inline int match(pli::Frontend& fe, const cv::Mat& desc1, const cv::Mat& desc2, float nnr, std::vector<int>& matches_12) {
  return fe.matchLines(desc1.data, desc1.rows, desc2.data, desc2.rows, nnr, matches_12);
}

}  // namespace ORB_SLAM3
