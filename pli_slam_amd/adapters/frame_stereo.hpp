// Frame-level replacements, as code: the bodies of
//
//   void Frame::ComputeStereoMatches()                 include/Frame.h:152   src/Frame.cc:976-1154
//   void Frame::ComputeStereoMatches_Lines(bool)       include/Frame.h:154   src/Frame.cc:1156-1259
//
// written against the members of the reference's Frame that those two functions read and write, as templates so that this
// header needs neither Frame.h nor Eigen.  In the PLI-SLAM tree the two member functions become one line each:
//
//   void Frame::ComputeStereoMatches()                    { ORB_SLAM3::pli_frame::ComputeStereoMatches(*this); }
//   void Frame::ComputeStereoMatches_Lines(bool initial)  { ORB_SLAM3::pli_frame::ComputeStereoMatches_Lines(*this, initial); }
//
// Both run on the device context of the Frame's left ORB extractor, on the tables and pyramids the four operator() calls of
// Frame.cc:128-135 left there (the four std::threads may stay: calls on a context are serialised inside the library).
// Members read:  N, mvKeys, mvKeysRight, mvKeys_Line, mvKeysRight_Line, mpORBextractorLeft, mbf, mK
// Members written: mvuRight, mvDepth (N floats each, -1 = no stereo); mvDisparity_l (pair<float,float>, (-1,-1) = mono),
//                  mvle_l (Vector3d(0,0,0) = mono)
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "orbslam_adapters.hpp"

namespace ORB_SLAM3 {
namespace pli_frame {

// The device context that holds this Frame's tables: the one of the left ORB extractor for the size of its level-0 image
// (the reference reads the same object: `mpORBextractorLeft->mvImagePyramid[0].rows`, Frame.cc:983).
template <class FrameT>
inline std::shared_ptr<pli::Frontend> contextOf(FrameT& F) {
  // (with ORBextractor::pliCopyPyramidBack(false) the member stays empty: the extractor remembers the size of its last image)
  int w = 0, h = 0;
  if (F.mpORBextractorLeft) F.mpORBextractorLeft->pliLastImageSize(w, h);
  if (!F.mpORBextractorLeft || w <= 0 || h <= 0)
    throw std::logic_error("pli_frame: the left ORB extractor has not run (Frame.cc:128-135 comes first)");
  return F.mpORBextractorLeft->pliContext(w, h);
}

inline void checkCount(const char* what, long frameSide, int deviceSide) {
  if (frameSide != deviceSide)
    throw std::logic_error(std::string("pli_frame: ") + what + " holds " + std::to_string(frameSide) + " entries but the device table " +
                           std::to_string(deviceSide) + " — another Frame has extracted on this context since (one Frame at a time per context)");
}

// void Frame::ComputeStereoMatches()  — Frame.cc:976-1154
template <class FrameT>
inline void ComputeStereoMatches(FrameT& F) {
  F.mvuRight = std::vector<float>(F.N, -1.0f);          // :978-979
  F.mvDepth = std::vector<float>(F.N, -1.0f);
  if (F.N == 0) return;                                 // the constructor returns before the call (Frame.cc:146); the reference's own body
                                                        // would index an empty vDistIdx (:1141)
  std::shared_ptr<pli::Frontend> fe = contextOf(F);
  int32_t cnt[4];
  fe->lastCounts(cnt);
  checkCount("mvKeys", (long)F.mvKeys.size(), cnt[0]);
  checkCount("mvKeysRight", (long)F.mvKeysRight.size(), cnt[1]);
  checkCount("N", (long)F.N, cnt[0]);
  // minZ = mb, maxD = mbf / minZ (:1005-1008): mb = mbf / fx is assigned after this call in the constructor (:197), so the
  // intended value is taken from the calibration the Frame carries: fx = mK(0,0)
  // (through the extractor's group: every context of the group, present and future, takes the rig; a fused Frame that
  // pli_frame_extract matched with another rig is matched again here, on its resident tables)
  F.mpORBextractorLeft->pliSetStereoCamera(F.mbf, F.mK.template at<float>(0, 0));
  std::vector<float> ur, depth;
  fe->computeStereoMatches(ur, depth);
  for (int i = 0; i < F.N; ++i) { F.mvuRight[i] = ur[i]; F.mvDepth[i] = depth[i]; }
}

// void Frame::ComputeStereoMatches_Lines(bool initial)  — Frame.cc:1156-1259 (doNotDropMonoLines = true, :1158: the line
// containers keep their size, unmatched lines get (-1,-1) and a zero line equation)
template <class FrameT>
inline void ComputeStereoMatches_Lines(FrameT& F, bool /*initial*/ = false) {
  typedef typename std::remove_reference<decltype(F.mvDisparity_l)>::type DispVec;
  typedef typename std::remove_reference<decltype(F.mvle_l)>::type LeVec;
  typedef typename LeVec::value_type Vec3;
  const size_t NL = F.mvKeys_Line.size();
  F.mvDisparity_l.clear();                              // :1162-1167
  F.mvle_l.clear();
  F.mvDisparity_l.resize(NL, typename DispVec::value_type(-1, -1));
  F.mvle_l.resize(NL, Vec3(0, 0, 0));
  if (F.mvKeys_Line.empty() || F.mvKeysRight_Line.empty()) return;     // :1176-1177
  std::shared_ptr<pli::Frontend> fe = contextOf(F);
  int32_t cnt[4];
  fe->lastCounts(cnt);
  checkCount("mvKeys_Line", (long)NL, cnt[2]);
  checkCount("mvKeysRight_Line", (long)F.mvKeysRight_Line.size(), cnt[3]);
  std::vector<float> disp;
  std::vector<double> le;
  fe->computeStereoMatchesLines(disp, le);
  for (size_t i = 0; i < NL; ++i) {
    F.mvDisparity_l[i] = typename DispVec::value_type(disp[2 * i], disp[2 * i + 1]);
    F.mvle_l[i] = Vec3(le[3 * i], le[3 * i + 1], le[3 * i + 2]);
  }
}

}  // namespace pli_frame
}  // namespace ORB_SLAM3
