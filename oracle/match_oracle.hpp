// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// CPU restatement of the reference matchers on the per-frame path:
//   ORBmatcher::DescriptorDistance            src/ORBmatcher.cc:2495-2511
//   distance()                                src/LineMatcher.cpp:231-247
//   Frame::ComputeStereoMatches               src/Frame.cc:976-1154
//   Frame::ComputeStereoMatches_Lines         src/Frame.cc:1156-1307
//   matchGrid(lines)                          src/LineMatcher.cpp:317-396
//   GridStructure / getLineCoords / LineIterator  src/gridStructure.cpp, src/LineIterator.cpp
//   matchNNR / match(desc,desc)               src/LineMatcher.cpp:139-159,201-229
//   ORBmatcher::SearchByProjection(F,F)       src/ORBmatcher.cc:2179-2323 (+ Frame.cc:451-482,774-855, ORBmatcher.cc:2449-2490)
// The Bresenham cell walk and the grid window query are pinned against the
// reference's own src/LineIterator.cpp + src/gridStructure.cpp (the only two
// path files that build without OpenCV): oracle/_ref + tests/golden/grid_*.json.
// Everything else: PARITY UNPINNED (no reference tests; not buildable here).
#pragma once
#include "ocv_prims.hpp"
#include "../include/pli_frontend.h"
#include <climits>
#include <map>
#include <set>
#include <unordered_set>
#include <list>

namespace orc {

static inline int descriptorDistance(const uint8_t* a, const uint8_t* b) {
  const int32_t* pa = (const int32_t*)a;
  const int32_t* pb = (const int32_t*)b;
  int dist = 0;
  for (int i = 0; i < 8; i++, pa++, pb++) {
    unsigned int v = *pa ^ *pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// ---------------------------------------------------------------------------
// Frame::ComputeStereoMatches, Frame.cc:976-1154
// `maxD`: the reference reads mb before it is set; the
// oracle takes it as a parameter (fx as intended, or +inf).
// ---------------------------------------------------------------------------
static inline void computeStereoMatches(const std::vector<pli_keypoint>& mvKeys, const uint8_t* mDescriptors,
                                        const std::vector<pli_keypoint>& mvKeysRight,
                                        const uint8_t* mDescriptorsRight, const std::vector<Img8>& pyrL,
                                        const std::vector<Img8>& pyrR, const std::vector<float>& mvScaleFactors,
                                        const std::vector<float>& mvInvScaleFactors, float mbf, float maxD,
                                        std::vector<float>& mvuRight, std::vector<float>& mvDepth,
                                        std::vector<int>* dbgBestIdx = nullptr, std::vector<int>* dbgSad = nullptr) {
  const int N = (int)mvKeys.size();
  mvuRight.assign(N, -1.0f);
  mvDepth.assign(N, -1.0f);
  if (dbgBestIdx) dbgBestIdx->assign(N, -1);
  if (dbgSad) dbgSad->assign(N, -1);
  const int TH_HIGH = 100, TH_LOW = 50;
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = pyrL[0].h;
  std::vector<std::vector<size_t>> vRowIndices(nRows);
  const int Nr = (int)mvKeysRight.size();
  for (int iR = 0; iR < Nr; iR++) {
    const pli_keypoint& kp = mvKeysRight[iR];
    const float kpY = kp.y;
    const float r = 2.0f * mvScaleFactors[kp.octave];
    const int maxr = (int)std::ceil(kpY + r);
    const int minr = (int)std::floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);   // reference indexes unchecked (kp >= 16 px inside)
  }
  const float minD = 0;
  std::vector<std::pair<int, int>> vDistIdx;
  for (int iL = 0; iL < N; iL++) {
    const pli_keypoint& kpL = mvKeys[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y;
    const float uL = kpL.x;
    const std::vector<size_t>& vCandidates = vRowIndices[(int)vL];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD;
    const float maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    const uint8_t* dL = mDescriptors + (size_t)iL * 32;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const size_t iR = vCandidates[iC];
      const pli_keypoint& kpR = mvKeysRight[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = descriptorDistance(dL, mDescriptorsRight + iR * 32);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      if (dbgBestIdx) (*dbgBestIdx)[iL] = (int)bestIdxR;
      const float uR0 = mvKeysRight[bestIdxR].x;
      const float scaleFactor = mvInvScaleFactors[kpL.octave];
      const float scaleduL = std::round(kpL.x * scaleFactor);
      const float scaledvL = std::round(kpL.y * scaleFactor);
      const float scaleduR0 = std::round(uR0 * scaleFactor);
      const int w = 5;
      const Img8& imL = pyrL[kpL.octave];
      const Img8& imR = pyrR[kpL.octave];
      const int cy = (int)scaledvL, cxl = (int)scaleduL, cxr = (int)scaleduR0;
      // cv::Mat::rowRange/colRange assert the window is inside the level image; the
      // reference aborts there.  Oracle (and kernels): such a keypoint gets no stereo.
      if (cy - w < 0 || cy + w + 1 > imL.h || cxl - w < 0 || cxl + w + 1 > imL.w) continue;
      int bestDistS = INT_MAX;
      int bestincR = 0;
      const int L = 5;
      float vDists[2 * 5 + 1];
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= imR.w) continue;
      if (cxr - L - w < 0) continue;   // same: colRange would assert
      const int cL = imL.at(cy, cxl);
      for (int incR = -L; incR <= +L; incR++) {
        const int cR = imR.at(cy, cxr + incR);
        int sad = 0;
        for (int dy = -w; dy <= w; ++dy)
          for (int dx = -w; dx <= w; ++dx) {
            int a = (int)imL.at(cy + dy, cxl + dx) - cL;
            int b = (int)imR.at(cy + dy, cxr + incR + dx) - cR;
            sad += std::abs(a - b);
          }
        float dist = (float)sad;
        if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
          disparity = 0.01f;
          bestuR = (float)((double)uL - 0.01);   // `bestuR = uL-0.01;` is evaluated in double
        }
        mvDepth[iL] = mbf / disparity;
        mvuRight[iL] = bestuR;
        vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
        if (dbgSad) (*dbgSad)[iL] = bestDistS;
      }
    }
  }
  if (vDistIdx.empty()) return;   // reference: vDistIdx[0] on an empty vector (UB); defined as "nothing to cut"
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist) break;
    mvuRight[vDistIdx[i].second] = -1;
    mvDepth[vDistIdx[i].second] = -1;
  }
}

// ---------------------------------------------------------------------------
// LineIterator (src/LineIterator.cpp:31-77) + getLineCoords (gridStructure.cpp:33-42)
// ---------------------------------------------------------------------------
static inline void getLineCoords(double x1_, double y1_, double x2_, double y2_,
                                 std::vector<std::pair<int, int>>& line_coords) {
  line_coords.clear();
  double x1 = x1_, y1 = y1_, x2 = x2_, y2 = y2_;
  const bool steep = std::abs(y2_ - y1_) > std::abs(x2_ - x1_);
  if (steep) { std::swap(x1, y1); std::swap(x2, y2); }
  if (x1 > x2) { std::swap(x1, x2); std::swap(y1, y2); }
  double dx = x2 - x1;
  double dy = std::abs(y2 - y1);
  double error = dx / 2.0;
  int ystep = (y1 < y2) ? 1 : -1;
  int x = static_cast<int>(x1);
  int y = static_cast<int>(y1);
  int maxX = static_cast<int>(x2);
  while (x <= maxX) {
    if (steep) line_coords.push_back(std::make_pair(y, x));
    else line_coords.push_back(std::make_pair(x, y));
    error -= dy;
    if (error < 0) { y += ystep; error += dx; }
    x++;
  }
}

// GridStructure (gridStructure.cpp:44-83): grid[x][y] lists, rows=48, cols=64.
struct Grid {
  int rows, cols;
  std::vector<std::vector<int>> cell;
  Grid(int r, int c) : rows(r), cols(c), cell((size_t)r * c) {}
  void push(int x, int y, int idx) {
    if (x >= 0 && x < cols && y >= 0 && y < rows) cell[(size_t)x * rows + y].push_back(idx);
  }
  void get(int x, int y, int wl, int wr, int hu, int hd, std::set<int>& indices) const {
    int min_x = std::max(0, x - wl);
    int max_x = std::min(cols, x + wr + 1);
    int min_y = std::max(0, y - hu);
    int max_y = std::min(rows, y + hd + 1);
    for (int x_ = min_x; x_ < max_x; ++x_)
      for (int y_ = min_y; y_ < max_y; ++y_) {
        const std::vector<int>& c = cell[(size_t)x_ * rows + y_];
        indices.insert(c.begin(), c.end());
      }
  }
};

struct LineMatchCfg {
  int matching_s_ws = 10;
  bool best_lr_matches = true;
  double line_sim_th = 0.75, stereo_overlap_th = 0.75, min_ratio_12_l = 0.9, ls_min_disp_ratio = 0.7,
         min_disp = 1.0, line_horiz_th = 0.1;
};

static inline void normalize2(double& a, double& b) {
  double magnitude = std::sqrt(a * a + b * b);
  a /= magnitude;
  b /= magnitude;
}

// matchGrid(lines), LineMatcher.cpp:317-396.  Candidates are iterated in
// ascending index order (the reference iterates an unordered_set; the result
// does not depend on the order: ties at the best distance are rejected).
static inline int matchGridLines(const std::vector<int>& sx, const std::vector<int>& sy, const std::vector<int>& ex,
                                 const std::vector<int>& ey, const uint8_t* desc1, int n1, const Grid& grid,
                                 const uint8_t* desc2, int n2, const std::vector<std::pair<double, double>>& dir2,
                                 const LineMatchCfg& C, std::vector<int>& matches_12) {
  int matches = 0;
  matches_12.assign(n1, -1);
  std::vector<int> matches_21, distances;
  if (C.best_lr_matches) {
    matches_21.assign(n2, -1);
    distances.assign(n2, INT_MAX);
  }
  for (int i1 = 0; i1 < n1; ++i1) {
    int best_d = INT_MAX, best_d2 = INT_MAX, best_idx = -1;
    double vx = (double)(ex[i1] - sx[i1]), vy = (double)(ey[i1] - sy[i1]);
    normalize2(vx, vy);
    std::set<int> candidates;
    grid.get(sx[i1], sy[i1], C.matching_s_ws, 0, 0, 0, candidates);
    grid.get(ex[i1], ey[i1], C.matching_s_ws, 0, 0, 0, candidates);
    if (candidates.empty()) continue;
    for (int i2 : candidates) {
      if (i2 < 0 || i2 >= n2) continue;
      if (std::abs(vx * dir2[i2].first + vy * dir2[i2].second) < C.line_sim_th) continue;
      const int d = descriptorDistance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
      if (C.best_lr_matches) {
        if (d < distances[i2]) { distances[i2] = d; matches_21[i2] = i1; }
        else continue;
      }
      if (d < best_d) { best_d2 = best_d; best_d = d; best_idx = i2; }
      else if (d < best_d2) best_d2 = d;
    }
    if (best_d < best_d2 * C.min_ratio_12_l) { matches_12[i1] = best_idx; matches++; }
  }
  if (C.best_lr_matches) {
    for (int i1 = 0; i1 < n1; ++i1) {
      int& i2 = matches_12[i1];
      if (i2 >= 0 && matches_21[i2] != i1) { i2 = -1; matches--; }
    }
  }
  return matches;
}

// Frame::lineSegmentOverlapStereo, Frame.cc:1261-1295
static inline double lineSegmentOverlapStereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj,
                                              double lineHorizTh) {
  double overlap = 1.f;
  if (std::fabs(epl_obs - spl_obs) > lineHorizTh) {
    double sln = std::min(spl_obs, epl_obs);
    double eln = std::max(spl_obs, epl_obs);
    double spn = std::min(spl_proj, epl_proj);
    double epn = std::max(spl_proj, epl_proj);
    double length = eln - spn;
    if ((epn < sln) || (spn > eln)) overlap = 0.f;
    else {
      if ((epn > eln) && (spn < sln)) overlap = eln - sln;
      else overlap = std::min(eln, epn) - std::max(sln, spn);
    }
    if (length > 0.01f) overlap = overlap / length;
    else overlap = 0.f;
    if (overlap > 1.f) overlap = 1.f;
  }
  return overlap;
}

// Frame::ComputeStereoMatches_Lines, Frame.cc:1156-1259 (doNotDropMonoLines = true)
static inline void computeStereoMatchesLines(const std::vector<pli_keyline>& KL, const uint8_t* descL,
                                             const std::vector<pli_keyline>& KR, const uint8_t* descR, int imgW,
                                             int imgH, const LineMatchCfg& C, std::vector<float>& disp /*n x 2*/,
                                             std::vector<double>& le /* n x 3 */, std::vector<int>* dbgMatches = nullptr) {
  const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;
  const int n1 = (int)KL.size(), n2 = (int)KR.size();
  disp.assign((size_t)n1 * 2, -1.f);
  le.assign((size_t)n1 * 3, 0.0);
  if (dbgMatches) dbgMatches->assign(n1, -1);
  if (KL.empty() || KR.empty()) return;
  const double inv_width = FRAME_GRID_COLS / static_cast<double>(imgW);
  const double inv_height = FRAME_GRID_ROWS / static_cast<double>(imgH);
  std::vector<int> sx(n1), sy(n1), ex(n1), ey(n1);
  for (int i = 0; i < n1; ++i) {
    sx[i] = (int)(KL[i].startPointX * inv_width);
    sy[i] = (int)(KL[i].startPointY * inv_height);
    ex[i] = (int)(KL[i].endPointX * inv_width);
    ey[i] = (int)(KL[i].endPointY * inv_height);
  }
  Grid grid(FRAME_GRID_ROWS, FRAME_GRID_COLS);
  std::vector<std::pair<double, double>> directions(n2);
  std::vector<std::pair<int, int>> line_coords;
  for (int idx = 0; idx < n2; ++idx) {
    const pli_keyline& kl = KR[idx];
    double vx = (kl.endPointX - kl.startPointX) * inv_width, vy = (kl.endPointY - kl.startPointY) * inv_height;
    normalize2(vx, vy);
    directions[idx] = std::make_pair(vx, vy);
    getLineCoords(kl.startPointX * inv_width, kl.startPointY * inv_height, kl.endPointX * inv_width,
                  kl.endPointY * inv_height, line_coords);
    for (const std::pair<int, int>& p : line_coords) grid.push(p.first, p.second, idx);
  }
  std::vector<int> matches_12;
  matchGridLines(sx, sy, ex, ey, descL, n1, grid, descR, n2, directions, C, matches_12);
  if (dbgMatches) *dbgMatches = matches_12;
  for (int i1 = 0; i1 < n1; ++i1) {
    const int i2 = matches_12[i1];
    if (i2 < 0) continue;
    double sp_l[3] = {KL[i1].startPointX, KL[i1].startPointY, 1.0};
    double ep_l[3] = {KL[i1].endPointX, KL[i1].endPointY, 1.0};
    double le_l[3] = {sp_l[1] * ep_l[2] - sp_l[2] * ep_l[1], sp_l[2] * ep_l[0] - sp_l[0] * ep_l[2],
                      sp_l[0] * ep_l[1] - sp_l[1] * ep_l[0]};
    double nrm = std::sqrt(le_l[0] * le_l[0] + le_l[1] * le_l[1]);
    le_l[0] = le_l[0] / nrm; le_l[1] = le_l[1] / nrm; le_l[2] = le_l[2] / nrm;
    double sp_r[3] = {KR[i2].startPointX, KR[i2].startPointY, 1.0};
    double ep_r[3] = {KR[i2].endPointX, KR[i2].endPointY, 1.0};
    double overlap = lineSegmentOverlapStereo(sp_l[1], ep_l[1], sp_r[1], ep_r[1], C.line_horiz_th);
    // sp_r << (...), sp_l(1), 1.0 : Eigen's comma initialiser evaluates all three
    // expressions from the OLD sp_r before assigning, then ep_r uses the NEW sp_r.
    double nsx = (sp_r[0] * (sp_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - sp_l[1])) / (sp_r[1] - ep_r[1]);
    sp_r[0] = nsx; sp_r[1] = sp_l[1];
    double nex = (sp_r[0] * (ep_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - ep_l[1])) / (sp_r[1] - ep_r[1]);
    ep_r[0] = nex; ep_r[1] = ep_l[1];
    // filterLineSegmentDisparity, Frame.cc:1297-1307
    double disp_s = sp_l[0] - sp_r[0];
    double disp_e = ep_l[0] - ep_r[0];
    if (std::min(disp_s, disp_e) / std::max(disp_s, disp_e) < C.ls_min_disp_ratio) { disp_s = -1.0; disp_e = -1.0; }
    if (disp_s >= C.min_disp && disp_e >= C.min_disp && std::abs(sp_l[1] - ep_l[1]) > C.line_horiz_th &&
        std::abs(sp_r[1] - ep_r[1]) > C.line_horiz_th && overlap > C.stereo_overlap_th) {
      disp[(size_t)i1 * 2] = (float)disp_s;
      disp[(size_t)i1 * 2 + 1] = (float)disp_e;
      le[(size_t)i1 * 3] = le_l[0];
      le[(size_t)i1 * 3 + 1] = le_l[1];
      le[(size_t)i1 * 3 + 2] = le_l[2];
    }
  }
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(k=2): two smallest, ties -> lower train index.
static inline void knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, std::vector<int>& idx,
                        std::vector<int>& dist) {
  idx.assign((size_t)nq * 2, -1);
  dist.assign((size_t)nq * 2, INT_MAX);
  for (int i = 0; i < nq; ++i) {
    int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
    for (int j = 0; j < nt; ++j) {
      int d = descriptorDistance(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
      else if (d < b1) { b1 = d; i1 = j; }
    }
    idx[2 * i] = i0; idx[2 * i + 1] = i1; dist[2 * i] = b0; dist[2 * i + 1] = b1;
  }
}

// matchNNR, LineMatcher.cpp:139-159.  The reference dereferences matches_[idx][1]
// unconditionally: with fewer than 2 train rows it is UB; defined here as "no match".
static inline int matchNNR(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, std::vector<int>& m12) {
  int matches = 0;
  m12.assign(n1, -1);
  if (n2 < 2) return 0;
  std::vector<int> idx, dist;
  knn2(d1, n1, d2, n2, idx, dist);
  for (int i = 0; i < n1; ++i) {
    if ((float)dist[2 * i] < (float)dist[2 * i + 1] * nnr) { m12[i] = idx[2 * i]; matches++; }
  }
  return matches;
}

// match(desc1, desc2, nnr, matches_12), LineMatcher.cpp:201-229
static inline int matchLines(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, bool bestLR,
                             std::vector<int>& m12) {
  if (bestLR) {
    std::vector<int> m21;
    int matches = matchNNR(d1, n1, d2, n2, nnr, m12);
    matchNNR(d2, n2, d1, n1, nnr, m21);
    for (int i1 = 0; i1 < n1; ++i1) {
      int& i2 = m12[i1];
      if (i2 >= 0 && m21[i2] != i1) { i2 = -1; matches--; }
    }
    return matches;
  }
  return matchNNR(d1, n1, d2, n2, nnr, m12);
}

// ---------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, match12)
// ORBmatcher.cc:2179-2323, with the SLAM state reduced to per-query records
// (see pli_proj_query in include/pli_frontend.h).
// ---------------------------------------------------------------------------
static inline int searchByProjection(const pli_proj_query* q, const uint8_t* qdesc, int nq,
                                     const pli_keypoint* kp, const uint8_t* desc, const float* uright, int ncur,
                                     float mnMinX, float mnMaxX, float mnMinY, float mnMaxY, bool checkOri,
                                     std::vector<int>& best_idx2, const uint8_t* occupied = nullptr,
                                     std::vector<int>* raw_idx2 = nullptr) {
  // occupied[i2]: CurrentFrame.mvpMapPoints[i2] holds a map point with Observations() > 0 before the call (:2255-2257);
  // q[i].valid bit 1 (PLI_PROJ_NO_OBSERVATIONS): LastFrame's map point i has no observations, so the keypoint it is written to
  // (:2280) stays available to the queries behind it.  best_idx2 = the matches after the rotation filter, raw_idx2 before it.
  const int COLS = 64, ROWS = 48, HISTO_LENGTH = 30, TH_HIGH = 100;
  best_idx2.assign(nq, -1);
  const float gwInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);
  const float ghInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
  // AssignFeaturesToGrid, Frame.cc:451-482 + PosInGrid :845-855
  std::vector<std::vector<int>> grid((size_t)COLS * ROWS);
  for (int i = 0; i < ncur; ++i) {
    int px = (int)std::round((kp[i].x - mnMinX) * gwInv);
    int py = (int)std::round((kp[i].y - mnMinY) * ghInv);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    grid[(size_t)px * ROWS + py].push_back(i);
  }
  std::vector<char> assigned(ncur, 0);
  if (occupied)
    for (int i = 0; i < ncur; ++i) assigned[i] = occupied[i] != 0;
  int nmatches = 0;
  std::vector<int> rotHist[30];      // (the QUERY of every histogram entry: its keypoint is best_idx2[query])
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i = 0; i < nq; ++i) {
    if (!q[i].valid) continue;
    const float u = q[i].u, v = q[i].v, radius = q[i].radius;
    if (u < mnMinX || u > mnMaxX) continue;
    if (v < mnMinY || v > mnMaxY) continue;
    // GetFeaturesInArea, Frame.cc:774-843
    const int minLevel = q[i].min_level, maxLevel = q[i].max_level;
    const int nMinCellX = std::max(0, (int)std::floor((u - mnMinX - radius) * gwInv));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((u - mnMinX + radius) * gwInv));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)std::floor((v - mnMinY - radius) * ghInv));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((v - mnMinY + radius) * ghInv));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestIdx2 = -1;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const std::vector<int>& vCell = grid[(size_t)ix * ROWS + iy];
        for (int i2 : vCell) {
          const pli_keypoint& k = kp[i2];
          if (bCheckLevels) {
            if (k.octave < minLevel) continue;
            if (maxLevel >= 0 && k.octave > maxLevel) continue;
          }
          const float distx = k.x - u, disty = k.y - v;
          if (!(std::fabs(distx) < radius && std::fabs(disty) < radius)) continue;
          if (assigned[i2]) continue;                 // mvpMapPoints[i2] with Observations()>0
          if (uright[i2] > 0) {
            const float er = std::fabs(q[i].ur - uright[i2]);
            if (er > radius) continue;
          }
          const int dist = descriptorDistance(qdesc + (size_t)i * 32, desc + (size_t)i2 * 32);
          if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
      }
    if (bestIdx2 >= 0 && bestDist <= TH_HIGH) {
      if (!(q[i].valid & 2)) assigned[bestIdx2] = 1;    // CurrentFrame.mvpMapPoints[bestIdx2] = pMP: not available while pMP has observations
      best_idx2[i] = bestIdx2;
      nmatches++;
      if (checkOri) {
        float rot = q[i].angle - kp[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(i);
      }
    }
  }
  if (raw_idx2) *raw_idx2 = best_idx2;
  if (checkOri) {
    // ComputeThreeMaxima, ORBmatcher.cc:2449-2490
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int s = (int)rotHist[i].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int iq : rotHist[i]) { best_idx2[iq] = -1; nmatches--; }
  }
  return nmatches;
}

// ---------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) ORBmatcher.cc:44-143
// (rectified stereo: F.Nleft == -1), SLAM state reduced to per-query records.
// ---------------------------------------------------------------------------
static inline int searchLocalMap(const pli_proj_query* q, const uint8_t* qdesc, int nq, const pli_keypoint* kp,
                                 const uint8_t* desc, const float* uright, const uint8_t* occupied, int ncur,
                                 float mnMinX, float mnMaxX, float mnMinY, float mnMaxY, float mfNNratio,
                                 std::vector<int>& best_idx2) {
  const int COLS = 64, ROWS = 48, TH_HIGH = 100;
  best_idx2.assign(nq, -1);
  const float gwInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);
  const float ghInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
  std::vector<std::vector<int>> grid((size_t)COLS * ROWS);
  for (int i = 0; i < ncur; ++i) {
    int px = (int)std::round((kp[i].x - mnMinX) * gwInv);
    int py = (int)std::round((kp[i].y - mnMinY) * ghInv);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    grid[(size_t)px * ROWS + py].push_back(i);
  }
  std::vector<char> taken(ncur, 0);
  if (occupied) for (int i = 0; i < ncur; ++i) taken[i] = occupied[i] != 0;
  int nmatches = 0;
  for (int i = 0; i < nq; ++i) {
    if (!q[i].valid) continue;
    const float x = q[i].u, y = q[i].v, r = q[i].radius;
    const int minLevel = q[i].min_level, maxLevel = q[i].max_level;
    const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * gwInv));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + r) * gwInv));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * ghInv));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + r) * ghInv));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
        for (int idx : grid[(size_t)ix * ROWS + iy]) {
          const pli_keypoint& k = kp[idx];
          if (bCheckLevels) {
            if (k.octave < minLevel) continue;
            if (maxLevel >= 0 && k.octave > maxLevel) continue;
          }
          if (!(std::fabs(k.x - x) < r && std::fabs(k.y - y) < r)) continue;
          if (taken[idx]) continue;
          if (uright[idx] > 0) {
            const float er = std::fabs(q[i].ur - uright[idx]);
            if (er > r) continue;
          }
          const int dist = descriptorDistance(qdesc + (size_t)i * 32, desc + (size_t)idx * 32);
          if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k.octave; bestIdx = idx; }
          else if (dist < bestDist2) { bestLevel2 = k.octave; bestDist2 = dist; }
        }
    if (bestIdx >= 0 && bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && (float)bestDist > mfNNratio * (float)bestDist2) continue;
      taken[bestIdx] = 1;
      best_idx2[i] = bestIdx;
      nmatches++;
    }
  }
  return nmatches;
}

// ---------------------------------------------------------------------------
// The same function for a frame of two fisheye cameras (F.Nleft != -1), ORBmatcher.cc:44-214 in full: per map point first
// the left camera (:62-143; no mvuRight gate; a match is also written to the keypoint's stereo partner in the right camera,
// mvLeftToRightMatch, :133-137), then — unless the left ratio test just failed, whose `continue` leaves the map point —
// the right camera (:145-211: GetFeaturesInArea(..., bRight = true), radius WITHOUT th, keypoints idx + Nleft, partner
// mvRightToLeftMatch).  F.mvpMapPoints is one array of Nleft + Nright slots: mpLeft / mpRight receive the index of the map
// point this call left in each slot (-1: none).  A slot counts as taken — "mvpMapPoints[idx] && Observations() > 0" — when
// the caller says so (occupied*) or once this call has written it: the local map's points have observations (they come from
// keyframes), the same reduction of the SLAM state as in searchLocalMap above.  Returns nmatches.
// ---------------------------------------------------------------------------
static inline int searchLocalMapFisheye(const pli_proj_query* qL, const pli_proj_query* qR, const uint8_t* qdesc, int nq,
                                        const pli_keypoint* kpL, const uint8_t* descL, const uint8_t* occL, const int* l2r, int nL,
                                        const pli_keypoint* kpR, const uint8_t* descR, const uint8_t* occR, const int* r2l, int nR,
                                        float mnMinX, float mnMaxX, float mnMinY, float mnMaxY, float mfNNratio,
                                        std::vector<int>& mpLeft, std::vector<int>& mpRight) {
  const int COLS = 64, ROWS = 48, TH_HIGH = 100;
  mpLeft.assign(nL, -1);
  mpRight.assign(nR, -1);
  const float gwInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);
  const float ghInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
  // Frame::AssignFeaturesToGrid (Frame.cc:451-482): mGrid for the left keypoints, mGridRight for the right ones
  auto makeGrid = [&](const pli_keypoint* kp, int n) {
    std::vector<std::vector<int>> grid((size_t)COLS * ROWS);
    for (int i = 0; i < n; ++i) {
      int px = (int)std::round((kp[i].x - mnMinX) * gwInv);
      int py = (int)std::round((kp[i].y - mnMinY) * ghInv);
      if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
      grid[(size_t)px * ROWS + py].push_back(i);
    }
    return grid;
  };
  const auto gridL = makeGrid(kpL, nL), gridR = makeGrid(kpR, nR);
  std::vector<char> takenL(nL, 0), takenR(nR, 0);
  if (occL) for (int i = 0; i < nL; ++i) takenL[i] = occL[i] != 0;
  if (occR) for (int i = 0; i < nR; ++i) takenR[i] = occR[i] != 0;
  // best / second best of one camera (GetFeaturesInArea order); false: the window holds no keypoint at all
  auto best2 = [&](const pli_proj_query& Q, const uint8_t* d, const pli_keypoint* kp, const uint8_t* desc,
                   const std::vector<std::vector<int>>& grid, const std::vector<char>& taken, int& bestDist, int& bestLevel,
                   int& bestDist2, int& bestLevel2, int& bestIdx) -> bool {
    bestDist = 256; bestLevel = -1; bestDist2 = 256; bestLevel2 = -1; bestIdx = -1;
    const float x = Q.u, y = Q.v, r = Q.radius;
    const int minLevel = Q.min_level, maxLevel = Q.max_level;
    const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * gwInv));
    if (nMinCellX >= COLS) return false;
    const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + r) * gwInv));
    if (nMaxCellX < 0) return false;
    const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * ghInv));
    if (nMinCellY >= ROWS) return false;
    const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + r) * ghInv));
    if (nMaxCellY < 0) return false;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    bool any = false;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
        for (int idx : grid[(size_t)ix * ROWS + iy]) {
          const pli_keypoint& k = kp[idx];
          if (bCheckLevels) {
            if (k.octave < minLevel) continue;
            if (maxLevel >= 0 && k.octave > maxLevel) continue;
          }
          if (!(std::fabs(k.x - x) < r && std::fabs(k.y - y) < r)) continue;
          any = true;
          if (taken[idx]) continue;
          const int dist = descriptorDistance(d, desc + (size_t)idx * 32);
          if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k.octave; bestIdx = idx; }
          else if (dist < bestDist2) { bestLevel2 = k.octave; bestDist2 = dist; }
        }
    return any;
  };
  int nmatches = 0;
  for (int i = 0; i < nq; ++i) {
    const uint8_t* d = qdesc + (size_t)i * 32;
    int bestDist, bestLevel, bestDist2, bestLevel2, bestIdx;
    if (qL[i].valid) {                                   // pMP->mbTrackInView
      if (best2(qL[i], d, kpL, descL, gridL, takenL, bestDist, bestLevel, bestDist2, bestLevel2, bestIdx)) {
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && (float)bestDist > mfNNratio * (float)bestDist2) continue;   // (leaves the map point: :126)
          mpLeft[bestIdx] = i; takenL[bestIdx] = 1;
          if (l2r[bestIdx] != -1) { mpRight[l2r[bestIdx]] = i; takenR[l2r[bestIdx]] = 1; nmatches++; }
          nmatches++;
        }
      }
    }
    if (qR[i].valid) {                                   // pMP->mbTrackInViewR && mnTrackScaleLevelR != -1
      if (!best2(qR[i], d, kpR, descR, gridR, takenR, bestDist, bestLevel, bestDist2, bestLevel2, bestIdx)) continue;
      if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && (float)bestDist > mfNNratio * (float)bestDist2) continue;
        if (r2l[bestIdx] != -1) { mpLeft[r2l[bestIdx]] = i; takenL[r2l[bestIdx]] = 1; nmatches++; }
        mpRight[bestIdx] = i; takenR[bestIdx] = 1;
        nmatches++;
      }
    }
  }
  return nmatches;
}

// Frame::ComputeStereoFromRGBD, Frame.cc:1309-1331 (rectified: mvKeysUn == mvKeys)
static inline void stereoFromDepth(const pli_keypoint* kp, int n, const float* depth, int64_t stride, int W, int H, float mbf,
                                   std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
  mvuRight.assign(n, -1.f);
  mvDepth.assign(n, -1.f);
  for (int i = 0; i < n; i++) {
    const float v = kp[i].y, u = kp[i].x;
    const int iv = (int)v, iu = (int)u;                  // cv::Mat::at<float>(int, int) called with floats
    if (iu < 0 || iv < 0 || iu >= W || iv >= H) continue;
    const float d = depth[(int64_t)iv * stride + iu];
    if (d > 0) {
      mvDepth[i] = d;
      mvuRight[i] = kp[i].x - mbf / d;
    }
  }
}

// ---------------------------------------------------------------------------
// DBoW2::TemplatedVocabulary<cv::Mat, FORB> (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) as far as
// Frame::ComputeBoW (Frame.cc:858-870) uses it: the tree as loadFromTextFile builds it (:1350-1433), the
// descent transform(feature, id, weight, nid, levelsup) (:1230-1272), and
// transform(features, BowVector, FeatureVector, levelsup) (:1139-1208) for the ORB vocabulary's
// TF_IDF weighting and L1 scoring (mustNormalize -> BowVector::normalize(L1), BowVector.cpp:59-81).
// ---------------------------------------------------------------------------
struct BowVocabulary {
  struct Node { int parent = 0; std::vector<int> children; uint8_t desc[32] = {0}; double weight = 0; int word_id = -1; };
  int m_k = 0, m_L = 0;
  std::vector<Node> m_nodes;
  int nwords = 0;
  void load(int k, int L, int n, const int* parent, const uint8_t* isLeaf, const uint8_t* desc, const double* weight) {
    m_k = k; m_L = L;
    m_nodes.assign(1, Node());
    nwords = 0;
    for (int i = 0; i < n; ++i) {
      const int nid = (int)m_nodes.size();
      m_nodes.resize(nid + 1);
      m_nodes[nid].parent = parent[i];
      m_nodes[parent[i]].children.push_back(nid);
      std::memcpy(m_nodes[nid].desc, desc + (size_t)i * 32, 32);
      m_nodes[nid].weight = weight[i];
      if (isLeaf[i]) m_nodes[nid].word_id = nwords++;
    }
  }
  void transformFeature(const uint8_t* feature, int& word_id, double& weight, int* nid, int levelsup) const {
    const int nid_level = m_L - levelsup;
    if (nid_level <= 0 && nid) *nid = 0;
    int final_id = 0, current_level = 0;
    do {
      ++current_level;
      const std::vector<int>& nodes = m_nodes[final_id].children;
      final_id = nodes[0];
      double best_d = descriptorDistance(feature, m_nodes[final_id].desc);
      for (size_t j = 1; j < nodes.size(); ++j) {
        const double d = descriptorDistance(feature, m_nodes[nodes[j]].desc);
        if (d < best_d) { best_d = d; final_id = nodes[j]; }
      }
      if (nid && current_level == nid_level) *nid = final_id;
    } while (!m_nodes[final_id].children.empty());
    word_id = m_nodes[final_id].word_id;
    weight = m_nodes[final_id].weight;
  }
  void transform(const uint8_t* features, int n, std::map<unsigned, double>& v, std::map<unsigned, std::vector<unsigned>>& fv,
                 int levelsup) const {
    v.clear(); fv.clear();
    if (m_nodes.size() <= 1) return;
    for (int i = 0; i < n; ++i) {
      int id = 0, nid = 0; double w = 0;
      transformFeature(features + (size_t)i * 32, id, w, &nid, levelsup);
      if (w > 0) { v[(unsigned)id] += w; fv[(unsigned)nid].push_back((unsigned)i); }     // addWeight / addFeature
    }
    double norm = 0.0;                                     // L1, in word order
    for (auto& kv : v) norm += std::fabs(kv.second);
    if (norm > 0.0) for (auto& kv : v) kv.second /= norm;
  }
};


// ---------------------------------------------------------------------------------------------------------------
// The projection part of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, match12)
// (ORBmatcher.cc:2190-2244): one pli_proj_query per keypoint of the last frame, with the last frame's stereo points
// (mvDepth > 0) standing for its map points as Tracking::UpdateLastFrame creates them (Frame::UnprojectStereo,
// Frame.cc:1334-1350; mRwc / mOw from Frame::UpdatePoseMatrices).  Tcw / Tlw: mTcw of the current / last frame, row
// major 3x4.  cv::Mat arithmetic (CV_32F): a product A*B (+ C) is OpenCV's gemm — restated as double accumulation of
// the float products and ONE rounding to float (OpenCV-3.3.1-compatible by intent; parity unpinned like ocv_prims.hpp).
// ---------------------------------------------------------------------------------------------------------------
static inline float cvmatDot3(const float* a, int sa, const float* b, double alpha, double c) {
  const double d = (double)a[0] * (double)b[0] + (double)a[sa] * (double)b[1] + (double)a[2 * sa] * (double)b[2];
  return (float)(alpha * d + c);
}

static inline void trackQueries(const pli_keypoint* lastKp, const float* lastDepth, int n, const float* Tlw, const float* Tcw,
                                float fx, float fy, float cx, float cy, float bf, float th, bool bMono,
                                const float* scaleFactors, pli_proj_query* q) {
  const float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]}, tlw[3] = {Tlw[3], Tlw[7], Tlw[11]};
  float twc[3], tlc[3];
  for (int r = 0; r < 3; ++r) twc[r] = cvmatDot3(Tcw + r, 4, tcw, -1.0, 0.0);                 // twc = -Rcw.t()*tcw
  for (int r = 0; r < 3; ++r) tlc[r] = cvmatDot3(Tlw + 4 * r, 1, twc, 1.0, (double)tlw[r]);   // tlc = Rlw*twc+tlw
  const float mb = bf / fx;
  const bool bForward = tlc[2] > mb && !bMono, bBackward = -tlc[2] > mb && !bMono;
  const float invfx = 1.0f / fx, invfy = 1.0f / fy;
  float Ow[3];
  for (int r = 0; r < 3; ++r) Ow[r] = cvmatDot3(Tlw + r, 4, tlw, -1.0, 0.0);                  // mOw = -mRcw.t()*mtcw
  for (int j = 0; j < n; ++j) {
    pli_proj_query Q;
    Q.u = 0.f; Q.v = 0.f; Q.radius = 0.f; Q.ur = 0.f; Q.min_level = 0; Q.max_level = -1; Q.angle = lastKp[j].angle; Q.valid = 0;
    const float z = lastDepth[j];
    if (z > 0) {
      const float u0 = lastKp[j].x, v0 = lastKp[j].y;
      const float xl[3] = {(u0 - cx) * z * invfx, (v0 - cy) * z * invfy, z};
      float xw[3], xc[3];
      for (int r = 0; r < 3; ++r) xw[r] = cvmatDot3(Tlw + r, 4, xl, 1.0, (double)Ow[r]);      // mRwc*x3Dc+mOw
      for (int r = 0; r < 3; ++r) xc[r] = cvmatDot3(Tcw + 4 * r, 1, xw, 1.0, (double)tcw[r]); // Rcw*x3Dw+tcw
      const float invzc = (float)(1.0 / xc[2]);
      if (!(invzc < 0)) {
        Q.u = fx * xc[0] * invzc + cx;
        Q.v = fy * xc[1] * invzc + cy;
        const int oct = lastKp[j].octave;
        Q.radius = th * scaleFactors[oct];
        Q.ur = Q.u - bf * invzc;
        if (bForward) { Q.min_level = oct; Q.max_level = -1; }
        else if (bBackward) { Q.min_level = 0; Q.max_level = oct; }
        else { Q.min_level = oct - 1; Q.max_level = oct + 1; }
        Q.valid = 1;
      }
    }
    q[j] = Q;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// SURVEY §8(f) row 4, fisheye stereo: Frame::ComputeStereoFishEyeMatches (Frame.cc:1577-1618) with
// KannalaBrandt8::unproject / project / TriangulateMatches / Triangulate (src/CameraModels/KannalaBrandt8.cpp:28-42,
// 103-130, 334-402, 422-435), and the lapping-area ordering of ORBextractor::operator() (ORBextractor.cc:1135-1144).
// Third-party arithmetic restated (OpenCV 3.3.1, parity unpinned like everything OpenCV-backed):
//   * cv::Mat products (CV_32F gemm): double accumulation, one rounding (as trackQueries above);
//   * Mat::dot / cv::norm(NORM_L2): sequential double accumulation;
//   * a*row - row (MatExpr -> addWeighted, float work type); Mat / scalar = convertTo with the float factor (float)(1/s);
//   * cv::SVD::compute(A 4x4 CV_32F, MODIFY_A | FULL_UV): one-sided Jacobi of lapack.cpp (JacobiSVDImpl_<float>, eps =
//     2*FLT_EPSILON, at most 30 sweeps, rotations in float, norms/dot products in double, rows sorted by singular value),
//     with hypot(p, beta) taken as sqrt(p*p + beta*beta) in double;
//   * libm: tan / atan2 / cos / sin of a float are evaluated in double and rounded (std::tan(float), atan2f of glibc are
//     not correctly rounded; the difference is at most the last bit of the float result).
// ---------------------------------------------------------------------------------------------------------------
struct Kb8Camera { float fx, fy, cx, cy, k0, k1, k2, k3; };
static constexpr double CV_PI_D = 3.1415926535897932384626433832795;   // CV_PI

static inline void kb8Unproject(const Kb8Camera& c, float u, float v, float r[3]) {
  const float pwx = (u - c.cx) / c.fx, pwy = (v - c.cy) / c.fy;
  float scale = 1.f;
  float theta_d = sqrtf(pwx * pwx + pwy * pwy);
  theta_d = fminf(fmaxf((float)(-CV_PI_D / 2.0), theta_d), (float)(CV_PI_D / 2.0));
  if ((double)theta_d > 1e-8) {
    float theta = theta_d;
    for (int j = 0; j < 10; j++) {
      float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      float k0_theta2 = c.k0 * theta2, k1_theta4 = c.k1 * theta4;
      float k2_theta6 = c.k2 * theta6, k3_theta8 = c.k3 * theta8;
      float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                        (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
      theta = theta - theta_fix;
      if (fabsf(theta_fix) < 1e-6f) break;          // KannalaBrandt8::precision
    }
    scale = (float)tan((double)theta) / theta_d;
  }
  r[0] = pwx * scale; r[1] = pwy * scale; r[2] = 1.f;
}

static inline void kb8Project(const Kb8Camera& c, const float p[3], float& u, float& v) {
  const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
  const float theta = (float)atan2((double)sqrtf(x2_plus_y2), (double)p[2]);
  const float psi = (float)atan2((double)p[1], (double)p[0]);
  const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
              theta9 = theta7 * theta2;
  const float r = theta + c.k0 * theta3 + c.k1 * theta5 + c.k2 * theta7 + c.k3 * theta9;
  u = (float)((double)(c.fx * r) * cos((double)psi) + (double)c.cx);
  v = (float)((double)(c.fy * r) * sin((double)psi) + (double)c.cy);
}

// cv::SVD::compute on a 4x4 CV_32F matrix: the last row of vt.  At = A transposed (row i of At = column i of A).
static inline void jacobiSvdLastVt4(float At[4][4], float out[4]) {
  const int n = 4, m = 4;
  double W[4];
  float Vt[4][4];
  const float eps = FLT_EPSILON * 2;
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) { const float t = At[i][k]; sd += (double)t * t; }
    W[i] = sd;
    for (int k = 0; k < n; k++) Vt[i][k] = 0;
    Vt[i][i] = 1;
  }
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        float *Ai = At[i], *Aj = At[j];
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
        if (fabs(p) <= (double)eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = sqrt(p * p + beta * beta);
        float c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = (float)sqrt(delta / gamma);
          c = (float)(p / (gamma * s * 2));
        } else {
          c = (float)sqrt((gamma + beta) / (gamma * 2));
          s = (float)(p / (gamma * c * 2));
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          const float t0 = c * Ai[k] + s * Aj[k];
          const float t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += (double)t0 * t0; b += (double)t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        float *Vi = Vt[i], *Vj = Vt[j];
        for (int k = 0; k < n; k++) {
          const float t0 = c * Vi[k] + s * Vj[k];
          const float t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0; Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) { const float t = At[i][k]; sd += (double)t * t; }
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < n - 1; i++) {
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < m; k++) std::swap(At[i][k], At[j][k]);
      for (int k = 0; k < n; k++) std::swap(Vt[i][k], Vt[j][k]);
    }
  }
  for (int k = 0; k < 4; k++) out[k] = Vt[3][k];
}

// KannalaBrandt8::TriangulateMatches: depth in the first camera (or -1), p3D filled on success.  R12: 3x3 row major.
static inline float kb8TriangulateMatches(const Kb8Camera& c1, const Kb8Camera& c2, const pli_keypoint& kp1, const pli_keypoint& kp2,
                                          const float* R12, const float* t12, float sigmaLevel, float unc, float p3D[3]) {
  float r1[3], r2[3], r21[3];
  kb8Unproject(c1, kp1.x, kp1.y, r1);
  kb8Unproject(c2, kp2.x, kp2.y, r2);
  for (int a = 0; a < 3; ++a) r21[a] = cvmatDot3(R12 + 3 * a, 1, r2, 1.0, 0.0);
  const double dot = (double)r1[0] * r21[0] + (double)r1[1] * r21[1] + (double)r1[2] * r21[2];
  const double n1 = sqrt((double)r1[0] * r1[0] + (double)r1[1] * r1[1] + (double)r1[2] * r1[2]);
  const double n2 = sqrt((double)r21[0] * r21[0] + (double)r21[1] * r21[1] + (double)r21[2] * r21[2]);
  const float cosParallaxRays = (float)(dot / (n1 * n2));
  if ((double)cosParallaxRays > 0.9998) return -1;
  float R21[9], t21[3];
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) R21[3 * a + b] = R12[3 * b + a];
  for (int a = 0; a < 3; ++a) t21[a] = cvmatDot3(R21 + 3 * a, 1, t12, -1.0, 0.0);
  const float T1[3][4] = {{1.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f}};
  float T2[3][4];
  for (int a = 0; a < 3; ++a) { T2[a][0] = R21[3 * a]; T2[a][1] = R21[3 * a + 1]; T2[a][2] = R21[3 * a + 2]; T2[a][3] = t21[a]; }
  float At[4][4];                                    // At[col][row] of A
  for (int b = 0; b < 4; ++b) {
    At[b][0] = r1[0] * T1[2][b] - T1[0][b];
    At[b][1] = r1[1] * T1[2][b] - T1[1][b];
    At[b][2] = r2[0] * T2[2][b] - T2[0][b];
    At[b][3] = r2[1] * T2[2][b] - T2[1][b];
  }
  float vh[4];
  jacobiSvdLastVt4(At, vh);
  const float inv = (float)(1.0 / (double)vh[3]);
  float x3D[3] = {vh[0] * inv + 0.f, vh[1] * inv + 0.f, vh[2] * inv + 0.f};
  const float z1 = x3D[2];
  if (z1 <= 0) return -1;
  const float z2 = (float)(((double)R21[6] * x3D[0] + (double)R21[7] * x3D[1] + (double)R21[8] * x3D[2]) + (double)t21[2]);
  if (z2 <= 0) return -1;
  float u1, v1;
  kb8Project(c1, x3D, u1, v1);
  const float errX1 = u1 - kp1.x, errY1 = v1 - kp1.y;
  if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaLevel) return -1;
  float x3D2[3];
  for (int a = 0; a < 3; ++a) x3D2[a] = cvmatDot3(R21 + 3 * a, 1, x3D, 1.0, (double)t21[a]);
  float u2, v2;
  kb8Project(c2, x3D2, u2, v2);
  const float errX2 = u2 - kp2.x, errY2 = v2 - kp2.y;
  if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)unc) return -1;
  p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
  return z1;
}

// Frame::ComputeStereoFishEyeMatches.  kpL / descL: the left table in lapping order (mono first), monoLeft = first lapping
// keypoint; same on the right.  Outputs sized Nleft / Nright: mvLeftToRightMatch, mvRightToLeftMatch, mvDepth (-1),
// mvStereo3Dpoints (zeros where unset).  Returns nMatches.
static inline int computeStereoFishEyeMatches(const pli_keypoint* kpL, const uint8_t* descL, int Nleft, int monoLeft,
                                              const pli_keypoint* kpR, const uint8_t* descR, int Nright, int monoRight,
                                              const Kb8Camera& c1, const Kb8Camera& c2, const float* Rlr, const float* tlr,
                                              const float* levelSigma2, int* l2r, int* r2l, float* depth, float* p3d) {
  for (int i = 0; i < Nleft; ++i) { l2r[i] = -1; depth[i] = -1.0f; p3d[3 * i] = p3d[3 * i + 1] = p3d[3 * i + 2] = 0.f; }
  for (int i = 0; i < Nright; ++i) r2l[i] = -1;
  const int nq = Nleft - monoLeft, nt = Nright - monoRight;
  std::vector<int> idx, dist;
  knn2(descL + (size_t)monoLeft * 32, nq, descR + (size_t)monoRight * 32, nt, idx, dist);
  int nMatches = 0;
  for (int i = 0; i < nq; ++i) {
    if (nt < 2) break;                                   // (*it).size() >= 2
    if (!((double)(float)dist[2 * i] < (double)(float)dist[2 * i + 1] * 0.7)) continue;
    const int li = i + monoLeft, ri = idx[2 * i] + monoRight;
    float p3D[3];
    const float d = kb8TriangulateMatches(c1, c2, kpL[li], kpR[ri], Rlr, tlr, levelSigma2[kpL[li].octave],
                                          levelSigma2[kpR[ri].octave], p3D);
    if (d > 0.0001f) {
      l2r[li] = ri;
      r2l[ri] = li;
      p3d[3 * li] = p3D[0]; p3d[3 * li + 1] = p3D[1]; p3d[3 * li + 2] = p3D[2];
      depth[li] = d;
      nMatches++;
    }
  }
  return nMatches;
}

// The keypoint order ORBextractor::operator() leaves with a lapping area (ORBextractor.cc:1135-1144): keypoints (already
// scaled to level-0 coordinates) with lap0 <= x <= lap1 fill the table from the back, the others from the front.
// order[dst] = src; returns monoIndex.
static inline int lappingOrder(const pli_keypoint* kp, int n, int lap0, int lap1, std::vector<int>& order) {
  order.assign(n, -1);
  int monoIndex = 0, stereoIndex = n - 1;
  for (int i = 0; i < n; ++i) {
    if (kp[i].x >= (float)lap0 && kp[i].x <= (float)lap1) order[stereoIndex--] = i;
    else order[monoIndex++] = i;
  }
  return monoIndex;
}

}  // namespace orc
