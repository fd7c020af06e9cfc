// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// CPU restatement of ORB_SLAM3::ORBextractor (reference src/ORBextractor.cc),
// OpenCV primitives from ocv_prims.hpp.  PARITY UNPINNED: the reference has no
// tests or golden vectors for this path and cannot be built here
// (OpenCV absent); this file follows the reference source line by line in
// behaviour, citing file:line, and defines the two things the reference leaves
// implementation-defined (see DistributeOctTree below).
#pragma once
#include "ocv_prims.hpp"
#include "../include/pli_frontend.h"
#include <list>
#include <utility>

namespace orc {

static const signed char kOrbPattern[1024] = {
#include "../include/pli_orb_pattern.inc"
};

static const int PATCH_SIZE = 31;        // ORBextractor.cc:70
static const int HALF_PATCH_SIZE = 15;   // :71
static const int EDGE_THRESHOLD = 19;    // :72

struct OrbCand { int x, y, score; };     // coordinates relative to (minBorderX, minBorderY)

struct OrbLevelDebug {
  std::vector<OrbCand> candidates;       // vToDistributeKeys, reference order
  std::vector<OrbCand> selected;         // after DistributeOctTree, list order
};

struct OrbExtractor {
  bool trigF32 = true;            // PLI_PARITY_TRIG_F32_ORB
  int nfeatures, nlevels, iniThFAST, minThFAST;
  float scaleFactor;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel;
  std::vector<int> umax;
  std::vector<Img8> mvImagePyramid;      // without the 19 px border (never read on this path)
  std::vector<Img8> mvBlurred;           // debug: GaussianBlur(7x7, 2) per level
  std::vector<OrbLevelDebug> dbg;

  // ORBextractor::ORBextractor, ORBextractor.cc:408-468
  OrbExtractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
      : nfeatures(_nfeatures), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
        scaleFactor(_scaleFactor) {
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale =
        nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      mnFeaturesPerLevel[level] = cvRoundf(nDesiredFeaturesPerScale);
      sumFeatures += mnFeaturesPerLevel[level];
      nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

    // umax, :451-467
    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // ORBextractor::ComputePyramid, ORBextractor.cc:1152-1177.  The reflect-101
  // border the reference adds is not materialised: nothing on the path reads it
  // (FAST stays >= 16 px inside, the blur clones the ROI, stereo SAD checks bounds).
  void ComputePyramid(const Img8& image) {
    for (int level = 0; level < nlevels; ++level) {
      float scale = mvInvScaleFactor[level];
      int sw = cvRoundf((float)image.w * scale), sh = cvRoundf((float)image.h * scale);
      if (level != 0) {
        const Img8& prev = mvImagePyramid[level - 1];
        double inv_x = (double)sw / prev.w, inv_y = (double)sh / prev.h;
        resizeLinear8u(prev, mvImagePyramid[level], sw, sh, 1. / inv_x, 1. / inv_y);
      } else {
        mvImagePyramid[level] = image;
      }
    }
  }

  // IC_Angle, ORBextractor.cc:75-102
  float IC_Angle(const Img8& image, int px, int py) const {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = image.row(py) + px;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    int step = image.w;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
      int v_sum = 0;
      int d = umax[v];
      for (int u = -d; u <= d; ++u) {
        int val_plus = center[u + v * step], val_minus = center[u - v * step];
        v_sum += (val_plus - val_minus);
        m_10 += u * (val_plus + val_minus);
      }
      m_01 += v * v_sum;
    }
    return fastAtan2((float)m_01, (float)m_10);
  }

  // computeOrbDescriptor, ORBextractor.cc:106-145
  // trigF32: `cos(angle)` with a float argument under `using namespace std;` (ORBextractor.cc:65) is std::cos(float) = cosf
  // (PLI_PARITY_TRIG_F32_ORB); false = the correctly rounded value
  static void computeOrbDescriptor(float kpAngle, int px, int py, const Img8& img, uint8_t* desc, bool trigF32) {
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = (float)kpAngle * factorPI;
    float a = cosOfFloat(angle, trigF32), b = sinOfFloat(angle, trigF32);
    const uint8_t* center = img.row(py) + px;
    const int step = img.w;
    const signed char* pattern = kOrbPattern;
    for (int i = 0; i < 32; ++i, pattern += 32) {
      int val = 0;
      for (int k = 0; k < 8; ++k) {
        int x0 = pattern[4 * k], y0 = pattern[4 * k + 1], x1 = pattern[4 * k + 2], y1 = pattern[4 * k + 3];
        int t0 = center[cvRoundf(x0 * b + y0 * a) * step + cvRoundf(x0 * a - y0 * b)];
        int t1 = center[cvRoundf(x1 * b + y1 * a) * step + cvRoundf(x1 * a - y1 * b)];
        val |= (t0 < t1) << k;
      }
      desc[i] = (uint8_t)val;
    }
  }

  // ----- DistributeOctTree, ORBextractor.cc:479-761 --------------------------
  struct Node {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::vector<int> keys;     // indices into the candidate array, input order preserved
    bool bNoMore = false;
    long seq = 0;              // creation order (replaces the heap address in the sort tie-break)
    std::list<Node>::iterator lit;
  };

  static void DivideNode(const Node& n, const std::vector<OrbCand>& K, Node& n1, Node& n2, Node& n3, Node& n4) {
    const int halfX = (int)std::ceil(static_cast<float>(n.URx - n.ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(n.BRy - n.ULy) / 2);
    n1.ULx = n.ULx; n1.ULy = n.ULy;
    n1.URx = n.ULx + halfX; n1.URy = n.ULy;
    n1.BLx = n.ULx; n1.BLy = n.ULy + halfY;
    n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = n.URx; n2.URy = n.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = n.BLx; n3.BLy = n.BLy;
    n3.BRx = n1.BRx; n3.BRy = n.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = n.BRx; n4.BRy = n.BRy;
    for (size_t i = 0; i < n.keys.size(); i++) {
      const OrbCand& kp = K[n.keys[i]];
      if ((float)kp.x < n1.URx) {
        if ((float)kp.y < n1.BRy) n1.keys.push_back(n.keys[i]);
        else n3.keys.push_back(n.keys[i]);
      } else if ((float)kp.y < n1.BRy) n2.keys.push_back(n.keys[i]);
      else n4.keys.push_back(n.keys[i]);
    }
    if (n1.keys.size() == 1) n1.bNoMore = true;
    if (n2.keys.size() == 1) n2.bNoMore = true;
    if (n3.keys.size() == 1) n3.bNoMore = true;
    if (n4.keys.size() == 1) n4.bNoMore = true;
  }

  // Two reference behaviours are implementation-defined and are fixed here:
  //  * :682 sorts pair<int, ExtractorNode*>: equal sizes are ordered by heap
  //    address.  Oracle: by creation order (`seq`), i.e. among equal sizes the
  //    most recently created node is expanded first.
  //  * nothing else: list order, push_front order, first-max-wins are as written.
  static std::vector<int> DistributeOctTree(const std::vector<OrbCand>& K, int minX, int maxX, int minY,
                                            int maxY, int N) {
    std::vector<int> result;
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return result;     // degenerate aspect (reference divides by zero)
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> lNodes;
    std::vector<Node*> vpIniNodes(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
      Node ni;
      ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
      ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
      ni.BLx = ni.ULx; ni.BLy = maxY - minY;
      ni.BRx = ni.URx; ni.BRy = maxY - minY;
      ni.seq = seq++;
      lNodes.push_back(ni);
      vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < K.size(); i++) {
      int idx = (int)((float)K[i].x / hX);
      if (idx >= nIni) idx = nIni - 1;   // cannot happen for in-range keys; guards the oracle only
      vpIniNodes[idx]->keys.push_back((int)i);
    }
    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
      if (lit->keys.size() == 1) { lit->bNoMore = true; lit++; }
      else if (lit->keys.empty()) lit = lNodes.erase(lit);
      else lit++;
    }
    bool bFinish = false;
    typedef std::pair<int, Node*> SP;
    auto cmp = [](const SP& a, const SP& b) {
      if (a.first != b.first) return a.first < b.first;
      return a.second->seq < b.second->seq;
    };
    std::vector<SP> vSizeAndPointerToNode;
    auto pushChild = [&](Node& c, int& nToExpand) {
      if (c.keys.size() > 0) {
        c.seq = seq++;
        lNodes.push_front(c);
        if (c.keys.size() > 1) {
          nToExpand++;
          vSizeAndPointerToNode.push_back(std::make_pair((int)c.keys.size(), &lNodes.front()));
          lNodes.front().lit = lNodes.begin();
        }
      }
    };
    while (!bFinish) {
      int prevSize = (int)lNodes.size();
      lit = lNodes.begin();
      int nToExpand = 0;
      vSizeAndPointerToNode.clear();
      while (lit != lNodes.end()) {
        if (lit->bNoMore) { lit++; continue; }
        Node n1, n2, n3, n4;
        DivideNode(*lit, K, n1, n2, n3, n4);
        pushChild(n1, nToExpand); pushChild(n2, nToExpand);
        pushChild(n3, nToExpand); pushChild(n4, nToExpand);
        lit = lNodes.erase(lit);
      }
      if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
        bFinish = true;
      } else if (((int)lNodes.size() + nToExpand * 3) > N) {
        while (!bFinish) {
          prevSize = (int)lNodes.size();
          std::vector<SP> vPrev = vSizeAndPointerToNode;
          vSizeAndPointerToNode.clear();
          std::sort(vPrev.begin(), vPrev.end(), cmp);
          for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
            Node n1, n2, n3, n4;
            DivideNode(*vPrev[j].second, K, n1, n2, n3, n4);
            int dummy = 0;
            pushChild(n1, dummy); pushChild(n2, dummy); pushChild(n3, dummy); pushChild(n4, dummy);
            lNodes.erase(vPrev[j].second->lit);
            if ((int)lNodes.size() >= N) break;
          }
          if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
        }
      }
    }
    // Retain the best point in each node (:741-757): first maximum wins.
    for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
      const std::vector<int>& vk = it->keys;
      int best = vk[0];
      int maxResponse = K[best].score;
      for (size_t k = 1; k < vk.size(); k++) {
        if (K[vk[k]].score > maxResponse) { best = vk[k]; maxResponse = K[vk[k]].score; }
      }
      result.push_back(best);
    }
    return result;
  }

  // Per-cell cv::FAST with the two thresholds, ORBextractor.cc:763-861.
  // The arc value is threshold independent, so one evaluation of the cell
  // serves both cv::FAST calls: FAST(t) keeps pixels with arc > t that are
  // strictly greater than their 8 neighbours INSIDE the evaluated interior of
  // the cell sub-image (rows/cols 3..dim-4); pixels outside count as 0.
  static void fastCell(const Img8& im, int x0, int y0, int x1, int y1, int iniTh, int minTh,
                       std::vector<OrbCand>& out /* coords relative to the cell sub-image */) {
    int cw = x1 - x0, ch = y1 - y0;
    if (cw < 7 || ch < 7) return;
    int iw = cw - 6, ih = ch - 6;      // interior
    std::vector<int> sc((size_t)iw * ih);
    for (int y = 0; y < ih; ++y)
      for (int x = 0; x < iw; ++x) {
        int arc = fastArcValue(im.row(y0 + 3 + y) + x0 + 3 + x, im.w);
        sc[(size_t)y * iw + x] = arc > minTh ? arc - 1 : 0;   // cornerScore; 0 = not a corner even at minTh
      }
    auto S = [&](int y, int x) { return (x < 0 || y < 0 || x >= iw || y >= ih) ? 0 : sc[(size_t)y * iw + x]; };
    std::vector<OrbCand> nms;
    bool anyIni = false;
    for (int y = 0; y < ih; ++y)
      for (int x = 0; x < iw; ++x) {
        int s = S(y, x);
        if (s == 0) continue;
        if (s > S(y - 1, x - 1) && s > S(y - 1, x) && s > S(y - 1, x + 1) && s > S(y, x - 1) &&
            s > S(y, x + 1) && s > S(y + 1, x - 1) && s > S(y + 1, x) && s > S(y + 1, x + 1)) {
          nms.push_back({x + 3, y + 3, s});
          if (s >= iniTh) anyIni = true;       // arc > iniTh  <=>  score >= iniTh
        }
      }
    for (const OrbCand& c : nms)
      if (!anyIni || c.score >= iniTh) out.push_back(c);
  }

  struct KP { float x, y, size, angle, response; int octave; int lx, ly; };

  // ComputeKeyPointsOctTree, ORBextractor.cc:763-878
  void ComputeKeyPointsOctTree(std::vector<std::vector<KP>>& allKeypoints) {
    allKeypoints.assign(nlevels, std::vector<KP>());
    dbg.assign(nlevels, OrbLevelDebug());
    const float W = 30;
    for (int level = 0; level < nlevels; ++level) {
      const Img8& im = mvImagePyramid[level];
      const int minBorderX = EDGE_THRESHOLD - 3;
      const int minBorderY = minBorderX;
      const int maxBorderX = im.w - EDGE_THRESHOLD + 3;
      const int maxBorderY = im.h - EDGE_THRESHOLD + 3;
      std::vector<OrbCand> vToDistributeKeys;
      const float width = (float)(maxBorderX - minBorderX);
      const float height = (float)(maxBorderY - minBorderY);
      const int nCols = (int)(width / W);
      const int nRows = (int)(height / W);
      if (nCols <= 0 || nRows <= 0) continue;   // image too small for this level (reference would divide by zero)
      const int wCell = (int)std::ceil(width / nCols);
      const int hCell = (int)std::ceil(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBorderX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = (float)maxBorderX;
          std::vector<OrbCand> vKeysCell;
          fastCell(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, iniThFAST, minThFAST, vKeysCell);
          for (OrbCand c : vKeysCell) {
            c.x += j * wCell;
            c.y += i * hCell;
            vToDistributeKeys.push_back(c);
          }
        }
      }
      std::vector<int> sel = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY,
                                               maxBorderY, mnFeaturesPerLevel[level]);
      dbg[level].candidates = vToDistributeKeys;
      const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
      for (int idx : sel) {
        const OrbCand& c = vToDistributeKeys[idx];
        dbg[level].selected.push_back(c);
        KP kp;
        kp.lx = c.x + minBorderX;
        kp.ly = c.y + minBorderY;
        kp.x = (float)kp.lx;
        kp.y = (float)kp.ly;
        kp.response = (float)c.score;
        kp.octave = level;
        kp.size = (float)scaledPatchSize;
        kp.angle = -1.f;
        allKeypoints[level].push_back(kp);
      }
    }
    for (int level = 0; level < nlevels; ++level)
      for (KP& kp : allKeypoints[level]) kp.angle = IC_Angle(mvImagePyramid[level], kp.lx, kp.ly);
  }

  // ORBextractor::operator(), ORBextractor.cc:1068-1150 with vLappingArea={0,0}
  // (Frame.cc:486): keypoints with 0 <= x <= 0 would go to the "stereo" tail; x is
  // always >= 16, so all keypoints are emitted level-major from index 0.
  int operator()(const Img8& image, std::vector<pli_keypoint>& keypoints, std::vector<uint8_t>& descriptors) {
    keypoints.clear();
    descriptors.clear();
    if (image.w == 0 || image.h == 0) return -1;
    ComputePyramid(image);
    std::vector<std::vector<KP>> allKeypoints;
    ComputeKeyPointsOctTree(allKeypoints);
    mvBlurred.assign(nlevels, Img8());
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KP>& kps = allKeypoints[level];
      if (kps.empty()) continue;
      gaussianBlur8u(mvImagePyramid[level], mvBlurred[level], 7, 2.0);
      float scale = mvScaleFactor[level];
      for (KP& kp : kps) {
        uint8_t d[32];
        computeOrbDescriptor(kp.angle, kp.lx, kp.ly, mvBlurred[level], d, trigF32);
        pli_keypoint o;
        o.x = kp.x; o.y = kp.y;
        if (level != 0) { o.x = kp.x * scale; o.y = kp.y * scale; }
        o.size = kp.size; o.angle = kp.angle; o.response = kp.response; o.octave = kp.octave;
        keypoints.push_back(o);
        descriptors.insert(descriptors.end(), d, d + 32);
      }
    }
    return (int)keypoints.size();
  }
};

}  // namespace orc
