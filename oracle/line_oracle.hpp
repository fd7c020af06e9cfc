// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// CPU restatement of the reference line front-end:
//   ORB_SLAM3::Lineextractor::operator()          src/LineExtractor.cc:31-70
//   cv::line_descriptor::LSDDetectorC::detectImpl Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:227-324
//   cv::LineSegmentDetector (OpenCV 3.3.1 imgproc/src/lsd.cpp; third-party, NOT
//     under /root/reference: restated from the published algorithm, refine = 0)
//   cv::line_descriptor::BinaryDescriptor::compute / computeLBD
//                                                 Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:217-259,350-412,524-687,1026-1372
// PARITY UNPINNED (no reference tests/golden vectors for this path, OpenCV not
// runnable here).  Choices the reference leaves open, fixed here and mirrored
// by the HIP kernels:
//   * LSD input: OpenCV 3.0-3.4 convert the image to CV_64FC1 first (double blur, double
//     resize, double gradient: LsdParams::input_f64, the default — the reference pins
//     3.3.1); the detector re-added in 4.5.x works on CV_8UC1 (u8 fixed-point 7x7
//     sigma-0.6 blur and x1.2 bilinear resize, integer 2x2 gradient): input_f64 = false.
//   * seeds of equal gradient bin are visited in raster order (OpenCV sorts the
//     bins with an unstable std::sort; the earlier linked-list version of the
//     same file visits them in raster order).
//   * Lineextractor's std::sort by response is unstable: equal responses keep
//     detection order (std::stable_sort).
#pragma once
#include "ocv_prims.hpp"
#include "../include/pli_frontend.h"

namespace orc {

struct LsdDebug {
  Img8 scaled;                       // blurred + resized image (CV_8UC1 pipeline)
  std::vector<double> scaled64;      // ... (CV_64FC1 pipeline)
  std::vector<float> angleDeg;       // fastAtan2 degrees, -1024 = NOTDEF
  std::vector<int> order;            // pixel index (y*W'+x) of every list entry, visiting order
  std::vector<float> segments;       // x1,y1,x2,y2 per detected segment (Vec4f), detection order
  int W = 0, H = 0;
};

static const double kNOTDEF = -1024.0;
static const double kPI = 3.14159265358979323846;
static const double kM_3_2_PI = (3 * kPI) / 2;
static const double kM_2__PI = 2 * kPI;
static const double kDEG_TO_RADS = kPI / 180;

struct LsdParams {
  int refine = 0;
  double scale = 1.2, sigma_scale = 0.6, quant = 2.0, ang_th = 22.5, log_eps = 1.0, density_th = 0.6;
  int n_bins = 1024;
  bool input_f64 = true;          // PLI_PARITY_LSD_F64: the CV_64FC1 pipeline of OpenCV 3.x
  bool trig_f32 = false;          // PLI_PARITY_TRIG_F32_LSD: cos(float(angle)) is cosf
};

// cv::LineSegmentDetectorImpl::detect -> flsd (refine == LSD_REFINE_NONE).
static inline void lsdDetect(const Img8& image, const LsdParams& P, LsdDebug& D) {
  const double prec = kPI * P.ang_th / 180;
  const double rho = P.quant / std::sin(prec);
  Img8 scaled;
  Img64 scaled64;
  if (P.scale != 1) {
    const double sigma = (P.scale < 1) ? (P.sigma_scale / P.scale) : (P.sigma_scale);
    const double sprec = 3;
    const unsigned int h = (unsigned int)(std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0))));
    int dw = cvRound(image.w * P.scale), dh = cvRound(image.h * P.scale);
    if (P.input_f64) {
      Img64 g;
      gaussianBlur64f(image, g, 1 + 2 * (int)h, sigma);
      resizeLinear64f(g, scaled64, dw, dh, 1. / P.scale, 1. / P.scale);
    } else {
      Img8 g;
      gaussianBlur8u(image, g, 1 + 2 * (int)h, sigma);
      resizeLinear8u(g, scaled, dw, dh, 1. / P.scale, 1. / P.scale);
    }
  } else {
    scaled = image;
    if (P.input_f64) {
      scaled64 = Img64(image.w, image.h);
      for (size_t i = 0; i < image.d.size(); ++i) scaled64.d[i] = (double)image.d[i];
    }
  }
  const int W = P.input_f64 ? scaled64.w : scaled.w, H = P.input_f64 ? scaled64.h : scaled.h;
  D.W = W; D.H = H;
  D.scaled = scaled;
  D.scaled64 = scaled64.d;
  // ll_angle
  std::vector<double> angles((size_t)W * H, kNOTDEF), modgrad((size_t)W * H, 0.0);
  std::vector<float> adeg((size_t)W * H, -1024.f);
  double max_grad = -1;
  for (int y = 0; y < H - 1; ++y) {
    for (int x = 0; x < W - 1; ++x) {
      double norm;
      float gxf, gyf;
      if (P.input_f64) {
        const double* r0 = scaled64.row(y);
        const double* r1 = scaled64.row(y + 1);
        const double DA = r1[x + 1] - r0[x];
        const double BC = r0[x + 1] - r1[x];
        const double gx = DA + BC;
        const double gy = DA - BC;
        norm = std::sqrt((gx * gx + gy * gy) / 4);
        gxf = float(gx); gyf = float(-gy);
      } else {
        const uint8_t* r0 = scaled.row(y);
        const uint8_t* r1 = scaled.row(y + 1);
        int DA = r1[x + 1] - r0[x];
        int BC = r0[x + 1] - r1[x];
        int gx = DA + BC;
        int gy = DA - BC;
        norm = std::sqrt((gx * gx + gy * gy) / 4.0);
        gxf = float(gx); gyf = float(-gy);
      }
      modgrad[(size_t)y * W + x] = norm;
      if (norm <= rho) {
        angles[(size_t)y * W + x] = kNOTDEF;
      } else {
        float deg = fastAtan2(gxf, gyf);
        adeg[(size_t)y * W + x] = deg;
        angles[(size_t)y * W + x] = deg * kDEG_TO_RADS;
        if (norm > max_grad) max_grad = norm;
      }
    }
  }
  D.angleDeg = adeg;
  // ordered list: bin descending, raster order inside a bin.  Covers x<W-1, y<H-1.
  const int n_bins = P.n_bins;
  double bin_coef = (max_grad > 0) ? double(n_bins - 1) / max_grad : 0;
  std::vector<std::vector<int>> bins(n_bins);
  for (int y = 0; y < H - 1; ++y)
    for (int x = 0; x < W - 1; ++x) {
      int i = int(modgrad[(size_t)y * W + x] * bin_coef);
      if (i >= n_bins) i = n_bins - 1;
      bins[i].push_back(y * W + x);
    }
  std::vector<int>& order = D.order;
  order.clear();
  order.reserve((size_t)W * H);
  for (int b = n_bins - 1; b >= 0; --b) order.insert(order.end(), bins[b].begin(), bins[b].end());

  const double p = P.ang_th / 180;
  const double LOG_NT = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
  const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
  std::vector<uint8_t> used((size_t)W * H, 0);
  struct RP { int x, y; double modgrad; };
  std::vector<RP> reg;
  D.segments.clear();

  auto isAligned = [&](int x, int y, double theta) -> bool {
    if (x < 0 || y < 0 || x >= W || y >= H) return false;
    const double a = angles[(size_t)y * W + x];
    if (a == kNOTDEF) return false;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > kM_3_2_PI) {
      n_theta -= kM_2__PI;
      if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
  };
  auto angle_diff = [&](double a, double b) {
    double diff = a - b;
    while (diff <= -kPI) diff += kM_2__PI;
    while (diff > kPI) diff -= kM_2__PI;
    return std::fabs(diff);
  };

  for (size_t i = 0; i < order.size(); ++i) {
    int sp = order[i];
    if (used[sp] || angles[sp] == kNOTDEF) continue;
    int sx = sp % W, sy = sp / W;
    // region_grow
    reg.clear();
    double reg_angle = angles[sp];
    reg.push_back({sx, sy, modgrad[sp]});
    float sumdx = float(std::cos(reg_angle));
    float sumdy = float(std::sin(reg_angle));
    used[sp] = 1;
    for (size_t k = 0; k < reg.size(); k++) {
      const RP rp = reg[k];
      int xx_min = std::max(rp.x - 1, 0), xx_max = std::min(rp.x + 1, W - 1);
      int yy_min = std::max(rp.y - 1, 0), yy_max = std::min(rp.y + 1, H - 1);
      for (int yy = yy_min; yy <= yy_max; ++yy)
        for (int xx = xx_min; xx <= xx_max; ++xx) {
          size_t q = (size_t)yy * W + xx;
          if (!used[q] && isAligned(xx, yy, reg_angle)) {
            const double angle = angles[q];
            used[q] = 1;
            reg.push_back({xx, yy, modgrad[q]});
            sumdx += cosOfFloat(float(angle), P.trig_f32);
            sumdy += sinOfFloat(float(angle), P.trig_f32);
            reg_angle = fastAtan2(sumdy, sumdx) * kDEG_TO_RADS;
          }
        }
    }
    if (reg.size() < min_reg_size) continue;
    // region2rect
    double x = 0, y = 0, sum = 0;
    for (size_t k = 0; k < reg.size(); ++k) {
      const double weight = reg[k].modgrad;
      x += double(reg[k].x) * weight;
      y += double(reg[k].y) * weight;
      sum += weight;
    }
    x /= sum;
    y /= sum;
    // get_theta
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (size_t k = 0; k < reg.size(); ++k) {
      const double regx = reg[k].x, regy = reg[k].y, weight = reg[k].modgrad;
      double dx = regx - x, dy = regy - y;
      Ixx += dy * dy * weight;
      Iyy += dx * dx * weight;
      Ixy -= dx * dy * weight;
    }
    double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fastAtan2(float(lambda - Ixx), float(Ixy)))
                                                     : double(fastAtan2(float(Ixy), float(lambda - Iyy)));
    theta *= kDEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += kPI;
    double dx = std::cos(theta), dy = std::sin(theta);
    double l_min = 0, l_max = 0;
    for (size_t k = 0; k < reg.size(); ++k) {
      double regdx = double(reg[k].x) - x;
      double regdy = double(reg[k].y) - y;
      double l = regdx * dx + regdy * dy;
      if (l > l_max) l_max = l;
      else if (l < l_min) l_min = l;
    }
    double x1 = x + l_min * dx, y1 = y + l_min * dy, x2 = x + l_max * dx, y2 = y + l_max * dy;
    x1 += 0.5; y1 += 0.5; x2 += 0.5; y2 += 0.5;
    if (P.scale != 1) { x1 /= P.scale; y1 /= P.scale; x2 /= P.scale; y2 /= P.scale; }
    D.segments.push_back(float(x1));
    D.segments.push_back(float(y1));
    D.segments.push_back(float(x2));
    D.segments.push_back(float(y2));
  }
}

// LSDDetectorC::detectImpl (1 octave), LSDDetector_custom.cpp:264-308
static inline void buildKeyLines(const std::vector<float>& segs, int imgW, int imgH, double min_length,
                                 std::vector<pli_keyline>& out) {
  out.clear();
  int class_counter = -1;
  for (size_t k = 0; k + 3 < segs.size(); k += 4) {
    float e[4] = {segs[k], segs[k + 1], segs[k + 2], segs[k + 3]};
    // checkLineExtremes :76-102
    if (e[0] < 0) e[0] = 0;
    if (e[0] >= imgW) e[0] = (float)imgW - 1.0f;
    if (e[2] < 0) e[2] = 0;
    if (e[2] >= imgW) e[2] = (float)imgW - 1.0f;
    if (e[1] < 0) e[1] = 0;
    if (e[1] >= imgH) e[1] = (float)imgH - 1.0f;
    if (e[3] < 0) e[3] = 0;
    if (e[3] >= imgH) e[3] = (float)imgH - 1.0f;
    double ddx = (double)(e[0] - e[2]), ddy = (double)(e[1] - e[3]);
    double length = (float)std::sqrt(ddx * ddx + ddy * ddy);   // pow(float,2) promotes to double
    if (length > min_length) {
      pli_keyline kl;
      const float octaveScale = 1.0f;   // pow((float)scale, 0)
      kl.startPointX = e[0] * octaveScale;
      kl.startPointY = e[1] * octaveScale;
      kl.endPointX = e[2] * octaveScale;
      kl.endPointY = e[3] * octaveScale;
      kl.sPointInOctaveX = e[0];
      kl.sPointInOctaveY = e[1];
      kl.ePointInOctaveX = e[2];
      kl.ePointInOctaveY = e[3];
      kl.lineLength = (float)length;
      kl.numOfPixels = lineIteratorCount(e[0], e[1], e[2], e[3]);
      kl.angle = (float)std::atan2((double)(kl.endPointY - kl.startPointY), (double)(kl.endPointX - kl.startPointX));
      kl.class_id = ++class_counter;
      kl.octave = 0;
      kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
      kl.response = kl.lineLength / (float)std::max(imgW, imgH);
      kl.pt_x = (kl.endPointX + kl.startPointX) / 2;
      kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
      out.push_back(kl);
    }
  }
}

// BinaryDescriptor weights, binary_descriptor_custom.cpp:217-259 (integer division quirks kept).
struct LbdWeights {
  float gaussCoefL[21];
  float gaussCoefG[63];
  LbdWeights() {
    const int widthOfBand = 7, NUM_OF_BANDS = 9;
    std::vector<double> L(21), G(63);
    double u = (widthOfBand * 3 - 1) / 2;
    double sigma = (widthOfBand * 2 + 1) / 2;
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < widthOfBand * 3; i++) {
      double dis = i - u;
      L[i] = std::exp(dis * dis * invsigma2);
    }
    u = (NUM_OF_BANDS * widthOfBand - 1) / 2;
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < NUM_OF_BANDS * widthOfBand; i++) {
      double dis = i - u;
      G[i] = std::exp(dis * dis * invsigma2);
    }
    for (int i = 0; i < 21; ++i) gaussCoefL[i] = (float)L[i];   // used as (float) gaussCoefL_[..]
    for (int i = 0; i < 63; ++i) gaussCoefG[i] = (float)G[i];
  }
};

static const int kLbdCombinations[32][2] = {
    {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
    {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
    {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

// computeLBD for one line, binary_descriptor_custom.cpp:1026-1340 (useDetectionData=false).
static inline void computeLBDLine(const pli_keyline& kl, const int16_t* pdxImg, const int16_t* pdyImg,
                                  int imgW, int imgH, const LbdWeights& Wt, float desVec[72], bool trigF32 = false) {
  const int NUM_OF_BANDS = 9, widthOfBand = 7;
  float dL[2], dO[2];
  short heightOfLSP = (short)(widthOfBand * NUM_OF_BANDS);
  float pgdLBandSum[9] = {0}, ngdLBandSum[9] = {0}, pgdL2BandSum[9] = {0}, ngdL2BandSum[9] = {0};
  float pgdOBandSum[9] = {0}, ngdOBandSum[9] = {0}, pgdO2BandSum[9] = {0}, ngdO2BandSum[9] = {0};
  short halfHeight = (heightOfLSP - 1) / 2;
  short realWidth = (short)imgW;
  short imageWidth = realWidth - 1;
  short imageHeight = (short)(imgH - 1);
  short lengthOfLSP = (short)kl.numOfPixels;
  short halfWidth = (lengthOfLSP - 1) / 2;
  float lineMiddlePointX = (float)(0.5 * (kl.sPointInOctaveX + kl.ePointInOctaveX));
  float lineMiddlePointY = (float)(0.5 * (kl.sPointInOctaveY + kl.ePointInOctaveY));
  dL[0] = cosOfFloat(kl.angle, trigF32);      // cos( pSingleLine->direction ), binary_descriptor_custom.cpp:1130 (PLI_PARITY_TRIG_F32_LBD)
  dL[1] = sinOfFloat(kl.angle, trigF32);
  dO[0] = -dL[1];
  dO[1] = dL[0];
  float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
  float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
  for (short hID = 0; hID < heightOfLSP; hID++) {
    float sCorX = sCorX0, sCorY = sCorY0;
    float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
    for (short wID = 0; wID < lengthOfLSP; wID++) {
      short tempCor = (short)std::round(sCorX);
      short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
      tempCor = (short)std::round(sCorY);
      short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
      short dx = pdxImg[yCor * realWidth + xCor];
      short dy = pdyImg[yCor * realWidth + xCor];
      float gDL = dx * dL[0] + dy * dL[1];
      float gDO = dx * dO[0] + dy * dO[1];
      if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
      if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
      sCorX += dL[0];
      sCorY += dL[1];
    }
    sCorX0 -= dL[1];
    sCorY0 += dL[0];
    float coefInGaussion = Wt.gaussCoefG[hID];
    pgdLRowSum = coefInGaussion * pgdLRowSum;
    ngdLRowSum = coefInGaussion * ngdLRowSum;
    float pgdL2RowSum = pgdLRowSum * pgdLRowSum;
    float ngdL2RowSum = ngdLRowSum * ngdLRowSum;
    pgdORowSum = coefInGaussion * pgdORowSum;
    ngdORowSum = coefInGaussion * ngdORowSum;
    float pgdO2RowSum = pgdORowSum * pgdORowSum;
    float ngdO2RowSum = ngdORowSum * ngdORowSum;
    short bandID = (short)(hID / widthOfBand);
    auto acc = [&](int b, float c) {
      pgdLBandSum[b] += c * pgdLRowSum;
      ngdLBandSum[b] += c * ngdLRowSum;
      pgdL2BandSum[b] += c * c * pgdL2RowSum;
      ngdL2BandSum[b] += c * c * ngdL2RowSum;
      pgdOBandSum[b] += c * pgdORowSum;
      ngdOBandSum[b] += c * ngdORowSum;
      pgdO2BandSum[b] += c * c * pgdO2RowSum;
      ngdO2BandSum[b] += c * c * ngdO2RowSum;
    };
    acc(bandID, Wt.gaussCoefL[hID % widthOfBand + widthOfBand]);
    bandID--;
    if (bandID >= 0) acc(bandID, Wt.gaussCoefL[hID % widthOfBand + 2 * widthOfBand]);
    bandID = bandID + 2;
    if (bandID < NUM_OF_BANDS) acc(bandID, Wt.gaussCoefL[hID % widthOfBand]);
  }
  float invN2 = (float)(1.0 / (widthOfBand * 2.0));
  float invN3 = (float)(1.0 / (widthOfBand * 3.0));
  for (int bandID = 0; bandID < NUM_OF_BANDS; bandID++) {
    float invN = (bandID == 0 || bandID == NUM_OF_BANDS - 1) ? invN2 : invN3;
    int desID = bandID * 8;
    float temp = pgdLBandSum[bandID] * invN;
    desVec[desID] = temp;
    desVec[desID + 4] = std::sqrt(pgdL2BandSum[bandID] * invN - temp * temp);
    temp = ngdLBandSum[bandID] * invN;
    desVec[desID + 1] = temp;
    desVec[desID + 5] = std::sqrt(ngdL2BandSum[bandID] * invN - temp * temp);
    temp = pgdOBandSum[bandID] * invN;
    desVec[desID + 2] = temp;
    desVec[desID + 6] = std::sqrt(pgdO2BandSum[bandID] * invN - temp * temp);
    temp = ngdOBandSum[bandID] * invN;
    desVec[desID + 3] = temp;
    desVec[desID + 7] = std::sqrt(ngdO2BandSum[bandID] * invN - temp * temp);
  }
  float tempM = 0, tempS = 0;
  for (int i = 0; i < 72; i += 8) {
    tempM += desVec[i] * desVec[i];
    tempM += desVec[i + 1] * desVec[i + 1];
    tempM += desVec[i + 2] * desVec[i + 2];
    tempM += desVec[i + 3] * desVec[i + 3];
    tempS += desVec[i + 4] * desVec[i + 4];
    tempS += desVec[i + 5] * desVec[i + 5];
    tempS += desVec[i + 6] * desVec[i + 6];
    tempS += desVec[i + 7] * desVec[i + 7];
  }
  tempM = 1 / std::sqrt(tempM);
  tempS = 1 / std::sqrt(tempS);
  for (int i = 0; i < 72; i += 8) {
    desVec[i] = desVec[i] * tempM;
    desVec[i + 1] = desVec[i + 1] * tempM;
    desVec[i + 2] = desVec[i + 2] * tempM;
    desVec[i + 3] = desVec[i + 3] * tempM;
    desVec[i + 4] = desVec[i + 4] * tempS;
    desVec[i + 5] = desVec[i + 5] * tempS;
    desVec[i + 6] = desVec[i + 6] * tempS;
    desVec[i + 7] = desVec[i + 7] * tempS;
  }
  for (int i = 0; i < 72; i++)
    if (desVec[i] > 0.4) desVec[i] = (float)0.4;
  float temp = 0;
  for (int i = 0; i < 72; i++) temp += desVec[i] * desVec[i];
  temp = 1 / std::sqrt(temp);
  for (int i = 0; i < 72; i++) desVec[i] = desVec[i] * temp;
}

// binaryConversion + combinations, binary_descriptor_custom.cpp:401-412,74-107,662-666
static inline void lbdBinarise(const float desVec[72], uint8_t out[32]) {
  for (int comb = 0; comb < 32; comb++) {
    const float* f1 = &desVec[8 * kLbdCombinations[comb][0]];
    const float* f2 = &desVec[8 * kLbdCombinations[comb][1]];
    uint8_t result = 0;
    for (int i = 0; i < 8; i++)
      if (f1[i] > f2[i]) result += (uint8_t)(1 << i);
    out[comb] = result;
  }
}

struct LineDebug {
  LsdDebug lsd;
  std::vector<int16_t> dx, dy;
  std::vector<float> lbdFloat;      // n x 72
};

struct LineExtractorCfg {
  int lsd_nfeatures = 500;
  double min_line_length = 0.025;
  LsdParams lsd;
  bool lbd_trig_f32 = false;      // PLI_PARITY_TRIG_F32_LBD
};

// Lineextractor::operator(), LineExtractor.cc:31-70
static inline void lineExtract(const Img8& img, const LineExtractorCfg& C, std::vector<pli_keyline>& keylines,
                               std::vector<uint8_t>& descriptors, LineDebug& D) {
  keylines.clear();
  descriptors.clear();
  lsdDetect(img, C.lsd, D.lsd);
  double min_length = C.min_line_length * (std::min(img.w, img.h));
  buildKeyLines(D.lsd.segments, img.w, img.h, min_length, keylines);
  if ((int)keylines.size() > C.lsd_nfeatures && C.lsd_nfeatures != 0) {
    std::stable_sort(keylines.begin(), keylines.end(),
                     [](const pli_keyline& a, const pli_keyline& b) { return a.response > b.response; });
    keylines.resize(C.lsd_nfeatures);
    for (int i = 0; i < C.lsd_nfeatures; i++) keylines[i].class_id = i;
  }
  // BinaryDescriptor::compute -> computeImpl -> computeSobel: blur 5x5 sigma 1, Sobel dx/dy
  Img8 blurred;
  gaussianBlur8u(img, blurred, 5, 1.0);
  sobel3x3_16s(blurred, D.dx, D.dy);
  D.lbdFloat.clear();
  if (keylines.empty()) return;     // "Error: keypoint list is empty": descriptors untouched
  static const LbdWeights Wt;
  descriptors.resize(keylines.size() * 32);
  D.lbdFloat.resize(keylines.size() * 72);
  for (size_t i = 0; i < keylines.size(); ++i) {
    float* des = &D.lbdFloat[i * 72];
    computeLBDLine(keylines[i], D.dx.data(), D.dy.data(), img.w, img.h, Wt, des, C.lbd_trig_f32);
    lbdBinarise(des, &descriptors[i * 32]);
  }
}

}  // namespace orc
