// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Restatement of the OpenCV 3.3.1 primitives the reference front-end calls.
// OpenCV is a third-party dependency of the reference (README.md:18-20,
// CMakeLists.txt:36 `find_package(OpenCV 3)`; the shipped ELF links
// libopencv_core3.so.3.3) that is neither vendored under /root/reference nor
// installed in the build image, so these functions restate its published
// algorithms ("OpenCV-3.3.1-compatible by intent").
// PARITY UNPINNED for this file: the reference holds no golden vectors at the
// OpenCV boundary and OpenCV itself cannot be run here.  Call sites anchoring
// each primitive are cited per function.
//
// Floating point: built with -ffp-contract=off; every expression below is
// evaluated in the written order with IEEE single/double operations, which is
// also what the HIP kernels do.
#pragma once
#include <cstdint>
#include <cmath>
#include <cfloat>
#include <vector>
#include <algorithm>
#include <cstring>

namespace orc {

// cvRound(double) / cvRound(float): nearest, ties to even (SSE cvtsd2si).
// Call sites: ORBextractor.cc:79,113,117-118,444,1157.
static inline int cvRound(double v) { return (int)std::nearbyint(v); }
static inline int cvRoundf(float v) { return (int)std::nearbyintf(v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

// cv::fastAtan2(y, x) in degrees [0,360). Call sites: ORBextractor.cc:101 and
// OpenCV's lsd.cpp (ll_angle, region_grow, get_theta).
static inline float fastAtan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

struct Img8 {
  int w = 0, h = 0;
  std::vector<uint8_t> d;
  Img8() {}
  Img8(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
  uint8_t* row(int y) { return d.data() + (size_t)y * w; }
  const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
  uint8_t at(int y, int x) const { return d[(size_t)y * w + x]; }
};

// BORDER_REFLECT_101 index (edge pixel not repeated). copyMakeBorder /
// GaussianBlur / Sobel default border. Call sites ORBextractor.cc:1115,1167,1172.
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

// cv::resize(src, dst, dsize, fx, fy, INTER_LINEAR) for CV_8UC1: 11-bit fixed
// point coefficients, horizontal pass to int, vertical pass
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
// `scale_x/scale_y` = source/dest ratio as OpenCV derives it:
//   dsize given  (ORBextractor.cc:1165): inv = (double)dst/src, scale = 1/inv
//   fx,fy given  (lsd.cpp resize(..., Size(), SCALE, SCALE)): scale = 1/fx.
// Coefficient tables as cv::resize builds them (imgproc/src/imgwarp.cpp, 3.3.1):
// the x table clamps at the borders (sx<0 -> sx=0,fx=0; sx>=w-1 -> sx=w-1,fx=0);
// the y table is NOT clamped, the row index is clipped when rows are fetched.
static inline void resizeLinearCoeffs(int ssize, int dsize, double scale, bool clampWeights,
                                      std::vector<int>& ofs, std::vector<short>& coef) {
  ofs.resize(dsize);
  coef.resize(2 * (size_t)dsize);
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cvFloor(f);
    f -= s;
    if (clampWeights) {
      if (s < 0) { f = 0; s = 0; }
      if (s >= ssize - 1) { f = 0; s = ssize - 1; }
    }
    ofs[d] = s;
    float c0 = 1.f - f, c1 = f;
    coef[2 * d] = (short)cvRoundf(c0 * 2048.f);      // saturate_cast<short>
    coef[2 * d + 1] = (short)cvRoundf(c1 * 2048.f);
  }
}

static inline int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

static inline void resizeLinear8u(const Img8& src, Img8& dst, int dw, int dh,
                                  double scale_x, double scale_y) {
  dst = Img8(dw, dh);
  std::vector<int> xofs, yofs;
  std::vector<short> alpha, beta;
  resizeLinearCoeffs(src.w, dw, scale_x, true, xofs, alpha);
  resizeLinearCoeffs(src.h, dh, scale_y, false, yofs, beta);
  std::vector<int> r0(dw), r1(dw);
  for (int dy = 0; dy < dh; ++dy) {
    int sy0 = clipi(yofs[dy], 0, src.h);
    int sy1 = clipi(yofs[dy] + 1, 0, src.h);
    const uint8_t* S0 = src.row(sy0);
    const uint8_t* S1 = src.row(sy1);
    for (int dx = 0; dx < dw; ++dx) {
      int sx = xofs[dx];
      int sx1 = std::min(sx + 1, src.w - 1);  // at the clamped edge alpha1 == 0
      int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
      r0[dx] = S0[sx] * a0 + S0[sx1] * a1;
      r1[dx] = S1[sx] * a0 + S1[sx1] * a1;
    }
    int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t* D = dst.row(dy);
    for (int dx = 0; dx < dw; ++dx) {
      int v = (((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2;
      D[dx] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
}

// cv::getGaussianKernel(n, sigma, CV_32F) then the 8-bit fixed-point
// conversion used by createSeparableLinearFilter for CV_8U smoothing kernels
// (convertTo(CV_32S, 256) = cvRound(k*256)).
static inline std::vector<int> gaussKernelFixed8(int n, double sigma) {
  std::vector<float> cf(n);
  double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    double t = std::exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  std::vector<int> k(n);
  for (int i = 0; i < n; ++i) {
    cf[i] = (float)(cf[i] * sum);
    k[i] = cvRound((double)cf[i] * 256.0);
  }
  return k;
}

// cv::GaussianBlur(src, dst, Size(n,n), sigma, sigma, BORDER_REFLECT_101) for
// CV_8UC1 in OpenCV 3.3.1 (filter-engine path, before the 3.4.1 rewrite):
// int row pass, int column pass, dst = saturate_u8((sum + 2^15) >> 16).
// Call sites: ORBextractor.cc:1115 (7x7, 2), binary_descriptor_custom.cpp:358
// (5x5, 1), lsd.cpp (7x7, 0.6).
static inline void gaussianBlur8u(const Img8& src, Img8& dst, int n, double sigma) {
  std::vector<int> k = gaussKernelFixed8(n, sigma);
  int r = n / 2;
  int w = src.w, h = src.h;
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; ++y) {
    const uint8_t* S = src.row(y);
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = -r; i <= r; ++i) s += k[i + r] * S[reflect101(x + i, w)];
      tmp[(size_t)y * w + x] = s;
    }
  }
  dst = Img8(w, h);
  for (int y = 0; y < h; ++y) {
    uint8_t* D = dst.row(y);
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = -r; i <= r; ++i) s += k[i + r] * tmp[(size_t)reflect101(y + i, h) * w + x];
      int v = (s + (1 << 15)) >> 16;
      D[x] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
}

// cv::Sobel(src, dst, CV_16S, dx, dy, 3) with BORDER_REFLECT_101, no scaling.
// Call sites: binary_descriptor_custom.cpp:395-396.
static inline void sobel3x3_16s(const Img8& src, std::vector<int16_t>& dxI, std::vector<int16_t>& dyI) {
  int w = src.w, h = src.h;
  dxI.assign((size_t)w * h, 0);
  dyI.assign((size_t)w * h, 0);
  for (int y = 0; y < h; ++y) {
    const uint8_t* r0 = src.row(reflect101(y - 1, h));
    const uint8_t* r1 = src.row(y);
    const uint8_t* r2 = src.row(reflect101(y + 1, h));
    for (int x = 0; x < w; ++x) {
      int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      int gx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
      int gy = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
      dxI[(size_t)y * w + x] = (int16_t)gx;
      dyI[(size_t)y * w + x] = (int16_t)gy;
    }
  }
}

// FAST-9/16 ring, OpenCV order (makeOffsets for patternSize 16).
static const int kFastRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                                     {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                     {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};

// "arc value" of a pixel: max over the 16 arcs of 9 contiguous ring pixels and
// both polarities of the minimum |centre - ring| difference with a consistent
// sign.  The pixel is a FAST-9 corner at threshold t iff arc > t, and OpenCV's
// cornerScore<16> then returns arc - 1 (cv::FAST, features2d/src/fast.cpp and
// fast_score.cpp in 3.3.1).  Call sites: ORBextractor.cc:808,827.
static inline int fastArcValue(const uint8_t* p, int stride) {
  int d[25];
  int v = p[0];
  for (int k = 0; k < 16; ++k) d[k] = v - p[kFastRing[k][1] * stride + kFastRing[k][0]];
  for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
  int best = -256;
  for (int k = 0; k < 16; ++k) {
    int mn = d[k], mx = d[k];
    for (int j = 1; j < 9; ++j) { mn = std::min(mn, d[k + j]); mx = std::max(mx, d[k + j]); }
    best = std::max(best, mn);     // all ring pixels darker than centre by at least mn
    best = std::max(best, -mx);    // all ring pixels brighter than centre by at least -mx
  }
  return best;
}

// cv::LineIterator(img, pt1, pt2).count for 8-connectivity with both points
// inside the image: max(|dx|,|dy|) + 1 on cvRound-ed endpoints.
// Call site: LSDDetector_custom.cpp:295-296.
static inline int lineIteratorCount(float x1, float y1, float x2, float y2) {
  int ix1 = cvRoundf(x1), iy1 = cvRoundf(y1), ix2 = cvRoundf(x2), iy2 = cvRoundf(y2);
  int dx = std::abs(ix2 - ix1), dy = std::abs(iy2 - iy1);
  return std::max(dx, dy) + 1;
}

// ---------------------------------------------------------------------------
// cv::remap(src, dst, map1 (CV_32FC1 x), map2 (CV_32FC1 y), INTER_LINEAR), 8U, BORDER_CONSTANT 0 —
// the stereo driver's rectification (Examples/Stereo/stereo_euroc.cc:166-167).  OpenCV 3.3.1
// imgwarp.cpp as recalled (parity unpinned): the float maps are converted per pixel to fixed point
// with cvRound(v * INTER_TAB_SIZE) (INTER_BITS = 5), integer part saturated to short; the bilinear
// weights come from initInterTab2D: float products of (1 - f/32, f/32), times 2^15,
// saturate_cast<short>; a block that does not sum to 2^15 is repaired by adding the difference to
// its largest (or smallest) entry — which happens exactly for f = (0,0), where 32768 saturates to
// 32767 and the search (whose k1,k2 loops start at ksize/2 = 1) ends on entry [1][1]: {32767,0,0,1}.
// Result: FixedPtCast<int, uchar, 15>: saturate_u8((sum + 2^14) >> 15).
// ---------------------------------------------------------------------------
static inline void remapBilinearWeights(int fx, int fy, int w[4]) {
  const float sc = 1.f / 32;
  const float vx[2] = {1.f - fx * sc, fx * sc}, vy[2] = {1.f - fy * sc, fy * sc};
  int isum = 0;
  for (int k1 = 0; k1 < 2; ++k1)
    for (int k2 = 0; k2 < 2; ++k2) {
      float v = vy[k1] * vx[k2] * 32768.f;
      int iv = cvRoundf(v);
      iv = iv > 32767 ? 32767 : (iv < -32768 ? -32768 : iv);
      w[k1 * 2 + k2] = iv;
      isum += iv;
    }
  if (isum != 32768) {
    const int diff = isum - 32768;
    // entries [1][1], [1][2], [2][1], [2][2] of the reference's search lie in this block only for [1][1]; the
    // others belong to the not yet initialised next block (zero)
    int M = w[3] > 0 ? w[3] : 0, m = w[3] < 0 ? w[3] : 0;
    (void)M; (void)m;
    if (diff < 0) { if (w[3] >= 0) w[3] -= diff; }   // [1][1] is the maximum of {w11, 0, 0, 0}
    else { if (w[3] <= 0) w[3] -= diff; }
  }
}

static inline void remapLinear8u(const Img8& src, const float* mapx, const float* mapy, Img8& dst, int dw, int dh) {
  dst = Img8(dw, dh);
  const int W = src.w, H = src.h;
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      const int sx = cvRoundf(mapx[(size_t)y * dw + x] * 32), sy = cvRoundf(mapy[(size_t)y * dw + x] * 32);
      auto sat = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
      const int ix = sat(sx >> 5), iy = sat(sy >> 5);
      int w[4];
      remapBilinearWeights(sx & 31, sy & 31, w);
      int out = 0;
      if (!(ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0)) {
        auto at = [&](int xx, int yy) -> int { return (xx >= 0 && yy >= 0 && xx < W && yy < H) ? src.at(yy, xx) : 0; };
        const int sum = at(ix, iy) * w[0] + at(ix + 1, iy) * w[1] + at(ix, iy + 1) * w[2] + at(ix + 1, iy + 1) * w[3];
        out = (sum + (1 << 14)) >> 15;
        out = out < 0 ? 0 : (out > 255 ? 255 : out);
      }
      dst.row(y)[x] = (uint8_t)out;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// CV_64F variants for OpenCV 3.3.x's LineSegmentDetector, which converts its input to CV_64FC1 before anything else
// (lsd.cpp, LineSegmentDetectorImpl::detect: `Mat_<double> img = _image.getMat(); img.convertTo(image, CV_64FC1);`;
// the CV_8UC1 form belongs to the detector that was re-added in 4.5.x).  OpenCV-3.3.1-compatible by intent, parity unpinned.
// ---------------------------------------------------------------------------------------------------------------
struct Img64 {
  int w = 0, h = 0;
  std::vector<double> d;
  Img64() {}
  Img64(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
  double* row(int y) { return d.data() + (size_t)y * w; }
  const double* row(int y) const { return d.data() + (size_t)y * w; }
};

// cv::getGaussianKernel(n, sigma, CV_64F) (ktype = max(depth, CV_32F) = CV_64F for a CV_64F image)
static inline std::vector<double> gaussKernel64(int n, double sigma) {
  std::vector<double> cd(n);
  const double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    cd[i] = std::exp(scale2X * x * x);
    sum += cd[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) cd[i] *= sum;
  return cd;
}

// cv::GaussianBlur on CV_64FC1 = sepFilter2D with the filter engine's generic double filters:
//   RowFilter<double,double>: s = kx[0]*S[0]; s += kx[k]*S[k] for k = 1..n-1 (left to right);
//   SymmColumnFilter<Cast<double,double>> (symmetric kernel): s = ky[c]*S[c]; s += ky[c+k]*(S[c+k] + S[c-k]) for k = 1..r.
// BORDER_REFLECT_101.  Input: the u8 image converted to double.
static inline void gaussianBlur64f(const Img8& src, Img64& dst, int n, double sigma) {
  const std::vector<double> k = gaussKernel64(n, sigma);
  const int r = n / 2, w = src.w, h = src.h;
  Img64 tmp(w, h);
  for (int y = 0; y < h; ++y) {
    const uint8_t* S = src.row(y);
    double* T = tmp.row(y);
    for (int x = 0; x < w; ++x) {
      double s = k[0] * (double)S[reflect101(x - r, w)];
      for (int i = 1; i < n; ++i) s += k[i] * (double)S[reflect101(x - r + i, w)];
      T[x] = s;
    }
  }
  dst = Img64(w, h);
  for (int y = 0; y < h; ++y) {
    double* D = dst.row(y);
    for (int x = 0; x < w; ++x) {
      double s = k[r] * tmp.row(y)[x];
      for (int i = 1; i <= r; ++i) s += k[r + i] * (tmp.row(reflect101(y + i, h))[x] + tmp.row(reflect101(y - i, h))[x]);
      D[x] = s;
    }
  }
}

// cv::resize INTER_LINEAR on CV_64FC1: HResizeLinear<double,double,float> / VResizeLinear<double,double,float,Cast>: float
// coefficients (1-f, f), double arithmetic: t = S[sx]*a0 + S[sx+1]*a1, dst = T0*b0 + T1*b1.  Index / weight tables as for
// the 8-bit path (resizeLinearCoeffs), without the fixed-point conversion.
static inline void resizeLinear64f(const Img64& src, Img64& dst, int dw, int dh, double scale_x, double scale_y) {
  dst = Img64(dw, dh);
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<float> alpha(2 * (size_t)dw), beta(2 * (size_t)dh);
  for (int d = 0; d < dw; ++d) {
    float f = (float)((d + 0.5) * scale_x - 0.5);
    int s = cvFloor(f);
    f -= s;
    if (s < 0) { f = 0; s = 0; }
    if (s >= src.w - 1) { f = 0; s = src.w - 1; }
    xofs[d] = s; alpha[2 * d] = 1.f - f; alpha[2 * d + 1] = f;
  }
  for (int d = 0; d < dh; ++d) {
    float f = (float)((d + 0.5) * scale_y - 0.5);
    int s = cvFloor(f);
    f -= s;
    yofs[d] = s; beta[2 * d] = 1.f - f; beta[2 * d + 1] = f;
  }
  for (int dy = 0; dy < dh; ++dy) {
    const double* S0 = src.row(clipi(yofs[dy], 0, src.h));
    const double* S1 = src.row(clipi(yofs[dy] + 1, 0, src.h));
    const double b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    double* D = dst.row(dy);
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = xofs[dx], sx1 = std::min(sx + 1, src.w - 1);
      const double a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
      const double t0 = S0[sx] * a0 + S0[sx1] * a1;
      const double t1 = S1[sx] * a0 + S1[sx1] * a1;
      D[dx] = t0 * b0 + t1 * b1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// cosf / sinf as glibc >= 2.28 computes them (sysdeps/ieee754/flt-32/s_sincosf.h, the ARM optimized-routines algorithm:
// reduction and a degree-7/8 polynomial in double, one rounding to float) — what `cos(float)` gives where the float
// overload is selected (e.g. `using namespace std;` in ORBextractor.cc:65 with the float `angle` of :110-111).
// Checked bit for bit against this image's glibc 2.35 on every third float of [2^-13, 2 pi) by tests/test_oracle_kat.py.
// Valid for |x| < 120 (the angles here are in [0, 2 pi)).
// ---------------------------------------------------------------------------------------------------------------
struct SinCosfTab { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
static const SinCosfTab kSinCosf[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static inline uint32_t f32bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline uint32_t abstop12(float x) { return (f32bits(x) >> 20) & 0x7ff; }
static inline float sinfPoly(double x, double x2, const SinCosfTab* p, int n) {
  if ((n & 1) == 0) {
    const double x3 = x * x2, s1 = p->s2 + x2 * p->s3, x7 = x3 * x2, s = x + x3 * p->s1;
    return (float)(s + x7 * s1);
  }
  const double x4 = x2 * x2, c2 = p->c3 + x2 * p->c4, c1 = p->c0 + x2 * p->c1, x6 = x4 * x2, c = c1 + x4 * p->c2;
  return (float)(c + x6 * c2);
}
static inline double sincosfReduce(double x, const SinCosfTab* p, int* np) {
  const double r = x * p->hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return x - n * p->hpi;
}
static inline float glibcSinf(float y) {
  double x = y;
  int n;
  const SinCosfTab* p = &kSinCosf[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return sinfPoly(x, x * x, p, 0);
  }
  x = sincosfReduce(x, p, &n);
  const double s = p->sign[n & 3];
  if (n & 2) p = &kSinCosf[1];
  return sinfPoly(x * s, x * x, p, n);
}
static inline float glibcCosf(float y) {
  double x = y;
  int n;
  const SinCosfTab* p = &kSinCosf[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return sinfPoly(x, x * x, p, 1);
  }
  x = sincosfReduce(x, p, &n);
  const double s = p->sign[(n + 1) & 3];
  if ((n + 1) & 2) p = &kSinCosf[1];
  return sinfPoly(x * s, x * x, p, n ^ 1);
}
// cos / sin of a float as the call site selects them: f32 = the float overload (glibc >= 2.28), else double libm + cast
static inline float cosOfFloat(float a, bool f32) { return f32 ? glibcCosf(a) : (float)std::cos((double)a); }
static inline float sinOfFloat(float a, bool f32) { return f32 ? glibcSinf(a) : (float)std::sin((double)a); }

}  // namespace orc
