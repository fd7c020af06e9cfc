// Device-side arithmetic primitives shared by the kernels.  Every float
// expression here must round exactly like the CPU path of the reference
// (IEEE single/double, no fused multiply-add): the library is compiled with
// -ffp-contract=off and these helpers use explicit _rn intrinsics where the
// order matters.
#pragma once
#include <hip/hip_runtime.h>
// NOTE: HIP's __fsqrt_rn is the *native* (approximate) square root; use sqrtf()/sqrt(), which are
// correctly rounded under the default -fhip-fp32-correctly-rounded-divide-sqrt.  __fadd_rn/__fmul_rn/
// __fdiv_rn are plain IEEE operations and rely on -ffp-contract=off to stay unfused.
#pragma clang fp contract(off)
#include <cfloat>
#include <cstdint>
#include "common.hpp"

namespace pli {

// cvRound(float): nearest, ties to even (OpenCV uses cvtss2si).
__device__ __forceinline__ int cv_round_f(float v) { return __float2int_rn(v); }

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

// cv::fastAtan2 (degrees, [0,360)); polynomial evaluated in the written order.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float ax = fabsf(x), ay = fabsf(y);
  // one division for both octant cases (the reference's two branches compute min/(max + eps) either way)
  const bool steep = !(ax >= ay);
  const float mn = steep ? ax : ay, mx = steep ? ay : ax;
  const float c = __fdiv_rn(mn, __fadd_rn(mx, (float)DBL_EPSILON));
  const float c2 = __fmul_rn(c, c);
  float a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  if (steep) a = __fsub_rn(90.f, a);
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// 256-bit Hamming distance of two 32-byte descriptors held as 4 x u64.
__device__ __forceinline__ int hamming256(const uint64_t a[4], const uint64_t b[4]) {
  return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide min of a 64-bit key (all lanes get the result)
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}


// cosf / sinf as glibc >= 2.28 computes them (the algorithm of ARM's optimized routines: reduction and polynomial in
// double, one rounding to float; tests/test_oracle_kat.py checks the same restatement against the host libm).
// Used where the reference's unqualified cos(float) selects the float overload (PLI_PARITY_TRIG_F32_*).  |x| < 120.
struct SinCosfTab { double c0, c1, c2, c3, c4, s1, s2, s3; };
__device__ __forceinline__ float glibc_sinf_poly(double x, double x2, bool neg, int n) {
  // table 0 / table 1 of glibc's __sincosf_table differ in the sign of the cosine coefficients
  const double sg = neg ? -1.0 : 1.0;
  if ((n & 1) == 0) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2, s1 = s2c + x2 * s3c, x7 = x3 * x2, s = x + x3 * s1c;
    return (float)(s + x7 * s1);
  }
  const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5, c3c = sg * -0x1.6c087e89a359dp-10,
               c4c = sg * 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2, c2 = c3c + x2 * c4c, c1 = c0 + x2 * c1c, x6 = x4 * x2, c = c1 + x4 * c2c;
  return (float)(c + x6 * c2);
}
__device__ __forceinline__ unsigned glibc_abstop12(float x) { return ((unsigned)__float_as_int(x) >> 20) & 0x7ffu; }
__device__ __forceinline__ void glibc_sincosf(float y, float* sn, float* cs) {
  const double x = (double)y;
  if (glibc_abstop12(y) < glibc_abstop12(0x1.921FB6p-1f)) {
    if (glibc_abstop12(y) < glibc_abstop12(0x1p-12f)) { *sn = y; *cs = 1.0f; return; }
    const double x2 = x * x;
    *sn = glibc_sinf_poly(x, x2, false, 0);
    *cs = glibc_sinf_poly(x, x2, false, 1);
    return;
  }
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int)r + 0x800000) >> 24;
  const double xr = x - (double)n * 0x1.921FB54442D18p0;
  const double x2 = xr * xr;
  // sine: sign[n & 3], table by (n & 2);  cosine: sign[(n + 1) & 3], table by ((n + 1) & 2), polynomial n ^ 1
  const double ss = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  const double sc = (((n + 1) & 3) == 1 || ((n + 1) & 3) == 2) ? -1.0 : 1.0;
  *sn = glibc_sinf_poly(xr * ss, x2, (n & 2) != 0, n);
  *cs = glibc_sinf_poly(xr * sc, x2, ((n + 1) & 2) != 0, n ^ 1);
}
// cos / sin of a float as the call site selects them (see include/pli_frontend.h, PLI_PARITY_TRIG_F32_*)
__device__ __forceinline__ void sincos_of_float(float a, bool f32, float* sn, float* cs) {
  if (f32) glibc_sincosf(a, sn, cs);
  else {
#if defined(PLI_DIAG_NOSINCOS) && defined(PLI_DEV)      // (diagnostic development build, wrong results: what the double sincos costs the front pass)
    *sn = a * 0.25f; *cs = 1.f - a * 0.125f;
#else
    double s, c;
    sincos((double)a, &s, &c);
    *sn = (float)s;
    *cs = (float)c;
#endif
  }
}

// (tile-sequential relaxation) the region of rank o was just stamped dirty for this round — the caller won the stamp, so this
// runs once per region and round: its seed goes on the list of the seed's tile (at most ts * ts seeds per tile: no overflow)
__device__ __forceinline__ void tx_dirty_append(const TxDirtyLists& DL, int img, int o) {
  if (!DL.list) return;
  const int sp = DL.rmask != -1 ? (o & DL.rmask) : DL.order[(int64_t)img * DL.npix + o];
  const int sy = sp / DL.W, sx = sp - sy * DL.W;
  const int64_t tile = (int64_t)img * DL.ntx * DL.nty + (sy / DL.ts) * DL.ntx + sx / DL.ts;
  const int slot = atomicAdd(&DL.cnt[tile], 1);
  // (an image that ran out of a capacity is left by its growers, which then no longer take — and clear — its lists, while the
  // bookkeeping passes of the remaining rounds may still stamp: nothing is written past a tile's list)
  if (slot < DL.ts * DL.ts) DL.list[tile * DL.ts * DL.ts + slot] = make_int2(o, sp);
}

// CV_64F pipeline (lsd_f64.hip): the gradient norm is a double plane, its maximum is kept as the bits of a double
__device__ __forceinline__ double lsd_bin_coef64(unsigned long long maxBits, int nBins) {
  const double maxGrad = __longlong_as_double((long long)maxBits);
  return maxBits ? (double)(nBins - 1) / maxGrad : 0.0;
}
__device__ __forceinline__ int lsd_bin64(double norm, double binCoef, int nBins) { return min((int)(norm * binCoef), nBins - 1); }

// Packed round 1 of the tile relaxation (lsd_tile.hip): the fourth word of a pixel record is the pixel's owner word.  Bit 31 set:
// nobody has claimed the pixel; the low bits then carry either the pixel's own id (written by k_tx_sort) or, in the LAZY form, its
// gradient norm in 2^-22 fixed point (written by the front pass, which does not know the ids yet) — enough to order the pixel's
// gradient bin against a region's without the double plane, except within a few units of a bin boundary (lsd_tile.hip decides those
// from the double).  Bit 31 clear: the id of the region that holds the pixel (claims are unsigned atomic minima).
constexpr unsigned TX_UNCLAIMED = 0x80000000u;
constexpr double TX_FIX22 = 4194304.0;
__device__ __forceinline__ float tx_unclaimed_norm_word(double norm) {
  return __int_as_float((int)(TX_UNCLAIMED | (unsigned)fmin(norm * TX_FIX22, 2147418112.0 /* 0x7FFF0000 */)));
}

}  // namespace pli
