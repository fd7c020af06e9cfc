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

}  // namespace pli
