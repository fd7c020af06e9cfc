// region2rect of the relaxations (shared by lsd_relax.hip: k_rx_rect, and lsd_tile.hip: the persistent tail kernel).
#pragma once
#include "kernels.hpp"
#include "device_prims.hpp"

namespace pli {

constexpr double RX_PI = 3.14159265358979323846;
constexpr double RX_DEG2RAD = RX_PI / 180;
constexpr double RX_3_2_PI = (3 * RX_PI) / 2;
constexpr double RX_2PI = 2 * RX_PI;

__device__ __forceinline__ double rx_angle_diff(double a, double b) {
  double diff = a - b;
  while (diff <= -RX_PI) diff += RX_2PI;
  while (diff > RX_PI) diff -= RX_2PI;
  return fabs(diff);
}

// ---- region2rect of the regions completed in this round (lsd.cpp region2rect / get_theta) --------------
// The running sums are accumulated in list order by three lanes (bit-exact with the sequential loop), the products are
// computed a row of lanes at a time.  Most regions are small (46 pixels on average at 752x480), so a wave takes FOUR list
// entries at a time, 16 lanes each (the three serial sums of the four regions run side by side); a region of more than
// RX_RECT_GROUP_MAX pixels is left to the whole wave (LANES = 64), which does it right after its group of four.
constexpr int RX_RECT_GROUP_MAX = 192;
constexpr int RX_RECT_CACHE = 4;          // chunks of a region's list whose weights stay in LDS between the passes

__device__ __forceinline__ void rx_wave_sync() {
  // single-wave workgroups: the LDS operations of a wave execute in order, the compiler only has to keep the order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LANES lanes (a whole wave, or an aligned group of 16) compute the segment of one region; `on` = this group has one
template <int LANES>
__device__ __forceinline__ void rx_rect_region(bool on, const RxRect& it, const DevParams& P, const float4* __restrict__ rec,
                                               const double* __restrict__ mg, const int* __restrict__ arena,
                                               float4* __restrict__ rgSeg, double (*st)[64], double (*wc)[64], int (*ec)[64], int rmask,
                                               const int2* __restrict__ hot = nullptr /* the 8-byte hot records {angle, ..}, or null */,
                                               const float2* __restrict__ cold = nullptr /* ... and the exact {cos, sin} beside them, or null: rec */) {
  // wc / ec: the weights and the packed coordinates of the first RX_RECT_CACHE chunks of the list, kept in LDS by the first pass: the
  // second pass and the end-point pass read them there instead of walking list entry -> weight through global memory again (two
  // dependent round trips per chunk and pass; most regions fit the cache: 46 pixels on average)
  const int lane = threadIdx.x & 63, gl = lane & (LANES - 1), g0 = lane & ~(LANES - 1);
  const int W = P.LW;
  const double prec = P.prec;
  const int cnt = on ? it.cnt : 0;
  const int* lst = arena + it.off;
  const double reg_angle = (double)fast_atan2_deg(it.sumdy, it.sumdx) * RX_DEG2RAD;
  // (the trip count of the group with the longest list: the passes below run in step over the wave)
  int cmax = cnt;
#pragma unroll
  for (int o = 32; o >= LANES; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o, 64));
  double acc = 0.0;
  for (int c0 = 0; c0 < cmax; c0 += LANES) {
    const int kk = c0 + gl;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;                  // lanes past the end add +0.0 (the sums are never -0.0)
    if (kk < cnt) {
      const int e = lst[kk];
      const int ex = e & 0xFFFF, ey = e >> 16;
      const double w = mg ? mg[ey * W + ex] : sqrt((double)__float_as_int(rec[ey * W + ex].w) / 4.0);
      v0 = (double)ex * w;
      v1 = (double)ey * w;
      v2 = w;
      if (c0 < RX_RECT_CACHE * LANES) { wc[c0 / LANES][lane] = w; ec[c0 / LANES][lane] = e; }
    }
    rx_wave_sync();
    st[0][lane] = v0; st[1][lane] = v1; st[2][lane] = v2;
    rx_wave_sync();
    if (gl < 3 && c0 < cnt) {
      // list-order sum, eight terms per trip: the LDS reads of a trip are issued together, the adds stay in order
      const int mm = (min(LANES, cnt - c0) + 7) & ~7;
      for (int tt = 0; tt < mm; tt += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = st[gl][g0 + tt + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += a[u];
      }
    }
  }
  // (the two centroid divisions are one division in lanes 0 and 1, the four end-point divisions one in lanes 0..3)
  const double cq = acc / __shfl(acc, g0 + 2, 64);
  const double x = __shfl(cq, g0, 64), y = __shfl(cq, g0 + 1, 64);
  acc = 0.0;
  for (int c0 = 0; c0 < cmax; c0 += LANES) {
    const int kk = c0 + gl;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;                  // past the end: acc + 0.0 and acc - 0.0 leave acc as it is
    if (kk < cnt) {
      const bool cached = c0 < RX_RECT_CACHE * LANES;
      const int e = cached ? ec[c0 / LANES][lane] : lst[kk];
      const int ex = e & 0xFFFF, ey = e >> 16;
      const double w = cached ? wc[c0 / LANES][lane] : (mg ? mg[ey * W + ex] : sqrt((double)__float_as_int(rec[ey * W + ex].w) / 4.0));
      const double dx = (double)ex - x, dy = (double)ey - y;
      v0 = dy * dy * w;
      v1 = dx * dx * w;
      v2 = dx * dy * w;
    }
    rx_wave_sync();
    st[0][lane] = v0; st[1][lane] = v1; st[2][lane] = v2;
    rx_wave_sync();
    if (gl < 3 && c0 < cnt) {
      const int mm = (min(LANES, cnt - c0) + 7) & ~7;
      for (int tt = 0; tt < mm; tt += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = st[gl][g0 + tt + u];
        if (gl < 2) {
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += a[u];
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) acc -= a[u];
        }
      }
    }
  }
  const double Ixx = __shfl(acc, g0, 64), Iyy = __shfl(acc, g0 + 1, 64), Ixy = __shfl(acc, g0 + 2, 64);
  const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
  const bool wide = fabs(Ixx) > fabs(Iyy);
  double theta = (double)fast_atan2_deg(wide ? (float)(lambda - Ixx) : (float)Ixy, wide ? (float)Ixy : (float)(lambda - Iyy));
  theta *= RX_DEG2RAD;
  double adiff = rx_angle_diff(theta, reg_angle);
  // A region grown on the hot records (lsd_tile.hip) hands over the sums of its vector filter: cos / sin from v_cos_f32 / v_sin_f32,
  // within 2e-4 rad of the exact sums in direction (5e-4 rad through fastAtan2's octant seams).  The region angle has this ONE use — on
  // which side of `prec` the inertia axis lies —, so the approximate one decides unless the difference is within rectApproxBand (2e-3 rad)
  // of prec; there (one region in a thousand) the group sums the exact cos / sin of the 16-byte records in list order, as region_grow does.
  {
    const bool need = on && it.approx != 0 && hot != nullptr && fabs(adiff - prec) < P.rectApproxBand;
    if (__builtin_amdgcn_ballot_w64(need)) {
      int cm = need ? cnt : 0;
#pragma unroll
      for (int o = 32; o >= LANES; o >>= 1) cm = max(cm, __shfl_xor(cm, o, 64));
      float sx = 0.f, sy = 0.f;
      for (int c0 = 0; c0 < cm; c0 += LANES) {
        const int kk = c0 + gl;
        float cy = 0.f, cz = 0.f;
        if (need && kk < cnt) {
          const int e = lst[kk];
          // (approx is only ever set by the growers of the hot records: the angle comes from there, the exact pair from the cold plane —
          // or from the 16-byte record where the front pass still writes it)
          const int p = (e >> 16) * W + (e & 0xFFFF);
          if (kk == 0) {                                  // the seed: region_grow starts its sums with cos / sin of the unrounded double angle
            double sn, cs;
            sincos((double)__int_as_float(hot[p].x) * RX_DEG2RAD, &sn, &cs);
            cy = (float)cs; cz = (float)sn;
          } else if (cold) {
            const float2 cp = cold[p];
            cy = cp.x; cz = cp.y;
          } else {
            const float4 r = rec[p];
            cy = r.y; cz = r.z;
          }
        }
        for (int j = 0; j < LANES; ++j) {
          const float vy = __shfl(cy, g0 + j, 64), vz = __shfl(cz, g0 + j, 64);
          if (need && c0 + j < cnt) {
            sx = (c0 + j == 0) ? vy : __fadd_rn(sx, vy);
            sy = (c0 + j == 0) ? vz : __fadd_rn(sy, vz);
          }
        }
      }
      if (need) adiff = rx_angle_diff(theta, (double)fast_atan2_deg(sy, sx) * RX_DEG2RAD);
    }
  }
  if (adiff > prec) theta += RX_PI;
  double dxr, dyr;
  sincos(theta, &dyr, &dxr);
  double l_min = 0, l_max = 0;
  for (int kk = gl; kk < cnt; kk += LANES) {
    const int e = kk < RX_RECT_CACHE * LANES ? ec[kk / LANES][lane] : lst[kk];
    const double l = ((double)(e & 0xFFFF) - x) * dxr + ((double)(e >> 16) - y) * dyr;
    l_max = fmax(l_max, l);
    l_min = fmin(l_min, l);
  }
#pragma unroll
  for (int o = LANES / 2; o > 0; o >>= 1) {
    l_max = fmax(l_max, __shfl_xor(l_max, o, 64));
    l_min = fmin(l_min, __shfl_xor(l_min, o, 64));
  }
  // lanes 0..3: x1, y1, x2, y2
  double e = ((gl & 1) ? y : x) + ((gl & 2) ? l_max : l_min) * ((gl & 1) ? dyr : dxr);
  e += 0.5;
  const double scale = P.lsdScale;
  if (scale != 1) e /= scale;
  if (on && gl < 4) reinterpret_cast<float*>(&rgSeg[it.rank & rmask])[gl] = (float)e;
}


// One wave's share of the round's rect list of one image: list entries first, first + stride, ... four at a time.
__device__ __forceinline__ void rx_rect_wave(const DevParams& P, const RxCtl& c, const float4* __restrict__ rec, const double* __restrict__ mg,
                                             const int* __restrict__ arena, const RxRect* __restrict__ rects, int rectCap,
                                             float4* __restrict__ rgSeg, int first, int stride, double (*st)[64], double (*wc)[64],
                                             int (*ec)[64], int rmask = -1, const int2* __restrict__ hot = nullptr,
                                             const float2* __restrict__ cold = nullptr) {
  const int nrect = min((int)(c.rectArena >> RX_ARENA_BITS), rectCap);
  const int lane = threadIdx.x & 63, g = lane >> 4;
  for (int w0 = first * 4; w0 < nrect; w0 += stride * 4) {
    RxRect it = {0, 0, 0, 1.f, 0.f, 0};
    const bool have = w0 + g < nrect;
    if (have) it = rects[w0 + g];
    const bool small = have && it.cnt <= RX_RECT_GROUP_MAX;
    if (__builtin_amdgcn_ballot_w64(small)) rx_rect_region<16>(small, it, P, rec, mg, arena, rgSeg, st, wc, ec, rmask, hot, cold);
    unsigned long long big = __builtin_amdgcn_ballot_w64(have && !small) & 0x0001000100010001ull;   // one bit per group
    while (big) {
      const int gl0 = __ffsll((long long)big) - 1;
      big &= big - 1ull;
      RxRect bt;
      bt.rank = __shfl(it.rank, gl0, 64); bt.off = __shfl(it.off, gl0, 64); bt.cnt = __shfl(it.cnt, gl0, 64);
      bt.sumdx = __shfl(it.sumdx, gl0, 64); bt.sumdy = __shfl(it.sumdy, gl0, 64); bt.approx = __shfl(it.approx, gl0, 64);
      rx_rect_region<64>(true, bt, P, rec, mg, arena, rgSeg, st, wc, ec, rmask, hot, cold);
    }
  }
}

}  // namespace pli
