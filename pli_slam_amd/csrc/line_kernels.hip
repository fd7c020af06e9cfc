// Line front-end kernels for gfx950 (wave64): LSD segment detection + LBD.
//
//   (k_blur, k_resize_level from orb_kernels.hip do LSD's 7x7 sigma-0.6 blur and x1.2 resize)
//   k_lsd_grad      ll_angle: 2x2 gradient, level-line angle, max gradient   (OpenCV lsd.cpp)
//   k_lsd_hist/scan/scatter   the 1024-bin ordering of seed pixels (bin desc, raster asc)
//   k_lsd_grow      region_grow + region2rect, refine = NONE
//   k_keylines      LSDDetectorC::detectImpl KeyLine fill + Lineextractor top-N
//                   (LSDDetector_custom.cpp:264-308, LineExtractor.cc:53-65)
//   k_sobel         cv::Sobel 3x3 dx/dy -> i16            (binary_descriptor_custom.cpp:395-396)
//   k_lbd           BinaryDescriptor::computeLBD + binaryConversion
//                   (binary_descriptor_custom.cpp:1026-1340, 401-412, 662-666)
#include "kernels.hpp"
#include "device_prims.hpp"
#include <type_traits>

namespace pli {

constexpr float LSD_NOTDEF = -1024.f;
constexpr double D_PI = 3.14159265358979323846;
constexpr double D_DEG2RAD = D_PI / 180;
constexpr double D_3_2_PI = (3 * D_PI) / 2;
constexpr double D_2PI = 2 * D_PI;

// ---------------------------------------------------------------------------
// k_lsd_grad.  One thread per pixel of the scaled image.  Per pixel it writes
//   rec = { ang, c, s, g2 } (16 bytes, what the region grower needs in one load):
//     ang : level-line angle in degrees (cv::fastAtan2) or -1024 (NOTDEF); the
//           region grower overwrites claimed pixels with -1024 (= USED)
//     c,s : (float)cos / (float)sin of (float)angle_rad, what region_grow adds per pixel
//     g2  : gx^2+gy^2 as int bits (modgrad = sqrt(g2/4))
//   g2o : the same g2 as a plain int plane for the bin-ordering kernels.
// ---------------------------------------------------------------------------
constexpr int LSD_GRAD_ROWS = 16;   // rows per block: one atomicMax per block (same-line atomics serialise in L2)

__global__ __launch_bounds__(256) void k_lsd_grad(const uint8_t* __restrict__ scaled, int64_t imgStride, int W, int H,
                                                  int WP /* row pitch of the planes written (>= W: the pad columns are undefined pixels) */,
                                                  int pitch, int g2Thresh, float4* __restrict__ rec,
                                                  int* __restrict__ g2o, int2* __restrict__ own,
                                                  int* __restrict__ maxG2, float* __restrict__ angDbg, int img0, int trigF32) {
  __shared__ int wmax[4];
  const int img = blockIdx.z + img0;
  const int x = blockIdx.x * 256 + threadIdx.x;
  int m = 0;
  if (x < WP) {
    const int yEnd = min((int)(blockIdx.y + 1) * LSD_GRAD_ROWS, H);
    for (int y = blockIdx.y * LSD_GRAD_ROWS; y < yEnd; ++y) {
      int g2 = 0;
      float a = LSD_NOTDEF;
      float cx = 0.f, sy = 0.f;
      if (x < W - 1 && y < H - 1) {
        const uint8_t* r0 = scaled + (int64_t)img * imgStride + (int64_t)y * pitch;
        const uint8_t* r1 = r0 + pitch;
        int DA = (int)r1[x + 1] - (int)r0[x];
        int BC = (int)r0[x + 1] - (int)r1[x];
        int gx = DA + BC, gy = DA - BC;
        g2 = gx * gx + gy * gy;
        if (g2 > g2Thresh) {
          m = max(m, g2);
          a = fast_atan2_deg((float)gx, (float)(-gy));
          // cos(float(angle)), sin(float(angle)) of lsd.cpp region_grow (PLI_PARITY_TRIG_F32_LSD: which overload)
          sincos_of_float((float)((double)a * D_DEG2RAD), trigF32 != 0, &sy, &cx);
        }
      }
      const int64_t o = (int64_t)img * WP * H + (int64_t)y * WP + x;
      rec[o] = make_float4(a, cx, sy, __int_as_float(g2));
      g2o[o] = g2;
      if (own) own[o] = make_int2(0x7FFFFFFF, 0x7FFFFFFF);   // owner planes of the relaxation (lsd_relax.hip): undefined pixels never belong to a region
      if (angDbg && x < W) angDbg[(int64_t)img * W * H + (int64_t)y * W + x] = a;      // (the debug plane: rows of the true width)
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    if (m > 0) atomicMax(&maxG2[img], m);
  }
}

__device__ __forceinline__ int lsd_bin(int g2, double binCoef) {
  double norm = sqrt((double)g2 / 4.0);
  return (int)(norm * binCoef);
}
__device__ __forceinline__ double lsd_bin_coef(int maxG2, int nBins) {
  double maxGrad = sqrt((double)maxG2 / 4.0);
  return maxG2 > 0 ? (double)(nBins - 1) / maxGrad : 0.0;
}

// LDS traffic of ONE wave is executed in order: a write by some lanes followed by a read by others needs no s_barrier,
// only the compiler must keep the order
__device__ __forceinline__ void lsd_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// per-chunk histogram of the bins of the defined pixels
__global__ __launch_bounds__(256) void k_lsd_hist(const int* __restrict__ g2a, int npix, int g2Thresh, int nBins,
                                                  const int* __restrict__ maxG2, unsigned short* __restrict__ chunkHist,
                                                  int nChunks, int img0, const double* __restrict__ mgAll,
                                                  const unsigned long long* __restrict__ maxMg, double rho,
                                                  const RxCtl* __restrict__ onlyUnsettled) {
  __shared__ int h[1024];
  const int img = blockIdx.y + img0, chunk = blockIdx.x, tid = threadIdx.x;
  // (the ordered list only for the images the relaxation has left to the sequential grower: key mode of the tile relaxation)
  if (onlyUnsettled && onlyUnsettled[img].state == 2 && !onlyUnsettled[img].overflow) return;
  for (int i = tid; i < nBins; i += 256) h[i] = 0;
  __syncthreads();
  if (mgAll) {                                          // CV_64F pipeline
    const double bc = lsd_bin_coef64(maxMg[img], nBins);
    const double* g = mgAll + (int64_t)img * npix;
    // (the thread's loads unconditional, from a clamped index, and all of them before the first use: a load under a lane predicate is
    // waited for at the end of its branch — sixteen round trips one after the other)
    double v[LSD_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < LSD_CHUNK / 256; ++k) v[k] = g[min(chunk * LSD_CHUNK + k * 256 + tid, npix - 1)];
#pragma unroll
    for (int k = 0; k < LSD_CHUNK / 256; ++k) {
      const int i = chunk * LSD_CHUNK + k * 256 + tid;
      if (i < npix && !(v[k] <= rho)) atomicAdd(&h[lsd_bin64(v[k], bc, nBins)], 1);
    }
  } else {
    const double bc = lsd_bin_coef(maxG2[img], nBins);
    const int* g = g2a + (int64_t)img * npix;
    int v[LSD_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < LSD_CHUNK / 256; ++k) v[k] = g[min(chunk * LSD_CHUNK + k * 256 + tid, npix - 1)];
#pragma unroll
    for (int k = 0; k < LSD_CHUNK / 256; ++k) {
      const int i = chunk * LSD_CHUNK + k * 256 + tid;
      if (i < npix && v[k] > g2Thresh) atomicAdd(&h[lsd_bin(v[k], bc)], 1);
    }
  }
  __syncthreads();
  unsigned short* out = chunkHist + ((int64_t)img * nChunks + chunk) * nBins;
  for (int i = tid; i < nBins; i += 256) out[i] = (unsigned short)h[i];
}

// per image: chunkBase[chunk][bin] = start of that chunk's pixels of that bin in the ordered list
__global__ __launch_bounds__(1024) void k_lsd_scan(const unsigned short* __restrict__ chunkHist, int nChunks, int nBins,
                                                   int* __restrict__ chunkBase, int* __restrict__ nDefined, int img0,
                                                   const RxCtl* __restrict__ onlyUnsettled) {
  __shared__ int tot[1024];
  __shared__ int start[1024];
  const int img = blockIdx.x + img0, b = threadIdx.x;
  if (onlyUnsettled && onlyUnsettled[img].state == 2 && !onlyUnsettled[img].overflow) return;
  const unsigned short* hin = chunkHist + (int64_t)img * nChunks * nBins;
  int* cb = chunkBase + (int64_t)img * nChunks * nBins;
  int run = 0;
  if (b < nBins) {
#pragma unroll 8
    for (int c = 0; c < nChunks; ++c) {
      int v = hin[(int64_t)c * nBins + b];
      cb[(int64_t)c * nBins + b] = run;
      run += v;
    }
  }
  tot[b] = b < nBins ? run : 0;
  __syncthreads();
  if (b == 0) {
    int acc = 0;
    for (int k = nBins - 1; k >= 0; --k) { start[k] = acc; acc += tot[k]; }   // bins in descending order
    nDefined[img] = acc;
  }
  __syncthreads();
  if (b < nBins) {
    int s = start[b];
#pragma unroll 8
    for (int c = 0; c < nChunks; ++c) cb[(int64_t)c * nBins + b] += s;
  }
}

// The same for large images (thousands of chunks: one block per image is a long serial walk), in two kernels: a block per
// (image, group of chunks) makes the group-local prefix and the group's bin totals; a block per image then scans the groups
// per bin and writes, per (group, bin), the offset k_lsd_scatter adds to the group-local value when it loads a chunk's bases.
__global__ __launch_bounds__(1024) void k_lsd_scan_part(const unsigned short* __restrict__ chunkHist, int nChunks, int nBins,
                                                        int chunksPerGroup, int* __restrict__ chunkBase,
                                                        int* __restrict__ groupOff, int img0,
                                                        const RxCtl* __restrict__ onlyUnsettled) {
  const int img = blockIdx.x + img0, g = blockIdx.y, b = threadIdx.x;
  if (onlyUnsettled && onlyUnsettled[img].state == 2 && !onlyUnsettled[img].overflow) return;
  if (b >= nBins) return;
  const int c0 = g * chunksPerGroup, c1 = min(c0 + chunksPerGroup, nChunks);
  const unsigned short* hin = chunkHist + (int64_t)img * nChunks * nBins;
  int* cb = chunkBase + (int64_t)img * nChunks * nBins;
  int run = 0;
#pragma unroll 8
  for (int c = c0; c < c1; ++c) {
    const int v = hin[(int64_t)c * nBins + b];
    cb[(int64_t)c * nBins + b] = run;
    run += v;
  }
  groupOff[((int64_t)img * gridDim.y + g) * nBins + b] = run;
}
__global__ __launch_bounds__(1024) void k_lsd_scan_groups(int nGroups, int nBins, int* __restrict__ groupOff,
                                                          int* __restrict__ nDefined, int img0,
                                                          const RxCtl* __restrict__ onlyUnsettled) {
  __shared__ int tot[1024];
  __shared__ int start[1024];
  const int img = blockIdx.x + img0, b = threadIdx.x;
  if (onlyUnsettled && onlyUnsettled[img].state == 2 && !onlyUnsettled[img].overflow) return;
  int* go = groupOff + (int64_t)img * nGroups * nBins;
  int run = 0;
  if (b < nBins)
    for (int g = 0; g < nGroups; ++g) {
      const int v = go[g * nBins + b];
      go[g * nBins + b] = run;
      run += v;
    }
  tot[b] = b < nBins ? run : 0;
  __syncthreads();
  if (b == 0) {
    int acc = 0;
    for (int k = nBins - 1; k >= 0; --k) { start[k] = acc; acc += tot[k]; }   // bins in descending order
    nDefined[img] = acc;
  }
  __syncthreads();
  if (b < nBins) {
    const int s0 = start[b];
    for (int g = 0; g < nGroups; ++g) go[g * nBins + b] += s0;
  }
}

// stable scatter: one wave per chunk walks its 1024 pixels in raster order
__global__ __launch_bounds__(64) void k_lsd_scatter(const int* __restrict__ g2a, int npix, int g2Thresh, int nBins,
                                                    const int* __restrict__ maxG2, const int* __restrict__ chunkBase,
                                                    int nChunks, int* __restrict__ order, int img0, int nimg,
                                                    const double* __restrict__ mgAll, const unsigned long long* __restrict__ maxMg,
                                                    double rho, int* __restrict__ rankAll,
                                                    const int* __restrict__ groupOff, int chunksPerGroup, int nGroups,
                                                    const RxCtl* __restrict__ onlyUnsettled) {
  __shared__ int base[1024];
  // XCD-aware order: workgroup L runs on XCD L % 8, so all chunks of an image are dealt to ONE XCD (consecutive slots of
  // that XCD, i.e. close in time): the 4-byte stores of different chunks into the same lines of the ordered list then
  // merge in that XCD's L2 instead of leaving partial lines in eight L2s.  Launched with 8 * ceil(nimg / 8) * nChunks blocks.
  const int L = blockIdx.x, slot = L >> 3;
  const int li = (L & 7) + 8 * (slot / nChunks), chunk = slot % nChunks, lane = threadIdx.x;
  if (li >= nimg) return;
  const int img = li + img0;
  if (onlyUnsettled && onlyUnsettled[img].state == 2 && !onlyUnsettled[img].overflow) return;
  const int* cb = chunkBase + ((int64_t)img * nChunks + chunk) * nBins;
  if (groupOff) {                                        // (large images: k_lsd_scan_part / k_lsd_scan_groups)
    const int* go = groupOff + ((int64_t)img * nGroups + chunk / chunksPerGroup) * nBins;
    for (int i = lane; i < nBins; i += 64) base[i] = cb[i] + go[i];
  } else {
    for (int i = lane; i < nBins; i += 64) base[i] = cb[i];
  }
  __syncthreads();
  const bool f64 = mgAll != nullptr;
  const double bc = f64 ? lsd_bin_coef64(maxMg[img], nBins) : lsd_bin_coef(maxG2[img], nBins);
  const int* g = g2a + (int64_t)img * npix;
  const double* gd = mgAll + (int64_t)img * npix;
  int* ord = order + (int64_t)img * npix;
  // (relaxations) the rank of every pixel — its slot in the ordered list, LSD_ID_INF for an undefined pixel — is written here,
  // in raster order, instead of being scattered from the list afterwards
  int* rk = rankAll ? rankAll + (int64_t)img * npix : nullptr;
  const int nbits = 32 - __clz(max(nBins - 1, 1));
  constexpr int GRP = 16;                        // rows of 64 pixels whose loads are in flight together
  for (int it0 = 0; it0 < LSD_CHUNK / 64; it0 += GRP) {
    if (chunk * LSD_CHUNK + it0 * 64 >= npix) break;
    int vals[GRP];                               // the bin of the pixel, -1 = not defined
    // (unconditional loads from a clamped index, all before the first use: under the lane predicate i < npix each load was waited for
    // at the end of its branch, and the group's loads were NOT in flight together)
    if (f64) {
      double vv[GRP];
#pragma unroll
      for (int u = 0; u < GRP; ++u) vv[u] = gd[min(chunk * LSD_CHUNK + (it0 + u) * 64 + lane, npix - 1)];
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int i = chunk * LSD_CHUNK + (it0 + u) * 64 + lane;
        vals[u] = (i < npix && !(vv[u] <= rho)) ? lsd_bin64(vv[u], bc, nBins) : -1;
      }
    } else {
      int vv[GRP];
#pragma unroll
      for (int u = 0; u < GRP; ++u) vv[u] = g[min(chunk * LSD_CHUNK + (it0 + u) * 64 + lane, npix - 1)];
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int i = chunk * LSD_CHUNK + (it0 + u) * 64 + lane;
        vals[u] = (i < npix && vv[u] > g2Thresh) ? lsd_bin(vv[u], bc) : -1;
      }
    }
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      const int i = chunk * LSD_CHUNK + (it0 + u) * 64 + lane;
      const int bin = vals[u];
      const bool def = bin >= 0;
      // lanes with the same bin (match-any by bits: one ballot per bin bit instead of one round per distinct bin)
      unsigned long long peers = __builtin_amdgcn_ballot_w64(def);
      if (!peers) {
        if (rk && i < npix) rk[i] = LSD_ID_INF;
        continue;
      }
      for (int b = 0; b < nbits; ++b) {
        const bool bit = (bin >> b) & 1;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(def && bit);
        peers &= bit ? bal : ~bal;
      }
      int rank = 0, cnt = 0;
      bool last = false;
      if (def) {
        rank = __popcll(peers & ((1ull << lane) - 1ull));
        cnt = __popcll(peers);
        last = (peers >> lane) == 1ull;
      }
      const int slot = def ? base[bin] + rank : LSD_ID_INF;
      if (def) ord[slot] = i;
      if (rk && i < npix) rk[i] = slot;
      // single-wave block: the LDS operations of a wave execute in order (a block barrier would also wait for the stores)
      lsd_wave_sync();
      if (def && last) base[bin] += cnt;
      lsd_wave_sync();
    }
  }
}

// -DLSD_STATS (make EXTRA=-DLSD_STATS): per-image counts of the sequential grower, read by tools/lsd_stats.py
#ifdef LSD_STATS
__device__ unsigned long long g_lsdStats[24];
__device__ unsigned long long g_lsdStatsMax;
extern "C" unsigned long long pli_lsd_stats_max() { unsigned long long v = 0, z = 0; (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lsdStatsMax), 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lsdStatsMax), &z, 8); return v; }
extern "C" void pli_lsd_stats(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lsdStats), 192); unsigned long long z[24] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lsdStats), z, 192); }
#define LSTAT(i, v) do { if (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) stt[i] += (v); } while (0)
#define LCLOCK() __builtin_readcyclecounter()
#define LWAIT() __builtin_amdgcn_s_waitcnt(0)
#define LTIME(i, t0) do { stt[i] += __builtin_readcyclecounter() - (t0); } while (0)
#else
#define LSTAT(i, v) do {} while (0)
#define LCLOCK() 0ull
#define LWAIT() do {} while (0)
#define LTIME(i, t0) do {} while (0)
#endif
constexpr int LSD_QCAP = 1024;      // region queue entries kept in LDS (8 KB/wave): longer regions spill to global memory

__device__ __forceinline__ double lsd_angle_diff(double a, double b) {
  double diff = a - b;
  while (diff <= -D_PI) diff += D_2PI;
  while (diff > D_PI) diff -= D_2PI;
  return fabs(diff);
}

// broadcast from a wave-uniform lane (v_readlane_b32, no LDS crossbar)
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }


// LDS read that stays a ds_read (see the note in k_lsd_grow)
__device__ __forceinline__ uint2 lsd_lds_read2(const uint2* p) {
  typedef __attribute__((address_space(3))) const volatile unsigned lds_cvu;
  lds_cvu* q = (lds_cvu*)p;
  return make_uint2(q[0], q[1]);
}

// (the LDS read is unconditional so that it stays a ds_read instead of a flat load)
__device__ __forceinline__ uint2 lsd_qget(const uint2* qs, const uint2* qg, int k, int qcap = LSD_QCAP) {
  if (qcap == 0) return qg[k];                          // (k_lsd_rect: the whole list is in global memory)
  uint2 e = lsd_lds_read2(&qs[min(k, qcap - 1)]);
  if (k >= qcap) e = qg[k - qcap];
  return e;
}

// region2rect + the end points of the segment (lsd.cpp region2rect / get_theta, refine = NONE): the weighted sums are
// accumulated in list order by three lanes (bit-exact with the sequential loop), the products 64 at a time.
__device__ __forceinline__ void lsd_region2rect(const uint2* qs, const uint2* qg, double (*st)[64], int cnt, double reg_angle,
                                                double prec, double scale, int lane, float* __restrict__ seg, int nseg, int maxSeg,
                                                int qcap = LSD_QCAP, const double* __restrict__ mg = nullptr, int W = 0) {
  // weight of a region pixel = its gradient norm: sqrt(g2 / 4) from the queue entry, or (CV_64F pipeline) the double plane
  auto weight = [&](const uint2 e) -> double {
    return mg ? mg[(int)(e.x >> 16) * W + (int)(e.x & 0xFFFFu)] : sqrt((double)(int)e.y / 4.0);
  };
  // pass 1: x = sum x*w, y = sum y*w, sum = sum w, in list order
  double acc = 0.0;                                   // lanes 0,1,2 hold x, y, sum
  for (int c0 = 0; c0 < cnt; c0 += 64) {
    const int k = c0 + lane;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;              // lanes past the end add +0.0 (the sums are never -0.0)
    if (k < cnt) {
      const uint2 e = lsd_qget(qs, qg, k, qcap);
      const double w = weight(e);
      v0 = (double)(int)(e.x & 0xFFFFu) * w;
      v1 = (double)(int)(e.x >> 16) * w;
      v2 = w;
    }
    st[0][lane] = v0; st[1][lane] = v1; st[2][lane] = v2;
    lsd_wave_sync();
    if (lane < 3) {
      // list-order sum, eight terms per trip: the LDS reads of a trip are issued together, the adds stay in order
      const int m = (min(64, cnt - c0) + 7) & ~7;
      for (int t = 0; t < m; t += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = st[lane][t + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += a[u];
      }
    }
    lsd_wave_sync();
  }
  // (wave-uniform f64 arithmetic costs a full wave instruction per operation: the two centroid divisions are one
  // division in lanes 0 and 1, the four end-point divisions below one division in lanes 0..3)
  const double cq = acc / __shfl(acc, 2, 64);
  const double x = __shfl(cq, 0, 64), y = __shfl(cq, 1, 64);
  // pass 2: inertia
  acc = 0.0;                                          // lanes 0,1,2 hold Ixx, Iyy, Ixy
  for (int c0 = 0; c0 < cnt; c0 += 64) {
    const int k = c0 + lane;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;              // past the end: acc + 0.0 and acc - 0.0 leave acc as it is
    if (k < cnt) {
      const uint2 e = lsd_qget(qs, qg, k, qcap);
      const double w = weight(e);
      const double dx = (double)(int)(e.x & 0xFFFFu) - x, dy = (double)(int)(e.x >> 16) - y;
      v0 = dy * dy * w;
      v1 = dx * dx * w;
      v2 = dx * dy * w;
    }
    st[0][lane] = v0; st[1][lane] = v1; st[2][lane] = v2;
    lsd_wave_sync();
    if (lane < 3) {
      const int m = (min(64, cnt - c0) + 7) & ~7;
      for (int t = 0; t < m; t += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = st[lane][t + u];
        if (lane < 2) {
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += a[u];
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) acc -= a[u];
        }
      }
    }
    lsd_wave_sync();
  }
  const double Ixx = __shfl(acc, 0, 64), Iyy = __shfl(acc, 1, 64), Ixy = __shfl(acc, 2, 64);
  const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
  const bool wide = fabs(Ixx) > fabs(Iyy);
  double theta = (double)fast_atan2_deg(wide ? (float)(lambda - Ixx) : (float)Ixy, wide ? (float)Ixy : (float)(lambda - Iyy));
  theta *= D_DEG2RAD;
  if (lsd_angle_diff(theta, reg_angle) > prec) theta += D_PI;
  double dxr, dyr;
  sincos(theta, &dyr, &dxr);
  // pass 3: extent along the main axis (min/max are order independent)
  double l_min = 0, l_max = 0;
  for (int k = lane; k < cnt; k += 64) {
    const uint2 e = lsd_qget(qs, qg, k, qcap);
    const double l = ((double)(int)(e.x & 0xFFFFu) - x) * dxr + ((double)(int)(e.x >> 16) - y) * dyr;
    l_max = fmax(l_max, l);
    l_min = fmin(l_min, l);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    l_max = fmax(l_max, __shfl_xor(l_max, o, 64));
    l_min = fmin(l_min, __shfl_xor(l_min, o, 64));
  }
  // lanes 0..3: x1, y1, x2, y2
  double e = ((lane & 1) ? y : x) + ((lane & 2) ? l_max : l_min) * ((lane & 1) ? dyr : dxr);
  e += 0.5;
  if (scale != 1) e /= scale;
  if (nseg < maxSeg && lane < 4) seg[4 * nseg + lane] = (float)e;
}

// ---------------------------------------------------------------------------
// k_lsd_grow: the sequential form of region_grow + region2rect, one wave per
// image (used for large batches, where one wave per SIMD per image keeps the
// chip busy, and as the fallback of the relaxation in lsd_relax.hip).
//   * 64 seeds of the ordered list are fetched per vector load; a ballot keeps
//     the still-unused ones and every later claim clears its bit;
//   * lanes 0..8 fetch the 3x3 neighbourhood records of the current region
//     pixel with one 16-byte load each; ballots over "unused & aligned"
//     reproduce the raster-order accept loop, re-testing only the neighbours
//     after an accepted one against the updated region angle;
//   * the region (x|y, g2) queue lives in LDS; the weighted sums of region2rect
//     are accumulated in list order by three lanes from products computed 64 at a time.
// Claimed pixels are marked by overwriting rec.x with NOTDEF.
// ---------------------------------------------------------------------------
template <int WPB>   // waves (= images) per block
__device__ __forceinline__ void lsd_grow_image(const DevParams* __restrict__ Pp, float4* __restrict__ recAll, const double* __restrict__ mgAll,
                                               const int* __restrict__ orderAll, const int* __restrict__ nDefined,
                                               uint2* __restrict__ regOverflow, float* __restrict__ segAll,
                                               int* __restrict__ nSeg, int maxSeg, int img0, int nimg) {
  __shared__ uint2 qsAll[WPB][LSD_QCAP];
  __shared__ double stAll[WPB][3][64];
  const DevParams& P = *Pp;
  const int wv = threadIdx.x >> 6;
  if ((int)blockIdx.x * WPB + wv >= nimg) return;       // (no block-wide barrier below: the waves are independent)
  const int img = blockIdx.x * WPB + wv + img0;
  const int lane = threadIdx.x & 63;
  uint2* qs = qsAll[wv];
  double (*st)[64] = stAll[wv];
  const int W = P.LW, H = P.LH;
  const int64_t npix = (int64_t)W * H;
  float4* rec = recAll + img * npix;
  const int* order = orderAll + img * npix;
  uint2* qg = regOverflow + img * npix;
  float* seg = segAll + (int64_t)img * maxSeg * 4;
  const int nOrder = nDefined[img];
  const int minReg = P.minRegSize;
  const double prec = P.prec, scale = P.lsdScale;
  const float alignLo = P.alignLo, alignHi = P.alignHi;
  const bool useFilter = P.alignFilter != 0;
  const bool smallImg = npix < (1 << 21);
  const float invW = 1.0f / (float)W;
  const int ndx = lane % 3 - 1, ndy = (lane / 3) % 3 - 1;   // lanes 0..8: raster order of the 3x3 block
  int nseg = 0;
#ifdef LSD_STATS
  unsigned long long stt[24] = {};
  const unsigned long long tKernel = LCLOCK();
#endif
  LSTAT(7, nOrder);

  for (int base = 0; base < nOrder; base += 64) {
    const unsigned long long tSeed = LCLOCK();
    const int idx = base + lane;
    const bool valid = idx < nOrder;
    const int sp_l = valid ? order[idx] : -1;
    float4 srec = make_float4(LSD_NOTDEF, 0.f, 0.f, 0.f);
    if (valid) srec = rec[sp_l];
    // seed sums: float(cos(reg_angle)), float(sin(reg_angle)) with reg_angle = angle as double
    float scos = 0.f, ssin = 0.f;
    if (srec.x != LSD_NOTDEF) {
      const double ra = (double)srec.x * D_DEG2RAD;
      double sn, cn;
      sincos(ra, &sn, &cn);
      scos = (float)cn;
      ssin = (float)sn;
    }
    unsigned long long unusedMask = __builtin_amdgcn_ballot_w64(srec.x != LSD_NOTDEF);
    LTIME(9, tSeed);
    while (unusedMask) {
      const int j = __ffsll((long long)unusedMask) - 1;
      unusedMask &= unusedMask - 1ull;
      const int sp = rl_i(sp_l, j);
      const float sa = rl_f(srec.x, j);
      float sumdx = rl_f(scos, j), sumdy = rl_f(ssin, j);
      const int sg2 = __float_as_int(rl_f(srec.w, j));
      double reg_angle = (double)sa * D_DEG2RAD;     // the seed's own angle until the first pixel is added
      bool angValid = true;                          // reg_angle is current (it is a function of the sums otherwise)
      // (row of a linear index without an integer division: exact while W * H < 2^21, see the bound in DESIGN.md §5)
      const int spy = smallImg ? (int)(((float)sp + 0.5f) * invW) : sp / W, spx = sp - spy * W;
      // single-lane work inside these wave-uniform loops is done by the first ACTIVE lane (or by all lanes with
      // the same value): a fixed lane such as lane 0 is not guaranteed to be in the exec mask here
      if (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) rec[sp].x = LSD_NOTDEF;
      qs[0] = make_uint2(((unsigned)spy << 16) | (unsigned)spx, (unsigned)sg2);
      int cnt = 1;
      LSTAT(0, 1);
      // One BFS step.  Two copies: while the queue fits in LDS the step touches global memory only for the
      // neighbourhood load and the USED marks.  (A queue read that may come from LDS or from the overflow area
      // becomes a flat load, and its wait — vmcnt(0) — also covers the USED stores of the previous step: a full
      // store round trip in front of every neighbourhood load.)
      auto step = [&](int k, auto spillTag) {
        constexpr bool SPILL = decltype(spillTag)::value;
        uint2 e = lsd_lds_read2(&qs[SPILL ? min(k, LSD_QCAP - 1) : k]);
        if (SPILL && k >= LSD_QCAP) e = qg[k - LSD_QCAP];
        e.x = __builtin_amdgcn_readfirstlane(e.x);
        e.y = __builtin_amdgcn_readfirstlane(e.y);
        const int px = (int)(e.x & 0xFFFFu), py = (int)(e.x >> 16);
        const int nx = px + ndx, ny = py + ndy;
        const bool inb = lane < 9 && nx >= 0 && ny >= 0 && nx < W && ny < H;
        const int qi = ny * W + nx;
        float4 r = make_float4(LSD_NOTDEF, 0.f, 0.f, 0.f);
        if (inb) r = rec[qi];
        const bool cand = r.x != LSD_NOTDEF;
        const double ad = (double)r.x * D_DEG2RAD;
        unsigned long long remaining = __builtin_amdgcn_ballot_w64(cand);
        if (!angValid) {
          reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
          angValid = true;
        }
        while (remaining) {
          double n_theta = fabs(reg_angle - ad);
          if (n_theta > D_3_2_PI) {
            n_theta = fabs(n_theta - D_2PI);
          }
          const unsigned long long m = __builtin_amdgcn_ballot_w64(cand && n_theta <= prec) & remaining;
          if (!m) break;
          const int j2 = __ffsll((long long)m) - 1;
          remaining &= ~((2ull << j2) - 1ull);
          const int qj = rl_i(qi, j2);
          const float cj = rl_f(r.y, j2), sj = rl_f(r.z, j2);
          const unsigned g2j = (unsigned)__float_as_int(rl_f(r.w, j2));
          const unsigned xyj = ((unsigned)(py + j2 / 3 - 1) << 16) | (unsigned)(px + j2 % 3 - 1);
          if (lane == j2) rec[qi].x = LSD_NOTDEF;
          if (!SPILL || cnt < LSD_QCAP) qs[cnt] = make_uint2(xyj, g2j);           // same value from every active lane
          else if (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) qg[cnt - LSD_QCAP] = make_uint2(xyj, g2j);
          ++cnt;
          sumdx = __fadd_rn(sumdx, cj);
          sumdy = __fadd_rn(sumdy, sj);
          reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
          unusedMask &= ~__builtin_amdgcn_ballot_w64(sp_l == qj);
        }
        if (SPILL && cnt > LSD_QCAP) __threadfence_block();   // overflow entries are read back through global memory
      };
      // Batched steps: up to 8 queue entries are popped together, their 8 x 8 neighbourhood records come back in one
      // round trip, and ONE accept loop walks the combined candidate list in (entry, raster) order = lane order —
      // exactly the order in which the sequential loop would test them.  A pixel can sit in the list more than
      // once (neighbour of several entries): it is tested again at each of its turns with the angle of that moment,
      // as in the sequential loop, and all its later copies are dropped once it is accepted.  Pixels accepted in
      // the batch are appended to the queue and popped by later batches (FIFO order is unchanged).
      auto batch = [&](int k, int nb) {
        const unsigned long long tB0 = LCLOCK();
        const int pi = lane >> 3, ni = (lane & 7) < 4 ? (lane & 7) : (lane & 7) + 1;   // 8 neighbours, raster order, centre skipped
        const bool act = pi < nb;
        const unsigned ex = lsd_lds_read2(&qs[k + (act ? pi : 0)]).x;
        const int nx = (int)(ex & 0xFFFFu) + ni % 3 - 1, ny = (int)(ex >> 16) + ni / 3 - 1;
        const bool inb = act && nx >= 0 && ny >= 0 && nx < W && ny < H;
        const int qi = inb ? ny * W + nx : -1;
        float4 r = make_float4(LSD_NOTDEF, 0.f, 0.f, 0.f);
        if (inb) r = rec[qi];
        const bool cand = r.x != LSD_NOTDEF;
        const unsigned myxy = ((unsigned)ny << 16) | (unsigned)nx;
        unsigned long long remaining = __builtin_amdgcn_ballot_w64(cand);
        LTIME(10, tB0);
        const unsigned long long tB1 = LCLOCK();
        // The alignment test in vector form (see DevParams::alignLo): "surely aligned" and "surely not" need no
        // region angle, so the arctangent after every accepted pixel is only evaluated for the rare pixel whose
        // angle falls inside the margin — that one is decided by the reference's expression.
        while (remaining) {
          // dot > 0 and dot^2 >= T  <=>  dot * |dot| >= T (T > 0).  This filter only has to be conservative (the margin is
          // three orders above float rounding), so it may use fused multiply-adds.
          const float n2 = __builtin_fmaf(sumdx, sumdx, sumdy * sumdy);
          const float dot = __builtin_fmaf(sumdx, r.y, sumdy * r.z);
          const float sd2 = dot * __builtin_fabsf(dot);
          // (plain lane masks combined with scalar ops: no per-lane short-circuit branches; `remaining` only holds candidates)
          unsigned long long mm = __builtin_amdgcn_ballot_w64(sd2 >= alignLo * n2);
          if (!useFilter) mm = ~0ull;
          const unsigned long long m = mm & remaining;
          if (!m) break;
          const int j2 = __ffsll((long long)m) - 1;
          remaining &= ~((2ull << j2) - 1ull);
          unsigned long long sure = __builtin_amdgcn_ballot_w64(sd2 >= alignHi * n2);
          if (!useFilter) sure = 0ull;
          if (!((sure >> j2) & 1ull)) {
            if (!angValid) {
              reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
              angValid = true;
            }
            double n_theta = fabs(reg_angle - (double)rl_f(r.x, j2) * D_DEG2RAD);
            if (n_theta > D_3_2_PI) {
              n_theta = fabs(n_theta - D_2PI);
            }
            if (!(n_theta <= prec)) continue;
          }
          const int qj = rl_i(qi, j2);
          const float cj = rl_f(r.y, j2), sj = rl_f(r.z, j2);
          if (lane == j2) {                            // lane j2 is in the exec mask: it is a set bit of a ballot
            rec[qi].x = LSD_NOTDEF;
            qs[cnt] = make_uint2(myxy, (unsigned)__float_as_int(r.w));
          }
          remaining &= ~__builtin_amdgcn_ballot_w64(qi == qj);            // the other copies of the accepted pixel
          ++cnt;
          sumdx = __fadd_rn(sumdx, cj);
          sumdy = __fadd_rn(sumdy, sj);
          angValid = false;
          unusedMask &= ~__builtin_amdgcn_ballot_w64(sp_l == qj);
        }
        LTIME(11, tB1);
      };
      for (int k = 0; k < cnt;) {
        if (cnt + 65 <= LSD_QCAP) {                    // a batch can append up to 8 x 8 entries
          const int nb = min(8, cnt - k);
          LSTAT(1, 1);
          batch(k, nb);
          k += nb;
        } else {
          LSTAT(13, 1);
          if (cnt + 9 > LSD_QCAP) step(k, std::true_type{});
          else step(k, std::false_type{});
          ++k;
        }
      }
      LSTAT(3, cnt);
      if (cnt + 65 > LSD_QCAP) { LSTAT(14, cnt); LSTAT(15, 1); }
      if (cnt == 1) LSTAT(5, 1);
      if (cnt <= 4) LSTAT(6, 1);
      if (cnt < minReg) continue;
      if (!angValid) reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
      LSTAT(4, 1);
      LSTAT(2, cnt);
      const unsigned long long tRect = LCLOCK();
      lsd_region2rect(qs, qg, st, cnt, reg_angle, prec, scale, lane, seg, nseg, maxSeg, LSD_QCAP, mgAll ? mgAll + img * npix : nullptr, W);
      ++nseg;
      LTIME(12, tRect);
    }
  }
  if (lane == 0) nSeg[img] = nseg < maxSeg ? nseg : maxSeg;
#ifdef LSD_STATS
  LTIME(8, tKernel);
  if (lane == 0) for (int i = 0; i < 24; ++i) atomicAdd(&g_lsdStats[i], stt[i]);
  if (lane == 0) atomicMax(&g_lsdStatsMax, stt[8]);
#endif
}


// ---------------------------------------------------------------------------
// The accept loop of a batched step, hand-scheduled (the compiler's rendering of the same loop carries ~48 instructions and
// seven branches per accepted pixel; this one 15 VALU + 13 SALU and three not-taken branches).  State: the float sums
// (replicated in every lane), `remaining` (candidates not yet decided, lane order = test order), `acc` (lanes accepted so far
// in this batch: they are accepted in increasing lane order, so a lane's queue slot is cnt0 + its rank in acc), cnt.
// Per iteration: vector filter of every candidate against the current sums (lo: "maybe aligned", hi: "surely aligned", see
// lsd_grow_image), first maybe-lane j; if it is not sure the loop hands over to the exact test (return 1, `remaining` already
// without the lanes up to j); else lane j is accepted: sums += (cos, sin)(j), the other copies of its pixel leave
// `remaining`; if the pixel carries a speculation tag the caller must see it (return 2, pixel index in qj).
// Return 0: no candidate left.  Wait states (gfx940 rules): a v_readlane result is used by a VALU instruction at least two
// instructions later; VALU-written VCC is only read by SALU instructions (interlocked).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lsd_accept_fast(float& sumdx, float& sumdy, float cosv, float sinv, int qi, unsigned long long& remaining,
                                               unsigned long long& acc, int& cnt, unsigned long long tagMask, float lo, float hi,
                                               int& j2, int& qj) {
  int code, sc, ss;
  float t0, t1, t2, t3;
  unsigned long long m, sh;
  // (wave-uniform values the register allocator may be holding in vector registers)
  cnt = __builtin_amdgcn_readfirstlane(cnt);
  lo = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lo)));
  hi = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hi)));
  asm volatile(
      "1:\n\t"
      "v_mul_f32_e32 %[t0], %[sy], %[sy]\n\t"
      "v_mul_f32_e32 %[t1], %[sy], %[sn]\n\t"
      "v_fmac_f32_e32 %[t0], %[sx], %[sx]\n\t"
      "v_fmac_f32_e32 %[t1], %[sx], %[cs]\n\t"
      "v_mul_f32_e32 %[t2], %[lo], %[t0]\n\t"
      "v_mul_f32_e64 %[t3], %[t1], |%[t1]|\n\t"
      "v_mul_f32_e32 %[t0], %[hi], %[t0]\n\t"
      "v_cmp_ge_f32_e32 vcc, %[t3], %[t2]\n\t"
      "s_and_b64 %[m], vcc, %[rem]\n\t"
      "s_cbranch_scc0 4f\n\t"
      "v_cmp_ge_f32_e32 vcc, %[t3], %[t0]\n\t"
      "s_ff1_i32_b64 %[j], %[m]\n\t"
      "s_lshl_b64 %[sh], -2, %[j]\n\t"
      "s_and_b64 %[rem], %[rem], %[sh]\n\t"
      "s_bitcmp1_b64 vcc, %[j]\n\t"
      "s_cbranch_scc0 5f\n\t"
      "v_readlane_b32 %[sc], %[cs], %[j]\n\t"
      "v_readlane_b32 %[ss], %[sn], %[j]\n\t"
      "v_readlane_b32 %[q], %[qi], %[j]\n\t"
      "s_bitset1_b64 %[acc], %[j]\n\t"
      "s_add_i32 %[cnt], %[cnt], 1\n\t"
      "v_add_f32_e32 %[sx], %[sc], %[sx]\n\t"
      "v_add_f32_e32 %[sy], %[ss], %[sy]\n\t"
      "v_cmp_eq_u32_e32 vcc, %[q], %[qi]\n\t"
      "s_bitcmp1_b64 %[tag], %[j]\n\t"
      "s_cbranch_scc1 6f\n\t"
      "s_andn2_b64 %[rem], %[rem], vcc\n\t"
      "s_cbranch_scc1 1b\n"
      "4:\n\t"
      "s_mov_b32 %[code], 0\n\t"
      "s_branch 9f\n"
      "5:\n\t"
      "s_mov_b32 %[code], 1\n\t"
      "s_branch 9f\n"
      "6:\n\t"
      "s_andn2_b64 %[rem], %[rem], vcc\n\t"
      "s_mov_b32 %[code], 2\n"
      "9:\n\t"
      : [sx] "+v"(sumdx), [sy] "+v"(sumdy), [rem] "+s"(remaining), [acc] "+s"(acc), [cnt] "+s"(cnt), [code] "=&s"(code), [j] "=&s"(j2),
        [q] "=&s"(qj), [sc] "=&s"(sc), [ss] "=&s"(ss), [m] "=&s"(m), [sh] "=&s"(sh), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
        [t3] "=&v"(t3)
      : [cs] "v"(cosv), [sn] "v"(sinv), [qi] "v"(qi), [tag] "s"(tagMask), [lo] "s"(lo), [hi] "s"(hi)
      : "vcc", "scc");
  return code;
}

// ---------------------------------------------------------------------------
// Speculative form of the sequential grower (lsd_grow_image_spec).  The sequential loop spends most of its round trips
// on regions of a few pixels (on the EuRoC-shaped stream 34 000 of the 46 000 regions of an image stay below 9 pixels)
// that a whole wave grows one after the other.  Here the wave first collects 64 LIVE seeds of the ordered list
// (a "super-row"), and every lane grows the region of its own seed ALONE, against the committed state only (USED marks
// of everything before the super-row), up to SPEC_CAP pixels; the pixels a lane takes are tagged with its lane number
// (spare bits of rec.w, plain stores: where two lanes take the same pixel one tag survives).  Then, in lane = seed order:
//   * a lane that still owns the tag of every pixel it took is CLEAN: no lower lane's speculative region touches its
//     pixels.  Its region is final as soon as every lower lane is final, and is committed (USED marks) without any
//     wave-wide work;
//   * a lane that hit the cap, or lost a tag, is regrown by the whole wave with the batched steps of the sequential
//     grower, in its turn (all lower lanes are final then).  Whenever that regrowth accepts a pixel that carries the tag
//     of a higher clean lane, that lane is no longer clean and will be regrown in its turn as well.
// A clean lane's speculative run equals its sequential run: the pixels it tested and rejected are rejected again
// whether or not somebody else has taken them meanwhile, and the pixels it accepted are still unused when its turn
// comes (nobody lower holds them: it owns their tags, and a regrowth that takes one of them un-cleans the lane).
// tools/sim/sim_tile_relax.cpp replays this protocol on the CPU ("lane speculation": wrong 0 for every cap).
// Tags never outlive a super-row: clean lanes' pixels are USED, every other lane clears the tags of its speculative
// pixels before the next super-row.
// ---------------------------------------------------------------------------
constexpr int SPEC_CAP = 8;                 // pixels a lane may take before its region is handed to the whole wave
constexpr int SPEC_Q = LSD_QCAP - SPEC_CAP * 64;   // wave-wide queue entries left in LDS beside the lanes' lists (2 x SPEC_CAP x 64 ints)
constexpr unsigned SPEC_G2MASK = 0x7FFFFu;  // g2 <= 2 * 510^2 < 2^19: rec.w bits 19.. hold the tag (lane + 1)

template <int WPB>
__device__ __forceinline__ void lsd_grow_image_spec(const DevParams* __restrict__ Pp, float4* __restrict__ recAll, const double* __restrict__ mgAll,
                                                    const int* __restrict__ orderAll, const int* __restrict__ nDefined,
                                                    uint2* __restrict__ regOverflow, float* __restrict__ segAll,
                                                    int* __restrict__ nSeg, int maxSeg, int img0, int nimg,
                                                    LsdRectItem* __restrict__ rectAll, double* __restrict__ rectWAll) {
  __shared__ uint2 qsAll[WPB][LSD_QCAP];
  __shared__ double stAll[WPB][3][64];
  const DevParams& P = *Pp;
  const int wv = threadIdx.x >> 6;
  if ((int)blockIdx.x * WPB + wv >= nimg) return;       // (no block-wide barrier below: the waves are independent)
  const int img = blockIdx.x * WPB + wv + img0;
  const int lane = threadIdx.x & 63;
  uint2* qs = qsAll[wv];
  double (*st)[64] = stAll[wv];
  // the lanes' pixel lists [SPEC_CAP][64] (y << 16 | x) and the g2 of those pixels live in the upper half of the queue area
  // until the super-row is resolved: the wave-wide queue keeps SPEC_Q entries in LDS, longer regions spill to global memory
  int* sq = reinterpret_cast<int*>(qs + SPEC_Q);
  int* sq2 = sq + SPEC_CAP * 64;
  int* stageSp = reinterpret_cast<int*>(&st[0][0]);     // staged seeds of the super-row: pixel, angle, g2 (64 entries each)
  float* stageA = reinterpret_cast<float*>(stageSp + 64);
  int* stageG = stageSp + 128;
  const int W = P.LW, H = P.LH;
  const int64_t npix = (int64_t)W * H;
  float4* rec = recAll + img * npix;
  const int* order = orderAll + img * npix;
  uint2* qg = regOverflow + img * npix;
  float* seg = segAll + (int64_t)img * maxSeg * 4;
  const int nOrder = nDefined[img];
  const int minReg = P.minRegSize;
  const double prec = P.prec, scale = P.lsdScale;
  const float alignLo = P.alignLo, alignHi = P.alignHi;
  const unsigned tag = (unsigned)(lane + 1);
  int cap = min(SPEC_CAP, minReg - 1);                  // a region that may yield a segment is never finished by a lane
  if (P.alignPad >> 4) cap = min(cap, P.alignPad >> 4);  // (dev switch PLI_LSD_SPEC_CAP)
  int nseg = 0;
  int pos = 0;
  int avgLive = 8;                                      // running estimate of live seeds per row of 64 list entries
  // region2rect off the serial wave (rectAll != null): the pixel list of every region that yields a segment is left in the
  // image's arena (= the overflow area: the lists of disjoint regions never exceed npix entries), entry apos.., and k_lsd_rect
  // computes the segments afterwards, many regions at a time.  The overflow of the region being grown is written where its list
  // will stay.
  LsdRectItem* items = rectAll ? rectAll + (int64_t)img * maxSeg : nullptr;
  // CV_64F pipeline: the weight of a pixel (its gradient norm) is a double in its own plane; the wave gathers the weights of
  // a finished region (consecutive list entries are neighbours: few sectors per load) and leaves them beside the list, so that
  // k_lsd_rect streams them instead of gathering one sector per pixel and lane
  double* rectW = (rectAll && rectWAll && mgAll) ? rectWAll + img * npix : nullptr;
  const double* mgImg = mgAll ? mgAll + img * npix : nullptr;
  int apos = 0;
#ifdef LSD_STATS
  unsigned long long stt[24] = {};
  const unsigned long long tKernel = LCLOCK();
#endif

  while (pos < nOrder) {
    // ---- fill: up to 64 live seeds, in list order ------------------------------------------------------------
    const unsigned long long tFill = LCLOCK();
    int nst = 0;
    bool full = false;
    while (!full && nst < 64 && pos < nOrder) {
      const int want = (64 - nst + avgLive - 1) / max(avgLive, 1);
      const int R = min(8, max(1, want));
      int sp[8];
      float4 sr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = pos + 64 * u + lane;
        sp[u] = (u < R && idx < nOrder) ? order[idx] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        sr[u] = make_float4(LSD_NOTDEF, 0.f, 0.f, 0.f);
        if (sp[u] >= 0) sr[u] = rec[sp[u]];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u >= R || full || pos >= nOrder) continue;
        const bool live = sr[u].x != LSD_NOTDEF;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(live);
        const int c = __popcll(bal);
        if (nst + c > 64) { full = true; continue; }      // this row does not fit: it starts the next super-row
        if (live) {
          const int at = nst + __popcll(bal & ((1ull << lane) - 1ull));
          stageSp[at] = sp[u];
          stageA[at] = sr[u].x;
          stageG[at] = (int)((unsigned)__float_as_int(sr[u].w) & SPEC_G2MASK);
        }
        nst += c;
        pos += 64;
        avgLive = max(1, (3 * avgLive + c + 2) >> 2);
      }
    }
    LTIME(9, tFill);
    if (nst == 0) continue;
    LSTAT(0, 1);
    const unsigned long long tSpec = LCLOCK();
    lsd_wave_sync();
    // ---- speculation: lane l < nst grows the region of staged seed l alone -------------------------------------
    const bool has = lane < nst;
    const int sp_l = has ? stageSp[lane] : -1;
    const float sa_l = has ? stageA[lane] : 0.f;
    const int sg_l = has ? stageG[lane] : 0;
    lsd_wave_sync();                                     // the staging area is the region2rect scratch: read before anything writes it
    float scos = 0.f, ssin = 0.f;
    double ang_l = 0.0;
    int spx = 0, spy = 0;
    if (has) {
      ang_l = (double)sa_l * D_DEG2RAD;
      double sn, cn;
      sincos(ang_l, &sn, &cn);
      scos = (float)cn;
      ssin = (float)sn;
      spy = sp_l / W;
      spx = sp_l - spy * W;
    }
    int scnt = has ? 1 : 0;       // pixels of the lane's region (for a lane that hit the cap: up to the last completed step)
    int tcnt = scnt;              // pixels that carry the lane's tag (scnt + those of an abandoned step)
    int hk = 0;                   // queue entry the whole wave continues with when it takes the region over
    float hsx = scos, hsy = ssin; // float sums at that point
    bool big = false;
    {
      float sumdx = scos, sumdy = ssin;
      double reg_angle = ang_l;
      int k = 0;
      bool active = has;
      if (has) {
        sq[lane] = (spy << 16) | spx;
        sq2[lane] = sg_l;
        rec[sp_l].w = __int_as_float((int)((tag << 19) | (unsigned)sg_l));     // (through the float member: the float4 loads below must alias it)
      }
      while (__builtin_amdgcn_ballot_w64(active)) {
        LSTAT(14, 1);
        if (active) {
          const int xy = sq[k * 64 + lane];
          const int px = xy & 0xFFFF, py = xy >> 16;
          const int cnt0 = scnt;                         // state at the start of the step: a step that hits the cap is abandoned
          const float sdx0 = sumdx, sdy0 = sumdy;
          float4 nr[8];
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            const int m = n < 4 ? n : n + 1;             // skip the centre of the 3x3 block
            const int nx = px + m % 3 - 1, ny = py + m / 3 - 1;
            nr[n] = make_float4(LSD_NOTDEF, 0.f, 0.f, 0.f);
            if (nx >= 0 && ny >= 0 && nx < W && ny < H) nr[n] = rec[ny * W + nx];
          }
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            if (big || nr[n].x == LSD_NOTDEF) continue;
            const unsigned wbits = (unsigned)__float_as_int(nr[n].w);
            if ((wbits >> 19) == tag) continue;          // already in my region
            double n_theta = fabs(reg_angle - (double)nr[n].x * D_DEG2RAD);
            if (n_theta > D_3_2_PI) {
              n_theta = fabs(n_theta - D_2PI);
            }
            if (!(n_theta <= prec)) continue;
            const int m = n < 4 ? n : n + 1;
            const int nx = px + m % 3 - 1, ny = py + m / 3 - 1;
            // "already mine" must not depend on the tag alone: another lane may have overwritten it since (that lane and this
            // one cannot both stay clean, but a pixel taken twice would corrupt this lane's own run)
            bool mine = false;
            if (wbits >> 19)                             // (a pixel of this lane always carries SOME tag)
              for (int i = 0; i < scnt; ++i) mine = mine || sq[i * 64 + lane] == ((ny << 16) | nx);
            if (mine) continue;
            if (scnt == cap) { big = true; continue; }
            sq[scnt * 64 + lane] = (ny << 16) | nx;
            sq2[scnt * 64 + lane] = (int)(wbits & SPEC_G2MASK);
            ++scnt;
            rec[ny * W + nx].w = __int_as_float((int)((tag << 19) | (wbits & SPEC_G2MASK)));
            sumdx = __fadd_rn(sumdx, nr[n].y);
            sumdy = __fadd_rn(sumdy, nr[n].z);
            reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
          }
          tcnt = scnt;
          if (big) { hk = k; scnt = cnt0; hsx = sdx0; hsy = sdy0; active = false; }
          else {
            ++k;
            hsx = sumdx; hsy = sumdy;
            if (k >= scnt) active = false;
          }
        }
      }
    }
    auto lin = [&](int xy) -> int { return (xy >> 16) * W + (xy & 0xFFFF); };
    lsd_wave_sync();
    LTIME(10, tSpec);
    const unsigned long long tVal = LCLOCK();
    // ---- validation: who still owns the tags of its pixels ------------------------------------------------------
    bool own = has, seedLost = false;                  // own: every pixel of the (prefix of the) region still carries this lane's tag
    {
      unsigned tg[SPEC_CAP];
#pragma unroll
      for (int i = 0; i < SPEC_CAP; ++i) {
        tg[i] = tag << 19;
        if (i < scnt) tg[i] = (unsigned)__float_as_int(const_cast<const volatile float4*>(rec)[lin(sq[i * 64 + lane])].w);
      }
#pragma unroll
      for (int i = 0; i < SPEC_CAP; ++i)
        if (i < scnt && (tg[i] >> 19) != tag) { own = false; if (i == 0) seedLost = true; }
    }
    const bool clean = own && !big;
    unsigned long long cleanMask = __builtin_amdgcn_ballot_w64(clean);
    unsigned long long handMask = __builtin_amdgcn_ballot_w64(own && big);   // capped regions whose prefix the wave can take over
    if (P.alignPad & 1) handMask = 0ull;                 // (dev switch PLI_LSD_SPEC=3: every capped region restarts from its seed)
    unsigned long long todoMask = __builtin_amdgcn_ballot_w64(has && !clean);
    if ((P.alignPad & 2) && todoMask) {                  // (dev switch PLI_LSD_SPEC=4: no lane above the first todo lane stays clean)
      const unsigned long long above = ~((todoMask & (0ull - todoMask)) - 1ull);
      todoMask |= cleanMask & above;
      cleanMask &= ~above;
      handMask = 0ull;
    }
    const unsigned long long everTodo0 = todoMask;
    unsigned long long committed = 0ull, killed = 0ull, everTodo = everTodo0;
    const unsigned long long seedLostMask = __builtin_amdgcn_ballot_w64(seedLost);
    auto commit = [&](unsigned long long m) {
      if ((m >> lane) & 1ull) {
#pragma unroll
        for (int i = 0; i < SPEC_CAP; ++i)
          if (i < scnt) rec[lin(sq[i * 64 + lane])].x = LSD_NOTDEF;
      }
    };
    LTIME(11, tVal);
    LSTAT(3, __popcll(cleanMask)); LSTAT(5, __popcll(__builtin_amdgcn_ballot_w64(big))); LSTAT(4, __popcll(todoMask) - __popcll(__builtin_amdgcn_ballot_w64(big)));
    const unsigned long long tRes = LCLOCK();
    // ---- ordered resolution -----------------------------------------------------------------------------------------
    while (todoMask) {
      const int i = __ffsll((long long)todoMask) - 1;
      todoMask &= todoMask - 1ull;
      const unsigned long long below = cleanMask & ((1ull << i) - 1ull) & ~committed;
      if (below) { commit(below); committed |= below; }
      if ((killed >> i) & 1ull) { LSTAT(6, 1); continue; }
      const int sp = rl_i(sp_l, i);
      const bool hand = (handMask >> i) & 1ull;         // the lane's run up to its last completed step is the sequential run
      // an accepted pixel that carries the tag of a higher lane: that lane is not clean any more; if it is its seed, the lane is dead
      auto tagged = [&](unsigned tg, int qj) {
        const int L = (int)(tg >> 19) - 1;
        if (L > i) {
          const unsigned long long b = 1ull << L;
          if (cleanMask & b & ~committed) { cleanMask &= ~b; todoMask |= b; everTodo |= b; }
          handMask &= ~b;                               // (a capped lane restarts from its seed)
          if (rl_i(sp_l, L) == qj) killed |= b;
        }
      };
      if (!hand && ((seedLostMask >> i) & 1ull)) {
        // the tag on this lane's seed belongs to somebody else: look whether the seed is still unused — and if it is, taking it
        // takes it away from the lane whose tag it carries
        float sx = LSD_NOTDEF, sw = 0.f;
        const int fl = __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1;
        if (lane == fl) { sx = const_cast<const volatile float4*>(rec)[sp].x; sw = const_cast<const volatile float4*>(rec)[sp].w; }
        sx = rl_f(sx, fl);
        if (sx == LSD_NOTDEF) continue;
        const unsigned tw = (unsigned)__float_as_int(rl_f(sw, fl));
        if (tw >> 19) tagged(tw, sp);
      }
      const float sa = rl_f(sa_l, i);
      float sumdx = hand ? rl_f(hsx, i) : rl_f(scos, i), sumdy = hand ? rl_f(hsy, i) : rl_f(ssin, i);
      const int sg2 = rl_i(sg_l, i);
      double reg_angle = (double)sa * D_DEG2RAD;
      const int ry = rl_i(spy, i), rx = rl_i(spx, i);
      int cnt = 1, k0 = 0;
      if (hand) {
        // take the region over where the lane left it: its pixels become USED, its queue becomes the wave's queue
        cnt = rl_i(scnt, i);
        k0 = rl_i(hk, i);
        if (lane == i) {
#pragma unroll
          for (int t = 0; t < SPEC_CAP; ++t)
            if (t < scnt) { const int e = sq[t * 64 + lane]; rec[lin(e)].x = LSD_NOTDEF; qs[t] = make_uint2((unsigned)e, (unsigned)sq2[t * 64 + lane]); }
        }
        lsd_wave_sync();
      } else {
        if (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) rec[sp].x = LSD_NOTDEF;
        qs[0] = make_uint2(((unsigned)ry << 16) | (unsigned)rx, (unsigned)sg2);
      }
      int angCnt = cnt == 1 ? 1 : -1;                    // reg_angle is the angle of the sums at this pixel count (the seed angle at 1)
      LSTAT(1, 1); if (hand) LSTAT(2, 1);
      const int cntStart = cnt;
      const int qoff = items ? apos : -SPEC_Q;           // queue entry k >= SPEC_Q lives at qg[qoff + k]
      auto step = [&](int k, auto spillTag) {
        constexpr bool SPILL = decltype(spillTag)::value;
        const int ndx = lane % 3 - 1, ndy = (lane / 3) % 3 - 1;   // lanes 0..8: raster order of the 3x3 block
        uint2 e = lsd_lds_read2(&qs[SPILL ? min(k, SPEC_Q - 1) : k]);
        if (SPILL && k >= SPEC_Q) e = qg[qoff + k];
        e.x = __builtin_amdgcn_readfirstlane(e.x);
        e.y = __builtin_amdgcn_readfirstlane(e.y);
        const int px = (int)(e.x & 0xFFFFu), py = (int)(e.x >> 16);
        const int nx = px + ndx, ny = py + ndy;
        const bool inb = lane < 9 && nx >= 0 && ny >= 0 && nx < W && ny < H;
        const int qi = ny * W + nx;
        float4 r = make_float4(LSD_NOTDEF, 0.f, 0.f, 0.f);
        if (inb) r = rec[qi];
        const bool cand = r.x != LSD_NOTDEF;
        const double ad = (double)r.x * D_DEG2RAD;
        unsigned long long remaining = __builtin_amdgcn_ballot_w64(cand);
        if (angCnt != cnt) reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
        while (remaining) {
          double n_theta = fabs(reg_angle - ad);
          if (n_theta > D_3_2_PI) {
            n_theta = fabs(n_theta - D_2PI);
          }
          const unsigned long long m = __builtin_amdgcn_ballot_w64(cand && n_theta <= prec) & remaining;
          if (!m) break;
          const int j2 = __ffsll((long long)m) - 1;
          remaining &= ~((2ull << j2) - 1ull);
          const int qj = rl_i(qi, j2);
          const float cj = rl_f(r.y, j2), sj = rl_f(r.z, j2);
          const unsigned wj = (unsigned)__float_as_int(rl_f(r.w, j2));
          const unsigned xyj = ((unsigned)(py + j2 / 3 - 1) << 16) | (unsigned)(px + j2 % 3 - 1);
          if (lane == j2) rec[qi].x = LSD_NOTDEF;
          if (!SPILL || cnt < SPEC_Q) qs[cnt] = make_uint2(xyj, wj & SPEC_G2MASK);           // same value from every active lane
          else if (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) qg[qoff + cnt] = make_uint2(xyj, wj & SPEC_G2MASK);
          ++cnt;
          sumdx = __fadd_rn(sumdx, cj);
          sumdy = __fadd_rn(sumdy, sj);
          reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
          if (wj >> 19) tagged(wj, qj);
        }
        angCnt = cnt;                                       // (this form keeps the angle current)
        if (SPILL && cnt > SPEC_Q) __threadfence_block();   // overflow entries are read back through global memory
      };
      // One round trip for up to 8 queue entries x 8 neighbours; the accept loop (lsd_accept_fast) walks the candidates in lane
      // order = the order of the sequential tests.  The loop is what the wave spends its issue slots on (300 000 iterations per
      // image), so it carries nothing that can wait: accepted lanes are only noted in a mask and write their USED mark and
      // their queue entry after the loop; the tag of an accepted pixel is looked at only where a tag is known to sit (ballot
      // before the loop); without the vector filter alignLo / alignHi are -inf / +inf (never sure, always maybe).
      auto batch = [&](int k, int nb) {
        const int pi = lane >> 3, ni = (lane & 7) < 4 ? (lane & 7) : (lane & 7) + 1;   // 8 neighbours, raster order, centre skipped
        const bool act = pi < nb;
        const unsigned ex = lsd_lds_read2(&qs[k + (act ? pi : 0)]).x;
        const int nx = (int)(ex & 0xFFFFu) + ni % 3 - 1, ny = (int)(ex >> 16) + ni / 3 - 1;
        const bool inb = act && nx >= 0 && ny >= 0 && nx < W && ny < H;
        const int qi = inb ? ny * W + nx : -1;
        float4 r = make_float4(LSD_NOTDEF, 0.f, 0.f, 0.f);
        const unsigned long long tF = LCLOCK();
        if (inb) r = rec[qi];
        LWAIT();
        LTIME(16, tF);
        const unsigned long long tA = LCLOCK();
        const bool cand = r.x != LSD_NOTDEF;
        const unsigned myxy = ((unsigned)ny << 16) | (unsigned)nx;
        const unsigned wbits = (unsigned)__float_as_int(r.w);
        unsigned long long remaining = __builtin_amdgcn_ballot_w64(cand);
        const unsigned long long tagMask = __builtin_amdgcn_ballot_w64(cand && (wbits >> 19) != 0u);
        unsigned long long acc = 0ull;                 // lanes accepted in this batch (in increasing lane order)
        const int cnt0 = cnt;
        LSTAT(18, nb);
        while (remaining) {
          int j2, qj;
          const int code = lsd_accept_fast(sumdx, sumdy, r.y, r.z, qi, remaining, acc, cnt, tagMask, alignLo, alignHi, j2, qj);
          if (code == 0) break;
          if (code == 1) {
            // lane j2 lies inside the margin of the vector filter: the reference's own expression decides
            if (angCnt != cnt) {
              reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
              angCnt = cnt;
            }
            double n_theta = fabs(reg_angle - (double)rl_f(r.x, j2) * D_DEG2RAD);
            if (n_theta > D_3_2_PI) {
              n_theta = fabs(n_theta - D_2PI);
            }
            // (the sums live in vector registers: the compiler takes everything derived from them for lane-dependent; say that
            // this decision is the same in every lane, or the loop state ends up in vector registers)
            if (!__builtin_amdgcn_readfirstlane((int)(n_theta <= prec))) continue;
            qj = rl_i(qi, j2);
            const float cj = rl_f(r.y, j2), sj = rl_f(r.z, j2);
            acc |= 1ull << j2;
            remaining &= ~__builtin_amdgcn_ballot_w64(qi == qj);          // the other copies of the accepted pixel
            ++cnt;
            sumdx = __fadd_rn(sumdx, cj);
            sumdy = __fadd_rn(sumdy, sj);
            if (!((tagMask >> j2) & 1ull)) continue;
          }
          tagged((unsigned)rl_i((int)wbits, j2), qj);
        }
        LSTAT(19, cnt - cnt0);
        if ((acc >> lane) & 1ull) {
          rec[qi].x = LSD_NOTDEF;
          qs[cnt0 + __popcll(acc & ((1ull << lane) - 1ull))] = make_uint2(myxy, wbits & SPEC_G2MASK);
        }
        LTIME(17, tA);
      };
      // (two loops, not one with a branch: as alternatives inside one loop the two forms share their loop-carried state and
      // the compiler moves a dozen registers per step between them)
      int k = k0;
      while (k < cnt && cnt + 65 <= SPEC_Q) {        // a batch can append up to 8 x 8 entries
        const int nb = min(8, cnt - k);
        LSTAT(15, 1);
        batch(k, nb);
        k += nb;
      }
      while (k < cnt) {
        if (cnt + 9 > SPEC_Q) step(k, std::true_type{});
        else step(k, std::false_type{});
        ++k;
      }
      LSTAT(7, cnt - cntStart);
      if (cnt < minReg) continue;
      if (angCnt != cnt) reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * D_DEG2RAD;
      const unsigned long long tRect = LCLOCK();
      if (items) {
        const int nl = min(cnt, SPEC_Q);
        for (int i2 = lane; i2 < nl; i2 += 64) qg[apos + i2] = lsd_lds_read2(&qs[i2]);
        if (rectW) {
          if (cnt > SPEC_Q) __threadfence_block();
          for (int i2 = lane; i2 < cnt; i2 += 64) {
            const unsigned exy = i2 < SPEC_Q ? lsd_lds_read2(&qs[min(i2, SPEC_Q - 1)]).x : qg[apos + i2].x;
            rectW[apos + i2] = mgImg[(int)(exy >> 16) * W + (int)(exy & 0xFFFFu)];
          }
        }
        if (nseg < maxSeg && lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) {
          LsdRectItem it;
          it.off = apos; it.cnt = cnt; it.reg_angle = reg_angle;
          items[nseg] = it;
        }
        apos += cnt;
      } else {
        lsd_region2rect(qs, qg, st, cnt, reg_angle, prec, scale, lane, seg, nseg, maxSeg, SPEC_Q, mgAll ? mgAll + img * npix : nullptr, W);
      }
      LTIME(13, tRect);
      ++nseg;
    }
    LTIME(12, tRes);
    // ---- the lanes that are still clean are final: commit them; everybody else clears its tags -------------------
    {
      const unsigned long long rest = cleanMask & ~committed;
      if (rest) commit(rest);
      if ((everTodo >> lane) & 1ull) {
#pragma unroll
        for (int i = 0; i < SPEC_CAP; ++i)
          if (i < tcnt) rec[lin(sq[i * 64 + lane])].w = __int_as_float(sq2[i * 64 + lane]);
      }
    }
  }
  if (lane == 0) nSeg[img] = nseg < maxSeg ? nseg : maxSeg;
#ifdef LSD_STATS
  LTIME(8, tKernel);
  if (lane == 0) for (int i = 0; i < 24; ++i) atomicAdd(&g_lsdStats[i], stt[i]);
  if (lane == 0) atomicMax(&g_lsdStatsMax, stt[8]);
#endif
}

// one image per wave; the 2-waves-per-block form keeps the two waves of a block on one CU, which spreads a large
// batch evenly (2 per SIMD at 1024 frames) however the dispatcher deals the blocks
__global__ __launch_bounds__(64) void k_lsd_grow(const DevParams* __restrict__ Pp, float4* __restrict__ recAll, const double* __restrict__ mgAll,
                                                 const int* __restrict__ orderAll, const int* __restrict__ nDefined,
                                                 uint2* __restrict__ regOverflow, float* __restrict__ segAll,
                                                 int* __restrict__ nSeg, int maxSeg, int img0, int nimg) {
  lsd_grow_image<1>(Pp, recAll, mgAll, orderAll, nDefined, regOverflow, segAll, nSeg, maxSeg, img0, nimg);
}
// The relaxations' fallback without a host look: one wave per image, and only the images whose relaxation did not reach its fixed
// point in the rounds that were launched (or ran out of a capacity) are grown again, sequentially; the others leave at once.
__global__ __launch_bounds__(64) void k_lsd_grow_unsettled(const DevParams* __restrict__ Pp, float4* __restrict__ recAll, const double* __restrict__ mgAll,
                                                           const int* __restrict__ orderAll, const int* __restrict__ nDefined,
                                                           uint2* __restrict__ regOverflow, float* __restrict__ segAll,
                                                           int* __restrict__ nSeg, int maxSeg, int img0, int nimg,
                                                           RxCtl* __restrict__ ctl) {
  RxCtl& c = ctl[blockIdx.x + img0];
  if (c.state == 2 && !c.overflow) return;
  if (threadIdx.x == 0) c.races = -1;                   // (noted for the host: this image took the slow path)
  lsd_grow_image<1>(Pp, recAll, mgAll, orderAll, nDefined, regOverflow, segAll, nSeg, maxSeg, img0, nimg);
}
__global__ __launch_bounds__(128) void k_lsd_grow2(const DevParams* __restrict__ Pp, float4* __restrict__ recAll, const double* __restrict__ mgAll,
                                                   const int* __restrict__ orderAll, const int* __restrict__ nDefined,
                                                   uint2* __restrict__ regOverflow, float* __restrict__ segAll,
                                                   int* __restrict__ nSeg, int maxSeg, int img0, int nimg) {
  lsd_grow_image<2>(Pp, recAll, mgAll, orderAll, nDefined, regOverflow, segAll, nSeg, maxSeg, img0, nimg);
}

__global__ __launch_bounds__(64, 4) void k_lsd_grow_spec(const DevParams* __restrict__ Pp, float4* __restrict__ recAll, const double* __restrict__ mgAll,
                                                      const int* __restrict__ orderAll, const int* __restrict__ nDefined,
                                                      uint2* __restrict__ regOverflow, float* __restrict__ segAll,
                                                      int* __restrict__ nSeg, int maxSeg, int img0, int nimg, LsdRectItem* __restrict__ rectAll,
    double* __restrict__ rectWAll) {
  lsd_grow_image_spec<1>(Pp, recAll, mgAll, orderAll, nDefined, regOverflow, segAll, nSeg, maxSeg, img0, nimg, rectAll, rectWAll);
}
__global__ __launch_bounds__(128, 4) void k_lsd_grow2_spec(const DevParams* __restrict__ Pp, float4* __restrict__ recAll, const double* __restrict__ mgAll,
                                                        const int* __restrict__ orderAll, const int* __restrict__ nDefined,
                                                        uint2* __restrict__ regOverflow, float* __restrict__ segAll,
                                                        int* __restrict__ nSeg, int maxSeg, int img0, int nimg, LsdRectItem* __restrict__ rectAll,
    double* __restrict__ rectWAll) {
  lsd_grow_image_spec<2>(Pp, recAll, mgAll, orderAll, nDefined, regOverflow, segAll, nSeg, maxSeg, img0, nimg, rectAll, rectWAll);
}

// ---------------------------------------------------------------------------
// k_lsd_rect: region2rect + segment end points (lsd.cpp region2rect / get_theta, refine = NONE) of the regions the sequential
// grower left in the arena, off the grower's serial chain.  RECT_WPI waves per image; a wave takes 64 regions at a time, ONE
// REGION PER LANE: the weighted sums are accumulated in list order by the lane (the same adds in the same order as the
// sequential loop; three independent chains), the inertia pass likewise, the extent pass is order independent.  A region
// longer than RECT_LANE_MAX pixels would hold its wave back: those are done afterwards by the whole wave
// (lsd_region2rect, list-order sums by three lanes, products 64 at a time).
// ---------------------------------------------------------------------------
constexpr int RECT_WPI = 16;
constexpr int RECT_LANE_MAX = 160;

__global__ __launch_bounds__(64) void k_lsd_rect(const DevParams* __restrict__ Pp, const LsdRectItem* __restrict__ rectAll,
                                                 const uint2* __restrict__ arenaAll, const double* __restrict__ mgAll,
                                                 const double* __restrict__ rectWAll, const int* __restrict__ nSeg,
                                                 float* __restrict__ segAll, int maxSeg, int img0) {
  __shared__ double st[3][64];
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0, lane = threadIdx.x;
  const int W = P.LW;
  const int64_t npix = (int64_t)W * P.LH;
  const int n = min(nSeg[img], maxSeg);
  const LsdRectItem* items = rectAll + (int64_t)img * maxSeg;
  const uint2* arena = arenaAll + img * npix;
  const double* mg = mgAll ? mgAll + img * npix : nullptr;
  const double* wlist = (mgAll && rectWAll) ? rectWAll + img * npix : nullptr;     // CV_64F pipeline: weights beside the lists
  float* seg = segAll + (int64_t)img * maxSeg * 4;
  const double prec = P.prec, scale = P.lsdScale;
  for (int t0 = blockIdx.x * 64; t0 < n; t0 += RECT_WPI * 64) {
    const int t = t0 + lane;
    LsdRectItem it;
    it.off = 0; it.cnt = 0; it.reg_angle = 0.0;
    if (t < n) it = items[t];
    const bool small = t < n && it.cnt <= RECT_LANE_MAX;
    if (small) {
      const uint2* lst = arena + it.off;
      const double* wl = wlist ? wlist + it.off : nullptr;
      auto weight = [&](const uint2 e, int k) -> double {
        return wl ? wl[k] : mg ? mg[(int)(e.x >> 16) * W + (int)(e.x & 0xFFFFu)] : sqrt((double)(int)e.y / 4.0);
      };
      const int cnt = it.cnt;
      double sx = 0.0, sy = 0.0, sw = 0.0;
      for (int k = 0; k < cnt; ++k) {
        const uint2 e = lst[k];
        const double w = weight(e, k);
        sx += (double)(int)(e.x & 0xFFFFu) * w;
        sy += (double)(int)(e.x >> 16) * w;
        sw += w;
      }
      const double x = sx / sw, y = sy / sw;
      double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
      for (int k = 0; k < cnt; ++k) {
        const uint2 e = lst[k];
        const double w = weight(e, k);
        const double dx = (double)(int)(e.x & 0xFFFFu) - x, dy = (double)(int)(e.x >> 16) - y;
        Ixx += dy * dy * w;
        Iyy += dx * dx * w;
        Ixy -= dx * dy * w;
      }
      const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
      const bool wide = fabs(Ixx) > fabs(Iyy);
      double theta = (double)fast_atan2_deg(wide ? (float)(lambda - Ixx) : (float)Ixy, wide ? (float)Ixy : (float)(lambda - Iyy));
      theta *= D_DEG2RAD;
      if (lsd_angle_diff(theta, it.reg_angle) > prec) theta += D_PI;
      double dxr, dyr;
      sincos(theta, &dyr, &dxr);
      double l_min = 0, l_max = 0;
      for (int k = 0; k < cnt; ++k) {
        const uint2 e = lst[k];
        const double l = ((double)(int)(e.x & 0xFFFFu) - x) * dxr + ((double)(int)(e.x >> 16) - y) * dyr;
        l_max = fmax(l_max, l);
        l_min = fmin(l_min, l);
      }
      float4 o;
      double e0 = x + l_min * dxr, e1 = y + l_min * dyr, e2 = x + l_max * dxr, e3 = y + l_max * dyr;
      e0 += 0.5; e1 += 0.5; e2 += 0.5; e3 += 0.5;
      if (scale != 1) { e0 /= scale; e1 /= scale; e2 /= scale; e3 /= scale; }
      o.x = (float)e0; o.y = (float)e1; o.z = (float)e2; o.w = (float)e3;
      reinterpret_cast<float4*>(seg)[t] = o;
    }
    // the long ones of this group of 64, one after the other, by the whole wave
    unsigned long long big = __builtin_amdgcn_ballot_w64(t < n && !small);
    while (big) {
      const int j = __ffsll((long long)big) - 1;
      big &= big - 1ull;
      const int off = rl_i(it.off, j), cnt = rl_i(it.cnt, j);
      const double ra = __longlong_as_double(((long long)rl_i(__double2hiint(it.reg_angle), j) << 32) |
                                             (unsigned long long)(unsigned)rl_i(__double2loint(it.reg_angle), j));
      lsd_wave_sync();
      lsd_region2rect(nullptr, arena + off, st, cnt, ra, prec, scale, lane, seg, t0 + j, maxSeg, 0, mg, W);
    }
  }
}

// ---------------------------------------------------------------------------
// k_keylines: one workgroup per image.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_keylines(const DevParams* __restrict__ Pp, const float* __restrict__ seg,
                                                  const int* __restrict__ nSeg, int maxSeg,
                                                  pli_keyline* __restrict__ tmpKL, uint8_t* __restrict__ table,
                                                  int64_t recordBytes, int64_t offCounts, int64_t offKl0,
                                                  int64_t offKl1, int img0) {
  __shared__ int s_wc[4];
  __shared__ int s_base;
  __shared__ unsigned long long keys[4096];
  const DevParams& P = *Pp;
  const int img = blockIdx.x + img0, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = nSeg[img];
  const float* sg = seg + (int64_t)img * maxSeg * 4;
  pli_keyline* tk = tmpKL + (int64_t)img * P.maxLines;
  const int imgW = P.W, imgH = P.H;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int i = c0 + tid;
    bool keep = false;
    pli_keyline kl;
    if (i < n) {
      float e0 = sg[4 * i], e1 = sg[4 * i + 1], e2 = sg[4 * i + 2], e3 = sg[4 * i + 3];
      if (e0 < 0) e0 = 0;
      if (e0 >= imgW) e0 = (float)imgW - 1.0f;
      if (e2 < 0) e2 = 0;
      if (e2 >= imgW) e2 = (float)imgW - 1.0f;
      if (e1 < 0) e1 = 0;
      if (e1 >= imgH) e1 = (float)imgH - 1.0f;
      if (e3 < 0) e3 = 0;
      if (e3 >= imgH) e3 = (float)imgH - 1.0f;
      const double ddx = (double)__fsub_rn(e0, e2), ddy = (double)__fsub_rn(e1, e3);
      const double length = (double)(float)sqrt(ddx * ddx + ddy * ddy);
      keep = length > P.minLength;
      if (keep) {
        kl.startPointX = e0; kl.startPointY = e1; kl.endPointX = e2; kl.endPointY = e3;
        kl.sPointInOctaveX = e0; kl.sPointInOctaveY = e1; kl.ePointInOctaveX = e2; kl.ePointInOctaveY = e3;
        kl.lineLength = (float)length;
        const int ix1 = cv_round_f(e0), iy1 = cv_round_f(e1), ix2 = cv_round_f(e2), iy2 = cv_round_f(e3);
        kl.numOfPixels = max(abs(ix2 - ix1), abs(iy2 - iy1)) + 1;
        kl.angle = (float)atan2((double)__fsub_rn(e3, e1), (double)__fsub_rn(e2, e0));
        kl.octave = 0;
        kl.size = __fmul_rn(__fsub_rn(e2, e0), __fsub_rn(e3, e1));
        kl.response = __fdiv_rn(kl.lineLength, (float)max(imgW, imgH));
        kl.pt_x = __fdiv_rn(__fadd_rn(e2, e0), 2.f);
        kl.pt_y = __fdiv_rn(__fadd_rn(e3, e1), 2.f);
      }
    }
    unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    if (lane == 0) s_wc[wv] = __popcll(bal);
    __syncthreads();
    int pre = s_base;
    for (int k = 0; k < wv; ++k) pre += s_wc[k];
    const int pos = pre + __popcll(bal & ((1ull << lane) - 1ull));
    if (keep && pos < P.maxLines) {
      kl.class_id = pos;
      tk[pos] = kl;
    }
    __syncthreads();
    if (tid == 0) s_base += s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    __syncthreads();
  }
  const int M = min(s_base, P.maxLines);
  uint8_t* rec = table + (int64_t)(img >> 1) * recordBytes;
  const int eye = img & 1;
  // truncation flag (counts[6], one byte per eye): more segments passed the length cut than max_lines holds, i.e. the top-N
  // selection did not see all of them (the reference keeps every segment: raise pli_frontend_config.max_lines)
  if (tid == 0) rec[offCounts + 24 + eye] = s_base > P.maxLines ? 1 : 0;
  pli_keyline* out = reinterpret_cast<pli_keyline*>(rec + (eye ? offKl1 : offKl0));
  const int nf = P.lsdNFeatures;
  __threadfence_block();
  __syncthreads();
  if (M > nf && nf != 0) {
    // top-N by response, equal responses keep detection order (stable): rank by counting.  A line's place in that order is ONE
    // 64-bit key — the bits of its (positive) response above the complement of its index — and its rank the number of larger keys;
    // the keys are tiled through LDS, a wave takes 64 of them with one load (a key per lane) and hands them round with readlane
    // (2 readlanes + a 64-bit compare + an add per key; the earlier form read every response from LDS in every thread, one
    // dependent LDS round trip per comparison: 189 us per image, a single pair's third-longest kernel).
    auto keyOf = [](float resp, int idx) -> unsigned long long {
      return ((unsigned long long)__float_as_uint(resp) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
    };
    if (M <= 4096) {
      // ... and only for the nf lines that are kept: the nf-th largest key by radix select (eight passes of 8 bits from the top: a
      // 256-bin LDS histogram of the keys that match the bits chosen so far, one wave finds the bin of the k-th from the top),
      // then the lines at or above it (keys are distinct: exactly nf) rank themselves among each other
      __shared__ int s_hist[256];
      __shared__ unsigned long long s_pref, s_mask;
      __shared__ int s_k, s_nsel;
      __shared__ unsigned short s_sel[4096];
      for (int j = tid; j < M; j += 256) keys[j] = keyOf(tk[j].response, j);
      if (tid == 0) { s_pref = 0ull; s_mask = 0ull; s_k = nf - 1; s_nsel = 0; }
      for (int shift = 56; shift >= 0; shift -= 8) {
        s_hist[tid] = 0;
        __syncthreads();
        const unsigned long long pref = s_pref, mask = s_mask;
        for (int j = tid; j < M; j += 256) {
          const unsigned long long kj = keys[j];
          if ((kj & mask) == pref) atomicAdd(&s_hist[(int)((kj >> shift) & 255ull)], 1);
        }
        __syncthreads();
        if (tid < 64) {                              // lane l: the bins 255 - 4l .. 252 - 4l, from the top
          const int b0 = 255 - 4 * tid;
          const int h0 = s_hist[b0], h1 = s_hist[b0 - 1], h2 = s_hist[b0 - 2], h3 = s_hist[b0 - 3];
          const int sum = h0 + h1 + h2 + h3;
          int inc = sum;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int t2 = __shfl_up(inc, o, 64);
            if (tid >= o) inc += t2;
          }
          const int k = s_k, before = inc - sum;
          if (before <= k && k < inc) {
            int b = b0, c = before;
            if (k >= c + h0) { c += h0; b = b0 - 1; if (k >= c + h1) { c += h1; b = b0 - 2; if (k >= c + h2) { c += h2; b = b0 - 3; } } }
            s_pref = pref | ((unsigned long long)b << shift);
            s_mask = mask | (255ull << shift);
            s_k = k - c;
          }
        }
        __syncthreads();
      }
      const unsigned long long kth = s_pref;
      for (int j = tid; j < M; j += 256)
        if (keys[j] >= kth) s_sel[atomicAdd(&s_nsel, 1)] = (unsigned short)j;
      __syncthreads();
      for (int i0 = 0; i0 < nf; i0 += 256) {
        const int si = i0 + tid;
        const int i = si < nf ? (int)s_sel[si] : 0;
        const unsigned long long ki = si < nf ? keys[i] : ~0ull;
        int rank = 0;
        for (int j = 0; j < nf; j += 64) {
          const unsigned long long kv = j + lane < nf ? keys[s_sel[j + lane]] : 0ull;
          const int lo = (int)(unsigned)kv, hi = (int)(unsigned)(kv >> 32);
#pragma unroll
          for (int u = 0; u < 64; ++u) {
            const unsigned long long kj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, u) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readlane(lo, u);
            rank += kj > ki ? 1 : 0;
          }
        }
        if (si < nf) {
          pli_keyline kl = tk[i];
          kl.class_id = rank;
          out[rank] = kl;
        }
      }
    } else
    for (int i0 = 0; i0 < M; i0 += 256) {
      const int i = i0 + tid;
      const unsigned long long ki = i < M ? keyOf(tk[i].response, i) : ~0ull;
      int rank = 0;
      for (int j0 = 0; j0 < M; j0 += 4096) {
        const int m = min(4096, M - j0);
        __syncthreads();
        for (int j = tid; j < m; j += 256) keys[j] = keyOf(tk[j0 + j].response, j0 + j);
        __syncthreads();
        for (int j = 0; j < m; j += 64) {
          const unsigned long long kv = j + lane < m ? keys[j + lane] : 0ull;      // (0: smaller than every key)
          const int lo = (int)(unsigned)kv, hi = (int)(unsigned)(kv >> 32);
#pragma unroll
          for (int u = 0; u < 64; ++u) {
            const unsigned long long kj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, u) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readlane(lo, u);
            rank += kj > ki ? 1 : 0;
          }
        }
      }
      if (i < M && rank < nf) {
        pli_keyline kl = tk[i];
        kl.class_id = rank;
        out[rank] = kl;
      }
    }
    if (tid == 0) reinterpret_cast<int*>(rec + offCounts)[2 + eye] = nf;
  } else {
    const int cnt = min(M, P.klCap);
    for (int i = tid; i < cnt; i += 256) out[i] = tk[i];
    if (tid == 0) reinterpret_cast<int*>(rec + offCounts)[2 + eye] = cnt;
  }
}

// ---------------------------------------------------------------------------
// k_sobel: 3x3 Sobel dx, dy (CV_16S) with REFLECT_101 on the blurred image.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sobel(const uint8_t* __restrict__ in, int64_t imgStride, int W, int H,
                                               int pitch, short2* __restrict__ dxy, int img0) {
  // 4 adjacent pixels per thread: ONE aligned dword of each of the three rows per thread — the byte on its left and the byte on its
  // right come from the neighbouring lanes' dwords (the first and the last lane of a wave load theirs), where the earlier form
  // issued 18 byte loads per thread — and one 16-byte store of interleaved (dx, dy).  (Rows are pitch = align64(W) bytes: a dword at
  // x0 < W is inside the row.)
  const int img = blockIdx.z + img0, y = blockIdx.y, x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int lane = threadIdx.x & 63;
  const uint8_t* base = in + (int64_t)img * imgStride;
  const uint8_t* rows[3] = {base + (int64_t)reflect101(y - 1, H) * pitch, base + (int64_t)y * pitch, base + (int64_t)reflect101(y + 1, H) * pitch};
  const bool live = x0 < W;
  int v[3][6];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    unsigned cur = 0u;
    if (live) cur = *reinterpret_cast<const unsigned*>(rows[r] + x0);
    unsigned prev = (unsigned)__shfl_up((int)cur, 1, 64), next = (unsigned)__shfl_down((int)cur, 1, 64);
    if (lane == 0 && live && x0 >= 4) prev = *reinterpret_cast<const unsigned*>(rows[r] + x0 - 4);
    if (lane == 63 && x0 + 4 < W) next = *reinterpret_cast<const unsigned*>(rows[r] + x0 + 4);
    v[r][1] = cur & 0xFF; v[r][2] = (cur >> 8) & 0xFF; v[r][3] = (cur >> 16) & 0xFF; v[r][4] = cur >> 24;
    v[r][0] = x0 == 0 ? v[r][2] : (int)(prev >> 24);              // REFLECT_101: column -1 is column 1
    v[r][5] = (int)(next & 0xFF);
  }
  if (!live) return;
  if (x0 + 4 >= W) {
    // the thread that holds the last column(s): columns past the image are not stored, column W is column W - 2
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 1; k < 6; ++k)
        if (x0 + k - 1 >= W) v[r][k] = rows[r][reflect101(min(x0 + k - 1, W), W)];
  }
  short2 o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int gx = (v[0][k + 2] - v[0][k]) + 2 * (v[1][k + 2] - v[1][k]) + (v[2][k + 2] - v[2][k]);
    const int gy = (v[2][k] - v[0][k]) + 2 * (v[2][k + 1] - v[0][k + 1]) + (v[2][k + 2] - v[0][k + 2]);
    o[k] = make_short2((short)gx, (short)gy);      // interleaved: k_lbd gathers both with one 4-byte load
  }
  short2* dst = dxy + (int64_t)img * W * H + (int64_t)y * W + x0;
  if (x0 + 3 < W && (((int64_t)y * W + x0) & 3) == 0 && ((int64_t)W * H & 3) == 0) {
    *reinterpret_cast<int4*>(dst) = make_int4(*reinterpret_cast<int*>(&o[0]), *reinterpret_cast<int*>(&o[1]),
                                              *reinterpret_cast<int*>(&o[2]), *reinterpret_cast<int*>(&o[3]));
  } else {
    for (int k = 0; k < 4 && x0 + k < W; ++k) dst[k] = o[k];
  }
}

// ---------------------------------------------------------------------------
// k_lbd: one wave per line.  Lane h walks row h of the 63-row line support
// region sequentially (the reference accumulates coordinates and row sums in
// float, so the order is part of the result); band sums are then accumulated
// in ascending row order, one lane per (band, quantity).
// ---------------------------------------------------------------------------
__constant__ int c_lbd_comb[32][2] = {
    {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
    {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
    {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};


__global__ __launch_bounds__(64) void k_lbd(const DevParams* __restrict__ Pp, const LbdCoef* __restrict__ coef,
                                            const short2* __restrict__ dxya,
                                            uint8_t* __restrict__ table, int64_t recordBytes, int64_t offCounts,
                                            int64_t offKl0, int64_t offKl1, int64_t offLd0, int64_t offLd1,
                                            float* __restrict__ dbgFloat, int img0) {
  __shared__ float rowS[63][4];      // pgdL, ngdL, pgdO, ngdO row sums (already x gaussCoefG)
  __shared__ float band[9][8];       // pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2
  __shared__ float des[72];
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0, li = blockIdx.x, lane = threadIdx.x;
  uint8_t* rec = table + (int64_t)(img >> 1) * recordBytes;
  const int eye = img & 1;
  const int n = reinterpret_cast<const int*>(rec + offCounts)[2 + eye];
  if (li >= n) return;
  const pli_keyline kl = reinterpret_cast<const pli_keyline*>(rec + (eye ? offKl1 : offKl0))[li];
  const int W = P.W, H = P.H;
  const short2* pdxy = dxya + (int64_t)img * W * H;
  const short heightOfLSP = 63, halfHeight = 31;
  const short imageWidth = (short)(W - 1), imageHeight = (short)(H - 1);
  const short lengthOfLSP = (short)kl.numOfPixels;
  const short halfWidth = (short)((lengthOfLSP - 1) / 2);
  const float midX = (float)(0.5 * ((double)__fadd_rn(kl.sPointInOctaveX, kl.ePointInOctaveX)));
  const float midY = (float)(0.5 * ((double)__fadd_rn(kl.sPointInOctaveY, kl.ePointInOctaveY)));
  float dL0, dL1;                                      // cos( direction ), sin( direction ), binary_descriptor_custom.cpp:1130-1131
  sincos_of_float(kl.angle, (P.parityFlags & PLI_PARITY_TRIG_F32_LBD) != 0, &dL1, &dL0);
  const float dO0 = -dL1, dO1 = dL0;
  if (lane < heightOfLSP) {
    float sCorX0 = __fadd_rn(__fadd_rn(__fmul_rn(-dL0, (float)halfWidth), __fmul_rn(dL1, (float)halfHeight)), midX);
    float sCorY0 = __fadd_rn(__fsub_rn(__fmul_rn(-dL1, (float)halfWidth), __fmul_rn(dL0, (float)halfHeight)), midY);
    for (int h = 0; h < lane; ++h) {
      sCorX0 = __fsub_rn(sCorX0, dL1);
      sCorY0 = __fadd_rn(sCorY0, dL0);
    }
    float sCorX = sCorX0, sCorY = sCorY0;
    float pgdL = 0, ngdL = 0, pgdO = 0, ngdO = 0;
    // (unrolled: the gathers of several steps are in flight together, the float sums stay in walking order)
#pragma unroll 8
    for (short wID = 0; wID < lengthOfLSP; wID++) {
      short tempCor = (short)roundf(sCorX);
      short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
      tempCor = (short)roundf(sCorY);
      short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
      const short2 g = pdxy[(int)yCor * W + xCor];
      const float dx = (float)g.x, dy = (float)g.y;
      const float gDL = __fadd_rn(__fmul_rn(dx, dL0), __fmul_rn(dy, dL1));
      const float gDO = __fadd_rn(__fmul_rn(dx, dO0), __fmul_rn(dy, dO1));
      if (gDL > 0) pgdL = __fadd_rn(pgdL, gDL); else ngdL = __fsub_rn(ngdL, gDL);
      if (gDO > 0) pgdO = __fadd_rn(pgdO, gDO); else ngdO = __fsub_rn(ngdO, gDO);
      sCorX = __fadd_rn(sCorX, dL0);
      sCorY = __fadd_rn(sCorY, dL1);
    }
    const float cg = coef->G[lane];
    rowS[lane][0] = __fmul_rn(cg, pgdL);
    rowS[lane][1] = __fmul_rn(cg, ngdL);
    rowS[lane][2] = __fmul_rn(cg, pgdO);
    rowS[lane][3] = __fmul_rn(cg, ngdO);
  }
  __syncthreads();
  for (int i = lane; i < 72; i += 64) {
    const int b = i >> 3, t = i & 7;
    // t: 0 pgdL 1 ngdL 2 pgdL2 3 ngdL2 4 pgdO 5 ngdO 6 pgdO2 7 ngdO2
    const int src = (t & 1) + ((t & 4) ? 2 : 0);
    const bool sq = (t & 2) != 0;
    float acc = 0;
    const int h0 = max(0, (b - 1) * 7), h1 = min(62, (b + 2) * 7 - 1);
    for (int h = h0; h <= h1; ++h) {
      const int hb = h / 7;
      const int off = (hb == b) ? 7 : (hb == b + 1 ? 14 : 0);   // own band / row below feeds the band above / row above feeds the band below
      const float c = coef->L[h % 7 + off];
      const float r = rowS[h][src];
      if (sq) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(c, c), __fmul_rn(r, r)));
      else acc = __fadd_rn(acc, __fmul_rn(c, r));
    }
    band[b][t] = acc;
  }
  __syncthreads();
  const float invN2 = (float)(1.0 / (7 * 2.0)), invN3 = (float)(1.0 / (7 * 3.0));
  for (int i = lane; i < 72; i += 64) {
    const int b = i >> 3, j = i & 7;
    const float invN = (b == 0 || b == 8) ? invN2 : invN3;
    // des index j: 0 mean pgdL, 1 mean ngdL, 2 mean pgdO, 3 mean ngdO, 4..7 the matching std
    const int m = j & 3;
    const int lin = (m & 1) + ((m & 2) ? 4 : 0);       // band[] index of the linear sum
    const float temp = __fmul_rn(band[b][lin], invN);
    float v = temp;
    if (j >= 4) v = sqrtf(__fsub_rn(__fmul_rn(band[b][lin + 2], invN), __fmul_rn(temp, temp)));
    des[i] = v;
  }
  __syncthreads();
  {
    // The three normalisation sums run over the 72 entries in index order (float addition order is part of the result).
    // Every lane keeps its entries in registers, the squares are computed in parallel and the ordered sums read them
    // with v_readlane (a scalar operand per term) instead of one dependent LDS read per term.
    const float d0 = des[lane], d1 = lane < 8 ? des[64 + lane] : 0.f;            // entry lane and entry 64 + lane
    const float q0 = __fmul_rn(d0, d0), q1 = __fmul_rn(d1, d1);
    float tempM = 0, tempS = 0;
#pragma unroll
    for (int i = 0; i < 72; ++i) {
      const float q = rl_f(i < 64 ? q0 : q1, i & 63);
      if (i & 4) tempS = __fadd_rn(tempS, q); else tempM = __fadd_rn(tempM, q);
    }
    tempM = __fdiv_rn(1.f, sqrtf(tempM));
    tempS = __fdiv_rn(1.f, sqrtf(tempS));
    float v0 = __fmul_rn(d0, (lane & 4) ? tempS : tempM), v1 = __fmul_rn(d1, (lane & 4) ? tempS : tempM);   // (64 + lane) & 4 == lane & 4
    if ((double)v0 > 0.4) v0 = (float)0.4;
    if ((double)v1 > 0.4) v1 = (float)0.4;
    const float p0 = __fmul_rn(v0, v0), p1 = __fmul_rn(v1, v1);
    float temp = 0;
#pragma unroll
    for (int i = 0; i < 72; ++i) temp = __fadd_rn(temp, rl_f(i < 64 ? p0 : p1, i & 63));
    temp = __fdiv_rn(1.f, sqrtf(temp));
    des[lane] = __fmul_rn(v0, temp);
    if (lane < 8) des[64 + lane] = __fmul_rn(v1, temp);
  }
  __syncthreads();
  if (dbgFloat) {
    for (int i = lane; i < 72; i += 64) dbgFloat[((int64_t)img * P.klCap + li) * 72 + i] = des[i];
  }
  if (lane < 32) {
    const float* f1 = &des[8 * c_lbd_comb[lane][0]];
    const float* f2 = &des[8 * c_lbd_comb[lane][1]];
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r |= (f1[i] > f2[i]) ? (1u << i) : 0u;
    (rec + (eye ? offLd1 : offLd0))[(int64_t)li * 32 + lane] = (uint8_t)r;
  }
}

}  // namespace pli
