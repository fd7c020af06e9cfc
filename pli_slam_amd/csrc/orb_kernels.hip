// ORB point front-end kernels for gfx950 (wave64).
//
//   k_ingest        input images -> level 0 of the device pyramid
//   k_resize_level  ORBextractor::ComputePyramid            (ORBextractor.cc:1152-1177)
//   k_fast_cells    per-cell cv::FAST x2 thresholds + NMS   (ORBextractor.cc:787-854)
//   k_octree        ORBextractor::DistributeOctTree         (ORBextractor.cc:479-761)
//   k_blur          cv::GaussianBlur u8 fixed point         (ORBextractor.cc:1115, also LSD/LBD)
//   k_describe      IC_Angle + computeOrbDescriptor + table (ORBextractor.cc:75-145,1105-1147)
//
// All arithmetic that decides a result is integer or explicitly ordered IEEE
// float (see device_prims.hpp); the library is built with -ffp-contract=off.
#include "kernels.hpp"
#include <type_traits>
#include "device_prims.hpp"

namespace pli {

__constant__ signed char c_orb_pattern[1024] = {
#include "../../include/pli_orb_pattern.inc"
};

// ---------------------------------------------------------------------------
// k_ingest: copy nframes x 2 images (arbitrary stride) into level 0 — or, when the eye has rectification
// maps, cv::remap(src, dst, mapx, mapy, INTER_LINEAR) (BORDER_CONSTANT 0) fused into the copy
// (the driver's rectification, Examples/Stereo/stereo_euroc.cc:166-167).  OpenCV 3.3.1 remap for 8U:
// coordinates to 1/32 px with cvRound(map*32) (saturated to short), bilinear weights (32-fx)(32-fy)*32 ...
// as shorts summing to 2^15 — the all-in-one-pixel weight 32768 does not fit a short and initInterTab2D
// repairs the block to {32767, 0, 0, 1} — result (sum + 2^14) >> 15.
// ---------------------------------------------------------------------------
// k_ingest_copy16: the plain copy (no rectification maps) when rows, strides and both image bases are 16-byte aligned and the width
// is a multiple of 16: one 16-byte load and one 16-byte store per thread, the (row, column) pairs of an image spread over the grid
// (the byte form below issues four byte loads per thread and leaves most of a 256-thread row block idle at 752 columns).
__global__ __launch_bounds__(256) void k_ingest_copy16(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right, int64_t stride,
                                                       int64_t frameStride, uint8_t* __restrict__ pyr, int64_t pyrBlock, int W16, int H,
                                                       int pitch, int img0) {
  const int img = blockIdx.y + img0;
  const uint8_t* src = ((img & 1) ? right : left) + (int64_t)(img >> 1) * frameStride;
  uint8_t* dst = pyr + (int64_t)img * pyrBlock;
  const int n = W16 * H;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int y = i / W16, x = (i - y * W16) * 16;
    *reinterpret_cast<uint4*>(dst + (int64_t)y * pitch + x) = *reinterpret_cast<const uint4*>(src + (int64_t)y * stride + x);
  }
}

__global__ __launch_bounds__(256) void k_ingest(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                int64_t stride, int64_t frameStride, uint8_t* __restrict__ pyr,
                                                int64_t pyrBlock, int W, int H, int pitch,
                                                const float2* __restrict__ mapL, const float2* __restrict__ mapR, int img0) {
  const int img = blockIdx.z + img0;
  const uint8_t* src = ((img & 1) ? right : left) + (int64_t)(img >> 1) * frameStride;
  const float2* map = (img & 1) ? mapR : mapL;
  uint8_t* dst = pyr + (int64_t)img * pyrBlock;
  const int y = blockIdx.y;
  const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x4 >= W) return;
  uint8_t v[4];
  if (!map) {
    const uint8_t* s = src + (int64_t)y * stride + x4;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (x4 + k < W) ? s[k] : 0;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = 0;
      if (x4 + k >= W) continue;
      const float2 m = map[(int64_t)y * W + x4 + k];
      const int sx = cv_round_f(__fmul_rn(m.x, 32.f)), sy = cv_round_f(__fmul_rn(m.y, 32.f));
      const int ix = min(max(sx >> 5, -32768), 32767), iy = min(max(sy >> 5, -32768), 32767);
      const int fx = sx & 31, fy = sy & 31;
      int w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
      if ((fx | fy) == 0) { w0 = 32767; w3 = 1; }
      if (ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0) continue;        // fully outside: border value 0
      auto at = [&](int xx, int yy) -> int {
        return (xx >= 0 && yy >= 0 && xx < W && yy < H) ? (int)src[(int64_t)yy * stride + xx] : 0;
      };
      const int sum = at(ix, iy) * w0 + at(ix + 1, iy) * w1 + at(ix, iy + 1) * w2 + at(ix + 1, iy + 1) * w3;
      v[k] = (uint8_t)min(max((sum + (1 << 14)) >> 15, 0), 255);
    }
  }
  *reinterpret_cast<uchar4*>(dst + (int64_t)y * pitch + x4) = make_uchar4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------
// k_resize_level: cv::resize INTER_LINEAR u8 (11-bit coefficients built on the
// host exactly as OpenCV builds them).
// tab: xofs[dw] | alpha[2*dw] (as int) | yofs[dh] | beta[2*dh]
// A workgroup makes a 256 x 16 block of the output: the source window of the block is staged in LDS with aligned
// 4-byte loads (0.4 loads per output pixel instead of four byte gathers), a thread keeps the column coefficients of
// its 4 adjacent outputs in registers for its 4 rows.  Windows that do not fit (scale factors above ~2.4) read
// the source directly.
// ---------------------------------------------------------------------------
constexpr int RZ_W = 640;     // bytes per staged source row
constexpr int RZ_H = 44;      // staged source rows

__global__ __launch_bounds__(256) void k_resize_level(const uint8_t* __restrict__ srcBase, int64_t srcImgStride,
                                                      int sw, int sh, int spitch, uint8_t* __restrict__ dstBase,
                                                      int64_t dstImgStride, int dw, int dh, int dpitch,
                                                      const int* __restrict__ tab, int img0) {
  __shared__ uint32_t S[RZ_H * RZ_W / 4];
  const int img = blockIdx.z + img0;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int dxa = blockIdx.x * 256, dya = blockIdx.y * 16;
  const int* xofs = tab;
  const int* alpha = tab + dw;
  const int* yofs = tab + 3 * dw;
  const int* beta = tab + 3 * dw + dh;
  const uint8_t* src = srcBase + (int64_t)img * srcImgStride;
  auto clampRow = [&](int y) { return y < 0 ? 0 : (y >= sh ? sh - 1 : y); };
  // source window of the block
  const int dxb = min(dxa + 255, dw - 1), dyb = min(dya + 15, dh - 1);
  const int colBase = xofs[dxa] & ~3;
  const int ndw = (min(xofs[dxb] + 1, sw - 1) - colBase + 4) >> 2;
  const int rya = clampRow(yofs[dya]), nrows = clampRow(yofs[dyb] + 1) - rya + 1;
  const bool staged = ndw * 4 <= RZ_W && nrows <= RZ_H;
  if (staged) {
    const float invNdw = 1.0f / (float)ndw;
    for (int i = tid; i < ndw * nrows; i += 256) {
      const int r = (int)(((float)i + 0.5f) * invNdw), d = i - r * ndw;          // exact: i < 160 * 44
      S[r * (RZ_W / 4) + d] = *reinterpret_cast<const uint32_t*>(src + (int64_t)(rya + r) * spitch + colBase + 4 * d);
    }
    __syncthreads();
  }
  const int dx0 = dxa + tx * 4;
  if (dx0 >= dw) return;
  int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int dx = min(dx0 + k, dw - 1);
    sx[k] = xofs[dx];
    sx1[k] = sx[k] + 1 < sw ? sx[k] + 1 : sw - 1;
    a0[k] = alpha[2 * dx];
    a1[k] = alpha[2 * dx + 1];
  }
  const uint8_t* Sb = reinterpret_cast<const uint8_t*>(S);
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int dy = dya + ty + 4 * r;
    if (dy >= dh) break;
    const int sy = yofs[dy];
    const int sy0 = clampRow(sy), sy1 = clampRow(sy + 1);
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t out[4];
    auto row = [&](auto p0, auto p1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r0 = p0[sx[k]] * a0[k] + p0[sx1[k]] * a1[k];
        const int r1 = p1[sx[k]] * a0[k] + p1[sx1[k]] * a1[k];
        int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        out[k] = (uint8_t)v;
      }
    };
    if (staged) {
      typedef __attribute__((address_space(3))) const uint8_t lds_u8;
      row((lds_u8*)(Sb + (sy0 - rya) * RZ_W) - colBase, (lds_u8*)(Sb + (sy1 - rya) * RZ_W) - colBase);
    } else {
      row(src + (int64_t)sy0 * spitch, src + (int64_t)sy1 * spitch);
    }
    uint8_t* dst = dstBase + (int64_t)img * dstImgStride + (int64_t)dy * dpitch + dx0;
    *reinterpret_cast<uchar4*>(dst) = make_uchar4(out[0], out[1], out[2], out[3]);
  }
}

// ---------------------------------------------------------------------------
// k_fast_cells: one workgroup per FAST cell.
// The cell sub-image (<= 66 x 66) is staged in LDS; every thread evaluates the
// 16-pixel ring of its pixels with two 16-bit sign masks (9-contiguous test by
// shift-and), the exact corner score only for pixels that pass at minTh; 3x3
// NMS inside the cell interior; the two-threshold rule; then an ordered
// (raster) compaction with wave ballots + popcounts.
// Output: packed (y<<20 | x<<8 | score), coordinates relative to (minBX,minBY).
// ---------------------------------------------------------------------------
constexpr int FT_PITCH = 72;
constexpr int FT_SP = 68;        // pitch of the padded score map (64 + 2, rounded to 4 bytes)

__device__ __forceinline__ bool has9(unsigned m16) {
  unsigned m = m16 | (m16 << 16);
  unsigned r = m & (m >> 1);
  r &= r >> 2;
  r &= r >> 4;      // bit i: bits i..i+7 set
  r &= m >> 8;      // bit i: bits i..i+8 set
  return (r & 0xFFFFu) != 0;
}

// max over the 16 arcs of 9 of the minimum of a[] (a has 16 entries, circular): 9 = 3 x 3, so the minimum of an arc is the minimum
// of three minima of three — 16 + 16 v_min3 and 8 v_max3 (round 6; the doubling form min2 -> min4 -> min8 -> +1 took 93 two-input ops)
__device__ __forceinline__ int tx_min3(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int tx_max3(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int arc9_maxmin(const int a[16]) {
  int m3[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) m3[i] = tx_min3(a[i], a[(i + 1) & 15], a[(i + 2) & 15]);
  int m9[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) m9[i] = tx_min3(m3[i], m3[(i + 3) & 15], m3[(i + 6) & 15]);
  int t[6];
#pragma unroll
  for (int i = 0; i < 5; ++i) t[i] = tx_max3(m9[3 * i], m9[3 * i + 1], m9[3 * i + 2]);
  t[5] = m9[15];
  return max(tx_max3(t[0], t[1], t[2]), tx_max3(t[3], t[4], t[5]));
}

__global__ __launch_bounds__(256) void k_fast_cells(const DevParams* __restrict__ Pp, const uint8_t* __restrict__ pyr,
                                                    uint32_t* __restrict__ cellCand, int* __restrict__ cellCount, int img0) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[FT_PITCH * 68];
  // the score map with a border of zeros (pitch FT_SP, pixel (x, y) at (y + 1) * FT_SP + x + 1): the 8 neighbours of the non-maximum
  // suppression are read without bounds tests; zeroed once per cell with 4-byte stores (everything that is not a corner scores 0)
  __shared__ __attribute__((aligned(16))) uint8_t score[FT_SP * 66 + 4];
  __shared__ unsigned nmsBits[64 * 2];        // one bit per pixel of the 64 x 64 box: a strict maximum of its 3 x 3 block
  // corners at minTh (pixel indices): the DARK ones from the front, the BRIGHT ones from the back — a ring of 16 cannot hold an arc of 9
  // darker and an arc of 9 brighter samples, so a corner is one or the other, and each kind takes its score from its own pass (round 6:
  // one list whose lanes mixed the kinds ran both passes for every wave)
  __shared__ unsigned short clist[64 * 64];
  __shared__ int s_nc, s_nb;
  __shared__ int s_any;
  __shared__ int s_wc[64];     // survivors per (chunk, wave)
  __shared__ int s_off[64];    // their exclusive prefix
  __shared__ int s_base;
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0;
  const int cell = blockIdx.x;
  int lvl = 0;
#pragma unroll 1
  for (int l = 1; l < P.nlevels; ++l)
    if (cell >= P.lv[l].cellBase) lvl = l;
  const LevelGeom& G = P.lv[lvl];
  const int c = cell - G.cellBase;
  const int ci = c / G.nCols, cj = c - ci * G.nCols;
  const int iniX = G.minBX + cj * G.wCell, iniY = G.minBY + ci * G.hCell;
  int maxX = iniX + G.wCell + 6, maxY = iniY + G.hCell + 6;
  const int tid = threadIdx.x;
  const int64_t cellIdx = (int64_t)img * P.cellsPerImage + cell;
  if (iniY >= G.maxBY - 3 || iniX >= G.maxBX - 6) {
    if (tid == 0) cellCount[cellIdx] = 0;
    return;
  }
  if (maxX > G.maxBX) maxX = G.maxBX;
  if (maxY > G.maxBY) maxY = G.maxBY;
  const int cw = maxX - iniX, ch = maxY - iniY;
  const int iw = cw - 6, ih = ch - 6;
  if (iw <= 0 || ih <= 0) {
    if (tid == 0) cellCount[cellIdx] = 0;
    return;
  }
  const uint8_t* im = pyr + (int64_t)img * P.pyrBlock + G.offset;
  // (row, column) of a linear index without an integer division: exact for these sizes (index < 4096, width <= 64)
  // the rows of a level start 64-byte aligned, so the misalignment of the cell is the same in every row: the tile
  // keeps it (column ox = first cell column) and is filled with aligned 4-byte loads
  const int ox = iniX & 3, nd = (cw + ox + 3) >> 2;
  const float invNd = 1.0f / (float)nd, invIw = 1.0f / (float)iw;
  for (int i = tid; i < nd * ch; i += 256) {
    const int y = (int)(((float)i + 0.5f) * invNd), k = i - y * nd;
    reinterpret_cast<uint32_t*>(tile)[y * (FT_PITCH / 4) + k] =
        *reinterpret_cast<const uint32_t*>(im + (int64_t)(iniY + y) * G.pitch + (iniX - ox) + 4 * k);
  }
  if (tid == 0) { s_any = 0; s_nc = 0; s_nb = 0; }
  for (int i = tid; i < (FT_SP * 66 + 4) / 4; i += 256) reinterpret_cast<uint32_t*>(score)[i] = 0u;
  if (tid < 128) nmsBits[tid] = 0u;
  __syncthreads();
  const int minTh = P.minTh, iniTh = P.iniTh;
  const int npix = iw * ih;
  // pass 1, every pixel: the two 16-bit sign masks of the ring at minTh (one add/sub and one v_alignbit per test:
  // mask = mask << 1 | sign) and the 9-contiguous test; corners are listed, everything else scores 0
  for (int i = tid; i < npix; i += 256) {
    const int y = (int)(((float)i + 0.5f) * invIw), x = i - y * iw;
    const uint8_t* p = &tile[(y + 3) * FT_PITCH + x + 3 + ox];
    const int v = p[0];
    const int tD = v - minTh - 1, tB = v + minTh + 1;     // ring darker: r <= tD; brighter: r >= tB
    unsigned mDark = 0, mBright = 0;
#define FT_RING(o)                                                                        \
    {                                                                                     \
      const int r = p[o];                                                                 \
      mDark = __builtin_amdgcn_alignbit(mDark, (unsigned)(r - tD - 1), 31);   /* sign: r <= tD  <=> v - r > minTh */ \
      mBright = __builtin_amdgcn_alignbit(mBright, (unsigned)(tB - r - 1), 31); /* sign: r >= tB <=> r - v > minTh */ \
    }
    FT_RING(3 * FT_PITCH + 0) FT_RING(3 * FT_PITCH + 1) FT_RING(2 * FT_PITCH + 2) FT_RING(1 * FT_PITCH + 3)
    FT_RING(3) FT_RING(-1 * FT_PITCH + 3) FT_RING(-2 * FT_PITCH + 2) FT_RING(-3 * FT_PITCH + 1)
    FT_RING(-3 * FT_PITCH + 0) FT_RING(-3 * FT_PITCH - 1) FT_RING(-2 * FT_PITCH - 2) FT_RING(-1 * FT_PITCH - 3)
    FT_RING(-3) FT_RING(1 * FT_PITCH - 3) FT_RING(2 * FT_PITCH - 2) FT_RING(3 * FT_PITCH - 1)
#undef FT_RING
    const bool cd = has9(mDark & 0xFFFFu), cb = has9(mBright & 0xFFFFu);   // (bit order reversed: a contiguous arc stays one)
    if (cd) clist[atomicAdd(&s_nc, 1)] = (unsigned short)i;
    else if (cb) clist[4095 - atomicAdd(&s_nb, 1)] = (unsigned short)i;
  }
  __syncthreads();
  // pass 2, corners only (densely packed lanes): the exact corner score = the largest threshold at which the pixel is still a corner
  // of its kind = max over the 16 arcs of 9 of the smallest (centre - ring) resp. (ring - centre), minus one (cornerScore<16>)
  const int ncd = s_nc, ncb = s_nb, nc = ncd + ncb;
  auto cornerAt = [&](int j) -> int { return j < ncd ? clist[j] : clist[4095 - (j - ncd)]; };
  auto scorePass = [&](auto brightTag) {
    constexpr bool BRIGHT = decltype(brightTag)::value;
    const int n = BRIGHT ? ncb : ncd;
    for (int j = tid; j < n; j += 256) {
      const int i = BRIGHT ? clist[4095 - j] : clist[j];
      const int y = (int)(((float)i + 0.5f) * invIw), x = i - y * iw;
      const uint8_t* p = &tile[(y + 3) * FT_PITCH + x + 3 + ox];
      const int v = p[0];
      auto df = [&](int o) -> int { return BRIGHT ? (int)p[o] - v : v - (int)p[o]; };     // ring - centre (bright), centre - ring (dark)
      int d[16];
      d[0] = df(3 * FT_PITCH + 0);   d[1] = df(3 * FT_PITCH + 1);   d[2] = df(2 * FT_PITCH + 2);   d[3] = df(1 * FT_PITCH + 3);
      d[4] = df(3);                  d[5] = df(-1 * FT_PITCH + 3);  d[6] = df(-2 * FT_PITCH + 2);  d[7] = df(-3 * FT_PITCH + 1);
      d[8] = df(-3 * FT_PITCH + 0);  d[9] = df(-3 * FT_PITCH - 1);  d[10] = df(-2 * FT_PITCH - 2); d[11] = df(-1 * FT_PITCH - 3);
      d[12] = df(-3);                d[13] = df(1 * FT_PITCH - 3);  d[14] = df(2 * FT_PITCH - 2);  d[15] = df(3 * FT_PITCH - 1);
      score[(y + 1) * FT_SP + x + 1] = (uint8_t)(arc9_maxmin(d) - 1);
    }
  };
  scorePass(std::false_type{});
  scorePass(std::true_type{});
  __syncthreads();
  // Non-maximum suppression does not depend on the threshold (a survivor is strictly greater than its 8 neighbours inside the
  // interior; everything outside it, and every non-corner, scores 0 in the padded map).  Only corners can survive: the pass runs over
  // the corner list — dense lanes, 8 byte reads, no bounds tests — and leaves one bit per survivor (round 6: it used to visit every
  // pixel of the cell, a fifth of the kernel's instructions).
  for (int j = tid; j < nc; j += 256) {
    const int i = cornerAt(j);
    const int y = (int)(((float)i + 0.5f) * invIw), x = i - y * iw;
    const uint8_t* sp = &score[(y + 1) * FT_SP + x + 1];
    const int s = sp[0];
    const int n = max(tx_max3(tx_max3(sp[-FT_SP - 1], sp[-FT_SP], sp[-FT_SP + 1]), sp[-1], sp[1]), tx_max3(sp[FT_SP - 1], sp[FT_SP], sp[FT_SP + 1]));
    if (s > n) {                                          // (s > n >= 0: a corner)
      atomicOr(&nmsBits[y * 2 + (x >> 5)], 1u << (x & 31));
      if (s >= iniTh) s_any = 1;                          // benign race: all writers store 1
    }
  }
  const int nchunks = (npix + 255) / 256;
  __syncthreads();
  const int thr = s_any ? iniTh : 1;     // score > 0 <=> corner at minTh
  // ordered (raster) compaction in one go: per (chunk, wave) survivor counts by ballot, one 64-entry scan, then the writes
  const int lane = tid & 63, wv = tid >> 6;
  unsigned survBits = 0;
  unsigned long long rk = 0;             // rank inside (chunk, wave): 16 x 4 bits would overflow (0..63) -> two words of 8 x 8 bits
  unsigned long long rk2 = 0;
#pragma unroll 1
  for (int ck = 0; ck < nchunks; ++ck) {
    bool surv = false;
    const int i = ck * 256 + tid;
    if (i < npix) {
      const int y = (int)(((float)i + 0.5f) * invIw), x = i - y * iw;
      surv = ((nmsBits[y * 2 + (x >> 5)] >> (x & 31)) & 1u) != 0u && score[(y + 1) * FT_SP + x + 1] >= thr;
    }
    const unsigned long long bal = __ballot(surv);
    if (lane == 0) s_wc[ck * 4 + wv] = __popcll(bal);
    if (surv) {
      survBits |= 1u << ck;
      const unsigned long long r = (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
      if (ck < 8) rk |= r << (8 * ck); else rk2 |= r << (8 * (ck - 8));
    }
  }
  __syncthreads();
  if (wv == 0) {
    const int v = lane < nchunks * 4 ? s_wc[lane] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    s_off[lane] = inc - v;
    if (lane == 63) s_base = inc;
  }
  __syncthreads();
  uint32_t* out = cellCand + cellIdx * CELL_CAP;
#pragma unroll 1
  for (int ck = 0; ck < nchunks; ++ck) {
    if (!((survBits >> ck) & 1u)) continue;
    const int i = ck * 256 + tid;
    const int y = (int)(((float)i + 0.5f) * invIw), x = i - y * iw;
    const int r = (int)(((ck < 8 ? rk >> (8 * ck) : rk2 >> (8 * (ck - 8)))) & 0xFFull);
    const int pos = s_off[ck * 4 + wv] + r;
    if (pos < CELL_CAP) {
      const int xr = x + 3 + cj * G.wCell, yr = y + 3 + ci * G.hCell;
      out[pos] = ((uint32_t)yr << 20) | ((uint32_t)xr << 8) | (uint32_t)score[(y + 1) * FT_SP + x + 1];
    }
  }
  if (tid == 0) cellCount[cellIdx] = s_base < CELL_CAP ? s_base : CELL_CAP;
}

// ---------------------------------------------------------------------------
// k_octree: ORBextractor::DistributeOctTree, one workgroup per (level, image).
//
// The reference keeps a std::list of nodes, splits nodes with push_front of
// the non-empty children and, when the next full pass could overshoot N,
// splits the largest nodes first until the list reaches N.  Here the list is
// an array in list order that is rebuilt every round:
//   new list = [children in reverse creation order] ++ [unsplit nodes, old order]
// which is exactly what push_front/erase produce.  Creation order = processing
// order of the parents (list order in a normal pass; size-descending, newest
// first among equal sizes in the "careful" rounds) then child n1..n4.
// Keys never move: node_of[key] holds the list position of the key's node.
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ int block_excl_scan_1024(int* a, int n, int* scratch /*>= NT / 64 ints*/) {
  // in-place exclusive scan of a[0..n) (n <= 1024) by NT threads (256 or 1024); returns the total
  constexpr int PER = 1024 / NT, NW = NT / 64;
  const int tid = threadIdx.x;
  int v[PER], s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int i = tid * PER + k;
    v[k] = i < n ? a[i] : 0;
    s += v[k];
  }
  // wave inclusive scan of s
  int lane = tid & 63, wv = tid >> 6;
  int inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) scratch[wv] = inc;
  __syncthreads();
  int wbase = 0, total = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int w = scratch[k];
    if (k < wv) wbase += w;
    total += w;
  }
  int run = wbase + inc - s;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int i = tid * PER + k;
    if (i < n) a[i] = run;
    run += v[k];
  }
  __syncthreads();
  return total;
}

template <int MAXN>
struct OtNodes {
  short x0[MAXN], y0[MAXN], x1[MAXN], y1[MAXN];
  int cnt[MAXN];
  unsigned char isNew[MAXN];
};

// MAXN: node capacity of the list = LDS footprint (the list ends at N..N+3 nodes, N = the level's quota; the host
// picks the smallest instance that holds the largest quota: 70 KB of LDS at 1024 nodes allow 2 workgroups per CU,
// 22 KB at 320 allow 7)
// NT: threads of the workgroup.  The passes over the keys (quadrant counts, node_of updates, best response) and the rank-by-counting
// of the careful rounds are spread over the threads; a batch whose (level, image) workgroups do not fill the chip takes 1024.
template <int MAXN, int NT>
__device__ __forceinline__ void octree_level(const DevParams* __restrict__ Pp, const uint32_t* __restrict__ cellCand,
                                             const int* __restrict__ cellCount, uint32_t* __restrict__ candAll,
                                             unsigned short* __restrict__ nodeOfAll, int* __restrict__ candCount,
                                             uint32_t* __restrict__ kpSel, int* __restrict__ kpSelCount, int img0) {
  constexpr int OT_MAXN = MAXN;
  __shared__ OtNodes<MAXN> A, B;
  __shared__ int c4[OT_MAXN * 4];          // per node, keys per quadrant
  __shared__ int procIdx[OT_MAXN];         // processing order -> node position
  __shared__ int isSplit[OT_MAXN];         // 1 when the node is split this round (later: creation base)
  __shared__ int cbase[OT_MAXN];           // creation index of the node's first child
  __shared__ int unsplitRank[OT_MAXN];
  __shared__ int scanbuf[OT_MAXN];
  __shared__ unsigned long long best[OT_MAXN];
  __shared__ int scratch[16];
  __shared__ int s_n, s_S, s_T, s_mode, s_done, s_nExp;

  const DevParams& P = *Pp;
  const int lvl = blockIdx.x, img = blockIdx.y + img0;
  const LevelGeom& G = P.lv[lvl];
  const int tid = threadIdx.x;
  const int N = G.nfeatures;
  uint32_t* cand = candAll + (int64_t)img * P.candPerImage + G.candBase;
  unsigned short* nodeOf = nodeOfAll + (int64_t)img * P.candPerImage + G.candBase;
  uint32_t* sel = kpSel + (int64_t)img * P.kpSlotsPerImage + G.kpBase;
  const int lc = img * P.nlevels + lvl;

  // ---- gather the cells' keypoints in reference order (cells row-major, raster inside) ----
  const int ncells = G.nCols * G.nRows;
  const int* cc = cellCount + (int64_t)img * P.cellsPerImage + G.cellBase;
  const uint32_t* cin = cellCand + ((int64_t)img * P.cellsPerImage + G.cellBase) * CELL_CAP;
  int n = 0;
  for (int base = 0; base < ncells; base += OT_MAXN) {
    int m = min(OT_MAXN, ncells - base);
    for (int i = tid; i < m; i += NT) scanbuf[i] = cc[base + i];
    __syncthreads();
    int tot = block_excl_scan_1024<NT>(scanbuf, m, scratch);
    for (int i = tid; i < m; i += NT) {
      int cnt = cc[base + i];
      int o = n + scanbuf[i];
      const uint32_t* src = cin + (int64_t)(base + i) * CELL_CAP;
      for (int k = 0; k < cnt; ++k)
        if (o + k < G.candCap) cand[o + k] = src[k];
    }
    n += tot;
    __syncthreads();
  }
  if (n > G.candCap) n = G.candCap;
  if (tid == 0) candCount[lc] = n;
  if (n == 0 || G.nIni <= 0) {     // (a quota of 0 still runs the first expansion, like the reference loop)
    if (tid == 0) kpSelCount[lc] = 0;
    return;
  }
  __threadfence_block();
  __syncthreads();

  // ---- root nodes (ORBextractor.cc:540-589) ----
  const int nIni = G.nIni;
  const float hX = G.hX;
  for (int i = tid; i < OT_MAXN; i += NT) scanbuf[i] = 0;
  __syncthreads();
  for (int k = tid; k < n; k += NT) {
    int x = (cand[k] >> 8) & 0xFFF;
    int r = (int)__fdiv_rn((float)x, hX);
    if (r >= nIni) r = nIni - 1;
    nodeOf[k] = (unsigned short)r;       // temporarily the root index
    atomicAdd(&scanbuf[r], 1);
  }
  __syncthreads();
  // compact non-empty roots, keep order
  if (tid == 0) {
    int S = 0;
    for (int i = 0; i < nIni; ++i) {
      int cnt = scanbuf[i];
      if (cnt > 0) {
        A.x0[S] = (short)(int)__fmul_rn(hX, (float)i);
        A.x1[S] = (short)(int)__fmul_rn(hX, (float)(i + 1));
        A.y0[S] = 0;
        A.y1[S] = (short)(G.maxBY - G.minBY);
        A.cnt[S] = cnt;
        A.isNew[S] = 0;
        procIdx[i] = S;                  // root -> list position
        ++S;
      } else procIdx[i] = -1;
    }
    s_S = S; s_mode = 0; s_done = 0;
  }
  __syncthreads();
  for (int k = tid; k < n; k += NT) nodeOf[k] = (unsigned short)procIdx[nodeOf[k]];
  __syncthreads();

  OtNodes<MAXN>* cur = &A;
  OtNodes<MAXN>* nxt = &B;
  int guard = 0;
  while (!s_done && guard++ < 64) {
    const int S = s_S;
    const int mode = s_mode;
    // A. split candidates
    for (int p = tid; p < S; p += NT) {
      bool cd = cur->cnt[p] > 1 && (mode == 0 || cur->isNew[p]);
      isSplit[p] = cd ? 1 : 0;
      c4[4 * p] = c4[4 * p + 1] = c4[4 * p + 2] = c4[4 * p + 3] = 0;
    }
    __syncthreads();
    // B. keys per quadrant (ExtractorNode::DivideNode, :479-535)
    for (int k = tid; k < n; k += NT) {
      int p = nodeOf[k];
      if (!isSplit[p]) continue;
      uint32_t cv = cand[k];
      float fx = (float)((cv >> 8) & 0xFFF), fy = (float)(cv >> 20);
      int halfX = (int)ceilf(__fdiv_rn((float)(cur->x1[p] - cur->x0[p]), 2.f));
      int halfY = (int)ceilf(__fdiv_rn((float)(cur->y1[p] - cur->y0[p]), 2.f));
      int mx = cur->x0[p] + halfX, my = cur->y0[p] + halfY;
      int q = (fx < (float)mx) ? ((fy < (float)my) ? 0 : 2) : ((fy < (float)my) ? 1 : 3);
      atomicAdd(&c4[4 * p + q], 1);
    }
    __syncthreads();
    // C/D. processing order and the set of nodes actually split
    int ncand;
    if (mode == 0) {
      for (int p = tid; p < S; p += NT) scanbuf[p] = isSplit[p];
      __syncthreads();
      ncand = block_excl_scan_1024<NT>(scanbuf, S, scratch);
      for (int p = tid; p < S; p += NT)
        if (isSplit[p]) procIdx[scanbuf[p]] = p;
      if (tid == 0) s_T = ncand;
      __syncthreads();
    } else {
      // order by (size desc, list position asc): newest-created first among equals
      for (int p = tid; p < S; p += NT) scanbuf[p] = isSplit[p];
      __syncthreads();
      ncand = block_excl_scan_1024<NT>(scanbuf, S, scratch);
      // rank by counting
      for (int p = tid; p < S; p += NT) {
        if (!isSplit[p]) continue;
        int cp = cur->cnt[p], r = 0;
        for (int j = 0; j < S; ++j) {
          if (!isSplit[j]) continue;
          int cj = cur->cnt[j];
          r += (cj > cp || (cj == cp && j < p)) ? 1 : 0;
        }
        procIdx[r] = p;
      }
      __syncthreads();
      // size after each split in processing order; stop at the first >= N
      for (int r = tid; r < ncand; r += NT) {
        int p = procIdx[r];
        int ne = (c4[4 * p] > 0) + (c4[4 * p + 1] > 0) + (c4[4 * p + 2] > 0) + (c4[4 * p + 3] > 0);
        scanbuf[r] = ne - 1;
      }
      __syncthreads();
      block_excl_scan_1024<NT>(scanbuf, ncand, scratch);
      if (tid == 0) s_T = ncand;
      __syncthreads();
      for (int r = tid; r < ncand; r += NT) {
        int p = procIdx[r];
        int ne = (c4[4 * p] > 0) + (c4[4 * p + 1] > 0) + (c4[4 * p + 2] > 0) + (c4[4 * p + 3] > 0);
        int after = S + scanbuf[r] + ne - 1;       // list size once this node is split
        if (after >= N) atomicMin(&s_T, r + 1);
      }
      __syncthreads();
      const int T = s_T;
      for (int r = T + tid; r < ncand; r += NT) isSplit[procIdx[r]] = 0;
      __syncthreads();
    }
    const int T = s_T;
    // E. creation index of every split node's first child
    for (int r = tid; r < T; r += NT) {
      int p = procIdx[r];
      scanbuf[r] = (c4[4 * p] > 0) + (c4[4 * p + 1] > 0) + (c4[4 * p + 2] > 0) + (c4[4 * p + 3] > 0);
    }
    __syncthreads();
    const int C = block_excl_scan_1024<NT>(scanbuf, T, scratch);
    for (int r = tid; r < T; r += NT) cbase[procIdx[r]] = scanbuf[r];
    __syncthreads();
    // F. positions of the unsplit nodes
    for (int p = tid; p < S; p += NT) scanbuf[p] = isSplit[p] ? 0 : 1;
    __syncthreads();
    const int U = block_excl_scan_1024<NT>(scanbuf, S, scratch);
    for (int p = tid; p < S; p += NT) unsplitRank[p] = scanbuf[p];
    __syncthreads();
    const int S2 = C + U;
    // G. build the new list
    if (S2 <= OT_MAXN) {
      for (int p = tid; p < S; p += NT) {
        if (isSplit[p]) {
          int halfX = (int)ceilf(__fdiv_rn((float)(cur->x1[p] - cur->x0[p]), 2.f));
          int halfY = (int)ceilf(__fdiv_rn((float)(cur->y1[p] - cur->y0[p]), 2.f));
          int x0 = cur->x0[p], y0 = cur->y0[p], x1 = cur->x1[p], y1 = cur->y1[p];
          int mx = x0 + halfX, my = y0 + halfY;
          int c = cbase[p];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int cnt = c4[4 * p + q];
            if (cnt == 0) continue;
            int pos = C - 1 - c;
            ++c;
            nxt->x0[pos] = (short)((q & 1) ? mx : x0);
            nxt->x1[pos] = (short)((q & 1) ? x1 : mx);
            nxt->y0[pos] = (short)((q & 2) ? my : y0);
            nxt->y1[pos] = (short)((q & 2) ? y1 : my);
            nxt->cnt[pos] = cnt;
            nxt->isNew[pos] = 1;
          }
        } else {
          int pos = C + unsplitRank[p];
          nxt->x0[pos] = cur->x0[p]; nxt->x1[pos] = cur->x1[p];
          nxt->y0[pos] = cur->y0[p]; nxt->y1[pos] = cur->y1[p];
          nxt->cnt[pos] = cur->cnt[p];
          nxt->isNew[pos] = 0;
        }
      }
      for (int k = tid; k < n; k += NT) {
        int p = nodeOf[k];
        int np;
        if (isSplit[p]) {
          uint32_t cv = cand[k];
          float fx = (float)((cv >> 8) & 0xFFF), fy = (float)(cv >> 20);
          int halfX = (int)ceilf(__fdiv_rn((float)(cur->x1[p] - cur->x0[p]), 2.f));
          int halfY = (int)ceilf(__fdiv_rn((float)(cur->y1[p] - cur->y0[p]), 2.f));
          int mx = cur->x0[p] + halfX, my = cur->y0[p] + halfY;
          int q = (fx < (float)mx) ? ((fy < (float)my) ? 0 : 2) : ((fy < (float)my) ? 1 : 3);
          int rq = 0;
          for (int j = 0; j < q; ++j) rq += c4[4 * p + j] > 0;
          np = C - 1 - (cbase[p] + rq);
        } else {
          np = C + unsplitRank[p];
        }
        nodeOf[k] = (unsigned short)np;
      }
    }
    __syncthreads();
    // H. termination (ORBextractor.cc:665-738)
    if (tid == 0) s_nExp = 0;
    __syncthreads();
    if (S2 <= OT_MAXN) {
      int loc = 0;
      for (int p = tid; p < C; p += NT) loc += nxt->cnt[p] > 1;
      if (loc) atomicAdd(&s_nExp, loc);
    }
    __syncthreads();
    if (tid == 0) {
      if (S2 > OT_MAXN) { s_done = 1; }     // capacity guard (never reached for nfeatures/level <= 1000)
      else {
        s_S = S2;
        if (S2 >= N || S2 == S) s_done = 1;
        else if (mode == 0 && (S2 + s_nExp * 3) > N) s_mode = 1;
      }
    }
    __syncthreads();
    if (S2 <= OT_MAXN) { OtNodes<MAXN>* t = cur; cur = nxt; nxt = t; }
  }
  // ---- best response per node, first maximum wins (:741-757) ----
  const int S = s_S;
  for (int p = tid; p < S; p += NT) best[p] = 0ull;
  __syncthreads();
  for (int k = tid; k < n; k += NT) {
    uint32_t cv = cand[k];
    unsigned long long key = ((unsigned long long)(cv & 0xFF) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)k);
    atomicMax(&best[nodeOf[k]], key);
  }
  __syncthreads();
  for (int p = tid; p < S; p += NT) {
    unsigned k = 0xFFFFFFFFu - (unsigned)(best[p] & 0xFFFFFFFFull);
    if (p < G.kpCap) sel[p] = cand[k];
  }
  if (tid == 0) kpSelCount[lc] = S < G.kpCap ? S : G.kpCap;
}

#define OCTREE_KERNEL(NAME, MAXN, NT)                                                                                             \
  __global__ __launch_bounds__(NT) void NAME(const DevParams* __restrict__ Pp, const uint32_t* __restrict__ cellCand,             \
                                             const int* __restrict__ cellCount, uint32_t* __restrict__ candAll,                    \
                                             unsigned short* __restrict__ nodeOfAll, int* __restrict__ candCount,                  \
                                             uint32_t* __restrict__ kpSel, int* __restrict__ kpSelCount, int img0) {               \
    octree_level<MAXN, NT>(Pp, cellCand, cellCount, candAll, nodeOfAll, candCount, kpSel, kpSelCount, img0);                        \
  }
OCTREE_KERNEL(k_octree, 1024, 256)
OCTREE_KERNEL(k_octree_512, 512, 256)
OCTREE_KERNEL(k_octree_320, 320, 256)
OCTREE_KERNEL(k_octree_w, 1024, 1024)
OCTREE_KERNEL(k_octree_512_w, 512, 1024)
OCTREE_KERNEL(k_octree_320_w, 320, 1024)
#undef OCTREE_KERNEL

// ---------------------------------------------------------------------------
// k_blur: separable Gaussian, u8 -> u8, 8-bit fixed-point kernel (radius <= 3),
// REFLECT_101.  64x32 output tile per workgroup, staged through LDS.
// Generic over an "image set": nimg images `imgStride` apart, several planes
// (pyramid levels) described by a tile table.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur(const BlurJob* __restrict__ Jp, const uint8_t* __restrict__ in,
                                              int64_t inImgStride, uint8_t* __restrict__ out,
                                              int64_t outImgStride, int img0) {
  __shared__ uint32_t t8[38 * 18 + 2];  // (32 + 2R) rows of 72 bytes (+ the dword a zero-weight tap of the last row may touch)
  __shared__ int hs[38 * 64];
  const BlurJob& J = *Jp;
  const int img = blockIdx.y + img0;
  int pi = 0;
#pragma unroll 1
  for (int l = 1; l < J.nplanes; ++l)
    if ((int)blockIdx.x >= J.pl[l].tileBase) pi = l;
  const BlurPlane& PL = J.pl[pi];
  const int t = blockIdx.x - PL.tileBase;
  const int ty = t / PL.tilesX, tx = t - ty * PL.tilesX;
  const int x0 = tx * 64, y0 = ty * 32;
  const int R = J.radius;
  const uint8_t* src = in + (int64_t)img * inImgStride + PL.offIn;
  const int tid = threadIdx.x;
  const int th = 32 + 2 * R;
  int kc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) kc[k] = k <= 2 * R ? J.k[k] : 0;
  const bool narrow = (kc[0] | kc[1] | kc[2] | kc[3] | kc[4] | kc[5] | kc[6]) < 256;
  const uint32_t kLo = (uint32_t)kc[0] | ((uint32_t)kc[1] << 8) | ((uint32_t)kc[2] << 16) | ((uint32_t)kc[3] << 24);
  const uint32_t kHi = (uint32_t)kc[4] | ((uint32_t)kc[5] << 8) | ((uint32_t)kc[6] << 16);
  uint8_t* t8b = reinterpret_cast<uint8_t*>(t8);
  // stage rows y0-R .. y0+31+R, columns x0-4 .. x0+67 (18 dwords per row; byte b of a row = column x0 - 4 + b): a dword
  // that lies inside the image is one aligned 4-byte load (x0 is a multiple of 64, the rows are 64-byte aligned), a dword
  // that crosses the left or right border is put together from REFLECT_101 bytes
  for (int i = tid; i < th * 18; i += 256) {
    const int y = (int)(((float)i + 0.5f) * (1.0f / 18.0f)), d = i - y * 18;     // exact for i < 38 * 18
    const uint8_t* row = src + (int64_t)reflect101(y0 + y - R, PL.h) * PL.pitchIn;
    const int px = x0 - 4 + 4 * d;
    uint32_t v;
    if (px >= 0 && px + 3 < PL.w) {
      v = *reinterpret_cast<const uint32_t*>(row + px);
    } else {
      v = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // (columns further than R outside the image are never used by a stored output: clamped to keep the reflection single)
        const int xx = reflect101(min(max(px + k, -R), PL.w - 1 + R), PL.w);
        v |= (uint32_t)row[xx] << (8 * k);
      }
    }
    t8[i] = v;
  }
  __syncthreads();
  // horizontal pass: a thread makes 4 adjacent sums; output i of group j takes the bytes 4j + i + 4 - R + k, k = 0..2R
  {
    const int j = tid & 15, yy = tid >> 4;
    for (int y = yy; y < th; y += 16) {
      const uint32_t w0 = t8[y * 18 + j], w1 = t8[y * 18 + j + 1], w2 = t8[y * 18 + j + 2], w3 = t8[y * 18 + j + 3];
      int s[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = i + 4 - R;                       // 1..6
        const bool q = off >= 4;
        const uint32_t a0 = q ? w1 : w0, a1 = q ? w2 : w1, a2 = q ? w3 : w2;
        const uint32_t lo = __builtin_amdgcn_alignbyte(a1, a0, off & 3);
        const uint32_t hi = __builtin_amdgcn_alignbyte(a2, a1, off & 3);
        if (narrow) {
          // taps 0..3 on one unaligned dword, taps 4..6 on the next: two v_dot4_u32_u8 (coefficients 0 beyond 2R)
          s[i] = (int)__builtin_amdgcn_udot4(hi, kHi, __builtin_amdgcn_udot4(lo, kLo, 0u, false), false);
        } else {                                         // a coefficient of 256 (a sigma so small that the kernel is a delta)
          s[i] = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) s[i] += kc[k] * (int)((lo >> (8 * k)) & 255u);
#pragma unroll
          for (int k = 4; k < 7; ++k) s[i] += kc[k] * (int)((hi >> (8 * (k - 4))) & 255u);
        }
      }
      *reinterpret_cast<int4*>(&hs[y * 64 + 4 * j]) = make_int4(s[0], s[1], s[2], s[3]);
    }
  }
  __syncthreads();
  // vertical pass: 4 adjacent outputs per thread, one 4-byte store
  uint8_t* dst = out + (int64_t)img * outImgStride + PL.offOut;
  {
    const int j = tid & 15, yy = tid >> 4;
    for (int y = yy; y < 32; y += 16) {
      if (y0 + y >= PL.h || x0 + 4 * j >= PL.w) continue;
      int s[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        if (k > 2 * R) break;
        const int4 v = *reinterpret_cast<const int4*>(&hs[(y + k) * 64 + 4 * j]);
        s[0] += kc[k] * v.x; s[1] += kc[k] * v.y; s[2] += kc[k] * v.z; s[3] += kc[k] * v.w;
      }
      uint8_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int v = (s[i] + (1 << 15)) >> 16;
        o[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
      uint8_t* d = dst + (int64_t)(y0 + y) * PL.pitchOut + x0 + 4 * j;
      if (x0 + 4 * j + 3 < PL.w) *reinterpret_cast<uchar4*>(d) = make_uchar4(o[0], o[1], o[2], o[3]);
      else for (int i = 0; i < 4 && x0 + 4 * j + i < PL.w; ++i) d[i] = o[i];
    }
  }
}

// ---------------------------------------------------------------------------
// k_describe: one wave per selected keypoint slot: IC_Angle on the level image,
// steered rBRIEF on the blurred level (ballot -> 4 x u64), and the final
// level-major tables (ORBextractor.cc:1105-1147) straight into the frame record.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_describe(const DevParams* __restrict__ Pp, const uint8_t* __restrict__ pyr,
                                                 const uint8_t* __restrict__ blur, const uint32_t* __restrict__ kpSel,
                                                 const int* __restrict__ kpSelCount, uint8_t* __restrict__ table,
                                                 int64_t recordBytes, int64_t offCounts, int64_t offKp0,
                                                 int64_t offKp1, int64_t offDesc0, int64_t offDesc1, int img0) {
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0, slot = blockIdx.x;
  int lvl = 0;
#pragma unroll 1
  for (int l = 1; l < P.nlevels; ++l)
    if (slot >= P.lv[l].kpBase) lvl = l;
  const LevelGeom& G = P.lv[lvl];
  const int p = slot - G.kpBase;
  const int* cnts = kpSelCount + img * P.nlevels;
  if (p >= cnts[lvl]) return;
  int before = 0, total = 0;
  for (int l = 0; l < P.nlevels; ++l) {
    int c = cnts[l];
    if (l < lvl) before += c;
    total += c;
  }
  const int lane = threadIdx.x;
  uint8_t* rec = table + (int64_t)(img >> 1) * recordBytes;
  const int eye = img & 1;
  (void)offCounts; (void)total;   // counts are written by k_kp_counts
  const int outIdx = before + p;
  if (outIdx >= P.kpCap) return;
  const uint32_t cv = kpSel[(int64_t)img * P.kpSlotsPerImage + slot];
  const int kx = (int)((cv >> 8) & 0xFFF) + G.minBX, ky = (int)(cv >> 20) + G.minBY;
  const int resp = (int)(cv & 0xFF);
  const uint8_t* im = pyr + (int64_t)img * P.pyrBlock + G.offset;
  // IC_Angle: moments over the radius-15 disc.  The disc rows (umax) go through LDS so that the 16 pixel loads of a lane
  // do not each wait for a table load: they are issued together.
  __shared__ int s_umax[16];
  if (lane < 16) s_umax[lane] = P.umax[lane];
  __syncthreads();
  int m10 = 0, m01 = 0;
  int val[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = lane + 64 * k;
    const int r = (i * 2115) >> 16;                // i / 31, exact for i < 1024
    const int v = r - 15, u = (i - r * 31) - 15;
    const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
    val[k] = 0;
    if (i < 31 * 31 && au <= s_umax[av]) val[k] = im[(int64_t)(ky + v) * G.pitch + kx + u];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = lane + 64 * k;
    const int r = (i * 2115) >> 16;
    m10 += ((i - r * 31) - 15) * val[k];
    m01 += (r - 15) * val[k];
  }
  m10 = wave_sum_i32(m10);
  m01 = wave_sum_i32(m01);
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  // steered rBRIEF
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float ang = __fmul_rn(angle, factorPI);
  float a, b;                                          // a = cos, b = sin (ORBextractor.cc:111)
  sincos_of_float(ang, (P.parityFlags & PLI_PARITY_TRIG_F32_ORB) != 0, &b, &a);
  const uint8_t* bl = blur + (int64_t)img * P.pyrBlock + G.offset;
  const uint8_t* center = bl + (int64_t)ky * G.pitch + kx;
  unsigned long long words[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const signed char* pt = c_orb_pattern + 4 * (64 * j + lane);
    float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
    int r0 = cv_round_f(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
    int c0 = cv_round_f(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
    int r1 = cv_round_f(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
    int c1 = cv_round_f(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
    int t0 = center[(int64_t)r0 * G.pitch + c0];
    int t1 = center[(int64_t)r1 * G.pitch + c1];
    words[j] = __ballot(t0 < t1);
  }
  if (lane == 0) {
    pli_keypoint kp;
    kp.x = (float)kx; kp.y = (float)ky;
    if (lvl != 0) { kp.x = __fmul_rn((float)kx, G.scale); kp.y = __fmul_rn((float)ky, G.scale); }
    kp.size = (float)(int)__fmul_rn(31.f, G.scale);
    kp.angle = angle;
    kp.response = (float)resp;
    kp.octave = lvl;
    reinterpret_cast<pli_keypoint*>(rec + (eye ? offKp1 : offKp0))[outIdx] = kp;
    unsigned long long* d = reinterpret_cast<unsigned long long*>(rec + (eye ? offDesc1 : offDesc0) + (int64_t)outIdx * 32);
    d[0] = words[0]; d[1] = words[1]; d[2] = words[2]; d[3] = words[3];
  }
}

// N (mvKeys.size()) per eye into the frame record
// k_zero_ranges: up to 12 buffers (ZeroRanges) cleared by ONE launch (what the relaxation zeroes at the start of a call: control blocks, stamp
// planes, cell tables, counters — eight hipMemsetAsync launches before; a single pair's call spent 15 of its ~110 launches on fills).
// blockIdx.y = the range; 16-byte stores on the aligned body, words at both ends.
__global__ __launch_bounds__(256) void k_zero_ranges(ZeroRanges Z) {
  const int r = blockIdx.y;
  uint32_t* p = Z.p[r];
  const int64_t n = Z.words[r];
  if (!p || n <= 0) return;
  const int64_t head = min<int64_t>(n, (int64_t)((16 - ((uintptr_t)p & 15)) & 15) >> 2);
  const int64_t body = (n - head) >> 2;                          // uint4 stores
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  uint4* b = reinterpret_cast<uint4*>(p + head);
  for (int64_t i = tid; i < body; i += stride) b[i] = make_uint4(0u, 0u, 0u, 0u);
  if (tid < head) p[tid] = 0u;
  const int64_t tail0 = head + 4 * body;
  if (tid < n - tail0) p[tail0 + tid] = 0u;
}

__global__ void k_kp_counts(const DevParams* __restrict__ Pp, const int* __restrict__ kpSelCount,
                            uint8_t* __restrict__ table, int64_t recordBytes, int64_t offCounts, int nimg, int img0) {
  const DevParams& P = *Pp;
  int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= nimg) return;
  img += img0;
  int total = 0;
  for (int l = 0; l < P.nlevels; ++l) total += kpSelCount[img * P.nlevels + l];
  uint8_t* rec = table + (int64_t)(img >> 1) * recordBytes;
  rec[offCounts + 26 + (img & 1)] = total > P.kpCap ? 1 : 0;       // truncation flag (counts[6], bytes 2 and 3)
  if (total > P.kpCap) total = P.kpCap;
  reinterpret_cast<int*>(rec + offCounts)[img & 1] = total;
}

}  // namespace pli
