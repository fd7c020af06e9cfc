// Matching kernels for gfx950 (wave64): 256-bit Hamming by XOR + popcount on
// 4 x u64, wave-wide argmin by 64-bit key reduction.
//
//   k_stereo_points + k_stereo_median   Frame::ComputeStereoMatches        (Frame.cc:976-1154)
//   k_stereo_lines                      Frame::ComputeStereoMatches_Lines  (Frame.cc:1156-1307)
//                                       + matchGrid(lines) (LineMatcher.cpp:317-396)
//                                       + GridStructure/LineIterator (gridStructure.cpp, LineIterator.cpp)
//   k_distance                          ORBmatcher::DescriptorDistance     (ORBmatcher.cc:2495-2511)
//   k_knn2, k_ratio, k_mutual           matchNNR / match                   (LineMatcher.cpp:139-229)
//   k_search_by_projection              ORBmatcher::SearchByProjection(F,F)(ORBmatcher.cc:2179-2323)
//   k_search_local_map                  ORBmatcher::SearchByProjection(F,MPs)(ORBmatcher.cc:44-143)
#include "kernels.hpp"
#include "device_prims.hpp"
#include <climits>

namespace pli {

__device__ __forceinline__ void load_desc(const uint8_t* p, uint64_t d[4]) {
  const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
  d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3];
}

// ---------------------------------------------------------------------------
// Stereo points: one wave per left keypoint.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_stereo_points(const DevParams* __restrict__ Pp, const uint8_t* __restrict__ pyr,
                                                      uint8_t* __restrict__ table, int64_t recordBytes,
                                                      int64_t offCounts, int64_t offKp0, int64_t offKp1,
                                                      int64_t offDesc0, int64_t offDesc1, int64_t offUr,
                                                      int64_t offDepth, int* __restrict__ sadOut,
                                                      int* __restrict__ bestIdxOut) {
  const DevParams& P = *Pp;
  const int frame = blockIdx.y, iL = blockIdx.x, lane = threadIdx.x;
  uint8_t* rec = table + (int64_t)frame * recordBytes;
  const int* counts = reinterpret_cast<const int*>(rec + offCounts);
  const int N = counts[0], Nr = counts[1];
  if (iL >= N) return;
  const pli_keypoint* kpsL = reinterpret_cast<const pli_keypoint*>(rec + offKp0);
  const pli_keypoint* kpsR = reinterpret_cast<const pli_keypoint*>(rec + offKp1);
  const uint8_t* descL = rec + offDesc0;
  const uint8_t* descR = rec + offDesc1;
  float* uright = reinterpret_cast<float*>(rec + offUr);
  float* depth = reinterpret_cast<float*>(rec + offDepth);
  const pli_keypoint kpL = kpsL[iL];
  const int levelL = kpL.octave;
  const float vL = kpL.y, uL = kpL.x;
  const int rowL = (int)vL;
  const float maxD = P.maxD, minD = 0.f;
  const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
  uint64_t dL[4];
  load_desc(descL + (int64_t)iL * 32, dL);
  unsigned long long bestKey = ~0ull;
  // (row band half-height per level through LDS: the scan below then has no load that waits for another)
  __shared__ float s_r[MAX_LEVELS];
  if (lane < MAX_LEVELS) s_r[lane] = lane < P.nlevels ? __fmul_rn(2.0f, P.lv[lane].scale) : 0.f;
  __syncthreads();
  if (!(maxU < 0)) {
#pragma unroll 4
    for (int iR = lane; iR < Nr; iR += 64) {
      const pli_keypoint kpR = kpsR[iR];
      const float r = s_r[kpR.octave];
      const int maxr = (int)ceilf(__fadd_rn(kpR.y, r));
      const int minr = (int)floorf(__fsub_rn(kpR.y, r));
      if (rowL < minr || rowL > maxr) continue;
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        uint64_t dR[4];
        load_desc(descR + (int64_t)iR * 32, dR);
        const int dist = hamming256(dL, dR);
        if (dist < 100) {     // ORBmatcher::TH_HIGH, strict
          unsigned long long key = ((unsigned long long)dist << 32) | (unsigned)iR;
          bestKey = key < bestKey ? key : bestKey;
        }
      }
    }
  }
  bestKey = wave_min_u64(bestKey);
  float outU = -1.f, outD = -1.f;
  int outSad = -1, outIdx = -1;
  const int thOrbDist = (100 + 50) / 2;
  if (bestKey != ~0ull && (int)(bestKey >> 32) < thOrbDist) {
    const int bestIdxR = (int)(bestKey & 0xFFFFFFFFu);
    outIdx = bestIdxR;
    const float uR0 = kpsR[bestIdxR].x;
    const LevelGeom& G = P.lv[levelL];
    const float scaleFactor = G.invScale;
    const float scaleduL = roundf(__fmul_rn(kpL.x, scaleFactor));
    const float scaledvL = roundf(__fmul_rn(kpL.y, scaleFactor));
    const float scaleduR0 = roundf(__fmul_rn(uR0, scaleFactor));
    const int w = 5, L = 5;
    const int cy = (int)scaledvL, cxl = (int)scaleduL, cxr = (int)scaleduR0;
    bool ok = !(cy - w < 0 || cy + w + 1 > G.h || cxl - w < 0 || cxl + w + 1 > G.w);
    const float iniu = __fadd_rn(scaleduR0, (float)(L - w));
    const float endu = __fadd_rn(scaleduR0, (float)(L + w + 1));
    if (iniu < 0 || endu >= (float)G.w) ok = false;
    if (cxr - L - w < 0) ok = false;
    if (ok) {
      const uint8_t* imL = pyr + (int64_t)(frame * 2) * P.pyrBlock + G.offset;
      const uint8_t* imR = pyr + (int64_t)(frame * 2 + 1) * P.pyrBlock + G.offset;
      const int cL = imL[(int64_t)cy * G.pitch + cxl];
      // each lane owns up to two pixels of the 11x11 window
      int aL[2], px[2], py[2];
      bool act[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        int i = lane + 64 * k;
        act[k] = i < 121;
        int dy = act[k] ? i / 11 - w : 0, dx = act[k] ? i % 11 - w : 0;
        px[k] = dx; py[k] = dy;
        aL[k] = act[k] ? (int)imL[(int64_t)(cy + dy) * G.pitch + cxl + dx] - cL : 0;
      }
      float vDists[11];
      int bestDist = INT_MAX, bestincR = 0;
#pragma unroll
      for (int incR = -L; incR <= L; ++incR) {
        const int cR = imR[(int64_t)cy * G.pitch + cxr + incR];
        int s = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (act[k]) {
            int b = (int)imR[(int64_t)(cy + py[k]) * G.pitch + cxr + incR + px[k]] - cR;
            s += abs(aL[k] - b);
          }
        s = wave_sum_i32(s);
        const float dist = (float)s;
        if (dist < (float)bestDist) { bestDist = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (!(bestincR == -L || bestincR == L)) {
        float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
        for (int k = 1; k < 10; ++k)
          if (k == L + bestincR) { dist1 = vDists[k - 1]; dist2 = vDists[k]; dist3 = vDists[k + 1]; }
        const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                       __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
        if (!(deltaR < -1 || deltaR > 1)) {
          float bestuR = __fmul_rn(G.scale, __fadd_rn(__fadd_rn(scaleduR0, (float)bestincR), deltaR));
          float disparity = __fsub_rn(uL, bestuR);
          if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
              disparity = 0.01f;
              bestuR = (float)((double)uL - 0.01);
            }
            outD = __fdiv_rn(P.bf, disparity);
            outU = bestuR;
            outSad = bestDist;
          }
        }
      }
    }
  }
  if (lane == 0) {
    uright[iL] = outU;
    depth[iL] = outD;
    sadOut[(int64_t)frame * P.kpCap + iL] = outSad;
    if (bestIdxOut) bestIdxOut[(int64_t)frame * P.kpCap + iL] = outIdx;
  }
}

// median-based outlier cut (Frame.cc:1140-1153), one workgroup per frame
__global__ __launch_bounds__(256) void k_stereo_median(const DevParams* __restrict__ Pp, uint8_t* __restrict__ table,
                                                       int64_t recordBytes, int64_t offCounts, int64_t offUr,
                                                       int64_t offDepth, const int* __restrict__ sadIn) {
  extern __shared__ int s_sad[];
  __shared__ int s_m, s_median, s_left;
  const DevParams& P = *Pp;
  const int frame = blockIdx.x, tid = threadIdx.x;
  uint8_t* rec = table + (int64_t)frame * recordBytes;
  int* counts = reinterpret_cast<int*>(rec + offCounts);
  const int N = counts[0];
  float* uright = reinterpret_cast<float*>(rec + offUr);
  float* depth = reinterpret_cast<float*>(rec + offDepth);
  const int* sad = sadIn + (int64_t)frame * P.kpCap;
  if (tid == 0) { s_m = 0; s_median = -1; s_left = 0; }
  __syncthreads();
  int loc = 0;
  for (int i = tid; i < N; i += 256) {
    int v = sad[i];
    s_sad[i] = v;
    loc += v >= 0;
  }
  if (loc) atomicAdd(&s_m, loc);
  __syncthreads();
  const int M = s_m;
  if (M == 0) {
    if (tid == 0) counts[4] = 0;
    return;
  }
  // the median = the (M / 2)-th smallest valid distance (Frame.cc:1143-1146 sorts and takes element size / 2): radix select, four
  // passes of 8 bits from the top — a 256-bin LDS histogram of the values that match the bits chosen so far, and one wave that finds
  // the bin holding the k-th.  (The earlier form counted, for every value, the smaller ones with one dependent LDS read per
  // comparison: 93 us for the 1200 keypoints of a single pair.)
  __shared__ int s_hist[256];
  __shared__ unsigned s_pref, s_mask;
  __shared__ int s_k;
  if (tid == 0) { s_pref = 0u; s_mask = 0u; s_k = M / 2; }
  for (int shift = 24; shift >= 0; shift -= 8) {
    s_hist[tid] = 0;
    __syncthreads();
    const unsigned pref = s_pref, mask = s_mask;
    for (int i = tid; i < N; i += 256) {
      const int v = s_sad[i];
      if (v >= 0 && ((unsigned)v & mask) == pref) atomicAdd(&s_hist[((unsigned)v >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid < 64) {
      const int h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
      const int sum = h0 + h1 + h2 + h3;
      int inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t2 = __shfl_up(inc, o, 64);
        if (tid >= o) inc += t2;
      }
      const int k = s_k, before = inc - sum;
      if (before <= k && k < inc) {                 // exactly one lane: the k-th lies in its four bins
        int b = 0, c = before;
        if (k >= c + h0) { c += h0; b = 1; if (k >= c + h1) { c += h1; b = 2; if (k >= c + h2) { c += h2; b = 3; } } }
        s_pref = pref | ((unsigned)(4 * tid + b) << shift);
        s_mask = mask | (255u << shift);
        s_k = k - c;
      }
    }
    __syncthreads();
  }
  if (tid == 0) s_median = (int)s_pref;
  __syncthreads();
  const float median = (float)s_median;
  const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), median);
  loc = 0;
  for (int i = tid; i < N; i += 256) {
    const int v = s_sad[i];
    if (v < 0) continue;
    if (!((float)v < thDist)) {
      uright[i] = -1.f;
      depth[i] = -1.f;
    } else ++loc;
  }
  if (loc) atomicAdd(&s_left, loc);
  __syncthreads();
  if (tid == 0) counts[4] = s_left;
}

// ---------------------------------------------------------------------------
// Stereo lines: one workgroup per frame.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void normalize2(double& a, double& b) {
  const double m = sqrt(a * a + b * b);
  a /= m;
  b /= m;
}

__device__ double line_overlap_stereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj, double horizTh) {
  double overlap = 1.f;
  if (fabs(epl_obs - spl_obs) > horizTh) {
    double sln = fmin(spl_obs, epl_obs);
    double eln = fmax(spl_obs, epl_obs);
    double spn = fmin(spl_proj, epl_proj);
    double epn = fmax(spl_proj, epl_proj);
    double length = eln - spn;
    if ((epn < sln) || (spn > eln)) overlap = 0.f;
    else {
      if ((epn > eln) && (spn < sln)) overlap = eln - sln;
      else overlap = fmin(eln, epn) - fmax(sln, spn);
    }
    if (length > 0.01f) overlap = overlap / length;
    else overlap = 0.f;
    if (overlap > 1.f) overlap = 1.f;
  }
  return overlap;
}

__global__ __launch_bounds__(256) void k_stereo_lines(const DevParams* __restrict__ Pp, uint8_t* __restrict__ table,
                                                      int64_t recordBytes, int64_t offCounts, int64_t offKl0,
                                                      int64_t offKl1, int64_t offLd0, int64_t offLd1, int64_t offDisp,
                                                      int64_t offLe, unsigned long long* __restrict__ maskAll,
                                                      double* __restrict__ dirAll, short* __restrict__ dmatAll,
                                                      int* __restrict__ m12All, int* __restrict__ m21All, int phase) {
  // phase 0: the whole matcher, one workgroup per frame.  With many lines (4K: 500 x 500 pairs per frame) the pair distances are
  // the bulk and one workgroup per frame leaves the chip idle: the host then launches phase 1 (tables of the right lines),
  // phase 2 (the pair distances, gridDim.y workgroups per frame) and phase 3 (the selection) — the phases already talk through
  // global memory.
  const DevParams& P = *Pp;
  const int frame = blockIdx.x, tid = threadIdx.x;
  uint8_t* rec = table + (int64_t)frame * recordBytes;
  int* counts = reinterpret_cast<int*>(rec + offCounts);
  const int n1 = counts[2], n2 = counts[3];
  const pli_keyline* KL = reinterpret_cast<const pli_keyline*>(rec + offKl0);
  const pli_keyline* KR = reinterpret_cast<const pli_keyline*>(rec + offKl1);
  const uint8_t* descL = rec + offLd0;
  const uint8_t* descR = rec + offLd1;
  float* disp = reinterpret_cast<float*>(rec + offDisp);
  double* le = reinterpret_cast<double*>(rec + offLe);
  const int cap = P.klCap;
  unsigned long long* mask = maskAll + (int64_t)frame * cap * GRID_ROWS;
  double* dir = dirAll + (int64_t)frame * cap * 2;
  short* dmat = dmatAll + (int64_t)frame * cap * cap;
  int* m12 = m12All + (int64_t)frame * cap;
  int* m21 = m21All + (int64_t)frame * cap;
  if (phase <= 1)
  for (int i = tid; i < n1; i += 256) {
    disp[2 * i] = -1.f; disp[2 * i + 1] = -1.f;
    le[3 * i] = 0.0; le[3 * i + 1] = 0.0; le[3 * i + 2] = 0.0;
    m12[i] = -1;
  }
  if (n1 == 0 || n2 == 0) {
    if (tid == 0 && phase != 2) counts[5] = 0;
    return;
  }
  const double inv_width = (double)GRID_COLS / (double)P.W;
  const double inv_height = (double)GRID_ROWS / (double)P.H;
  // right lines: direction + Bresenham cell masks (getLineCoords / LineIterator)
  if (phase <= 1)
  for (int i2 = tid; i2 < n2; i2 += 256) {
    const pli_keyline kl = KR[i2];
    double vx = (double)__fsub_rn(kl.endPointX, kl.startPointX) * inv_width;
    double vy = (double)__fsub_rn(kl.endPointY, kl.startPointY) * inv_height;
    normalize2(vx, vy);
    dir[2 * i2] = vx; dir[2 * i2 + 1] = vy;
    unsigned long long* mk = mask + (int64_t)i2 * GRID_ROWS;
    for (int r = 0; r < GRID_ROWS; ++r) mk[r] = 0ull;
    double x1 = (double)kl.startPointX * inv_width, y1 = (double)kl.startPointY * inv_height;
    double x2 = (double)kl.endPointX * inv_width, y2 = (double)kl.endPointY * inv_height;
    const bool steep = fabs(y2 - y1) > fabs(x2 - x1);
    if (steep) { double t = x1; x1 = y1; y1 = t; t = x2; x2 = y2; y2 = t; }
    if (x1 > x2) { double t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    const double dx = x2 - x1, dy = fabs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int x = (int)x1, y = (int)y1;
    const int maxX = (int)x2;
    while (x <= maxX) {
      const int cx = steep ? y : x, cy = steep ? x : y;
      if (cx >= 0 && cx < GRID_COLS && cy >= 0 && cy < GRID_ROWS) mk[cy] |= 1ull << cx;
      error -= dy;
      if (error < 0) { y += ystep; error += dx; }
      x++;
    }
  }
  if (phase == 1) return;
  __threadfence_block();
  __syncthreads();
  // pair distances for candidate pairs passing the direction gate, else -1
  const int ws = P.sWs;
  if (phase == 0 || phase == 2)
  for (int pidx = tid + 256 * (int)blockIdx.y; pidx < n1 * n2; pidx += 256 * (int)gridDim.y) {
    const int i1 = pidx / n2, i2 = pidx - i1 * n2;
    const pli_keyline kl = KL[i1];
    const int sx = (int)((double)kl.startPointX * inv_width), sy = (int)((double)kl.startPointY * inv_height);
    const int ex = (int)((double)kl.endPointX * inv_width), ey = (int)((double)kl.endPointY * inv_height);
    const unsigned long long* mk = mask + (int64_t)i2 * GRID_ROWS;
    bool cand = false;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int qx = e ? ex : sx, qy = e ? ey : sy;
      const int min_x = max(0, qx - ws), max_x = min(GRID_COLS, qx + 0 + 1);
      if (qy >= 0 && qy < GRID_ROWS && max_x > min_x) {
        const unsigned long long hi = max_x >= 64 ? ~0ull : ((1ull << max_x) - 1ull);
        const unsigned long long wm = hi & ~((1ull << min_x) - 1ull);
        cand = cand || ((mk[qy] & wm) != 0ull);
      }
    }
    short d = -1;
    if (cand) {
      double vx = (double)(ex - sx), vy = (double)(ey - sy);
      normalize2(vx, vy);
      const double dt = vx * dir[2 * i2] + vy * dir[2 * i2 + 1];
      if (!(fabs(dt) < P.lineSimTh)) {
        uint64_t a[4], b[4];
        load_desc(descL + (int64_t)i1 * 32, a);
        load_desc(descR + (int64_t)i2 * 32, b);
        d = (short)hamming256(a, b);
      }
    }
    dmat[(int64_t)i1 * cap + i2] = d;
  }
  if (phase == 2) return;
  __threadfence_block();
  __syncthreads();
  // bestLRMatches: a pair only counts if it lowers the running minimum of its column
  if (P.bestLR) {
    for (int i2 = tid; i2 < n2; i2 += 256) {
      int run = INT_MAX, who = -1;
      for (int i1 = 0; i1 < n1; ++i1) {
        const int d = dmat[(int64_t)i1 * cap + i2];
        if (d < 0) continue;
        if (d < run) { run = d; who = i1; }
        else dmat[(int64_t)i1 * cap + i2] = -1;
      }
      m21[i2] = who;
    }
    __threadfence_block();
    __syncthreads();
  }
  for (int i1 = tid; i1 < n1; i1 += 256) {
    int best_d = INT_MAX, best_d2 = INT_MAX, best_idx = -1;
    for (int i2 = 0; i2 < n2; ++i2) {
      const int d = dmat[(int64_t)i1 * cap + i2];
      if (d < 0) continue;
      if (d < best_d) { best_d2 = best_d; best_d = d; best_idx = i2; }
      else if (d < best_d2) best_d2 = d;
    }
    int m = -1;
    if ((double)best_d < (double)best_d2 * P.ratio12L) m = best_idx;
    if (P.bestLR && m >= 0 && m21[m] != i1) m = -1;
    m12[i1] = m;
  }
  __threadfence_block();
  __syncthreads();
  int loc = 0;
  for (int i1 = tid; i1 < n1; i1 += 256) {
    const int i2 = m12[i1];
    if (i2 < 0) continue;
    const pli_keyline a = KL[i1], b = KR[i2];
    const double spl0 = a.startPointX, spl1 = a.startPointY, epl0 = a.endPointX, epl1 = a.endPointY;
    double l0 = spl1 * 1.0 - 1.0 * epl1, l1 = 1.0 * epl0 - spl0 * 1.0, l2 = spl0 * epl1 - spl1 * epl0;
    const double nrm = sqrt(l0 * l0 + l1 * l1);
    l0 = l0 / nrm; l1 = l1 / nrm; l2 = l2 / nrm;
    double spr0 = b.startPointX, spr1 = b.startPointY, epr0 = b.endPointX, epr1 = b.endPointY;
    const double overlap = line_overlap_stereo(spl1, epl1, spr1, epr1, P.horizTh);
    const double nsx = (spr0 * (spl1 - epr1) + epr0 * (spr1 - spl1)) / (spr1 - epr1);
    spr0 = nsx; spr1 = spl1;
    const double nex = (spr0 * (epl1 - epr1) + epr0 * (spr1 - epl1)) / (spr1 - epr1);
    epr0 = nex; epr1 = epl1;
    double disp_s = spl0 - spr0, disp_e = epl0 - epr0;
    if (fmin(disp_s, disp_e) / fmax(disp_s, disp_e) < P.minDispRatio) { disp_s = -1.0; disp_e = -1.0; }
    if (disp_s >= P.minDisp && disp_e >= P.minDisp && fabs(spl1 - epl1) > P.horizTh && fabs(spr1 - epr1) > P.horizTh &&
        overlap > P.overlapTh) {
      disp[2 * i1] = (float)disp_s;
      disp[2 * i1 + 1] = (float)disp_e;
      le[3 * i1] = l0; le[3 * i1 + 1] = l1; le[3 * i1 + 2] = l2;
      ++loc;
    }
  }
  __shared__ int s_cnt;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  if (loc) atomicAdd(&s_cnt, loc);
  __syncthreads();
  if (tid == 0) counts[5] = s_cnt;
}

// ---------------------------------------------------------------------------
// Stateless matchers on caller tables
// ---------------------------------------------------------------------------
__global__ void k_distance(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int n, int* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[4], y[4];
  load_desc(a + (int64_t)i * 32, x);
  load_desc(b + (int64_t)i * 32, y);
  out[i] = hamming256(x, y);
}

// knnMatch(k=2): one wave per query; ties -> lower train index
__global__ __launch_bounds__(64) void k_knn2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                             int* __restrict__ idx, int* __restrict__ dist) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= nq) return;
  uint64_t dq[4];
  load_desc(q + (int64_t)i * 32, dq);
  unsigned long long k1 = ~0ull, k2 = ~0ull;
  for (int j = lane; j < nt; j += 64) {
    uint64_t dt[4];
    load_desc(t + (int64_t)j * 32, dt);
    unsigned long long key = ((unsigned long long)hamming256(dq, dt) << 32) | (unsigned)j;
    if (key < k1) { k2 = k1; k1 = key; }
    else if (key < k2) k2 = key;
  }
  const unsigned long long m1 = wave_min_u64(k1);
  const unsigned long long c2 = (k1 == m1) ? k2 : k1;
  const unsigned long long m2 = wave_min_u64(c2);
  if (lane == 0) {
    idx[2 * i] = m1 == ~0ull ? -1 : (int)(m1 & 0xFFFFFFFFu);
    dist[2 * i] = m1 == ~0ull ? INT_MAX : (int)(m1 >> 32);
    idx[2 * i + 1] = m2 == ~0ull ? -1 : (int)(m2 & 0xFFFFFFFFu);
    dist[2 * i + 1] = m2 == ~0ull ? INT_MAX : (int)(m2 >> 32);
  }
}

// matchNNR ratio test on knn2 output
__global__ void k_ratio(const int* __restrict__ idx, const int* __restrict__ dist, int n, int nTrain, float nnr,
                        int* __restrict__ m) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r = -1;
  if (nTrain >= 2 && (float)dist[2 * i] < __fmul_rn((float)dist[2 * i + 1], nnr)) r = idx[2 * i];
  m[i] = r;
}

__global__ void k_mutual(int* __restrict__ m12, const int* __restrict__ m21, int n1, int* __restrict__ nmatches) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int ok = 0;
  if (i < n1) {
    int i2 = m12[i];
    if (i2 >= 0 && m21 && m21[i2] != i) { m12[i] = -1; i2 = -1; }
    ok = i2 >= 0;
  }
  unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(nmatches, __popcll(b));
}

// ---------------------------------------------------------------------------
// SearchByProjection(CurrentFrame, LastFrame): the queries are consumed in
// order by one wave because a current keypoint taken by an earlier query is
// not available to later ones; the 64 lanes scan the current keypoints.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_search_by_projection(const pli_proj_query* __restrict__ q,
                                                             const uint8_t* __restrict__ qdesc, int nq,
                                                             const pli_keypoint* __restrict__ kp,
                                                             const uint8_t* __restrict__ desc,
                                                             const float* __restrict__ uright, int ncur, float minX,
                                                             float maxX, float minY, float maxY, int checkOri,
                                                             int* __restrict__ owner /* ncur */,
                                                             int* __restrict__ bestIdx2 /* nq */,
                                                             int* __restrict__ nmatchesOut,
                                                             const uint8_t* __restrict__ occupied /* ncur or null */,
                                                             int* __restrict__ rawIdx2 /* nq or null: the matches before the rotation filter */) {
  __shared__ int hist[30];
  __shared__ int keep[30];
  const int lane = threadIdx.x;
  const float gwInv = __fdiv_rn((float)GRID_COLS, __fsub_rn(maxX, minX));
  const float ghInv = __fdiv_rn((float)GRID_ROWS, __fsub_rn(maxY, minY));
  // (a keypoint that holds a map point with observations before the call is not available, ORBmatcher.cc:2259-2261; a match made in
  // this call takes its keypoint away from the later queries only if ITS map point has observations — PLI_PROJ_NO_OBSERVATIONS, bit 1 of
  // `valid`, marks the queries whose map point has none: Tracking::UpdateLastFrame's temporal points in localisation mode)
  for (int i = lane; i < ncur; i += 64) owner[i] = (occupied && occupied[i]) ? INT_MAX : -1;
  for (int i = lane; i < nq; i += 64) bestIdx2[i] = -1;
  if (lane < 30) hist[lane] = 0;
  __syncthreads();
  int nmatches = 0;
  for (int i = 0; i < nq; ++i) {
    const pli_proj_query Q = q[i];
    if (!Q.valid) continue;
    const float u = Q.u, v = Q.v, radius = Q.radius;
    if (u < minX || u > maxX) continue;
    if (v < minY || v > maxY) continue;
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(u, minX), radius), gwInv)));
    if (nMinCellX >= GRID_COLS) continue;
    const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(u, minX), radius), gwInv)));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(v, minY), radius), ghInv)));
    if (nMinCellY >= GRID_ROWS) continue;
    const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(v, minY), radius), ghInv)));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (Q.min_level > 0) || (Q.max_level >= 0);
    uint64_t dq[4];
    load_desc(qdesc + (int64_t)i * 32, dq);
    unsigned long long best = ~0ull;
    for (int i2 = lane; i2 < ncur; i2 += 64) {
      const pli_keypoint k = kp[i2];
      const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, minX), gwInv));
      const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, minY), ghInv));
      if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;   // PosInGrid
      if (px < nMinCellX || px > nMaxCellX || py < nMinCellY || py > nMaxCellY) continue;
      if (bCheckLevels) {
        if (k.octave < Q.min_level) continue;
        if (Q.max_level >= 0 && k.octave > Q.max_level) continue;
      }
      const float distx = __fsub_rn(k.x, u), disty = __fsub_rn(k.y, v);
      if (!(fabsf(distx) < radius && fabsf(disty) < radius)) continue;
      if (owner[i2] >= 0) continue;
      const float ur2 = uright[i2];
      if (ur2 > 0) {
        const float er = fabsf(__fsub_rn(Q.ur, ur2));
        if (er > radius) continue;
      }
      uint64_t d2[4];
      load_desc(desc + (int64_t)i2 * 32, d2);
      const int dist = hamming256(dq, d2);
      if (dist < 256) {
        // first minimum in the reference's visiting order: cells ix asc, iy asc, then insertion order
        unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)px << 34) |
                                 ((unsigned long long)py << 28) | (unsigned long long)i2;
        best = key < best ? key : best;
      }
    }
    best = wave_min_u64(best);
    if (best != ~0ull && (int)(best >> 40) <= 100) {
      const int b2 = (int)(best & 0xFFFFFFFull);
      if (lane == 0) {
        if (!(Q.valid & 2)) owner[b2] = i;
        bestIdx2[i] = b2;
        if (checkOri) {
          float rot = __fsub_rn(Q.angle, kp[b2].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
          if (bin == 30) bin = 0;
          if (bin >= 0 && bin < 30) hist[bin]++;
        }
      }
      ++nmatches;
      __threadfence_block();
    }
    __syncthreads();
  }
  if (checkOri) {
    if (lane == 0) {
      int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
      for (int i = 0; i < 30; i++) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
      }
      if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
      else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
      for (int i = 0; i < 30; ++i) keep[i] = (i == ind1 || i == ind2 || i == ind3);
    }
    __syncthreads();
  }
  // the rotation filter, per accepted query (:2303-2320: every entry of a rejected bin takes one off nmatches)
  __syncthreads();
  int removed = 0;
  for (int i = lane; i < nq; i += 64) {
    const int b = bestIdx2[i];
    if (rawIdx2) rawIdx2[i] = b;
    if (b < 0 || !checkOri) continue;
    float rot = __fsub_rn(q[i].angle, kp[b].angle);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
    if (bin == 30) bin = 0;
    if (!(bin >= 0 && bin < 30 && keep[bin])) { bestIdx2[i] = -1; ++removed; }
  }
  removed = wave_sum_i32(removed);
  nmatches -= removed;
  if (lane == 0) *nmatchesOut = nmatches;
}

// ---------------------------------------------------------------------------
// Local-map search (ORBmatcher.cc:44-143, rectified stereo branch).  One wave walks the
// map points in order (the assignment of a keypoint is visible to the following points);
// the lanes share the scan over the frame's keypoints.  The reference's running
// best / second-best pair is the two smallest (distance, visiting order) keys.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_search_local_map(const pli_proj_query* __restrict__ q,
                                                         const uint8_t* __restrict__ qdesc, int nq,
                                                         const pli_keypoint* __restrict__ kp,
                                                         const uint8_t* __restrict__ desc,
                                                         const float* __restrict__ uright,
                                                         const uint8_t* __restrict__ occupied, int ncur, float minX,
                                                         float maxX, float minY, float maxY, float nnratio,
                                                         int* __restrict__ owner /* ncur */,
                                                         int* __restrict__ bestIdx2 /* nq */,
                                                         int* __restrict__ nmatchesOut) {
  const int lane = threadIdx.x;
  const float gwInv = __fdiv_rn((float)GRID_COLS, __fsub_rn(maxX, minX));
  const float ghInv = __fdiv_rn((float)GRID_ROWS, __fsub_rn(maxY, minY));
  for (int i = lane; i < ncur; i += 64) owner[i] = (occupied && occupied[i]) ? INT_MAX : -1;
  for (int i = lane; i < nq; i += 64) bestIdx2[i] = -1;
  __syncthreads();
  int nmatches = 0;
  for (int i = 0; i < nq; ++i) {
    const pli_proj_query Q = q[i];
    if (!Q.valid) continue;
    const float u = Q.u, v = Q.v, radius = Q.radius;
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(u, minX), radius), gwInv)));
    if (nMinCellX >= GRID_COLS) continue;
    const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(u, minX), radius), gwInv)));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(v, minY), radius), ghInv)));
    if (nMinCellY >= GRID_ROWS) continue;
    const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(v, minY), radius), ghInv)));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (Q.min_level > 0) || (Q.max_level >= 0);
    uint64_t dq[4];
    load_desc(qdesc + (int64_t)i * 32, dq);
    unsigned long long k1 = ~0ull, k2 = ~0ull;
    for (int i2 = lane; i2 < ncur; i2 += 64) {
      const pli_keypoint k = kp[i2];
      const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, minX), gwInv));
      const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, minY), ghInv));
      if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;   // PosInGrid
      if (px < nMinCellX || px > nMaxCellX || py < nMinCellY || py > nMaxCellY) continue;
      if (bCheckLevels) {
        if (k.octave < Q.min_level) continue;
        if (Q.max_level >= 0 && k.octave > Q.max_level) continue;
      }
      const float distx = __fsub_rn(k.x, u), disty = __fsub_rn(k.y, v);
      if (!(fabsf(distx) < radius && fabsf(disty) < radius)) continue;
      if (owner[i2] >= 0) continue;
      const float ur2 = uright[i2];
      if (ur2 > 0) {
        const float er = fabsf(__fsub_rn(Q.ur, ur2));
        if (er > radius) continue;
      }
      uint64_t d2[4];
      load_desc(desc + (int64_t)i2 * 32, d2);
      const int dist = hamming256(dq, d2);
      if (dist < 256) {
        const unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)px << 34) |
                                       ((unsigned long long)py << 28) | (unsigned long long)i2;
        if (key < k1) { k2 = k1; k1 = key; }
        else if (key < k2) k2 = key;
      }
    }
    const unsigned long long m1 = wave_min_u64(k1);
    const unsigned long long c2 = (k1 == m1) ? k2 : k1;
    const unsigned long long m2 = wave_min_u64(c2);
    if (m1 != ~0ull && (int)(m1 >> 40) <= 100) {
      const int b1 = (int)(m1 & 0xFFFFFFFull);
      const int bestDist = (int)(m1 >> 40);
      const int bestLevel = kp[b1].octave;
      int bestDist2 = 256, bestLevel2 = -1;
      if (m2 != ~0ull) { bestDist2 = (int)(m2 >> 40); bestLevel2 = kp[(int)(m2 & 0xFFFFFFFull)].octave; }
      const bool reject = bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2);
      if (!reject) {
        // every lane stores the same value (lane 0 need not be active in a divergent wave)
        owner[b1] = i;
        bestIdx2[i] = b1;
        ++nmatches;
        __threadfence_block();
      }
    }
    __syncthreads();
  }
  if (lane == 0) *nmatchesOut = nmatches;
}

// ---------------------------------------------------------------------------
// Frame::ComputeStereoFromRGBD (Frame.cc:1309-1331): depth of every left keypoint from a registered float
// depth image, mvuRight = x - bf / d.  (cv::Mat::at<float>(v, u) with float arguments truncates them.)
// ---------------------------------------------------------------------------
__global__ void k_stereo_from_depth(const DevParams* __restrict__ Pp, const float* __restrict__ depthImg, int64_t pitch,
                                    int W, int H, uint8_t* __restrict__ table, int64_t offCounts, int64_t offKp0,
                                    int64_t offUr, int64_t offDepth) {
  const DevParams& P = *Pp;
  const int N = reinterpret_cast<const int*>(table + offCounts)[0];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const pli_keypoint kp = reinterpret_cast<const pli_keypoint*>(table + offKp0)[i];
  const int u = (int)kp.x, v = (int)kp.y;
  float ur = -1.f, dp = -1.f;
  if (u >= 0 && v >= 0 && u < W && v < H) {
    const float d = depthImg[(int64_t)v * pitch + u];
    if (d > 0) { dp = d; ur = __fsub_rn(kp.x, __fdiv_rn(P.bf, d)); }
  }
  reinterpret_cast<float*>(table + offUr)[i] = ur;
  reinterpret_cast<float*>(table + offDepth)[i] = dp;
}

// ---------------------------------------------------------------------------
// Two-phase form of the two projection searches (used when the frame's keypoints fit the LDS owner table).
// Which keypoints a query may take (grid window, pyramid levels, radius, uRight gate) and their Hamming distances do
// not depend on the other queries; only "already taken" does.  Phase 1 (one wave per query, all queries in parallel)
// lists the candidates of every query as (distance, visiting order, index) keys; phase 2 (one wave, queries in
// order) takes the smallest key whose keypoint is still free — the reference's running minimum in visiting order —
// with the owner table in LDS and the next query's keys already in flight.
// ---------------------------------------------------------------------------
constexpr int PROJ_K = 64;          // candidates kept per query; more -> that query is rescanned in phase 2

__device__ __forceinline__ bool proj_window(const pli_proj_query& Q, float minX, float maxX, float minY, float maxY,
                                            float gwInv, float ghInv, bool checkBounds, int& c0, int& c1, int& r0, int& r1) {
  if (!Q.valid) return false;
  const float u = Q.u, v = Q.v, radius = Q.radius;
  if (checkBounds && (u < minX || u > maxX || v < minY || v > maxY)) return false;
  c0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(u, minX), radius), gwInv)));
  if (c0 >= GRID_COLS) return false;
  c1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(u, minX), radius), gwInv)));
  if (c1 < 0) return false;
  r0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(v, minY), radius), ghInv)));
  if (r0 >= GRID_ROWS) return false;
  r1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(v, minY), radius), ghInv)));
  return r1 >= 0;
}

// key of keypoint i2 for query Q (~0: not a candidate); everything except "already taken"
__device__ __forceinline__ unsigned long long proj_key(const pli_proj_query& Q, const uint64_t dq[4], int i2,
                                                       const pli_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
                                                       const float* __restrict__ uright, float minX, float minY, float gwInv,
                                                       float ghInv, int c0, int c1, int r0, int r1) {
  const pli_keypoint k = kp[i2];
  const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, minX), gwInv));
  const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, minY), ghInv));
  if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) return ~0ull;   // PosInGrid
  if (px < c0 || px > c1 || py < r0 || py > r1) return ~0ull;
  if ((Q.min_level > 0) || (Q.max_level >= 0)) {
    if (k.octave < Q.min_level) return ~0ull;
    if (Q.max_level >= 0 && k.octave > Q.max_level) return ~0ull;
  }
  if (!(fabsf(__fsub_rn(k.x, Q.u)) < Q.radius && fabsf(__fsub_rn(k.y, Q.v)) < Q.radius)) return ~0ull;
  const float ur2 = uright[i2];
  if (ur2 > 0 && fabsf(__fsub_rn(Q.ur, ur2)) > Q.radius) return ~0ull;
  uint64_t d2[4];
  load_desc(desc + (int64_t)i2 * 32, d2);
  const int dist = hamming256(dq, d2);
  return ((unsigned long long)dist << 40) | ((unsigned long long)px << 34) | ((unsigned long long)py << 28) | (unsigned long long)i2;
}

__device__ __forceinline__ void proj_candidates_dev(int i, int lane, const pli_proj_query* __restrict__ q,
                                                    const uint8_t* __restrict__ qdesc, int nq,
                                                    const pli_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
                                                    const float* __restrict__ uright, int ncur, float minX, float maxX,
                                                    float minY, float maxY, int checkBounds, int distLimit,
                                                    unsigned long long* __restrict__ candKeys, int* __restrict__ candCount) {
  if (i >= nq) return;
  const float gwInv = __fdiv_rn((float)GRID_COLS, __fsub_rn(maxX, minX));
  const float ghInv = __fdiv_rn((float)GRID_ROWS, __fsub_rn(maxY, minY));
  const pli_proj_query Q = q[i];
  int c0, c1, r0, r1;
  int count = 0;
  if (proj_window(Q, minX, maxX, minY, maxY, gwInv, ghInv, checkBounds != 0, c0, c1, r0, r1)) {
    uint64_t dq[4];
    load_desc(qdesc + (int64_t)i * 32, dq);
    for (int b = 0; b < ncur; b += 64) {
      const int i2 = b + lane;
      unsigned long long key = ~0ull;
      if (i2 < ncur) key = proj_key(Q, dq, i2, kp, desc, uright, minX, minY, gwInv, ghInv, c0, c1, r0, r1);
      const bool keep = key != ~0ull && (int)(key >> 40) <= distLimit;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
      const int pos = count + __popcll(bal & ((1ull << lane) - 1ull));
      if (keep && pos < PROJ_K) candKeys[(int64_t)i * PROJ_K + pos] = key;
      count += __popcll(bal);
    }
  }
  if (lane == 0) candCount[i] = count <= PROJ_K ? count : -1;
}

__global__ __launch_bounds__(64) void k_proj_candidates(const pli_proj_query* __restrict__ q, const uint8_t* __restrict__ qdesc,
                                                        int nq, const pli_keypoint* __restrict__ kp,
                                                        const uint8_t* __restrict__ desc, const float* __restrict__ uright,
                                                        int ncur, float minX, float maxX, float minY, float maxY,
                                                        int checkBounds, int distLimit, unsigned long long* __restrict__ candKeys,
                                                        int* __restrict__ candCount) {
  proj_candidates_dev(blockIdx.x, threadIdx.x, q, qdesc, nq, kp, desc, uright, ncur, minX, maxX, minY, maxY, checkBounds, distLimit,
                      candKeys, candCount);
}

// mode 0: SearchByProjection(CurrentFrame, LastFrame) (ORBmatcher.cc:2179-2323); mode 1: (Frame, MapPoints) (:44-143)
// owner: ncur ints of LDS: -1 free, else the query that took the keypoint (INT_MAX: occupied before)
__device__ __forceinline__ void proj_assign_dev(int lane, int* owner, const pli_proj_query* __restrict__ q,
                                                const uint8_t* __restrict__ qdesc, int nq,
                                                const pli_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
                                                const float* __restrict__ uright, const uint8_t* __restrict__ occupied,
                                                int ncur, float minX, float maxX, float minY, float maxY, int mode,
                                                int checkOri, float nnratio, const unsigned long long* __restrict__ candKeys,
                                                const int* __restrict__ candCount, int* __restrict__ bestIdx2,
                                                int* __restrict__ nmatchesOut, int* __restrict__ rawIdx2 /* mode 0, or null */) {
  __shared__ int hist[30];
  __shared__ int keep[30];
  const float gwInv = __fdiv_rn((float)GRID_COLS, __fsub_rn(maxX, minX));
  const float ghInv = __fdiv_rn((float)GRID_ROWS, __fsub_rn(maxY, minY));
  for (int i = lane; i < ncur; i += 64) owner[i] = (occupied && occupied[i]) ? INT_MAX : -1;
  for (int i = lane; i < nq; i += 64) bestIdx2[i] = -1;
  if (lane < 30) hist[lane] = 0;
  __syncthreads();
  int nmatches = 0;
  int cntNext = nq > 0 ? candCount[0] : 0;
  unsigned long long keyNext = (nq > 0 && lane < cntNext) ? candKeys[lane] : ~0ull;
  for (int i = 0; i < nq; ++i) {
    const int cnt = cntNext;
    unsigned long long key = keyNext;
    if (i + 1 < nq) {                                   // the next query's keys travel while this one is decided
      cntNext = candCount[i + 1];
      keyNext = lane < cntNext ? candKeys[(int64_t)(i + 1) * PROJ_K + lane] : ~0ull;
    }
    unsigned long long k1 = ~0ull, k2 = ~0ull;
    if (cnt >= 0) {
      if (key != ~0ull && owner[(int)(key & 0xFFFFFFFull)] >= 0) key = ~0ull;
      k1 = key;
    } else {                                            // more than PROJ_K candidates: scan the frame for this query
      const pli_proj_query Q = q[i];
      int c0, c1, r0, r1;
      if (proj_window(Q, minX, maxX, minY, maxY, gwInv, ghInv, mode == 0, c0, c1, r0, r1)) {
        uint64_t dq[4];
        load_desc(qdesc + (int64_t)i * 32, dq);
        for (int i2 = lane; i2 < ncur; i2 += 64) {
          if (owner[i2] >= 0) continue;
          const unsigned long long kk = proj_key(Q, dq, i2, kp, desc, uright, minX, minY, gwInv, ghInv, c0, c1, r0, r1);
          if (kk < k1) { k2 = k1; k1 = kk; }
          else if (kk < k2) k2 = kk;
        }
      }
    }
    const unsigned long long m1 = wave_min_u64(k1);
    if (m1 != ~0ull && (int)(m1 >> 40) <= 100) {
      const int b1 = (int)(m1 & 0xFFFFFFFull);
      bool accept = true;
      if (mode == 1) {
        const unsigned long long c2 = (k1 == m1) ? k2 : k1;
        const unsigned long long m2 = wave_min_u64(c2);
        const int bestDist = (int)(m1 >> 40), bestLevel = kp[b1].octave;
        int bestDist2 = 256, bestLevel2 = -1;
        if (m2 != ~0ull) { bestDist2 = (int)(m2 >> 40); bestLevel2 = kp[(int)(m2 & 0xFFFFFFFull)].octave; }
        accept = !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2));
      }
      if (accept) {
        // (every lane stores the same value.  Mode 0: a query whose map point has no observations — bit 1 of `valid` — does not take
        // its keypoint away from the queries behind it, ORBmatcher.cc:2259-2261)
        if (mode == 1 || !(q[i].valid & 2)) owner[b1] = i;
        bestIdx2[i] = b1;
        if (mode == 0 && checkOri && lane == 0) {
          float rot = __fsub_rn(q[i].angle, kp[b1].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
          if (bin == 30) bin = 0;
          if (bin >= 0 && bin < 30) hist[bin]++;
        }
        ++nmatches;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // single wave: LDS is executed in order
  }
  __syncthreads();
  if (mode == 0) {
    if (checkOri) {
      if (lane == 0) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; i++) {
          const int s = hist[i];
          if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
          else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
          else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        for (int i = 0; i < 30; ++i) keep[i] = (i == ind1 || i == ind2 || i == ind3);
      }
      __syncthreads();
    }
    // the rotation filter, per accepted query (:2303-2320: every entry of a rejected bin takes one off nmatches)
    int removed = 0;
    for (int i = lane; i < nq; i += 64) {
      const int b = bestIdx2[i];
      if (rawIdx2) rawIdx2[i] = b;
      if (b < 0 || !checkOri) continue;
      float rot = __fsub_rn(q[i].angle, kp[b].angle);
      if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
      int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
      if (bin == 30) bin = 0;
      if (!(bin >= 0 && bin < 30 && keep[bin])) { bestIdx2[i] = -1; ++removed; }
    }
    removed = wave_sum_i32(removed);
    nmatches -= removed;
  }
  if (lane == 0) *nmatchesOut = nmatches;
}

__global__ __launch_bounds__(64) void k_proj_assign(const pli_proj_query* __restrict__ q, const uint8_t* __restrict__ qdesc, int nq,
                                                    const pli_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
                                                    const float* __restrict__ uright, const uint8_t* __restrict__ occupied,
                                                    int ncur, float minX, float maxX, float minY, float maxY, int mode,
                                                    int checkOri, float nnratio, const unsigned long long* __restrict__ candKeys,
                                                    const int* __restrict__ candCount, int* __restrict__ bestIdx2,
                                                    int* __restrict__ nmatchesOut, int* __restrict__ rawIdx2) {
  extern __shared__ int owner[];
  proj_assign_dev(threadIdx.x, owner, q, qdesc, nq, kp, desc, uright, occupied, ncur, minX, maxX, minY, maxY, mode, checkOri, nnratio,
                  candKeys, candCount, bestIdx2, nmatchesOut, rawIdx2);
}

// ---------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th) for a frame of two fisheye cameras (ORBmatcher.cc:44-214 with
// F.Nleft != -1).  The candidates of every (map point, camera)
// come from k_proj_candidates, all in parallel; this wave walks the map points in order — left camera, then right unless
// the left ratio test sent the map point away — with the two halves of F.mvpMapPoints as owner tables in LDS (-1 free,
// INT_MAX taken before the call, else the map point that wrote the slot last), and writes a match to the stereo partner
// of the keypoint as the reference does (mvLeftToRightMatch / mvRightToLeftMatch; the partner's slot is overwritten, taken
// or not).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_proj_assign_fisheye(
    const pli_proj_query* __restrict__ qL, const pli_proj_query* __restrict__ qR, const uint8_t* __restrict__ qdesc, int nq,
    const pli_keypoint* __restrict__ kpL, const uint8_t* __restrict__ descL, const uint8_t* __restrict__ occL,
    const int* __restrict__ l2r, int nL, const pli_keypoint* __restrict__ kpR, const uint8_t* __restrict__ descR,
    const uint8_t* __restrict__ occR, const int* __restrict__ r2l, int nR, const float* __restrict__ noUright, float minX,
    float maxX, float minY, float maxY, float nnratio, const unsigned long long* __restrict__ keysL,
    const int* __restrict__ cntL, const unsigned long long* __restrict__ keysR, const int* __restrict__ cntR,
    int* __restrict__ mpL, int* __restrict__ mpR, int* __restrict__ nmatchesOut) {
  extern __shared__ int owner[];                         // [nL] left slots, [nR] right slots
  int* ownL = owner;
  int* ownR = owner + nL;
  const int lane = threadIdx.x;
  const float gwInv = __fdiv_rn((float)GRID_COLS, __fsub_rn(maxX, minX));
  const float ghInv = __fdiv_rn((float)GRID_ROWS, __fsub_rn(maxY, minY));
  for (int i = lane; i < nL; i += 64) ownL[i] = (occL && occL[i]) ? INT_MAX : -1;
  for (int i = lane; i < nR; i += 64) ownR[i] = (occR && occR[i]) ? INT_MAX : -1;
  __syncthreads();
  int nmatches = 0;
  // best and second best of one camera among the free keypoints; returns 0 nothing within TH_HIGH, 1 accepted (b1), 2 the
  // ratio test failed (ORBmatcher.cc:124-126 / :197-199)
  auto search = [&](int i, const pli_proj_query* q, const pli_keypoint* kp, const uint8_t* desc, int ncur, const int* own,
                    const unsigned long long* keys, const int* cnts, int& b1) -> int {
    const int cnt = cnts[i];
    unsigned long long k1 = ~0ull, k2 = ~0ull;
    if (cnt >= 0) {
      unsigned long long key = lane < cnt ? keys[(int64_t)i * PROJ_K + lane] : ~0ull;
      if (key != ~0ull && own[(int)(key & 0xFFFFFFFull)] >= 0) key = ~0ull;
      k1 = key;
    } else {                                             // more than PROJ_K candidates: scan the camera for this query
      const pli_proj_query Q = q[i];
      int c0, c1, r0, r1;
      if (proj_window(Q, minX, maxX, minY, maxY, gwInv, ghInv, false, c0, c1, r0, r1)) {
        uint64_t dq[4];
        load_desc(qdesc + (int64_t)i * 32, dq);
        for (int i2 = lane; i2 < ncur; i2 += 64) {
          if (own[i2] >= 0) continue;
          const unsigned long long kk = proj_key(Q, dq, i2, kp, desc, noUright, minX, minY, gwInv, ghInv, c0, c1, r0, r1);
          if (kk < k1) { k2 = k1; k1 = kk; }
          else if (kk < k2) k2 = kk;
        }
      }
    }
    const unsigned long long m1 = wave_min_u64(k1);
    if (m1 == ~0ull || (int)(m1 >> 40) > 100) return 0;
    b1 = (int)(m1 & 0xFFFFFFFull);
    const unsigned long long c2 = (k1 == m1) ? k2 : k1;
    const unsigned long long m2 = wave_min_u64(c2);
    const int bestDist = (int)(m1 >> 40), bestLevel = kp[b1].octave;
    int bestDist2 = 256, bestLevel2 = -1;
    if (m2 != ~0ull) { bestDist2 = (int)(m2 >> 40); bestLevel2 = kp[(int)(m2 & 0xFFFFFFFull)].octave; }
    return (bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)) ? 2 : 1;
  };
  for (int i = 0; i < nq; ++i) {
    int b1 = -1;
    bool leave = false;
    if (qL[i].valid) {
      const int r = search(i, qL, kpL, descL, nL, ownL, keysL, cntL, b1);
      if (r == 2) leave = true;
      if (r == 1) {
        ownL[b1] = i;                                     // every lane stores the same value
        const int p = l2r[b1];
        if (p != -1) { ownR[p] = i; ++nmatches; }
        ++nmatches;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // single wave: LDS is executed in order
    }
    if (!leave && qR[i].valid) {
      const int r = search(i, qR, kpR, descR, nR, ownR, keysR, cntR, b1);
      if (r == 1) {
        const int p = r2l[b1];
        if (p != -1) { ownL[p] = i; ++nmatches; }
        ownR[b1] = i;
        ++nmatches;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
  __syncthreads();
  for (int i = lane; i < nL; i += 64) { const int o = ownL[i]; mpL[i] = (o >= 0 && o != INT_MAX) ? o : -1; }
  for (int i = lane; i < nR; i += 64) { const int o = ownR[i]; mpR[i] = (o >= 0 && o != INT_MAX) ? o : -1; }
  if (lane == 0) *nmatchesOut = nmatches;
}

// ---------------------------------------------------------------------------
// Frame-to-frame track matching of a whole batch on the device tables (pli_batch_track): frame i against frame i-1.
//   k_track_queries     the projection part of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
//                       (ORBmatcher.cc:2190-2244) with LastFrame's stereo points standing for its map points
//                       (Frame::UnprojectStereo, Frame.cc:1334-1350, as Tracking::UpdateLastFrame creates them)
//   k_track_candidates / k_track_assign   the two-phase search above, one launch for all frame pairs
//   k_track_lines       match(last.mDescriptors_Line, cur.mDescriptors_Line, nnr, matches_12) (LineMatcher.cpp:201-229)
// cv::Mat arithmetic (CV_32F): a product A*B (+ C) is OpenCV's gemm, restated here as double accumulation of the float
// products, then ONE rounding to float ("OpenCV-3.3.1-compatible by intent", like the other OpenCV primitives).
// ---------------------------------------------------------------------------

__device__ __forceinline__ float cvmat_dot3(const float* a, int sa, const float* b, double alpha, double c) {
  const double d = (double)a[0] * (double)b[0] + (double)a[sa] * (double)b[1] + (double)a[2 * sa] * (double)b[2];
  return (float)(alpha * d + c);
}

__global__ __launch_bounds__(256) void k_track_queries(const DevParams* __restrict__ Pp, const uint8_t* __restrict__ table,
                                                       const float* __restrict__ poses, TrackParams tp,
                                                       pli_proj_query* __restrict__ qAll) {
  const DevParams& P = *Pp;
  const int frame = blockIdx.y + 1, j = blockIdx.x * 256 + threadIdx.x;     // frame >= 1 is matched against frame - 1
  const uint8_t* last = table + (int64_t)(frame - 1) * tp.recordBytes;
  const int nq = reinterpret_cast<const int*>(last + tp.offCounts)[0];
  if (j >= nq) return;
  const float* Tc = poses + (int64_t)frame * 12;        // mTcw of the current frame, row major 3x4
  const float* Tl = poses + (int64_t)(frame - 1) * 12;  // mTcw of the last frame
  const float tcw[3] = {Tc[3], Tc[7], Tc[11]}, tlw[3] = {Tl[3], Tl[7], Tl[11]};
  // twc = -Rcw.t()*tcw; tlc = Rlw*twc+tlw (ORBmatcher.cc:2193-2200)
  float twc[3], tlc[3];
  for (int r = 0; r < 3; ++r) twc[r] = cvmat_dot3(Tc + r, 4, tcw, -1.0, 0.0);
  for (int r = 0; r < 3; ++r) tlc[r] = cvmat_dot3(Tl + 4 * r, 1, twc, 1.0, (double)tlw[r]);
  const float mb = __fdiv_rn(tp.bf, tp.fx);             // Frame.cc:197
  const bool bForward = tlc[2] > mb && !tp.mono, bBackward = -tlc[2] > mb && !tp.mono;
  const pli_keypoint k = reinterpret_cast<const pli_keypoint*>(last + tp.offKp0)[j];
  const float z = reinterpret_cast<const float*>(last + tp.offDepth)[j];
  pli_proj_query Q;
  Q.u = 0.f; Q.v = 0.f; Q.radius = 0.f; Q.ur = 0.f; Q.min_level = 0; Q.max_level = -1; Q.angle = k.angle; Q.valid = 0;
  if (z > 0) {
    // LastFrame.UnprojectStereo(j): x3Dc = ((u-cx)*z*invfx, (v-cy)*z*invfy, z); x3Dw = mRwc*x3Dc+mOw with mRwc = Rlw.t(), mOw = -Rlw.t()*tlw
    const float invfx = __fdiv_rn(1.0f, tp.fx), invfy = __fdiv_rn(1.0f, tp.fy);
    const float xl[3] = {__fmul_rn(__fmul_rn(__fsub_rn(k.x, tp.cx), z), invfx), __fmul_rn(__fmul_rn(__fsub_rn(k.y, tp.cy), z), invfy), z};
    float Ow[3], xw[3], xc[3];
    for (int r = 0; r < 3; ++r) Ow[r] = cvmat_dot3(Tl + r, 4, tlw, -1.0, 0.0);
    for (int r = 0; r < 3; ++r) xw[r] = cvmat_dot3(Tl + r, 4, xl, 1.0, (double)Ow[r]);
    // x3Dc = Rcw*x3Dw+tcw (ORBmatcher.cc:2212)
    for (int r = 0; r < 3; ++r) xc[r] = cvmat_dot3(Tc + 4 * r, 1, xw, 1.0, (double)tcw[r]);
    const float invzc = (float)(1.0 / (double)xc[2]);
    if (!(invzc < 0)) {
      Q.u = __fadd_rn(__fmul_rn(__fmul_rn(tp.fx, xc[0]), invzc), tp.cx);
      Q.v = __fadd_rn(__fmul_rn(__fmul_rn(tp.fy, xc[1]), invzc), tp.cy);
      Q.radius = __fmul_rn(tp.th, P.lv[k.octave].scale);
      Q.ur = __fsub_rn(Q.u, __fmul_rn(tp.bf, invzc));
      if (bForward) { Q.min_level = k.octave; Q.max_level = -1; }
      else if (bBackward) { Q.min_level = 0; Q.max_level = k.octave; }
      else { Q.min_level = k.octave - 1; Q.max_level = k.octave + 1; }
      Q.valid = 1;
    }
  }
  qAll[(int64_t)frame * tp.kpCap + j] = Q;
}

__global__ __launch_bounds__(64) void k_track_candidates(const uint8_t* __restrict__ table, TrackParams tp,
                                                         const pli_proj_query* __restrict__ qAll,
                                                         unsigned long long* __restrict__ candKeysAll, int* __restrict__ candCountAll) {
  const int frame = blockIdx.y + 1;
  const uint8_t* last = table + (int64_t)(frame - 1) * tp.recordBytes;
  const uint8_t* cur = table + (int64_t)frame * tp.recordBytes;
  const int nq = reinterpret_cast<const int*>(last + tp.offCounts)[0], ncur = reinterpret_cast<const int*>(cur + tp.offCounts)[0];
  proj_candidates_dev(blockIdx.x, threadIdx.x, qAll + (int64_t)frame * tp.kpCap, last + tp.offDesc0, nq,
                      reinterpret_cast<const pli_keypoint*>(cur + tp.offKp0), cur + tp.offDesc0,
                      reinterpret_cast<const float*>(cur + tp.offUr), ncur, tp.minX, tp.maxX, tp.minY, tp.maxY, 1, 100,
                      candKeysAll + (int64_t)frame * tp.kpCap * PROJ_K, candCountAll + (int64_t)frame * tp.kpCap);
}

__global__ __launch_bounds__(64) void k_track_assign(const uint8_t* __restrict__ table, TrackParams tp,
                                                     const pli_proj_query* __restrict__ qAll,
                                                     const unsigned long long* __restrict__ candKeysAll,
                                                     const int* __restrict__ candCountAll, uint8_t* __restrict__ track) {
  extern __shared__ int owner[];
  const int frame = blockIdx.x + 1;
  const uint8_t* last = table + (int64_t)(frame - 1) * tp.recordBytes;
  const uint8_t* cur = table + (int64_t)frame * tp.recordBytes;
  uint8_t* out = track + (int64_t)frame * tp.trackBytes;
  const int nq = reinterpret_cast<const int*>(last + tp.offCounts)[0], ncur = reinterpret_cast<const int*>(cur + tp.offCounts)[0];
  int* counts = reinterpret_cast<int*>(out + tp.toffCounts);
  if (threadIdx.x == 0) counts[0] = nq;
  proj_assign_dev(threadIdx.x, owner, qAll + (int64_t)frame * tp.kpCap, last + tp.offDesc0, nq,
                  reinterpret_cast<const pli_keypoint*>(cur + tp.offKp0), cur + tp.offDesc0,
                  reinterpret_cast<const float*>(cur + tp.offUr), nullptr, ncur, tp.minX, tp.maxX, tp.minY, tp.maxY, 0, tp.checkOri, 0.f,
                  candKeysAll + (int64_t)frame * tp.kpCap * PROJ_K, candCountAll + (int64_t)frame * tp.kpCap,
                  reinterpret_cast<int*>(out + tp.toffBest), counts + 1, nullptr);
}

// one workgroup per frame pair: knn2 both ways, ratio tests, mutual check (descriptor tables of <= klCap lines)
__global__ __launch_bounds__(256) void k_track_lines(const uint8_t* __restrict__ table, TrackParams tp, int bestLR,
                                                     int* __restrict__ scratch, uint8_t* __restrict__ track) {
  __shared__ int s_cnt;
  const int frame = blockIdx.x + 1, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint8_t* last = table + (int64_t)(frame - 1) * tp.recordBytes;
  const uint8_t* cur = table + (int64_t)frame * tp.recordBytes;
  uint8_t* out = track + (int64_t)frame * tp.trackBytes;
  const int n1 = reinterpret_cast<const int*>(last + tp.offCounts)[2], n2 = reinterpret_cast<const int*>(cur + tp.offCounts)[2];
  const uint8_t* d1 = last + tp.offLd0;
  const uint8_t* d2 = cur + tp.offLd0;
  int* m12 = reinterpret_cast<int*>(out + tp.toffLines);
  int* m21 = scratch + (int64_t)frame * tp.klCap;
  if (tid == 0) s_cnt = 0;
  for (int dir = 0; dir < (bestLR ? 2 : 1); ++dir) {
    const uint8_t* q = dir ? d2 : d1;
    const uint8_t* t = dir ? d1 : d2;
    const int nq = dir ? n2 : n1, nt = dir ? n1 : n2;
    int* m = dir ? m21 : m12;
    for (int i = wv; i < nq; i += 4) {
      uint64_t dq[4];
      load_desc(q + (int64_t)i * 32, dq);
      unsigned long long k1 = ~0ull, k2 = ~0ull;
      for (int j = lane; j < nt; j += 64) {
        uint64_t dt[4];
        load_desc(t + (int64_t)j * 32, dt);
        const unsigned long long key = ((unsigned long long)hamming256(dq, dt) << 32) | (unsigned)j;
        if (key < k1) { k2 = k1; k1 = key; }
        else if (key < k2) k2 = key;
      }
      const unsigned long long m1 = wave_min_u64(k1);
      const unsigned long long c2 = (k1 == m1) ? k2 : k1;
      const unsigned long long m2 = wave_min_u64(c2);
      if (lane == 0) {
        int r = -1;
        if (nt >= 2 && (float)(int)(m1 >> 32) < __fmul_rn((float)(int)(m2 >> 32), tp.nnrLines)) r = (int)(m1 & 0xFFFFFFFFu);
        m[i] = r;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  int ok = 0;
  for (int i = tid; i < n1; i += 256) {
    int i2 = m12[i];
    if (i2 >= 0 && bestLR && n2 > 0 && m21[i2] != i) { m12[i] = -1; i2 = -1; }
    ok += i2 >= 0;
  }
  if (ok) atomicAdd(&s_cnt, ok);
  __syncthreads();
  if (tid == 0) {
    int* counts = reinterpret_cast<int*>(out + tp.toffCounts);
    counts[2] = n1;
    counts[3] = s_cnt;
  }
}

// ---------------------------------------------------------------------------
// DBoW2 vocabulary descent (TemplatedVocabulary.h:1230-1272): one wave per feature; at every level the lanes take
// one child each (k <= 64), a 64-bit (distance, position) key wave-min picks the nearest child, the first on ties.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bow_descend(const uint8_t* __restrict__ feat, int n, const int* __restrict__ childOff,
                                                    const int* __restrict__ childList, const uint8_t* __restrict__ nodeDesc,
                                                    const int* __restrict__ nodeWord, const double* __restrict__ nodeWeight,
                                                    int nidLevel, int* __restrict__ wordId, double* __restrict__ weight,
                                                    int* __restrict__ nodeId) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= n) return;
  uint64_t f[4];
  load_desc(feat + (int64_t)i * 32, f);
  int cur = 0, level = 0, nid = 0;
  for (;;) {
    const int c0 = childOff[cur], c1 = childOff[cur + 1];
    if (c0 == c1) break;                                   // Node::isLeaf(): no children
    ++level;
    unsigned long long best = ~0ull;
    for (int c = c0 + lane; c < c1; c += 64) {
      uint64_t d[4];
      load_desc(nodeDesc + (int64_t)childList[c] * 32, d);
      const unsigned long long key = ((unsigned long long)hamming256(f, d) << 32) | (unsigned)(c - c0);
      best = key < best ? key : best;
    }
    best = wave_min_u64(best);
    cur = childList[c0 + (int)(best & 0xFFFFFFFFull)];
    if (level == nidLevel) nid = cur;
  }
  if (lane == 0) {
    wordId[i] = nodeWord[cur];
    weight[i] = nodeWeight[cur];
    nodeId[i] = nid;
  }
}

}  // namespace pli

namespace pli {

// ---------------------------------------------------------------------------
// SURVEY §8(f) row 4, fisheye stereo.
// Frame::ComputeStereoFishEyeMatches (Frame.cc:1577-1618): after knnMatch(k = 2) of the lapping-area descriptors (k_knn2),
// Lowe's ratio 0.7 and KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:334-402) per surviving pair:
// unproject both keypoints (Newton on the distortion polynomial, :103-130), parallax test, linear triangulation (:422-435),
// positive depths, reprojection errors against 5.991 * sigma^2 (project, :28-42).
// One thread per left lapping-area keypoint; cv::Mat arithmetic as OpenCV 3.3.1 evaluates it: gemm = double accumulation and
// one rounding, Mat::dot / norm in double, cv::SVD::compute = the one-sided float Jacobi of lapack.cpp (4x4, <= 30 sweeps,
// hypot as sqrt(p*p + beta*beta)), libm calls on floats in double and rounded.  Same statement as the checker's
// (match_oracle.hpp kb8TriangulateMatches), operation for operation.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float mat_dot3(const float* a, int sa, const float* b, double alpha, double c) {
  const double d = (double)a[0] * (double)b[0] + (double)a[sa] * (double)b[1] + (double)a[2 * sa] * (double)b[2];
  return (float)(alpha * d + c);
}

__device__ __forceinline__ void kb8_unproject(const Kb8& c, float u, float v, float r[3]) {
  const float pwx = (u - c.cx) / c.fx, pwy = (v - c.cy) / c.fy;
  float scale = 1.f;
  float theta_d = sqrtf(pwx * pwx + pwy * pwy);
  theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.0), theta_d), (float)(3.1415926535897932384626433832795 / 2.0));
  if ((double)theta_d > 1e-8) {
    float theta = theta_d;
    for (int j = 0; j < 10; j++) {
      const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      const float k0t2 = c.k0 * theta2, k1t4 = c.k1 * theta4, k2t6 = c.k2 * theta6, k3t8 = c.k3 * theta8;
      const float fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) / (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
      theta = theta - fix;
      if (fabsf(fix) < 1e-6f) break;                  // KannalaBrandt8::precision
    }
    scale = (float)tan((double)theta) / theta_d;
  }
  r[0] = pwx * scale; r[1] = pwy * scale; r[2] = 1.f;
}

__device__ __forceinline__ void kb8_project(const Kb8& c, const float p[3], float& u, float& v) {
  const float x2y2 = p[0] * p[0] + p[1] * p[1];
  const float theta = (float)atan2((double)sqrtf(x2y2), (double)p[2]);
  const float psi = (float)atan2((double)p[1], (double)p[0]);
  const float t2 = theta * theta, t3 = theta * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
  const float r = theta + c.k0 * t3 + c.k1 * t5 + c.k2 * t7 + c.k3 * t9;
  u = (float)((double)(c.fx * r) * cos((double)psi) + (double)c.cx);
  v = (float)((double)(c.fy * r) * sin((double)psi) + (double)c.cy);
}

// last row of vt of cv::SVD::compute(A) for a 4x4 CV_32F matrix; At = A transposed
__device__ __forceinline__ void jacobi_svd_last_vt4(float At[4][4], float out[4]) {
  double W[4];
  float Vt[4][4];
  const float eps = 2.384185791015625e-07f;             // FLT_EPSILON * 2
  for (int i = 0; i < 4; i++) {
    double sd = 0;
    for (int k = 0; k < 4; k++) { const float t = At[i][k]; sd += (double)t * t; }
    W[i] = sd;
    for (int k = 0; k < 4; k++) Vt[i][k] = 0;
    Vt[i][i] = 1;
  }
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = i + 1; j < 4; j++) {
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < 4; k++) p += (double)At[i][k] * At[j][k];
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
        for (int k = 0; k < 4; k++) {
          const float t0 = c * At[i][k] + s * At[j][k];
          const float t1 = -s * At[i][k] + c * At[j][k];
          At[i][k] = t0; At[j][k] = t1;
          a += (double)t0 * t0; b += (double)t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        for (int k = 0; k < 4; k++) {
          const float t0 = c * Vt[i][k] + s * Vt[j][k];
          const float t1 = -s * Vt[i][k] + c * Vt[j][k];
          Vt[i][k] = t0; Vt[j][k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < 4; i++) {
    double sd = 0;
    for (int k = 0; k < 4; k++) { const float t = At[i][k]; sd += (double)t * t; }
    W[i] = sqrt(sd);
  }
  // the descending selection sort of lapack.cpp only matters for where the smallest singular value ends up: row 3 takes
  // part in a swap at step i iff it holds the maximum of W[i..3] (the first maximum wins: W[j] < W[k] is strict)
  int idx[4] = {0, 1, 2, 3};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int j = i;
#pragma unroll
    for (int k = i + 1; k < 4; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      const double tw = W[i]; W[i] = W[j]; W[j] = tw;
      const int ti = idx[i]; idx[i] = idx[j]; idx[j] = ti;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (idx[3] == r)
      for (int k = 0; k < 4; k++) out[k] = Vt[r][k];
}

__global__ __launch_bounds__(64) void k_fisheye_triangulate(const pli_keypoint* __restrict__ kpL, const pli_keypoint* __restrict__ kpR,
                                                            const int* __restrict__ knnIdx, const int* __restrict__ knnDist, int nl, int nr,
                                                            int monoL, int monoR, Kb8 c1, Kb8 c2, const float* __restrict__ R12t12,
                                                            const float* __restrict__ sigma2, int* __restrict__ l2r, int* __restrict__ r2l,
                                                            float* __restrict__ depth, float* __restrict__ p3d, int* __restrict__ nmatches) {
  const int i = blockIdx.x * 64 + threadIdx.x;          // index into the lapping-area (stereo) part of the left table
  if (i >= nl || nr < 2) return;                        // (*it).size() >= 2
  const int d0 = knnDist[2 * i], d1 = knnDist[2 * i + 1], j = knnIdx[2 * i];
  if (!((double)(float)d0 < (double)(float)d1 * 0.7)) return;
  const pli_keypoint k1 = kpL[monoL + i], k2 = kpR[monoR + j];
  const float* R12 = R12t12;
  const float* t12 = R12t12 + 9;
  float r1[3], r2[3], r21[3];
  kb8_unproject(c1, k1.x, k1.y, r1);
  kb8_unproject(c2, k2.x, k2.y, r2);
  for (int a = 0; a < 3; ++a) r21[a] = mat_dot3(R12 + 3 * a, 1, r2, 1.0, 0.0);
  const double dot = (double)r1[0] * r21[0] + (double)r1[1] * r21[1] + (double)r1[2] * r21[2];
  const double n1 = sqrt((double)r1[0] * r1[0] + (double)r1[1] * r1[1] + (double)r1[2] * r1[2]);
  const double n2 = sqrt((double)r21[0] * r21[0] + (double)r21[1] * r21[1] + (double)r21[2] * r21[2]);
  const float cosPar = (float)(dot / (n1 * n2));
  if ((double)cosPar > 0.9998) return;
  // Tcw1 = [I | 0], Tcw2 = [R21 | t21] with R21 = R12^T, t21 = -R21 t12
  float R21[9], t21[3];
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) R21[3 * a + b] = R12[3 * b + a];
  for (int a = 0; a < 3; ++a) t21[a] = mat_dot3(R21 + 3 * a, 1, t12, -1.0, 0.0);
  const float T1[3][4] = {{1.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f}};
  float T2[3][4];
  for (int a = 0; a < 3; ++a) { T2[a][0] = R21[3 * a]; T2[a][1] = R21[3 * a + 1]; T2[a][2] = R21[3 * a + 2]; T2[a][3] = t21[a]; }
  float At[4][4];
  for (int b = 0; b < 4; ++b) {
    At[b][0] = r1[0] * T1[2][b] - T1[0][b];
    At[b][1] = r1[1] * T1[2][b] - T1[1][b];
    At[b][2] = r2[0] * T2[2][b] - T2[0][b];
    At[b][3] = r2[1] * T2[2][b] - T2[1][b];
  }
  float vh[4];
  jacobi_svd_last_vt4(At, vh);
  const float inv = (float)(1.0 / (double)vh[3]);
  const float x3[3] = {vh[0] * inv + 0.f, vh[1] * inv + 0.f, vh[2] * inv + 0.f};
  const float z1 = x3[2];
  if (!(z1 > 0)) return;                                // z1 <= 0 (or NaN: `depth > 0.0001f` fails for it below in the reference)
  const float z2 = (float)(((double)R21[6] * x3[0] + (double)R21[7] * x3[1] + (double)R21[8] * x3[2]) + (double)t21[2]);
  if (z2 <= 0) return;
  float u1, v1;
  kb8_project(c1, x3, u1, v1);
  const float ex1 = u1 - k1.x, ey1 = v1 - k1.y;
  if ((double)(ex1 * ex1 + ey1 * ey1) > 5.991 * (double)sigma2[k1.octave]) return;
  float x32[3];
  for (int a = 0; a < 3; ++a) x32[a] = mat_dot3(R21 + 3 * a, 1, x3, 1.0, (double)t21[a]);
  float u2, v2;
  kb8_project(c2, x32, u2, v2);
  const float ex2 = u2 - k2.x, ey2 = v2 - k2.y;
  if ((double)(ex2 * ex2 + ey2 * ey2) > 5.991 * (double)sigma2[k2.octave]) return;
  if (!(z1 > 0.0001f)) return;
  l2r[monoL + i] = monoR + j;
  atomicMax(&r2l[monoR + j], monoL + i);        // the reference's loop overwrites: the LAST left keypoint that takes a right one stays
  depth[monoL + i] = z1;
  p3d[3 * (monoL + i)] = x3[0]; p3d[3 * (monoL + i) + 1] = x3[1]; p3d[3 * (monoL + i) + 2] = x3[2];
  atomicAdd(nmatches, 1);
}

__global__ void k_fill_f32(float* __restrict__ dst, int n, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}

// ---------------------------------------------------------------------------
// The keypoint order ORBextractor::operator() leaves when a lapping area is given (ORBextractor.cc:1135-1144): keypoints
// with lap0 <= x <= lap1 (level-0 coordinates) fill the table from the back in visiting order, the others from the front.
// One workgroup per image: ordered counts by block scans over chunks of 1024 keypoints; src = snapshot of the table.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_lapping_order(const pli_keypoint* __restrict__ srcKp, const uint8_t* __restrict__ srcDesc,
                                                        int n, float lap0, float lap1, pli_keypoint* __restrict__ dstKp,
                                                        uint8_t* __restrict__ dstDesc, int* __restrict__ monoCount) {
  __shared__ int waveCnt[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + tid;
    pli_keypoint k;
    bool mono = false;
    if (i < n) { k = srcKp[i]; mono = !(k.x >= lap0 && k.x <= lap1); }
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(mono);
    if (lane == 0) waveCnt[wv] = __popcll(bal);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wv; ++w) before += waveCnt[w];
    before += __popcll(bal & ((1ull << lane) - 1ull));       // mono keypoints before i
    if (i < n) {
      const int dst = mono ? before : n - 1 - (i - before);
      dstKp[dst] = k;
      const uint4* s4 = reinterpret_cast<const uint4*>(srcDesc + (int64_t)i * 32);
      uint4* d4 = reinterpret_cast<uint4*>(dstDesc + (int64_t)dst * 32);
      d4[0] = s4[0]; d4[1] = s4[1];
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += waveCnt[w]; base += t; }
    __syncthreads();
  }
  if (tid == 0) *monoCount = base;
}

}  // namespace pli
