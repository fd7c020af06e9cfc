// LSD region growing as a TILE-SEQUENTIAL rank-ordered relaxation (lsd_mode 3) — exact at convergence,
// like lsd_relax.hip, but with the sequential order kept INSIDE every tile:
//
//   * the scaled image is cut into tiles of ts x ts pixels; one wave per tile walks the seeds of ITS tile in rank
//     order (k_tx_sort builds the per-tile lists once) and grows each alive one with the batched wave-wide steps of
//     the sequential grower (8 queue entries x 8 neighbours per round trip, line_kernels.hip);
//   * ownership is the relaxation's: owner_t[q] = rank of the region that takes q in round t, claimed with
//     atomicMin; a pixel is used for region r iff owner_{t-1}[q] < r or a lower rank claimed it in this round.
//     The claims of the wave's own earlier regions are always visible to it, so every dependency chain that stays
//     inside a tile is resolved in ONE round; only chains that cross tile borders need further rounds.
//
// Correctness is the relaxation's induction on rank (lsd_relax.hip header): if every rank below r is final in
// owner_{t-1}, region r reads only final data in round t (claims seen in the round are claims of lower ranks and
// persist in owner_t) and is computed exactly, for ANY owner_0 and any interleaving of the tile waves; hence
// owner_t == owner_{t-1} implies the sequential result.  Measured on the EuRoC-shaped stream (CPU simulation
// tools/sim/sim_tile_relax.cpp and the GPU counters agree): 7-8 rounds instead of 13-17, 1.4x the sequential
// steps in round 1 (tiles of 64) and a few per cent of them from round 3 on, where only the regions next to a
// change are regrown (k_rx_diff / k_rx_mark of lsd_relax.hip decide which, k_tx_prep pre-claims the others).
//
// owner_0 is the trivial map (every pixel its own region): rounds 1 and 2 regrow everything, round t >= 3 only
// what k_rx_mark stamps dirty.
#include "kernels.hpp"
#include "device_prims.hpp"
#include "lsd_rect.hpp"
#include <climits>
#include <type_traits>

namespace pli {

constexpr float TX_NOTDEF = -1024.f;
constexpr int TX_INF = LSD_ID_INF;          // rank / id plane of undefined pixels
constexpr double TX_PI = 3.14159265358979323846;
constexpr double TX_DEG2RAD = TX_PI / 180;
constexpr double TX_3_2_PI = (3 * TX_PI) / 2;
constexpr double TX_2PI = 2 * TX_PI;
// Variants of the step that were built, measured at 256 frames and shelved — all exact (parity tests and fuzzers green); the code is
// in the history (commits 04f13dc .. 1fb47bd):
//   * claims not waited for (non-returning atomicMin, or returning ones consumed after the next step's loads are issued): the loads
//     are issued by the same wave in program order and meet the claims in the same L2 channel, so they still see them; slower with
//     32 tile waves per CU (k_tx_grow 24.4 -> 26.8 / 26.7 ms), a hair faster at 32 frames;
//   * the wave's own claims of the round as an LDS bitmap, the owner map read only for owner_{t-1} (one round trip per step, claims
//     fire-and-forget): k_tx_grow 24.4 -> 28.2 ms, later rounds 24.3 -> 31.0 ms — the claims of OTHER tiles' waves made in the same
//     round are no longer seen and are regrown a round later;
//   * the same bitmap only to skip the loads of seeds the wave itself has taken: a third of the L2 misses gone (854 M -> 577 M),
//     time unchanged.  (What binds the round-1 grower, as far as rounds 4 and 5 could separate it: VALU issue time — 71 % of the
//     chip's —, the two dependent trips of a step at 8 waves per SIMD, and the gather rate, in comparable shares: DESIGN.md 5 "Round 5".)
// (the diagnostic macros of this file — TX_CLAIM_SCOPE, TX_GROUP, TX_OWN_LOAD, TX_DIAG_PAD, TX_DIAG_NOWAIT, TX_DIAG_HOT_NOFOLD — belong to
// development builds: tools/build_variant.sh passes -DPLI_DEV with them)
#if !defined(PLI_DEV) && (defined(TX_CLAIM_SCOPE) || defined(TX_GROUP) || defined(TX_OWN_LOAD) || defined(TX_DIAG_PAD) || defined(TX_DIAG_NOWAIT) || defined(TX_DIAG_HOT_NOFOLD))
#error "diagnostic builds of lsd_tile.hip need -DPLI_DEV"
#endif
#ifndef TX_CLAIM_SCOPE             // (diagnostic builds only: -DTX_CLAIM_SCOPE=__HIP_MEMORY_SCOPE_WORKGROUP times the claims as L2 atomics —
#define TX_CLAIM_SCOPE __HIP_MEMORY_SCOPE_AGENT   // NOT coherent between the XCDs' L2s, so not exact unless an image stays on one XCD)
#endif
// Shared first steps (round 5): the alive seeds of a list row start in GROUPS of up to TX_GROUP — octet g of the wave fetches the 8
// neighbours of member g's seed, ONE load round trip for the group; the fetched records are parked in LDS and every member takes
// its first step from them at its turn, in id order, with nothing claimed before that (see tx_grow_tile).  0: every region fetches
// its own first step (the schedule of rounds 2-4; A/B builds).
#ifndef TX_GROUP
#define TX_GROUP 8
#endif
static_assert(TX_GROUP == 0 || TX_GROUP == 2 || TX_GROUP == 4 || TX_GROUP == 8, "a group member is an octet of lanes");
constexpr int TX_GQ = 768;        // queue entries of a region kept in LDS
constexpr int TX_PARK = 4 * 64;   // the parked first-step records of a group: angle, cos, sin, owner_{t-1} per lane
constexpr int TX_GQ_SPEC = 512;   // ... in the speculative round-1 kernel (its lanes' parked states take 2 KB of the wave's LDS)
constexpr int TX_SPEC_CAP = 8;    // pixels a lane may take by itself before its region is handed to the whole wave (< minRegSize)
constexpr int TX_HOT_RESYNC = 2048;   // hot records: pixels after which the filter's sums are replaced by the exact ones (tx_grow_tile)
constexpr int TX_BBLK = 256;      // arena block for the overflow of a large region's queue
constexpr int TX_BMAXBLK = 128;   // => regions of up to TX_GQ + 32768 pixels

__device__ __forceinline__ int2 tx_load_own(const int2* p) {
  // bypass the per-CU L1: claims (atomics) are performed in L2 / memory
#if defined(TX_OWN_LOAD) && TX_OWN_LOAD == 1      // diagnostic build: plain cached load (stale owner words cost regrowth, never exactness)
  return *p;
#elif defined(TX_OWN_LOAD) && TX_OWN_LOAD == 2    // diagnostic build: workgroup-scope load
  unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
  return make_int2((int)(v & 0xFFFFFFFFull), (int)(v >> 32));
#else
  unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
  return make_int2((int)(v & 0xFFFFFFFFull), (int)(v >> 32));
#endif
}
// PACKED ROUND 1 (round 5).  In round 1 owner_0 is the trivial map, so the one owner word a test needs is owner_1 — and a step's
// record and owner fetches are two gathers of the same 3 x 3 blocks from two planes (4.5 + 3.75 sixty-four-byte sectors per octet of
// lanes).  With the CV_64F detector the fourth word of the 16-byte pixel record is free (the gradient norm lives in its own double
// plane): owner_1's start value goes there — written by the front pass as "unclaimed | gradient norm" (lazy ids, key mode) or by
// k_tx_sort as "unclaimed | id" (rank mode) —, round 1 reads ONE 16-byte word per neighbour and claims into it (unsigned atomicMin:
// an unclaimed word is above every id), and k_tx_round2 moves the result into the owner plane.  The load bypasses the per-CU L1
// like tx_load_own (claims are performed in L2).
typedef float tx_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 tx_load_rec16(const float4* p) {
  tx_v4f v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int* tx_rec_owner(const float4* p) { return reinterpret_cast<int*>(const_cast<float4*>(p)) + 3; }
// HOT RECORDS (round 6).  The 16-byte record puts 8 pixels on a 128-byte line; what a round-1 test needs of a neighbour is its level-line
// angle and its owner word.  With keys.hot set the front pass writes those two words to a plane of their own (8 bytes per pixel, 16
// pixels per line: half the footprint of a tile wave's working set), round 1 gathers and claims there, and cos / sin of a candidate
// come from the angle (v_cos_f32 / v_sin_f32) for the vector filter — the exact values of the 16-byte record are summed only where
// the reference's own expression needs them (see growRegion: foldExact).
typedef int tx_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int2 tx_load_hot8(const int2* p) {
  tx_v2i v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_int2(v.x, v.y);
}
__device__ __forceinline__ float tx_hw_cos_deg(float a) { return __builtin_amdgcn_cosf(a * (1.f / 360.f)); }
__device__ __forceinline__ float tx_hw_sin_deg(float a) { return __builtin_amdgcn_sinf(a * (1.f / 360.f)); }
__device__ __forceinline__ int tx_lds_read(const int* p) {
  typedef __attribute__((address_space(3))) const volatile int lds_cvint;
  return *(lds_cvint*)p;
}
__device__ __forceinline__ int tx_rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float tx_rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// k_tx_hot_trig_err (self-test, pli_selftest_hot_trig): the largest |tx_hw_cos_deg(a) - cos(a deg)| and |tx_hw_sin_deg(a) - sin(a deg)| over
// EVERY float a in [0, 360] — the `d` of the error budget in tx_grow_tile.  The level-line angles are fastAtan2 results: floats of that range.
__global__ __launch_bounds__(256) void k_tx_hot_trig_err(unsigned long long* __restrict__ maxErrBits) {
  const unsigned last = 0x43B40000u;                     // 360.0f
  double worst = 0.0;
  for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b <= last; b += (unsigned long long)gridDim.x * 256) {
    const float a = __int_as_float((int)(unsigned)b);
    double sn, cs;
    sincos((double)a * TX_DEG2RAD, &sn, &cs);
    worst = fmax(worst, fmax(fabs((double)tx_hw_cos_deg(a) - cs), fabs((double)tx_hw_sin_deg(a) - sn)));
  }
  unsigned long long w = (unsigned long long)__double_as_longlong(worst);   // (non-negative doubles order like their bits)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(w, o, 64); w = t > w ? t : w; }
  if ((threadIdx.x & 63) == 0) atomicMax(maxErrBits, w);
}

// ---------------------------------------------------------------------------
// k_tx_sort: the seeds of every tile in rank order.  One workgroup of 256 threads per tile sorts the ranks of the tile's
// pixels (bitonic network over ts*ts keys); list entry = (rank, pixel).  Also writes owner_0 = (rank, rank).
// Thread t holds the KPT = ts*ts/256 consecutive keys t*KPT .. t*KPT+KPT-1 in registers: a compare distance j < KPT stays
// inside the thread, j < 64*KPT pairs lanes of one wave (ds_bpermute, no LDS memory, no barrier), and only the 3 of the 78
// stages (ts = 64) whose partner sits in another wave go through LDS — the earlier form did every stage there (6.6 ms at 256
// frames).
// ---------------------------------------------------------------------------
// KEY MODE (keys.mg != null; images of up to 2^20 scaled pixels, CV_64F detector): the id of a region is not its rank in the ordered
// list but the KEY  (nBins - 1 - gradient bin) << pixbits | seed pixel  — the same total order (bins from the strongest gradient
// down, raster order inside a bin), computed here from the gradient norm, so the histogram / scan / scatter passes that build the
// ordered list (4.1 ms of the 56 ms step at 256 frames) do not run at all; per-region planes are indexed by id & rmask = the
// seed pixel.  This kernel then also writes the plane of the pixels' own ids (what k_lsd_scatter writes in rank mode).
// (struct TxKeys: common.hpp)

// The bitonic network over KPT * NT keys, thread t holding the KPT consecutive keys t*KPT .. t*KPT+KPT-1 in registers (ascending
// result): a compare distance j < KPT stays inside the thread, j < 64*KPT pairs lanes of one wave (ds_bpermute, no LDS memory, no
// barrier), and only the stages whose partner sits in another wave go through xch (KPT * NT words of LDS).
// (Round 5, measured at 256 frames: the kernel's 2.3 ms follow neither its VALU count — the compare-exchanges as one v_med3_u32 each
// instead of min + max + select halve it: 2.29 -> 2.30 ms — nor the crossbar: lane exchanges by DPP row shifts and gfx950's
// v_permlane16/32_swap instead of ds_bpermute: 8.1 ms.  Its 7.8 GB of loads and 4-byte row-tiled stores are what it waits for.)
__device__ __forceinline__ unsigned tx_umed3(unsigned a, unsigned b, unsigned c) {      // median of three (v_med3_u32)
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <int KPT, int NT>
__device__ __forceinline__ void tx_bitonic_regs(unsigned (&key)[KPT], unsigned* xch, int tid) {
  constexpr int n2 = KPT * NT;
#pragma unroll
  for (int k = 2; k <= n2; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {             // (fully unrolled: the register array needs constant indices)
      if (j >= KPT) {
        const int pt = j / KPT;                           // partner thread = tid ^ pt
        const bool asc = ((tid * KPT) & k) == 0;           // (k > j >= KPT: the direction depends on the thread only)
        const bool lower = (tid & pt) == 0;
        // (the smaller or the larger of two keys in ONE instruction: the median of the two and 0 resp. UINT_MAX)
        const unsigned sel = (lower == asc) ? 0u : 0xFFFFFFFFu;
        if (pt < 64) {
#pragma unroll
          for (int u = 0; u < KPT; ++u) {
            const unsigned o = (unsigned)__shfl_xor((int)key[u], pt, 64);
            key[u] = tx_umed3(key[u], o, sel);
          }
        } else {
          __syncthreads();
#pragma unroll
          for (int u = 0; u < KPT; ++u) xch[u * NT + tid] = key[u];
          __syncthreads();
#pragma unroll
          for (int u = 0; u < KPT; ++u) {
            const unsigned o = xch[u * NT + (tid ^ pt)];
            key[u] = tx_umed3(key[u], o, sel);
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
          if ((u & j) == 0) {
            const unsigned sel = (((tid * KPT + u) & k) == 0) ? 0u : 0xFFFFFFFFu;      // ascending: the smaller key first
            const unsigned a = key[u], b = key[u | j];
            key[u] = tx_umed3(a, b, sel);
            key[u | j] = tx_umed3(a, b, ~sel);
          }
        }
      }
    }
  }
}

template <int KPT, int NT>
__device__ __forceinline__ void tx_sort_tile(const int* __restrict__ rankAll, const int* __restrict__ orderAll,
                                             int2* __restrict__ ownAll, int2* __restrict__ listAll,
                                             int* __restrict__ tileCntAll, int W, int H, int ts, int ntx, int nty, int img0,
                                             unsigned* xch, const TxKeys& keys) {
  __shared__ int s_cnt;
  const int img = blockIdx.y + img0, tile = blockIdx.x, tid = threadIdx.x;
  constexpr int n2 = KPT * NT;
  const int tx0 = (tile % ntx) * ts, ty0 = (tile / ntx) * ts;
  const int64_t npix = (int64_t)W * H;
  const int* rank = rankAll + img * npix;
  int2* own = ownAll + img * npix;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  // coalesced read of the tile (element i = m*NT + tid), transposed to "KPT consecutive keys per thread" through LDS
  // (padded by one word per 16: the strided side of the transposition is bank-conflict free)
  auto pad = [](int i) -> int { return i + (i >> 4); };
  int valid = 0;
  const double binCoef = keys.mg ? lsd_bin_coef64(keys.maxMg[img], keys.nBins) : 0.0;     // (one division per thread, not one per pixel)
  // (all loads of the thread first — the stores below may alias them for all the compiler knows, and a load that waits for the
  // previous pixel's stores made this phase KPT dependent round trips)
  double vIn[KPT];
  int rIn[KPT];
  // (sort-written ids on the hot records are 4-byte stores into the records' second words.  Rewriting the whole 8-byte record — the
  // angle loaded here with the rest — was measured: k_tx_sort 3.2 -> 3.8 ms at 752 x 480, the sixteen extra loads per thread cost more
  // than the partial sectors)
  // (unconditional, from clamped coordinates: a load under a lane predicate is waited for at the end of its branch)
  if (keys.mg) {
#pragma unroll
    for (int m = 0; m < KPT; ++m) {
      const int i = m * NT + tid;
      vIn[m] = keys.mg[img * npix + min(ty0 + i / ts, H - 1) * W + min(tx0 + i % ts, W - 1)];
      rIn[m] = TX_INF;
    }
  } else {
#pragma unroll
    for (int m = 0; m < KPT; ++m) {
      const int i = m * NT + tid;
      rIn[m] = rank[min(ty0 + i / ts, H - 1) * W + min(tx0 + i % ts, W - 1)];
      vIn[m] = 0.0;
    }
  }
#pragma unroll
  for (int m = 0; m < KPT; ++m) {
    const int i = m * NT + tid;
    const int x = tx0 + i % ts, y = ty0 + i / ts;
    unsigned k = 0xFFFFFFFFu;
    if (x < W && y < H) {
      int r;
      if (keys.mg) {
        const double v = vIn[m];
        r = TX_INF;
        if (!(v <= keys.rho))
          r = ((keys.nBins - 1 - lsd_bin64(v, binCoef, keys.nBins)) << keys.pixbits) | (y * W + x);
        keys.idPlane[img * npix + y * W + x] = r;
      } else {
        r = rIn[m];
      }
      // owner_0 = the trivial map, written for EVERY pixel of the tile (an undefined pixel is nobody's: INT_MAX in both components):
      // the front pass does not initialise the plane for this schedule.  Packed round 1: owner_1's start value into the pixel
      // record instead, with the "unclaimed" bit (k_tx_round2 writes the owner plane); LAZY ids: the front pass has written the word
      // (the two stamp planes of the relaxation — one word per pixel index each — start every call at zero: cleared here, beside the id
      // plane's store, instead of by a fill kernel of their own)
      if (keys.zeroA) keys.zeroA[img * npix + y * W + x] = 0;
      if (keys.zeroB) keys.zeroB[img * npix + y * W + x] = 0;
      if (keys.pack == 1) {
        if (r != TX_INF) {
          if (keys.hot) keys.hot[img * npix + y * W + x].y = (int)(TX_UNCLAIMED | (unsigned)r);
          else *tx_rec_owner(keys.recPack + img * npix + y * W + x) = (int)(TX_UNCLAIMED | (unsigned)r);
        }
      } else if (keys.pack == 0) own[y * W + x] = r != TX_INF ? make_int2(r, r) : make_int2(INT_MAX, INT_MAX);
      if (r != TX_INF) {
        k = (unsigned)r;
        ++valid;
      }
    }
    xch[pad(i)] = k;
  }
  valid = wave_sum_i32(valid);
  if ((tid & 63) == 0 && valid) atomicAdd(&s_cnt, valid);
  __syncthreads();
  unsigned key[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) key[u] = xch[pad(tid * KPT + u)];
  tx_bitonic_regs<KPT, NT>(key, xch, tid);
  int2* list = listAll + ((int64_t)img * ntx * nty + tile) * n2;
  const int* order = orderAll + img * npix;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KPT; ++u) xch[pad(tid * KPT + u)] = key[u];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < KPT; ++m) {
    const int i = m * NT + tid;
    const unsigned k = xch[pad(i)];
    // (key mode: the id's low bits ARE the seed pixel — 4-byte entries, at the start of the tile's slot of the list buffer: half the
    // bytes for this kernel to store and for the growers to fetch)
    if (k != 0xFFFFFFFFu) {
      if (keys.mg) reinterpret_cast<int*>(listAll)[((int64_t)img * ntx * nty + tile) * n2 + i] = (int)k;
      else list[i] = make_int2((int)k, order[k]);
    }
  }
  if (tid == 0) tileCntAll[(int64_t)img * ntx * nty + tile] = s_cnt;
}

__global__ __launch_bounds__(256) void k_tx_sort(const int* __restrict__ rankAll, const int* __restrict__ orderAll,
                                                 int2* __restrict__ ownAll, int2* __restrict__ listAll,
                                                 int* __restrict__ tileCntAll, int W, int H, int ts, int ntx, int nty, int img0, TxKeys keys) {
  __shared__ unsigned xch[16 * 256 + 256];
  if (ts == 64) tx_sort_tile<16, 256>(rankAll, orderAll, ownAll, listAll, tileCntAll, W, H, ts, ntx, nty, img0, xch, keys);
  else if (ts == 32) tx_sort_tile<4, 256>(rankAll, orderAll, ownAll, listAll, tileCntAll, W, H, ts, ntx, nty, img0, xch, keys);
  else tx_sort_tile<1, 256>(rankAll, orderAll, ownAll, listAll, tileCntAll, W, H, ts, ntx, nty, img0, xch, keys);
}
#ifdef PLI_DEV     // (dev switch PLI_TX_TS=128: measured, slower)
// tiles of 128 x 128 (large batches): 1024 threads x 16 keys, 10 of the 105 stages through LDS
__global__ __launch_bounds__(1024) void k_tx_sort128(const int* __restrict__ rankAll, const int* __restrict__ orderAll,
                                                     int2* __restrict__ ownAll, int2* __restrict__ listAll,
                                                     int* __restrict__ tileCntAll, int W, int H, int ts, int ntx, int nty, int img0, TxKeys keys) {
  __shared__ unsigned xch[16 * 1024 + 1024];
  tx_sort_tile<16, 1024>(rankAll, orderAll, ownAll, listAll, tileCntAll, W, H, ts, ntx, nty, img0, xch, keys);
}
#endif

// ---------------------------------------------------------------------------
// k_tx_diff2 (round 2 only; takes the place of k_rx_diff).  Round 1 ran against the trivial owner_0, so "what changed
// since the last run" is not a difference of two owner maps.  What a region saw in round 1 is: the claims of the lower
// ranks of ITS OWN tile for certain (the wave walks them in order), the claims of other tiles' regions maybe.  Region r
// would repeat its round-1 run exactly in round 2 iff no pixel it tested is owned in owner_1 by a lower rank that is
// foreign to r's tile.  Per 8x8 cell: the lowest owner rank whose seed lies outside the cell's tile (k_rx_mark then
// stamps every region with a pixel within one pixel of such a cell and a higher rank); a region whose bounding box
// (+1) leaves its tile is stamped here, and so is a region that ran in round 1 and lost its SEED afterwards (dead in
// owner_1: its remaining pixels are released).  Checked against the sequential result by tools/sim/sim_tile_relax.cpp
// (SIM_CARRY=1), which replays these rules on the CPU.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void tx_mark_dirty(int o, int t, int* __restrict__ rgDirty, const int2* __restrict__ rgBox,
                                              int* __restrict__ tileAct, int TW, int TH, const TxDirtyLists& DL, int img,
                                              bool withBox = true) {
  const int oi = o & DL.rmask;                          // (the region's slot in the per-region planes)
  if (rgDirty[oi] == t) return;
  if (atomicExch(&rgDirty[oi], t) == t) return;
  tx_dirty_append(DL, img, o);
  if (!withBox) return;
  const int2 b = rgBox[oi];
  const int tx0 = min(max((b.x & 0xFFFF) >> 3, 0), TW - 1), ty0 = min(max((b.x >> 16) >> 3, 0), TH - 1);
  const int tx1 = min(max((b.y & 0xFFFF) >> 3, 0), TW - 1), ty1 = min(max((b.y >> 16) >> 3, 0), TH - 1);
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) tileAct[ty * TW + tx] = t;
}

#ifdef PLI_DEV     // (the unfused round 2 of the dev switches PLI_TX_NOFUSE2 / PLI_TX_BOXRULE; the product's round 2 is k_tx_round2)
__global__ __launch_bounds__(256) void k_tx_diff2(RxCtl* __restrict__ ctl, const int2* __restrict__ ownAll,
                                                  const int* __restrict__ orderAll, const int2* __restrict__ rgBoxAll,
                                                  int* __restrict__ rgDirtyAll, int* __restrict__ tileMinAll,
                                                  int* __restrict__ tileActAll, int W, int H, int TW, int TH, int ts, int t,
                                                  int img0, const int* __restrict__ rgLostAll, TxDirtyLists DL) {
  __shared__ int tmin[4];
  const int img = blockIdx.z + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { c.nSmall = 0; c.nBig = 0; c.nHand = 0; c.nextBig = 0; c.rectArena = 0ull; c.changed = 1; }
  if (tid < 4) tmin[tid] = INT_MAX;
  __syncthreads();
  const int x = blockIdx.x * 32 + (tid & 31), y = blockIdx.y * 8 + (tid >> 5);
  const int64_t base = (int64_t)img * W * H;
  if (x < W && y < H) {
    const int2 ow = ownAll[base + y * W + x];
    const int o = (t & 1) ? ow.x : ow.y;               // owner_{t-1}
    if (o != INT_MAX) {
      const int sp = DL.rmask != -1 ? (o & DL.rmask) : orderAll[base + o];
      const int sx = sp % W, sy = sp / W;
      if (!rgLostAll && (sx / ts != x / ts || sy / ts != y / ts)) atomicMin(&tmin[(tid & 31) >> 3], o);
      if (sp != y * W + x) {
        // the owner ran in round 1 and lost its seed afterwards (to a lower rank of another tile): it is dead in owner_1, its
        // remaining pixels must be released (the later rounds do this with k_rx_mark's "seed changed hands" rule)
        const int2 so = ownAll[base + sp];
        if (((t & 1) ? so.x : so.y) != o) tx_mark_dirty(o, t, rgDirtyAll + base, rgBoxAll + base, tileActAll + (int64_t)img * TW * TH, TW, TH, DL, img);
      } else if (rgLostAll) {
        // the seed of an alive region.  Exact rule (rgLostAll): its round-1 run stands unless it LOST a pixel it claimed — a lower
        // rank claimed the pixel too; the growers note the loser of every such claim.  (Why nothing else: take the lowest-ranked
        // region whose run differs from the sequential one; everything below it is final, so a pixel it wrongly took is claimed
        // by its rightful lower owner as well, and a pixel it was wrongly refused would have to be held by a lower region that
        // does not hold it in the end — which is then a change of the later rounds' kind.  tools/sim/sim_tile_relax.cpp replays
        // the rule: SIM_CARRY=1 SIM_LOST=1.)
        if (rgLostAll[base + (o & DL.rmask)] != 0)
          tx_mark_dirty(o, t, rgDirtyAll + base, rgBoxAll + base, tileActAll + (int64_t)img * TW * TH, TW, TH, DL, img);
      } else {                                          // conservative rule: does its box (+1) leave the tile?
        const int2 b = rgBoxAll[base + (o & DL.rmask)];
        const int tx0 = (x / ts) * ts, ty0 = (y / ts) * ts;
        if ((b.x & 0xFFFF) - 1 < tx0 || (b.x >> 16) - 1 < ty0 || (b.y & 0xFFFF) + 1 >= tx0 + ts || (b.y >> 16) + 1 >= ty0 + ts)
          tx_mark_dirty(o, t, rgDirtyAll + base, rgBoxAll + base, tileActAll + (int64_t)img * TW * TH, TW, TH, DL, img);
      }
    }
  }
  __syncthreads();
  if (tid < 4) {
    const int tx = blockIdx.x * 4 + tid;
    if (tx < TW) {
      tileMinAll[(int64_t)img * TW * TH + blockIdx.y * TW + tx] = tmin[tid];
      if (tmin[tid] != INT_MAX) tileActAll[(int64_t)img * TW * TH + blockIdx.y * TW + tx] = t;
    }
  }
}
#endif

// ---------------------------------------------------------------------------
// k_tx_round2 (round 2, lost-pixel rule): k_tx_diff2 and the round's k_tx_prep in one pass over the owner map.  Whether the
// owner o of a pixel is regrown is decided where the pixel is — o is dead in owner_1 (its seed belongs to another region), or
// it lost a contested claim in round 1 (rgLost) — so the pixel can take its start value for owner_2 at once: o when o is
// carried, its own rank otherwise.  The first pixel that finds o stamps it (rgDirty, the cells under its box, the dirty list
// of its tile).  32 x 32 pixels per block, the four rows of a thread in flight together.
// ---------------------------------------------------------------------------
template <bool PACKED>
__device__ __forceinline__ void tx_round2_block(RxCtl* __restrict__ ctl, int2* __restrict__ ownAll,
                                                   const int* __restrict__ rankAll, const int* __restrict__ orderAll,
                                                   const int2* __restrict__ rgBoxAll, int* __restrict__ rgDirtyAll,
                                                   int* __restrict__ tileActAll, int W, int H, int TW, int TH, int t, int img0,
                                                   const int* __restrict__ rgLostAll, TxDirtyLists DL,
                                                   int* __restrict__ tileTouchAll, const float4* __restrict__ recPack, int keepRect,
                                                   const int2* __restrict__ hotPack) {
  const int img = blockIdx.z + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  const int tid = threadIdx.x;
  // (round 1's owner words: the fourth word of the 16-byte records, or the second of the 8-byte hot records)
  auto packWord = [&](int64_t p) -> int { return hotPack ? hotPack[p].y : *tx_rec_owner(recPack + p); };
  // (keepRect: round 1's region2rect pass runs beside this kernel and still reads the round's list counter: k_tx_reset_rect clears it later)
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    c.nSmall = 0; c.nBig = 0; c.nHand = 0; c.nextBig = 0; c.changed = 0; c.changedOdd = 0;
    if (!keepRect) c.rectArena = 0ull;
  }
  // (packed round 1: owner_1 comes from the pixel records and the owner plane is WRITTEN here, both components, every pixel; t == 2)
  const int64_t base = (int64_t)img * W * H;
  const int ci = t & 1;
  const int x = blockIdx.x * 32 + (tid & 31);
  int2 ow[4];
  int o[4], r[4], sp[4], lost[4];
  int2 so[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = blockIdx.y * 32 + i * 8 + (tid >> 5);
    ow[i] = make_int2(INT_MAX, INT_MAX);
    r[i] = TX_INF;
    if (x < W && y < H) {
      if (PACKED) ow[i].y = packWord(base + y * W + x);
      else ow[i] = ownAll[base + y * W + x];
      r[i] = rankAll[base + y * W + x];
    }
    // (packed: an unclaimed word — bit 31 — stands for the pixel's own id; an undefined pixel is nobody's)
    if (PACKED) ow[i].y = r[i] == TX_INF ? INT_MAX : (ow[i].y < 0 ? r[i] : ow[i].y);
    o[i] = ci ? ow[i].x : ow[i].y;                     // owner_{t-1}
  }
  // (the gathers of the four rows unconditional, from an index that is valid either way — slot 0 for a pixel without an owner —, so
  // that they are in flight together: a load under a lane predicate is waited for at the end of its branch)
  if (DL.rmask != -1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sp[i] = o[i] != INT_MAX ? (o[i] & DL.rmask) : -1;
      lost[i] = rgLostAll[base + max(sp[i], 0)];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sp[i] = -1; lost[i] = 0;
      if (o[i] != INT_MAX) {
        sp[i] = orderAll[base + o[i]];
        lost[i] = rgLostAll[base + (o[i] & DL.rmask)];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (PACKED) so[i] = make_int2(0, packWord(base + max(sp[i], 0)));
    else so[i] = ownAll[base + max(sp[i], 0)];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (sp[i] < 0) { so[i] = make_int2(0, 0); lost[i] = 0; }
    else if (PACKED && so[i].y < 0) so[i].y = o[i];    // (sp is o's seed pixel: unclaimed means o holds it)
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = blockIdx.y * 32 + i * 8 + (tid >> 5);
    const int p = y * W + x;
    if (o[i] == INT_MAX) {
      if (PACKED && x < W && y < H) ownAll[base + p] = make_int2(INT_MAX, INT_MAX);     // (an undefined pixel)
      continue;
    }
    // (p is o's seed: o holds it, so o is alive)
    const bool dead = sp[i] != p && (ci ? so[i].x : so[i].y) != o[i];
    const bool dirty = dead || lost[i] != 0;
    if (dirty) tx_mark_dirty(o[i], t, rgDirtyAll + base, rgBoxAll + base, tileActAll + (int64_t)img * TW * TH, TW, TH, DL, img);
    // (a pixel that falls back to its own rank makes its cell differ between owner_1 and owner_2: noted for round 3's diff)
    if (dirty && tileTouchAll) tileTouchAll[(int64_t)img * TW * TH + (y >> 3) * TW + (x >> 3)] = t;
    const int cur = dirty ? r[i] : o[i];
    if (PACKED || cur != (ci ? ow[i].y : ow[i].x)) {
      if (ci) ow[i].y = cur; else ow[i].x = cur;
      ownAll[base + p] = ow[i];
    }
  }
}

// (two instances: a test of the argument inside the unrolled row loops keeps the four rows' loads from being issued together)
__global__ __launch_bounds__(256) void k_tx_round2(RxCtl* __restrict__ ctl, int2* __restrict__ ownAll,
                                                   const int* __restrict__ rankAll, const int* __restrict__ orderAll,
                                                   const int2* __restrict__ rgBoxAll, int* __restrict__ rgDirtyAll,
                                                   int* __restrict__ tileActAll, int W, int H, int TW, int TH, int t, int img0,
                                                   const int* __restrict__ rgLostAll, TxDirtyLists DL,
                                                   int* __restrict__ tileTouchAll, const float4* __restrict__ recPack, int keepRect,
                                                   const int2* __restrict__ hotPack) {
  if (recPack || hotPack) tx_round2_block<true>(ctl, ownAll, rankAll, orderAll, rgBoxAll, rgDirtyAll, tileActAll, W, H, TW, TH, t, img0, rgLostAll, DL, tileTouchAll, recPack, keepRect, hotPack);
  else tx_round2_block<false>(ctl, ownAll, rankAll, orderAll, rgBoxAll, rgDirtyAll, tileActAll, W, H, TW, TH, t, img0, rgLostAll, DL, tileTouchAll, recPack, keepRect, hotPack);
}
// (after round 1's region2rect pass, when it ran beside k_tx_round2: the list counter of the images that are still relaxing starts round 2 at zero)
__global__ __launch_bounds__(256) void k_tx_reset_rect(RxCtl* __restrict__ ctl, int nimg, int img0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nimg && ctl[img0 + i].state != 2) ctl[img0 + i].rectArena = 0ull;
}

// ---------------------------------------------------------------------------
// k_tx_mark (rounds >= 3, exact rule; takes the place of k_rx_mark, whose comment states the rule).  Everything the rule does
// starts from a pixel whose owner changed between owner_{t-2} and owner_{t-1}, and those are few: a block (4 x 8 cells) first
// lists the changed pixels of its changed cells in LDS, then spreads (changed pixel, neighbour) pairs over its threads, so the
// dependent chain neighbour owner -> rgLost -> stamp runs once per block, not once per row of cells.
// ---------------------------------------------------------------------------
#ifdef PLI_DEV     // (the unfused marks of the dev switch PLI_TX_NOFUSEDM; the product runs k_tx_diffmark / the cell lists)
__global__ __launch_bounds__(256) void k_tx_mark(RxCtl* __restrict__ ctl, const int2* __restrict__ ownAll,
                                                 const int* __restrict__ rankAll, const int2* __restrict__ rgBoxAll,
                                                 int* __restrict__ rgDirtyAll, const int* __restrict__ tileMinAll,
                                                 int* __restrict__ tileActAll, int W, int H, int TW, int TH, int t, int img0,
                                                 const int* __restrict__ rgLostAll, TxDirtyLists DL) {
  __shared__ int s_chg[8][4];
  __shared__ int s_n;
  __shared__ unsigned short lst[2048];
  const int img = blockIdx.z + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  if (c.changed == 0) {                              // owner_{t-1} == owner_{t-2}: exact (every block sees the same flag)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { c.state = 2; c.rounds = t; }
    return;
  }
  const int tid = threadIdx.x;
  if (tid == 0) s_n = 0;
  if (tid < 32) {
    const int ty = (int)blockIdx.y * 8 + (tid >> 2), tx = (int)blockIdx.x * 4 + (tid & 3);
    s_chg[tid >> 2][tid & 3] = (ty < TH && tx < TW) ? tileMinAll[(int64_t)img * TW * TH + ty * TW + tx] != INT_MAX : 0;
  }
  __syncthreads();
  const int64_t base = (int64_t)img * W * H;
  const int lx = tid & 31, ly = tid >> 5;
  const int x = blockIdx.x * 32 + lx;
  int2 o[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int y = (blockIdx.y * 8 + rr) * 8 + ly;
    o[rr] = make_int2(0, 0);
    if (s_chg[rr][lx >> 3] && x < W && y < H) o[rr] = ownAll[base + y * W + x];
  }
#pragma unroll
  for (int rr = 0; rr < 8; ++rr)
    if (o[rr].x != o[rr].y) lst[atomicAdd(&s_n, 1)] = (unsigned short)(((rr * 8 + ly) << 5) | lx);
  __syncthreads();
  const int n = s_n;
  if (n == 0) return;
  int* rgDirty = rgDirtyAll + base;
  const int2* rgBox = rgBoxAll + base;
  int* tileAct = tileActAll + (int64_t)img * TW * TH;
  const int ci = t & 1;
  for (int i = tid; i < n * 9; i += 256) {
    const int e = (int)(((unsigned)i * 7282u) >> 16);   // i / 9 for i < 2048 * 9
    const int k = i - 9 * e;
    const int li = lst[e];
    const int px0 = blockIdx.x * 32 + (li & 31), py0 = blockIdx.y * 64 + (li >> 5);
    const int2 oc = ownAll[base + py0 * W + px0];
    const int prevv = ci ? oc.x : oc.y, prev2 = ci ? oc.y : oc.x;
    if (k == 4) {
      // the seed of a region changed hands: it died (its last box is released) or is newly alive (only this pixel)
      const int r = rankAll[base + py0 * W + px0];
      if (r == TX_INF) continue;
      const bool a1 = prevv == r, a2 = prev2 == r;
      if (a1 != a2) tx_mark_dirty(r, t, rgDirty, rgBox, tileAct, TW, TH, DL, img, a2);
    } else {
      const int px = px0 + k % 3 - 1, py = py0 + k / 3 - 1;
      if (px < 0 || py < 0 || px >= W || py >= H) continue;
      const int2 op2 = ownAll[base + py * W + px];
      const int op = ci ? op2.x : op2.y;               // owner_{t-1} of the neighbour
      if (op == INT_MAX) continue;
      if ((prev2 < op && prevv > op) || (prev2 == op && prevv < op) || rgLostAll[base + (op & DL.rmask)] == t - 1)
        tx_mark_dirty(op, t, rgDirty, rgBox, tileAct, TW, TH, DL, img);
    }
  }
}

#endif
// ---------------------------------------------------------------------------
// k_tx_diffmark (rounds >= 3): k_rx_diff and k_tx_mark in one pass.  The cells whose two owner components can differ are the
// ones tileTouch stamps with t-1 — the growers' claims of round t-1 and the cells k_tx_prep rewrote in round t-1 (it stamps
// them too in this mode) — so a block reads its touched cells once, lists their changed pixels and goes on to the rule.
// "Nothing changed anywhere" is known only when the kernel ends: the round's k_tx_prep tests the flag (one flag per round
// parity: it clears the other one for the next round) and declares the fixed point.
// ---------------------------------------------------------------------------
// (one block's share: the 4 x 8 cells at (bx, by) of image img; a kernel of its own — k_tx_diffmark — and a phase of k_tx_tail)
__device__ __forceinline__ void tx_diffmark_block(RxCtl* ctl, const int2* ownAll, const int* rankAll, const int2* rgBoxAll,
                                                  int* rgDirtyAll, int* tileActAll, const int* tileTouchAll, int W, int H, int TW,
                                                  int TH, int t, int bx, int by, int img, const int* rgLostAll,
                                                  const TxDirtyLists& DL) {
  __shared__ int s_rel[8][4];
  __shared__ int s_chg[8][4];
  __shared__ int s_n;
  __shared__ unsigned short lst[2048];
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  const int tid = threadIdx.x;
  if (bx == 0 && by == 0 && tid == 0) { c.nSmall = 0; c.nBig = 0; c.nHand = 0; c.nextBig = 0; c.rectArena = 0ull; }
  if (tid == 0) s_n = 0;
  if (tid < 32) {
    const int ty = by * 8 + (tid >> 2), tx = bx * 4 + (tid & 3);
    s_rel[tid >> 2][tid & 3] = (ty < TH && tx < TW) ? tileTouchAll[(int64_t)img * TW * TH + ty * TW + tx] == t - 1 : 0;
    s_chg[tid >> 2][tid & 3] = 0;
  }
  __syncthreads();
  const int64_t base = (int64_t)img * W * H;
  const int lx = tid & 31, ly = tid >> 5;
  const int x = bx * 32 + lx;
  int2 o[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int y = (by * 8 + rr) * 8 + ly;
    o[rr] = make_int2(0, 0);
    if (s_rel[rr][lx >> 3] && x < W && y < H) o[rr] = ownAll[base + y * W + x];
  }
#pragma unroll
  for (int rr = 0; rr < 8; ++rr)
    if (o[rr].x != o[rr].y) {
      lst[atomicAdd(&s_n, 1)] = (unsigned short)(((rr * 8 + ly) << 5) | lx);
      s_chg[rr][lx >> 3] = 1;                           // (benign race: every writer stores 1)
    }
  __syncthreads();
  const int n = s_n;
  if (n == 0) return;
  int* tileAct = tileActAll + (int64_t)img * TW * TH;
  if (tid < 32 && s_chg[tid >> 2][tid & 3]) {
    const int ty = by * 8 + (tid >> 2), tx = bx * 4 + (tid & 3);
    tileAct[ty * TW + tx] = t;                          // a changed cell is active
  }
  if (tid == 0) atomicOr((t & 1) ? &c.changedOdd : &c.changed, 1);
  int* rgDirty = rgDirtyAll + base;
  const int2* rgBox = rgBoxAll + base;
  const int ci = t & 1;
  for (int i = tid; i < n * 9; i += 256) {
    const int e = (int)(((unsigned)i * 7282u) >> 16);   // i / 9 for i < 2048 * 9
    const int k = i - 9 * e;
    const int li = lst[e];
    const int px0 = bx * 32 + (li & 31), py0 = by * 64 + (li >> 5);
    const int2 oc = ownAll[base + py0 * W + px0];
    const int prevv = ci ? oc.x : oc.y, prev2 = ci ? oc.y : oc.x;
    if (k == 4) {
      const int r = rankAll[base + py0 * W + px0];
      if (r == TX_INF) continue;
      const bool a1 = prevv == r, a2 = prev2 == r;
      if (a1 != a2) tx_mark_dirty(r, t, rgDirty, rgBox, tileAct, TW, TH, DL, img, a2);
    } else {
      const int px = px0 + k % 3 - 1, py = py0 + k / 3 - 1;
      if (px < 0 || py < 0 || px >= W || py >= H) continue;
      const int2 op2 = ownAll[base + py * W + px];
      const int op = ci ? op2.x : op2.y;               // owner_{t-1} of the neighbour
      if (op == INT_MAX) continue;
      if ((prev2 < op && prevv > op) || (prev2 == op && prevv < op) || rgLostAll[base + (op & DL.rmask)] == t - 1)
        tx_mark_dirty(op, t, rgDirty, rgBox, tileAct, TW, TH, DL, img);
    }
  }
}

__global__ __launch_bounds__(256) void k_tx_diffmark(RxCtl* __restrict__ ctl, const int2* __restrict__ ownAll,
                                                     const int* __restrict__ rankAll, const int2* __restrict__ rgBoxAll,
                                                     int* __restrict__ rgDirtyAll, int* __restrict__ tileActAll,
                                                     const int* __restrict__ tileTouchAll, int W, int H, int TW, int TH, int t,
                                                     int img0, const int* __restrict__ rgLostAll, TxDirtyLists DL) {
  tx_diffmark_block(ctl, ownAll, rankAll, rgBoxAll, rgDirtyAll, tileActAll, tileTouchAll, W, H, TW, TH, t, blockIdx.x, blockIdx.y,
                    blockIdx.z + img0, rgLostAll, DL);
}

// ---------------------------------------------------------------------------
// k_tx_prep (rounds >= 2, after the diff and k_rx_mark): owner_t is rewritten in the active 8x8 cells only:
// a pixel whose previous owner is carried (not stamped dirty) stays with it, any other falls back to its own
// rank.  Elsewhere owner_{t-2} == owner_{t-1} and the owner is carried: the word that is there is right.
// ---------------------------------------------------------------------------
// (256 threads walk the 32 x 32 pixels of a block in four steps: most blocks leave after the activity test, and the cost of
// that is per wave)
__device__ __forceinline__ void tx_prep_block(RxCtl* ctl, int2* ownAll, const int* rankAll, const int* rgDirtyAll,
                                              const int* tileActAll, int W, int H, int TW, int TH, int t, int bx, int by, int img,
                                              int full, int* tileTouchAll, int rmask) {
  __shared__ int s_act;
  __shared__ int s_cell[4][4];
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  const int tid = threadIdx.x;
  if (tileTouchAll) {
    // after k_tx_diffmark: the round's flag decides (every block reads the same value: nobody writes it in this kernel)
    if (((t & 1) ? c.changedOdd : c.changed) == 0) {
      if (bx == 0 && by == 0 && tid == 0) { c.state = 2; c.rounds = t; }
      return;
    }
    if (bx == 0 && by == 0 && tid == 0) { if (t & 1) c.changed = 0; else c.changedOdd = 0; }
  } else if (bx == 0 && by == 0 && tid == 0) c.changed = 0;
  if (tid == 0) s_act = full;                          // round 2: owner_t still holds the trivial map, every cell is rewritten
  __syncthreads();
  if (tid < 16) {
    const int tx = bx * 4 + (tid & 3), ty = by * 4 + (tid >> 2);
    const int a = full || (tx < TW && ty < TH && tileActAll[(int64_t)img * TW * TH + ty * TW + tx] == t);
    s_cell[tid >> 2][tid & 3] = a;
    if (a) s_act = 1;
    if (a && tileTouchAll && tx < TW && ty < TH) tileTouchAll[(int64_t)img * TW * TH + ty * TW + tx] = t;   // (this cell is rewritten)
  }
  __syncthreads();
  if (!s_act) return;
  const int64_t base = (int64_t)img * W * H;
  const int ci = t & 1;
  const int x = bx * 32 + (tid & 31);
  // (the loads of the four rows first, then the dependent stamps, then the stores: a store between them would order them)
  int r[4], dirtyAt[4];
  int2 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = by * 32 + i * 8 + (tid >> 5);
    r[i] = TX_INF;
    o[i] = make_int2(0, 0);
    // (a cell that is not active keeps its words: they are right)
    if (x < W && y < H && s_cell[i][(tid & 31) >> 3]) {
      r[i] = rankAll[base + y * W + x];
      o[i] = ownAll[base + y * W + x];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) dirtyAt[i] = r[i] != TX_INF ? rgDirtyAll[base + ((ci ? o[i].x : o[i].y) & rmask)] : 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (r[i] == TX_INF) continue;
    const int y = by * 32 + i * 8 + (tid >> 5);
    const int prevv = ci ? o[i].x : o[i].y;
    const int cur = dirtyAt[i] != t ? prevv : r[i];
    if (cur != (ci ? o[i].y : o[i].x)) {
      if (ci) o[i].y = cur; else o[i].x = cur;
      ownAll[base + y * W + x] = o[i];
    }
  }
}

__global__ __launch_bounds__(256) void k_tx_prep(RxCtl* __restrict__ ctl, int2* __restrict__ ownAll,
                                                 const int* __restrict__ rankAll, const int* __restrict__ rgDirtyAll,
                                                 const int* __restrict__ tileActAll, int W, int H, int TW, int TH, int t, int img0,
                                                 int full, int* __restrict__ tileTouchAll, int rmask) {
  tx_prep_block(ctl, ownAll, rankAll, rgDirtyAll, tileActAll, W, H, TW, TH, t, blockIdx.x, blockIdx.y, blockIdx.z + img0, full, tileTouchAll, rmask);
}

// ---------------------------------------------------------------------------
// Rounds >= 3 of large batches, CELL LISTS (round 5).  k_tx_diffmark and k_tx_prep walk every block of every image and let most
// of them leave after a look at their cells' stamps; from round 3 on a round touches a few per cent of the cells, and what the
// 133 000 + 267 000 workgroups of a 256-frame batch cost is the dependent chain "stamps -> barrier -> pixels -> barrier" of the many
// that find one or two cells (rounds 3 to 7: 3.5 ms).  Here the cells with work are LISTED first (a streaming pass over the stamp
// plane, eight cells per thread; the per-image control updates of the two kernels ride on it) and one WAVE per listed cell does the
// cell's 64 pixels — lane = pixel, no LDS beyond 64 bytes, no block barrier:
//   k_tx_cells(mode 0) -> k_tx_diffmark_cells -> k_tx_cells(mode 1) -> k_tx_prep_cells -> k_tx_grow_sparse -> k_rx_rect.
// Same marks, same rewritten cells as the block kernels (which stay: small batches — fewer launches —, k_tx_tail, dev switches).
// ---------------------------------------------------------------------------
// mode 0 (before the diff of round t): cells touched in round t - 1; resets the per-round counters of the image (what block (0, 0)
// of k_tx_diffmark does).  mode 1 (after it): cells active in round t, stamped as touched (k_tx_prep rewrites them); declares the
// fixed point of an image whose round-t flag stayed clear and clears the other round's flag (block (0, 0) of k_tx_prep).
// (256 threads, 8 KB of LDS: the kernel runs while the ORB chain's workgroups fill the CUs, and a 1024-thread workgroup waits for
// sixteen wave slots of ONE CU to be free at once — 0.5 ms a launch in the default line against 0.017 alone)
constexpr int TX_CELL_CHUNK = 2048;      // cells a workgroup of k_tx_cells lists (one atomic on the image's counter per workgroup)
__global__ __launch_bounds__(256) void k_tx_cells(RxCtl* __restrict__ ctl, const int* __restrict__ stampAll, int* __restrict__ tileTouchAll,
                                                   int ncell, int nimg, int img0, int t, int mode, int* __restrict__ list, int* __restrict__ cnt) {
  // A workgroup per (chunk of 2048 cells, image): its cells with work are compacted in order (a prefix over the lanes and the 4 waves)
  // and take their places in the image's list with ONE atomic (752 x 480: four workgroups per image; one counter for the whole batch
  // serialised 65 000 wave atomics on one address, 0.5 ms a launch).  The counters — per image, round and mode — are zeroed once
  // per call with the control blocks.
  __shared__ int wsum[4];
  __shared__ int s_base;
  __shared__ int buf[TX_CELL_CHUNK];
  const int il = blockIdx.y, img = img0 + il, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  if (mode == 0) {
    if (blockIdx.x == 0 && tid == 0) { c.nSmall = 0; c.nBig = 0; c.nHand = 0; c.nextBig = 0; c.rectArena = 0ull; }
  } else {
    // (nobody writes THIS round's flag in this kernel, and `state` only moves to 2 for an image whose flag is clear: every workgroup
    // of the image decides the same way whatever it sees of the first one's stores)
    const bool settled = ((t & 1) ? c.changedOdd : c.changed) == 0;
    if (blockIdx.x == 0 && tid == 0) {
      if (settled) { c.state = 2; c.rounds = t; }
      else if (t & 1) c.changed = 0;
      else c.changedOdd = 0;
    }
    if (settled) return;
  }
  // every thread looks at 8 consecutive cells (their stamps are in flight together), the counts are scanned once over the workgroup
  const int* stamp = stampAll + (int64_t)img * ncell;
  const int want0 = mode == 0 ? t - 1 : t;
  const int c0 = blockIdx.x * TX_CELL_CHUNK + tid * 8;
  int st[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) st[u] = c0 + u < ncell ? stamp[c0 + u] : want0 - 1;
  unsigned m = 0u;
#pragma unroll
  for (int u = 0; u < 8; ++u) m |= (st[u] == want0 ? 1u : 0u) << u;
  const int mine = __popc(m);
  int incl = mine;                                       // inclusive prefix over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int before = incl - mine, total = 0;
  for (int w = 0; w < 4; ++w) {
    const int v = wsum[w];
    if (w < wv) before += v;
    total += v;
  }
  if (total == 0) return;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if ((m >> u) & 1u) {
      buf[before++] = c0 + u;
      if (mode == 1) tileTouchAll[(int64_t)img * ncell + c0 + u] = t;          // (this cell is rewritten)
    }
  if (tid == 0) s_base = atomicAdd(&cnt[il], total);
  __syncthreads();
  int* out = list + (int64_t)il * ncell + s_base;
  for (int i = tid; i < total; i += 256) out[i] = buf[i];
}

// one wave per listed cell (four cells at a time: their pixels' owner words are fetched together): the changed pixels of a cell
// (owner_{t-1} != owner_{t-2}) and the rule of k_tx_mark on their neighbours
// (single-wave workgroups: the waves do not talk to each other, and the kernel runs beside the ORB chain — the smaller the workgroup,
// the sooner it finds its wave slots)
__global__ __launch_bounds__(64) void k_tx_diffmark_cells(RxCtl* __restrict__ ctl, const int2* __restrict__ ownAll, const int* __restrict__ rankAll,
                                                           const int2* __restrict__ rgBoxAll, int* __restrict__ rgDirtyAll,
                                                           int* __restrict__ tileActAll, int W, int H, int TW, int TH, int t, int img0,
                                                           const int* __restrict__ rgLostAll, TxDirtyLists DL, const int* __restrict__ list,
                                                           const int* __restrict__ cnt) {
  constexpr int U = 4;
  __shared__ unsigned char slots[1][64];
  const int lane = threadIdx.x & 63, wv = 0;
  const int il = blockIdx.y, ncell = TW * TH;
  const int n = cnt[il];
  const int ci = t & 1;
  const int img = img0 + il;
  const int64_t base = (int64_t)img * W * H;
  RxCtl& c = ctl[img];
  int* tileAct = tileActAll + (int64_t)img * ncell;
  int* rgDirty = rgDirtyAll + base;
  const int2* rgBox = rgBoxAll + base;
  const int* lst = list + (int64_t)il * ncell;
  for (int i0 = blockIdx.x * U; i0 < n; i0 += gridDim.x * U) {
    int cells[U];
    int2 ow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cells[u] = i0 + u < n ? lst[i0 + u] : -1;
      ow[u] = make_int2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (cells[u] < 0) continue;
      const int x = (cells[u] % TW) * 8 + (lane & 7), y = (cells[u] / TW) * 8 + (lane >> 3);
      if (x < W && y < H) ow[u] = ownAll[base + y * W + x];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int2 o = ow[u];
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(o.x != o.y);
      if (bal == 0ull) continue;
      const int cell = cells[u], cx = cell % TW, cy = cell / TW;
      if (lane == 0) {
        tileAct[cell] = t;                                // a changed cell is active
        int* flag = (t & 1) ? &c.changedOdd : &c.changed;
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
      }
      // the (changed pixel, neighbour) pairs of the cell, 64 at a time: pair j = (the (j / 9)-th changed pixel, position j % 9 of its 3 x 3 block)
      if ((bal >> lane) & 1ull) slots[wv][__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned char)lane;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int npairs = 9 * __popcll(bal);
      for (int j0 = 0; j0 < npairs; j0 += 64) {
        const int j = j0 + lane;
        const int e = min(j, npairs - 1) / 9, k = j - 9 * (j / 9);
        const int src = slots[wv][e];
        const int ox = __shfl(o.x, src, 64), oy = __shfl(o.y, src, 64);
        if (j >= npairs) continue;
        const int px0 = cx * 8 + (src & 7), py0 = cy * 8 + (src >> 3);
        const int prevv = ci ? ox : oy, prev2 = ci ? oy : ox;
        if (k == 4) {
          // the seed of a region changed hands: it died (its last box is released) or is newly alive (only this pixel)
          const int r = rankAll[base + py0 * W + px0];
          if (r == TX_INF) continue;
          const bool a1 = prevv == r, a2 = prev2 == r;
          if (a1 != a2) tx_mark_dirty(r, t, rgDirty, rgBox, tileAct, TW, TH, DL, img, a2);
        } else {
          const int px = px0 + k % 3 - 1, py = py0 + k / 3 - 1;
          if (px < 0 || py < 0 || px >= W || py >= H) continue;
          const int2 op2 = ownAll[base + py * W + px];
          const int op = ci ? op2.x : op2.y;             // owner_{t-1} of the neighbour
          if (op == INT_MAX) continue;
          if ((prev2 < op && prevv > op) || (prev2 == op && prevv < op) || rgLostAll[base + (op & DL.rmask)] == t - 1)
            tx_mark_dirty(op, t, rgDirty, rgBox, tileAct, TW, TH, DL, img);
        }
      }
      __builtin_amdgcn_wave_barrier();                   // (the slots are reused by the wave's next cell)
    }
  }
}

// one wave per listed cell, four cells at a time: owner_t of the cells' pixels (k_tx_prep's rule)
__global__ __launch_bounds__(64) void k_tx_prep_cells(const RxCtl* __restrict__ ctl, int2* __restrict__ ownAll, const int* __restrict__ rankAll,
                                                       const int* __restrict__ rgDirtyAll, int W, int H, int TW, int TH, int t, int img0, int rmask,
                                                       const int* __restrict__ list, const int* __restrict__ cnt) {
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const int il = blockIdx.y, ncell = TW * TH;
  const int n = cnt[il];
  const int ci = t & 1;
  const int img = img0 + il;
  const int64_t base = (int64_t)img * W * H;
  const int* lst = list + (int64_t)il * ncell;
  for (int i0 = blockIdx.x * U; i0 < n; i0 += gridDim.x * U) {
    int p[U], r[U], dirtyAt[U];
    int2 o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cell = i0 + u < n ? lst[i0 + u] : -1;
      const int x = cell < 0 ? W : (cell % TW) * 8 + (lane & 7), y = cell < 0 ? H : (cell / TW) * 8 + (lane >> 3);
      p[u] = (x < W && y < H) ? y * W + x : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      r[u] = TX_INF;
      o[u] = make_int2(0, 0);
      if (p[u] >= 0) {
        r[u] = rankAll[base + p[u]];
        o[u] = ownAll[base + p[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) dirtyAt[u] = r[u] != TX_INF ? rgDirtyAll[base + ((ci ? o[u].x : o[u].y) & rmask)] : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r[u] == TX_INF) continue;
      const int prevv = ci ? o[u].x : o[u].y;
      const int cur = dirtyAt[u] != t ? prevv : r[u];
      if (cur != (ci ? o[u].y : o[u].x)) {
        if (ci) o[u].y = cur; else o[u].x = cur;
        ownAll[base + p[u]] = o[u];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_tx_grow: one wave per tile walks the tile's seeds in rank order.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// The accept loop of a batched step, hand-scheduled like lsd_accept_fast (line_kernels.hip): half of the round-1 grower's vector
// instructions are issued here (an accepted pixel costs 20 VALU + 14 SALU; 413 000 accepted pixels per 752 x 480 image), and the
// kernel's time follows its instruction count at about half weight (DESIGN.md 5 "Round 5").
// State: the float sums, `remaining` (candidates not yet decided; lane order = test order), `acc` (lanes accepted in this
// batch, in increasing lane order: a lane's queue slot is cnt0 + its rank in acc), `seeds` (live seeds of the current list
// row: a seed whose pixel is taken leaves it), cnt, and the bounding box as packed 16-bit (y, x) minima / maxima.
// Return 0: no candidate left; 1: lane j2 lies inside the margin of the vector filter and needs the exact test (`remaining`
// already without the lanes up to j2).
// ---------------------------------------------------------------------------
// Diagnostic builds (make EXTRA=-DTX_DIAG_PAD=n, never shipped): n x 4 idle VALU instructions (or, negative, |n| x 4 idle SALU
// instructions) per iteration of the accept loop (at its top, where no hazard slot hides them) — how the kernel's time follows its instruction count (NOTEBOOK.md "R4-5").
#ifndef TX_DIAG_PAD
#define TX_DIAG_PAD_ASM
#elif TX_DIAG_PAD == 1
#define TX_DIAG_PAD_ASM "v_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\t"
#elif TX_DIAG_PAD == 2
#define TX_DIAG_PAD_ASM "v_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\t" \
                        "v_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\tv_mov_b32 %[t2], %[t2]\n\t"
#elif TX_DIAG_PAD == -1
#define TX_DIAG_PAD_ASM "s_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\t"
#elif TX_DIAG_PAD == -2
#define TX_DIAG_PAD_ASM "s_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\t" \
                        "s_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\ts_mov_b32 %[code], %[code]\n\t"
#endif
// A pixel is identified by its packed (y << 16 | x) throughout (candidates `myxy`, the row's seeds `seedXY`): one v_readlane serves
// the duplicate test, the seed test and the bounding box.  (Lanes that are not candidates may hold anything there — an
// out-of-image neighbour packs to a negative value, a lane past the step's entries to some pixel: they are not in `remaining` /
// `seeds`, so a match on them clears nothing.)
// `grp` / grpXY: the candidates of the current group's parked first steps and their pixels — an accepted pixel leaves that mask as it
// leaves `seeds` (the members that come later must not take what a lower id of this wave holds by then).
__device__ __forceinline__ int tx_accept_fast(float& sumdx, float& sumdy, float cosv, float sinv, int myxy, int seedXY,
                                              unsigned long long& remaining, unsigned long long& acc, unsigned long long& seeds,
                                              int& cnt, int& bmin, int& bmax, float lo, float hi, int& j2,
                                              unsigned long long& grp, int grpXY) {
  int code, sc, ss, sxy;
  float t0, t1, t2, t3;
  unsigned long long m, sh;
  cnt = __builtin_amdgcn_readfirstlane(cnt);            // (wave-uniform values the register allocator may hold in vector registers)
  lo = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lo)));
  hi = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hi)));
  asm volatile(
      "1:\n\t"
      TX_DIAG_PAD_ASM
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
      "v_readlane_b32 %[xy], %[mxy], %[j]\n\t"
      "s_bitset1_b64 %[acc], %[j]\n\t"
      "s_add_i32 %[cnt], %[cnt], 1\n\t"
      "v_add_f32_e32 %[sx], %[sc], %[sx]\n\t"
      "v_add_f32_e32 %[sy], %[ss], %[sy]\n\t"
      "v_cmp_eq_u32_e32 vcc, %[xy], %[sp]\n\t"
      "v_pk_min_u16 %[bmin], %[bmin], %[xy]\n\t"
      "s_andn2_b64 %[seeds], %[seeds], vcc\n\t"
#if TX_GROUP
      "v_cmp_eq_u32_e32 vcc, %[xy], %[gxy]\n\t"
      "v_pk_max_u16 %[bmax], %[bmax], %[xy]\n\t"
      "s_andn2_b64 %[grp], %[grp], vcc\n\t"
      "v_cmp_eq_u32_e32 vcc, %[xy], %[mxy]\n\t"
#else
      "v_cmp_eq_u32_e32 vcc, %[xy], %[mxy]\n\t"
      "v_pk_max_u16 %[bmax], %[bmax], %[xy]\n\t"
#endif
      "s_andn2_b64 %[rem], %[rem], vcc\n\t"
      "s_cbranch_scc1 1b\n"
      "4:\n\t"
      "s_mov_b32 %[code], 0\n\t"
      "s_branch 9f\n"
      "5:\n\t"
      "s_mov_b32 %[code], 1\n"
      "9:\n\t"
      : [sx] "+v"(sumdx), [sy] "+v"(sumdy), [bmin] "+v"(bmin), [bmax] "+v"(bmax), [rem] "+s"(remaining), [acc] "+s"(acc),
        [seeds] "+s"(seeds), [cnt] "+s"(cnt), [code] "=&s"(code), [j] "=&s"(j2), [xy] "=&s"(sxy), [sc] "=&s"(sc),
        [ss] "=&s"(ss), [m] "=&s"(m), [sh] "=&s"(sh), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
        [grp] "+s"(grp)
      : [cs] "v"(cosv), [sn] "v"(sinv), [mxy] "v"(myxy), [sp] "v"(seedXY), [lo] "s"(lo), [hi] "s"(hi), [gxy] "v"(grpXY)
      : "vcc", "scc");
  return code;
}
__device__ __forceinline__ int tx_pk_min_u16(int a, int b) {
  return (int)((min((unsigned)a >> 16, (unsigned)b >> 16) << 16) | min((unsigned)a & 0xFFFFu, (unsigned)b & 0xFFFFu));
}
__device__ __forceinline__ int tx_pk_max_u16(int a, int b) {
  return (int)((max((unsigned)a >> 16, (unsigned)b >> 16) << 16) | max((unsigned)a & 0xFFFFu, (unsigned)b & 0xFFFFu));
}

// PACK (t == 1, not SPARSE, not SPEC): round 1 with owner_1 inside the pixel records (see tx_load_rec16).  1: the unclaimed words
// carry the pixels' own ids (k_tx_sort wrote them); 2 (key mode): they carry the gradient norm (the front pass wrote them, LAZY ids).
template <bool SPARSE, bool SPEC = false, int GQ = TX_GQ, int PACK = 0, bool HOT = false>
__device__ __forceinline__ void tx_grow_tile(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                             const float4* recAll /* (no __restrict__: the packed round 1 claims into its fourth words) */, int2* __restrict__ ownAll,
                                             const int2* __restrict__ listAll, const int* __restrict__ tileCntAll,
                                             int ts, int ntx, int nty, int* __restrict__ rgSizeAll,
                                             int2* __restrict__ rgBoxAll, const int* __restrict__ rgDirtyAll,
                                             const int* __restrict__ tileActAll, int TW, int TH,
                                             int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,
                                             int rectCap, int img, int tile, int t, const int* __restrict__ rankAll,
                                             int* __restrict__ rgLostAll, int* __restrict__ tileTouchAll,
                                             const TxDirtyLists& DL, int* q /* LDS, GQ */, int* gb /* LDS, TX_BMAXBLK */,
                                             int* park /* LDS, TX_PARK (SPEC: 8 x 64) */, const TxKeys* keysp = nullptr /* PACK == 2 */,
                                             int2* hotAll = nullptr /* HOT */, const float2* coldAll = nullptr /* HOT: or null (rec) */) {
  const DevParams& P = *Pp;
  RxCtl& c = ctl[img];
  if (c.state == 2 || c.overflow) return;
  const int ntile = ntx * nty;
  int n = tileCntAll[(int64_t)img * ntile + tile];
  if (n == 0) return;
  const int lane = threadIdx.x & 63;
  const int W = P.LW, H = P.LH;
  const int64_t npix = (int64_t)W * H;
  // (later rounds) the seeds stamped dirty for this round were listed per tile by whoever stamped them: up to 64 of them are
  // sorted by rank in registers and taken as the tile's only row; a longer list falls back to the walk over all the seeds
  // (the row of list entries a lane works on; the dirty list is sorted straight into it)
  int2 se = make_int2(TX_INF, -1);
  bool useDirty = false;
  if (SPARSE && DL.list) {
    const int nd = DL.cnt[(int64_t)img * ntile + tile];
    if (nd == 0) return;
    // (the list is taken: the next round's stamps start a new one — the host does not clear the counters between the rounds)
    if (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) DL.cnt[(int64_t)img * ntile + tile] = 0;
    if (nd <= 64) {
      useDirty = true;
      if (lane < nd) se = DL.list[((int64_t)img * ntile + tile) * ts * ts + lane];
      // bitonic sort of the 64 (rank, seed pixel) pairs, ascending by rank (ranks are distinct; the padding is TX_INF)
#pragma unroll
      for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
          const int ox = __shfl_xor(se.x, j2, 64), oy = __shfl_xor(se.y, j2, 64);
          const bool up = (lane & k2) == 0, lowHalf = (lane & j2) == 0;
          const bool takeMin = lowHalf == up;
          const bool swap = takeMin ? ox < se.x : ox > se.x;
          if (swap) { se.x = ox; se.y = oy; }
        }
      }
      n = nd;
    }
  }
  if (SPARSE && !useDirty) {
    // a dirty region activates the cells under its bounding box, its seed's cell among them: no active cell, nothing to do
    const int cpt = ts >> 3;                              // 8x8 cells per tile side (ts = 32, 64 or 128: up to 256 cells)
    bool act = false;
    for (int cidx = lane; cidx < cpt * cpt; cidx += 64) {
      const int cx = (tile % ntx) * cpt + cidx % cpt, cy = (tile / ntx) * cpt + cidx / cpt;
      act = act || (cx < TW && cy < TH && tileActAll[(int64_t)img * TW * TH + cy * TW + cx] == t);
    }
    if (!__builtin_amdgcn_ballot_w64(act)) return;
  }
  const float4* rec = recAll + img * npix;
  int2* own = ownAll + img * npix;
  // (HOT: {level-line angle, round 1's owner word} per pixel.  Round 1 — PACK — gathers and claims there; the later rounds take the
  // angle from it and the owner pair from the owner plane: 4 + 8 bytes per neighbour instead of 16 + 8, and nobody reads the 16-byte
  // records any more, so the front pass does not write them)
  // APPROX (round 1 on the hot records): the vector filter on v_cos / v_sin of the angle, the exact sums where asked for.  COLDP (the
  // later rounds): nothing approximate — a test gathers the EXACT {cos, sin} of the cold plane (8 bytes; cos = TX_NOTDEF marks a pixel
  // without a level-line angle) beside the owner pair, and the angle itself, which only the reference's own expression wants, is
  // fetched from the hot record for the one candidate that gets there.  Measured first with the later rounds on the approximate
  // filter too: k_tx_grow_sparse 5.3 -> 5.7 ms on photographs — their steps are few and latency-bound, the filter's extra instructions
  // buy nothing there.
  constexpr bool APPROX = HOT && PACK != 0, COLDP = HOT && PACK == 0;
  int2* hot = HOT ? hotAll + img * npix : nullptr;
  // (... the exact {cos, sin} of the records, for the pixels whose exact sums are asked for: the cold plane the front pass writes beside
  // the hot records when it does not write the 16-byte ones, else those)
  const float2* cold = (HOT && coldAll) ? coldAll + img * npix : nullptr;
  auto exactPair = [&](const int p, float& cy, float& cz) {
    if (cold) {
      const float2 cp = cold[p];
      cy = cp.x; cz = cp.y;
    } else {
      const float4 cr = rec[p];
      cy = cr.y; cz = cr.z;
    }
  };
  const int2* list = listAll + ((int64_t)img * ntile + tile) * ts * ts;
  // (key mode: 4-byte entries — the id, whose low bits are the seed pixel — at the start of the tile's slot: k_tx_sort)
  const int* listK = reinterpret_cast<const int*>(listAll) + ((int64_t)img * ntile + tile) * ts * ts;
  auto seedEntry = [&](int i) -> int2 {
    if (DL.rmask != -1) { const int v = listK[i]; return make_int2(v, v & DL.rmask); }
    return list[i];
  };
  int* rgSize = rgSizeAll + img * npix;
  int2* rgBox = rgBoxAll + img * npix;
  const int* rgDirty = rgDirtyAll + img * npix;
  int* arena = arenaAll + (int64_t)img * arenaCap;
  // the loser of every contested claim is noted (a region that lost a pixel it claimed is regrown in the next round)
  const bool noteLost = rgLostAll != nullptr;             // (stamped with the round: k_tx_diff2 reads round 1's, k_rx_mark the last round's)
  int* rgLost = rgLostAll ? rgLostAll + img * npix : nullptr;
  const int* rankOfPix = rankAll + img * npix;
  int* tileTouch = tileTouchAll ? tileTouchAll + (int64_t)img * TW * TH : nullptr;
  RxRect* rects = rectAll + (int64_t)img * rectCap;
  const int ci = t & 1;                                   // owner_t lives in component ci, owner_{t-1} in the other
  const double prec = P.prec;
  const int minReg = P.minRegSize;
  const float alignLo = P.alignLo, alignHi = P.alignHi;
  const int pi = lane >> 3;
  const int nm = (lane & 7) < 4 ? (lane & 7) : (lane & 7) + 1;      // 8 neighbours, raster order, centre skipped
  const int ndx = nm % 3 - 1, ndy = nm / 3 - 1;

  // ---- PACK: is the pixel with owner word w free for region id r? ----
  // A claimed word (bit 31 clear) holds the id of the region that has the pixel: free iff that id is higher.  An unclaimed word
  // stands for the pixel's own id.  PACK == 1: the id is in the word.  PACK == 2 (LAZY): the id (gradient bin from the strongest
  // down, then the pixel index) is not, but the word has the gradient norm in 2^-22 fixed point F, and the region knows where the
  // bins below and above its own begin (T0f, T1f, from its bin b and the image's bin width, +- 4 units: kfix is rounded to 2^-8, the
  // shifts truncate, F is truncated): F > T1f + margin: a higher bin, i.e. a lower id, used; F < T0f - margin: a lower bin, free;
  // strictly between: the region's own bin, the pixel index decides; within the margin of a bin boundary (~1e-5 of the pixels) the double plane decides.
  unsigned lzKfix = 0u;
  int lzMargin = 1 << 30, lzNb1 = 0, lzPixBits = 0, lzCoefLo = 0, lzCoefHi = 0;     // (the bin coefficient: a double in two scalar registers)
  const double* lzMg = nullptr;
  if (PACK == 2) {
    const unsigned long long mb = keysp->maxMg[img];
    const double maxGrad = __longlong_as_double((long long)mb);
    lzNb1 = keysp->nBins - 1;
    lzPixBits = keysp->pixbits;
    lzMg = keysp->mg + img * npix;
    {
      const long long cb = __double_as_longlong(lsd_bin_coef64(mb, keysp->nBins));
      lzCoefLo = __builtin_amdgcn_readfirstlane((int)(cb & 0xFFFFFFFFll));
      lzCoefHi = __builtin_amdgcn_readfirstlane((int)(cb >> 32));
    }
    // (an image whose largest norm does not fit the fixed point — no 8-bit image has one — leaves every unclaimed pixel to the double plane)
    // ... and so does an image whose bin width does not fit the 32-bit kfix (fewer than ~130 bins: the host does not choose the lazy form then)
    if (mb != 0ull && maxGrad < 500.0 && maxGrad / (double)lzNb1 < 3.99) {
      lzKfix = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(maxGrad / (double)lzNb1 * 1073741824.0 + 0.5));
      lzMargin = keysp->lazyMargin;
    }
  }
  // (bin b begins at b * kfix >> 8 in the fixed point; kfix < 2^30 and b < 2^10: two 24-bit multiplies instead of a 64-bit one)
  auto lazyT = [&](const int b) -> int {
    return (int)((__umul24((unsigned)b, lzKfix >> 12) << 4) + (__umul24((unsigned)b, lzKfix & 0xFFFu) >> 8));
  };
  // One region's view of the bins: [tBelow, ...) the bins below its own end before tBelow (F < tBelow: free), its own bin's interior
  // is tSame .. tSame + tSpan (the pixel index decides), F > tAbove: a higher bin (used); everything else is within the margin of a
  // boundary.  (Signed compares: F < 2^31.)
  struct LazyBins { int tBelow, tSame; unsigned tSpan; int tAbove; };
  auto lazyBins = [&](const int b) -> LazyBins {
    const int T0 = lazyT(b), T1 = b + 1 > lzNb1 ? 0x7FFFFFEF - min(lzMargin, 1 << 20) : lazyT(b + 1);
    LazyBins L;
    if (lzMargin >= (1 << 20)) {                          // (no fixed point to trust, or the test switch: everything unclaimed is left to the double plane)
      L.tBelow = INT_MIN; L.tSame = 0; L.tSpan = 0u; L.tAbove = INT_MAX;
      return L;
    }
    L.tBelow = T0 - lzMargin;
    L.tSame = T0 + lzMargin + 1;
    L.tSpan = T1 - lzMargin > L.tSame ? (unsigned)(T1 - lzMargin - L.tSame) : 0u;
    L.tAbove = T1 + lzMargin;
    return L;
  };
  // (as a wave mask, already restricted to the lanes `def` that hold a defined pixel: the compiler keeps lane predicates that meet at a
  // branch as 0 / 1 registers and compares them back into masks)
  auto freeMask = [&](const int w, const int r, const int b, const int spix, const LazyBins& L, const int qi, const bool def) -> unsigned long long {
    if (PACK != 2) return __builtin_amdgcn_ballot_w64(def && (int)((unsigned)w & 0x7FFFFFFFu) > r);
    const int F = w & 0x7FFFFFFF;
    const bool uncl = w < 0, below = F < L.tBelow, same = (unsigned)(F - L.tSame) < L.tSpan, above = F > L.tAbove;
    // (a claimed word is a non-negative id; an unclaimed one is negative: never > r)
    unsigned long long m = __builtin_amdgcn_ballot_w64(def & ((w > r) | (uncl & (below | (same & (qi > spix))))));
    const unsigned long long amb = __builtin_amdgcn_ballot_w64(def & uncl & !(below | same | above));
    if (amb != 0ull) {
      bool fr = false;
      if (__builtin_amdgcn_inverse_ballot_w64(amb)) {
        const double coef = __longlong_as_double(((long long)lzCoefHi << 32) | (long long)(unsigned)lzCoefLo);
        const int bq = lsd_bin64(lzMg[qi], coef, lzNb1 + 1);
        fr = bq < b || (bq == b && qi > spix);
      }
      m |= __builtin_amdgcn_ballot_w64(fr);
    }
    return m;
  };

  auto qget = [&](int k) -> int {
    int e = tx_lds_read(&q[min(k, GQ - 1)]);
    if (k >= GQ) {
      const int o = k - GQ;
      e = arena[gb[o / TX_BBLK] + o % TX_BBLK];
    }
    return e;
  };

  // One region, grown by the whole wave with the batched steps, from a given start: the seed alone (cnt = 1, k = 0, q[0] = the
  // seed — every region of the plain schedule) or the prefix a lane has grown by itself (the speculative schedule below: q[0..cnt)
  // hold its pixels, k is the first queue entry that has not been expanded, the sums are the lane's).  seedMask / seedPixV: the live
  // seeds of the current list row and their pixels, packed (y << 16 | x) (a seed whose pixel is taken leaves the mask).  false: a capacity ran out.
  // grpCand / grpXY / preOct (shared first steps): the candidates of the current group's parked first steps, their pixels, and the
  // octet of lanes that holds THIS region's (-1: the region fetches its first step itself).
  // (the current list row's seeds: cos / sin of their angles, one seed per lane — declared here so that growRegion can reach them: the
  // hot-record path re-reads the seed's pair, lane seedLaneOfRow, when it first folds the exact sums)
  float scos = 0.f, ssin = 0.f;
  int seedLaneOfRow = 0;
  auto growRegion = [&](const int r, const float sa, float sumdx, float sumdy, int cnt, int k, int bmin, int bmax,
                        unsigned long long& seedMask, const int seedPixV, const int pend0, const int pendRank0,
                        unsigned long long& grpCand, const int grpXY, const int preOct) -> bool {
      // the region angle for the reference's own expression, taken when a candidate needs it (the angle of the sums at pixel count
      // angCnt; with the seed alone it is the seed's level-line angle itself)
      double reg_angle = 0.0;
      int angCnt = -1;
      // HOT: sumdx / sumdy are the FILTER's sums — the seed's exact cos / sin plus v_cos_f32 / v_sin_f32 of every accepted pixel's angle.
      // Where the reference's own expression is evaluated (a candidate inside the filter's margin) the region angle must come from the
      // sums region_grow forms: the exact cos / sin of the 16-byte records added in the order of acceptance.  exSx / exSy hold them for
      // the first exCnt queue entries and are brought up to date only when asked for (foldQueue: one gather of the records per 128
      // entries, then the adds in order).  Why the filter may run on the approximate sums: an accepted pixel lies within prec + margin
      // of the sum's direction, so |sum| grows by at least cos(prec + margin) >= 0.54 per pixel (prec <= 1 rad, checked by the host), and
      // after m pixels from a point where the two pairs of sums agreed they differ, per component, by at most m d + 2 u m |sum| (d = the
      // largest |v_cos - cos|, tests/test_gpu_parity.py measures it over every float angle: < 4e-6; u = 2^-24, one rounding per add and
      // pair) — in direction by sqrt(2) (1.85 d + 2 u m) = 1.05e-5 + 1.7e-7 m rad.  The sums are made to agree at the latest every
      // TX_HOT_RESYNC = 2048 pixels (and whenever the exact ones are computed): below 3.8e-4 rad with the candidate's own 1.5 d.  The
      // margin is 0.05 deg = 8.7e-4 rad, of which fastAtan2 uses 1.7e-4 (NOTEBOOK.md R1).
      // (the three words live in LDS — exs = the words behind the queue-block table —: the step loops have neither vector nor scalar
      // registers to spare, and a scalar register the compiler cannot keep costs a v_readlane per use)
      int* const exs = gb + TX_BMAXBLK;                   // [0] exCnt, [1] / [2] the bits of exSx / exSy, [3] the pixel count at which the filter's sums were last exact
      // (a region starts with ONE store — exs[0] = 0: nothing folded yet, the exact sums are the seed's, the filter's sums were exact
      // at pixel count 1 —: three more per region were 0.2 G vector instructions per launch)
      if (APPROX) exs[0] = 0;
      auto foldQueue = [&](const int upto) {              // (the entries exCnt .. upto - 1 are in the queue)
#if defined(TX_DIAG_HOT_NOFOLD)      // diagnostic build, NOT exact: what the exact sums cost (events decided on the filter's sums)
        return;
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int exCnt = __builtin_amdgcn_readfirstlane(tx_lds_read(&exs[0]));
        float ex = __int_as_float(tx_lds_read(&exs[1])), ey = __int_as_float(tx_lds_read(&exs[2]));
        if (exCnt == 0) {                                 // (the first fold of this region: the seed, as region_grow starts its sums)
          ex = tx_rlf(scos, seedLaneOfRow); ey = tx_rlf(ssin, seedLaneOfRow);
          exCnt = 1;
          exs[3] = 1;
        }
        while (exCnt < upto) {
          // (2 x 64 entries per trip; the gathers unconditional, from an entry that exists either way, so that they are in flight together)
          const int n4 = min(128, upto - exCnt);
          float cy[2], cz[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = qget(min(exCnt + 64 * u + lane, upto - 1));
            exactPair((e >> 16) * W + (e & 0xFFFF), cy[u], cz[u]);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int nn = min(64, n4 - 64 * u);
            for (int j = 0; j < nn; ++j) {
              ex = __fadd_rn(ex, tx_rlf(cy[u], j));
              ey = __fadd_rn(ey, tx_rlf(cz[u], j));
            }
          }
          exCnt += n4;
        }
        exs[0] = exCnt; exs[1] = __float_as_int(ex); exs[2] = __float_as_int(ey);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      };
      auto exSx = [&]() -> float { return __int_as_float(tx_lds_read(&exs[1])); };
      auto exSy = [&]() -> float { return __int_as_float(tx_lds_read(&exs[2])); };
      // the exact sums at this moment of a step: the queue up to the step's start, then the lanes accepted so far (in lane order)
      auto exactSums = [&](const int cnt0, const unsigned long long accSoFar, const int myxy, float& ax, float& ay) {
        foldQueue(cnt0);
        ax = exSx(); ay = exSy();
        if (accSoFar) {
          float cy = 0.f, cz = 0.f;
          if ((accSoFar >> lane) & 1ull) {
            // (the pixel index made opaque: the compiler hoisted this rare gather's address arithmetic — four vector instructions — out of
            // the accept loop into every step)
            int pxy = myxy;
            asm volatile("" : "+v"(pxy));
            exactPair((pxy >> 16) * W + (pxy & 0xFFFF), cy, cz);
          }
          unsigned long long m = accSoFar;
          while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            ax = __fadd_rn(ax, tx_rlf(cy, j));
            ay = __fadd_rn(ay, tx_rlf(cz, j));
          }
        }
      };
      // (whenever the exact sums of the moment are known the filter goes on from them: the two agree again at this pixel count)
      auto filterFrom = [&](const float ax, const float ay) {
        sumdx = ax; sumdy = ay;
        exs[3] = cnt;
      };
      // The reference's own expression for a candidate of level-line angle angJ (a lane inside the margin of the vector filter).  The
      // region angle is fastAtan2 of the sums — cached per pixel count —; exactAt brings the exact sums on the hot records.  There a
      // SECOND band comes first: fastAtan2 of the FILTER's sums is within the direction bound of fastAtan2 of the exact ones when both lie
      // in one octant of the approximation, i.e. when the filter's sums are farther than that bound from the axes and diagonals (inside
      // an octant the polynomial's slope is below 1.002, and its float evaluation adds 2e-6 rad) — so a candidate farther than that from
      // prec is decided without the exact sums.  The bound grows with the pixels accepted since the two pairs of sums last agreed (they
      // are made to agree whenever the exact ones are known): a few 1e-5 rad against the margin's 8.7e-4, so few events are left that
      // gather the 16-byte records of the region's pixels.
      // (the level-line angle of lane j2's candidate: in the lane's register, or — COLDP — in the hot record of its pixel)
      auto candAngle = [&](const float angReg, const int xyReg, const int j2) -> float {
        if (!COLDP) return tx_rlf(angReg, j2);
        const int xy = tx_rl(xyReg, j2);
        return __int_as_float(hot[(xy >> 16) * W + (xy & 0xFFFF)].x);
      };
      auto alignedExact = [&](const float angJ, auto&& exactAt) -> bool {
        const double aj = (double)angJ * TX_DEG2RAD;
        auto ntheta = [&](const double ra) -> double {
          double n = fabs(ra - aj);
          if (n > TX_3_2_PI) {
            n = fabs(n - TX_2PI);
          }
          return n;
        };
        if (APPROX) {
          // (nothing cached between the events of a region: they are rare, and the step loops have no registers to spare)
          if (cnt == 1) return ntheta((double)sa * TX_DEG2RAD) <= prec;
          // (the direction bound of the comment above for the m pixels since the sums last agreed: sqrt(2) (1.85 d + 2 u m) = 1.05e-5 +
          // 1.7e-7 m, times the polynomial's slope 1.002, plus 2e-6 for its evaluation — with room to spare)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          const int syncCnt = tx_lds_read(&exs[0]) == 0 ? 1 : tx_lds_read(&exs[3]);
          const float dirBound = 1.5e-5f + 1.8e-7f * (float)(cnt + 64 - syncCnt);
          const float fx = fabsf(sumdx), fy = fabsf(sumdy), mn = fminf(fx, fy), mx = fmaxf(fx, fy);
          if (mn > dirBound * mx && mx - mn > 2.1f * dirBound * mx) {
            const double n1 = ntheta((double)fast_atan2_deg(sumdy, sumdx) * TX_DEG2RAD);
            if (fabs(n1 - prec) > (double)dirBound + P.hotBand2) return n1 <= prec;
          }
          float ax, ay;
          exactAt(ax, ay);
          filterFrom(ax, ay);
          return ntheta((double)fast_atan2_deg(ay, ax) * TX_DEG2RAD) <= prec;
        }
        if (angCnt != cnt) {
          reg_angle = (double)(cnt == 1 ? sa : fast_atan2_deg(sumdy, sumdx)) * TX_DEG2RAD;
          angCnt = cnt;
        }
        return ntheta(reg_angle) <= prec;
      };
      // pendOld = what stood in the owner word when this lane's claim of the last step arrived, pendRank = the own rank of the pixel it
      // claimed — what an unclaimed word holds — (r and r: the lane claimed nothing): a lower rank -> this region does not hold the
      // pixel it took; a higher rank that is not the pixel's own -> that region just lost the pixel.  Contested claims are rare:
      // pendOld != pendRank covers both cases, and one compare and one ballot decide for the wave.
      int pendOld = PACK ? -1 : pend0, pendRank = pendRank0;      // (PACK: "unclaimed" stands for "this lane claimed nothing")
      // (LAZY ids: the region's bin, its seed pixel, and where the bins around its own begin)
      const int rBin = PACK == 2 ? lzNb1 - (r >> lzPixBits) : 0, rSeedPix = PACK == 2 ? (r & ((1 << lzPixBits) - 1)) : 0;
      LazyBins rBins{};
      if (PACK == 2) rBins = lazyBins(rBin);
      auto stampLosers = [&]() {
        if (noteLost && PACK) {
          // (the word a claim found: unclaimed — bit 31 —, or a region's id: a lower one keeps the pixel and THIS region has lost it,
          // a higher one has just lost it)
          // (a claim never finds the region's own id: the accept loop takes a pixel once)
          if (__builtin_amdgcn_ballot_w64(pendOld >= 0) != 0ull) {
            if (pendOld >= 0) rgLost[max(pendOld, r) & DL.rmask] = t;
          }
          pendOld = -1;
          return;
        } else if (noteLost) {
          if (__builtin_amdgcn_ballot_w64(pendOld != pendRank) != 0ull) {
            const bool contested = pendOld < r || (pendOld != r && pendOld != pendRank);
            if (contested) rgLost[(pendOld < r ? r : pendOld) & DL.rmask] = t;
          }
        }
        pendOld = r;
        pendRank = r;
      };
      bool dead = false;
      // The accept loop of a step whose queue entries fit the LDS queue (tx_accept_fast): walks the candidates `remaining` in lane order;
      // accepted lanes write their queue entries after the loop (they are accepted in increasing lane order).  True for the accepted lanes.
      auto acceptBatch = [&](unsigned long long remaining, const float ang, const float cosv, const float sinv, const int myxy) -> bool {
        unsigned long long acc = 0ull;
        const int cnt0 = cnt;
        while (remaining) {
          int j2;
          const int code = tx_accept_fast(sumdx, sumdy, cosv, sinv, myxy, seedPixV, remaining, acc, seedMask, cnt, bmin, bmax,
                                          alignLo, alignHi, j2, grpCand, grpXY);
          if (code == 0) break;
          // lane j2 lies inside the margin of the vector filter: the reference's own expression decides
          // (the sums live in vector registers, so the compiler takes this decision for lane-dependent: say it is not)
          if (!__builtin_amdgcn_readfirstlane((int)alignedExact(candAngle(ang, myxy, j2), [&](float& ax, float& ay) { exactSums(cnt0, acc, myxy, ax, ay); })))
            continue;
          const int xyj = tx_rl(myxy, j2);
          const float cj = tx_rlf(cosv, j2), sj = tx_rlf(sinv, j2);
          acc |= 1ull << j2;
          remaining &= ~__builtin_amdgcn_ballot_w64(myxy == xyj);        // the other copies of the accepted pixel
          seedMask &= ~__builtin_amdgcn_ballot_w64(seedPixV == xyj);      // a seed of this row that was just taken
          if (TX_GROUP) grpCand &= ~__builtin_amdgcn_ballot_w64(grpXY == xyj);   // ... a candidate of a member that comes later
          ++cnt;
          bmin = tx_pk_min_u16(bmin, xyj);
          bmax = tx_pk_max_u16(bmax, xyj);
          sumdx = __fadd_rn(sumdx, cj);
          sumdy = __fadd_rn(sumdy, sj);
        }
        // (the accepted lanes straight from the scalar mask: one s_and_saveexec; their queue slots by v_mbcnt)
        const bool accepted = __builtin_amdgcn_inverse_ballot_w64(acc);
        if (accepted) q[cnt0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(acc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)acc, 0u))] = myxy;
        return accepted;
      };
      // One step = up to 8 queue entries x 8 neighbours in one round trip (record + owner pair per lane), the accept loop, the
      // claims.  Two loops, the usual one first (the queue entries of the step fit the LDS queue) and the overflow form after
      // it: as two instances inside ONE loop they shared their loop-carried state and the compiler moved ~20 registers per step.
      auto oneStep = [&](auto spillTag) {
        constexpr bool SPILL = decltype(spillTag)::value;
        // single-wave block: the LDS operations of a wave execute in order (the compiler only has to keep the order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // The claims of the previous step are returning atomics: their results are consumed here, before the owner
        // loads of this step are issued, so those loads see the region's own claims.
        asm volatile("" ::"v"(pendOld) : "memory");
        stampLosers();
        // (the filter's sums and the exact ones agree again every TX_HOT_RESYNC pixels: a step adds at most 64, so the count cannot
        // pass a multiple of TX_HOT_RESYNC without a step starting in the 64 behind it)
        if (APPROX && cnt >= TX_HOT_RESYNC && (cnt & (TX_HOT_RESYNC - 1)) < 64) {
          foldQueue(cnt);
          filterFrom(exSx(), exSy());
        }
        bool accepted = false;
        const int nb = min(8, cnt - k);
        // (one predicate for the whole fetch: the lanes outside it keep whatever their registers hold, and stay out of `cand`)
        const int e = SPILL ? qget(min(k + pi, cnt - 1)) : tx_lds_read(&q[min(k + pi, cnt - 1)]);
        const int nx = (e & 0xFFFF) + ndx, ny = (e >> 16) + ndy;
        const bool ok = pi < nb && nx >= 0 && ny >= 0 && nx < W && ny < H;
        // (the pixel's index with a 24-bit multiply, full rate; lanes outside `ok` hold anything: nothing of theirs is used)
        const int qi = (int)(__umul24((unsigned)ny, (unsigned)W) + (unsigned)nx);
        const int myxy = (ny << 16) | nx;
        float4 rr;
        int2 oo;
        asm volatile("" : "=v"(rr.x), "=v"(rr.y), "=v"(rr.z), "=v"(oo.x), "=v"(oo.y));
        rr.w = 0.f;
        if (PACK) asm volatile("" : "=v"(rr.w));
        if (HOT) {
          if (PACK) {
            if (ok) {
              const int2 h = tx_load_hot8(hot + (unsigned)qi);
              rr.x = __int_as_float(h.x);
              rr.w = __int_as_float(h.y);
            }
          } else if (ok) {                                 // (later rounds: the exact pair of the cold plane, the owner pair of the owner plane)
            const float2 cp = cold[(unsigned)qi];
            rr.x = cp.x; rr.y = cp.x; rr.z = cp.y;         // (rr.x: TX_NOTDEF or not — the angle is fetched where it is needed)
            oo = tx_load_own(own + (unsigned)qi);
          }
          // (the filter's cos / sin.  Making the two conditional on "a candidate is left" — a third of the steps have none — was measured:
          // the branch costs more than the instructions, k_tx_grow 18.7 -> 19.2 ms)
          if (APPROX) {
            rr.y = tx_hw_cos_deg(rr.x);
            rr.z = tx_hw_sin_deg(rr.x);
          }
        } else if (PACK) {
          if (ok) rr = tx_load_rec16(rec + (unsigned)qi);
        } else if (ok) {
          rr = rec[(unsigned)qi];
          oo = tx_load_own(own + (unsigned)qi);
        }
        const int prevv = ci ? oo.x : oo.y, curv = ci ? oo.y : oo.x;
        unsigned long long remaining;
        if (PACK) remaining = freeMask(__float_as_int(rr.w), r, rBin, rSeedPix, rBins, qi, ok && rr.x != TX_NOTDEF);
        else remaining = __builtin_amdgcn_ballot_w64(ok && rr.x != TX_NOTDEF && !(prevv < r || curv <= r));
        if (!SPILL) {
          accepted = acceptBatch(remaining, rr.x, rr.y, rr.z, myxy);
        } else {
          while (remaining) {
            // the alignment test in vector form with the exact expression inside the margin (see k_lsd_grow)
            const float n2 = __builtin_fmaf(sumdx, sumdx, sumdy * sumdy);
            const float dot = __builtin_fmaf(sumdx, rr.y, sumdy * rr.z);
            const float sd2 = dot * __builtin_fabsf(dot);
            // (without the vector filter alignLo / alignHi are -inf / +inf: always maybe, never sure)
            const unsigned long long m = __builtin_amdgcn_ballot_w64(sd2 >= alignLo * n2) & remaining;
            if (!m) break;
            const int j2 = __ffsll((long long)m) - 1;
            remaining &= ~1ull << j2;
            const unsigned long long sure = __builtin_amdgcn_ballot_w64(sd2 >= alignHi * n2);
            if (__builtin_expect(!((sure >> j2) & 1ull), 0)) {
              // (this loop puts a pixel into the queue as it accepts it)
              if (!__builtin_amdgcn_readfirstlane((int)alignedExact(candAngle(rr.x, myxy, j2), [&](float& ax, float& ay) {
                    foldQueue(cnt);
                    ax = exSx(); ay = exSy();
                  })))
                continue;
            }
            const int xyj = tx_rl(myxy, j2);
            const float cj = tx_rlf(rr.y, j2), sj = tx_rlf(rr.z, j2);
            accepted = accepted || lane == j2;          // the claims are issued together after the loop
            remaining &= ~__builtin_amdgcn_ballot_w64(myxy == xyj);        // the other copies of the accepted pixel
            if (cnt < GQ) {
              q[cnt] = xyj;                              // every active lane stores the same value
            } else {
              const int o = cnt - GQ;
              if (o % TX_BBLK == 0) {
                if (o / TX_BBLK >= TX_BMAXBLK) { dead = true; break; }
                int nbk = 0;
                if (lane == j2) {                        // lane j2 is active: it is a set bit of a ballot
                  const unsigned long long ra = atomicAdd(&c.rectArena, (unsigned long long)TX_BBLK) & ((1ull << RX_ARENA_BITS) - 1ull);
                  nbk = ra + TX_BBLK > (unsigned long long)arenaCap ? -1 : (int)ra;
                }
                nbk = tx_rl(nbk, j2);
                if (nbk < 0) { dead = true; break; }
                gb[o / TX_BBLK] = nbk;
              }
              if (lane == j2) arena[gb[o / TX_BBLK] + o % TX_BBLK] = xyj;
              __threadfence_block();
            }
            ++cnt;
            bmin = tx_pk_min_u16(bmin, xyj);
            bmax = tx_pk_max_u16(bmax, xyj);
            sumdx = __fadd_rn(sumdx, cj);
            sumdy = __fadd_rn(sumdy, sj);
            seedMask &= ~__builtin_amdgcn_ballot_w64(seedPixV == xyj);  // a seed of this row that was just taken
            if (TX_GROUP) grpCand &= ~__builtin_amdgcn_ballot_w64(grpXY == xyj);
          }
        }
        if (accepted) {
#if defined(TX_DIAG_NOWAIT)     // diagnostic build (NOT exact: contested claims go unnoticed): non-returning claims, nothing to wait for
          (void)__hip_atomic_fetch_min(ci ? &own[qi].y : &own[qi].x, r, __ATOMIC_RELAXED, TX_CLAIM_SCOPE);
#else
          if (HOT && PACK) pendOld = (int)__hip_atomic_fetch_min(reinterpret_cast<unsigned*>(&hot[(unsigned)qi].y), (unsigned)r, __ATOMIC_RELAXED, TX_CLAIM_SCOPE);
          else if (PACK) pendOld = (int)__hip_atomic_fetch_min(reinterpret_cast<unsigned*>(tx_rec_owner(rec + (unsigned)qi)), (unsigned)r, __ATOMIC_RELAXED, TX_CLAIM_SCOPE);
          else pendOld = __hip_atomic_fetch_min(ci ? &own[(unsigned)qi].y : &own[(unsigned)qi].x, r, __ATOMIC_RELAXED, TX_CLAIM_SCOPE);
#endif
          // (the pixel's own rank, which an unclaimed owner word holds: in round 1 owner_0 is the trivial map, so the word just read has it)
          if (noteLost && !PACK) pendRank = (SPARSE || t != 1) ? rankOfPix[qi] : prevv;
          // (later rounds) the 8x8 cell of every claimed pixel is noted: the next round's k_rx_diff only looks where a claim or the
          // round's k_tx_prep wrote
          if (SPARSE && tileTouch) tileTouch[(myxy >> 19) * TW + ((myxy & 0xFFFF) >> 3)] = t;
        }
        k += nb;
      };
      if (TX_GROUP && preOct >= 0) {
        // The region's FIRST step from the group's parked fetch (cnt == 1, k == 0): the lanes of octet preOct hold the seed's 8
        // neighbours as they stood at group time, minus what this wave's own lower ids have taken since (the accept loops keep
        // grpCand) — what a fetch of its own would show now, except for claims other tiles' waves made in between, which the
        // claim's returned value settles like any other stale read.  No candidate left: the region is its seed, without a step.
        const unsigned long long octet = 0xFFull << (8 * preOct);
        const unsigned long long remaining = grpCand & octet;
        grpCand &= ~octet;
        if (remaining) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          const float pa = __int_as_float(tx_lds_read(&park[lane])), pc = __int_as_float(tx_lds_read(&park[64 + lane])),
                      ps = __int_as_float(tx_lds_read(&park[128 + lane]));
          if (acceptBatch(remaining, pa, pc, ps, grpXY)) {
            const int qi = (grpXY >> 16) * W + (grpXY & 0xFFFF);
            if (HOT && PACK) pendOld = (int)__hip_atomic_fetch_min(reinterpret_cast<unsigned*>(&hot[qi].y), (unsigned)r, __ATOMIC_RELAXED, TX_CLAIM_SCOPE);
            else if (PACK) pendOld = (int)__hip_atomic_fetch_min(reinterpret_cast<unsigned*>(tx_rec_owner(&rec[qi])), (unsigned)r, __ATOMIC_RELAXED, TX_CLAIM_SCOPE);
            else pendOld = __hip_atomic_fetch_min(ci ? &own[qi].y : &own[qi].x, r, __ATOMIC_RELAXED, TX_CLAIM_SCOPE);
            if (noteLost && !PACK) pendRank = (SPARSE || t != 1) ? rankOfPix[qi] : tx_lds_read(&park[192 + lane]);
            if (SPARSE && tileTouch) tileTouch[(grpXY >> 19) * TW + ((grpXY & 0xFFFF) >> 3)] = t;
          }
        }
        k = 1;
      }
      while (k < cnt && cnt + 8 * 8 + 1 <= GQ) oneStep(std::false_type{});
      while (k < cnt && !dead) oneStep(std::true_type{});
      asm volatile("" ::"v"(pendOld) : "memory");       // the claims of the region's last step
      stampLosers();
      asm volatile("" ::"v"(pendOld) : "memory");
      if (dead) { c.overflow = 5; return false; }
      // ---- the region is complete ----
      const bool first = lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1;   // the first ACTIVE lane
      if (first) {
        rgSize[r & DL.rmask] = cnt;
        rgBox[r & DL.rmask] = make_int2(bmin, bmax);
      }
      if (cnt >= minReg) {                              // the pixel list goes to k_rx_rect (region2rect)
        int off = 0, slot = 0;
        if (first) {
          const unsigned long long ra = atomicAdd(&c.rectArena, (1ull << RX_ARENA_BITS) | (unsigned long long)cnt);
          const unsigned long long o64 = ra & ((1ull << RX_ARENA_BITS) - 1ull);
          off = o64 + cnt > (unsigned long long)arenaCap ? -1 : (int)o64;
          slot = (int)(ra >> RX_ARENA_BITS);
        }
        const int fl = __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1;
        off = tx_rl(off, fl); slot = tx_rl(slot, fl);
        if (off < 0 || slot >= rectCap) { c.overflow = off < 0 ? 3 : 4; return false; }
        for (int i = lane; i < cnt; i += 64) arena[off + i] = qget(i);
        if (first) {
          RxRect& it = rects[slot];
          it.rank = r; it.off = off; it.cnt = cnt; it.sumdx = sumdx; it.sumdy = sumdy;
          it.approx = APPROX ? 1 : 0;                     // (the filter's sums; region2rect asks for the exact ones when it needs them)
        }
      }
      return true;
  };

  if constexpr (SPEC) {
    // ------------------------------------------------------------------------------------------------------------------
    // Speculative schedule of round 1 (the protocol of lsd_grow_image_spec, line_kernels.hip, with the owner words as the
    // record of who holds what).  Most regions stay below nine pixels, and the plain schedule spends a whole wave's steps on
    // each of them.  Here the wave collects 64 seeds of its list that are ALIVE (a "super-row", in rank order), every lane grows
    // the region of its seed ALONE — no claims, against the owner words as they stand — up to TX_SPEC_CAP pixels, and then, in
    // rank order:  a run of lanes that stayed below the cap is validated and committed together (a lane's speculative run equals
    // its sequential run iff every pixel it ACCEPTED is still free of lower ranks when its turn comes: what it rejected stays
    // rejected — owner words only go down within a round — and what it accepted decides everything it did afterwards); a lane
    // that hit the cap is taken over by the whole wave where its last complete step ended (same condition on its prefix),
    // or from its seed if the prefix does not stand; a lane whose seed is gone is dropped.  Two lanes of one run that want the
    // same pixel settle it by their claims (atomicMin), and the loser is stamped for round 2, exactly as between tiles.
    // ------------------------------------------------------------------------------------------------------------------
    constexpr int CAP = TX_SPEC_CAP;
    const int cap = min(CAP, minReg - 1);                 // a region that may yield a segment is never finished by a lane
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    auto parkAt = [&](int row, int l) -> int& { return park[row * 64 + l]; };
    auto wsync = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto entryXY = [&](unsigned long long lst, int i, int sxy) -> int {          // packed (y << 16 | x) of list entry i
      const int off = (int)((lst >> (8 * i)) & 0xFFull);
      return (((sxy >> 16) + (off >> 4) - 8) << 16) | ((sxy & 0xFFFF) + (off & 15) - 8);
    };
    int pos = 0;
    while (pos < n) {
      // ---- collect: up to 64 alive seeds, in list order (several list rows per trip: their loads are issued together) ----
      int nst = 0;
      bool full = false;
      while (!full && nst < 64 && pos < n) {
        const int R = nst < 32 ? 4 : 2;
        int2 se4[4];
        int2 so4[4];
        float an4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = pos + 64 * u + lane;
          se4[u] = make_int2(TX_INF, -1);
          if (u < R && idx < n) se4[u] = seedEntry(idx);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          so4[u] = make_int2(0, 0);
          an4[u] = 0.f;
          if (se4[u].y >= 0) {
            so4[u] = tx_load_own(&own[se4[u].y]);
            an4[u] = rec[se4[u].y].x;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u >= R || full || pos >= n) continue;
          const bool alive = se4[u].y >= 0 && so4[u].x == se4[u].x && so4[u].y == se4[u].x;
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(alive);
          const int c = __popcll(bal);
          if (nst + c > 64) { full = true; continue; }        // this row does not fit: it starts the next super-row
          if (alive) {
            const int at = nst + __popcll(bal & ltMask);
            parkAt(0, at) = se4[u].x;
            parkAt(1, at) = se4[u].y;
            parkAt(2, at) = __float_as_int(an4[u]);
          }
          nst += c;
          pos += 64;
        }
      }
      if (nst == 0) continue;
      wsync();
      // ---- speculation: lane l < nst grows the region of staged seed l alone ----
      // (everything a lane knows about its region is parked in LDS afterwards and read back where it is needed: nothing of it
      // may stay in registers across the whole-wave growth below, whose loops have none to spare)
      unsigned long long smallMask, bigMask;
      {
      const bool has = lane < nst;
      const int r_l = has ? parkAt(0, lane) : TX_INF;
      const int sp_l = has ? parkAt(1, lane) : 0;
      const float sa_l = has ? __int_as_float(parkAt(2, lane)) : 0.f;
      wsync();
      const int spy = sp_l / W, spx = sp_l - spy * W;
      float sumdx = 0.f, sumdy = 0.f;
      double reg_angle = (double)sa_l * TX_DEG2RAD;
      if (has) {
        double sn, cn;
        sincos(reg_angle, &sn, &cn);
        sumdx = (float)cn;
        sumdy = (float)sn;
      }
      unsigned long long lst = 0x88ull;                   // byte i: entry i as (dy + 8) << 4 | (dx + 8) relative to the seed
      int cnt = 1, k = 0;
      bool big = false;
      {
        bool active = has;
        while (__builtin_amdgcn_ballot_w64(active)) {
          if (active) {
            const int off = (int)((lst >> (8 * k)) & 0xFFull);
            const int px = spx + (off & 15) - 8, py = spy + (off >> 4) - 8;
            const int cnt0 = cnt;                          // a step that hits the cap is abandoned: the state at its start stands
            const unsigned long long lst0 = lst;
            const float sdx0 = sumdx, sdy0 = sumdy;
            int2 no[8];
            float na[8];
#pragma unroll
            for (int nn = 0; nn < 8; ++nn) {
              const int m = nn < 4 ? nn : nn + 1;         // the 3 x 3 block in raster order, centre skipped
              const int nx = px + m % 3 - 1, ny = py + m / 3 - 1;
              na[nn] = TX_NOTDEF;
              no[nn] = make_int2(0, 0);
              if (nx >= 0 && ny >= 0 && nx < W && ny < H) {
                na[nn] = rec[ny * W + nx].x;
                no[nn] = tx_load_own(&own[ny * W + nx]);
              }
            }
#pragma unroll
            for (int nn = 0; nn < 8; ++nn) {
              if (big || na[nn] == TX_NOTDEF) continue;
              const int prevv = ci ? no[nn].x : no[nn].y, curv = ci ? no[nn].y : no[nn].x;
              if (prevv < r_l || curv <= r_l) continue;
              const int m = nn < 4 ? nn : nn + 1;
              const int nx = px + m % 3 - 1, ny = py + m / 3 - 1;
              const int ox = nx - spx + 8, oy = ny - spy + 8;
              if (((ox | oy) & ~15) == 0) {                // (beyond +-7 of the seed no pixel of the list lies)
                const unsigned long long x = lst ^ ((unsigned long long)((oy << 4) | ox) * 0x0101010101010101ull);
                const unsigned long long z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
                const unsigned long long inList = cnt >= 8 ? ~0ull : ((1ull << (8 * cnt)) - 1ull);
                if (z & inList) continue;                  // already mine
              }
              double n_theta = fabs(reg_angle - (double)na[nn] * TX_DEG2RAD);
              if (n_theta > TX_3_2_PI) {
                n_theta = fabs(n_theta - TX_2PI);
              }
              if (!(n_theta <= prec)) continue;
              if (cnt == cap) { big = true; continue; }
              const float4 rq = rec[ny * W + nx];
              lst |= (unsigned long long)((oy << 4) | ox) << (8 * cnt);
              ++cnt;
              sumdx = __fadd_rn(sumdx, rq.y);
              sumdy = __fadd_rn(sumdy, rq.z);
              reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * TX_DEG2RAD;
            }
            if (big) { cnt = cnt0; lst = lst0; sumdx = sdx0; sumdy = sdy0; active = false; }
            else {
              ++k;
              if (k >= cnt) active = false;
            }
          }
        }
      }
      // ---- park the lanes' states; the whole wave works on one lane's region at a time from here on ----
      const int sxy_l = (spy << 16) | spx;
      parkAt(0, lane) = r_l;
      parkAt(1, lane) = sxy_l;
      parkAt(2, lane) = (int)(unsigned)(lst & 0xFFFFFFFFull);
      parkAt(3, lane) = (int)(unsigned)(lst >> 32);
      parkAt(4, lane) = cnt | (k << 8);
      parkAt(5, lane) = __float_as_int(sumdx);
      parkAt(6, lane) = __float_as_int(sumdy);
      parkAt(7, lane) = __float_as_int(sa_l);
      wsync();
      smallMask = __builtin_amdgcn_ballot_w64(has && !big);
      bigMask = __builtin_amdgcn_ballot_w64(has && big);
      }
      unsigned long long regrowMask = 0ull;                // lanes below the cap whose run does not stand (their seed does)
      int lo = 0;
      unsigned long long noSeeds = 0ull;
      for (;;) {
        int j;
        bool scratch;
        if (regrowMask) {
          j = __ffsll((long long)regrowMask) - 1;
          regrowMask &= regrowMask - 1ull;
          scratch = true;
        } else {
          const unsigned long long bigAbove = lo < 64 ? (bigMask >> lo) << lo : 0ull;
          const int jb = bigAbove ? __ffsll((long long)bigAbove) - 1 : 64;
          const unsigned long long seg = smallMask & (jb < 64 ? (1ull << jb) - 1ull : ~0ull);
          if (seg) {
            // ---- a run of lanes below the cap: validated and committed together ----
            smallMask &= ~seg;
            const bool mine = (seg >> lane) & 1ull;
            const int r = parkAt(0, lane), sxy_l = parkAt(1, lane), cnt = parkAt(4, lane) & 0xFF;
            const unsigned long long lst = ((unsigned long long)(unsigned)parkAt(3, lane) << 32) | (unsigned long long)(unsigned)parkAt(2, lane);
            int2 ow[CAP];
#pragma unroll
            for (int i = 0; i < CAP; ++i) {
              ow[i] = make_int2(0, 0);
              if (mine && i < cnt) {
                const int xy = entryXY(lst, i, sxy_l);
                ow[i] = tx_load_own(&own[(xy >> 16) * W + (xy & 0xFFFF)]);
              }
            }
            bool ok = mine, seedGone = false;
#pragma unroll
            for (int i = 0; i < CAP; ++i) {
              if (!(mine && i < cnt)) continue;
              const int pv = ci ? ow[i].x : ow[i].y, cv = ci ? ow[i].y : ow[i].x;
              if (pv < r || cv < r) { ok = false; if (i == 0) seedGone = true; }
            }
            if (ok) {
              int bmin = sxy_l, bmax = sxy_l;
              int olds[CAP];
#pragma unroll
              for (int i = 1; i < CAP; ++i) {
                olds[i] = r;
                if (i < cnt) {
                  const int xy = entryXY(lst, i, sxy_l);
                  int2* w = &own[(xy >> 16) * W + (xy & 0xFFFF)];
                  olds[i] = __hip_atomic_fetch_min(ci ? &w->y : &w->x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  bmin = tx_pk_min_u16(bmin, xy);
                  bmax = tx_pk_max_u16(bmax, xy);
                }
              }
              rgSize[r & DL.rmask] = cnt;
              rgBox[r & DL.rmask] = make_int2(bmin, bmax);
              if (noteLost) {
#pragma unroll
                for (int i = 1; i < CAP; ++i) {
                  if (i >= cnt) continue;
                  const int pv = ci ? ow[i].x : ow[i].y;       // round 1: owner_0 is the trivial map, i.e. the pixel's own rank
                  if (olds[i] < r) rgLost[r & DL.rmask] = t;                              // a lower rank slipped in between the look and the claim
                  else if (olds[i] != r && olds[i] != pv) rgLost[olds[i] & DL.rmask] = t; // a higher rank held it: it has just lost the pixel
                }
              }
            }
            regrowMask = __builtin_amdgcn_ballot_w64(mine && !ok && !seedGone);
            lo = jb;
            if (regrowMask) continue;
          }
          if (jb >= 64) break;
          j = jb;
          bigMask &= ~(1ull << jb);
          lo = jb + 1;
          scratch = false;
        }
        // ---- lane j's region by the whole wave: from the prefix it has grown (if that stands), or from its seed ----
        const int r = parkAt(0, j), sxy = parkAt(1, j);
        const unsigned long long lstj = ((unsigned long long)(unsigned)parkAt(3, j) << 32) | (unsigned long long)(unsigned)parkAt(2, j);
        const int meta = parkAt(4, j);
        const int cntj = scratch ? 1 : (meta & 0xFF), kj = scratch ? 0 : (meta >> 8);
        const float saj = __int_as_float(parkAt(7, j));
        const int sp = (sxy >> 16) * W + (sxy & 0xFFFF);
        // the prefix, one pixel per lane: does it still stand?
        const int myxy = lane < cntj ? entryXY(lstj, lane, sxy) : sxy;
        int2 ow = make_int2(0, 0);
        if (lane < cntj) ow = tx_load_own(&own[(myxy >> 16) * W + (myxy & 0xFFFF)]);
        const int pv = ci ? ow.x : ow.y, cv = ci ? ow.y : ow.x;
        const unsigned long long taken = __builtin_amdgcn_ballot_w64(lane < cntj && (pv < r || cv < r));
        if (taken & 1ull) continue;                        // the seed belongs to a lower rank: no region
        float sdx, sdy;
        int cnt2 = cntj, k2 = kj, bmin = sxy, bmax = sxy, pend0 = r, pendRank0 = r;
        if (taken == 0ull && cntj > 1) {
          sdx = __int_as_float(parkAt(5, j));
          sdy = __int_as_float(parkAt(6, j));
          if (lane > 0 && lane < cntj) {
            int2* w = &own[(myxy >> 16) * W + (myxy & 0xFFFF)];
            pend0 = __hip_atomic_fetch_min(ci ? &w->y : &w->x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pendRank0 = pv;                                // (round 1: the pixel's own rank)
          }
          if (lane < cntj) q[lane] = myxy;
          int mn = lane < cntj ? myxy : sxy, mx = mn;
#pragma unroll
          for (int o = 1; o < CAP; o <<= 1) {
            mn = tx_pk_min_u16(mn, __shfl_xor(mn, o, 64));
            mx = tx_pk_max_u16(mx, __shfl_xor(mx, o, 64));
          }
          bmin = tx_rl(mn, 0);
          bmax = tx_rl(mx, 0);
        } else {
          double sn, cn;
          sincos((double)saj * TX_DEG2RAD, &sn, &cn);
          sdx = (float)cn;
          sdy = (float)sn;
          cnt2 = 1;
          k2 = 0;
          q[0] = sxy;                                       // every lane stores the same value
        }
        unsigned long long noGroup = 0ull;
        if (!growRegion(r, saj, sdx, sdy, cnt2, k2, bmin, bmax, noSeeds, -1, pend0, pendRank0, noGroup, -1, -1)) return;
      }
    }
    return;
  }
  for (int base = 0; base < n; base += 64) {
    const bool valid = base + lane < n;
    if (!useDirty) {                                      // (the dirty list: n <= 64, one row, already there)
      se = make_int2(TX_INF, -1);
      if (valid) se = seedEntry(base + lane);
    }
    bool d = valid;
    if (SPARSE && !useDirty) d = valid && rgDirty[se.x & DL.rmask] == t;
    float4 srec = make_float4(TX_NOTDEF, 0.f, 0.f, 0.f);
    int2 so = make_int2(0, 0);
    if (PACK) {
      // (the seed's angle and its owner word by two small loads: the 16-byte load's result is a 4-register tuple that the allocator
      // would keep — and spill — whole for as long as the angles of the row are in use)
      if (HOT) {
        if (d) {
          const int2 h = tx_load_own(&hot[se.y]);           // (one 8-byte load past the per-CU L1, like the owner pairs of the later rounds)
          srec.x = __int_as_float(h.x);
          srec.w = __int_as_float(h.y);
        }
      } else if (d) {
        srec.x = rec[se.y].x;
        srec.w = __int_as_float(__hip_atomic_load(tx_rec_owner(&rec[se.y]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
    } else if (HOT) {
      if (d) {
        srec.x = __int_as_float(hot[se.y].x);
        so = tx_load_own(&own[se.y]);
      }
    } else if (d) {
      srec = rec[se.y];
      so = tx_load_own(&own[se.y]);
    }
    // (PACK: a seed is alive while nobody has claimed its pixel)
    const bool alive = PACK ? d && __float_as_int(srec.w) < 0 : d && (ci ? so.x : so.y) == se.x && (ci ? so.y : so.x) == se.x;
    const float seedAng = srec.x;
    // region_grow seeds its sums with cos/sin of the unrounded double angle
    scos = 0.f; ssin = 0.f;
    if (alive) {
      double sn, cn;
      sincos((double)seedAng * TX_DEG2RAD, &sn, &cn);
      scos = (float)cn;
      ssin = (float)sn;
    }
    // (the packed (y, x) of the row's seeds, 64 at a time: a scalar division per region costs more than this one per row)
    const unsigned spyv = (unsigned)max(se.y, 0) / (unsigned)W;
    const int sxyv = se.y < 0 ? -1 : (int)((spyv << 16) | ((unsigned)se.y - spyv * (unsigned)W));   // (-1: no pixel packs to it)
    unsigned long long unusedMask = __builtin_amdgcn_ballot_w64(alive);
    // the current group: its members (lanes of this row, as they stood when it formed), what is left of their parked candidates and
    // the candidates' pixels, packed like sxyv; onePix: the seeds of the row that turned out to be regions of one pixel
    unsigned long long grpMembers = 0ull, grpCand = 0ull, onePix = 0ull;
    int grpXY = -1;
    while (unusedMask) {
      const int j = __ffsll((long long)unusedMask) - 1;
      int preOct = -1;
      if (TX_GROUP) {
        if (!((grpMembers >> j) & 1ull)) {
          // ---- a new group: seed j and the next alive seeds of the row, up to TX_GROUP; octet g fetches member g's 8 neighbours.
          // The members hand their pixel and id to their octets through the first words of the queue (no region is being grown).
          const int idx = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(unusedMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)unusedMask, 0u));
          const bool member = ((unusedMask >> lane) & 1ull) != 0ull && idx < TX_GROUP;
          grpMembers = __builtin_amdgcn_ballot_w64(member);
          if (member) {
            q[idx] = sxyv;
            q[8 + idx] = se.x;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          const int msxy = tx_lds_read(&q[pi]), mr = tx_lds_read(&q[8 + pi]);
          const int nx = (msxy & 0xFFFF) + ndx, ny = (msxy >> 16) + ndy;
          const bool ok = pi < __popcll(grpMembers) && nx >= 0 && ny >= 0 && nx < W && ny < H;
          float4 rr = make_float4(TX_NOTDEF, 0.f, 0.f, 0.f);
          int2 oo = make_int2(0, 0);
          if (HOT) {
            if (PACK) {
              if (ok) {
                const int2 h = tx_load_hot8(&hot[ny * W + nx]);
                rr.x = __int_as_float(h.x);
                rr.w = __int_as_float(h.y);
              }
            } else if (ok) {
              const float2 cp = cold[ny * W + nx];
              rr.x = cp.x; rr.y = cp.x; rr.z = cp.y;
              oo = tx_load_own(&own[ny * W + nx]);
            }
            if (APPROX) {
              rr.y = tx_hw_cos_deg(rr.x);
              rr.z = tx_hw_sin_deg(rr.x);
            }
          } else if (PACK) {
            if (ok) rr = tx_load_rec16(&rec[ny * W + nx]);
          } else if (ok) {
            rr = rec[ny * W + nx];
            oo = tx_load_own(&own[ny * W + nx]);
          }
          const int prevv = ci ? oo.x : oo.y, curv = ci ? oo.y : oo.x;
          if (PACK) {
            const int mBin = PACK == 2 ? lzNb1 - (mr >> lzPixBits) : 0;
            const bool def = ok && rr.x != TX_NOTDEF;
            LazyBins mBins{};
            if (PACK == 2) mBins = lazyBins(mBin);
            grpCand = freeMask(__float_as_int(rr.w), mr, mBin, PACK == 2 ? (mr & ((1 << lzPixBits) - 1)) : 0, mBins, ny * W + nx, def);
          } else
          grpCand = __builtin_amdgcn_ballot_w64(ok && rr.x != TX_NOTDEF && !(prevv < mr || curv <= mr));
          grpXY = ok ? ((ny << 16) | nx) : -1;
          park[lane] = __float_as_int(rr.x);
          park[64 + lane] = __float_as_int(rr.y);
          park[128 + lane] = __float_as_int(rr.z);
          park[192 + lane] = prevv;
        }
        preOct = __popcll(grpMembers & ((1ull << j) - 1ull));
      }
      unusedMask &= unusedMask - 1ull;
      if (TX_GROUP && minReg > 1 && (grpCand & (0xFFull << (8 * preOct))) == 0ull) {
        // nothing to take around the seed (as parked, minus what the lower ids of this wave took since): a region of one pixel —
        // no step, no claim; its size and box are stored with the row's other such regions below
        onePix |= 1ull << j;
        continue;
      }
      const int sxy = tx_rl(sxyv, j);
      q[0] = sxy;                                         // every lane stores the same value
      const int rj = tx_rl(se.x, j);
      seedLaneOfRow = j;
      if (!growRegion(rj, tx_rlf(seedAng, j), tx_rlf(scos, j), tx_rlf(ssin, j), 1, 0, sxy, sxy, unusedMask, sxyv, rj, rj,
                      grpCand, grpXY, preOct))
        return;
    }
    if ((onePix >> lane) & 1ull) {
      rgSize[se.x & DL.rmask] = 1;
      rgBox[se.x & DL.rmask] = make_int2(sxyv, sxyv);
    }
  }
}

// k_tx_order: the (image, tile) pairs of a launch of k_tx_grow in order of decreasing seed count (64 buckets of 64 seeds; inside a
// bucket any order), by one workgroup: a histogram, its scan from the top, and a scatter with one cursor per bucket.
#ifdef PLI_DEV     // (dev switch PLI_TX_ORDER: measured, slower)
__global__ __launch_bounds__(1024) void k_tx_order(const int* __restrict__ tileCntAll, int ntile, int nimg, int img0, int* __restrict__ perm) {
  __shared__ int hist[64], cursor[64];
  const int tid = threadIdx.x, n = ntile * nimg;
  if (tid < 64) hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) atomicAdd(&hist[min(63, tileCntAll[(int64_t)img0 * ntile + i] >> 6)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 63; b >= 0; --b) { cursor[b] = run; run += hist[b]; }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) perm[atomicAdd(&cursor[min(63, tileCntAll[(int64_t)img0 * ntile + i] >> 6)], 1)] = i;
}
#endif

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tx_grow(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                const int2* __restrict__ listAll, const int* __restrict__ tileCntAll,
                                                int ts, int ntx, int nty, int* __restrict__ rgSizeAll,
                                                int2* __restrict__ rgBoxAll, const int* __restrict__ rgDirtyAll,
                                                const int* __restrict__ tileActAll, int TW, int TH,
                                                int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,
                                                int rectCap, int img0, int t, const int* __restrict__ rankAll,
                                                int* __restrict__ rgLostAll, int* __restrict__ tileTouchAll, TxDirtyLists DL) {
  __shared__ int q[TX_GQ];
  __shared__ int gb[TX_BMAXBLK];
  __shared__ int park[TX_PARK];
  // (the tiles with the most seeds first, k_tx_order: the waves that take longest start first and the launch does not end on a few of them)
  int img = blockIdx.y, tile = blockIdx.x;
  if (DL.perm) {
    const int v = DL.perm[blockIdx.y * gridDim.x + blockIdx.x];
    img = v / (int)gridDim.x;
    tile = v - img * (int)gridDim.x;
  } else if (DL.xcdAffine) {
    // workgroup L runs on XCD L % 8: the k-th workgroup of an XCD takes tile k % ntile of image 8 * (k / ntile) + xcd, so that the
    // tiles of one image — neighbours share their border pixels' records and owner words — meet in one L2 (gridDim.y % 8 == 0)
    const int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, k = L >> 3;
    img = (k / (int)gridDim.x) * 8 + xcd;
    tile = k % (int)gridDim.x;
  }
  tx_grow_tile<false>(Pp, ctl, recAll, ownAll, listAll, tileCntAll, ts, ntx, nty, rgSizeAll, rgBoxAll, rgDirtyAll, tileActAll, TW, TH,
                   arenaAll, arenaCap, rectAll, rectCap, img + img0, tile, t, rankAll, rgLostAll, tileTouchAll, DL, q, gb, park);
}
// round 1 with owner_1 packed into the pixel records (CV_64F detector; the host picks it, pli_capi.hip)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tx_grow_p1(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                float4* recAll /* written: the claims */, int2* __restrict__ ownAll,
                                                const int2* __restrict__ listAll, const int* __restrict__ tileCntAll,
                                                int ts, int ntx, int nty, int* __restrict__ rgSizeAll,
                                                int2* __restrict__ rgBoxAll, const int* __restrict__ rgDirtyAll,
                                                const int* __restrict__ tileActAll, int TW, int TH,
                                                int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,
                                                int rectCap, int img0, int t, const int* __restrict__ rankAll,
                                                int* __restrict__ rgLostAll, int* __restrict__ tileTouchAll, TxDirtyLists DL, TxKeys keys) {
  __shared__ int q[TX_GQ];
  __shared__ int gb[TX_BMAXBLK];
  __shared__ int park[TX_PARK];
  int img = blockIdx.y, tile = blockIdx.x;
  if (DL.xcdAffine) {                                     // (as in k_tx_grow)
    const int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, k = L >> 3;
    img = (k / (int)gridDim.x) * 8 + xcd;
    tile = k % (int)gridDim.x;
  }
  tx_grow_tile<false, false, TX_GQ, 1>(Pp, ctl, recAll, ownAll, listAll, tileCntAll, ts, ntx, nty, rgSizeAll, rgBoxAll, rgDirtyAll, tileActAll,
                                       TW, TH, arenaAll, arenaCap, rectAll, rectCap, img + img0, tile, 1, rankAll, rgLostAll, tileTouchAll, DL,
                                       q, gb, park);
}
// ... and with LAZY ids (key mode: the front pass wrote the unclaimed words, k_tx_sort touches neither the records nor the owner plane)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tx_grow_p2(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                float4* recAll /* written: the claims */, int2* __restrict__ ownAll,
                                                const int2* __restrict__ listAll, const int* __restrict__ tileCntAll,
                                                int ts, int ntx, int nty, int* __restrict__ rgSizeAll,
                                                int2* __restrict__ rgBoxAll, const int* __restrict__ rgDirtyAll,
                                                const int* __restrict__ tileActAll, int TW, int TH,
                                                int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,
                                                int rectCap, int img0, int t, const int* __restrict__ rankAll,
                                                int* __restrict__ rgLostAll, int* __restrict__ tileTouchAll, TxDirtyLists DL, TxKeys keys) {
  __shared__ int q[TX_GQ];
  __shared__ int gb[TX_BMAXBLK];
  __shared__ int park[TX_PARK];
  int img = blockIdx.y, tile = blockIdx.x;
  if (DL.xcdAffine) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, k = L >> 3;
    img = (k / (int)gridDim.x) * 8 + xcd;
    tile = k % (int)gridDim.x;
  }
  tx_grow_tile<false, false, TX_GQ, 2>(Pp, ctl, recAll, ownAll, listAll, tileCntAll, ts, ntx, nty, rgSizeAll, rgBoxAll, rgDirtyAll, tileActAll,
                                       TW, TH, arenaAll, arenaCap, rectAll, rectCap, img + img0, tile, 1, rankAll, rgLostAll, tileTouchAll, DL,
                                       q, gb, park, &keys);
}
// ... and both with round 1's words in the 8-byte hot records (round 6; keys.hot)
#define TX_GROW_HOT_KERNEL(NAME, PACKV)                                                                                                     \
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void NAME(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl, \
                                                const float4* __restrict__ recAll, int2* __restrict__ ownAll,                                \
                                                const int2* __restrict__ listAll, const int* __restrict__ tileCntAll,                        \
                                                int ts, int ntx, int nty, int* __restrict__ rgSizeAll,                                       \
                                                int2* __restrict__ rgBoxAll, const int* __restrict__ rgDirtyAll,                             \
                                                const int* __restrict__ tileActAll, int TW, int TH,                                          \
                                                int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,                      \
                                                int rectCap, int img0, int t, const int* __restrict__ rankAll,                               \
                                                int* __restrict__ rgLostAll, int* __restrict__ tileTouchAll, TxDirtyLists DL, TxKeys keys) { \
  __shared__ int q[TX_GQ];                                                                                                                   \
  __shared__ int gb[TX_BMAXBLK + 4];        /* (+ the exact sums' three words, tx_grow_tile) */                                              \
  __shared__ int park[TX_PARK];                                                                                                              \
  int img = blockIdx.y, tile = blockIdx.x;                                                                                                   \
  if (DL.xcdAffine) {                                                                                                                        \
    const int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, k = L >> 3;                                                              \
    img = (k / (int)gridDim.x) * 8 + xcd;                                                                                                    \
    tile = k % (int)gridDim.x;                                                                                                               \
  }                                                                                                                                          \
  tx_grow_tile<false, false, TX_GQ, PACKV, true>(Pp, ctl, recAll, ownAll, listAll, tileCntAll, ts, ntx, nty, rgSizeAll, rgBoxAll, rgDirtyAll, \
                                                 tileActAll, TW, TH, arenaAll, arenaCap, rectAll, rectCap, img + img0, tile, 1, rankAll,      \
                                                 rgLostAll, tileTouchAll, DL, q, gb, park, &keys, keys.hot, keys.cold);                       \
}
TX_GROW_HOT_KERNEL(k_tx_grow_h1, 1)
TX_GROW_HOT_KERNEL(k_tx_grow_h2, 2)
#undef TX_GROW_HOT_KERNEL
#ifdef PLI_DEV     // (dev switch PLI_TX_SPEC: round 1 in the speculative lane schedule — exact, measured, slower)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tx_grow_spec(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                const int2* __restrict__ listAll, const int* __restrict__ tileCntAll,
                                                int ts, int ntx, int nty, int* __restrict__ rgSizeAll,
                                                int2* __restrict__ rgBoxAll, const int* __restrict__ rgDirtyAll,
                                                const int* __restrict__ tileActAll, int TW, int TH,
                                                int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,
                                                int rectCap, int img0, int t, const int* __restrict__ rankAll,
                                                int* __restrict__ rgLostAll, int* __restrict__ tileTouchAll, TxDirtyLists DL) {
  __shared__ int q[TX_GQ_SPEC];
  __shared__ int gb[TX_BMAXBLK];
  __shared__ int park[8 * 64];
  tx_grow_tile<false, true, TX_GQ_SPEC>(Pp, ctl, recAll, ownAll, listAll, tileCntAll, ts, ntx, nty, rgSizeAll, rgBoxAll, rgDirtyAll, tileActAll,
                                        TW, TH, arenaAll, arenaCap, rectAll, rectCap, blockIdx.y + img0, blockIdx.x, t, rankAll, rgLostAll,
                                        tileTouchAll, DL, q, gb, park);
}
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tx_grow_sparse(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                       const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                       const int2* __restrict__ listAll, const int* __restrict__ tileCntAll,
                                                       int ts, int ntx, int nty, int* __restrict__ rgSizeAll,
                                                       int2* __restrict__ rgBoxAll, const int* __restrict__ rgDirtyAll,
                                                       const int* __restrict__ tileActAll, int TW, int TH,
                                                       int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,
                                                       int rectCap, int img0, int t, const int* __restrict__ rankAll,
                                                       int* __restrict__ rgLostAll, int* __restrict__ tileTouchAll, TxDirtyLists DL) {
  __shared__ int q[TX_GQ];
  __shared__ int gb[TX_BMAXBLK];
  __shared__ int park[TX_PARK];
  tx_grow_tile<true>(Pp, ctl, recAll, ownAll, listAll, tileCntAll, ts, ntx, nty, rgSizeAll, rgBoxAll, rgDirtyAll, tileActAll, TW, TH,
                   arenaAll, arenaCap, rectAll, rectCap, blockIdx.y + img0, blockIdx.x, t, rankAll, rgLostAll, tileTouchAll, DL, q, gb, park);
}
// ... on the hot records (round 6): the angle from there, the owner pair from the owner plane; recAll is not read
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tx_grow_sparse_h(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                       const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                       const int2* __restrict__ listAll, const int* __restrict__ tileCntAll,
                                                       int ts, int ntx, int nty, int* __restrict__ rgSizeAll,
                                                       int2* __restrict__ rgBoxAll, const int* __restrict__ rgDirtyAll,
                                                       const int* __restrict__ tileActAll, int TW, int TH,
                                                       int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,
                                                       int rectCap, int img0, int t, const int* __restrict__ rankAll,
                                                       int* __restrict__ rgLostAll, int* __restrict__ tileTouchAll, TxDirtyLists DL,
                                                       int2* __restrict__ hotAll, const float2* __restrict__ coldAll) {
  __shared__ int q[TX_GQ];
  __shared__ int gb[TX_BMAXBLK + 4];
  __shared__ int park[TX_PARK];
  tx_grow_tile<true, false, TX_GQ, 0, true>(Pp, ctl, recAll, ownAll, listAll, tileCntAll, ts, ntx, nty, rgSizeAll, rgBoxAll, rgDirtyAll, tileActAll,
                                            TW, TH, arenaAll, arenaCap, rectAll, rectCap, blockIdx.y + img0, blockIdx.x, t, rankAll, rgLostAll,
                                            tileTouchAll, DL, q, gb, park, nullptr, hotAll, coldAll);
}

// ---------------------------------------------------------------------------
// k_tx_rec_from_hot: with every round on the hot records the front pass does not write the 16-byte records.  The sequential grower —
// the device-side fallback of an image that ran out of a capacity or did not settle — reads them: written here, for those images only
// (the workgroups of the others leave at once), from the hot and cold planes: the words the front pass would have written.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tx_rec_from_hot(const RxCtl* __restrict__ ctl, const int2* __restrict__ hotAll,
                                                         const float2* __restrict__ coldAll, float4* __restrict__ recAll, int64_t npix, int img0) {
  const int img = blockIdx.y + img0;
  if (ctl[img].state == 2 && ctl[img].overflow == 0) return;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const float2 cp = coldAll[img * npix + p];
    recAll[img * npix + p] = make_float4(__int_as_float(hotAll[img * npix + p].x), cp.x == TX_NOTDEF ? 0.f : cp.x, cp.y, 0.f);
  }
}

// ---------------------------------------------------------------------------
// k_tx_collect / k_tx_emit_sorted (key mode; take the place of k_rx_count / k_rx_emit, which walk the ordered list): the segments of
// the regions that are alive at the fixed point and large enough, in the order of their ids.  k_tx_collect lists the ids (any
// order; the workgroup gathers them in LDS and takes its places in the image's list with ONE atomic — one per wave cost 7.8 ms at 256
// frames, all on the same word), k_tx_emit_sorted sorts the list of an image — ids are distinct — with one workgroup (bitonic in LDS,
// up to TX_EMIT_CAP = 16384 ids) and writes the first maxSeg segments.  An image with more candidates than that is left to the sequential grower (overflow 6).
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_tx_collect(RxCtl* __restrict__ ctl, const int* __restrict__ idPlaneAll, const int2* __restrict__ ownAll,
                                                    const int* __restrict__ rgSizeAll, int64_t npix, int minReg, int* __restrict__ candAll,
                                                    int* __restrict__ candCnt, int img0, int emitCap) {
  constexpr int CAP = 1024;           // ids a workgroup gathers in LDS before its one atomic on the image's counter
  __shared__ int s_n, s_base;
  __shared__ int s_buf[CAP];
  const int img = blockIdx.y + img0;
  RxCtl& c = ctl[img];
  if (c.state != 2 || c.overflow) return;
  const int64_t base = (int64_t)img * npix;
  const int tid = threadIdx.x, lane = tid & 63;
  int* cand = candAll + (int64_t)img * TX_EMIT_CAP;
  if (tid == 0) s_n = 0;
  __syncthreads();
  // the size plane is streamed sixteen pixels per thread and trip (four 16-byte loads in flight); the few pixels that are — or, stale,
  // once were — seeds of a large enough region go on to the two gathers that decide (their own id, the owner of their pixel)
  const int* rgSize = rgSizeAll + base;
  const bool vec = (npix & 3) == 0 && ((uintptr_t)rgSize & 15) == 0;
  for (int64_t p1 = (int64_t)blockIdx.x * 4096; p1 < npix; p1 += (int64_t)gridDim.x * 4096) {
    int sz[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p1 + u * 1024 + tid * 4;
      int4 v = make_int4(0, 0, 0, 0);
      if (vec && p + 3 < npix) v = *reinterpret_cast<const int4*>(rgSize + p);
      else {
        if (p < npix) v.x = rgSize[p];
        if (p + 1 < npix) v.y = rgSize[p + 1];
        if (p + 2 < npix) v.z = rgSize[p + 2];
        if (p + 3 < npix) v.w = rgSize[p + 3];
      }
      sz[4 * u] = v.x; sz[4 * u + 1] = v.y; sz[4 * u + 2] = v.z; sz[4 * u + 3] = v.w;
    }
    unsigned hits = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) hits |= (sz[j] >= minReg ? 1u : 0u) << j;
    while (hits) {
      const int j = __ffs(hits) - 1;
      hits &= hits - 1u;
      const int64_t p = p1 + (j >> 2) * 1024 + tid * 4 + (j & 3);
      const int id = idPlaneAll[base + p];
      if (id == TX_INF || ownAll[base + p].x != id) continue;
      const int at = atomicAdd(&s_n, 1);
      if (at < CAP) s_buf[at] = id;
      else {                                        // (more than CAP in one workgroup's share: straight to the image's list)
        const int g = atomicAdd(&candCnt[img], 1);
        if (g < emitCap) cand[g] = id;
        else c.overflow = 6;
      }
    }
  }
  __syncthreads();
  const int n = min(s_n, CAP);
  if (tid == 0 && n) s_base = atomicAdd(&candCnt[img], n);
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    if (s_base + i < emitCap) cand[s_base + i] = s_buf[i];
    else c.overflow = 6;
  }
}

template <int KPT>
__device__ __forceinline__ void tx_emit_sort(unsigned* ks, const int* __restrict__ cand, int n, int tid) {
  // (thread t: the keys t*KPT .. t*KPT+KPT-1; the padding sorts to the end)
  unsigned key[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) key[u] = tid * KPT + u < n ? (unsigned)cand[tid * KPT + u] : 0xFFFFFFFFu;
  tx_bitonic_regs<KPT, 1024>(key, ks, tid);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KPT; ++u) ks[tid * KPT + u] = key[u];
  __syncthreads();
}

__global__ __launch_bounds__(1024) void k_tx_emit_sorted(const RxCtl* __restrict__ ctl, const int* __restrict__ candAll, const int* __restrict__ candCnt,
                                                         const float4* __restrict__ rgSegAll, int64_t npix, int rmask, float* __restrict__ segAll,
                                                         int* __restrict__ nSeg, int maxSeg, int img0, int emitCap) {
  extern __shared__ unsigned ks[];
  const int img = blockIdx.x + img0;
  if (ctl[img].state != 2 || ctl[img].overflow) return;
  const int n = min(candCnt[img], emitCap), tid = threadIdx.x;
  const int* cand = candAll + (int64_t)img * TX_EMIT_CAP;
  // the sorting network of k_tx_sort, keys in registers: with n <= 1024 (the usual case) 10 of its 55 stages cross a wave and need
  // the LDS and a barrier; the earlier all-LDS network spent a barrier on every stage (114 us for a single pair)
  if (n <= 1024) tx_emit_sort<1>(ks, cand, n, tid);
  else if (n <= 2048) tx_emit_sort<2>(ks, cand, n, tid);
  else if (n <= 4096) tx_emit_sort<4>(ks, cand, n, tid);
  else if (n <= 8192) tx_emit_sort<8>(ks, cand, n, tid);
  else tx_emit_sort<16>(ks, cand, n, tid);
  float* seg = segAll + (int64_t)img * maxSeg * 4;
  for (int i = tid; i < min(n, maxSeg); i += 1024) {
    const float4 v = rgSegAll[img * npix + (int)(ks[i] & (unsigned)rmask)];
    seg[4 * i + 0] = v.x; seg[4 * i + 1] = v.y; seg[4 * i + 2] = v.z; seg[4 * i + 3] = v.w;
  }
  if (tid == 0) nSeg[img] = min(n, maxSeg);
}

// ---------------------------------------------------------------------------
// k_tx_tail: the rounds t >= t0 of the tile relaxation in ONE persistent launch.  From the fourth round on a round is four
// kernels of almost no work (at 256 frames: 133 000 + 267 000 + 69 000 + 32 000 workgroups that look at a flag and leave;
// a single stereo pair: four launches of a few microseconds of work each), ~10 rounds of them, and the host cannot know the
// last round without looking.  Here a resident grid walks the same blocks — tx_diffmark_block, tx_prep_block, tx_grow_tile,
// rx_rect_wave: the code of the four kernels, unchanged — as VIRTUAL blocks / waves, with a grid barrier where a kernel
// boundary was, and stops by itself when every image is at its fixed point (or out of a capacity).
// The grid must be co-resident: the host sizes it from the occupancy of this kernel on a device it has to itself, and chains
// the tails of one process per device (pli_capi.hip).  A barrier that is not passed within ~2^21 polls (about a second; a barrier
// normally takes microseconds, the longest round of real work milliseconds) — another PROCESS holding the compute units the
// rest of the grid needs, see "Sharing a device" in include/pli_frontend.h — raises the abort word, every block leaves, and the
// images that have not settled take the device-side fallback like any other unsettled image (exact; counted in
// pli_lsd_round_stats out[2]).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool tx_grid_barrier(unsigned* bar, unsigned target) {
  // One release (L2 write-back towards the other XCDs) and one acquire (L1 / stale-L2 invalidation) per BLOCK, by its first wave:
  // the agent-scope fences are the expensive part of a grid barrier on this chip (every wave fencing on both sides of every barrier
  // cost ~140 us per barrier with 2048 waves; the polls are relaxed loads for the same reason).  Each wave first waits for its own
  // stores to be acknowledged by the L2, so that the one write-back covers them; the per-CU L1 is shared by the waves of a block, so
  // the one invalidation serves them all.
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if ((++spins & 63u) == 0u &&
          (__hip_atomic_load(&bar[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > (1u << 21))) {
        __hip_atomic_store(&bar[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_ok = ok;
  }
  __syncthreads();
  asm volatile("s_dcache_inv" ::: "memory");           // (the scalar cache: uniform loads of flags and counters)
  return s_ok != 0;
}

__global__ __launch_bounds__(256) void k_tx_tail(TxTailArgs A) {
  __shared__ int qs[4][TX_GQ];
  __shared__ int gbs[4][TX_BMAXBLK + 4];
  __shared__ int parks[4][TX_PARK];
  __shared__ double sts[4][3][64];
  __shared__ double wcs[4][RX_RECT_CACHE][64];
  __shared__ int ecs[4][RX_RECT_CACHE][64];
  __shared__ int s_live, s_n;
  __shared__ int s_list[256];
  if (A.forceAbort) {                                    // (test switch: what every block does when a barrier times out)
    if (threadIdx.x == 0) __hip_atomic_store(&A.bar[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const DevParams& P = *A.Pp;
  const int W = P.LW, H = P.LH;
  const int64_t npix = (int64_t)W * H;
  const int tid = threadIdx.x, wv = tid >> 6;
  const int G = gridDim.x;
  const int nbx = (W + 31) / 32, nbyA = (A.TH + 7) / 8, nbyB = (H + 31) / 32, ntile = A.ntx * A.nty, ncell = A.TW * A.TH;
  const int64_t nA = (int64_t)nbx * nbyA * A.nimg, nB = (int64_t)nbx * nbyB * A.nimg, nC = (int64_t)ntile * A.nimg;
  const int nwaves = G * 4, gw = blockIdx.x * 4 + wv;
  auto relaxing = [&](int il) -> bool {
    const RxCtl& c = A.ctl[A.img0 + il];
    return c.state != 2 && c.overflow == 0;
  };
  // A block's candidates are the virtual blocks blockIdx.x, blockIdx.x + G, ...: 256 of them are TESTED at a time, one per thread
  // (is there anything to do there at all?), the few that pass are listed in LDS and then run, one after the other, by the whole
  // block.  The tests cost a pass over the cell stamps; the virtual blocks that do something are spread evenly over the grid.
  unsigned epoch = 0;
  for (int t = A.t0; t <= A.maxRounds; ++t) {
    // ---- k_tx_diffmark: the 4 x 8 cell blocks with a cell touched in round t - 1 ----
    for (int il = blockIdx.x * 256 + tid; il < A.nimg; il += G * 256) {          // (what block (0, 0) of the kernel does for its image)
      RxCtl& c = A.ctl[A.img0 + il];
      if (c.state != 2) { c.nSmall = 0; c.nBig = 0; c.nHand = 0; c.nextBig = 0; c.rectArena = 0ull; }
    }
    for (int64_t base = blockIdx.x; base < nA; base += (int64_t)G * 256) {
      if (tid == 0) s_n = 0;
      __syncthreads();
      const int64_t v = base + (int64_t)tid * G;
      if (v < nA) {
        const int il = (int)(v / (nbx * nbyA)), rem = (int)(v - (int64_t)il * (nbx * nbyA));
        const int bx = rem % nbx, by = rem / nbx;
        bool act = false;
        if (relaxing(il)) {
          const int* tt = A.tileTouch + (int64_t)(A.img0 + il) * ncell;
          for (int cy = by * 8; cy < min(by * 8 + 8, A.TH); ++cy)
            for (int cx = bx * 4; cx < min(bx * 4 + 4, A.TW); ++cx) act = act || tt[cy * A.TW + cx] == t - 1;
        }
        if (act) s_list[atomicAdd(&s_n, 1)] = (int)((v - base) / G);
      }
      __syncthreads();
      const int n = s_n;
      for (int i = 0; i < n; ++i) {
        const int64_t va = base + (int64_t)s_list[i] * G;
        const int il = (int)(va / (nbx * nbyA)), rem = (int)(va - (int64_t)il * (nbx * nbyA));
        tx_diffmark_block(A.ctl, A.own, A.rank, A.rgBox, A.rgDirty, A.tileAct, A.tileTouch, W, H, A.TW, A.TH, t, rem % nbx, rem / nbx,
                          A.img0 + il, A.rgLost, A.DL);
        __syncthreads();                                 // (the block's LDS lists are reused by the next virtual block)
      }
    }
    if (!tx_grid_barrier(A.bar, ++epoch * G)) return;
    // ---- k_tx_prep: the fixed point of an image whose round-t flag stayed clear; the 32 x 32 blocks with an active cell ----
    for (int il = blockIdx.x * 256 + tid; il < A.nimg; il += G * 256) {
      RxCtl& c = A.ctl[A.img0 + il];
      if (c.state == 2) continue;
      if (((t & 1) ? c.changedOdd : c.changed) == 0) { c.state = 2; c.rounds = t; }
      else if (t & 1) c.changed = 0;
      else c.changedOdd = 0;
    }
    for (int64_t base = blockIdx.x; base < nB; base += (int64_t)G * 256) {
      if (tid == 0) s_n = 0;
      __syncthreads();
      const int64_t v = base + (int64_t)tid * G;
      if (v < nB) {
        const int il = (int)(v / (nbx * nbyB)), rem = (int)(v - (int64_t)il * (nbx * nbyB));
        const int bx = rem % nbx, by = rem / nbx;
        bool act = false;
        if (relaxing(il)) {
          const int* ta = A.tileAct + (int64_t)(A.img0 + il) * ncell;
          for (int cy = by * 4; cy < min(by * 4 + 4, A.TH); ++cy)
            for (int cx = bx * 4; cx < min(bx * 4 + 4, A.TW); ++cx) act = act || ta[cy * A.TW + cx] == t;
        }
        if (act) s_list[atomicAdd(&s_n, 1)] = (int)((v - base) / G);
      }
      __syncthreads();
      const int n = s_n;
      for (int i = 0; i < n; ++i) {
        const int64_t va = base + (int64_t)s_list[i] * G;
        const int il = (int)(va / (nbx * nbyB)), rem = (int)(va - (int64_t)il * (nbx * nbyB));
        tx_prep_block(A.ctl, A.own, A.rank, A.rgDirty, A.tileAct, W, H, A.TW, A.TH, t, rem % nbx, rem / nbx, A.img0 + il, 0, A.tileTouch, A.DL.rmask);
        __syncthreads();
      }
    }
    if (!tx_grid_barrier(A.bar, ++epoch * G)) return;
    // ---- is anybody still relaxing?  (every block counts the same control words: the answer is the same everywhere) ----
    if (tid == 0) s_live = 0;
    __syncthreads();
    int live = 0;
    for (int i = tid; i < A.nimg; i += 256) live += relaxing(i) ? 1 : 0;
    if (live) atomicAdd(&s_live, live);
    __syncthreads();
    if (s_live == 0) return;
    // ---- k_tx_grow_sparse: one wave per tile that has seeds to regrow ----
    for (int64_t base = blockIdx.x; base < nC; base += (int64_t)G * 256) {
      if (tid == 0) s_n = 0;
      __syncthreads();
      const int64_t v = base + (int64_t)tid * G;
      if (v < nC && A.DL.cnt[(int64_t)A.img0 * ntile + v] > 0 && relaxing((int)(v / ntile))) s_list[atomicAdd(&s_n, 1)] = (int)((v - base) / G);
      __syncthreads();
      const int n = s_n;
      for (int i = wv; i < n; i += 4) {
        const int64_t va = base + (int64_t)s_list[i] * G;
        const int il = (int)(va / ntile);
        if (A.hot)
          tx_grow_tile<true, false, TX_GQ, 0, true>(A.Pp, A.ctl, A.rec, A.own, A.list, A.tileCnt, A.ts, A.ntx, A.nty, A.rgSize, A.rgBox, A.rgDirty,
                                                    A.tileAct, A.TW, A.TH, A.arena, A.arenaCap, A.rects, A.rectCap, A.img0 + il,
                                                    (int)(va - (int64_t)il * ntile), t, A.rank, A.rgLost, A.tileTouch, A.DL, qs[wv], gbs[wv],
                                                    parks[wv], nullptr, A.hot, A.cold);
        else
        tx_grow_tile<true>(A.Pp, A.ctl, A.rec, A.own, A.list, A.tileCnt, A.ts, A.ntx, A.nty, A.rgSize, A.rgBox, A.rgDirty, A.tileAct, A.TW,
                           A.TH, A.arena, A.arenaCap, A.rects, A.rectCap, A.img0 + il, (int)(va - (int64_t)il * ntile), t, A.rank, A.rgLost,
                           A.tileTouch, A.DL, qs[wv], gbs[wv], parks[wv]);
      }
      __syncthreads();
    }
    if (!tx_grid_barrier(A.bar, ++epoch * G)) return;
    // ---- k_rx_rect of the regions this round completed; the dirty lists are consumed: emptied for the next round ----
    for (int64_t v = (int64_t)blockIdx.x * 256 + tid; v < nC; v += (int64_t)G * 256) A.DL.cnt[(int64_t)A.img0 * ntile + v] = 0;
    {
      const int per = max(1, nwaves / A.nimg);            // waves per image (few images), or images per wave (many)
      for (int il = gw / per; il < A.nimg; il += max(1, nwaves / per)) {
        const int img = A.img0 + il;
        const RxCtl& c = A.ctl[img];
        if (c.state == 2 || c.overflow) continue;
        rx_rect_wave(P, c, A.rec + img * npix, A.mg ? A.mg + img * npix : nullptr, A.arena + (int64_t)img * A.arenaCap,
                     A.rects + (int64_t)img * A.rectCap, A.rectCap, A.rgSeg + img * npix, gw % per, per, sts[wv], wcs[wv], ecs[wv], A.DL.rmask,
                     A.hot ? A.hot + img * npix : nullptr, A.cold ? A.cold + img * npix : nullptr);
      }
    }
    if (!tx_grid_barrier(A.bar, ++epoch * G)) return;
  }
}

}  // namespace pli
