// LSD region growing as an incremental rank-ordered relaxation — the parallel form of
// OpenCV lsd.cpp's region_grow loop (refine = NONE), exact at convergence.
//
// Sequential definition (what the CPU path does): seeds are visited in rank order
// (gradient bin descending, raster ascending); an unused seed grows a region over unused
// aligned pixels (8-neighbour BFS, the region angle is re-estimated after every accepted
// pixel); every pixel a region takes is USED for all later seeds.  Let owner[q] be the rank
// of the region that takes q.
//
// Relaxation: keep an estimate owner_t.  In round t every seed r that is alive in
// owner_{t-1} (owner_{t-1}[seed] == r) grows its region ALONE, treating a pixel q as used
// iff owner_{t-1}[q] < r (or a lower rank has already claimed it in this round), and claims
// its pixels with atomicMin(owner_t[q], r); unclaimed pixels fall back to their own rank.
// By induction on rank the lowest-ranked region that is not yet final only reads final data
// and is computed correctly in the next round, for ANY owner_0; hence owner_t == owner_{t-1}
// implies owner_t is the sequential result.
//
// Three things keep the work near the sequential step count (measured: ~6x the sequential
// steps in 13-15 rounds, tens of thousands of independent growers per round):
//  * owner_0 is a guess, not the identity: a pixel is given to the lowest-ranked 3x3
//    neighbour whose own angle accepts it, so round 1 starts ~30 000 seeds instead of 411 000;
//  * clean regions are carried, not regrown: a region that was alive in the last two owner
//    maps and whose tested pixels (bounding box + 1) saw no ownership change involving a lower
//    rank (per 8x8 tile: the minimum rank taking part in a change) would repeat its last run
//    exactly, so its pixels are pre-claimed and the seed is not listed;
//  * the segment of a region (region2rect) is computed whenever the region is grown and kept
//    per rank, so there is no separate emit round.
//
// Small regions are grown one per lane; a lane that reaches RX_HAND pixels hands its state
// (queue, sums, bounding box) to the wave-per-region grower, which also takes the seeds whose
// last run was that large.
#include "kernels.hpp"
#include "device_prims.hpp"
#include "lsd_rect.hpp"
#include <climits>
#include <type_traits>

namespace pli {

constexpr float RX_NOTDEF = -1024.f;
constexpr int RX_INF = LSD_ID_INF;          // rank plane of undefined pixels

__device__ __forceinline__ int2 rx_load_own(const int2* p) {
  // bypass the per-CU L1: claims of other workgroups (and our own atomics) are served from L2 / memory
  unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
  return make_int2((int)(v & 0xFFFFFFFFull), (int)(v >> 32));
}

// LDS read that stays a ds_read: left to itself the compiler merges "queue entry from LDS or from the arena" into
// a pointer select and one flat load, and the wait of a flat load also covers the global claims in flight
__device__ __forceinline__ int rx_lds_read(const int* p) {
  typedef __attribute__((address_space(3))) const volatile int lds_cvint;
  return *(lds_cvint*)p;
}


// Block-wide stream compaction: position of this thread's element in a list whose counter gets ONE atomic per
// block (same-address returning atomics serialise at their L2 channel, ~150 ns each).  lds: 17 ints, up to 1024 threads.
__device__ __forceinline__ int rx_block_append(bool flag, int* counter, int* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(flag);
  if (lane == 0) lds[wv] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < nw; ++w) tot += lds[w];
    lds[16] = tot ? atomicAdd(counter, tot) : 0;
  }
  __syncthreads();
  int pos = lds[16] + __popcll(bal & ((1ull << lane) - 1ull));
  for (int w = 0; w < wv; ++w) pos += lds[w];
  __syncthreads();
  return pos;
}

// Everything from here to k_rx_rect is the LANE relaxation (lsd_mode 1: round 1's schedule, one region per lane / lane group, and its
// bookkeeping passes, which the tile relaxation's unfused dev switches also launch): kept for cross-checks, compiled into the
// development build only (make dev, -DPLI_DEV).  The product library has k_rx_rect / k_rx_count / k_rx_emit of this file.
#ifdef PLI_DEV
// ---- setup -------------------------------------------------------------------------------------
// (the rank plane — rank of every pixel, LSD_ID_INF where undefined — is written by k_lsd_scatter, line_kernels.hip)

// owner_0: the lowest-ranked 3x3 neighbour whose own angle accepts the pixel (a heuristic: any owner_0 is valid)
__global__ __launch_bounds__(256) void k_rx_guess(const float4* __restrict__ recAll, const int* __restrict__ rankAll,
                                                  int2* __restrict__ ownAll, int W, int H, float precDeg, int img0) {
  const int img = blockIdx.z + img0;
  const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int64_t base = (int64_t)img * W * H;
  const int p = y * W + x;
  const int r = rankAll[base + p];
  if (r == RX_INF) return;                         // own stays (INT_MAX, INT_MAX) from k_lsd_grad
  const float a = recAll[base + p].x;
  int best = r;
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    if (m == 4) continue;
    const int nx = x + m % 3 - 1, ny = y + m / 3 - 1;
    if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
    const int q = ny * W + nx;
    const int rq = rankAll[base + q];
    if (rq >= best) continue;
    float d = fabsf(recAll[base + q].x - a);
    if (d > 270.f) d = fabsf(d - 360.f);
    if (d <= precDeg) best = rq;
  }
  ownAll[base + p] = make_int2(best, best);
}

// ---- round bookkeeping -------------------------------------------------------------------------
// owner_{t-1} against owner_{t-2}: per 8x8 tile the lowest rank that takes part in a change
// (RX_DIFF_ROWS rows of 8x8 cells per workgroup: with one row the grid is a million blocks of a few instructions at 256 frames
// and the kernel is bound by the rate at which workgroups are dispatched)
constexpr int RX_DIFF_ROWS = 8;
__global__ __launch_bounds__(256) void k_rx_diff(RxCtl* __restrict__ ctl, const int2* __restrict__ ownAll,
                                                 int* __restrict__ tileMinAll, int* __restrict__ tileActAll, int W, int H,
                                                 int TW, int TH, int t, int img0, const int* __restrict__ tileTouchAll) {
  __shared__ int tmin[RX_DIFF_ROWS][4];
  __shared__ int s_cell[RX_DIFF_ROWS][4];
  __shared__ int s_rel;
  const int img = blockIdx.z + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { c.nSmall = 0; c.nBig = 0; c.nHand = 0; c.nextBig = 0; c.rectArena = 0ull; }
  if (tid < 4 * RX_DIFF_ROWS) s_cell[tid >> 2][tid & 3] = 1;
  if (tileTouchAll) {
    // tile-sequential relaxation, later rounds: the two owner components of a cell can only differ where round t-1 wrote one of
    // them — k_tx_prep rewrites the cells with tileAct == t-1, the growers note the cells of their claims in tileTouch —
    // everywhere else the carried owner stands in both: only those cells are read (a block without one leaves at once)
    if (tid == 0) s_rel = 0;
    __syncthreads();
    if (tid < 4 * RX_DIFF_ROWS) {
      const int tx = blockIdx.x * 4 + (tid & 3), ty = blockIdx.y * RX_DIFF_ROWS + (tid >> 2);
      int rel = 0;
      if (tx < TW && ty < TH) {
        const int64_t cell = (int64_t)img * TW * TH + ty * TW + tx;
        rel = tileActAll[cell] == t - 1 || tileTouchAll[cell] == t - 1;
      }
      s_cell[tid >> 2][tid & 3] = rel;
      if (rel) s_rel = 1;
    }
    __syncthreads();
    if (!s_rel) {
      if (tid < 4 * RX_DIFF_ROWS) {
        const int tx = blockIdx.x * 4 + (tid & 3), ty = blockIdx.y * RX_DIFF_ROWS + (tid >> 2);
        if (tx < TW && ty < TH) tileMinAll[(int64_t)img * TW * TH + ty * TW + tx] = INT_MAX;
      }
      return;
    }
  }
  if (tid < 4 * RX_DIFF_ROWS) tmin[tid >> 2][tid & 3] = INT_MAX;
  __syncthreads();
  const int x = blockIdx.x * 32 + (tid & 31);
  bool ch = false;
#pragma unroll
  for (int rr = 0; rr < RX_DIFF_ROWS; ++rr) {
    const int y = (blockIdx.y * RX_DIFF_ROWS + rr) * 8 + (tid >> 5);
    if (x < W && y < H && s_cell[rr][(tid & 31) >> 3]) {
      const int2 o = ownAll[(int64_t)img * W * H + y * W + x];
      if (o.x != o.y) { ch = true; atomicMin(&tmin[rr][(tid & 31) >> 3], min(o.x, o.y)); }
    }
  }
  const int any = __syncthreads_or(ch ? 1 : 0);
  if (tid < 4 * RX_DIFF_ROWS) {
    const int tx = blockIdx.x * 4 + (tid & 3), ty = blockIdx.y * RX_DIFF_ROWS + (tid >> 2);
    if (tx < TW && ty < TH) {
      const int v = tmin[tid >> 2][tid & 3];
      tileMinAll[(int64_t)img * TW * TH + ty * TW + tx] = v;
      if (v != INT_MAX) tileActAll[(int64_t)img * TW * TH + ty * TW + tx] = t;   // a changed tile is active
    }
  }
  if (any && tid == 0) atomicOr(&c.changed, 1);
}

// fixed point test, and which alive regions would repeat their last run exactly: those get this round's stamp
// in rgClean (everything else keeps an older stamp)
__global__ __launch_bounds__(256) void k_rx_classify(RxCtl* __restrict__ ctl, const int2* __restrict__ ownAll,
                                                     const int* __restrict__ rankAll, const int2* __restrict__ rgBoxAll,
                                                     uint8_t* __restrict__ rgCleanAll, const int* __restrict__ tileMinAll,
                                                     int W, int H, int TW, int TH, int t, int img0) {
  const int img = blockIdx.z + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  if (c.changed == 0) {                              // owner_{t-1} == owner_{t-2}: exact (every block sees the same flag)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { c.state = 2; c.rounds = t; }
    return;
  }
  const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int64_t base = (int64_t)img * W * H;
  const int p = y * W + x;
  const int2 o = ownAll[base + p];
  if (o.x != o.y) return;
  const int r = rankAll[base + p];
  if (r == RX_INF || o.x != r) return;               // (undefined pixels)
  // alive in both maps: grown (or carried) in round t-1
  const int2 b = rgBoxAll[base + r];
  const int tx0 = max((b.x & 0xFFFF) - 1, 0) >> 3, ty0 = max((b.x >> 16) - 1, 0) >> 3;
  const int tx1 = min((b.y & 0xFFFF) + 1, W - 1) >> 3, ty1 = min((b.y >> 16) + 1, H - 1) >> 3;
  const int* tm = tileMinAll + (int64_t)img * TW * TH;
  bool clean = true;
  for (int ty = ty0; ty <= ty1 && clean; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx)
      if (tm[ty * TW + tx] < r) { clean = false; break; }
  if (clean) rgCleanAll[base + r] = (uint8_t)t;
}

// owner_t starts as "carried regions keep their pixels, everything else falls back to its own rank";
// the alive seeds that have to be regrown are listed as ready-to-run records
__global__ __launch_bounds__(1024) void k_rx_seed(RxCtl* __restrict__ ctl, int2* __restrict__ ownAll,
                                                 const int* __restrict__ rankAll, const float4* __restrict__ recAll,
                                                 const int* __restrict__ rgSizeAll,
                                                 const uint8_t* __restrict__ rgCleanAll, RxSeed* __restrict__ smallAll,
                                                 RxSeed* __restrict__ bigAll, int bigCap, int W, int H, int bigThresh,
                                                 int t, int img0) {
  __shared__ int scan[17];
  const int img = blockIdx.z + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) c.changed = 0;
  const int y = blockIdx.y, x = blockIdx.x * 1024 + threadIdx.x;
  const int64_t npix = (int64_t)W * H;
  const int64_t base = (int64_t)img * npix;
  const int p = y * W + x;
  const int ci = t & 1;
  const uint8_t stamp = (uint8_t)t;
  bool alive = false, big = false;
  RxSeed sd;
  if (x < W) {
    const int r = rankAll[base + p];
    if (r != RX_INF) {
      int2 o = ownAll[base + p];
      const int prevv = ci ? o.x : o.y;
      const bool carried = t >= 2 && rgCleanAll[base + prevv] == stamp;
      const int cur = carried ? prevv : r;
      if (cur != (ci ? o.y : o.x)) {
        if (ci) o.y = cur; else o.x = cur;
        ownAll[base + p] = o;
      }
      alive = prevv == r && !carried;
      if (alive) {
        big = t >= 2 && rgSizeAll[base + r] >= bigThresh;
        const float ang = recAll[base + p].x;
        // region_grow seeds its sums with cos/sin of the unrounded double angle (the per-pixel c,s of rec are of
        // the float-rounded angle)
        double sn, cn;
        sincos((double)ang * RX_DEG2RAD, &sn, &cn);
        sd.rank = r; sd.xy = (y << 16) | x; sd.ang = ang; sd.sx = (float)cn; sd.sy = (float)sn;
      }
    }
  }
  const int ps = rx_block_append(alive && !big, &c.nSmall, scan);
  if (alive && !big) smallAll[base + ps] = sd;
  if (t >= 2) {
    const int pb = rx_block_append(alive && big, &c.nBig, scan);
    if (alive && big) {
      if (pb < bigCap) bigAll[(int64_t)img * bigCap + pb] = sd;
      else c.overflow = 2;                           // (codes: 1 hand-over list, 2 large-seed list, 3 arena, 4 rect list, 5 queue blocks)
    }
  }
}

// ---- rounds >= 3: bookkeeping restricted to where something happened ------------------------------------------
// From round 3 on most of the image is settled.  k_rx_mark looks only at tiles next to a change: a region with a
// pixel within one pixel of a tile whose change involves a lower rank may see different inputs and is stamped
// dirty (this is the exact footprint, tighter than the bounding-box rule of k_rx_classify), as are the regions whose
// seed changed hands (died or newly alive); the tiles under the bounding box of a dirty region become active.
// k_rx_seed_sparse then rewrites owner_t only in active tiles: everywhere else owner_{t-2} == owner_{t-1} and the
// owner is carried, i.e. the word that is already there is the right start value.
__device__ __forceinline__ void rx_mark_dirty(int o, bool withBox, int t, int* __restrict__ rgDirty,
                                              const int2* __restrict__ rgBox, int* __restrict__ tileAct, int TW, int TH,
                                              const TxDirtyLists& DL, int img) {
  if (rgDirty[o] == t) return;
  if (atomicExch(&rgDirty[o], t) == t) return;      // one marker per region activates the tiles
  tx_dirty_append(DL, img, o);
  if (!withBox) return;
  const int2 b = rgBox[o];
  const int tx0 = min(max((b.x & 0xFFFF) >> 3, 0), TW - 1), ty0 = min(max((b.x >> 16) >> 3, 0), TH - 1);
  const int tx1 = min(max((b.y & 0xFFFF) >> 3, 0), TW - 1), ty1 = min(max((b.y >> 16) >> 3, 0), TH - 1);
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) tileAct[ty * TW + tx] = t;
}

__global__ __launch_bounds__(256) void k_rx_mark(RxCtl* __restrict__ ctl, const int2* __restrict__ ownAll,
                                                 const int* __restrict__ rankAll, const int2* __restrict__ rgBoxAll,
                                                 int* __restrict__ rgDirtyAll, const int* __restrict__ tileMinAll,
                                                 int* __restrict__ tileActAll, int W, int H, int TW, int TH, int t, int img0,
                                                 int seedRule, const int* __restrict__ rgLostAll, TxDirtyLists DL) {
  // (RX_DIFF_ROWS rows of cells per workgroup, like k_rx_diff: the grid of one-row blocks is bound by the dispatch rate)
  __shared__ int nt[RX_DIFF_ROWS + 2][6];
  __shared__ int s_any[RX_DIFF_ROWS];
  const int img = blockIdx.z + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  if (c.changed == 0) {                              // owner_{t-1} == owner_{t-2}: exact (every block sees the same flag)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { c.state = 2; c.rounds = t; }
    return;
  }
  const int tid = threadIdx.x;
  const int* tm = tileMinAll + (int64_t)img * TW * TH;
  if (tid < RX_DIFF_ROWS) s_any[tid] = 0;
  __syncthreads();
  if (tid < 6 * (RX_DIFF_ROWS + 2)) {
    const int ty = (int)blockIdx.y * RX_DIFF_ROWS + tid / 6 - 1, tx = (int)blockIdx.x * 4 + tid % 6 - 1;
    const int v = (ty >= 0 && ty < TH && tx >= 0 && tx < TW) ? tm[ty * TW + tx] : INT_MAX;
    nt[tid / 6][tid % 6] = v;
    if (v != INT_MAX) {
      if (rgLostAll) {                               // exact rule: evaluated from the changed pixels, i.e. inside the changed cells only
        const int rr = tid / 6 - 1, cc = tid % 6;
        if (rr >= 0 && rr < RX_DIFF_ROWS && cc >= 1 && cc <= 4) s_any[rr] = 1;
      } else {                                       // a change in row tid/6 concerns the cell rows tid/6 - 1 .. tid/6 + 1 (block-local: -2 .. 0)
        for (int d = -2; d <= 0; ++d) {
          const int rr = tid / 6 + d;
          if (rr >= 0 && rr < RX_DIFF_ROWS) s_any[rr] = 1;
        }
      }
    }
  }
  __syncthreads();
  const int64_t base = (int64_t)img * W * H;
  int* rgDirty = rgDirtyAll + base;
  const int2* rgBox = rgBoxAll + base;
  int* tileAct = tileActAll + (int64_t)img * TW * TH;
  const int ci = t & 1;
  for (int rr = 0; rr < RX_DIFF_ROWS; ++rr) {
    if (!s_any[rr]) continue;                        // nothing changed in or next to these four tiles
    const int x = blockIdx.x * 32 + (tid & 31), y = (blockIdx.y * RX_DIFF_ROWS + rr) * 8 + (tid >> 5);
    if (x >= W || y >= H) continue;
    const int lx = tid & 31, ly = tid >> 5;
    if (rgLostAll && nt[rr + 1][1 + (lx >> 3)] == INT_MAX) continue;      // (exact rule) no pixel of this cell changed
    const int2 o = ownAll[base + y * W + x];
    const int prevv = ci ? o.x : o.y, prev2 = ci ? o.y : o.x;
    if (rgLostAll && prevv == prev2) continue;       // (exact rule) everything below starts from a changed pixel
    const int r = rankAll[base + y * W + x];
    if (r == RX_INF) continue;
    // lowest rank of a change in the tiles within one pixel of (x, y)
    const int cx0 = (lx + 7) >> 3, cx1 = (lx + 9) >> 3;        // nt column of x-1 and x+1 (nt column 1 = first own tile)
    const int cy0 = rr + ((ly + 7) >> 3), cy1 = rr + ((ly + 9) >> 3);
    const int m = min(min(nt[cy0][cx0], nt[cy0][cx1]), min(nt[cy1][cx0], nt[cy1][cx1]));
    if (rgLostAll) {
      // Exact rule (tile-sequential relaxation).  A region is regrown iff
      //   * it lost a contested claim in round t-1 (the growers stamp the loser), or a pixel next to one of its own was its own in
      //     owner_{t-2} and belongs to a lower rank now (it lost that pixel across the rounds), or
      //   * a pixel next to one of its own was held by a lower rank in owner_{t-2} and is not any more (released, or passed to a
      //     higher rank): the region may take it now.
      // All three start from a pixel whose owner CHANGED, so they are evaluated from that pixel's side: a changed (x, y) looks at
      // its eight neighbours and stamps their owners (a few thousand pixels per image and round do so).
      // (The 8x8-cell rule below — any change that involves a lower rank within a pixel — regrew ten times as many regions;
      // tools/sim/sim_tile_relax.cpp replays both: SIM_CARRY=1 SIM_LOST=1 SIM_EXACT=1 SIM_GPURULE=1.)
      if (prevv != prev2) {
        int lastOp = INT_MAX;                          // (neighbours mostly share their owner: a rank is looked at once in a row)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if (k == 4) continue;
          const int px = x + k % 3 - 1, py = y + k / 3 - 1;
          if (px < 0 || py < 0 || px >= W || py >= H) continue;
          const int2 op2 = ownAll[base + py * W + px];
          const int op = ci ? op2.x : op2.y;           // owner_{t-1} of the neighbour
          if (op == INT_MAX || op == lastOp) continue;
          lastOp = op;
          if ((prev2 < op && prevv > op) || (prev2 == op && prevv < op) || rgLostAll[base + op] == t - 1)
            rx_mark_dirty(op, true, t, rgDirty, rgBox, tileAct, TW, TH, DL, img);
        }
      }
    } else if (m < prevv) rx_mark_dirty(prevv, true, t, rgDirty, rgBox, tileAct, TW, TH, DL, img);
    const bool a1 = prevv == r, a2 = prev2 == r;
    // (seedRule 0: round 2 of the tile-sequential relaxation, where owner_{t-2} is the trivial map and the seeds that died in
    // round 1 never ran)
    if (seedRule && a1 != a2) rx_mark_dirty(r, a2, t, rgDirty, rgBox, tileAct, TW, TH, DL, img);   // died: its last box; newly alive: only this pixel
  }
}

// (32 x 32 pixels = 16 tiles per block: one list atomic per 1024 pixels)
__global__ __launch_bounds__(1024) void k_rx_seed_sparse(RxCtl* __restrict__ ctl, int2* __restrict__ ownAll,
                                                        const int* __restrict__ rankAll, const float4* __restrict__ recAll,
                                                        const int* __restrict__ rgSizeAll, const int* __restrict__ rgDirtyAll,
                                                        const int* __restrict__ tileActAll, RxSeed* __restrict__ smallAll,
                                                        RxSeed* __restrict__ bigAll, int bigCap, int W, int H, int TW, int TH,
                                                        int bigThresh, int t, int img0) {
  __shared__ int scan[17];
  __shared__ int s_act;
  const int img = blockIdx.z + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2) return;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) c.changed = 0;
  if (tid == 0) s_act = 0;
  __syncthreads();
  if (tid < 16) {
    const int tx = blockIdx.x * 4 + (tid & 3), ty = blockIdx.y * 4 + (tid >> 2);
    if (tx < TW && ty < TH && tileActAll[(int64_t)img * TW * TH + ty * TW + tx] == t) s_act = 1;
  }
  __syncthreads();
  if (!s_act) return;
  const int x = blockIdx.x * 32 + (tid & 31), y = blockIdx.y * 32 + (tid >> 5);
  const int64_t npix = (int64_t)W * H;
  const int64_t base = (int64_t)img * npix;
  const int p = y * W + x;
  const int ci = t & 1;
  bool alive = false, big = false;
  RxSeed sd;
  if (x < W && y < H) {
    const int r = rankAll[base + p];
    if (r != RX_INF) {
      int2 o = ownAll[base + p];
      const int prevv = ci ? o.x : o.y;
      const bool carried = rgDirtyAll[base + prevv] != t;
      const int cur = carried ? prevv : r;
      if (cur != (ci ? o.y : o.x)) {
        if (ci) o.y = cur; else o.x = cur;
        ownAll[base + p] = o;
      }
      alive = prevv == r && !carried;
      if (alive) {
        big = rgSizeAll[base + r] >= bigThresh;
        const float ang = recAll[base + p].x;
        double sn, cn;
        sincos((double)ang * RX_DEG2RAD, &sn, &cn);   // see k_rx_seed
        sd.rank = r; sd.xy = (y << 16) | x; sd.ang = ang; sd.sx = (float)cn; sd.sy = (float)sn;
      }
    }
  }
  const int ps = rx_block_append(alive && !big, &c.nSmall, scan);
  if (alive && !big) smallAll[base + ps] = sd;
  const int pb = rx_block_append(alive && big, &c.nBig, scan);
  if (alive && big) {
    if (pb < bigCap) bigAll[(int64_t)img * bigCap + pb] = sd;
    else c.overflow = 2;
  }
}

// ---- lane-per-region grower --------------------------------------------------------------------
// A wave takes 64 seed records at a time (one atomic per wave, one coalesced load per lane); the serial
// BFS of a lane is the CPU loop verbatim with its queue in LDS.  At RX_HAND pixels the lane stops at a
// step boundary and hands the region over to the wave-per-region grower.
__global__ __launch_bounds__(256) void k_rx_grow(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                 const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                 const RxSeed* __restrict__ smallAll, int* __restrict__ rgSizeAll,
                                                 int2* __restrict__ rgBoxAll, RxHand* __restrict__ handAll, int handCap,
                                                 int* __restrict__ arenaAll, int arenaCap, RxRect* __restrict__ rectAll,
                                                 int rectCap, int img0, int t) {
  __shared__ int mq[RX_QCAP * 256];
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2 || c.overflow) return;
  const int nlive = c.nSmall;
  if ((int)blockIdx.x * 256 >= nlive) return;      // more lanes than seeds
  const int W = P.LW, H = P.LH;
  const int64_t npix = (int64_t)W * H;
  const float4* rec = recAll + img * npix;
  int2* own = ownAll + img * npix;
  const RxSeed* seeds = smallAll + img * npix;
  int* rgSize = rgSizeAll + img * npix;
  int2* rgBox = rgBoxAll + img * npix;
  RxHand* hand = handAll + (int64_t)img * handCap;
  int* arena = arenaAll + (int64_t)img * arenaCap;
  RxRect* rects = rectAll + (int64_t)img * rectCap;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ci = t & 1;                              // owner_t lives in component ci, owner_{t-1} in the other
  const double prec = P.prec;
  const int minReg = P.minRegSize;
  int batch = 0;

  for (;;) {
    // batches of 64 seeds are dealt to the waves round robin (no work counter: a returning atomic per batch on a
    // per-image counter costs more than the imbalance of these short regions)
    const int base = (batch * (int)(gridDim.x * 4) + (int)blockIdx.x * 4 + (tid >> 6)) * 64;
    ++batch;
    if (base >= nlive) break;
    bool active = base + lane < nlive;
    int r = 0, cnt = 0, k = 0;
    int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
    int mark0 = 1, mark1 = 1;
    int pend[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) pend[n] = 0x7FFFFFFF;
    bool raced = false;
    float sumdx = 0.f, sumdy = 0.f;
    double reg_angle = 0.0;
    if (active) {
      const RxSeed sd = seeds[base + lane];
      r = sd.rank;
      reg_angle = (double)sd.ang * RX_DEG2RAD;
      sumdx = sd.sx;
      sumdy = sd.sy;
      mq[tid] = sd.xy;
      bx0 = bx1 = sd.xy & 0xFFFF;
      by0 = by1 = sd.xy >> 16;
      cnt = 1;
    }
    while (__builtin_amdgcn_ballot_w64(active)) {
      if (!active) continue;
      // ---- one BFS step of this lane's region ----------------------------------------
      const int xy = mq[k * 256 + tid];
      const int px = xy & 0xFFFF, py = xy >> 16;
      float4 nr[8];
      int2 no[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int m = n < 4 ? n : n + 1;               // skip the centre of the 3x3 block
        const int nx = px + m % 3 - 1, ny = py + m / 3 - 1;
        const bool inb = nx >= 0 && ny >= 0 && nx < W && ny < H;
        nr[n] = make_float4(RX_NOTDEF, 0.f, 0.f, 0.f);
        no[n] = make_int2(0, 0);
        if (inb) {
          const int q = ny * W + nx;
          nr[n] = rec[q];
          no[n] = rx_load_own(&own[q]);
        }
      }
      // The claims are returning atomics whose results are only folded here, one step later: they return
      // in issue order with the loads above, so a claim of step k-1 is performed before the loads of step
      // k+1 are issued; the loads of step k may still miss it, hence the look at the last two steps' queue
      // entries below.  A lower rank that slipped in between our load and our claim is ignored: such a
      // region is not final in this round anyway (the lowest non-final region never sees that race,
      // because the pixels of final lower ranks are already theirs in owner_{t-1}).
#pragma unroll
      for (int n = 0; n < 8; ++n) { raced = raced || pend[n] <= r; pend[n] = 0x7FFFFFFF; }
      const int recent0 = mark1;                        // queue entries [recent0, cnt) were claimed in the last two steps
      mark1 = mark0;
      mark0 = cnt;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        if (nr[n].x == RX_NOTDEF) continue;
        const int prevv = ci ? no[n].x : no[n].y;
        const int curv = ci ? no[n].y : no[n].x;
        if (prevv < r || curv <= r) continue;          // taken by a lower rank (last round / this round) or already mine
        double n_theta = fabs(reg_angle - (double)nr[n].x * RX_DEG2RAD);
        if (n_theta > RX_3_2_PI) {
          n_theta = fabs(n_theta - RX_2PI);
        }
        if (!(n_theta <= prec)) continue;
        const int m = n < 4 ? n : n + 1;
        const int nx = px + m % 3 - 1, ny = py + m / 3 - 1;
        const int e = (ny << 16) | nx;
        bool mine = false;
        for (int i = recent0; i < cnt; ++i) mine = mine || mq[i * 256 + tid] == e;
        if (mine) continue;                            // claimed a moment ago, the owner load did not see it yet
        const int q = ny * W + nx;
        pend[n] = __hip_atomic_fetch_min(ci ? &own[q].y : &own[q].x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mq[cnt * 256 + tid] = e;                       // cnt < RX_HAND + 8 <= RX_QCAP at a step boundary
        ++cnt;
        bx0 = min(bx0, nx); bx1 = max(bx1, nx); by0 = min(by0, ny); by1 = max(by1, ny);
        sumdx = __fadd_rn(sumdx, nr[n].y);
        sumdy = __fadd_rn(sumdy, nr[n].z);
        reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * RX_DEG2RAD;
      }
      ++k;
      if (k < cnt && cnt < RX_HAND) continue;
#pragma unroll
      for (int n = 0; n < 8; ++n) raced = raced || pend[n] <= r;
      if (raced) atomicAdd(&c.races, 1);                 // statistics; also what keeps the claims 'returning'
      if (k < cnt) {
        // ---- too large for a lane: hand the state over ---------------------------------
        active = false;
        const int slot = atomicAdd(&c.nHand, 1);
        if (slot >= handCap) { c.overflow = 1; continue; }
        RxHand& hd = hand[slot];
        hd.rank = r; hd.k = k; hd.cnt = cnt; hd.sumdx = sumdx; hd.sumdy = sumdy;
        hd.box0 = (by0 << 16) | bx0; hd.box1 = (by1 << 16) | bx1; hd.pad = 0;
        for (int i = 0; i < cnt; ++i) hd.q[i] = mq[i * 256 + tid];
        continue;
      }
      // ---- the region is complete -------------------------------------------------------
      active = false;
      rgSize[r] = cnt;
      rgBox[r] = make_int2((by0 << 16) | bx0, (by1 << 16) | bx1);
      if (cnt < minReg) continue;
      // the pixel list goes to k_rx_rect (region2rect)
      const unsigned long long ra = atomicAdd(&c.rectArena, (1ull << RX_ARENA_BITS) | (unsigned long long)cnt);
      const long long off = (long long)(ra & ((1ull << RX_ARENA_BITS) - 1ull));
      const int slot = (int)(ra >> RX_ARENA_BITS);
      if (off + cnt > arenaCap || slot >= rectCap) { c.overflow = off + cnt > arenaCap ? 3 : 4; continue; }
      for (int i = 0; i < cnt; ++i) arena[off + i] = mq[i * 256 + tid];
      RxRect& it = rects[slot];
      it.rank = r; it.off = off; it.cnt = cnt; it.sumdx = sumdx; it.sumdy = sumdy; it.approx = 0;
    }
  }
}

// ---- group-per-region grower ------------------------------------------------------------------------
// A wave carries eight regions, one per group of 8 lanes (the issue slots of a wave are the scarce
// resource here, and a BFS step has exactly 8 neighbours to look at).  The lanes of a group fetch the 3x3
// neighbourhood (centre skipped) of the group's current queue entry (record + owner pair) in one round trip; ballots over
// "available & aligned", shifted to the group, reproduce the raster-order accept loop.  The queue of a
// group lives in LDS (first RX_GQ entries) and in arena blocks beyond.  Work items: the seeds listed as
// large, then the regions handed over by the lane grower in this round.  Everything that is uniform per
// group (r, cnt, k, sums, angle) is replicated in the group's lanes.
constexpr int RX_BBLK = 256;      // arena block for the overflow of a large region's queue
constexpr int RX_BMAXBLK = 128;   // => regions of up to RX_GQ + 32768 pixels

// RX_GL lanes per group (8, 16 or 64 = one region per wave), RX_GQ LDS queue entries per group.
template <int RX_GL, int RX_GQ>
__device__ __forceinline__ void rx_grow_groups(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                               const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                               const RxSeed* __restrict__ bigAll, int bigCap,
                                               const RxHand* __restrict__ handAll, int handCap,
                                               int* __restrict__ rgSizeAll, int2* __restrict__ rgBoxAll,
                                               int* __restrict__ arenaAll, int arenaCap,
                                               RxRect* __restrict__ rectAll, int rectCap, int img0, int t) {
  constexpr int RX_NG = 64 / RX_GL;   // groups (regions) per wave
  __shared__ int qs[RX_NG][RX_GQ];
  __shared__ int blk[RX_NG][RX_BMAXBLK];
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0;
  RxCtl& c = ctl[img];
  if (c.state == 2 || c.overflow) return;
  const int nbig = min(c.nBig, bigCap);
  const int nitems = nbig + min(c.nHand, handCap);
  if ((int)blockIdx.x * RX_NG >= nitems) return;
  const int W = P.LW, H = P.LH;
  const int64_t npix = (int64_t)W * H;
  const float4* rec = recAll + img * npix;
  int2* own = ownAll + img * npix;
  const RxSeed* seeds = bigAll + (int64_t)img * bigCap;
  const RxHand* hand = handAll + (int64_t)img * handCap;
  int* rgSize = rgSizeAll + img * npix;
  int2* rgBox = rgBoxAll + img * npix;
  int* arena = arenaAll + (int64_t)img * arenaCap;
  RxRect* rects = rectAll + (int64_t)img * rectCap;
  const int lane = threadIdx.x, g = lane / RX_GL, gl = lane % RX_GL, gbase = RX_GL == 64 ? 0 : lane - gl;
  // with one group per wave the group-uniform state is wave-uniform: say so (scalar registers, scalar branches)
  auto U = [](int v) -> int { if constexpr (RX_GL == 64) return __builtin_amdgcn_readfirstlane(v); else return v; };
  auto Uf = [](float v) -> float { if constexpr (RX_GL == 64) return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); else return v; };
  auto bcast = [&](float v, int j) -> float {   // value of lane j of this group
    if constexpr (RX_GL == 64) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); else return __shfl(v, gbase + j, 64);
  };
  // Batched steps (as in k_lsd_grow): a group pops up to NP queue entries at a time, 8 lanes per entry (the 3x3
  // neighbours in raster order, centre skipped); lane order inside the group = the order of the sequential tests.
  constexpr int NP = RX_GL / 8;
  constexpr unsigned long long GMASK = RX_GL == 64 ? ~0ull : ((1ull << RX_GL) - 1ull);
  const int pi = gl >> 3;
  const int nm = (gl & 7) < 4 ? (gl & 7) : (gl & 7) + 1;
  const int ci = t & 1;
  const double prec = P.prec;
  const int minReg = P.minRegSize;
  const int ndx = nm % 3 - 1, ndy = nm / 3 - 1;
  int* q = qs[g];
  int* gb = blk[g];

  // the LDS read is unconditional (clamped index) so that it stays a ds_read: a pointer select between LDS and
  // the arena would become a flat load, whose wait also covers the claims in flight
  auto qget = [&](int k) -> int {
    int e = rx_lds_read(&q[min(k, RX_GQ - 1)]);
    if (k >= RX_GQ) {
      const int o = k - RX_GQ;
      e = arena[gb[o / RX_BBLK] + o % RX_BBLK];
    }
    return e;
  };

  bool active = false, exhausted = false, raced = false;
  int pendOld = 0x7FFFFFFF;
  int r = 0, cnt = 0, k = 0, bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
  float sumdx = 0.f, sumdy = 0.f;
  double reg_angle = 0.0;
  for (;;) {
    // ---- groups without a region take the next item ---------------------------------------------
    if (!active && !exhausted) {
      int wi = 0;
      if (gl == 0) wi = atomicAdd(&c.nextBig, 1);
      wi = U(__shfl(wi, gbase, 64));
      if (wi >= nitems) exhausted = true;
      else {
        active = true;
        if (wi < nbig) {
          const RxSeed sd = seeds[wi];
          r = sd.rank;
          reg_angle = (double)sd.ang * RX_DEG2RAD;
          sumdx = sd.sx; sumdy = sd.sy;
          q[0] = sd.xy;                               // every lane of the group stores the same value
          cnt = 1; k = 0;
          bx0 = bx1 = sd.xy & 0xFFFF; by0 = by1 = sd.xy >> 16;
        } else {
          const RxHand& hd = hand[wi - nbig];
          r = hd.rank; cnt = hd.cnt; k = hd.k;
          sumdx = hd.sumdx; sumdy = hd.sumdy;
          reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * RX_DEG2RAD;   // cnt >= 2: the angle is a function of the sums
          bx0 = hd.box0 & 0xFFFF; by0 = hd.box0 >> 16; bx1 = hd.box1 & 0xFFFF; by1 = hd.box1 >> 16;
          for (int i = gl; i < cnt; i += RX_GL) q[i] = hd.q[i];
        }
        r = U(r); cnt = U(cnt); k = U(k);
        bx0 = U(bx0); bx1 = U(bx1); by0 = U(by0); by1 = U(by1);
        sumdx = Uf(sumdx); sumdy = Uf(sumdy);
      }
    }
    if constexpr (RX_GL == 64) { active = U(active) != 0; exhausted = U(exhausted) != 0; }
    if (!__builtin_amdgcn_ballot_w64(active)) break;
    // single-wave block: LDS operations of a wave execute in order, so the queue writes of the last step are
    // visible to the reads below; the compiler only has to keep the order (no s_barrier: __syncthreads would
    // also wait for the claims in flight)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // ---- one BFS step of every active group -------------------------------------------------------
    // Two copies: while every queue of the wave fits in LDS the step has no global access besides the
    // neighbourhood loads and the claims (any other load would make the compiler wait for the claims too).
    bool dead = false, accepted = false;
    int qi = 0, nb = 0;
    // The claims of the previous step are returning atomics: their results are consumed here, before the owner
    // loads of this step are issued, so those loads see the group's own claims (a lower rank that slipped in
    // between our load and our claim is ignored: such a region is not final in this round anyway, and the lowest
    // non-final region never sees that race — the pixels of final lower ranks are already theirs in owner_{t-1}).
    raced = raced || pendOld <= r;
    asm volatile("" ::"v"(pendOld) : "memory");
    pendOld = 0x7FFFFFFF;
    auto step = [&](auto spillTag) {
      constexpr bool SPILL = decltype(spillTag)::value;
      int myxy = -1;
      qi = -1;
      float4 rr = make_float4(RX_NOTDEF, 0.f, 0.f, 0.f);
      int2 oo = make_int2(0, 0);
      nb = active ? min(NP, cnt - k) : 0;
      if (pi < nb) {
        const int e = SPILL ? qget(k + pi) : rx_lds_read(&q[k + pi]);
        const int nx = (e & 0xFFFF) + ndx, ny = (e >> 16) + ndy;
        if (nx >= 0 && ny >= 0 && nx < W && ny < H) {
          qi = ny * W + nx;
          myxy = (ny << 16) | nx;
          rr = rec[qi];
          oo = rx_load_own(&own[qi]);
        }
      }
      const int prevv = ci ? oo.x : oo.y, curv = ci ? oo.y : oo.x;
      const bool cand = rr.x != RX_NOTDEF && !(prevv < r || curv <= r);
      const double ad = (double)rr.x * RX_DEG2RAD;
      unsigned long long rem = (__builtin_amdgcn_ballot_w64(cand) >> gbase) & GMASK;      // the group's candidates in test order
      dead = false; accepted = false;
      while (__builtin_amdgcn_ballot_w64(rem != 0)) {
        double n_theta = fabs(reg_angle - ad);
        if (n_theta > RX_3_2_PI) {
          n_theta = fabs(n_theta - RX_2PI);
        }
        const unsigned long long m = ((__builtin_amdgcn_ballot_w64(cand && n_theta <= prec) >> gbase) & GMASK) & rem;
        if (!m) { rem = 0; continue; }
        const int j2 = __ffsll((long long)m) - 1;
        rem &= ~((2ull << j2) - 1ull);
        const float cj = bcast(rr.y, j2), sj = bcast(rr.z, j2);
        const int xyj = __float_as_int(bcast(__int_as_float(myxy), j2));
        const int qj = __float_as_int(bcast(__int_as_float(qi), j2));
        rem &= ~((__builtin_amdgcn_ballot_w64(qi == qj) >> gbase) & GMASK);             // the other copies of the accepted pixel
        const int ax = xyj & 0xFFFF, ay = xyj >> 16;
        accepted = accepted || gl == j2;               // the claims are issued together after the loop
        if (!SPILL || cnt < RX_GQ) {
          q[cnt] = xyj;                                // every lane of the group stores the same value
        } else {
          const int o = cnt - RX_GQ;
          if (o % RX_BBLK == 0) {
            int nbk = 0;
            if (o / RX_BBLK >= RX_BMAXBLK) { dead = true; rem = 0; continue; }
            if (gl == 0) {
              const unsigned long long ra = atomicAdd(&c.rectArena, (unsigned long long)RX_BBLK) & ((1ull << RX_ARENA_BITS) - 1ull);
              nbk = ra + RX_BBLK > (unsigned long long)arenaCap ? -1 : (int)ra;
            }
            nbk = __shfl(nbk, gbase, 64);
            if (nbk < 0) { dead = true; rem = 0; continue; }
            gb[o / RX_BBLK] = nbk;
          }
          if (gl == 0) arena[gb[o / RX_BBLK] + o % RX_BBLK] = xyj;
          __threadfence_block();
        }
        ++cnt;
        bx0 = min(bx0, ax); bx1 = max(bx1, ax); by0 = min(by0, ay); by1 = max(by1, ay);
        sumdx = __fadd_rn(sumdx, cj);
        sumdy = __fadd_rn(sumdy, sj);
        reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * RX_DEG2RAD;
      }
    };
    if (__builtin_amdgcn_ballot_w64(active && cnt + 8 * NP + 1 > RX_GQ)) step(std::true_type{});
    else step(std::false_type{});
    if (accepted) pendOld = __hip_atomic_fetch_min(ci ? &own[qi].y : &own[qi].x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (active) {
      k += nb;
      if (dead) { c.overflow = 5; active = false; exhausted = true; }
      else if (k >= cnt) {
        // ---- the region is complete ---------------------------------------------------------------
        active = false;
        if (raced || pendOld <= r) atomicAdd(&c.races, 1);   // statistics
        raced = false; pendOld = 0x7FFFFFFF;
        if (gl == 0) {
          rgSize[r] = cnt;
          rgBox[r] = make_int2((by0 << 16) | bx0, (by1 << 16) | bx1);
        }
        if (cnt >= minReg) {                          // the pixel list goes to k_rx_rect (region2rect)
          int off = 0, slot = 0;
          if (gl == 0) {
            const unsigned long long ra = atomicAdd(&c.rectArena, (1ull << RX_ARENA_BITS) | (unsigned long long)cnt);
            const unsigned long long o64 = ra & ((1ull << RX_ARENA_BITS) - 1ull);
            off = o64 + cnt > (unsigned long long)arenaCap ? -1 : (int)o64;
            slot = (int)(ra >> RX_ARENA_BITS);
          }
          off = __shfl(off, gbase, 64); slot = __shfl(slot, gbase, 64);
          if (off < 0 || slot >= rectCap) c.overflow = off < 0 ? 3 : 4;
          else {
            for (int i = gl; i < cnt; i += RX_GL) arena[off + i] = qget(i);
            if (gl == 0) {
              RxRect& it = rects[slot];
              it.rank = r; it.off = off; it.cnt = cnt; it.sumdx = sumdx; it.sumdy = sumdy; it.approx = 0;
            }
          }
        }
      }
    }
  }
}

// two regions per wave (32-lane groups, 4 queue entries per step): the throughput form (batches)
__global__ __launch_bounds__(64) void k_rx_grow_big(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                    const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                    const RxSeed* __restrict__ bigAll, int bigCap,
                                                    const RxHand* __restrict__ handAll, int handCap,
                                                    int* __restrict__ rgSizeAll, int2* __restrict__ rgBoxAll,
                                                    int* __restrict__ arenaAll, int arenaCap,
                                                    RxRect* __restrict__ rectAll, int rectCap, int img0, int t) {
  rx_grow_groups<32, 512>(Pp, ctl, recAll, ownAll, bigAll, bigCap, handAll, handCap, rgSizeAll, rgBoxAll, arenaAll, arenaCap,
                          rectAll, rectCap, img0, t);
}

// one region per wave: the latency form (a few images; the longest region of a round is the critical path, and a
// wave that carries nothing else steps it ~25 % faster)
__global__ __launch_bounds__(64) void k_rx_grow_wave(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                     const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                     const RxSeed* __restrict__ bigAll, int bigCap,
                                                     const RxHand* __restrict__ handAll, int handCap,
                                                     int* __restrict__ rgSizeAll, int2* __restrict__ rgBoxAll,
                                                     int* __restrict__ arenaAll, int arenaCap,
                                                     RxRect* __restrict__ rectAll, int rectCap, int img0, int t) {
  rx_grow_groups<64, 1024>(Pp, ctl, recAll, ownAll, bigAll, bigCap, handAll, handCap, rgSizeAll, rgBoxAll, arenaAll, arenaCap,
                           rectAll, rectCap, img0, t);
}

#endif   // PLI_DEV

__global__ __launch_bounds__(64) void k_rx_rect(const DevParams* __restrict__ Pp, RxCtl* __restrict__ ctl,
                                                const float4* __restrict__ recAll, const int* __restrict__ arenaAll,
                                                int arenaCap, const RxRect* __restrict__ rectAll, int rectCap,
                                                float4* __restrict__ rgSegAll, int img0, const double* __restrict__ mgAll, int rmask,
                                                const int2* __restrict__ hotAll /* or null: the hot records (regions with approximate sums) */,
                                                const float2* __restrict__ coldAll /* or null: the exact {cos, sin} plane beside them */) {
  __shared__ double st[3][64];
  __shared__ double wc[RX_RECT_CACHE][64];
  __shared__ int ec[RX_RECT_CACHE][64];
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0;
  const RxCtl& c = ctl[img];
  if (c.state == 2 || c.overflow) return;
  const int64_t npix = (int64_t)P.LW * P.LH;
  rx_rect_wave(P, c, recAll + img * npix, mgAll ? mgAll + img * npix : nullptr /* CV_64F pipeline: the gradient norm as a double plane */,
               arenaAll + (int64_t)img * arenaCap, rectAll + (int64_t)img * rectCap, rectCap, rgSegAll + img * npix, blockIdx.x, gridDim.x, st, wc, ec, rmask,
               hotAll ? hotAll + img * npix : nullptr, coldAll ? coldAll + img * npix : nullptr);
}

// ---- segments in seed-rank order (= detection order of the sequential algorithm) ------------------------
constexpr int RX_CCHUNK = 2048;   // ranks per block

// (the size first: it is read in rank order — by the callers, for all the ranks of a thread at once and from a clamped index: a load
// under the lane predicate i < n is waited for at the end of its branch —, and only the ~1.5 % of the ranks whose region is large
// enough go on to the two dependent gathers)
__device__ __forceinline__ bool rx_emits(const int* order, const int2* own, int size, int i, int n, int minReg) {
  if (i >= n) return false;
  return size >= minReg && own[order[i]].x == i;
}

__global__ __launch_bounds__(256) void k_rx_count(const RxCtl* __restrict__ ctl, const int* __restrict__ orderAll,
                                                  const int* __restrict__ nDefined, const int2* __restrict__ ownAll,
                                                  const int* __restrict__ rgSizeAll, int* __restrict__ chunkCntAll,
                                                  int nChunks, int64_t npix, int minReg, int img0) {
  __shared__ int wsum[4];
  const int img = blockIdx.y + img0;
  if (ctl[img].state != 2 || ctl[img].overflow) return;
  const int n = nDefined[img];
  const int tid = threadIdx.x;
  int cntv = 0;
  int sz[RX_CCHUNK / 256];
#pragma unroll
  for (int j = 0; j < RX_CCHUNK / 256; ++j) sz[j] = rgSizeAll[img * npix + min((int)blockIdx.x * RX_CCHUNK + j * 256 + tid, max(n - 1, 0))];
#pragma unroll
  for (int j = 0; j < RX_CCHUNK / 256; ++j)
    cntv += rx_emits(orderAll + img * npix, ownAll + img * npix, sz[j], blockIdx.x * RX_CCHUNK + j * 256 + tid, n, minReg) ? 1 : 0;
  cntv = wave_sum_i32(cntv);
  if ((tid & 63) == 0) wsum[tid >> 6] = cntv;
  __syncthreads();
  if (tid == 0) chunkCntAll[(int64_t)img * nChunks + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void k_rx_emit(const RxCtl* __restrict__ ctl, const int* __restrict__ orderAll,
                                                 const int* __restrict__ nDefined, const int2* __restrict__ ownAll,
                                                 const int* __restrict__ rgSizeAll, const float4* __restrict__ rgSegAll,
                                                 const int* __restrict__ chunkCntAll, int nChunks, int64_t npix,
                                                 int minReg, float* __restrict__ segAll, int* __restrict__ nSeg,
                                                 int maxSeg, int img0) {
  __shared__ int wsum[4];
  __shared__ int sbase;
  const int img = blockIdx.y + img0;
  if (ctl[img].state != 2 || ctl[img].overflow) return;
  const int n = nDefined[img];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int* cc = chunkCntAll + (int64_t)img * nChunks;
  int part = 0;
  for (int j = tid; j < (int)blockIdx.x; j += 256) part += cc[j];
  part = wave_sum_i32(part);
  if (lane == 0) wsum[wv] = part;
  __syncthreads();
  if (tid == 0) sbase = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  int base = sbase;
  float* seg = segAll + (int64_t)img * maxSeg * 4;
  int sz[RX_CCHUNK / 256];
#pragma unroll
  for (int j = 0; j < RX_CCHUNK / 256; ++j) sz[j] = rgSizeAll[img * npix + min((int)blockIdx.x * RX_CCHUNK + j * 256 + tid, max(n - 1, 0))];
#pragma unroll
  for (int j = 0; j < RX_CCHUNK / 256; ++j) {
    const int i = blockIdx.x * RX_CCHUNK + j * 256 + tid;
    const bool e = rx_emits(orderAll + img * npix, ownAll + img * npix, sz[j], i, n, minReg);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(e);
    __syncthreads();
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w2 = 0; w2 < wv; ++w2) pos += wsum[w2];
    if (e && pos < maxSeg) {
      const float4 v = rgSegAll[img * npix + i];
      seg[4 * pos + 0] = v.x; seg[4 * pos + 1] = v.y; seg[4 * pos + 2] = v.z; seg[4 * pos + 3] = v.w;
    }
    base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  if ((int)blockIdx.x == nChunks - 1 && tid == 0) nSeg[img] = min(base, maxSeg);
}

}  // namespace pli
