// LSD region growing as a rank-ordered relaxation — the parallel form of
// OpenCV lsd.cpp's region_grow loop (refine = NONE), exact at convergence.
//
// Sequential definition (what the CPU path does): seeds are visited in rank
// order (gradient bin descending, raster ascending); an unused seed grows a
// region over unused aligned pixels (8-neighbour BFS, the region angle is
// re-estimated after every accepted pixel); every pixel a region takes is USED
// for all later seeds.  Let owner[q] be the rank of the region that takes q.
//
// Relaxation: keep an estimate owner_t.  In round t every seed r that is not
// taken by a lower rank in owner_{t-1} grows its region ALONE, treating a
// pixel q as used iff owner_{t-1}[q] < r (or a lower rank has already claimed
// it in this round), and claims its pixels with atomicMin(owner_t[q], r).
// owner_0[q] = rank[q] (a pixel is never taken later than by its own seed).
// By induction on rank the lowest-ranked region whose estimate is still wrong
// only reads correct data and is therefore computed correctly in the next
// round; hence owner_t == owner_{t-1} implies owner_t is the sequential
// result, and at least one more region becomes final every round.  On the
// EuRoC-shaped stream this takes 12-15 rounds with tens of thousands of
// independent growers per round instead of 411 000 dependent steps.
//
// One more round with the final owner map re-grows every region (now all
// exact, including the visiting order inside each region, which fixes the
// floating point sums) and fits the rectangles (region2rect); segments are
// then put back into seed-rank order, which is the CPU path's output order.
#include "kernels.hpp"
#include "device_prims.hpp"

namespace pli {

constexpr float JR_NOTDEF = -1024.f;
constexpr double JR_PI = 3.14159265358979323846;
constexpr double JR_DEG2RAD = JR_PI / 180;
constexpr double JR_3_2_PI = (3 * JR_PI) / 2;
constexpr double JR_2PI = 2 * JR_PI;
constexpr int JR_MQ = 16;        // queue entries a lane keeps in LDS
constexpr int JR_CHUNK = 8;      // arena chunk: 1 link + 7 entries
constexpr int JR_R1_CAP = 64;    // round 1 grows against owner_0 = rank only: cap the speculative regions
constexpr int JR_BIG = 48;       // regions at least this large last round are grown by a whole wave

__device__ __forceinline__ int2 jr_load_own(const int2* p) {
  // bypass the per-CU L1: claims of other workgroups (and our own atomics) are served from L2 / memory
  unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
  return make_int2((int)(v & 0xFFFFFFFFull), (int)(v >> 32));
}

// ---- round bookkeeping -------------------------------------------------------
__global__ void k_jr_begin(JrCtl* __restrict__ ctl, int nimg, int img0, int t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nimg) return;
  JrCtl& c = ctl[img0 + i];
  if (t == 1) { c.state = 0; c.changed = 0; c.overflow = 0; c.nSegRaw = 0; c.rounds = 0; }
  if (c.state == 1) c.state = 2;          // the emit round ran in the previous iteration
  for (int b = 0; b < JR_K; ++b) { c.liveCount[b] = 0; c.bigCount[b] = 0; c.next[b] = 0; c.nextBig[b] = 0; }
  c.arenaHead = 0;
}

// reset owner_t to rank, compare owner_{t-1} with owner_{t-2}, and list the seeds that are alive in owner_{t-1}
// as ready-to-run records.  Seeds whose region had >= JR_BIG pixels in the previous round go to the list of the
// wave-cooperative grower, the rest to the lane-per-region grower.
__global__ __launch_bounds__(256) void k_jr_prepare(JrCtl* __restrict__ ctl, const int* __restrict__ orderAll,
                                                    const int* __restrict__ nDefined, int2* __restrict__ ownAll,
                                                    const float4* __restrict__ recAll, const float2* __restrict__ seedAll,
                                                    const int* __restrict__ lastSizeAll, JrSeed* __restrict__ smallAll,
                                                    JrSeed* __restrict__ bigAll, int bigCap, int64_t npix, int W,
                                                    int bigThresh, int kUse, int img0, int t) {
  const int img = blockIdx.y + img0;
  JrCtl& c = ctl[img];
  if (c.state != 0) return;
  const int n = nDefined[img];
  const int* order = orderAll + img * npix;
  int2* own = ownAll + img * npix;
  const float4* rec = recAll + img * npix;
  const float2* seedcs = seedAll + img * npix;
  const int* lastSize = lastSizeAll + img * npix;
  JrSeed* smallL = smallAll + img * (npix + 64 * JR_K);      // bucket b starts at b * smallSeg
  JrSeed* bigL = bigAll + (int64_t)img * bigCap * JR_K;        // bucket b starts at b * bigCap
  const int smallSeg = (int)(npix / kUse) + 64;
  const int lane = threadIdx.x & 63;
  const int pi = (t - 1) & 1;
  bool changed = false;
  for (int i0 = blockIdx.x * 256; i0 < n; i0 += 256 * gridDim.x) {
    const int i = i0 + threadIdx.x;
    bool alive = false, big = false;
    int p = 0;
    if (i < n) {
      p = order[i];
      int2 o = own[p];
      int prevv, curOld;
      if (t == 1) { prevv = i; curOld = i; }
      else { prevv = pi ? o.y : o.x; curOld = pi ? o.x : o.y; }
      changed = changed || (t >= 2 && prevv != curOld);
      if (t == 1) o = make_int2(i, i);
      else if (pi) o.x = i;
      else o.y = i;
      own[p] = o;
      alive = prevv == i;
      big = alive && t > 1 && lastSize[i] >= bigThresh;
    }
    JrSeed sd;
    if (alive) {
      const float4 r = rec[p];
      const float2 sc = seedcs[p];
      const int py = p / W;
      sd.rank = i; sd.xy = (py << 16) | (p - py * W); sd.ang = r.x; sd.sx = sc.x; sd.sy = sc.y;
    }
    // rank bucket of this seed; a wave covers 64 consecutive ranks, i.e. at most two buckets
    const int bkt = (i < n) ? (int)(((long long)i * kUse) / n) : 0;
    const int bLo = __shfl(bkt, 0, 64);
    for (int pass = 0; pass < 2; ++pass) {
      const int bb = bLo + pass;
      const unsigned long long balS = __ballot(alive && !big && bkt == bb), balB = __ballot(big && bkt == bb);
      if (balS) {
        const int leader = __ffsll((long long)balS) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&c.liveCount[bb], __popcll(balS));
        base = __shfl(base, leader, 64);
        if (alive && !big && bkt == bb) smallL[(int64_t)bb * smallSeg + base + __popcll(balS & ((1ull << lane) - 1ull))] = sd;
      }
      if (balB) {
        const int leader = __ffsll((long long)balB) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&c.bigCount[bb], __popcll(balB));
        base = __shfl(base, leader, 64);
        const int pos = base + __popcll(balB & ((1ull << lane) - 1ull));
        if (big && bkt == bb) {
          if (pos < bigCap) bigL[(int64_t)bb * bigCap + pos] = sd;
          else c.overflow = 1;
        }
      }
    }
  }
  if (__ballot(changed) && lane == 0) atomicOr(&c.changed, 1);
}

__global__ void k_jr_decide(JrCtl* __restrict__ ctl, int nimg, int img0, int t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nimg) return;
  JrCtl& c = ctl[img0 + i];
  if (c.state == 0 && t >= 3 && c.changed == 0) { c.state = 1; c.rounds = t; }   // owner_{t-1} == owner_{t-2}: exact
  c.changed = 0;
}

// ---- the growers -----------------------------------------------------------------
struct JrQueue {
  int first, wBase, wPos, rBase, rPos;
};

__device__ __forceinline__ double jr_angle_diff(double a, double b) {
  double diff = a - b;
  while (diff <= -JR_PI) diff += JR_2PI;
  while (diff > JR_PI) diff -= JR_2PI;
  return fabs(diff);
}

// ---- lane-per-region grower (small regions) -----------------------------------------------
// A wave takes 64 seed records at a time (one atomic per wave, one coalesced load per lane) and runs
// them to completion; the serial BFS of a lane is the CPU loop verbatim.
__global__ __launch_bounds__(256) void k_jr_grow(const DevParams* __restrict__ Pp, JrCtl* __restrict__ ctl,
                                                 const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                 const JrSeed* __restrict__ smallAll, int* __restrict__ lastSizeAll,
                                                 int* __restrict__ arenaAll, int arenaCap,
                                                 float4* __restrict__ segRawAll, int* __restrict__ segRankAll,
                                                 int maxSeg, int img0, int t, int bkt, int kUse) {
  __shared__ int mq[JR_MQ * 256];
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0;
  JrCtl& c = ctl[img];
  const int state = c.state;
  if (state == 2 || c.overflow) return;
  const bool emit = state == 1;
  const int nlive = c.liveCount[bkt];
  if ((int)blockIdx.x * 256 >= nlive) return;      // more lanes than seeds
  const int W = P.LW, H = P.LH;
  const int64_t npix = (int64_t)W * H;
  const float4* rec = recAll + img * npix;
  int2* own = ownAll + img * npix;
  const JrSeed* seeds = smallAll + img * (npix + 64 * JR_K) + (int64_t)bkt * ((int)(npix / kUse) + 64);
  int* lastSize = lastSizeAll + img * npix;
  int* arena = arenaAll + (int64_t)img * arenaCap;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ci = t & 1;                              // owner_t lives in component ci, owner_{t-1} in the other
  const double prec = P.prec;
  const int minReg = P.minRegSize;
  const int cap = (t == 1) ? JR_R1_CAP : 0x7FFFFFFF;

  for (;;) {
    int base = 0;
    if (lane == 0) base = atomicAdd(&c.next[bkt], 64);
    base = __shfl(base, 0, 64);
    if (base >= nlive) break;
    bool active = base + lane < nlive;
    int r = 0, cnt = 0, k = 0;
    float sumdx = 0.f, sumdy = 0.f;
    double reg_angle = 0.0;
    JrQueue Q = {-1, -1, 0, -1, 0};
    if (active) {
      const JrSeed sd = seeds[base + lane];
      r = sd.rank;
      reg_angle = (double)sd.ang * JR_DEG2RAD;
      sumdx = sd.sx;
      sumdy = sd.sy;
      mq[tid] = sd.xy;
      cnt = 1;
    }
    while (__ballot(active)) {
      if (!active) continue;
      // ---- one BFS step of this lane's region ----------------------------------------
      int xy;
      if (k < JR_MQ) xy = mq[k * 256 + tid];
      else {
        if (Q.rBase < 0) { Q.rBase = Q.first; Q.rPos = 1; }
        else if (Q.rPos == JR_CHUNK) { Q.rBase = arena[Q.rBase]; Q.rPos = 1; }
        xy = arena[Q.rBase + Q.rPos++];
      }
      const int px = xy & 0xFFFF, py = xy >> 16;
      float4 nr[8];
      int2 no[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int m = n < 4 ? n : n + 1;               // skip the centre of the 3x3 block
        const int nx = px + m % 3 - 1, ny = py + m / 3 - 1;
        const bool inb = nx >= 0 && ny >= 0 && nx < W && ny < H;
        nr[n] = make_float4(JR_NOTDEF, 0.f, 0.f, 0.f);
        no[n] = make_int2(0, 0);
        if (inb) {
          const int q = ny * W + nx;
          nr[n] = rec[q];
          no[n] = jr_load_own(&own[q]);
        }
      }
      bool dead = false;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        if (nr[n].x == JR_NOTDEF || dead) continue;
        const int prevv = ci ? no[n].x : no[n].y;
        const int curv = ci ? no[n].y : no[n].x;
        if (prevv < r || curv <= r) continue;          // taken by a lower rank (last round / this round) or already mine
        double n_theta = reg_angle - (double)nr[n].x * JR_DEG2RAD;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > JR_3_2_PI) {
          n_theta -= JR_2PI;
          if (n_theta < 0) n_theta = -n_theta;
        }
        if (!(n_theta <= prec)) continue;
        const int m = n < 4 ? n : n + 1;
        const int nx = px + m % 3 - 1, ny = py + m / 3 - 1;
        const int q = ny * W + nx;
        // the returned value orders this claim before the owner loads of the next steps (a no-return atomic may
        // still be in flight when they are issued); a lower rank that got there first keeps the pixel
        const int old = __hip_atomic_fetch_min(ci ? &own[q].y : &own[q].x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old <= r) continue;
        const int e = (ny << 16) | nx;
        if (cnt < JR_MQ) mq[cnt * 256 + tid] = e;
        else {
          if (Q.wBase < 0 || Q.wPos == JR_CHUNK) {
            const int nb = atomicAdd(&c.arenaHead, JR_CHUNK);
            if (nb + JR_CHUNK > arenaCap) { c.overflow = 1; dead = true; continue; }
            if (Q.wBase < 0) Q.first = nb; else arena[Q.wBase] = nb;
            Q.wBase = nb;
            Q.wPos = 1;
          }
          arena[Q.wBase + Q.wPos++] = e;
        }
        ++cnt;
        sumdx = __fadd_rn(sumdx, nr[n].y);
        sumdy = __fadd_rn(sumdy, nr[n].z);
        reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * JR_DEG2RAD;
      }
      ++k;
      if (dead) { active = false; continue; }
      if (k < cnt && cnt < cap) continue;
      // ---- the region is complete -------------------------------------------------------
      active = false;
      lastSize[r] = cnt;
      if (!emit || cnt < minReg) continue;
      // region2rect, sums in list order (lsd.cpp region2rect / get_theta)
      auto qat = [&](int i, JrQueue& it) -> int {
        if (i < JR_MQ) return mq[i * 256 + tid];
        if (it.rBase < 0) { it.rBase = Q.first; it.rPos = 1; }
        else if (it.rPos == JR_CHUNK) { it.rBase = arena[it.rBase]; it.rPos = 1; }
        return arena[it.rBase + it.rPos++];
      };
      double x = 0, y = 0, sum = 0;
      {
        JrQueue it = {Q.first, -1, 0, -1, 0};
        for (int i = 0; i < cnt; ++i) {
          const int e = qat(i, it);
          const int ex = e & 0xFFFF, ey = e >> 16;
          const double w = sqrt((double)__float_as_int(rec[ey * W + ex].w) / 4.0);
          x += (double)ex * w;
          y += (double)ey * w;
          sum += w;
        }
      }
      x /= sum;
      y /= sum;
      double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
      {
        JrQueue it = {Q.first, -1, 0, -1, 0};
        for (int i = 0; i < cnt; ++i) {
          const int e = qat(i, it);
          const int ex = e & 0xFFFF, ey = e >> 16;
          const double w = sqrt((double)__float_as_int(rec[ey * W + ex].w) / 4.0);
          const double dx = (double)ex - x, dy = (double)ey - y;
          Ixx += dy * dy * w;
          Iyy += dx * dx * w;
          Ixy -= dx * dy * w;
        }
      }
      const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
      double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                             : (double)fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
      theta *= JR_DEG2RAD;
      if (jr_angle_diff(theta, reg_angle) > prec) theta += JR_PI;
      const double dxr = cos(theta), dyr = sin(theta);
      double l_min = 0, l_max = 0;
      {
        JrQueue it = {Q.first, -1, 0, -1, 0};
        for (int i = 0; i < cnt; ++i) {
          const int e = qat(i, it);
          const double l = ((double)(e & 0xFFFF) - x) * dxr + ((double)(e >> 16) - y) * dyr;
          if (l > l_max) l_max = l;
          else if (l < l_min) l_min = l;
        }
      }
      double x1 = x + l_min * dxr, y1 = y + l_min * dyr, x2 = x + l_max * dxr, y2 = y + l_max * dyr;
      x1 += 0.5; y1 += 0.5; x2 += 0.5; y2 += 0.5;
      const double scale = P.lsdScale;
      if (scale != 1) { x1 /= scale; y1 /= scale; x2 /= scale; y2 /= scale; }
      const int si = atomicAdd(&c.nSegRaw, 1);
      if (si < maxSeg) {
        segRawAll[(int64_t)img * maxSeg + si] = make_float4((float)x1, (float)y1, (float)x2, (float)y2);
        segRankAll[(int64_t)img * maxSeg + si] = r;
      }
    }
  }
}

// ---- wave-per-region grower (regions that were large in the previous round) ------------------------
// Lanes 0..8 fetch the 3x3 neighbourhood of the current queue entry (record + owner pair) in one round
// trip; ballots over "available & aligned" reproduce the raster-order accept loop; the queue lives in
// LDS (first JR_BQ entries) and in arena blocks beyond.
constexpr int JR_BQ = 2048;
constexpr int JR_BBLK = 2048;     // arena block for the overflow of a big region's queue
constexpr int JR_BMAXBLK = 64;

__device__ __forceinline__ int jr_rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float jr_rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__global__ __launch_bounds__(64) void k_jr_grow_big(const DevParams* __restrict__ Pp, JrCtl* __restrict__ ctl,
                                                    const float4* __restrict__ recAll, int2* __restrict__ ownAll,
                                                    const JrSeed* __restrict__ bigAll, int bigCap,
                                                    int* __restrict__ lastSizeAll, int* __restrict__ arenaAll,
                                                    int arenaCap, float4* __restrict__ segRawAll,
                                                    int* __restrict__ segRankAll, int maxSeg, int img0, int t, int bkt,
                                                    int* __restrict__ dbgQ, int dbgRank) {
  __shared__ int qs[JR_BQ];
  __shared__ int blk[JR_BMAXBLK];
  __shared__ double st[3][64];
  const DevParams& P = *Pp;
  const int img = blockIdx.y + img0;
  JrCtl& c = ctl[img];
  const int state = c.state;
  if (state == 2 || c.overflow) return;
  const bool emit = state == 1;
  const int nbig = min(c.bigCount[bkt], bigCap);
  if ((int)blockIdx.x >= nbig) return;
  const int W = P.LW, H = P.LH;
  const int64_t npix = (int64_t)W * H;
  const float4* rec = recAll + img * npix;
  int2* own = ownAll + img * npix;
  const JrSeed* seeds = bigAll + ((int64_t)img * JR_K + bkt) * bigCap;
  int* lastSize = lastSizeAll + img * npix;
  int* arena = arenaAll + (int64_t)img * arenaCap;
  const int lane = threadIdx.x;
  const int ci = t & 1;
  const double prec = P.prec;
  const int minReg = P.minRegSize;
  const int ndx = lane % 3 - 1, ndy = (lane / 3) % 3 - 1;

  auto qget = [&](int k) -> int {
    if (k < JR_BQ) return qs[k];
    const int o = k - JR_BQ;
    return arena[blk[o / JR_BBLK] + o % JR_BBLK];
  };

  for (;;) {
    int wi = 0;
    if (lane == 0) wi = atomicAdd(&c.nextBig[bkt], 1);
    wi = __shfl(wi, 0, 64);
    if (wi >= nbig) break;
    const JrSeed sd = seeds[wi];
    const int r = sd.rank;
    double reg_angle = (double)sd.ang * JR_DEG2RAD;
    float sumdx = sd.sx, sumdy = sd.sy;
    qs[0] = sd.xy;
    int cnt = 1;
    bool dead = false;
    for (int k = 0; k < cnt && !dead; ++k) {
      __syncthreads();                              // single-wave block: orders lane 0's queue writes before the reads
      int e;
      if (k < JR_BQ) e = qs[k];                     // plain ds_read (never a flat access)
      else {
        const int o = k - JR_BQ;
        e = arena[blk[o / JR_BBLK] + o % JR_BBLK];
      }
      e = __builtin_amdgcn_readfirstlane(e);
      const int px = e & 0xFFFF, py = e >> 16;
      const int nx = px + ndx, ny = py + ndy;
      const bool inb = lane < 9 && lane != 4 && nx >= 0 && ny >= 0 && nx < W && ny < H;
      const int qi = inb ? ny * W + nx : 0;
      float4 rr = make_float4(JR_NOTDEF, 0.f, 0.f, 0.f);
      int2 oo = make_int2(0, 0);
      if (inb) {
        rr = rec[qi];
        oo = jr_load_own(&own[qi]);
      }
      const int prevv = ci ? oo.x : oo.y, curv = ci ? oo.y : oo.x;
      const bool cand = rr.x != JR_NOTDEF && !(prevv < r || curv <= r);
      const double ad = (double)rr.x * JR_DEG2RAD;
      unsigned long long remaining = __ballot(cand);
      while (remaining) {
        double n_theta = reg_angle - ad;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > JR_3_2_PI) {
          n_theta -= JR_2PI;
          if (n_theta < 0) n_theta = -n_theta;
        }
        const unsigned long long m = __ballot(cand && n_theta <= prec) & remaining;
        if (!m) break;
        const int j2 = __ffsll((long long)m) - 1;
        remaining &= ~((2ull << j2) - 1ull);
        const float cj = jr_rl_f(rr.y, j2), sj = jr_rl_f(rr.z, j2);
        const int xyj = ((py + j2 / 3 - 1) << 16) | (px + j2 % 3 - 1);
        int old = 0;
        if (lane == j2) old = __hip_atomic_fetch_min(ci ? &own[qi].y : &own[qi].x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = jr_rl_i(old, j2);                     // waits for the claim: later owner loads see it
        if (old <= r) continue;                     // a lower rank took the pixel first in this round
        if (cnt < JR_BQ) {
          qs[cnt] = xyj;                            // every lane stores the same value: no dependence on lane 0
        } else {
          const int o = cnt - JR_BQ;
          // NOTE: inside these wave-uniform loops a given lane (e.g. lane 0) is not guaranteed to be in the
          // exec mask, so single-lane work is done by the first ACTIVE lane and results are broadcast from it.
          const int leader = __ffsll((long long)__ballot(true)) - 1;
          if (o % JR_BBLK == 0) {
            int nb = 0;
            if (o / JR_BBLK >= JR_BMAXBLK) { dead = true; break; }
            if (lane == leader) nb = atomicAdd(&c.arenaHead, JR_BBLK);
            nb = __builtin_amdgcn_readlane(nb, leader);
            if (nb + JR_BBLK > arenaCap) { dead = true; break; }
            blk[o / JR_BBLK] = nb;
            __syncthreads();
          }
          if (lane == leader) arena[blk[o / JR_BBLK] + o % JR_BBLK] = xyj;
          __threadfence_block();
        }
        cnt = __builtin_amdgcn_readfirstlane(cnt + 1);
        sumdx = __fadd_rn(sumdx, cj);
        sumdy = __fadd_rn(sumdy, sj);
        reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * JR_DEG2RAD;
      }
    }
    if (dead) {
      if (lane == 0) c.overflow = 1;
      break;
    }
    if (lane == 0) lastSize[r] = cnt;
    if (emit && dbgQ && r == dbgRank) {
      for (int i = lane; i < cnt && i < 4095; i += 64) dbgQ[1 + i] = qget(i);
      if (lane == 0) dbgQ[0] = cnt;
    }
    if (!emit || cnt < minReg) continue;
    // ---- region2rect: three lanes accumulate the running sums in list order -------------------
    __syncthreads();
    double acc = 0.0;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      const int kk = c0 + lane;
      if (kk < cnt) {
        const int e = qget(kk);
        const int ex = e & 0xFFFF, ey = e >> 16;
        const double w = sqrt((double)__float_as_int(rec[ey * W + ex].w) / 4.0);
        st[0][lane] = (double)ex * w;
        st[1][lane] = (double)ey * w;
        st[2][lane] = w;
      }
      __syncthreads();
      if (lane < 3) {
        const int mm = min(64, cnt - c0);
        for (int tt = 0; tt < mm; ++tt) acc += st[lane][tt];
      }
      __syncthreads();
    }
    const double sum = __shfl(acc, 2, 64);
    const double x = __shfl(acc, 0, 64) / sum, y = __shfl(acc, 1, 64) / sum;
    acc = 0.0;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      const int kk = c0 + lane;
      if (kk < cnt) {
        const int e = qget(kk);
        const int ex = e & 0xFFFF, ey = e >> 16;
        const double w = sqrt((double)__float_as_int(rec[ey * W + ex].w) / 4.0);
        const double dx = (double)ex - x, dy = (double)ey - y;
        st[0][lane] = dy * dy * w;
        st[1][lane] = dx * dx * w;
        st[2][lane] = dx * dy * w;
      }
      __syncthreads();
      if (lane < 2) {
        const int mm = min(64, cnt - c0);
        for (int tt = 0; tt < mm; ++tt) acc += st[lane][tt];
      } else if (lane == 2) {
        const int mm = min(64, cnt - c0);
        for (int tt = 0; tt < mm; ++tt) acc -= st[2][tt];
      }
      __syncthreads();
    }
    const double Ixx = __shfl(acc, 0, 64), Iyy = __shfl(acc, 1, 64), Ixy = __shfl(acc, 2, 64);
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                           : (double)fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
    theta *= JR_DEG2RAD;
    if (jr_angle_diff(theta, reg_angle) > prec) theta += JR_PI;
    const double dxr = cos(theta), dyr = sin(theta);
    double l_min = 0, l_max = 0;
    for (int kk = lane; kk < cnt; kk += 64) {
      const int e = qget(kk);
      const double l = ((double)(e & 0xFFFF) - x) * dxr + ((double)(e >> 16) - y) * dyr;
      l_max = fmax(l_max, l);
      l_min = fmin(l_min, l);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      l_max = fmax(l_max, __shfl_xor(l_max, o, 64));
      l_min = fmin(l_min, __shfl_xor(l_min, o, 64));
    }
    double x1 = x + l_min * dxr, y1 = y + l_min * dyr, x2 = x + l_max * dxr, y2 = y + l_max * dyr;
    x1 += 0.5; y1 += 0.5; x2 += 0.5; y2 += 0.5;
    const double scale = P.lsdScale;
    if (scale != 1) { x1 /= scale; y1 /= scale; x2 /= scale; y2 /= scale; }
    if (lane == 0) {
      const int si = atomicAdd(&c.nSegRaw, 1);
      if (si < maxSeg) {
        segRawAll[(int64_t)img * maxSeg + si] = make_float4((float)x1, (float)y1, (float)x2, (float)y2);
        segRankAll[(int64_t)img * maxSeg + si] = r;
      }
    }
  }
}

// segments back into seed-rank order (= detection order of the sequential algorithm); ranks are unique
__global__ __launch_bounds__(256) void k_jr_sort(const JrCtl* __restrict__ ctl, const float4* __restrict__ segRawAll,
                                                 const int* __restrict__ segRankAll, int maxSeg,
                                                 float* __restrict__ segAll, int* __restrict__ nSeg, int img0) {
  __shared__ int tile[2048];
  const int img = blockIdx.y + img0;
  const int n = min(ctl[img].nSegRaw, maxSeg);
  if (blockIdx.x == 0 && threadIdx.x == 0) nSeg[img] = n;
  if ((int)blockIdx.x * 256 >= n) return;
  const float4* raw = segRawAll + (int64_t)img * maxSeg;
  const int* rk = segRankAll + (int64_t)img * maxSeg;
  float* seg = segAll + (int64_t)img * maxSeg * 4;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  const int mine = i < n ? rk[i] : 0x7FFFFFFF;
  int pos = 0;
  for (int j0 = 0; j0 < n; j0 += 2048) {
    const int m = min(2048, n - j0);
    __syncthreads();
    for (int j = tid; j < m; j += 256) tile[j] = rk[j0 + j];
    __syncthreads();
    for (int j = 0; j < m; ++j) pos += tile[j] < mine;
  }
  if (i < n) {
    const float4 v = raw[i];
    seg[4 * pos + 0] = v.x; seg[4 * pos + 1] = v.y; seg[4 * pos + 2] = v.z; seg[4 * pos + 3] = v.w;
  }
}

}  // namespace pli
