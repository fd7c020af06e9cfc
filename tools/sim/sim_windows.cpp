// Dev tool (CPU): event-driven model of ROUND 1 of the tile relaxation with tile waves that advance through the seed order IN STEP
// (VERDICT r3 item 4a): every 64x64 tile is a wave with its own clock (instruction units: 75 per batched step + 31 per accepted
// pixel, the ISA counts of k_tx_grow), claims are atomicMin on one owner map and visible to every tile from the
// moment they are made (the real launch: L1-bypassing owner loads), and a tile may start the seeds of window w only when its 8
// neighbours have finished the windows below w (a finished tile counts as +inf); a wait ends `latency` units after the last
// neighbour got there.  Windows: quantiles of the seed order (SIMW_K = K), or the gradient bins themselves (SIMW_K = 0).
// Reports, per policy: pixels grown in round 1 (the sequential run = 1.0x), pixels wrong after round 1, the image's makespan
// against the longest tile without waiting, and the share of wave-time spent waiting.
//   g++ -O2 -std=c++17 -I../../oracle -I../../include sim_windows.cpp -o /tmp/sim_windows
//   /tmp/sim_windows img.raw W H [tile] [latency]
#include "line_oracle.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <queue>

using namespace orc;

struct Field {
  int W, H;
  std::vector<double> ang;
  std::vector<float> c, s;
  std::vector<int> order, rankOf;
  double prec;
};

static bool aligned(const Field& F, int q, double theta) {
  const double a = F.ang[q];
  if (a == kNOTDEF) return false;
  double n = theta - a;
  if (n < 0) n = -n;
  if (n > kM_3_2_PI) { n -= kM_2__PI; if (n < 0) n = -n; }
  return n <= F.prec;
}

// grows region r; returns (#batched steps of 8 queue entries, pixel list)
template <class Used, class Claim>
static long grow(const Field& F, int r, Used used, Claim claim, std::vector<int>& reg) {
  const int W = F.W, H = F.H, sp = F.order[r];
  reg.clear();
  double reg_angle = F.ang[sp];
  float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
  reg.push_back(sp);
  claim(sp);
  long steps = 0;
  size_t k = 0;
  while (k < reg.size()) {
    const size_t nb = std::min<size_t>(8, reg.size() - k);       // one batched step: the entries that exist when it starts
    ++steps;
    for (size_t e = k; e < k + nb; ++e) {
      const int px = reg[e] % W, py = reg[e] / W;
      for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
        for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
          const int q = yy * W + xx;
          if (!used(q) && aligned(F, q, reg_angle)) {
            claim(q);
            reg.push_back(q);
            sumdx += F.c[q]; sumdy += F.s[q];
            reg_angle = fastAtan2(sumdy, sumdx) * kDEG_TO_RADS;
          }
        }
    }
    k += nb;
  }
  return steps;
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: img.raw W H [tile] [latency]\n"); return 2; }
  const int iw = atoi(argv[2]), ih = atoi(argv[3]);
  const int TS = argc > 4 ? atoi(argv[4]) : 64;
  const double LAT = argc > 5 ? atof(argv[5]) : 600.0;   // instruction units a released waiter loses (a poll round trip at 8 waves per SIMD)
  Img8 img(iw, ih);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(img.d.data(), 1, img.d.size(), f) != img.d.size()) { std::fprintf(stderr, "read failed\n"); return 2; }
  fclose(f);
  LsdParams P; LsdDebug D;
  lsdDetect(img, P, D);
  Field F;
  F.W = D.W; F.H = D.H;
  const int W = F.W, H = F.H, N = W * H;
  F.ang.assign(N, kNOTDEF); F.c.assign(N, 0); F.s.assign(N, 0);
  for (int i = 0; i < N; ++i)
    if (D.angleDeg[i] != -1024.f) {
      F.ang[i] = D.angleDeg[i] * kDEG_TO_RADS;
      F.c[i] = (float)std::cos((double)float(F.ang[i]));
      F.s[i] = (float)std::sin((double)float(F.ang[i]));
    }
  std::vector<int> binOf;                    // bin index (0 = strongest) per list entry: a new bin starts where the raster order restarts
  {
    int b = 0, last = -1;
    for (int p : D.order) {
      if (p < last) ++b;
      last = p;
      if (F.ang[p] != kNOTDEF) { F.order.push_back(p); binOf.push_back(b); }
    }
  }
  const int R = (int)F.order.size();
  F.rankOf.assign(N, INT32_MAX);
  for (int r = 0; r < R; ++r) F.rankOf[F.order[r]] = r;
  F.prec = kPI * P.ang_th / 180;
  std::vector<int> truth(N, INT32_MAX), reg;
  long seqPix = 0, seqSteps = 0, seqRegions = 0;
  for (int r = 0; r < R; ++r) {
    if (truth[F.order[r]] != INT32_MAX) continue;
    seqSteps += grow(F, r, [&](int q) { return truth[q] != INT32_MAX; }, [&](int q) { truth[q] = r; }, reg);
    seqPix += (long)reg.size(); ++seqRegions;
  }
  const double seqCost = 75.0 * seqSteps + 31.0 * seqPix;
  std::printf("scaled %dx%d, %d seeds in %d bins, sequential: %ld regions, %ld steps, %ld px, cost %.2f M units\n", W, H, R, binOf.back() + 1,
              seqRegions, seqSteps, seqPix, seqCost / 1e6);
  const int TW = (W + TS - 1) / TS, TH = (H + TS - 1) / TS, NT = TW * TH;
  std::vector<std::vector<int>> tileSeeds(NT);
  for (int r = 0; r < R; ++r) { const int p = F.order[r]; tileSeeds[(p / W / TS) * TW + (p % W) / TS].push_back(r); }

  // ---- shared first steps (round-4 idea): the alive seeds of a 64-entry list row start in groups of up to G; the first step of
  // a group's members (the seed's 8 neighbours) is decided against the state at group time — the members do not see each other —,
  // their claims go out together (lowest rank wins a pixel, the loser is noted), then the members continue one after the other
  // in rank order.  Counted: regions started, first steps saved (members - groups), members whose seed a lower member took,
  // regions that lose a pixel to a lower rank AFTER claiming it (they are regrown in round 2), pixels wrong after the round.
  for (int G : {1, 4, 8}) {
    std::vector<int> own(N, INT32_MAX);
    std::vector<char> lost(R, 0);
    long regionsStarted = 0, groups = 0, members = 0, deadInGroup = 0, steps = 0, firstSteps = 0;
    auto claim = [&](int q, int r) {
      if (own[q] != INT32_MAX && own[q] > r) lost[own[q]] = 1;
      if (own[q] < r) lost[r] = 1;
      own[q] = std::min(own[q], r);
    };
    // tiles in a round-robin of rows (all tiles advance one row at a time: a coarse stand-in for concurrent tile waves)
    size_t maxRows = 0;
    for (auto& S : tileSeeds) maxRows = std::max(maxRows, (S.size() + 63) / 64);
    for (size_t row = 0; row < maxRows; ++row)
      for (int T = 0; T < NT; ++T) {
        const std::vector<int>& S = tileSeeds[T];
        if (row * 64 >= S.size()) continue;
        const size_t e = std::min(S.size(), row * 64 + 64);
        std::vector<int> alive;
        for (size_t i = row * 64; i < e; ++i) if (own[F.order[S[i]]] >= S[i]) alive.push_back(S[i]);   // (alive at the row fetch)
        size_t a = 0;
        while (a < alive.size()) {
          // the next group: up to G seeds that are still alive now
          std::vector<int> grp;
          while (a < alive.size() && (int)grp.size() < G) { const int r = alive[a++]; if (own[F.order[r]] >= r) grp.push_back(r); }
          if (grp.empty()) break;
          ++groups; members += (long)grp.size();
          // first steps against the state at group time
          std::vector<std::vector<int>> firstAcc(grp.size());
          std::vector<float> sdx(grp.size()), sdy(grp.size());
          const std::vector<int> snap = own;                    // (small images: a copy per group is affordable)
          for (size_t m = 0; m < grp.size(); ++m) {
            const int r = grp[m], sp = F.order[r], px = sp % W, py = sp / W;
            double reg_angle = F.ang[sp];
            float sx = float(std::cos(reg_angle)), sy = float(std::sin(reg_angle));
            for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
              for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
                const int q = yy * W + xx;
                if (q == sp) continue;
                const bool used = snap[q] < r || F.rankOf[q] < r;
                if (!used && aligned(F, q, reg_angle)) {
                  firstAcc[m].push_back(q);
                  sx += F.c[q]; sy += F.s[q];
                  reg_angle = fastAtan2(sy, sx) * kDEG_TO_RADS;
                }
              }
            sdx[m] = sx; sdy[m] = sy;
          }
          ++firstSteps;
          // a member whose seed a lower member accepted is dropped before the claims
          std::vector<char> dead(grp.size(), 0);
          for (size_t m = 1; m < grp.size(); ++m)
            for (size_t l = 0; l < m && !dead[m]; ++l)
              if (!dead[l]) for (int q : firstAcc[l]) if (q == F.order[grp[m]]) { dead[m] = 1; break; }
          for (size_t m = 0; m < grp.size(); ++m) {
            if (dead[m]) { ++deadInGroup; continue; }
            claim(F.order[grp[m]], grp[m]);
            for (int q : firstAcc[m]) claim(q, grp[m]);
          }
          // continuations, one after the other in rank order (a member that lost its seed in the meantime does not continue)
          for (size_t m = 0; m < grp.size(); ++m) {
            if (dead[m]) continue;
            const int r = grp[m], sp = F.order[r];
            ++regionsStarted;
            if (own[sp] != r) continue;
            // continue the BFS from the first step's state: queue = seed + first accepts, k = 1
            std::vector<int> regq; regq.push_back(sp);
            for (int q : firstAcc[m]) regq.push_back(q);
            float sx = sdx[m], sy = sdy[m];
            double reg_angle = regq.size() > 1 ? fastAtan2(sy, sx) * kDEG_TO_RADS : F.ang[sp];
            size_t k = 1;
            while (k < regq.size()) {
              const size_t nb = std::min<size_t>(8, regq.size() - k);
              ++steps;
              for (size_t e2 = k; e2 < k + nb; ++e2) {
                const int px = regq[e2] % W, py = regq[e2] / W;
                for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
                  for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
                    const int q = yy * W + xx;
                    const bool used = own[q] <= r || F.rankOf[q] < r;
                    if (!used && aligned(F, q, reg_angle)) {
                      claim(q, r);
                      regq.push_back(q);
                      sx += F.c[q]; sy += F.s[q];
                      reg_angle = fastAtan2(sy, sx) * kDEG_TO_RADS;
                    }
                  }
              }
              k += nb;
            }
          }
        }
      }
    long wrong = 0, losers = 0;
    for (int q = 0; q < N; ++q) wrong += own[q] != truth[q];
    for (int r = 0; r < R; ++r) losers += lost[r];
    std::printf("shared first steps, groups of %d: %ld regions started in %ld groups (%.2f per group), %ld members dropped (seed taken in the group), first steps %ld "
                "+ later steps %ld = %ld (sequential: %ld); regions that lost a claimed pixel %ld; wrong after round 1 %ld px\n",
                G, regionsStarted, groups, (double)members / groups, deadInGroup, firstSteps, steps, firstSteps + steps, seqSteps, losers, wrong);
  }
  // ---- the same with the members' claims DEFERRED to their turn: at its turn a member's first step stands iff its seed is still its
  // own and none of the pixels it accepted has been claimed by a lower rank since the group formed (what it rejected stays rejected:
  // owners only go down within a round); a member whose first step does not stand starts again from its seed.  No speculative claim
  // ever reaches the owner map, so nothing is regrown because of the grouping.
  for (int G : {4, 8}) {
    std::vector<int> own(N, INT32_MAX);
    std::vector<char> lost(R, 0);
    long regionsStarted = 0, groups = 0, members = 0, redo = 0, steps = 0, firstSteps = 0, onePix = 0, seedGone = 0;
    auto claim = [&](int q, int r) {
      if (own[q] != INT32_MAX && own[q] > r) lost[own[q]] = 1;
      if (own[q] < r) lost[r] = 1;
      own[q] = std::min(own[q], r);
    };
    size_t maxRows = 0;
    for (auto& S : tileSeeds) maxRows = std::max(maxRows, (S.size() + 63) / 64);
    for (size_t row = 0; row < maxRows; ++row)
      for (int T = 0; T < NT; ++T) {
        const std::vector<int>& S = tileSeeds[T];
        if (row * 64 >= S.size()) continue;
        const size_t e = std::min(S.size(), row * 64 + 64);
        std::vector<int> alive;
        for (size_t i = row * 64; i < e; ++i) if (own[F.order[S[i]]] >= S[i]) alive.push_back(S[i]);
        size_t a = 0;
        while (a < alive.size()) {
          std::vector<int> grp;
          while (a < alive.size() && (int)grp.size() < G) { const int r = alive[a++]; if (own[F.order[r]] >= r) grp.push_back(r); }
          if (grp.empty()) break;
          ++groups; members += (long)grp.size(); ++firstSteps;
          std::vector<std::vector<int>> firstAcc(grp.size());
          std::vector<float> sdx(grp.size()), sdy(grp.size());
          for (size_t m = 0; m < grp.size(); ++m) {                 // (nothing is claimed here: the state is the same for every member)
            const int r = grp[m], sp = F.order[r], px = sp % W, py = sp / W;
            double reg_angle = F.ang[sp];
            float sx = float(std::cos(reg_angle)), sy = float(std::sin(reg_angle));
            for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
              for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
                const int q = yy * W + xx;
                if (q == sp) continue;
                const bool used = own[q] < r || F.rankOf[q] < r;
                if (!used && aligned(F, q, reg_angle)) { firstAcc[m].push_back(q); sx += F.c[q]; sy += F.s[q]; reg_angle = fastAtan2(sy, sx) * kDEG_TO_RADS; }
              }
            sdx[m] = sx; sdy[m] = sy;
          }
          for (size_t m = 0; m < grp.size(); ++m) {
            const int r = grp[m], sp = F.order[r];
            if (own[sp] < r) { ++seedGone; continue; }               // taken by a lower member in the meantime
            ++regionsStarted;
            bool stands = true;
            for (int q : firstAcc[m]) if (own[q] < r) { stands = false; break; }
            std::vector<int> regq; regq.push_back(sp);
            float sx, sy; double reg_angle; size_t k;
            claim(sp, r);
            if (stands) {
              for (int q : firstAcc[m]) { claim(q, r); regq.push_back(q); }
              sx = sdx[m]; sy = sdy[m];
              reg_angle = regq.size() > 1 ? fastAtan2(sy, sx) * kDEG_TO_RADS : F.ang[sp];
              k = 1;
              if (regq.size() == 1) ++onePix;
            } else {
              ++redo;
              reg_angle = F.ang[sp]; sx = float(std::cos(reg_angle)); sy = float(std::sin(reg_angle)); k = 0;
            }
            while (k < regq.size()) {
              const size_t nb = std::min<size_t>(8, regq.size() - k);
              ++steps;
              for (size_t e2 = k; e2 < k + nb; ++e2) {
                const int px = regq[e2] % W, py = regq[e2] / W;
                for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
                  for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
                    const int q = yy * W + xx;
                    const bool used = own[q] <= r || F.rankOf[q] < r;
                    if (!used && aligned(F, q, reg_angle)) { claim(q, r); regq.push_back(q); sx += F.c[q]; sy += F.s[q]; reg_angle = fastAtan2(sy, sx) * kDEG_TO_RADS; }
                  }
              }
              k += nb;
            }
          }
        }
      }
    long wrong = 0, losers = 0;
    for (int q = 0; q < N; ++q) wrong += own[q] != truth[q];
    for (int r = 0; r < R; ++r) losers += lost[r];
    std::printf("shared first steps, claims deferred, groups of %d: %ld regions in %ld groups (%.2f per group), %ld members found their seed gone, %ld first steps "
                "redone, %ld one-pixel regions closed without a step of their own; shared steps %ld + later steps %ld = %ld (sequential %ld); regions that lost a "
                "claimed pixel %ld; wrong after round 1 %ld px\n", G, regionsStarted, groups, (double)members / groups, seedGone, redo, onePix, firstSteps, steps,
                firstSteps + steps, seqSteps, losers, wrong);
  }
  // ---- small regions grown 8 at a time for ALL their steps (the generalisation): the alive seeds of a row form groups of up to G; a
  // group advances in lockstep, one queue entry per member and step (an octet of lanes: its 8 neighbours), claims go out after every
  // step (members see each other's claims from the next step on, like tiles do); a member that reaches CAP pixels leaves the group
  // and is finished by the whole wave (8 entries per step) after the group; a member that finishes is NOT replaced (no refill).
  // Counted: wave steps (group steps + whole-wave steps), regions that lost a claimed pixel, pixels wrong after round 1.
  for (int G : {8}) for (int CAP : {8, 16}) {
    std::vector<int> own(N, INT32_MAX);
    std::vector<char> lost(R, 0);
    long regionsStarted = 0, groups = 0, members = 0, groupSteps = 0, waveSteps = 0, handed = 0;
    auto claim = [&](int q, int r) {
      if (own[q] != INT32_MAX && own[q] > r) lost[own[q]] = 1;
      if (own[q] < r) lost[r] = 1;
      own[q] = std::min(own[q], r);
    };
    struct M { int r; std::vector<int> q; size_t k; float sx, sy; double ang; bool done, big; };
    size_t maxRows = 0;
    for (auto& S : tileSeeds) maxRows = std::max(maxRows, (S.size() + 63) / 64);
    for (size_t row = 0; row < maxRows; ++row)
      for (int T = 0; T < NT; ++T) {
        const std::vector<int>& S = tileSeeds[T];
        if (row * 64 >= S.size()) continue;
        const size_t e = std::min(S.size(), row * 64 + 64);
        std::vector<int> alive;
        for (size_t i = row * 64; i < e; ++i) if (own[F.order[S[i]]] >= S[i]) alive.push_back(S[i]);
        size_t a = 0;
        while (a < alive.size()) {
          std::vector<M> grp;
          while (a < alive.size() && (int)grp.size() < G) {
            const int r = alive[a++];
            if (own[F.order[r]] < r) continue;
            const int sp = F.order[r];
            claim(sp, r);
            grp.push_back(M{r, {sp}, 0, float(std::cos(F.ang[sp])), float(std::sin(F.ang[sp])), F.ang[sp], false, false});
          }
          if (grp.empty()) break;
          ++groups; members += (long)grp.size(); regionsStarted += (long)grp.size();
          // lockstep: one queue entry per member and step
          for (;;) {
            bool any = false;
            std::vector<std::pair<int, int>> claims;            // (pixel, rank) of this step, applied after it
            for (auto& m : grp) {
              if (m.done || m.big) continue;
              if (own[m.q[0]] != m.r) { m.done = true; continue; }              // its seed was taken: the member dies
              if (m.k >= m.q.size()) { m.done = true; continue; }
              any = true;
              const int px = m.q[m.k] % W, py = m.q[m.k] / W;
              for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
                for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
                  const int q = yy * W + xx;
                  bool mine = false;
                  for (int z : m.q) if (z == q) { mine = true; break; }
                  const bool used = mine || own[q] < m.r || F.rankOf[q] < m.r;
                  if (!used && aligned(F, q, m.ang)) {
                    if ((int)m.q.size() == CAP) { m.big = true; break; }
                    m.q.push_back(q); claims.push_back({q, m.r});
                    m.sx += F.c[q]; m.sy += F.s[q];
                    m.ang = fastAtan2(m.sy, m.sx) * kDEG_TO_RADS;
                  }
                }
              if (!m.big) ++m.k;
            }
            if (!any) break;
            ++groupSteps;
            for (auto& c : claims) claim(c.first, c.second);
          }
          // the members that outgrew the group: finished one after the other by the whole wave (the step in which a member hit the cap
          // is repeated there: the model restarts that queue entry)
          for (auto& m : grp) {
            if (!m.big || own[m.q[0]] != m.r) continue;
            ++handed;
            const int r = m.r;
            size_t k = m.k;
            while (k < m.q.size()) {
              const size_t nb = std::min<size_t>(8, m.q.size() - k);
              ++waveSteps;
              for (size_t e2 = k; e2 < k + nb; ++e2) {
                const int px = m.q[e2] % W, py = m.q[e2] / W;
                for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
                  for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
                    const int q = yy * W + xx;
                    const bool used = own[q] <= r || F.rankOf[q] < r;
                    if (!used && aligned(F, q, m.ang)) { claim(q, r); m.q.push_back(q); m.sx += F.c[q]; m.sy += F.s[q]; m.ang = fastAtan2(m.sy, m.sx) * kDEG_TO_RADS; }
                  }
              }
              k += nb;
            }
          }
        }
      }
    long wrong = 0, losers = 0;
    for (int q = 0; q < N; ++q) wrong += own[q] != truth[q];
    for (int r = 0; r < R; ++r) losers += lost[r];
    std::printf("octet groups of %d, cap %d px: %ld regions in %ld groups (%.2f per group), %ld handed to the whole wave; group steps %ld + whole-wave steps %ld = %ld "
                "(today's schedule in this model: 151063, sequential %ld); regions that lost a claimed pixel %ld; wrong after round 1 %ld px\n",
                G, CAP, regionsStarted, groups, (double)members / groups, handed, groupSteps, waveSteps, groupSteps + waveSteps, seqSteps, losers, wrong);
  }
  if (getenv("SIMW_GROUPS_ONLY")) return 0;
  const int Ks[] = {1, 8, 32, 128, 512, 2048, 0};
  for (int K : Ks) {
    auto winOf = [&](int r) -> int { return K == 0 ? binOf[r] : (int)((long)r * K / R); };
    std::vector<int> own(N, INT32_MAX);
    std::vector<size_t> idx(NT, 0);
    std::vector<double> clk(NT, 0.0), waited(NT, 0.0), busy(NT, 0.0);
    // prog[T] = the window of the tile's NEXT seed: everything below it is finished (published when the previous seed completes)
    std::vector<int> prog(NT, INT32_MAX);
    std::vector<double> progTime(NT, 0.0);
    std::vector<char> doneT(NT, 0);
    int left = 0;
    for (int T = 0; T < NT; ++T) { if (tileSeeds[T].empty()) doneT[T] = 1; else { prog[T] = winOf(tileSeeds[T][0]); ++left; } }
    long grownPix = 0, grownSteps = 0, grownRegions = 0, skipped = 0;
    auto nbReady = [&](int T, int w, double& when) -> bool {     // all 8 neighbours at window >= w?  when: the last one's arrival
      const int tx = T % TW, ty = T / TW;
      when = 0;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          if (!dx && !dy) continue;
          const int x = tx + dx, y = ty + dy;
          if (x < 0 || y < 0 || x >= TW || y >= TH) continue;
          const int U = y * TW + x;
          if (prog[U] < w) return false;
          when = std::max(when, progTime[U]);
        }
      return true;
    };
    std::vector<int> checked(NT, -1);                       // the window the tile has been admitted to
    while (left > 0) {
      int best = -1;
      double bestClk = 0;
      for (int T = 0; T < NT; ++T) {
        if (doneT[T]) continue;
        const int w = prog[T];
        double start = clk[T];
        if (K != 1 && w > checked[T]) {                     // a new window: the neighbours must have left the ones below
          double when;
          if (!nbReady(T, w, when)) continue;               // blocked
          start = when > clk[T] ? when + LAT : clk[T] + 0.1 * LAT;      // (a look that passes at once still costs a little)
        }
        if (best < 0 || start < bestClk) { best = T; bestClk = start; }
      }
      if (best < 0) { std::printf("deadlock?!\n"); return 1; }
      const int T = best;
      const int r = tileSeeds[T][idx[T]];
      if (prog[T] > checked[T]) { waited[T] += bestClk - clk[T]; checked[T] = prog[T]; }
      clk[T] = bestClk;
      const int sp = F.order[r];
      if (own[sp] < r) { ++skipped; clk[T] += 1.5; busy[T] += 1.5; }        // (a dead seed: its share of the row's liveness test)
      else {
        const long st = grow(F, r, [&](int q) { return own[q] < r || own[q] == r; }, [&](int q) { own[q] = std::min(own[q], r); }, reg);
        const double cost = 75.0 * st + 31.0 * (double)reg.size() + 40.0;
        clk[T] += cost; busy[T] += cost;
        grownPix += (long)reg.size(); grownSteps += st; ++grownRegions;
      }
      if (++idx[T] == tileSeeds[T].size()) { doneT[T] = 1; prog[T] = INT32_MAX; --left; }
      else prog[T] = winOf(tileSeeds[T][idx[T]]);
      progTime[T] = clk[T];
    }
    long wrong = 0;
    for (int q = 0; q < N; ++q) wrong += own[q] != truth[q];
    double makespan = 0, longest = 0, sumBusy = 0, sumWait = 0, sumLife = 0;
    for (int T = 0; T < NT; ++T) { makespan = std::max(makespan, clk[T]); longest = std::max(longest, busy[T]); sumBusy += busy[T]; sumWait += waited[T]; sumLife += clk[T]; }
    std::printf("K=%5d%s: grown %6ld regions %7ld px (%.3fx), cost %.2f M (%.3fx of sequential), wrong after round 1 %6ld px | makespan %.2f M, longest tile %.2f M, "
                "waiting %.1f%% of wave time\n", K, K == 0 ? " (bins)" : K == 1 ? " (free)" : "", grownRegions, grownPix, (double)grownPix / seqPix, sumBusy / 1e6,
                sumBusy / seqCost, wrong, makespan / 1e6, longest / 1e6, 100.0 * sumWait / sumLife);
  }
  return 0;
}
