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
