// Dev tool (CPU): simulates candidate parallel schedules of LSD region growing on one image and checks them
// against the sequential result, counting rounds and work.  Uses the oracle's ll_angle / ordering.
//   g++ -O2 -std=c++17 -I../../oracle -I../../include sim_tile_relax.cpp -o /tmp/sim_tile_relax
//   /tmp/sim_tile_relax img.raw W H [tile]
#include "line_oracle.hpp"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>
#include <utility>

using namespace orc;

struct Field {
  int W, H;
  std::vector<double> ang;      // radians or NOTDEF
  std::vector<float> c, s;      // per-pixel (float)cos/(float)sin of (double)(float)angle
  std::vector<int> order, rankOf;
  double prec;
  int minReg;
};

static bool aligned(const Field& F, int q, double theta) {
  const double a = F.ang[q];
  if (a == kNOTDEF) return false;
  double n = theta - a;
  if (n < 0) n = -n;
  if (n > kM_3_2_PI) { n -= kM_2__PI; if (n < 0) n = -n; }
  return n <= F.prec;
}

// Grow region of rank r; used(q) decides; claim(q) records.  Returns the pixel list.
template <class Used, class Claim>
static void grow(const Field& F, int r, Used used, Claim claim, std::vector<int>& reg, long& tests) {
  const int W = F.W, H = F.H;
  const int sp = F.order[r];
  reg.clear();
  double reg_angle = F.ang[sp];
  float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
  reg.push_back(sp);
  claim(sp);
  for (size_t k = 0; k < reg.size(); ++k) {
    const int px = reg[k] % W, py = reg[k] / W;
    for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
      for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
        const int q = yy * W + xx;
        ++tests;
        if (!used(q) && aligned(F, q, reg_angle)) {
          claim(q);
          reg.push_back(q);
          sumdx += F.c[q];
          sumdy += F.s[q];
          reg_angle = fastAtan2(sumdy, sumdx) * kDEG_TO_RADS;
        }
      }
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc < 4) { std::fprintf(stderr, "usage: img.raw W H [tile]\n"); return 2; }
  const int iw = atoi(argv[2]), ih = atoi(argv[3]);
  const int TS = argc > 4 ? atoi(argv[4]) : 64;
  Img8 img(iw, ih);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(img.d.data(), 1, img.d.size(), f) != img.d.size()) { std::fprintf(stderr, "read failed\n"); return 2; }
  fclose(f);
  LsdParams P;
  LsdDebug D;
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
  // the oracle's order covers every pixel with x<W-1,y<H-1; keep the defined ones only (the undefined never seed)
  for (int p : D.order) if (F.ang[p] != kNOTDEF) F.order.push_back(p);
  const int R = (int)F.order.size();
  F.rankOf.assign(N, INT32_MAX);
  for (int r = 0; r < R; ++r) F.rankOf[F.order[r]] = r;
  F.prec = kPI * P.ang_th / 180;
  {
    const double p = P.ang_th / 180;
    const double LOG_NT = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
    F.minReg = (int)size_t(-LOG_NT / std::log10(p));
  }
  std::printf("image %dx%d scaled %dx%d, defined %d, minReg %d, tile %d\n", iw, ih, W, H, R, F.minReg, TS);

  // ---- truth: sequential ----
  std::vector<int> truth(N, INT32_MAX);
  std::vector<int> reg;
  long tests = 0, regions = 0, bigRegions = 0, bigPix = 0, accepts = 0, aloneByAngle = 0;
  std::map<int, int> sizeHist;
  for (int r = 0; r < R; ++r) {
    if (truth[F.order[r]] != INT32_MAX) continue;
    grow(F, r, [&](int q) { return truth[q] != INT32_MAX; }, [&](int q) { truth[q] = r; }, reg, tests);
    ++regions;
    if (reg.size() == 1) {                         // (how many of the one-pixel regions are alone by ANGLE: no neighbour aligned with the seed, used or not)
      const int sp = F.order[r], px = sp % W, py = sp / W;
      bool alone = true;
      for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1); ++yy)
        for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx)
          if (yy * W + xx != sp && aligned(F, yy * W + xx, F.ang[sp])) alone = false;
      aloneByAngle += alone;
    }
    accepts += (long)reg.size();
    sizeHist[std::min((int)reg.size(), 32)]++;
    if ((int)reg.size() >= F.minReg) { ++bigRegions; bigPix += (long)reg.size(); }
  }
  std::printf("sequential: %ld regions (%ld >= minReg holding %ld px), %ld pixels, %ld neighbour tests\n", regions, bigRegions, bigPix,
              accepts, tests);
  std::printf("one-pixel regions alone by angle (no neighbour aligned with the seed's own angle): %ld\n", aloneByAngle);
  std::printf("size histogram (size:count, 32 = 32+):");
  for (auto& kv : sizeHist) std::printf(" %d:%d", kv.first, kv.second);
  std::printf("\n");

  // ---- tile-sequential relaxation ----
  const int TW = (W + TS - 1) / TS, TH = (H + TS - 1) / TS, NT = TW * TH;
  std::vector<std::vector<int>> tileSeeds(NT);          // ranks, ascending
  for (int r = 0; r < R; ++r) {
    const int p = F.order[r];
    tileSeeds[(p / W / TS) * TW + (p % W) / TS].push_back(r);
  }
  for (int rule = 0; rule < 2; ++rule) {
    std::vector<int> prev(N, INT32_MAX), cur(N);
    std::printf("tile-sequential relaxation, rule %c:\n", rule ? 'B' : 'A');
    for (int t = 1; t <= 40; ++t) {
      std::fill(cur.begin(), cur.end(), INT32_MAX);
      long grown = 0, acc = 0, tst = 0;
      std::vector<int> mine(N, INT32_MAX);      // claims visible inside the running tile-wave (reset per tile via list)
      std::vector<int> touched;
      std::vector<char> doneRank;               // rule B: rank already processed this round by the running tile
      // SIM_PHASES=K: round 1 in K phases by rank (quantiles of the ordered list), a grid-wide barrier between them: a phase sees every
      // claim of the earlier phases (vis), whatever tile made it
      const int K = (t == 1 && getenv("SIM_PHASES")) ? std::max(1, atoi(getenv("SIM_PHASES"))) : 1;
      std::vector<int> vis(N, INT32_MAX);
      for (int ph = 0; ph < K; ++ph) {
      const int rLo = (int)((long)R * ph / K), rHi = (int)((long)R * (ph + 1) / K);
      for (int T = 0; T < NT; ++T) {
        touched.clear();
        const std::vector<int>& S = tileSeeds[T];
        const int tx0 = (T % TW) * TS, ty0 = (T / TW) * TS;
        for (int r : S) {
          if (r < rLo || r >= rHi) continue;
          const int sp = F.order[r];
          auto used = [&](int q) {
            if (vis[q] < r) return true;                     // claimed in an earlier phase of this round by a lower rank
            if (mine[q] <= r) return true;                   // claimed earlier this round by this tile-wave (lower rank)
            const int po = prev[q];
            if (po < r) {
              if (rule == 1) {
                // the previous owner is a seed of THIS tile with a lower rank: it has already been regrown this round by
                // this wave, and did not claim q (else mine[q] < r) -> q is free as far as that region is concerned
                const int op = F.order[po];
                const int ox = op % W, oy = op / W;
                if (ox >= tx0 && ox < tx0 + TS && oy >= ty0 && oy < ty0 + TS) return false;
              }
              return true;
            }
            return false;
          };
          if (used(sp)) continue;
          grow(F, r, used, [&](int q) { if (mine[q] == INT32_MAX) touched.push_back(q); mine[q] = std::min(mine[q], r); }, reg, tst);
          ++grown;
          acc += (long)reg.size();
        }
        for (int q : touched) { cur[q] = std::min(cur[q], mine[q]); mine[q] = INT32_MAX; }
      }
      if (K > 1) vis = cur;
      }
      long changed = 0, wrong = 0;
      std::vector<char> tileChanged(NT, 0);
      for (int q = 0; q < N; ++q) {
        if (cur[q] != prev[q]) { ++changed; tileChanged[(q / W / TS) * TW + (q % W) / TS] = 1; }
        if (cur[q] != truth[q]) ++wrong;
      }
      int tc = 0;
      for (char c : tileChanged) tc += c;
      std::printf("  round %2d: grown %6ld regions, %7ld px, changed %7ld px in %4d/%d tiles, wrong vs truth %7ld\n", t, grown, acc,
                  changed, tc, NT, wrong);
      prev.swap(cur);
      if (changed == 0) break;
    }
  }


  // ---- tile-sequential relaxation WITH carrying (the GPU schedule): rounds 1 full, round 2 by the foreign-owner rule,
  //      rounds >= 3 by the change rule; cells of 8x8; checks owner map AND every alive region's last-run list ----
  if (getenv("SIM_CARRY")) {
    const int CW = (W + 7) / 8, CH = (H + 7) / 8;
    std::vector<int> prev(N), cur(N), prev2(N);
    for (int q = 0; q < N; ++q) prev[q] = F.rankOf[q];          // trivial owner_0 (INT_MAX for undefined)
    std::vector<std::vector<int>> lastRun(R);                    // pixel list of the last run of every region
    std::vector<int> lastRound(R, 0);
    std::vector<int> lostStamp(R, 0);                            // GPU form: round in which the region last lost a contested claim
    const bool gpuRule = getenv("SIM_GPURULE") != nullptr;       // flags + map comparison instead of the pixel lists
    const bool full2 = getenv("SIM_FULL2") != nullptr;
    std::vector<int> boxes(4 * (size_t)R, 0);
    // truth lists
    std::vector<std::vector<int>> truthRun(R);
    {
      std::vector<int> own(N, INT32_MAX);
      long tt = 0;
      for (int r = 0; r < R; ++r) {
        if (own[F.order[r]] != INT32_MAX) continue;
        grow(F, r, [&](int q) { return own[q] != INT32_MAX; }, [&](int q) { own[q] = r; }, reg, tt);
        truthRun[r] = reg;
      }
    }
    auto tileOf = [&](int p) { return (p / W / TS) * TW + (p % W) / TS; };
    for (int t = 1; t <= 40; ++t) {
      std::vector<char> dirty(R, (t <= 1 || (t == 2 && full2)) ? 1 : 0);
      std::vector<int> cellMin(CW * CH, INT32_MAX);
      long changed = 0;
      if (t == 2 && getenv("SIM_LOST")) {
        // candidate rule: a region is regrown in round 2 iff it LOST a pixel it claimed in round 1 (to a lower rank), or died
        for (int r = 0; r < R; ++r) {
          if (lastRound[r] != 1) continue;
          if (gpuRule) { if (lostStamp[r] == 1) dirty[r] = 1; }
          else for (int q : lastRun[r]) if (prev[q] != r) { dirty[r] = 1; break; }
        }
        for (int q = 0; q < N; ++q) {
          const int o = prev[q];
          if (o == INT32_MAX) continue;
          if (prev[F.order[o]] != o) dirty[o] = 1;
        }
        changed = 1;
      } else if (t == 2) {
        for (int q = 0; q < N; ++q) {
          const int o = prev[q];
          if (o == INT32_MAX) continue;
          if (prev[F.order[o]] != o) dirty[o] = 1;     // the owner died after it ran: its pixels are released
          if (tileOf(F.order[o]) != tileOf(q)) cellMin[(q / W / 8) * CW + (q % W) / 8] = std::min(cellMin[(q / W / 8) * CW + (q % W) / 8], o);
          if (F.order[o] == q) {
            const int T = tileOf(q), tx0 = (T % TW) * TS, ty0 = (T / TW) * TS;
            const int* b = &boxes[4 * (size_t)o];
            if (b[0] - 1 < tx0 || b[1] - 1 < ty0 || b[2] + 1 >= tx0 + TS || b[3] + 1 >= ty0 + TS) dirty[o] = 1;
          }
        }
        changed = 1;
      } else if (t >= 3) {
        for (int q = 0; q < N; ++q)
          if (prev[q] != prev2[q]) {
            ++changed;
            int& m = cellMin[(q / W / 8) * CW + (q % W) / 8];
            m = std::min(m, std::min(prev[q], prev2[q]));
          }
        if (!changed) {
          std::printf("  carry: fixed point detected at round %d\n", t);
          int shown = 0;
          for (int r = 0; r < R && shown < 6; ++r)
            if (prev[F.order[r]] == r && lastRun[r] != truthRun[r]) {
              ++shown;
              const int sp = F.order[r];
              std::printf("   bad region r=%d seed (%d,%d) tile %d last run in round %d: %zu px, truth %zu px\n", r, sp % W, sp / W, tileOf(sp),
                          lastRound[r], lastRun[r].size(), truthRun[r].size());
              for (size_t k = 0; k < std::max(lastRun[r].size(), truthRun[r].size()); ++k) {
                const int a = k < lastRun[r].size() ? lastRun[r][k] : -1, b = k < truthRun[r].size() ? truthRun[r][k] : -1;
                if (a != b) {
                  std::printf("     first difference at list position %zu: last (%d,%d) truth (%d,%d)\n", k, a % W, a / W, b % W, b / W);
                  for (int q : {a, b}) if (q >= 0) {
                    const int o = prev[q];
                    std::printf("       pixel (%d,%d): rank %d, final owner %d (seed tile %d), truth owner %d, pixel tile %d\n", q % W, q / W, F.rankOf[q], o,
                                o == INT32_MAX ? -1 : tileOf(F.order[o]), truth[q], tileOf(q));
                  }
                  break;
                }
              }
            }
          break;
        }
      }
      if (t >= 3 && getenv("SIM_EXACT")) {
        // candidate rule for the later rounds: region o (alive in owner_{t-1}) is regrown iff
        //  (a) it lost a pixel of its last run (owner_{t-1}[q] != o for a q of the list), or
        //  (b) a neighbour q of one of its pixels was held by a lower rank in owner_{t-2} and is not any more (released or passed
        //      to a higher rank): o may take it now;  plus the "seed changed hands" rule.
        if (!gpuRule)
          for (int r = 0; r < R; ++r) {
            if (prev[F.order[r]] != r || lastRound[r] == 0) continue;
            for (int q : lastRun[r]) if (prev[q] != r) { dirty[r] = 1; break; }
          }
        for (int p = 0; p < N; ++p) {
          const int o = prev[p];
          if (o == INT32_MAX) continue;
          if (gpuRule && lostStamp[o] == t - 1) dirty[o] = 1;       // lost a contested claim in the last round
          const int x = p % W, y = p / W;
          for (int yy = std::max(y - 1, 0); yy <= std::min(y + 1, H - 1); ++yy)
            for (int xx = std::max(x - 1, 0); xx <= std::min(x + 1, W - 1); ++xx) {
              const int q = yy * W + xx;
              if (prev2[q] < o && prev[q] > o && prev[q] != INT32_MAX) dirty[o] = 1;
              if (gpuRule && prev2[q] == o && prev[q] < o) dirty[o] = 1;      // lost q across the rounds
            }
          const int r = F.rankOf[p];
          const bool a1 = prev[p] == r, a2 = prev2[p] == r;
          if (a1 != a2) dirty[r] = 1;
        }
      } else if (t >= 2) {
        for (int q = 0; q < N; ++q) {
          const int o = prev[q];
          if (o == INT32_MAX) continue;
          const int x = q % W, y = q / W;
          int m = INT32_MAX;
          for (int dy = -1; dy <= 1; dy += 2)
            for (int dx = -1; dx <= 1; dx += 2) {
              const int xx = std::min(std::max(x + dx, 0), W - 1), yy = std::min(std::max(y + dy, 0), H - 1);
              m = std::min(m, cellMin[(yy / 8) * CW + xx / 8]);
            }
          if (m < o) dirty[o] = 1;
          if (t >= 3) {
            const int r = F.rankOf[q];
            const bool a1 = prev[q] == r, a2 = prev2[q] == r;
            if (a1 != a2) dirty[r] = 1;
          }
        }
      }
      // owner_t start: carried regions keep their pixels, everything else its own rank
      for (int q = 0; q < N; ++q) {
        const int o = prev[q];
        cur[q] = o == INT32_MAX ? INT32_MAX : (dirty[o] ? F.rankOf[q] : o);
      }
      long grown = 0, acc = 0, tst = 0;
      // SIM_LANESPEC=cap (round 1 only): the alive seeds of a row of 64 list entries grow at the same time, one per lane, alone
      // against the state at the start of the row, up to `cap` pixels; the ones that stay below the cap claim their pixels together
      // (lowest rank wins a pixel, the others are stamped losers and regrown in round 2 by the usual rule), the ones that hit the cap
      // are grown afterwards by the whole wave, in rank order, like every region is without this switch.
      const int laneCap = (t == 1 && getenv("SIM_LANESPEC")) ? atoi(getenv("SIM_LANESPEC")) : 0;
      long specSmall = 0, specBig = 0, specLosers = 0, specKilledBig = 0, specRows = 0, specSteps = 0;
      for (int T = 0; T < NT; ++T) {
        std::vector<std::pair<int, int>> claims;
        std::vector<int> mineIdx;
        static std::vector<int> mine;
        if ((int)mine.size() != N) mine.assign(N, INT32_MAX);
        if (laneCap > 0) {
          const std::vector<int>& S = tileSeeds[T];
          auto record = [&](int r, const std::vector<int>& rg) {
            lastRun[r] = rg;
            lastRound[r] = t;
            int* b = &boxes[4 * (size_t)r];
            b[0] = b[2] = rg[0] % W; b[1] = b[3] = rg[0] / W;
            for (int q : rg) { b[0] = std::min(b[0], q % W); b[2] = std::max(b[2], q % W); b[1] = std::min(b[1], q / W); b[3] = std::max(b[3], q / W); }
            ++grown; acc += (long)rg.size();
          };
          auto claim = [&](int q, int r) {
            if (mine[q] == INT32_MAX) mineIdx.push_back(q);
            if (mine[q] != INT32_MAX && mine[q] != r) lostStamp[std::max(mine[q], r)] = t;     // contested inside the tile: the higher rank loses
            mine[q] = std::min(mine[q], r);
          };
          // SIM_SUPERROW: a row = the next 64 seeds that are ALIVE when the row is formed (the tile's list is scanned on), not 64 list entries
          const bool superRow = getenv("SIM_SUPERROW") != nullptr;
          std::vector<int> SR;
          size_t scan = 0;
          for (size_t base = 0; superRow ? scan < S.size() : base < S.size(); base += 64) {
            if (superRow) {
              SR.clear();
              while (scan < S.size() && SR.size() < 64) {
                const int r = S[scan++], sp = F.order[r];
                if (!(prev[sp] != r || mine[sp] < r || cur[sp] < r)) SR.push_back(r);
              }
              if (SR.empty()) break;
            }
            const std::vector<int>& S_ = superRow ? SR : S;
            const size_t base_ = superRow ? 0 : base;
            ++specRows;
            const size_t n = superRow ? SR.size() : std::min<size_t>(64, S.size() - base);
            std::vector<std::vector<int>> RG(n);
            std::vector<int> st(n, -1);            // -1 dead at row start, 0 small, 1 big
            long maxSteps = 0;
            for (size_t j = 0; j < n; ++j) {
              const int r = S_[base_ + j], sp = F.order[r];
              if (prev[sp] != r || mine[sp] < r || cur[sp] < r) continue;
              auto used = [&](int q) { return mine[q] <= r || prev[q] < r || cur[q] < r; };
              std::vector<int>& rg = RG[j];
              double reg_angle = F.ang[sp];
              float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
              rg.push_back(sp);
              bool big = false;
              size_t k = 0;
              for (; k < rg.size() && !big; ++k) {
                const int px = rg[k] % W, py = rg[k] / W;
                for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1) && !big; ++yy)
                  for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
                    const int q = yy * W + xx;
                    if (used(q) || std::find(rg.begin(), rg.end(), q) != rg.end()) continue;
                    if (aligned(F, q, reg_angle)) {
                      if ((int)rg.size() == laneCap) { big = true; break; }
                      rg.push_back(q);
                      sumdx += F.c[q]; sumdy += F.s[q];
                      reg_angle = fastAtan2(sumdy, sumdx) * kDEG_TO_RADS;
                    }
                  }
              }
              maxSteps = std::max<long>(maxSteps, (long)k);
              st[j] = big ? 1 : 0;
            }
            specSteps += maxSteps;
            for (size_t j = 0; j < n; ++j)
              if (st[j] == 0) { for (int q : RG[j]) claim(q, S_[base_ + j]); record(S_[base_ + j], RG[j]); ++specSmall; }
            for (size_t j = 0; j < n; ++j)
              if (st[j] == 0) for (int q : RG[j]) if (mine[q] != S_[base_ + j]) { ++specLosers; break; }
            for (size_t j = 0; j < n; ++j) {
              if (st[j] != 1) continue;
              const int r = S_[base_ + j], sp = F.order[r];
              if (mine[sp] < r) { ++specKilledBig; continue; }
              auto used = [&](int q) { return mine[q] <= r || prev[q] < r || cur[q] < r; };
              grow(F, r, used, [&](int q) { claim(q, r); }, reg, tst);
              record(r, reg);
              ++specBig;
            }
          }
        } else
        for (int r : tileSeeds[T]) {
          const int sp = F.order[r];
          if (!dirty[r] || prev[sp] != r) continue;
          auto used = [&](int q) { return mine[q] <= r || prev[q] < r || cur[q] < r; };   // (cur: the pre-claims of carried lower ranks)
          if (mine[sp] < r || cur[sp] < r) continue;
          grow(F, r, used, [&](int q) { if (mine[q] == INT32_MAX) mineIdx.push_back(q); mine[q] = std::min(mine[q], r); }, reg, tst);
          lastRun[r] = reg;
          lastRound[r] = t;
          int* b = &boxes[4 * (size_t)r];
          b[0] = b[2] = reg[0] % W; b[1] = b[3] = reg[0] / W;
          for (int q : reg) { b[0] = std::min(b[0], q % W); b[2] = std::max(b[2], q % W); b[1] = std::min(b[1], q / W); b[3] = std::max(b[3], q / W); }
          ++grown; acc += (long)reg.size();
        }
        for (int q : mineIdx) { claims.push_back({q, mine[q]}); mine[q] = INT32_MAX; }
        // claims of this tile land after the tile is done (no in-round visibility across tiles in this model)
        static std::vector<std::pair<int, int>> all;
        if (T == 0) all.clear();
        all.insert(all.end(), claims.begin(), claims.end());
        if (T == NT - 1)
          for (auto& c : all) {
            const int old = cur[c.first];
            if (old < c.second) lostStamp[c.second] = t;
            else if (old > c.second && old != F.rankOf[c.first] && old != INT32_MAX) lostStamp[old] = t;
            cur[c.first] = std::min(old, c.second);
          }
      }
      long wrong = 0, wrongRuns = 0;
      for (int q = 0; q < N; ++q) if (cur[q] != truth[q]) ++wrong;
      for (int r = 0; r < R; ++r) if (cur[F.order[r]] == r && lastRun[r] != truthRun[r]) ++wrongRuns;
      std::printf("  carry round %2d: grown %6ld regions, %7ld px, changed-in %7ld, owner wrong %6ld, alive regions with a wrong last run %5ld\n",
                  t, grown, acc, changed, wrong, wrongRuns);
      if (laneCap > 0)
        std::printf("    lane speculation (cap %d): %ld rows, %ld lane-parallel steps, %ld small regions (%ld of them lost a pixel), %ld big, %ld big seeds taken first\n",
                    laneCap, specRows, specSteps, specSmall, specLosers, specBig, specKilledBig);
      prev2 = prev;
      prev = cur;
    }
  }

  // ---- lane speculation inside a sequential wave: super-rows of 64 live seeds, cap C ----
  for (int C : {4, 6, 8, 12}) {
    std::vector<int> own(N, INT32_MAX);
    long rows = 0, lanesLive = 0, lanesBig = 0, lanesLoser = 0, lanesClean = 0, lanesKilled = 0, regrowAcc = 0, cleanAcc = 0, specPops = 0;
    int r = 0;
    std::vector<int> lane(64);
    std::vector<std::vector<int>> R_(64);
    std::vector<int> status(64);      // 0 clean small, 1 big, 2 loser
    std::vector<int> tag(N, -1);
    while (r < R) {
      int n = 0;
      while (r < R && n < 64) { if (own[F.order[r]] == INT32_MAX) lane[n++] = r; ++r; }
      if (!n) break;
      ++rows;
      lanesLive += n;
      long maxPops = 0;
      for (int j = 0; j < n; ++j) {
        // speculative growth against the committed state, cap C
        std::vector<int>& rg = R_[j];
        rg.clear();
        const int rk = lane[j];
        const int sp = F.order[rk];
        double reg_angle = F.ang[sp];
        float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
        rg.push_back(sp);
        bool big = false;
        size_t k = 0;
        for (; k < rg.size() && !big; ++k) {
          const int px = rg[k] % W, py = rg[k] / W;
          for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H - 1) && !big; ++yy)
            for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W - 1); ++xx) {
              const int q = yy * W + xx;
              if (own[q] != INT32_MAX) continue;
              if (std::find(rg.begin(), rg.end(), q) != rg.end()) continue;
              if (aligned(F, q, reg_angle)) {
                if ((int)rg.size() == C) { big = true; break; }
                rg.push_back(q);
                sumdx += F.c[q]; sumdy += F.s[q];
                reg_angle = fastAtan2(sumdy, sumdx) * kDEG_TO_RADS;
              }
            }
        }
        maxPops = std::max<long>(maxPops, (long)k);
        status[j] = big ? 1 : 0;
      }
      specPops += maxPops;
      // tags: arbitrary winner among the small lanes (here: the last writer), losers = lanes that do not own all their pixels
      for (int j = 0; j < n; ++j) if (status[j] == 0) for (int q : R_[j]) tag[q] = j;
      for (int j = 0; j < n; ++j) if (status[j] == 0) for (int q : R_[j]) if (tag[q] != j) { status[j] = 2; break; }
      // ordered resolution
      for (int j = 0; j < n; ++j) {
        const int rk = lane[j];
        if (status[j] == 0) {
          for (int q : R_[j]) own[q] = rk;
          ++lanesClean;
          cleanAcc += (long)R_[j].size();
          continue;
        }
        if (status[j] == 1) ++lanesBig; else ++lanesLoser;
        if (own[F.order[rk]] != INT32_MAX) { ++lanesKilled; continue; }
        long tt = 0;
        grow(F, rk, [&](int q) { return own[q] != INT32_MAX; },
             [&](int q) {
               own[q] = rk;
               const int L = tag[q];
               if (L > j && L < n && status[L] == 0 && std::find(R_[L].begin(), R_[L].end(), q) != R_[L].end()) status[L] = 2;
             }, reg, tt);
        regrowAcc += (long)reg.size();
      }
      for (int j = 0; j < n; ++j) for (int q : R_[j]) tag[q] = -1;
    }
    long wrong = 0;
    for (int q = 0; q < N; ++q) if (own[q] != truth[q]) ++wrong;
    std::printf("lane speculation C=%2d: %ld super-rows, live %ld: clean %ld (%ld px), big %ld, losers %ld (killed %ld), wave-wide regrown px %ld, "
                "spec pops %ld, wrong %ld\n", C, rows, lanesLive, lanesClean, cleanAcc, lanesBig, lanesLoser, lanesKilled, regrowAcc, specPops, wrong);
  }
  return 0;
}
