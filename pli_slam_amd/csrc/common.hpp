// Shared host/device declarations of the MI355X point+line front-end.
// gfx950 only: wave64, no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include "../../include/pli_frontend.h"

namespace pli {

constexpr int MAX_LEVELS = 16;
constexpr int CELL_CAP = 512;          // FAST survivors a 30..60 px cell can hold after NMS
constexpr int GRID_COLS = 64;          // FRAME_GRID_COLS, Frame.h:60
constexpr int GRID_ROWS = 48;          // FRAME_GRID_ROWS, Frame.h:59
constexpr int LSD_CHUNK = 4096;        // pixels per chunk of the ordered-list build (histogram, scan, stable scatter)

// Geometry of one pyramid level and of its FAST cell grid
// (ORBextractor.cc:763-806 restated once on the host).
struct LevelGeom {
  int w, h, pitch;           // level image
  int64_t offset;            // byte offset of the level inside one image's pyramid block
  int minBX, minBY, maxBX, maxBY;
  int nCols, nRows, wCell, hCell;
  int cellBase;              // index of this level's first cell in the per-image cell array
  int nfeatures;             // mnFeaturesPerLevel
  int kpBase;                // first slot of this level in the per-image keypoint slot array
  int kpCap;                 // slots (nfeatures + 4)
  int candBase;              // first entry of this level in the per-image candidate array
  int candCap;
  float scale, invScale;     // mvScaleFactor, mvInvScaleFactor
  int nIni;                  // quadtree root nodes
  float hX;
};

// Everything the kernels need to know about the configuration, by value.
struct DevParams {
  int W, H;                  // input image
  int nlevels;
  int iniTh, minTh;
  LevelGeom lv[MAX_LEVELS];
  int cellsPerImage, kpSlotsPerImage, candPerImage;
  int64_t pyrBlock;          // bytes of one image's pyramid (levels 1..n-1; level 0 aliases the input)
  int kpCap, klCap;
  int umax[16];
  // LSD
  int LW, LH, lpitch;        // scaled image: LW = the row PITCH of the detector's per-pixel planes (LWt rounded up to 16), lpitch: the u8 plane's
  int LWt;                   // true width of the scaled image (front pass, resize tables, LOG_NT, debug getters: nothing else)
  int g2Thresh;              // defined  <=>  g2 > g2Thresh   (norm > rho)
  int nBins;
  int minRegSize;
  double prec, lsdScale;
  float alignLo, alignHi;     // cos^2(prec + margin), cos^2(prec - margin): bounds of the vector form of the alignment test
  double hotBand2;            // hot records, test switch: added to the band inside which a candidate of the exact test takes the exact sums (rad; 0)
  double rectApproxBand;      // region2rect, a region with approximate sums: |angle_diff(theta, reg_angle) - prec| below this -> the exact sums (rad)
  int alignFilter, alignPad;  // 0: prec too wide for the vector form, every test takes the exact path
  int parityFlags, parityPad; // pli_frontend_config::parity_flags (PLI_PARITY_*)
  double rho;                 // LSD gradient threshold quant / sin(prec) (the CV_64F pipeline compares the double norm with it)
  int maxLines;
  int lsdNFeatures;
  double minLength;
  // stereo
  float bf, maxD;
  // line matching
  int sWs, bestLR;
  double lineSimTh, overlapTh, ratio12L, minDispRatio, minDisp, horizTh;
};

// per-image control block of the LSD relaxation (lsd_relax.hip)
// Returning atomics on one cache line serialise in its L2 channel (~40 ns each), so every hot counter has its own
// 128-byte line, and the two counters every finished region needs are one 64-bit word.
struct alignas(128) RxCtl {
  int state;       // 0 relaxing, 2 exact owner map found
  int changed;     // owner_{t-1} != owner_{t-2} somewhere
  int overflow;    // a capacity was exhausted (code): the image falls back to the sequential grower
  int rounds;      // round in which the fixed point was detected
  int nSmall, nBig;   // work lists of this round: lane grower, wave grower
  int changedOdd;     // (tile relaxation, fused diff + mark: the flag of the odd rounds; `changed` serves the even ones)
  int pad0[25];
  int nHand;       // regions handed from the lane grower to the wave grower in this round
  int pad1[31];
  int nextBig;     // work counter of the wave grower
  int pad2[31];
  // rect list entries (high 24 bits) | arena words (low 40 bits): lists for k_rx_rect and queue overflow blocks
  unsigned long long rectArena;
  int pad3[30];
  int races;       // regions that saw a lower rank slip between their owner load and their claim
  int pad4[31];
};
constexpr int RX_ARENA_BITS = 40;

// an alive seed, ready to grow (written by k_rx_seed)
struct RxSeed {
  int rank, xy;
  float ang, sx, sy;
};

constexpr int RX_QCAP = 16;            // queue entries a lane keeps in LDS
constexpr int RX_HAND = RX_QCAP - 8;   // a lane hands its region to the wave grower at this size (step boundary)

// a completed region of at least minRegSize pixels: its pixel list (arena) goes to k_rx_rect
struct RxRect {
  int rank, off, cnt;
  float sumdx, sumdy;
  int approx;      // 1: the sums are the hot-record grower's filter sums (lsd_tile.hip "HOT RECORDS"): within 2e-4 rad of the exact ones in
                   // direction — region2rect recomputes the exact ones where its one use of the region angle is that close to call (lsd_rect.hpp)
};

// tile-sequential relaxation, rounds >= 2: the seeds stamped dirty in a round, per tile of their seed pixel (appended by whoever
// stamps the region first; the tile's wave sorts its few entries by rank instead of walking its whole seed list).
// list == nullptr: no lists (the rank-ordered relaxation shares k_rx_mark).
struct TxDirtyLists {
  int2* list;          // [image][tile][ts * ts] (rank, seed pixel)
  int* cnt;            // [image][tile]
  const int* order;    // [image][npix] seed pixel of a rank
  int ts, ntx, nty, W;
  int64_t npix;
  int rmask = -1;              // index of a region in the per-region planes = its id & rmask: -1 when ids are dense ranks (the ordered
                               // list), (1 << pixbits) - 1 when they are KEYS (gradient bin << pixbits | seed pixel, k_tx_sort): the seed pixel
  const int* perm = nullptr;   // round 1: workgroup -> (image, tile) in order of decreasing work (k_tx_order), or null: the grid order
  int xcdAffine = 0;           // round 1: all tiles of an image on ONE XCD (workgroups go to the 8 XCDs round robin)
};

// "no rank / no id": what the rank plane (k_lsd_scatter) and the id plane (k_tx_sort, key mode) hold for a pixel without a level-line
// angle.  INT_MAX since round 5 (0x7F7F7F7F before): key-mode ids use all 31 bits on images of more than 2^20 scaled pixels.
constexpr int LSD_ID_INF = 0x7FFFFFFF;
constexpr int TX_EMIT_CAP = 16384;   // key mode: candidate segments of an image that k_tx_emit_sorted can order (LDS)
// k_zero_ranges (orb_kernels.hip): buffers cleared by one launch
struct ZeroRanges {
  uint32_t* p[12];
  int64_t words[12];
};

// k_tx_sort in key mode (lsd_tile.hip)
struct TxKeys {
  const double* mg;                 // gradient norm plane, or null: rank mode
  const unsigned long long* maxMg;  // per image: bits of the largest norm
  double rho;
  int nBins, pixbits;
  int* idPlane;                     // out: own id of every pixel (TX_INF: undefined)
  float4* recPack = nullptr;        // packed round 1 (lsd_tile.hip): the pixel records, whose fourth word is owner_1 during round 1
  int* zeroA = nullptr;             // planes of one word per pixel index that the sort clears on its way (rgDirty, rgLost), or null
  int* zeroB = nullptr;
  int lazyMargin = 8;               // LAZY ids: units of the 2^-22 fixed point around a bin boundary that the double plane decides (>= 4; test switch)
  int2* hot = nullptr;              // round 6: round 1's words live in 8-byte hot records {angle, owner word} instead (lsd_tile.hip "HOT RECORDS")
  const float2* cold = nullptr;     // ... and the records' exact {cos, sin} in a plane of their own when the 16-byte records are not written
  int pack = 0;                     // 0: owner plane; 1: k_tx_sort writes the unclaimed words (ids); 2: the front pass wrote them (LAZY ids)
};

// arguments of k_tx_tail (lsd_tile.hip): the rounds t >= t0 of the tile relaxation in one persistent launch
struct TxTailArgs {
  const DevParams* Pp;
  RxCtl* ctl;
  const float4* rec;
  int2* own;
  const int2* list;
  const int* tileCnt;
  int ts, ntx, nty;
  int* rgSize;
  int2* rgBox;
  int* rgDirty;
  int* tileAct;
  int* tileTouch;
  int TW, TH;
  int* arena;
  int arenaCap;
  RxRect* rects;
  int rectCap;
  float4* rgSeg;
  const double* mg;
  const int* rank;
  int* rgLost;
  TxDirtyLists DL;
  int img0, nimg, t0, maxRounds;
  unsigned* bar;       // [0] arrivals (zeroed by the host before the launch), [32] abort word (its own line)
  int forceAbort = 0;  // test switch: behave as if the first grid barrier had timed out
  int2* hot = nullptr; // the hot records {angle, ..}: the growers take the angles from there (and `rec` is not read), or null
  const float2* cold = nullptr;   // ... with the exact {cos, sin} plane beside them
};

// a region in mid-growth, handed from the lane grower to the wave grower
struct RxHand {
  int rank, k, cnt;
  float sumdx, sumdy;
  int box0, box1, pad;
  int q[RX_QCAP];
};

// pli_batch_track: parameters + table / track record offsets, by value (match_kernels.hip: k_track_*)
struct TrackParams {
  float fx, fy, cx, cy, bf, th, minX, maxX, minY, maxY;
  int mono, checkOri;
  float nnrLines;
  int kpCap, klCap;
  int64_t recordBytes, offCounts, offKp0, offDesc0, offUr, offDepth, offLd0;
  int64_t trackBytes, toffCounts, toffBest, toffLines;
};

// a region handed from the sequential LSD grower to k_lsd_rect: its pixel list in the image's arena and its angle
struct LsdRectItem { int off, cnt; double reg_angle; };

// KannalaBrandt8 parameters (pli_kb8_camera), by value (match_kernels.hip: k_fisheye_triangulate)
struct Kb8 { float fx, fy, cx, cy, k0, k1, k2, k3; };

}  // namespace pli
