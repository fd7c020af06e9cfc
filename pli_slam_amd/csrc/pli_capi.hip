// Host side of libpli_frontend.so: context, geometry tables, stage scheduling
// on one HIP stream, and the extern "C" entry points of include/pli_frontend.h.
// There is no CPU fallback: without a gfx950 device every compute entry point
// fails with PLI_ERR_NO_DEVICE.
#include "kernels.hpp"
#include <cmath>
#include <cfloat>
#include <algorithm>
#include <memory>
#include <limits>
#include <cstdlib>
#include <set>
#include <string>
#include <functional>
#include <mutex>
#include <atomic>
#include <dlfcn.h>

using namespace pli;

// Development switches (tools/README.md "Environment switches"): schedules and kernels that were measured and shelved, test switches,
// tuning knobs.  They are read ONLY by the development build of the library (make dev: -DPLI_DEV -> libpli_frontend_dev.so, which the dev-switch
// tests and the tools load); in the product library DEVENV is a null constant — the variable names are not even in the binary, and a
// stray variable in an integrator's environment cannot change a schedule.  The product reads four documented variables: PLI_ROCTX (roctx
// ranges), PLI_SYNC_DEBUG (synchronise after every launch), PLI_TX_TAIL (= 0: no persistent kernel, include/pli_frontend.h "Sharing a
// device") and PLI_LSD_MODE (overrides pli_frontend_config.lsd_mode).
#ifdef PLI_DEV
#define DEVENV(name) getenv(name)
#else
static inline const char* pli_no_devenv() { return nullptr; }
#define DEVENV(name) (pli_no_devenv())
#endif

namespace {

thread_local std::string g_err;

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e__);                            \
      return PLI_ERR_HIP;                                                                    \
    }                                                                                        \
  } while (0)

inline int64_t alignUp(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline int cvRoundf(float v) { return (int)std::nearbyintf(v); }
inline int cvRoundd(double v) { return (int)std::nearbyint(v); }
inline int cvFloorf(float v) { int i = (int)v; return i - (i > v); }

// lsd_mode auto: batches of at least this many images (2 per stereo frame) take the sequential wave grower (one wave
// per image: work-efficient, its throughput keeps growing with the batch); below it the tile-sequential relaxation
// (lsd_tile.hip: parallel inside an image, ~1.5x the sequential work per frame at large batches).  Measured, 752x480,
// stereo frames/s (end of round 3, tile vs sequential): 768 frames 5489 vs 3715, 1024 frames 5440 vs 4584, 1280 frames 5497 vs 5435,
// 1536 frames 5550 vs 5250, 1792 frames 5557 vs 5702, 2048 frames 5995 (sequential).  (Round 2: crossover at 1024 frames.)
// The hand-over is put at 1280 frames, where the two are level: the relaxations' buffers cost 25 MB per image, and a context of
// 1600 frames with them would take 250 of the 288 GB.
constexpr int RX_AUTO_IMAGES = 2560;     // images (2 per stereo frame): 1280 frames
// tiles of 32 for contexts of up to 32 frames of 752x480 (four times the waves where 64-pixel tiles leave the chip under-occupied: a
// single stereo pair takes 4.6 instead of 7.3 ms with the round-1 code; now, frames/s with 32 vs 64: 16 frames 2228 vs 2066,
// 32 frames 2930 vs 2777; from 64 frames on the extra border conflicts cost more: 3343 vs 3445, 128 frames 3568 vs 3782)
// — in tile waves: 32 frames of 752x480 are 64 x 135 = 8640 waves of 64-pixel tiles, about what the chip holds at once (7168)
constexpr int TX_SMALL_TILE_WAVES = 12000;

struct ProfEntry { const char* name; hipEvent_t a, b; };

}  // namespace

struct pli_ctx {
  // Every entry point that takes a context holds this lock for the whole call: the reference drives its four extractors
  // from four std::threads (Frame.cc:128-135) and the adapters put them on ONE context, so concurrent calls on a context
  // are legal and are serialised here (recursive: entry points call each other).  Different contexts run concurrently.
  mutable std::recursive_mutex mu;
  pli_frontend_config cfg;
  DevParams hp;
  DevParams* dP = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  bool ownStream = true;
  // side stream: for small batches the ORB chain runs beside the line chain (fork after the ingest, join before the stereo
  // stage); both chains are launch/latency bound there (+4 % on a single pair, +2.5 % at 32 frames, nothing from 256 frames on)
  hipStream_t aux = nullptr;
  hipEvent_t evFork = nullptr, evJoin = nullptr, evLbdPre = nullptr;
  hipStream_t aux2 = nullptr;        // round 1's region2rect pass beside round 2's pass over the owner map (runLines)
  hipEvent_t evRectFork = nullptr, evRectDone = nullptr;
  std::function<pli_status()> sideChain;     // pli_batch_run -> runLines: enqueues the side stream's work (see pli_batch_run)
  bool lbdPreOnSide = false;                 // the LBD's blur + Sobel of this call ran on the side stream (pli_batch_run)
  bool syncDebug = getenv("PLI_SYNC_DEBUG") != nullptr;
  int NI = 0;
  pli_table_layout lay;
  std::vector<void*> allocs;
  // ORB
  uint8_t *pyr = nullptr, *blur = nullptr;
  int* resizeTab[MAX_LEVELS] = {nullptr};
  uint32_t* cellCand = nullptr; int* cellCount = nullptr;
  uint32_t* candAll = nullptr; unsigned short* nodeOf = nullptr; int* candCount = nullptr;
  uint32_t* kpSel = nullptr; int* kpSelCount = nullptr;
  BlurJob *jobOrb = nullptr, *jobLsd = nullptr, *jobLbd = nullptr;
  int orbTiles = 0, l0Tiles = 0;
  // LSD / LBD
  uint8_t *tmp8 = nullptr, *lsdScaled = nullptr;
  int64_t tmp8Stride = 0, lsdStride = 0;
  int tmpPitch = 0;
  int* lsdTab = nullptr;
  float4* rec = nullptr; int* g2 = nullptr; int* maxG2 = nullptr;
  // CV_64F pipeline of the LSD front (PLI_PARITY_LSD_F64, lsd_f64.hip)
  bool lsdF64 = false; double* mg = nullptr; unsigned long long* maxMg = nullptr; double* tmp64 = nullptr; double* kern64 = nullptr;
  int* lsdTab64 = nullptr; int lsdRadius = 0;
  bool lsdFront64 = false;                   // the fused blur -> resize -> gradient pass applies (lsd_f64.hip: k_lsd_front64)
  int2* hot = nullptr;                       // tile relaxation, CV_64F detector: round 1's 8-byte hot records {angle, owner word} (lsd_tile.hip)
  float2* cold = nullptr;                    // ... and the records' exact {cos, sin}: written instead of the 16-byte records when every round runs on the hot ones
  int2* own = nullptr; RxSeed* smallSeeds = nullptr; RxSeed* bigSeeds = nullptr; int bigCap = 0;
  RxHand* hand = nullptr; int handCap = 0; RxRect* rects = nullptr; int rectCap = 0; int* rankOf = nullptr; int2* rgBox = nullptr; float4* rgSeg = nullptr; uint8_t* rgClean = nullptr;
  int* rgLost = nullptr;                     // tile relaxation: round in which a region last lost a contested claim (per rank)
  int* tileTouch = nullptr;                  // tile relaxation: round in which a grower last claimed a pixel of the 8x8 cell
  int* tileMin = nullptr; int* tileAct = nullptr; int* rgDirty = nullptr; int tilesW = 0, tilesH = 0; int* rxChunkCnt = nullptr; int rxChunks = 0;
  int* lastSize = nullptr; int* arena = nullptr; int arenaCap = 0; int arenaFactor = 0;    // (arena words per scaled pixel the context got)
  RxCtl* jrCtl = nullptr;
  int2* txList = nullptr; int* txTileCnt = nullptr; int txTs = 64, txNtx = 0, txNty = 0;   // tile-sequential relaxation (lsd_tile.hip)
  bool txKeys = false; int txPixBits = 0; int* txCand = nullptr; int* txCandCnt = nullptr;  // ... its key mode (region id = gradient bin | seed pixel: no ordered list)
  int2* txDirtyList = nullptr; int* txDirtyCnt = nullptr;                                  // per tile: the seeds stamped dirty in a round
  std::vector<RxCtl> jrHost;
  int rxImages = 0;    // images the relaxations' buffers are sized for
  int lsdMode = 0;     // 0 auto, 1 relaxation, 2 sequential, 3 tile-sequential relaxation (cfg.lsd_mode, or PLI_LSD_MODE)
  bool lsdSpec = !(DEVENV("PLI_LSD_SPEC") != nullptr && atoi(DEVENV("PLI_LSD_SPEC")) == 0);  // speculative sequential grower (PLI_LSD_SPEC=0: the plain one)
  int rxLastRounds = 0; // rounds the relaxation needed in an earlier call (the last one whose control blocks the host has seen)
  // Rounds without a host look (the default once rxLastRounds is known): the call launches rxLastRounds + rxMargin rounds — kernels
  // of a settled image leave at once —, k_lsd_grow_unsettled redoes, on the device, whatever image did not settle in them, and the
  // control blocks come back asynchronously (rxSeen / evRxSeen) to be read at the start of a later call.  The call never drains the
  // stream, so copies and kernels of neighbouring calls overlap (pli_batch_submit_host, the multi-GPU gather).
  // the rounds t >= tail t0 of the tile relaxation run in one persistent launch (lsd_tile.hip: k_tx_tail)
  unsigned* tailBar = nullptr;      // [0] barrier arrivals, [32] abort word
  int* txPerm = nullptr;            // round 1: (image, tile) pairs in order of decreasing seed count (k_tx_order)
  int tailBlocks = 0;               // resident grid: CUs x blocks per CU (from the kernel's occupancy)
  int* txCellList = nullptr;        // rounds >= 3 of large batches: the cells with work, listed (k_tx_cells) ...
  int* txCellCnt = nullptr;         // ... and the lists' lengths, two per round (zeroed with the control blocks)
  bool tailLaunched = false;        // this call launched k_tx_tail ...
  bool rxSeenTail = false;          // ... and so did the call whose control blocks are on their way to the host (rxSeen)
  int rxMargin = 3;
  int rxPlanned = 0;                // rounds the last call launched without looking (0: it looked)
  int* rxSeen = nullptr;            // pinned: {state, changed, overflow, rounds} per image of the last look-free call
  hipEvent_t evRxSeen = nullptr;
  int rxSeenImages = 0;             // images behind rxSeen while the copy is in flight (0: nothing pending)
  int64_t rxSlowImages = 0;         // images that took the device-side fallback so far (pli_prof_report prints it)
  unsigned short* chunkHist = nullptr; int* chunkBase = nullptr; int* nDefined = nullptr;
  int* order = nullptr; uint2* regScratch = nullptr; float* seg = nullptr; int* nSeg = nullptr;
  LsdRectItem* rectItems = nullptr;          // sequential grower -> k_lsd_rect (maxSeg per image)
  double* rectW = nullptr;                   // CV_64F pipeline: the weights of the listed pixels, beside the lists
  double* scaled64Dbg = nullptr;             // debug copy of the CV_64F scaled image (its plane is reused as the growers' arena)
  int nChunks = 0, maxSeg = 0;
  int scanGroups = 0, scanChunksPerGroup = 0; int* scanGroupOff = nullptr;     // ordered-list scan of large images, in groups of chunks
  pli_keyline* tmpKL = nullptr;
  short2* dxy = nullptr;           // Sobel (dx, dy) of level 0, interleaved
  LbdCoef* lbdCoef = nullptr;
  // stereo
  int *sad = nullptr, *bestIdx = nullptr;
  unsigned long long* lmask = nullptr; double* ldir = nullptr; short* dmat = nullptr; int *m12 = nullptr, *m21 = nullptr;
  // drop-in state
  uint8_t* inStage[2] = {nullptr, nullptr};
  float2* rectMap[2] = {nullptr, nullptr};   // per eye: (mapx, mapy) of the driver's rectification, or null
  uint8_t* ownTable = nullptr;
  std::vector<uint8_t> hostRec;
  bool orbDone[2] = {false, false}, lineDone[2] = {false, false};
  bool frameFresh = false;                   // ownTable / hostRec hold the COMPLETE record of one Frame (pli_frame_extract): the stereo
                                             // matchers hand it out without running again; any per-call extraction ends that
  bool pointsFresh = false;                  // ... and its mvuRight / mvDepth were computed with the CURRENT rig (pli_set_stereo_camera
                                             // clears it: the point matcher then re-runs on the resident tables)
  uint8_t* recPinned = nullptr;              // pinned staging of that record
  uint8_t* pyrHost[2] = {nullptr, nullptr};  // pinned copy of an eye's pyramid block (pli_orb_pyramid_level)
  bool pyrHostValid[2] = {false, false};     // ... is the one of the last extraction on that eye
  int orbCount[2] = {0, 0};                  // keypoints the last pli_orb_extract(_lapping) left in each eye's table
  int lineCount[2] = {0, 0};                 // keylines the last pli_line_extract left in each eye's table
  int monoCount[2] = {0, 0};                 // pli_orb_extract_lapping: first lapping-area keypoint of each eye's table
  // pipelined host entry point: two slots of device staging (images + table), H2D and D2H copy streams
  struct HostSlot { uint8_t* dimg = nullptr; uint8_t* dtab = nullptr; size_t imgBytes = 0, tabBytes = 0;
                    hipEvent_t h2d = nullptr, kern = nullptr, d2h = nullptr; bool busy = false; };
  HostSlot hs[2];
  hipStream_t sH2D = nullptr, sD2H = nullptr;
  uint64_t hsSubmitted = 0, hsWaited = 0;
  // generic scratch for the stateless matchers
  void* scratch = nullptr; size_t scratchBytes = 0;
  // debug
  bool debug = false;
  float* angDbg = nullptr; float* lbdFloat = nullptr;
  // profiling
  bool prof = false;
  std::vector<ProfEntry> profLog;
  std::vector<hipEvent_t> evPool;
  size_t evNext = 0;

  template <class T> pli_status dalloc(T** p, size_t n) {
    void* q = nullptr;
    if (n == 0) n = 1;
    HIPCHK(hipMalloc(&q, n * sizeof(T)));
    allocs.push_back(q);
    *p = (T*)q;
    return PLI_OK;
  }
  hipEvent_t ev() {
    if (evNext == evPool.size()) {
      hipEvent_t e;
      hipEventCreate(&e);
      evPool.push_back(e);
    }
    return evPool[evNext++];
  }
  void profBegin(const char* name) {
    if (!prof) return;
    ProfEntry pe{name, ev(), ev()};
    hipEventRecord(pe.a, stream);
    profLog.push_back(pe);
  }
  void profEnd() {
    if (!prof) return;
    hipEventRecord(profLog.back().b, stream);
  }
};

// roctx ranges ("rocprofv3 counters + roctx ranges per kernel").  With PLI_ROCTX=1 in the environment every stage of a call
// (ingest, ORB chain, line chain, the two stereo matchers, the entry points themselves) and every kernel launch is bracketed by
// roctxRangePush / roctxRangePop, so that `rocprofv3 --kernel-trace --marker-trace` shows which call and stage a kernel belongs
// to.  The marker library is looked up at run time (rocprofiler-sdk's, then roctracer's): no link-time dependency, nothing is
// loaded and nothing is pushed without the switch.  pli_trace_ranges() counts the ranges pushed so far.
struct PliRoctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  std::atomic<long long> pushed{0};
  PliRoctx() {
    const char* e = getenv("PLI_ROCTX");
    if (!e || atoi(e) == 0) return;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) break;
      push = nullptr; pop = nullptr;
    }
  }
};
static PliRoctx g_roctx;
struct TraceRange {
  bool on;
  explicit TraceRange(const char* name) : on(g_roctx.push != nullptr) {
    if (on) { g_roctx.push(name); g_roctx.pushed.fetch_add(1, std::memory_order_relaxed); }
  }
  ~TraceRange() { if (on) g_roctx.pop(); }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

constexpr int TX_CELL_ROUNDS = 100;   // rounds of a call that can run on cell lists (two list lengths per round and image, zeroed per call)
// the last k_tx_tail launch on every device of this process (see the launch site)
static std::mutex g_tailMu;
constexpr int TAIL_MAX_DEVICES = 64;
static hipEvent_t g_tailEv[TAIL_MAX_DEVICES] = {};
// ... and whether a tail of this process has ever been ABORTED on a device (its grid barrier timed out: somebody else — another
// process — holds compute units the resident grid needs).  From then on this process runs the planned-rounds schedule on that device
// instead of stalling a second per call (include/pli_frontend.h "Sharing a device").
static std::atomic<bool> g_tailAborted[TAIL_MAX_DEVICES] = {};       // (read without the mutex by every call: relaxed atomics)
static bool g_tailWarned = false;

struct CtxGuard {
  const pli_ctx* c;
  explicit CtxGuard(const pli_ctx* c_) : c(c_) { if (c) c->mu.lock(); }
  ~CtxGuard() { if (c) c->mu.unlock(); }
  CtxGuard(const CtxGuard&) = delete;
  CtxGuard& operator=(const CtxGuard&) = delete;
};

#define LAUNCH(c, name, kern, grid, block, shmem, ...)                       \
  do {                                                                       \
    TraceRange tr__(name);                                                   \
    (c)->profBegin(name);                                                    \
    hipLaunchKernelGGL(kern, grid, block, shmem, (c)->stream, __VA_ARGS__);  \
    (c)->profEnd();                                                          \
    HIPCHK(hipGetLastError());                                               \
    if ((c)->syncDebug) {   /* PLI_SYNC_DEBUG: find the kernel that faults */ \
      std::fprintf(stderr, "[pli] %s ...", name);                            \
      HIPCHK(hipStreamSynchronize((c)->stream));                             \
      std::fprintf(stderr, " ok\n");                                         \
    }                                                                        \
  } while (0)

namespace {

// cv::resize coefficient tables (see oracle/ocv_prims.hpp for the derivation):
// xofs[dw] | alpha[2*dw] | yofs[dh] | beta[2*dh], all int32.
std::vector<int> buildResizeTab(int sw, int sh, int dw, int dh, double scale_x, double scale_y) {
  std::vector<int> t((size_t)3 * dw + 3 * dh);
  for (int d = 0; d < dw; ++d) {
    float f = (float)((d + 0.5) * scale_x - 0.5);
    int s = cvFloorf(f);
    f -= s;
    if (s < 0) { f = 0; s = 0; }
    if (s >= sw - 1) { f = 0; s = sw - 1; }
    t[d] = s;
    t[dw + 2 * d] = (short)cvRoundf((1.f - f) * 2048.f);
    t[dw + 2 * d + 1] = (short)cvRoundf(f * 2048.f);
  }
  for (int d = 0; d < dh; ++d) {
    float f = (float)((d + 0.5) * scale_y - 0.5);
    int s = cvFloorf(f);
    f -= s;
    t[3 * dw + d] = s;
    t[3 * dw + dh + 2 * d] = (short)cvRoundf((1.f - f) * 2048.f);
    t[3 * dw + dh + 2 * d + 1] = (short)cvRoundf(f * 2048.f);
  }
  return t;
}

void gaussKernelFixed8(int n, double sigma, int* k) {
  std::vector<float> cf(n);
  double scale2X = -0.5 / (sigma * sigma), sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    cf[i] = (float)std::exp(scale2X * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) {
    cf[i] = (float)(cf[i] * sum);
    k[i] = cvRoundd((double)cf[i] * 256.0);
  }
}

pli_status validate(const pli_frontend_config& c) {
  if (c.width < 64 || c.height < 64 || c.width > 4095 || c.height > 4095) { g_err = "width/height must be in [64,4095]"; return PLI_ERR_INVALID; }
  if (c.max_frames < 1) { g_err = "max_frames < 1"; return PLI_ERR_INVALID; }
  if (c.orb_nlevels < 1 || c.orb_nlevels > MAX_LEVELS) { g_err = "orb_nlevels out of range"; return PLI_ERR_INVALID; }
  if (c.orb_nfeatures < 1 || !(c.orb_scale_factor > 1.f)) { g_err = "orb_nfeatures/scale_factor invalid"; return PLI_ERR_INVALID; }
  if (c.orb_min_th_fast < 1 || c.orb_ini_th_fast < c.orb_min_th_fast || c.orb_ini_th_fast > 254) { g_err = "FAST thresholds invalid"; return PLI_ERR_INVALID; }
  if (c.lsd_refine != 0) { g_err = "only lsd_refine = 0 (LSD_REFINE_NONE) is on the reference path"; return PLI_ERR_INVALID; }
  if (c.lsd_n_bins < 1 || c.lsd_n_bins > 1024) { g_err = "lsd_n_bins must be in [1,1024]"; return PLI_ERR_INVALID; }
  if (c.max_lines < 1 || c.max_lines > (1 << 20)) { g_err = "max_lines must be in [1,2^20]"; return PLI_ERR_INVALID; }
  if (c.parity_flags < 0 || c.parity_flags > 15) { g_err = "parity_flags: unknown bits"; return PLI_ERR_INVALID; }
  if (c.lsd_mode < 0 || c.lsd_mode > 3) { g_err = "lsd_mode must be 0, 1, 2 or 3"; return PLI_ERR_INVALID; }
  if (c.lsd_nfeatures < 0 || c.lsd_nfeatures > c.max_lines) { g_err = "lsd_nfeatures must be in [0,max_lines]"; return PLI_ERR_INVALID; }
  if (!(c.lsd_scale > 0) || !(c.lsd_ang_th > 0 && c.lsd_ang_th < 180)) { g_err = "lsd_scale/ang_th invalid"; return PLI_ERR_INVALID; }
  return PLI_OK;
}

pli_status buildGeometry(pli_ctx* c) {
  const pli_frontend_config& cfg = c->cfg;
  DevParams& P = c->hp;
  std::memset(&P, 0, sizeof(P));
  P.W = cfg.width; P.H = cfg.height;
  P.nlevels = cfg.orb_nlevels;
  P.iniTh = cfg.orb_ini_th_fast; P.minTh = cfg.orb_min_th_fast;
  const int L = P.nlevels;
  // scale tables and per-level quotas, ORBextractor.cc:413-444
  std::vector<float> sc(L), inv(L);
  sc[0] = 1.0f;
  for (int i = 1; i < L; i++) sc[i] = sc[i - 1] * cfg.orb_scale_factor;
  for (int i = 0; i < L; i++) inv[i] = 1.0f / sc[i];
  std::vector<int> quota(L);
  {
    float factor = 1.0f / cfg.orb_scale_factor;
    float nDesired = cfg.orb_nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
      quota[l] = cvRoundf(nDesired);
      sum += quota[l];
      nDesired *= factor;
    }
    quota[L - 1] = std::max(cfg.orb_nfeatures - sum, 0);
  }
  int64_t off = 0;
  int cellBase = 0, kpBase = 0, candBase = 0;
  for (int l = 0; l < L; ++l) {
    LevelGeom& G = P.lv[l];
    G.scale = sc[l]; G.invScale = inv[l];
    G.w = cvRoundf((float)P.W * inv[l]);
    G.h = cvRoundf((float)P.H * inv[l]);
    G.pitch = (int)alignUp(G.w, 64);
    G.offset = off;
    off += (int64_t)G.pitch * G.h;
    off = alignUp(off, 256);
    G.minBX = 19 - 3; G.minBY = 19 - 3;
    G.maxBX = G.w - 19 + 3; G.maxBY = G.h - 19 + 3;
    const float width = (float)(G.maxBX - G.minBX), height = (float)(G.maxBY - G.minBY);
    G.nCols = width > 0 ? (int)(width / 30.f) : 0;
    G.nRows = height > 0 ? (int)(height / 30.f) : 0;
    if (G.nCols <= 0 || G.nRows <= 0) { G.nCols = 0; G.nRows = 0; G.wCell = 0; G.hCell = 0; }
    else {
      G.wCell = (int)std::ceil(width / G.nCols);
      G.hCell = (int)std::ceil(height / G.nRows);
    }
    G.cellBase = cellBase;
    cellBase += G.nCols * G.nRows;
    G.nfeatures = quota[l];
    if (G.nfeatures + 8 > 1000) { g_err = "per-level feature quota above 992 is not supported by k_octree"; return PLI_ERR_INVALID; }
    G.nIni = (G.maxBY - G.minBY) > 0 ? (int)std::round(static_cast<float>(G.maxBX - G.minBX) / (G.maxBY - G.minBY)) : 0;
    if (G.nCols == 0) G.nIni = 0;
    G.hX = G.nIni > 0 ? static_cast<float>(G.maxBX - G.minBX) / G.nIni : 1.f;
    G.kpBase = kpBase;
    G.kpCap = std::max(G.nfeatures + 4, 4 * G.nIni + 4);
    kpBase += G.kpCap;
    const int perCell = std::min(CELL_CAP, ((G.wCell + 1) / 2) * ((G.hCell + 1) / 2));
    G.candBase = candBase;
    G.candCap = G.nCols * G.nRows * perCell;
    candBase += (int)alignUp(G.candCap, 4);
  }
  P.pyrBlock = alignUp(off, 256);
  P.cellsPerImage = cellBase;
  P.kpSlotsPerImage = kpBase;
  P.candPerImage = candBase;
  P.kpCap = pli_kp_capacity(&cfg);
  P.klCap = pli_kl_capacity(&cfg);
  // umax, ORBextractor.cc:451-467
  {
    const int HP = 15;
    int umax[HP + 2] = {0};
    int v, v0, vmax = cvFloorf(HP * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(HP * std::sqrt(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRoundd(std::sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
    for (int i = 0; i < 16; ++i) P.umax[i] = umax[i];
  }
  // LSD (OpenCV lsd.cpp flsd)
  P.lsdScale = cfg.lsd_scale;
  P.LWt = cfg.lsd_scale != 1 ? cvRoundd(P.W * cfg.lsd_scale) : P.W;
  P.LH = cfg.lsd_scale != 1 ? cvRoundd(P.H * cfg.lsd_scale) : P.H;
  // LW is the ROW PITCH, in pixels, of every per-pixel plane of the detector (records, norms, owner pairs, region tables' pixel
  // indices): the true width LWt rounded up to 16 pixels, so that a row of 8-byte records starts on a 128-byte line and a row of 4-byte
  // words on a 64-byte sector (902 -> 912 at 752 x 480 x 1.2: k_tx_sort's id stores and the row-wise passes touch whole sectors).  The
  // pad columns are undefined pixels — as the image's last column already is — written by the front pass; nothing after it knows the
  // true width.  Pixel indices (seed order within a bin, region ids in key mode) keep their order: (y, x) lexicographic either way.
  P.LW = (int)alignUp(P.LWt, 16);
  if (DEVENV("PLI_LSD_NOPAD")) P.LW = P.LWt;                 // dev switch (A/B)
  P.lpitch = (int)alignUp(P.LWt, 64);
  P.prec = 3.14159265358979323846 * cfg.lsd_ang_th / 180;
  {
    // Vector form of isAligned (lsd.cpp region_grow): the angle between the region's (sumdx, sumdy) and a pixel's
    // (cos, sin) is within prec  <=>  dot > 0 and dot^2 >= cos^2(prec) |sum|^2.  The sequential grower uses it with a
    // margin on both sides (fastAtan2 is within 0.01 deg of atan2) and decides only the pixels inside the margin with
    // the reference's own expression.
    double margin = 0.05 * 3.14159265358979323846 / 180;
    // (test switch PLI_ALIGN_MARGIN_DEG: a wide margin sends many more candidates through the exact expression — and, on the hot records,
    // through the exact sums; the results must not change)
    if (const char* e = DEVENV("PLI_ALIGN_MARGIN_DEG")) margin = std::max(0.05, std::min(20.0, atof(e))) * 3.14159265358979323846 / 180;
    P.hotBand2 = 0.0;
    if (const char* e = DEVENV("PLI_TX_HOT_BAND2")) P.hotBand2 = std::max(0.0, atof(e));              // test switch: 10 = the exact sums at every event
    P.rectApproxBand = 2e-3;
    if (const char* e = DEVENV("PLI_RECT_APPROX_BAND")) P.rectApproxBand = std::max(2e-3, atof(e));     // test switch: 10 = every region
    P.alignFilter = (P.prec + margin < 1.5) && (P.prec - margin > 0.01);
    const double cl = std::cos(P.prec + margin), ch = std::cos(P.prec - margin);
    // (without the filter: every candidate is a "maybe", none is "sure")
    P.alignLo = P.alignFilter ? (float)(cl * cl * (1 - 1e-5)) : -INFINITY;
    P.alignHi = P.alignFilter ? (float)(ch * ch * (1 + 1e-5)) : INFINITY;
    P.alignPad = 0;                                       // dev switches of the speculative grower
    if (const char* e = DEVENV("PLI_LSD_SPEC")) P.alignPad = atoi(e) == 3 ? 1 : atoi(e) == 4 ? 3 : 0;
    if (const char* e = DEVENV("PLI_LSD_SPEC_CAP")) P.alignPad |= std::max(0, std::min(8, atoi(e))) << 4;
  }
  {
    const double rho = cfg.lsd_quant / std::sin(P.prec);
    int g = 0;
    while (std::sqrt(g / 4.0) <= rho) ++g;     // defined  <=>  sqrt(g2/4) > rho
    P.g2Thresh = g - 1;
  }
  P.nBins = cfg.lsd_n_bins;
  P.parityFlags = cfg.parity_flags; P.parityPad = 0;
  P.rho = cfg.lsd_quant / std::sin(P.prec);
  {
    const double p = cfg.lsd_ang_th / 180;
    const double LOG_NT = 5 * (std::log10(double(P.LWt)) + std::log10(double(P.LH))) / 2 + std::log10(11.0);
    P.minRegSize = (int)size_t(-LOG_NT / std::log10(p));
  }
  P.maxLines = cfg.max_lines;
  P.lsdNFeatures = cfg.lsd_nfeatures;
  P.minLength = cfg.min_line_length * (std::min(P.W, P.H));
  P.bf = cfg.bf;
  P.maxD = cfg.stereo_maxd_inf ? std::numeric_limits<float>::infinity() : cfg.bf / (cfg.bf / cfg.fx);
  P.sWs = cfg.matching_s_ws; P.bestLR = cfg.best_lr_matches;
  P.lineSimTh = cfg.line_sim_th; P.overlapTh = cfg.stereo_overlap_th; P.ratio12L = cfg.min_ratio_12_l;
  P.minDispRatio = cfg.ls_min_disp_ratio; P.minDisp = cfg.min_disp; P.horizTh = cfg.line_horiz_th;
  return PLI_OK;
}

void buildLayout(pli_ctx* c) {
  pli_table_layout& L = c->lay;
  const int kc = c->hp.kpCap, lc = c->hp.klCap;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { int64_t r = o; o = alignUp(o + bytes, 16); return r; };
  L.kp_cap = kc; L.kl_cap = lc;
  L.off_counts = take(8 * 4);
  L.off_kp[0] = take((int64_t)kc * sizeof(pli_keypoint));
  L.off_kp[1] = take((int64_t)kc * sizeof(pli_keypoint));
  L.off_desc[0] = take((int64_t)kc * 32);
  L.off_desc[1] = take((int64_t)kc * 32);
  L.off_uright = take((int64_t)kc * 4);
  L.off_depth = take((int64_t)kc * 4);
  L.off_kl[0] = take((int64_t)lc * sizeof(pli_keyline));
  L.off_kl[1] = take((int64_t)lc * sizeof(pli_keyline));
  L.off_ldesc[0] = take((int64_t)lc * 32);
  L.off_ldesc[1] = take((int64_t)lc * 32);
  L.off_disp = take((int64_t)lc * 8);
  L.off_le = take((int64_t)lc * 24);
  L.record_bytes = alignUp(o, 256);
}

pli_status allocAll(pli_ctx* c) {
  const DevParams& P = c->hp;
  const int NI = c->NI;
  pli_status st;
#define A(ptr, n) if ((st = c->dalloc(&(ptr), (size_t)(n))) != PLI_OK) return st
  A(c->dP, 1);
  HIPCHK(hipMemcpy(c->dP, &c->hp, sizeof(DevParams), hipMemcpyHostToDevice));
  A(c->pyr, (size_t)P.pyrBlock * NI);
  A(c->blur, (size_t)P.pyrBlock * NI);
  for (int l = 1; l < P.nlevels; ++l) {
    const LevelGeom &S = P.lv[l - 1], &D = P.lv[l];
    std::vector<int> t = buildResizeTab(S.w, S.h, D.w, D.h, 1. / ((double)D.w / S.w), 1. / ((double)D.h / S.h));
    A(c->resizeTab[l], t.size());
    HIPCHK(hipMemcpy(c->resizeTab[l], t.data(), t.size() * 4, hipMemcpyHostToDevice));
  }
  A(c->cellCand, (size_t)NI * P.cellsPerImage * CELL_CAP);
  A(c->cellCount, (size_t)NI * P.cellsPerImage);
  A(c->candAll, (size_t)NI * P.candPerImage);
  A(c->nodeOf, (size_t)NI * P.candPerImage);
  A(c->candCount, (size_t)NI * P.nlevels);
  A(c->kpSel, (size_t)NI * P.kpSlotsPerImage);
  A(c->kpSelCount, (size_t)NI * P.nlevels);
  // blur jobs
  {
    BlurJob J;
    std::memset(&J, 0, sizeof(J));
    J.nplanes = P.nlevels;
    J.radius = 3;
    gaussKernelFixed8(7, 2.0, J.k);
    int tb = 0;
    for (int l = 0; l < P.nlevels; ++l) {
      BlurPlane& p = J.pl[l];
      p.w = P.lv[l].w; p.h = P.lv[l].h; p.pitchIn = p.pitchOut = P.lv[l].pitch;
      p.offIn = p.offOut = P.lv[l].offset;
      p.tilesX = (p.w + 63) / 64;
      p.tileBase = tb;
      tb += p.tilesX * ((p.h + 31) / 32);
    }
    c->orbTiles = tb;
    A(c->jobOrb, 1);
    HIPCHK(hipMemcpy(c->jobOrb, &J, sizeof(J), hipMemcpyHostToDevice));
    // LSD pre-filter: level 0 -> tmp8, kernel size from sigma (lsd.cpp)
    c->tmpPitch = (int)alignUp(P.W, 64);
    c->tmp8Stride = alignUp((int64_t)c->tmpPitch * P.H, 256);
    std::memset(&J, 0, sizeof(J));
    J.nplanes = 1;
    const double sigma = (c->cfg.lsd_scale < 1) ? (c->cfg.lsd_sigma_scale / c->cfg.lsd_scale) : c->cfg.lsd_sigma_scale;
    const unsigned h = (unsigned)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0)));
    if (h > 3) { g_err = "LSD pre-filter radius > 3 not supported"; return PLI_ERR_INVALID; }
    J.radius = (int)h;
    gaussKernelFixed8(1 + 2 * (int)h, sigma, J.k);
    J.pl[0].w = P.W; J.pl[0].h = P.H; J.pl[0].pitchIn = P.lv[0].pitch; J.pl[0].pitchOut = c->tmpPitch;
    J.pl[0].offIn = P.lv[0].offset; J.pl[0].offOut = 0;
    J.pl[0].tilesX = (P.W + 63) / 64; J.pl[0].tileBase = 0;
    c->l0Tiles = J.pl[0].tilesX * ((P.H + 31) / 32);
    A(c->jobLsd, 1);
    HIPCHK(hipMemcpy(c->jobLsd, &J, sizeof(J), hipMemcpyHostToDevice));
    // LBD: 5x5 sigma 1 (binary_descriptor_custom.cpp:358)
    J.radius = 2;
    std::memset(J.k, 0, sizeof(J.k));
    gaussKernelFixed8(5, 1.0, J.k);
    A(c->jobLbd, 1);
    HIPCHK(hipMemcpy(c->jobLbd, &J, sizeof(J), hipMemcpyHostToDevice));
  }
  A(c->tmp8, (size_t)c->tmp8Stride * NI);
  c->lsdStride = alignUp((int64_t)P.lpitch * P.LH, 256);
  A(c->lsdScaled, (size_t)c->lsdStride * NI);
  if (c->cfg.lsd_scale != 1) {
    std::vector<int> t = buildResizeTab(P.W, P.H, P.LWt, P.LH, 1. / c->cfg.lsd_scale, 1. / c->cfg.lsd_scale);
    A(c->lsdTab, t.size());
    HIPCHK(hipMemcpy(c->lsdTab, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  }
  const size_t npix = (size_t)P.LW * P.LH;
  A(c->rec, npix * NI);
  c->lsdF64 = (c->cfg.parity_flags & PLI_PARITY_LSD_F64) != 0;
  if (c->lsdF64) {
    A(c->mg, npix * NI);
    A(c->maxMg, NI);
    // Gaussian kernel in double (cv::getGaussianKernel(n, sigma, CV_64F)); radius 0 / kernel {1} converts u8 -> double (lsd_scale 1)
    const double sigma = (c->cfg.lsd_scale < 1) ? (c->cfg.lsd_sigma_scale / c->cfg.lsd_scale) : c->cfg.lsd_sigma_scale;
    const unsigned h = c->cfg.lsd_scale != 1 ? (unsigned)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0))) : 0u;
    if (h > 3) { g_err = "LSD pre-filter radius > 3 not supported"; return PLI_ERR_INVALID; }
    c->lsdRadius = (int)h;
    double k[7] = {0, 0, 0, 0, 0, 0, 0};
    const int n = 1 + 2 * (int)h;
    if (h == 0) k[0] = 1.0;
    else {
      const double scale2X = -0.5 / (sigma * sigma);
      double sum = 0;
      for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = std::exp(scale2X * x * x); sum += k[i]; }
      sum = 1. / sum;
      for (int i = 0; i < n; ++i) k[i] *= sum;
    }
    A(c->kern64, 7);
    HIPCHK(hipMemcpy(c->kern64, k, sizeof(k), hipMemcpyHostToDevice));
    if (c->cfg.lsd_scale != 1) {
      A(c->tmp64, (size_t)P.W * P.H * NI);
      // cv::resize coefficient tables for CV_64F (imgwarp.cpp): float weights (1 - f, f), x clamped at the borders, y not
      std::vector<int> t((size_t)3 * P.LWt + 3 * P.LH);
      const double sc = 1. / c->cfg.lsd_scale;
      auto fbits = [](float f) { int b; std::memcpy(&b, &f, 4); return b; };
      for (int d = 0; d < P.LWt; ++d) {
        float f = (float)((d + 0.5) * sc - 0.5);
        int sidx = cvFloorf(f);
        f -= sidx;
        if (sidx < 0) { f = 0; sidx = 0; }
        if (sidx >= P.W - 1) { f = 0; sidx = P.W - 1; }
        t[d] = sidx; t[P.LWt + 2 * d] = fbits(1.f - f); t[P.LWt + 2 * d + 1] = fbits(f);
      }
      for (int d = 0; d < P.LH; ++d) {
        float f = (float)((d + 0.5) * sc - 0.5);
        int sidx = cvFloorf(f);
        f -= sidx;
        t[3 * P.LWt + d] = sidx; t[3 * P.LWt + P.LH + 2 * d] = fbits(1.f - f); t[3 * P.LWt + P.LH + 2 * d + 1] = fbits(f);
      }
      A(c->lsdTab64, t.size());
      HIPCHK(hipMemcpy(c->lsdTab64, t.data(), t.size() * 4, hipMemcpyHostToDevice));
      // the fused front (k_lsd_front64) stages the source window of a 64 x 16 tile of the scaled image in LDS: 64 x 20 at most
      bool fits = (int)h <= 3;
      for (int x0 = 0; x0 < P.LWt && fits; x0 += 64) {
        const int x1 = std::min(x0 + 64, P.LWt - 1);
        fits = std::min(t[x1] + 1, P.W - 1) - t[x0] + 1 <= 64;
      }
      for (int y0 = 0; y0 < P.LH && fits; y0 += 16) {
        const int y1 = std::min(y0 + 16, P.LH - 1);
        const int a = std::min(std::max(t[3 * P.LWt + y0], 0), P.H - 1), b = std::min(std::max(t[3 * P.LWt + y1] + 1, 0), P.H - 1);
        fits = b - a + 1 <= 20;
      }
      c->lsdFront64 = fits && !DEVENV("PLI_LSD_NOFUSE");
    }
  } else {
    A(c->g2, npix * NI);
  }
  c->lsdMode = c->cfg.lsd_mode;
  if (const char* e = getenv("PLI_LSD_MODE")) {
    const int m = atoi(e);
    if (m < 0 || m > 3) { g_err = "PLI_LSD_MODE must be 0, 1, 2 or 3"; return PLI_ERR_INVALID; }
    c->lsdMode = m;
  }
#ifndef PLI_DEV
  if (c->lsdMode == 1) { g_err = "lsd_mode 1 (the lane relaxation, round 1's schedule) is kept for cross-checks in the development build only (libpli_frontend_dev.so)"; return PLI_ERR_INVALID; }
#endif
  if (c->lsdMode != 2) {     // buffers of the relaxations (in auto mode only batches below RX_AUTO_IMAGES use them)
    // (auto mode: a context sized for the sequential regime keeps the relaxations' buffers — 20-30 MB per image — for 2047 images,
    // as in round 2, and batches above that take the sequential grower: 2048 frames stay at ~160 GB of HBM instead of 289)
    const size_t NR = c->lsdMode == 0 ? ((int)NI >= RX_AUTO_IMAGES ? std::min<size_t>(NI, 2047) : NI) : NI;
    c->rxImages = (int)NR;
    const bool lane = c->lsdMode == 1, tiles = c->lsdMode == 0 || c->lsdMode == 3;
    A(c->own, npix * NR);
    if (tiles && c->lsdF64) { A(c->hot, npix * NR); A(c->cold, npix * NR); }
    if (lane) {               // lane / lane-group growers of lsd_relax.hip
      A(c->smallSeeds, npix * NR);
      c->bigCap = (int)(npix / RX_HAND + 64);               // a region listed as large had >= RX_HAND pixels of its own
      A(c->bigSeeds, (size_t)c->bigCap * NR);
      c->handCap = (int)(npix / RX_HAND + 64);
      A(c->hand, (size_t)c->handCap * NR);
      A(c->rgClean, npix * NR);
    }
    c->rectCap = 2 * ((int)(npix / std::max(P.minRegSize, 1)) + 64);
    A(c->rects, (size_t)c->rectCap * NR);
    A(c->lastSize, npix * NR);                            // region tables, indexed by seed rank
    A(c->rgBox, npix * NR);
    A(c->rgSeg, npix * NR);
    A(c->rankOf, npix * NR);
    c->tilesW = (P.LW + 7) / 8; c->tilesH = (P.LH + 7) / 8;
    A(c->tileMin, (size_t)c->tilesW * c->tilesH * NR);
    A(c->tileAct, (size_t)c->tilesW * c->tilesH * NR);
    A(c->tileTouch, (size_t)c->tilesW * c->tilesH * NR);
    A(c->rgDirty, npix * NR);
    A(c->rgLost, npix * NR);
    c->rxChunks = (int)((npix + 2047) / 2048);
    A(c->rxChunkCnt, (size_t)c->rxChunks * NR);
    // pixel lists for region2rect (<= npix per round) + queue overflow blocks; the lane growers also park hand-overs here
    // Words of arena per scaled pixel.  A round needs one word per pixel for the finished regions' lists — and, in the tile relaxation,
    // queue overflow blocks for every region of more than TX_GQ pixels that is being grown, by EVERY tile it has seeds in at the
    // same time in round 1: long parallel structures (blinds, corrugated walls; the "stripes" image of the tests) multiply that
    // by the tiles a region crosses.  3 words per pixel sent all 512 stripes images to the sequential grower (455 ms per batch
    // instead of 50); they need < 16 (8 is not enough: round 3's last commit budgeted 8 GiB, which gave the 256-frame context 8 words,
    // and the stripes batch took the fallback again, 507 ms — found by tools/rounds_sweep.py in round 4).  A context gets 16 words per
    // pixel as long as its arenas stay below 18 GiB (256 frames of 752 x 480: 17.0 GB of the 288 GB HBM); larger contexts share that
    // budget (512 frames: 8 words; a 2047-image context of the auto mode: ~10 words, 19 GB).  The budget is also clamped to a
    // quarter of what the device has FREE when the context is created (several large contexts in one process, a part with less HBM),
    // and an allocation that still fails is retried with half the words down to 3 before the context gives up.
    size_t arenaFactor = lane ? 8 : 16;
    if (!lane) {
      size_t budget = (size_t)18 << 30, freeB = 0, totalB = 0;
      if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB > 0) budget = std::min(budget, freeB / 4);
      arenaFactor = std::max<size_t>(3, std::min<size_t>(16, budget / (npix * 4 * NR)));
    }
    if (const char* e = DEVENV("PLI_RX_ARENA")) arenaFactor = (size_t)std::max(1, atoi(e));     // dev: words of arena per scaled pixel
    const size_t arenaAsked = arenaFactor;
    for (;;) {
      c->arenaCap = (int)std::min<size_t>(arenaFactor * npix + 65536, (size_t)1 << 30);
      const pli_status as = c->dalloc(&c->arena, (size_t)c->arenaCap * NR);
      if (as == PLI_OK) break;
      if (lane || arenaFactor <= 3) return as;
      (void)hipGetLastError();                              // (the failed allocation's error state)
      arenaFactor = std::max<size_t>(3, arenaFactor / 2);
    }
    c->arenaFactor = (int)arenaFactor;
    // (results stay exact below 16 words per pixel, but hostile batches — long parallel structures — then take the ~500 ms fallback: say so once)
    if (!lane && arenaFactor < 16 && arenaAsked >= arenaFactor && !DEVENV("PLI_RX_ARENA")) {
      static std::atomic<bool> warned{false};
      if (!warned.exchange(true))
        std::fprintf(stderr, "pli_frontend: the LSD relaxation's arena of this context holds %d words per scaled pixel (16 wanted: %zu images, "
                             "free device memory); results are unchanged, batches of long parallel structures may take the slow fallback "
                             "(pli_lsd_arena_words)\n", (int)arenaFactor, NR);
    }
    if (tiles) {
      // (the measure is the number of 64-pixel tile waves the context can put on the chip, not the number of images)
      c->txTs = (int64_t)NI * ((P.LW + 63) / 64) * ((P.LH + 63) / 64) <= TX_SMALL_TILE_WAVES ? 32 : 64;
      if (const char* e = DEVENV("PLI_TX_TS")) c->txTs = atoi(e) == 16 ? 16 : atoi(e) == 32 ? 32 : atoi(e) == 128 ? 128 : 64;
      c->txNtx = (P.LW + c->txTs - 1) / c->txTs; c->txNty = (P.LH + c->txTs - 1) / c->txTs;
      A(c->txList, (size_t)c->txNtx * c->txNty * c->txTs * c->txTs * NR);
      A(c->txTileCnt, (size_t)c->txNtx * c->txNty * NR);
      A(c->txDirtyList, (size_t)c->txNtx * c->txNty * c->txTs * c->txTs * NR);
      A(c->txDirtyCnt, (size_t)c->txNtx * c->txNty * NR);
      A(c->tailBar, 64);
      A(c->txCellList, (size_t)c->tilesW * c->tilesH * NR);
      A(c->txCellCnt, (size_t)2 * TX_CELL_ROUNDS * NR);
      A(c->txPerm, (size_t)c->txNtx * c->txNty * NR);
      // key mode (lsd_tile.hip, k_tx_sort): ids of (bits of nBins - 1) + pixbits bits must stay below LSD_ID_INF = 2^31 - 1: with 1024 bins
      // up to 2^21 scaled pixels (1280 x 720 scales to 1536 x 864 = 1.33 M; 3840 x 2160 to 11.9 M pixels = 24 bits: ranks).
      // PLI_TX_KEYS=0: ranks.  (debug contexts keep the ordered list for PLI_DBG_LSD_ORDER: checked at run time)
      c->txPixBits = 1;
      while (((size_t)1 << c->txPixBits) < npix) ++c->txPixBits;
      int binBits = 1;
      while ((1 << binBits) < P.nBins) ++binBits;
      // (the largest id is (nBins - 1) << pixbits | npix - 1: strictly below 2^31 - 1 unless every bit is used AND the image fills 2^pixbits)
      c->txKeys = c->lsdF64 && P.nBins <= 1024 && (binBits + c->txPixBits < 31 || (binBits + c->txPixBits == 31 && npix < ((size_t)1 << c->txPixBits))) &&
                  !(DEVENV("PLI_TX_KEYS") && atoi(DEVENV("PLI_TX_KEYS")) == 0);
      if (c->txKeys) {
        A(c->txCand, (size_t)TX_EMIT_CAP * NR);
        A(c->txCandCnt, NR);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tx_emit_sorted), hipFuncAttributeMaxDynamicSharedMemorySize, TX_EMIT_CAP * 4));
      }
    }
  }
  A(c->jrCtl, NI);
  c->jrHost.resize(NI);

  A(c->maxG2, NI);
  c->nChunks = (int)((npix + LSD_CHUNK - 1) / LSD_CHUNK);
  A(c->chunkHist, (size_t)NI * c->nChunks * P.nBins);
  A(c->chunkBase, (size_t)NI * c->nChunks * P.nBins);
  if (c->nChunks > 256 && !DEVENV("PLI_LSD_SCAN1")) {     // (dev switch: the one-block scan)
    c->scanChunksPerGroup = 64;
    c->scanGroups = (c->nChunks + c->scanChunksPerGroup - 1) / c->scanChunksPerGroup;
    A(c->scanGroupOff, (size_t)NI * c->scanGroups * P.nBins);
  }
  A(c->nDefined, NI);
  A(c->order, npix * NI);
  A(c->regScratch, npix * NI);
  c->maxSeg = (int)(npix / std::max(P.minRegSize, 1)) + 64;   // a region needs minRegSize pixels: no image can yield more segments
  A(c->seg, (size_t)NI * c->maxSeg * 4);
  A(c->rectItems, (size_t)NI * c->maxSeg);
  if (c->lsdF64) A(c->rectW, npix * NI);
  A(c->nSeg, NI);
  A(c->tmpKL, (size_t)NI * P.maxLines);
  A(c->dxy, (size_t)P.W * P.H * NI);
  {
    // BinaryDescriptor ctor weights, binary_descriptor_custom.cpp:217-259 (integer divisions kept)
    LbdCoef C;
    const int widthOfBand = 7, NUM_OF_BANDS = 9;
    double u = (widthOfBand * 3 - 1) / 2;
    double sigma = (widthOfBand * 2 + 1) / 2;
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < widthOfBand * 3; i++) { double dis = i - u; C.L[i] = (float)std::exp(dis * dis * invsigma2); }
    u = (NUM_OF_BANDS * widthOfBand - 1) / 2;
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < NUM_OF_BANDS * widthOfBand; i++) { double dis = i - u; C.G[i] = (float)std::exp(dis * dis * invsigma2); }
    A(c->lbdCoef, 1);
    HIPCHK(hipMemcpy(c->lbdCoef, &C, sizeof(C), hipMemcpyHostToDevice));
  }
  const int NF = c->cfg.max_frames;
  A(c->sad, (size_t)NF * P.kpCap);
  A(c->bestIdx, (size_t)NF * P.kpCap);
  A(c->lmask, (size_t)NF * P.klCap * GRID_ROWS);
  A(c->ldir, (size_t)NF * P.klCap * 2);
  A(c->dmat, (size_t)NF * P.klCap * P.klCap);
  A(c->m12, (size_t)NF * P.klCap);
  A(c->m21, (size_t)NF * P.klCap);
  A(c->inStage[0], (size_t)P.W * P.H);
  A(c->inStage[1], (size_t)P.W * P.H);
  A(c->ownTable, (size_t)c->lay.record_bytes * NF);
  HIPCHK(hipMemset(c->ownTable, 0, (size_t)c->lay.record_bytes * NF));
  c->hostRec.resize((size_t)c->lay.record_bytes);
#undef A
  return PLI_OK;
}

// ---- stage scheduling -------------------------------------------------------
// Folds the control blocks a look-free call sent back (rxSeen) into the plan of the next call.  Returns the rounds the slowest
// settled image needed, or -1 when an image took the device-side fallback.
int foldRoundStats(pli_ctx* c) {
  int settled = 0, unsettled = 0, overflowed = 0;
  for (int i = 0; i < c->rxSeenImages; ++i) {
    const int* h = c->rxSeen + 4 * i;
    if (h[2]) ++overflowed;                            // a capacity ran out (adversarial images): more rounds would not help
    else if (h[0] != 2) ++unsettled;
    else settled = std::max(settled, h[3]);
  }
  c->rxSlowImages += unsettled + overflowed;
  // An image that has not settled although the self-terminating tail kernel ran (it walks up to 96 rounds): the tail was aborted —
  // its grid barrier timed out because the resident grid did not fit beside somebody else's work (another PROCESS on the device).
  // This process runs the planned-rounds schedule on that device from now on instead of stalling ~1 s in every call.
  // (the abort word of the tail's grid barrier came back with the control blocks: an image that merely needed more than the 96 rounds the
  // tail walks — or a small PLI_RX_MAXROUNDS — is not an aborted tail and does not switch the process's schedule)
  const bool aborted = c->rxSeen[4 * (size_t)c->NI] != 0;
  if (unsettled && c->rxSeenTail && aborted && c->device >= 0 && c->device < TAIL_MAX_DEVICES) {
    std::lock_guard<std::mutex> lk(g_tailMu);
    g_tailAborted[c->device].store(true, std::memory_order_relaxed);
    if (!g_tailWarned) {
      g_tailWarned = true;
      std::fprintf(stderr, "pli_frontend: the persistent relaxation kernel did not finish on device %d (another process on the device?); "
                           "this process uses the planned-rounds schedule there from now on (as with PLI_TX_TAIL=0)\n", c->device);
    }
  }
  if (unsettled) c->rxLastRounds = std::min(96, c->rxLastRounds + c->rxMargin);        // not enough rounds: plan more next time
  else if (settled > 0) c->rxLastRounds = std::max(settled, c->rxLastRounds - 1);       // (follows a calmer stream down slowly)
  c->rxSeenImages = 0;
  return (unsettled || overflowed) ? -1 : settled;
}

pli_status runIngest(pli_ctx* c, const uint8_t* dl, const uint8_t* dr, int64_t stride, int64_t frameStride, int img0, int nimg) {
  TraceRange range__("pli:ingest");
  const DevParams& P = c->hp;
  const bool aligned16 = !c->rectMap[0] && !c->rectMap[1] && (P.W % 16) == 0 && (stride % 16) == 0 && (frameStride % 16) == 0 &&
                         (reinterpret_cast<uintptr_t>(dl) % 16) == 0 && (reinterpret_cast<uintptr_t>(dr) % 16) == 0 && (P.lv[0].pitch % 16) == 0 &&
                         (P.pyrBlock % 16) == 0 && !DEVENV("PLI_INGEST_BYTES");
  if (aligned16) {
    const int n16 = (P.W / 16) * P.H;
    LAUNCH(c, "k_ingest", k_ingest_copy16, dim3(std::max(1, std::min((n16 + 255) / 256, 64)), nimg), dim3(256), 0, dl, dr, stride, frameStride,
           c->pyr, P.pyrBlock, P.W / 16, P.H, P.lv[0].pitch, img0);
    return PLI_OK;
  }
  dim3 g((P.W + 1023) / 1024, P.H, nimg);
  LAUNCH(c, "k_ingest", k_ingest, g, dim3(256), 0, dl, dr, stride, frameStride, c->pyr, P.pyrBlock, P.W, P.H, P.lv[0].pitch,
         (const float2*)c->rectMap[0], (const float2*)c->rectMap[1], img0);
  return PLI_OK;
}

pli_status runOrb(pli_ctx* c, int img0, int nimg, uint8_t* table) {
  TraceRange range__("pli:orb chain");
  const DevParams& P = c->hp;
  const pli_table_layout& Y = c->lay;
  for (int l = 1; l < P.nlevels; ++l) {
    const LevelGeom &S = P.lv[l - 1], &D = P.lv[l];
    dim3 g((D.w + 255) / 256, (D.h + 15) / 16, nimg);
    LAUNCH(c, "k_resize_level", k_resize_level, g, dim3(256), 0, c->pyr + S.offset, P.pyrBlock, S.w, S.h, S.pitch,
           c->pyr + D.offset, P.pyrBlock, D.w, D.h, D.pitch, c->resizeTab[l], img0);
  }
  if (P.cellsPerImage > 0)
    LAUNCH(c, "k_fast_cells", k_fast_cells, dim3(P.cellsPerImage, nimg), dim3(256), 0, c->dP, c->pyr, c->cellCand, c->cellCount, img0);
  {
    // node capacity of the octree instance: the list ends at quota..quota+3 nodes (first pass: 4 per root)
    int need = 0;
    for (int l = 0; l < P.nlevels; ++l) need = std::max(need, std::max(P.lv[l].nfeatures + 8, 4 * P.lv[l].nIni + 8));
    // workgroups of 1024 threads while the (level, image) workgroups do not fill the chip (4K: 32 images x 8 levels; a single pair:
    // 16 workgroups): the passes over a level's keys are the long pole, 4x the threads on them (dev switch PLI_OCTREE_WIDE=0/1)
    bool wide = P.nlevels * nimg <= 1024;
    if (const char* e = DEVENV("PLI_OCTREE_WIDE")) wide = atoi(e) != 0;
    const dim3 og(P.nlevels, nimg), ob(wide ? 1024 : 256);
#define OCTREE_LAUNCH(K) LAUNCH(c, "k_octree", K, og, ob, 0, c->dP, c->cellCand, c->cellCount, c->candAll, c->nodeOf, c->candCount, c->kpSel, c->kpSelCount, img0)
    if (need <= 320) { if (wide) OCTREE_LAUNCH(k_octree_320_w); else OCTREE_LAUNCH(k_octree_320); }
    else if (need <= 512) { if (wide) OCTREE_LAUNCH(k_octree_512_w); else OCTREE_LAUNCH(k_octree_512); }
    else { if (wide) OCTREE_LAUNCH(k_octree_w); else OCTREE_LAUNCH(k_octree); }
#undef OCTREE_LAUNCH
  }
  LAUNCH(c, "k_blur_orb", k_blur, dim3(c->orbTiles, nimg), dim3(256), 0, c->jobOrb, c->pyr, P.pyrBlock, c->blur, P.pyrBlock, img0);
  LAUNCH(c, "k_kp_counts", k_kp_counts, dim3((nimg + 63) / 64), dim3(64), 0, c->dP, c->kpSelCount, table, Y.record_bytes,
         Y.off_counts, nimg, img0);
  LAUNCH(c, "k_describe", k_describe, dim3(P.kpSlotsPerImage, nimg), dim3(64), 0, c->dP, c->pyr, c->blur, c->kpSel,
         c->kpSelCount, table, Y.record_bytes, Y.off_counts, Y.off_kp[0], Y.off_kp[1], Y.off_desc[0], Y.off_desc[1], img0);
  return PLI_OK;
}

// LAUNCH plus, under PLI_RX_TRACE, a sync and a line on stderr (finds a kernel that does not return)
#define TRL(c, name, ...)                                                      \
  do {                                                                         \
    LAUNCH(c, rxn(name), __VA_ARGS__);                                         \
    if (trace) {                                                               \
      HIPCHK(hipStreamSynchronize((c)->stream));                               \
      std::fprintf(stderr, "[rx] %s done\n", name);                           \
    }                                                                          \
  } while (0)

// BinaryDescriptor::compute's image work (binary_descriptor_custom.cpp:350-398): GaussianBlur 5x5 + Sobel of level 0.  It
// needs the image only, not the lines: pli_batch_run puts it on the side stream (after the ORB chain) when the LSD front does
// not use the u8 scratch plane itself (CV_64F detector).
pli_status runLbdPre(pli_ctx* c, int img0, int nimg) {
  TraceRange range__("pli:lbd blur+sobel");
  const DevParams& P = c->hp;
  LAUNCH(c, "k_blur_lbd", k_blur, dim3(c->l0Tiles, nimg), dim3(256), 0, c->jobLbd, c->pyr, P.pyrBlock, c->tmp8, c->tmp8Stride, img0);
  dim3 g((P.W + 1023) / 1024, P.H, nimg);
  LAUNCH(c, "k_sobel", k_sobel, g, dim3(256), 0, c->tmp8, c->tmp8Stride, P.W, P.H, c->tmpPitch, c->dxy, img0);
  return PLI_OK;
}

pli_status runLines(pli_ctx* c, int img0, int nimg, uint8_t* table) {
  TraceRange range__("pli:line chain");
  const DevParams& P = c->hp;
  const pli_table_layout& Y = c->lay;
  const int npix = P.LW * P.LH;
  const bool sequential = c->lsdMode == 2 || (c->lsdMode == 0 && (nimg >= RX_AUTO_IMAGES || nimg > c->rxImages));
  // (the owner plane's start value: the lane relaxation takes it from the front pass; the tile relaxation's k_tx_sort writes the
  // trivial map for every pixel itself — 8 bytes per scaled pixel less for the front pass to store)
  int2* ownPlane = (sequential || c->lsdMode != 1) ? (int2*)nullptr : c->own;
  // key mode of the tile relaxation (lsd_tile.hip): no ordered list — it is only built, at the end, for images left to the sequential grower
  const bool lostRule0 = c->lsdMode != 1 && DEVENV("PLI_TX_BOXRULE") == nullptr;
  const bool keyMode = !sequential && c->lsdMode != 1 && c->txKeys && !c->debug && lostRule0 && !DEVENV("PLI_TX_FULL2") &&
                       !DEVENV("PLI_TX_OLDMARK") && !DEVENV("PLI_TX_CELLRULE");   // (k_rx_mark, lsd_relax.hip, indexes the region planes by rank)
  // packed round 1 (lsd_tile.hip, tx_load_rec16): owner_1 lives in the fourth word of the pixel records during round 1 — one
  // 16-byte gather per neighbour instead of two.  Needs the CV_64F detector (the word is free: the gradient norms have their own
  // plane) and the default schedule (k_tx_round2 moves the result into the owner plane).  In key mode the front pass writes the
  // unclaimed words itself (2: LAZY ids), otherwise k_tx_sort does (1).  Dev switch PLI_TX_PACK1=0 / 1.
  int packMode = 0;
  if (!sequential && c->lsdMode != 1 && c->lsdF64 && c->mg && lostRule0 && !c->debug && !DEVENV("PLI_TX_FULL2") && !DEVENV("PLI_TX_NOFUSE2") &&
      !(DEVENV("PLI_TX_SPEC") && atoi(DEVENV("PLI_TX_SPEC")) != 0) && !(DEVENV("PLI_TX_ORDER") && atoi(DEVENV("PLI_TX_ORDER")) != 0))
    packMode = (keyMode && P.nBins > 128) ? 2 : 1;      // (the lazy form's fixed-point bin width needs maxGrad / (nBins - 1) < 4: tests/test_lazy_ids_cpu.py)
  const int packCap = packMode;                           // (what this call could do: the choice follows the hot-record level below)
  // round 6: round 1's words in 8-byte hot records of their own (lsd_tile.hip "HOT RECORDS"); dev switch PLI_TX_HOT=0 / 1
  // 2 (the default): every round — the later rounds take the angle from the hot record and the owner pair from the owner plane, nobody
  // reads the 16-byte records, and the front pass does not write them (the device-side fallback of an unsettled image rebuilds them
  // from the hot records first: k_tx_rec_from_hot); 1: round 1 only; 0: the 16-byte records everywhere (round 5's form)
  int hotLevel = (packMode != 0 && c->hot != nullptr && P.prec <= 1.0 && P.alignFilter != 0) ? 2 : 0;
  if (const char* e = DEVENV("PLI_TX_HOT")) hotLevel = std::min(hotLevel, std::max(0, atoi(e)));
  if (hotLevel == 2 && (DEVENV("PLI_RX_FULL") || DEVENV("PLI_TX_NODIRTYLIST"))) hotLevel = 1;     // (dev schedules that keep to the plain later-round kernels)
  // On the hot records the ids are SORT-WRITTEN (1) by default, in key mode too: k_tx_sort rewrites the whole 8-byte record — angle and
  // `unclaimed | id`, full sectors — and round 1 does without the lazy-id test, ~23 of a step's ~150 vector instructions: k_tx_grow
  // 18.6 -> 17.1 ms, k_tx_sort 2.1 -> 3.2 with 4-byte stores into the records' second words (read-modify-writes of sectors) and less
  // with whole records (DESIGN.md 5 "Round 6").  On the 16-byte records the same trade was a wash (round 5) and lazy ids stay the
  // default there.  Dev switch PLI_TX_PACK1 = 0 / 1 / 2.
  if (hotLevel != 0 && packMode == 2) packMode = 1;
  if (const char* e = DEVENV("PLI_TX_PACK1")) packMode = std::min(packCap, std::max(0, atoi(e)));
  if (packMode == 0) hotLevel = 0;
  const bool hotMode = hotLevel != 0, hotAll = hotLevel == 2;
  int2* hotPlane = hotMode ? c->hot : (int2*)nullptr;
  int2* hotLater = hotAll ? c->hot : (int2*)nullptr;
  float4* recPlane = hotAll ? (float4*)nullptr : c->rec;         // (what the front pass writes: the 16-byte records, or the cold plane)
  float2* coldPlane = hotAll ? c->cold : (float2*)nullptr;
  const int trigF32 = ((c->cfg.parity_flags & PLI_PARITY_TRIG_F32_LSD) ? 1 : 0) | (packMode == 2 ? 2 : 0);
  if (c->lsdF64) {
    // OpenCV 3.x: the detector works on the CV_64FC1 copy of the image (lsd_f64.hip).  The scaled double image lives in
    // the grower's overflow area (same size, not in use before the growers run).
    double* scaled64 = reinterpret_cast<double*>(c->regScratch);
    const dim3 gb((P.W + 63) / 64, (P.H + 15) / 16, nimg);
    if (c->lsdFront64 && !c->debug) {
      // blur -> resize -> gradient in one pass over LDS tiles: neither double plane goes to HBM
      HIPCHK(hipMemsetAsync(c->maxMg + img0, 0, sizeof(unsigned long long) * nimg, c->stream));
      LAUNCH(c, "k_lsd_front", k_lsd_front64, dim3((P.LW + 63) / 64, (P.LH + 15) / 16, nimg), dim3(256), 0, c->pyr + P.lv[0].offset,
             P.pyrBlock, P.W, P.H, P.lv[0].pitch, c->kern64, c->lsdRadius, c->lsdTab64, P.LWt, P.LH, P.LW, P.rho, recPlane, c->mg, ownPlane,
             c->maxMg, img0, trigF32, hotPlane, coldPlane);
    } else {
    if (c->cfg.lsd_scale != 1) {
      LAUNCH(c, "k_blur_lsd", k_lsd_blur64, gb, dim3(256), 0, c->pyr + P.lv[0].offset, P.pyrBlock, P.W, P.H, P.lv[0].pitch,
             c->kern64, c->lsdRadius, c->tmp64, (int64_t)P.W * P.H, img0);
      LAUNCH(c, "k_resize_lsd", k_lsd_resize64, dim3((P.LWt + 255) / 256, P.LH, nimg), dim3(256), 0, c->tmp64, P.W, P.H, scaled64, P.LWt,
             P.LH, (int64_t)npix, c->lsdTab64, img0);
    } else {
      LAUNCH(c, "k_blur_lsd", k_lsd_blur64, gb, dim3(256), 0, c->pyr + P.lv[0].offset, P.pyrBlock, P.W, P.H, P.lv[0].pitch,
             c->kern64, 0, scaled64, (int64_t)npix, img0);
    }
    // (the scaled double plane: rows of the TRUE width, images npix = pitch x height apart)
    if (c->debug && c->scaled64Dbg)                    // the plane becomes the growers' arena: keep a copy for PLI_DBG_LSD_SCALED
      HIPCHK(hipMemcpyAsync(c->scaled64Dbg + (int64_t)img0 * npix, scaled64 + (int64_t)img0 * npix, (size_t)nimg * npix * 8,
                            hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->maxMg + img0, 0, sizeof(unsigned long long) * nimg, c->stream));
    LAUNCH(c, "k_lsd_grad", k_lsd_grad64, dim3((P.LW + 255) / 256, (P.LH + 15) / 16, nimg), dim3(256), 0, scaled64, P.LWt, P.LH, P.LW,
           (int64_t)npix, P.rho, recPlane, c->mg, ownPlane, c->maxMg, c->debug ? c->angDbg : (float*)nullptr, img0, trigF32, hotPlane, coldPlane);
    }
  } else {
    const uint8_t* scaled;
    int64_t sStride;
    int sPitch;
    if (c->cfg.lsd_scale != 1) {
      LAUNCH(c, "k_blur_lsd", k_blur, dim3(c->l0Tiles, nimg), dim3(256), 0, c->jobLsd, c->pyr, P.pyrBlock, c->tmp8, c->tmp8Stride, img0);
      dim3 g((P.LWt + 255) / 256, (P.LH + 15) / 16, nimg);
      LAUNCH(c, "k_resize_lsd", k_resize_level, g, dim3(256), 0, c->tmp8, c->tmp8Stride, P.W, P.H, c->tmpPitch, c->lsdScaled,
             c->lsdStride, P.LWt, P.LH, P.lpitch, c->lsdTab, img0);
      scaled = c->lsdScaled; sStride = c->lsdStride; sPitch = P.lpitch;
    } else {
      scaled = c->pyr + P.lv[0].offset; sStride = P.pyrBlock; sPitch = P.lv[0].pitch;
    }
    HIPCHK(hipMemsetAsync(c->maxG2 + img0, 0, sizeof(int) * nimg, c->stream));
    dim3 g((P.LW + 255) / 256, (P.LH + 15) / 16, nimg);   // 16 = LSD_GRAD_ROWS
    LAUNCH(c, "k_lsd_grad", k_lsd_grad, g, dim3(256), 0, scaled, sStride, P.LWt, P.LH, P.LW, sPitch, P.g2Thresh, c->rec, c->g2, ownPlane,
           c->maxG2, c->debug ? c->angDbg : (float*)nullptr, img0, trigF32);
  }
  auto orderPasses = [&](const RxCtl* only) -> pli_status {
  LAUNCH(c, "k_lsd_hist", k_lsd_hist, dim3(c->nChunks, nimg), dim3(256), 0, c->g2, npix, P.g2Thresh, P.nBins, c->maxG2,
         c->chunkHist, c->nChunks, img0, c->mg, c->maxMg, P.rho, only);
  // (thousands of chunks per image — 4K —: the scan over the chunks in groups, lsd_scanGroups > 0)
  if (c->scanGroups > 0) {
    LAUNCH(c, "k_lsd_scan", k_lsd_scan_part, dim3(nimg, c->scanGroups), dim3(1024), 0, c->chunkHist, c->nChunks, P.nBins,
           c->scanChunksPerGroup, c->chunkBase, c->scanGroupOff, img0, only);
    LAUNCH(c, "k_lsd_scan", k_lsd_scan_groups, dim3(nimg), dim3(1024), 0, c->scanGroups, P.nBins, c->scanGroupOff, c->nDefined, img0, only);
  } else
  LAUNCH(c, "k_lsd_scan", k_lsd_scan, dim3(nimg), dim3(1024), 0, c->chunkHist, c->nChunks, P.nBins, c->chunkBase, c->nDefined, img0, only);
  // 16 KB of unused dynamic LDS per single-wave workgroup caps the scatter at 8 waves per CU: with all chunks of an image on
  // one XCD (see the kernel) that keeps the ordered lists "open" in an XCD's 4 MB L2 to ~2 images, so the 4-byte stores of
  // different chunks merge into whole lines before they are evicted (2048 frames: 21.8 -> 15.3 ms; 32 KB starves the CUs)
  constexpr size_t scatterOccupancyPad = 16384;
  LAUNCH(c, "k_lsd_scatter", k_lsd_scatter, dim3((unsigned)(8 * ((nimg + 7) / 8) * c->nChunks)), dim3(64), scatterOccupancyPad, c->g2, npix, P.g2Thresh,
         P.nBins, c->maxG2, c->chunkBase, c->nChunks, c->order, img0, nimg, c->mg, c->maxMg, P.rho,
         (sequential || only) ? (int*)nullptr : c->rankOf, (const int*)(c->scanGroups > 0 ? c->scanGroupOff : nullptr), c->scanChunksPerGroup,
         c->scanGroups, only);
    return PLI_OK;
  };
  if (!keyMode) {
    pli_status os = orderPasses(nullptr);
    if (os != PLI_OK) return os;
  }
  if (sequential) {
    // speculative form (line_kernels.hip: lsd_grow_image_spec): the small regions of 64 seeds at a time, one per lane
    const bool spec = c->lsdSpec && P.minRegSize >= 2;
    const bool two = nimg >= 64 && !DEVENV("PLI_GROW_WPB1");
    // region2rect off the grower's serial chain: the spec grower leaves the pixel lists in the arena, k_lsd_rect does the segments
    LsdRectItem* items = (spec && DEVENV("PLI_LSD_RECT_OFFLOAD")) ? c->rectItems : (LsdRectItem*)nullptr;      // dev switch: measured, not a gain
    if (two && spec)
      LAUNCH(c, "k_lsd_grow2", k_lsd_grow2_spec, dim3((nimg + 1) / 2), dim3(128), 0, c->dP, c->rec, c->mg, c->order, c->nDefined,
             c->regScratch, c->seg, c->nSeg, c->maxSeg, img0, nimg, items, c->rectW);
    else if (two)
      LAUNCH(c, "k_lsd_grow2", k_lsd_grow2, dim3((nimg + 1) / 2), dim3(128), 0, c->dP, c->rec, c->mg, c->order, c->nDefined,
             c->regScratch, c->seg, c->nSeg, c->maxSeg, img0, nimg);
    else if (spec)
      LAUNCH(c, "k_lsd_grow", k_lsd_grow_spec, dim3(nimg), dim3(64), 0, c->dP, c->rec, c->mg, c->order, c->nDefined,
             c->regScratch, c->seg, c->nSeg, c->maxSeg, img0, nimg, items, c->rectW);
    else
      LAUNCH(c, "k_lsd_grow", k_lsd_grow, dim3(nimg), dim3(64), 0, c->dP, c->rec, c->mg, c->order, c->nDefined,
             c->regScratch, c->seg, c->nSeg, c->maxSeg, img0, nimg);
#ifdef PLI_DEV
    if (items)
      LAUNCH(c, "k_lsd_rect", k_lsd_rect, dim3(16, nimg), dim3(64), 0, c->dP, (const LsdRectItem*)items, (const uint2*)c->regScratch,
             (const double*)c->mg, (const double*)c->rectW, (const int*)c->nSeg, c->seg, c->maxSeg, img0);     // 16 = RECT_WPI
#endif
  } else {
    const bool trace = DEVENV("PLI_RX_TRACE") != nullptr;
    const bool fullPasses = DEVENV("PLI_RX_FULL") != nullptr;      // dev: full-image bookkeeping in every round
    const bool perRound = DEVENV("PLI_RX_PROFROUNDS") != nullptr;    // profile names carry the round number
    int curT = 0;
    auto rxn = [&](const char* n) -> const char* {
      if (!perRound) return n;
      static std::set<std::string> pool;
      char b[64];
      std::snprintf(b, sizeof(b), "%s@%02d", n, curT);
      return pool.insert(b).first->c_str();
    };
    // incremental rank-ordered relaxation (lsd_relax.hip): rounds until every image's owner map is a fixed point
    const int64_t npix64 = (int64_t)npix;
    int growBlocks = std::max(16, std::min(256, 2048 / nimg));
    if (const char* e = DEVENV("PLI_JR_BLOCKS")) growBlocks = std::max(1, atoi(e));
    int bigBlocks = std::max(64, std::min(1024, 16384 / nimg));
    if (const char* e = DEVENV("PLI_JR_BIGBLOCKS")) bigBlocks = std::max(1, atoi(e));
    int rectBlocks = std::max(64, std::min(2048, 32768 / nimg));
    if (const char* e = DEVENV("PLI_RECT_BLOCKS")) rectBlocks = std::max(1, atoi(e));      // dev: waves per image of k_rx_rect
    int bigThresh = RX_HAND;
    if (const char* e = DEVENV("PLI_JR_BIG")) bigThresh = atoi(e);
    int maxRounds = 96;
    if (const char* e = DEVENV("PLI_RX_MAXROUNDS")) maxRounds = std::max(1, atoi(e));
    const float precDeg = (float)(P.prec * 180.0 / 3.14159265358979323846);
    (void)growBlocks; (void)bigBlocks; (void)bigThresh; (void)precDeg;     // (the lane relaxation's: development build)
    if (c->rgClean) HIPCHK(hipMemsetAsync(c->rgClean + (int64_t)img0 * npix, 0, npix64 * nimg, c->stream));   // round stamps
    const bool lostRule = c->lsdMode != 1 && DEVENV("PLI_TX_BOXRULE") == nullptr;     // dev switch: the conservative round-2 rule
    {
      // everything the relaxation wants zeroed at the start of a call, in one launch (k_zero_ranges): control blocks, the stamp planes,
      // the cell tables, and the tile relaxation's per-tile dirty counters (cleared by their consumers from then on), the barrier words
      // of k_tx_tail and the candidate counters of k_tx_collect
      const int64_t cells = (int64_t)c->tilesW * c->tilesH;
      const int64_t ntile64 = (int64_t)c->txNtx * c->txNty;
      static_assert(sizeof(RxCtl) % 4 == 0, "RxCtl is cleared as words");
      ZeroRanges Z{};
      int zr = 0;
      constexpr int zmax = (int)(sizeof(Z.p) / sizeof(Z.p[0]));
      bool zfull = false;
      auto add = [&](void* p, int64_t words) {
        if (!p || words <= 0) return;
        if (zr == zmax) { zfull = true; return; }
        Z.p[zr] = (uint32_t*)p; Z.words[zr] = words; ++zr;
      };
      add(c->jrCtl + img0, (int64_t)(sizeof(RxCtl) / 4) * nimg);
      // (tile relaxation: k_tx_sort clears the two per-pixel stamp planes while it writes the id plane)
      if (c->lsdMode == 1 || DEVENV("PLI_TX_ZERO_BY_FILL")) {
        add(c->rgDirty + (int64_t)img0 * npix, npix64 * nimg);
        if (lostRule) add(c->rgLost + (int64_t)img0 * npix, npix64 * nimg);
      }
      add(c->tileTouch + (int64_t)img0 * cells, cells * nimg);
      add(c->tileAct + (int64_t)img0 * cells, cells * nimg);
      if (c->txDirtyCnt) add(c->txDirtyCnt + (int64_t)img0 * ntile64, ntile64 * nimg);
      if (c->tailBar) add(c->tailBar, 64);
      if (c->txCellCnt) add(c->txCellCnt, (int64_t)2 * TX_CELL_ROUNDS * c->rxImages);
      if (c->txCandCnt) add(c->txCandCnt + img0, nimg);
      if (zfull) { g_err = "k_zero_ranges: more buffers than ZeroRanges holds"; return PLI_ERR_INVALID; }
      const int zb = (int)std::max<int64_t>(1, std::min<int64_t>(8192, (npix64 * nimg / 4 + 255) / 256));
      LAUNCH(c, "k_zero_ranges", k_zero_ranges, dim3(zb, zr), dim3(256), 0, Z);
    }
    // (the rank plane was written by k_lsd_scatter)
    const dim3 raster((P.LW + 255) / 256, P.LH, nimg);
    const bool tile = c->lsdMode != 1;      // auto below RX_AUTO_IMAGES and mode 3: the tile-sequential relaxation
    bool allDone = false;
    // what the last look-free call left behind (if its copy has arrived): how many rounds its slowest image needed
    if (c->rxSeenImages > 0 && hipEventQuery(c->evRxSeen) == hipSuccess) foldRoundStats(c);
    if (tile && c->tailBar && c->tailBlocks == 0) {       // the resident grid of k_tx_tail: CUs x blocks per CU (-1: it does not fit)
      int perCu = 0, cus = 0;
      HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_tx_tail, 256, 0));
      HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
      int bpc = 1;
      if (const char* e = DEVENV("PLI_TX_TAIL_BPC")) bpc = std::max(1, atoi(e));
      c->tailBlocks = perCu < 1 ? -1 : std::min(perCu, bpc) * std::max(1, cus);
    }
    c->tailLaunched = false;
    const bool tailLatched = c->device < 0 || c->device >= TAIL_MAX_DEVICES || g_tailAborted[c->device].load(std::memory_order_relaxed);   // (no chain slot for the device: no tail)
    const bool tailPossible = tile && c->tailBar && c->tailBlocks > 0 && !tailLatched && !DEVENV("PLI_RX_PLAN") && !trace && !perRound && !DEVENV("PLI_RX_BLOCKING") &&
                              !(getenv("PLI_TX_TAIL") && atoi(getenv("PLI_TX_TAIL")) == 0) && lostRule && !DEVENV("PLI_TX_FULL2") &&
                              !DEVENV("PLI_TX_CELLRULE") && !DEVENV("PLI_TX_OLDMARK") && !DEVENV("PLI_TX_FULLDIFF") &&
                              !DEVENV("PLI_TX_NOFUSEDM") && !DEVENV("PLI_TX_NODIRTYLIST");
    const bool blocking = !tailPossible && (c->rxLastRounds == 0 || trace || DEVENV("PLI_RX_BLOCKING") != nullptr);
    if (!blocking && !tailPossible) maxRounds = std::min(maxRounds, c->rxLastRounds + c->rxMargin);
    if (!blocking) if (const char* e = DEVENV("PLI_RX_PLAN")) maxRounds = std::max(1, atoi(e));   // dev / test: a plan that is too short
    c->rxPlanned = blocking ? 0 : (tailPossible ? -1 : maxRounds);
    const int firstLook = c->rxLastRounds > 0 ? std::max(4, c->rxLastRounds) : 4;
    auto look = [&](int t) -> pli_status {
      // the host looks at the state every second round, starting where the previous call on this context ended (a
      // stream of similar frames settles after a similar number of rounds; each look drains the stream)
      if (!blocking) return PLI_OK;
      if ((t >= firstLook && ((t - firstLook) % 2) == 0) || t == maxRounds) {
        HIPCHK(hipMemcpyAsync(c->jrHost.data(), c->jrCtl + img0, sizeof(RxCtl) * nimg, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        allDone = true;
        int settled = 0;
        for (int i = 0; i < nimg; ++i) {
          allDone = allDone && (c->jrHost[i].state == 2 || c->jrHost[i].overflow);
          settled = std::max(settled, c->jrHost[i].rounds);
        }
        if (allDone) c->rxLastRounds = settled;       // the round in which the last image reached its fixed point
      }
      return PLI_OK;
    };
    if (tile) {
      // tile-sequential relaxation (lsd_tile.hip): per-tile seed lists once, then rounds of one wave per tile
      const int ts = c->txTs, ntile = c->txNtx * c->txNty;
      TxKeys keys{};
      if (keyMode) keys = TxKeys{c->mg, c->maxMg, P.rho, P.nBins, c->txPixBits, c->rankOf};
      const bool pack1 = packMode != 0;
      if (pack1) { keys.recPack = c->rec; keys.pack = packMode; keys.hot = hotPlane; keys.cold = coldPlane; }
      if (!DEVENV("PLI_TX_ZERO_BY_FILL")) { keys.zeroA = c->rgDirty; keys.zeroB = lostRule ? c->rgLost : (int*)nullptr; }
      // (test switch: a wide margin sends every unclaimed pixel of the LAZY form through the double plane; the results must not change)
      if (const char* e = DEVENV("PLI_TX_LAZY_MARGIN")) keys.lazyMargin = std::max(4, std::min(0x3FFFFFFF, atoi(e)));
#ifdef PLI_DEV
      if (ts == 128)
        TRL(c, "k_tx_sort", k_tx_sort128, dim3(ntile, nimg), dim3(1024), 0, c->rankOf, c->order, c->own, c->txList,
            c->txTileCnt, P.LW, P.LH, ts, c->txNtx, c->txNty, img0, keys);
      else
#endif
        TRL(c, "k_tx_sort", k_tx_sort, dim3(ntile, nimg), dim3(256), 0, c->rankOf, c->order, c->own, c->txList,
            c->txTileCnt, P.LW, P.LH, ts, c->txNtx, c->txNty, img0, keys);
      const bool fullRound2 = DEVENV("PLI_TX_FULL2") != nullptr;       // dev: regrow everything in round 2
      const size_t txPad = DEVENV("PLI_TX_LDSPAD") ? (size_t)atoi(DEVENV("PLI_TX_LDSPAD")) : 0;   // dev: occupancy cap of the tile growers
      TxDirtyLists DL{c->txDirtyList, c->txDirtyCnt, c->order, ts, c->txNtx, c->txNty, P.LW, npix64};
      if (keyMode) DL.rmask = (1 << c->txPixBits) - 1;
      if (DEVENV("PLI_TX_NODIRTYLIST")) DL.list = nullptr;             // dev: every active tile walks its whole seed list
      TxDirtyLists noDL = DL; noDL.list = nullptr;
      // rounds >= 3: k_rx_diff + k_tx_mark as one kernel (k_tx_diffmark; the dev rules keep the separate passes)
      const bool fusedDM = lostRule && !fullRound2 && !DEVENV("PLI_TX_CELLRULE") && !DEVENV("PLI_TX_OLDMARK") && !DEVENV("PLI_TX_FULLDIFF") &&
                           !DEVENV("PLI_TX_NOFUSEDM");
      // rounds >= tailT0 in one persistent launch that ends by itself when every image is at its fixed point (no host look, no
      // planned round count; dev switches: PLI_TX_TAIL=0, PLI_TX_TAIL_T0, PLI_TX_TAIL_BPC)
      // Where the deferred ORB chain forks from the line chain (pli_batch_run): behind round 1's growth up to 64 images, behind round 2's
      // pass over the owner map (-2) above — what waits in a large batch is the END of the line chain (the 1024-thread / 64 KB workgroups
      // of k_tx_emit_sorted, the gated k_lsd_scan, k_keylines find no room while the ORB chain fills every hole), so the chain should end
      // early, and k_tx_round2, which the ORB kernels stretch most, should run alone.  Same box, 256 frames, fork behind round 1 /
      // round 2's owner pass / round 2 / round 3: synthetic 45.0 / 45.0 / 45.4-45.5 / 46.2 ms, photographs 28.6 / 28.5 / 29.5 / 30.0
      // (DESIGN.md 5 has the history).  Dev switch PLI_SIDE_FORK_ROUND.
      const int sideForkEnv = DEVENV("PLI_SIDE_FORK_ROUND") ? atoi(DEVENV("PLI_SIDE_FORK_ROUND")) : 0;
      const int sideForkRound = sideForkEnv ? sideForkEnv : (nimg <= 64 ? 1 : -2);
      // (round 1's region2rect pass on a stream of its own beside k_tx_round2: the default schedule only; dev switch PLI_RECT_ASIDE=0)
      // (a single pair pays 0.06 ms for the two events and the reset launch and has nothing to overlap: from 8 images on)
      // (the ORB chain's stream when the chain is waiting to fork behind round 2 or later: idle until then)
      const bool rectOnSide = c->sideChain && c->aux && (sideForkRound >= 2 || sideForkRound == -2) && !DEVENV("PLI_RECT_OWN_STREAM");
      const bool rectAside = lostRule && !fullRound2 && !DEVENV("PLI_TX_NOFUSE2") && !trace && !perRound && !blocking && maxRounds >= 2 && !c->syncDebug &&
                             (DEVENV("PLI_RECT_ASIDE") ? atoi(DEVENV("PLI_RECT_ASIDE")) != 0 : (nimg >= 8 && (rectOnSide || c->sH2D == nullptr)));
      const bool useTail = tailPossible && fusedDM && DL.list;
      // rounds 3 .. tail start of a large batch by cell lists (six lean launches per round instead of four that walk every block;
      // a small batch keeps the four: launches are what it pays for).  Dev switch PLI_TX_CELLS=0 / 1.
      // (measured: 256 frames of 752 x 480 +2 %, 64 of 1280 x 720 +1 %, 16 of 4K +2 %; 32 frames of 752 x 480 -4 %: a million cells is the line)
      bool useCells = fusedDM && DL.list && c->txCellList && (int64_t)c->tilesW * c->tilesH * nimg >= (1 << 20);
      if (const char* e = DEVENV("PLI_TX_CELLS")) useCells = atoi(e) != 0 && fusedDM && DL.list && c->txCellList;
      // The persistent kernel takes over at round 8 — at round 12 when the rounds before it run on cell lists: their launches are lean, and
      // a large batch still has the ORB chain on the chip around round 8, beside which the tail's 256 workgroups of 50 KB LDS, which must
      // ALL be resident, wait for their compute units (k_tx_tail 0.7 ms alone, 2.6 ms in the default line).  256 frames, same box, start
      // at 8 / 12 / 14 / 16: synthetic 46.0-46.1 / 45.7 / 45.6 / 45.6 ms, photographs 29.2-29.3 / 29.3 / 29.4 / 29.7.
      int tailT0 = useCells ? 12 : 8;
      if (const char* e = DEVENV("PLI_TX_TAIL_T0")) tailT0 = std::max(3, atoi(e));
      for (int t = 1; t <= maxRounds && !allDone; ++t) {
        curT = t;
        if (useTail && t == tailT0) {
          // (the dirty counters are zero: every list of the last round was taken by its tile's wave; the barrier words were cleared
          // with the control blocks)
          TxTailArgs ta{c->dP, c->jrCtl, c->rec, c->own, c->txList, c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty,
                        c->tileAct, c->tileTouch, c->tilesW, c->tilesH, c->arena, c->arenaCap, c->rects, c->rectCap, c->rgSeg, c->mg,
                        c->rankOf, c->rgLost, DL, img0, nimg, t, maxRounds, c->tailBar};
          ta.hot = hotLater; ta.cold = coldPlane;
          // (a grid barrier costs 8 us with 256 workgroups and less with fewer: a few images do not need the whole chip in the late rounds)
          // (single pair: k_tx_tail 149 us with 256 workgroups, 90 with 64)
          int tb = std::min(c->tailBlocks, std::max(64, 8 * nimg));
          if (const char* e = DEVENV("PLI_TX_TAIL_BLOCKS")) tb = std::max(1, std::min(c->tailBlocks, atoi(e)));     // dev switch
          // k_tx_tail spins on a grid barrier, so its grid must be co-resident; the grid is sized for a device it has to itself.  Two
          // tails in flight at once (two contexts of this process on one device) could each hold a part of the other's slots and spin
          // until the barrier's timeout: tails of one process are chained per device — a tail starts when the previous one has ended.
          // (Other PROCESSES on the device are out of reach of this chain: PLI_TX_TAIL=0 selects the planned-rounds schedule there,
          // include/pli_frontend.h "Sharing a device".)
          {
            std::lock_guard<std::mutex> lk(g_tailMu);
            hipEvent_t& ev = g_tailEv[c->device];             // (0 <= device < TAIL_MAX_DEVICES: tailPossible)
            c->tailLaunched = true;
            ta.forceAbort = DEVENV("PLI_TX_TAIL_FORCE_ABORT") ? 1 : 0;   // test switch: the kernel leaves at once, as after a barrier timeout
            if (ev) HIPCHK(hipStreamWaitEvent(c->stream, ev, 0));
            else HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            LAUNCH(c, "k_tx_tail", k_tx_tail, dim3(tb), dim3(256), 0, ta);
            HIPCHK(hipEventRecord(ev, c->stream));
          }
          break;
        }
        // (the per-tile dirty counters: cleared once per call with the control blocks, then by the wave that takes a tile's list;
        // the dev schedule that regrows everything in round 2 does not read the lists it stamped)
        if (t == 3 && DL.list && fullRound2)
          HIPCHK(hipMemsetAsync(c->txDirtyCnt + (int64_t)img0 * ntile, 0, sizeof(int) * (size_t)ntile * nimg, c->stream));
        const bool fused2 = t == 2 && !fullRound2 && lostRule && !DEVENV("PLI_TX_NOFUSE2");    // (dev switch: the two passes)
        bool cellsDone = false;
        if (fused2) {
          TRL(c, "k_tx_round2", k_tx_round2, dim3((P.LW + 31) / 32, (P.LH + 31) / 32, nimg), dim3(256), 0, c->jrCtl, c->own, c->rankOf, c->order,
              c->rgBox, c->rgDirty, c->tileAct, P.LW, P.LH, c->tilesW, c->tilesH, t, img0, (const int*)c->rgLost, DL, c->tileTouch,
              (pack1 && !hotPlane) ? (const float4*)c->rec : (const float4*)nullptr, rectAside ? 1 : 0, (const int2*)hotPlane);
          if (rectAside) {                               // (round 1's region2rect pass has run beside this kernel: join, then the counter)
            HIPCHK(hipStreamWaitEvent(c->stream, c->evRectDone, 0));
            TRL(c, "k_tx_reset_rect", k_tx_reset_rect, dim3((nimg + 255) / 256), dim3(256), 0, c->jrCtl, nimg, img0);
          }
          if (c->sideChain && sideForkRound == -2) {      // (dev switch: the ORB chain forks behind round 2's pass over the owner map, before its growth)
            auto f = std::move(c->sideChain);
            c->sideChain = nullptr;
            pli_status ss = f();
            if (ss != PLI_OK) return ss;
          }
        } else if (t >= 3 && fusedDM && useCells && t < TX_CELL_ROUNDS) {
          // cell lists (lsd_tile.hip "CELL LISTS"): the cells touched in round t - 1 -> their changed pixels' marks -> the active cells
          // -> their owner words.  A list and its length per image (and round: zeroed once per call) stay on the device; a few
          // workgroups per image stride over its list.
          const int ncell = c->tilesW * c->tilesH;
          // (2048 = TX_CELL_CHUNK cells per listing workgroup; the consumers: 32 waves per image, more for large images)
          const dim3 lg((unsigned)((ncell + 2047) / 2048), nimg), wg((unsigned)(4 * std::max(8, std::min(256, std::max(ncell / 1024, 4096 / nimg)))), nimg);      // (single-wave workgroups)
          int* cntA = c->txCellCnt + ((int64_t)2 * t) * c->rxImages + img0;
          int* cntB = c->txCellCnt + ((int64_t)2 * t + 1) * c->rxImages + img0;
          int* lst = c->txCellList + (int64_t)img0 * ncell;
          TRL(c, "k_tx_cells", k_tx_cells, lg, dim3(256), 0, c->jrCtl, (const int*)c->tileTouch, c->tileTouch, ncell, nimg, img0, t, 0, lst, cntA);
          TRL(c, "k_tx_diffmark", k_tx_diffmark_cells, wg, dim3(64), 0, c->jrCtl, (const int2*)c->own, (const int*)c->rankOf, (const int2*)c->rgBox,
              c->rgDirty, c->tileAct, P.LW, P.LH, c->tilesW, c->tilesH, t, img0, (const int*)c->rgLost, DL, (const int*)lst, (const int*)cntA);
          TRL(c, "k_tx_cells", k_tx_cells, lg, dim3(256), 0, c->jrCtl, (const int*)c->tileAct, c->tileTouch, ncell, nimg, img0, t, 1, lst, cntB);
          TRL(c, "k_tx_prep", k_tx_prep_cells, wg, dim3(64), 0, (const RxCtl*)c->jrCtl, c->own, (const int*)c->rankOf, (const int*)c->rgDirty, P.LW,
              P.LH, c->tilesW, c->tilesH, t, img0, DL.rmask, (const int*)lst, (const int*)cntB);
          cellsDone = true;
        } else if (t >= 3 && fusedDM)
          TRL(c, "k_tx_diffmark", k_tx_diffmark, dim3((P.LW + 31) / 32, (c->tilesH + 7) / 8, nimg), dim3(256), 0, c->jrCtl, c->own, c->rankOf,
              c->rgBox, c->rgDirty, c->tileAct, (const int*)c->tileTouch, P.LW, P.LH, c->tilesW, c->tilesH, t, img0, (const int*)c->rgLost, DL);
#ifdef PLI_DEV     // (the unfused passes of the dev switches: the product's round 2 is k_tx_round2, its rounds >= 3 k_tx_diffmark or the cell lists)
        else if (t == 2 && !fullRound2)
          TRL(c, "k_tx_diff2", k_tx_diff2, dim3((P.LW + 31) / 32, c->tilesH, nimg), dim3(256), 0, c->jrCtl, c->own, c->order, c->rgBox,
              c->rgDirty, c->tileMin, c->tileAct, P.LW, P.LH, c->tilesW, c->tilesH, ts, t, img0, lostRule ? (const int*)c->rgLost : (const int*)nullptr, DL);
        else if (t >= 2)                     // (round 1 starts from the trivial map: nothing to compare, the control block is zeroed)
          TRL(c, "k_rx_diff", k_rx_diff, dim3((P.LW + 31) / 32, (c->tilesH + 7) / 8, nimg), dim3(256), 0, c->jrCtl, c->own, c->tileMin, c->tileAct,
              P.LW, P.LH, c->tilesW, c->tilesH, t, img0,
              (t >= 3 && !fullRound2 && !DEVENV("PLI_TX_FULLDIFF")) ? (const int*)c->tileTouch : (const int*)nullptr);
#endif
        if (t >= 3 || (t == 2 && !fullRound2)) {
          // (round 2 under the lost-pixel rule: k_tx_diff2 has stamped the regions itself, k_rx_mark would find nothing)
          if (t >= 3 && fusedDM) {
            // (k_tx_diffmark has done it)
          }
#ifdef PLI_DEV
          else if (lostRule && t >= 3 && !DEVENV("PLI_TX_CELLRULE") && !DEVENV("PLI_TX_OLDMARK"))
          TRL(c, "k_rx_mark", k_tx_mark, dim3((P.LW + 31) / 32, (c->tilesH + 7) / 8, nimg), dim3(256), 0, c->jrCtl, c->own, c->rankOf, c->rgBox,
              c->rgDirty, c->tileMin, c->tileAct, P.LW, P.LH, c->tilesW, c->tilesH, t, img0, (const int*)c->rgLost, DL);
          else if (t >= 3 || !lostRule)
          TRL(c, "k_rx_mark", k_rx_mark, dim3((P.LW + 31) / 32, (c->tilesH + 7) / 8, nimg), dim3(256), 0, c->jrCtl, c->own, c->rankOf, c->rgBox,
              c->rgDirty, c->tileMin, c->tileAct, P.LW, P.LH, c->tilesW, c->tilesH, t, img0, t >= 3 ? 1 : 0,
              (lostRule && t >= 3 && !DEVENV("PLI_TX_CELLRULE")) ? (const int*)c->rgLost : (const int*)nullptr, DL);
#endif
          if (!fused2 && !cellsDone)
          TRL(c, "k_tx_prep", k_tx_prep, dim3((P.LW + 31) / 32, (P.LH + 31) / 32, nimg), dim3(256), 0, c->jrCtl, c->own, c->rankOf,
              c->rgDirty, c->tileAct, P.LW, P.LH, c->tilesW, c->tilesH, t, img0, t == 2 ? 1 : 0, fusedDM ? c->tileTouch : (int*)nullptr, DL.rmask);
          if (hotLater)
          TRL(c, "k_tx_grow_sparse", k_tx_grow_sparse_h, dim3(ntile, nimg), dim3(64), txPad, c->dP, c->jrCtl, c->rec, c->own, c->txList,
              c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty, c->tileAct, c->tilesW, c->tilesH, c->arena,
              c->arenaCap, c->rects, c->rectCap, img0, t, (const int*)c->rankOf, lostRule ? c->rgLost : (int*)nullptr, c->tileTouch,
              fullRound2 && t == 2 ? noDL : DL, hotLater, (const float2*)coldPlane);
          else
          TRL(c, "k_tx_grow_sparse", k_tx_grow_sparse, dim3(ntile, nimg), dim3(64), txPad, c->dP, c->jrCtl, c->rec, c->own, c->txList,
              c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty, c->tileAct, c->tilesW, c->tilesH, c->arena,
              c->arenaCap, c->rects, c->rectCap, img0, t, (const int*)c->rankOf, lostRule ? c->rgLost : (int*)nullptr, c->tileTouch,
              fullRound2 && t == 2 ? noDL : DL);
        } else {
          // round 1 in the speculative schedule (lanes grow the small regions of 64 alive seeds at a time, lsd_tile.hip): exact, built
          // and measured in round 3, NOT the default — 37 ms against 22 ms of the plain schedule at 256 frames (the
          // divergent per-lane accept path costs more instructions than the whole-wave steps it replaces).  Dev switch PLI_TX_SPEC=1.
#ifdef PLI_DEV
          const bool specRound1 = lostRule && P.minRegSize >= 3 && DEVENV("PLI_TX_SPEC") && atoi(DEVENV("PLI_TX_SPEC")) != 0;
          if (specRound1)
          TRL(c, "k_tx_grow", k_tx_grow_spec, dim3(ntile, nimg), dim3(64), txPad, c->dP, c->jrCtl, c->rec, c->own, c->txList,
              c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty, c->tileAct, c->tilesW, c->tilesH, c->arena,
              c->arenaCap, c->rects, c->rectCap, img0, t, (const int*)c->rankOf, lostRule ? c->rgLost : (int*)nullptr, c->tileTouch, noDL);
          else
#endif
          {
          // dev switch PLI_TX_ORDER=1: the heaviest tiles first (k_tx_order).  Measured, not the default: the launch does end on fewer
          // stragglers, but consecutive workgroups no longer grow neighbouring tiles of one image and k_tx_grow goes from 23.2 to
          // 31.3 ms at 256 frames (4K: 33.1 -> 44.2 ms) — the grid order's locality is worth more than the balance
          TxDirtyLists r1DL = noDL;
          if ((nimg % 8) == 0 && DEVENV("PLI_TX_XCD") && atoi(DEVENV("PLI_TX_XCD")) != 0) r1DL.xcdAffine = 1;
#ifdef PLI_DEV
          if (DEVENV("PLI_TX_ORDER") && atoi(DEVENV("PLI_TX_ORDER")) != 0) {
            TRL(c, "k_tx_order", k_tx_order, dim3(1), dim3(1024), 0, (const int*)c->txTileCnt, ntile, nimg, img0, c->txPerm + (int64_t)img0 * ntile);
            r1DL.perm = c->txPerm + (int64_t)img0 * ntile;
          }
#endif
          if (hotPlane && packMode == 2)
          TRL(c, "k_tx_grow", k_tx_grow_h2, dim3(ntile, nimg), dim3(64), txPad, c->dP, c->jrCtl, c->rec, c->own, c->txList,
              c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty, c->tileAct, c->tilesW, c->tilesH, c->arena,
              c->arenaCap, c->rects, c->rectCap, img0, t, (const int*)c->rankOf, lostRule ? c->rgLost : (int*)nullptr, c->tileTouch, r1DL, keys);
          else if (hotPlane)
          TRL(c, "k_tx_grow", k_tx_grow_h1, dim3(ntile, nimg), dim3(64), txPad, c->dP, c->jrCtl, c->rec, c->own, c->txList,
              c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty, c->tileAct, c->tilesW, c->tilesH, c->arena,
              c->arenaCap, c->rects, c->rectCap, img0, t, (const int*)c->rankOf, lostRule ? c->rgLost : (int*)nullptr, c->tileTouch, r1DL, keys);
          else if (packMode == 2)
          TRL(c, "k_tx_grow", k_tx_grow_p2, dim3(ntile, nimg), dim3(64), txPad, c->dP, c->jrCtl, c->rec, c->own, c->txList,
              c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty, c->tileAct, c->tilesW, c->tilesH, c->arena,
              c->arenaCap, c->rects, c->rectCap, img0, t, (const int*)c->rankOf, lostRule ? c->rgLost : (int*)nullptr, c->tileTouch, r1DL, keys);
          else if (packMode == 1)
          TRL(c, "k_tx_grow", k_tx_grow_p1, dim3(ntile, nimg), dim3(64), txPad, c->dP, c->jrCtl, c->rec, c->own, c->txList,
              c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty, c->tileAct, c->tilesW, c->tilesH, c->arena,
              c->arenaCap, c->rects, c->rectCap, img0, t, (const int*)c->rankOf, lostRule ? c->rgLost : (int*)nullptr, c->tileTouch, r1DL, keys);
          else
          TRL(c, "k_tx_grow", k_tx_grow, dim3(ntile, nimg), dim3(64), txPad, c->dP, c->jrCtl, c->rec, c->own, c->txList,
              c->txTileCnt, ts, c->txNtx, c->txNty, c->lastSize, c->rgBox, c->rgDirty, c->tileAct, c->tilesW, c->tilesH, c->arena,
              c->arenaCap, c->rects, c->rectCap, img0, t, (const int*)c->rankOf, lostRule ? c->rgLost : (int*)nullptr, c->tileTouch, r1DL);
          }
          if (c->sideChain && sideForkRound <= 1 && sideForkRound != -2) {       // (pli_batch_run: the ORB chain forks here, behind round 1)
            auto f = std::move(c->sideChain);
            c->sideChain = nullptr;
            pli_status ss = f();
            if (ss != PLI_OK) return ss;
          }
        }
        if (c->sideChain && t >= 2 && t == sideForkRound) {   // (dev switch PLI_SIDE_FORK_ROUND: ... behind the growth of a later round)
          auto f = std::move(c->sideChain);
          c->sideChain = nullptr;
          pli_status ss = f();
          if (ss != PLI_OK) return ss;
        }
        if (t == 1 && rectAside) {
          // Round 1's region2rect pass (gather / latency bound, 2.9 ms at 256 frames) beside round 2's pass over the owner map (bandwidth
          // bound, 2.1 ms): the two touch disjoint data — the pass reads the round's pixel lists and writes the segment plane, k_tx_round2
          // reads the packed owner words and writes the owner plane — except the list counter, which k_tx_round2 leaves alone here and
          // k_tx_reset_rect clears after the join, before round 2's growth allocates from it.
          if (!c->evRectFork) {
            HIPCHK(hipEventCreateWithFlags(&c->evRectFork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->evRectDone, hipEventDisableTiming));
          }
          // Which stream: the runtime maps streams onto FOUR hardware queues, and a context that pipelines host batches already has
          // four (kernels, the ORB chain's side stream, the two copy streams) — a fifth shares a queue with a copy stream, and the
          // host-inclusive rate fell from 0.99x to 0.93x of the resident one.  The ORB chain's stream is idle until the chain forks
          // (behind round 2 in large batches): the pass borrows it then; otherwise a stream of its own, unless the host pipeline is in use.
          hipStream_t rectS = rectOnSide ? c->aux : c->aux2;
          if (!rectS) {
            HIPCHK(hipStreamCreateWithFlags(&c->aux2, hipStreamNonBlocking));
            rectS = c->aux2;
          }
          hipStream_t mainS = c->stream;
          HIPCHK(hipEventRecord(c->evRectFork, mainS));
          HIPCHK(hipStreamWaitEvent(rectS, c->evRectFork, 0));
          c->stream = rectS;
          hipError_t le = hipSuccess;
          {
            auto launchRect = [&]() -> pli_status {
              TRL(c, "k_rx_rect", k_rx_rect, dim3(rectBlocks, nimg), dim3(64), 0, c->dP, c->jrCtl, c->rec, c->arena, c->arenaCap, c->rects,
                  c->rectCap, c->rgSeg, img0, c->mg, DL.rmask, (const int2*)hotPlane, (const float2*)coldPlane);
              return PLI_OK;
            };
            const pli_status rs = launchRect();
            c->stream = mainS;
            if (rs != PLI_OK) return rs;
          }
          (void)le;
          HIPCHK(hipEventRecord(c->evRectDone, rectS));
        } else
        TRL(c, "k_rx_rect", k_rx_rect, dim3(rectBlocks, nimg), dim3(64), 0, c->dP, c->jrCtl, c->rec, c->arena, c->arenaCap, c->rects,
            c->rectCap, c->rgSeg, img0, c->mg, DL.rmask, (const int2*)hotPlane, (const float2*)coldPlane);
        if (trace) {
          HIPCHK(hipMemcpyAsync(c->jrHost.data(), c->jrCtl + img0, sizeof(RxCtl) * nimg, hipMemcpyDeviceToHost, c->stream));
          HIPCHK(hipStreamSynchronize(c->stream));
          const RxCtl& h = c->jrHost[0];
          std::fprintf(stderr, "[tx] t=%d state=%d changed=%d overflow=%d rect=%d arena=%lld\n", t, h.state, h.changed, h.overflow,
                       (int)(h.rectArena >> RX_ARENA_BITS), (long long)(h.rectArena & ((1ull << RX_ARENA_BITS) - 1ull)));
        }
        pli_status ls = look(t);
        if (ls != PLI_OK) return ls;
      }
    } else {
#ifndef PLI_DEV
      g_err = "lsd_mode 1 (the lane relaxation) is a development schedule: libpli_frontend_dev.so";
      return PLI_ERR_INVALID;
#else
    TRL(c, "k_rx_guess", k_rx_guess, raster, dim3(256), 0, c->rec, c->rankOf, c->own, P.LW, P.LH, precDeg, img0);
    for (int t = 1; t <= maxRounds && !allDone; ++t) {
      curT = t;
      TRL(c, "k_rx_diff", k_rx_diff, dim3((P.LW + 31) / 32, (c->tilesH + 7) / 8, nimg), dim3(256), 0, c->jrCtl, c->own, c->tileMin, c->tileAct,
             P.LW, P.LH, c->tilesW, c->tilesH, t, img0, (const int*)nullptr);
      if (t >= 3 && !fullPasses) {
        // bookkeeping only where something happened (lsd_relax.hip)
        TRL(c, "k_rx_mark", k_rx_mark, dim3((P.LW + 31) / 32, (c->tilesH + 7) / 8, nimg), dim3(256), 0, c->jrCtl, c->own, c->rankOf, c->rgBox,
            c->rgDirty, c->tileMin, c->tileAct, P.LW, P.LH, c->tilesW, c->tilesH, t, img0, 1, (const int*)nullptr, TxDirtyLists{});
        TRL(c, "k_rx_seed_sparse", k_rx_seed_sparse, dim3((P.LW + 31) / 32, (P.LH + 31) / 32, nimg), dim3(1024), 0, c->jrCtl, c->own, c->rankOf,
            c->rec, c->lastSize, c->rgDirty, c->tileAct, c->smallSeeds, c->bigSeeds, c->bigCap, P.LW, P.LH, c->tilesW, c->tilesH,
            bigThresh, t, img0);
      } else {
        if (t >= 2)
          TRL(c, "k_rx_classify", k_rx_classify, raster, dim3(256), 0, c->jrCtl, c->own, c->rankOf, c->rgBox,
                 c->rgClean, c->tileMin, P.LW, P.LH, c->tilesW, c->tilesH, t, img0);
        TRL(c, "k_rx_seed", k_rx_seed, dim3((P.LW + 1023) / 1024, P.LH, nimg), dim3(1024), 0, c->jrCtl, c->own, c->rankOf, c->rec, c->lastSize,
               c->rgClean, c->smallSeeds, c->bigSeeds, c->bigCap, P.LW, P.LH, bigThresh, t, img0);
      }
      TRL(c, "k_rx_grow", k_rx_grow, dim3(growBlocks, nimg), dim3(256), 0, c->dP, c->jrCtl, c->rec, c->own, c->smallSeeds,
             c->lastSize, c->rgBox, c->hand, c->handCap, c->arena, c->arenaCap, c->rects, c->rectCap, img0, t);
      if (nimg <= 4)
        TRL(c, "k_rx_grow_wave", k_rx_grow_wave, dim3(bigBlocks * 4, nimg), dim3(64), 0, c->dP, c->jrCtl, c->rec, c->own, c->bigSeeds,
            c->bigCap, c->hand, c->handCap, c->lastSize, c->rgBox, c->arena, c->arenaCap, c->rects, c->rectCap, img0, t);
      else
        TRL(c, "k_rx_grow_big", k_rx_grow_big, dim3(bigBlocks, nimg), dim3(64), 0, c->dP, c->jrCtl, c->rec, c->own, c->bigSeeds,
            c->bigCap, c->hand, c->handCap, c->lastSize, c->rgBox, c->arena, c->arenaCap, c->rects, c->rectCap, img0, t);
      TRL(c, "k_rx_rect", k_rx_rect, dim3(rectBlocks, nimg), dim3(64), 0, c->dP, c->jrCtl, c->rec, c->arena, c->arenaCap, c->rects,
          c->rectCap, c->rgSeg, img0, c->mg, -1, (const int2*)nullptr, (const float2*)nullptr);
      if (trace) {
        HIPCHK(hipMemcpyAsync(c->jrHost.data(), c->jrCtl + img0, sizeof(RxCtl) * nimg, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const RxCtl& h = c->jrHost[0];
        std::fprintf(stderr, "[rx] t=%d state=%d changed=%d overflow=%d small=%d big=%d hand=%d nextBig=%d rect=%d arena=%lld races=%d\n", t,
                     h.state, h.changed, h.overflow, h.nSmall, h.nBig, h.nHand, h.nextBig, (int)(h.rectArena >> RX_ARENA_BITS),
                     (long long)(h.rectArena & ((1ull << RX_ARENA_BITS) - 1ull)), h.races);
      }
      { pli_status ls_ = look(t); if (ls_ != PLI_OK) return ls_; }
    }
#endif
    }
    if (keyMode) {
      int emitCap = TX_EMIT_CAP;                   // (test switch: a small list, so that images take the overflow path to the sequential grower)
      if (const char* e = DEVENV("PLI_TX_EMITCAP")) emitCap = std::max(1, std::min(TX_EMIT_CAP, atoi(e)));
      TRL(c, "k_tx_collect", k_tx_collect, dim3(std::max(1, std::min(64, (int)((npix64 + 2047) / 2048))), nimg), dim3(256), 0, c->jrCtl,
          (const int*)c->rankOf, (const int2*)c->own, (const int*)c->lastSize, npix64, P.minRegSize, c->txCand, c->txCandCnt, img0, emitCap);
      TRL(c, "k_tx_emit_sorted", k_tx_emit_sorted, dim3(nimg), dim3(1024), TX_EMIT_CAP * 4, (const RxCtl*)c->jrCtl, (const int*)c->txCand,
          (const int*)c->txCandCnt, (const float4*)c->rgSeg, npix64, (1 << c->txPixBits) - 1, c->seg, c->nSeg, c->maxSeg, img0, emitCap);
      // (the ordered lists of the images left to the sequential grower, decided on the device: normally none, and the three passes end at once)
      pli_status os = orderPasses(c->jrCtl);
      if (os != PLI_OK) return os;
    } else {
    TRL(c, "k_rx_count", k_rx_count, dim3(c->rxChunks, nimg), dim3(256), 0, c->jrCtl, c->order, c->nDefined, c->own, c->lastSize,
           c->rxChunkCnt, c->rxChunks, npix64, P.minRegSize, img0);
    TRL(c, "k_rx_emit", k_rx_emit, dim3(c->rxChunks, nimg), dim3(256), 0, c->jrCtl, c->order, c->nDefined, c->own, c->lastSize,
           c->rgSeg, c->rxChunkCnt, c->rxChunks, npix64, P.minRegSize, c->seg, c->nSeg, c->maxSeg, img0);
    }
    // images that ran out of a capacity (or did not settle) take the sequential grower
    // (... which reads the 16-byte records: with every round on the hot records nobody has written them — for those images, now)
    if (hotAll)
      LAUNCH(c, "k_tx_rec_from_hot", k_tx_rec_from_hot, dim3(std::max(1, std::min(64, (int)((npix64 + 4095) / 4096))), nimg), dim3(256), 0,
             (const RxCtl*)c->jrCtl, (const int2*)c->hot, (const float2*)c->cold, c->rec, npix64, img0);
    if (blocking && !keyMode) {
      for (int i = 0; i < nimg; ++i) {
        if (c->jrHost[i].overflow || c->jrHost[i].state != 2) {
          LAUNCH(c, "k_lsd_grow", k_lsd_grow, dim3(1), dim3(64), 0, c->dP, c->rec, c->mg, c->order, c->nDefined,
                 c->regScratch, c->seg, c->nSeg, c->maxSeg, img0 + i, 1);
        }
      }
    } else {
      // ... decided on the device: the waves of the settled images leave at once
      LAUNCH(c, "k_lsd_grow_unsettled", k_lsd_grow_unsettled, dim3(nimg), dim3(64), 0, c->dP, c->rec, c->mg, c->order, c->nDefined,
             c->regScratch, c->seg, c->nSeg, c->maxSeg, img0, nimg, c->jrCtl);
      if (!blocking) {
      if (!c->rxSeen) {
        HIPCHK(hipHostMalloc((void**)&c->rxSeen, sizeof(int) * (4 * (size_t)c->NI + 1), hipHostMallocDefault));     // (+ the tail's abort word)
        c->rxSeen[4 * (size_t)c->NI] = 0;
        HIPCHK(hipEventCreateWithFlags(&c->evRxSeen, hipEventDisableTiming));
      }
      if (c->rxSeenImages == 0) {                       // (an earlier copy still in flight keeps the buffer: this call is not sampled)
        HIPCHK(hipMemcpy2DAsync(c->rxSeen, 16, c->jrCtl + img0, sizeof(RxCtl), 16, nimg, hipMemcpyDeviceToHost, c->stream));
        if (c->tailBar && c->tailLaunched)
          HIPCHK(hipMemcpyAsync(c->rxSeen + 4 * (size_t)c->NI, c->tailBar + 32, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        else c->rxSeen[4 * (size_t)c->NI] = 0;
        HIPCHK(hipEventRecord(c->evRxSeen, c->stream));
        c->rxSeenImages = nimg;
        c->rxSeenTail = c->tailLaunched;
      }
      }
    }
  }
  LAUNCH(c, "k_keylines", k_keylines, dim3(nimg), dim3(256), 0, c->dP, c->seg, c->nSeg, c->maxSeg, c->tmpKL, table,
         Y.record_bytes, Y.off_counts, Y.off_kl[0], Y.off_kl[1], img0);
  if (c->lbdPreOnSide) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->evLbdPre, 0));   // (the blur + Sobel of the LBD ran on the side stream, see pli_batch_run)
  } else {
    pli_status ls = runLbdPre(c, img0, nimg);
    if (ls != PLI_OK) return ls;
  }
  LAUNCH(c, "k_lbd", k_lbd, dim3(P.klCap, nimg), dim3(64), 0, c->dP, c->lbdCoef, c->dxy, table, Y.record_bytes,
         Y.off_counts, Y.off_kl[0], Y.off_kl[1], Y.off_ldesc[0], Y.off_ldesc[1], c->debug ? c->lbdFloat : (float*)nullptr, img0);
  return PLI_OK;
}

pli_status runStereoPoints(pli_ctx* c, int nframes, uint8_t* table) {
  TraceRange range__("pli:stereo points");
  const DevParams& P = c->hp;
  const pli_table_layout& Y = c->lay;
  LAUNCH(c, "k_stereo_points", k_stereo_points, dim3(P.kpCap, nframes), dim3(64), 0, c->dP, c->pyr, table, Y.record_bytes,
         Y.off_counts, Y.off_kp[0], Y.off_kp[1], Y.off_desc[0], Y.off_desc[1], Y.off_uright, Y.off_depth, c->sad,
         c->debug ? c->bestIdx : (int*)nullptr);
  LAUNCH(c, "k_stereo_median", k_stereo_median, dim3(nframes), dim3(256), (size_t)P.kpCap * 4, c->dP, table, Y.record_bytes,
         Y.off_counts, Y.off_uright, Y.off_depth, c->sad);
  return PLI_OK;
}

pli_status runStereoLines(pli_ctx* c, int nframes, uint8_t* table) {
  TraceRange range__("pli:stereo lines");
  const pli_table_layout& Y = c->lay;
  // (many lines per frame: the pair distances by several workgroups per frame, see the kernel)
  const int cap = c->hp.klCap;
  const int slices = (cap >= 192 && !DEVENV("PLI_STEREO_LINES_1")) ? std::max(1, std::min(64, (cap * cap) / 8192)) : 1;
  for (int phase = slices > 1 ? 1 : 0; phase <= (slices > 1 ? 3 : 0); ++phase)
    LAUNCH(c, "k_stereo_lines", k_stereo_lines, dim3(nframes, phase == 2 ? slices : 1), dim3(256), 0, c->dP, table, Y.record_bytes,
           Y.off_counts, Y.off_kl[0], Y.off_kl[1], Y.off_ldesc[0], Y.off_ldesc[1], Y.off_disp, Y.off_le, c->lmask, c->ldir, c->dmat,
           c->m12, c->m21, phase);
  return PLI_OK;
}

pli_status ensureScratch(pli_ctx* c, size_t bytes) {
  if (bytes <= c->scratchBytes) return PLI_OK;
  if (c->scratch) hipFree(c->scratch);
  c->scratch = nullptr; c->scratchBytes = 0;
  HIPCHK(hipMalloc(&c->scratch, bytes));
  c->scratchBytes = bytes;
  return PLI_OK;
}

pli_status checkImage(pli_ctx* c, const uint8_t* img, int w, int h, int64_t stride) {
  if (!img || w <= 0 || h <= 0) { g_err = "empty image"; return PLI_ERR_EMPTY_IMAGE; }
  if (w != c->cfg.width || h != c->cfg.height) { g_err = "image size differs from the context's"; return PLI_ERR_INVALID; }
  if (stride < w) { g_err = "stride < width"; return PLI_ERR_INVALID; }
  return PLI_OK;
}

pli_status stageImage(pli_ctx* c, int eye, const uint8_t* img, int w, int h, int64_t stride) {
  HIPCHK(hipMemcpy2DAsync(c->inStage[eye], w, img, stride, w, h, hipMemcpyHostToDevice, c->stream));
  return runIngest(c, c->inStage[0], c->inStage[1], w, 0, eye, 1);
}

}  // namespace

extern "C" {

const char* pli_version(void) { return "pli-frontend-mi355x 0.1 (gfx950)"; }
const char* pli_last_error(void) { return g_err.c_str(); }

void pli_config_default(pli_frontend_config* c, int32_t width, int32_t height) {
  std::memset(c, 0, sizeof(*c));
  c->width = width; c->height = height; c->max_frames = 1;
  c->orb_nfeatures = 1200; c->orb_scale_factor = 1.2f; c->orb_nlevels = 8;
  c->orb_ini_th_fast = 20; c->orb_min_th_fast = 7;
  c->lsd_nfeatures = 500; c->lsd_refine = 0; c->lsd_n_bins = 1024;
  c->max_lines = std::max(4096, width * height / 64);   // the reference keeps every segment above the length cut
  c->min_line_length = 0.025; c->lsd_scale = 1.2; c->lsd_sigma_scale = 0.6; c->lsd_quant = 2.0;
  c->lsd_ang_th = 22.5; c->lsd_log_eps = 1.0; c->lsd_density_th = 0.6;
  c->bf = 47.90639384423901f; c->fx = 435.2046959714599f; c->stereo_maxd_inf = 0;
  c->matching_s_ws = 10; c->best_lr_matches = 1;
  c->line_sim_th = 0.75; c->stereo_overlap_th = 0.75; c->min_ratio_12_l = 0.9;
  c->ls_min_disp_ratio = 0.7; c->min_disp = 1.0; c->line_horiz_th = 0.1;
  c->lsd_mode = 0;
  c->parity_flags = PLI_PARITY_TRIG_F32_ORB | PLI_PARITY_LSD_F64;
}

int32_t pli_kp_capacity(const pli_frontend_config* c) {
  // DistributeOctTree returns up to quota+3 keypoints per level (ORBextractor.cc:713-733) — or, when the quota is tiny,
  // the 4 children of each of its nIni = round(width/height) root nodes (:540-589, first expansion)
  int64_t extra = 0;
  float sc = 1.0f;
  for (int l = 0; l < c->orb_nlevels; ++l) {
    const float inv = 1.0f / sc;
    const int w = cvRoundf((float)c->width * inv), h = cvRoundf((float)c->height * inv);
    const int bw = w - 32, bh = h - 32;                     // maxBorder - minBorder, EDGE_THRESHOLD - 3 = 16 each side
    const int nIni = (bw > 0 && bh > 0) ? (int)std::round((float)bw / (float)bh) : 0;
    extra += std::max(4, 4 * nIni);
    sc *= c->orb_scale_factor;
  }
  return (int32_t)alignUp((int64_t)c->orb_nfeatures + extra + 8, 64);
}
int32_t pli_kl_capacity(const pli_frontend_config* c) { return c->lsd_nfeatures != 0 ? c->lsd_nfeatures : c->max_lines; }

pli_status pli_ctx_create(const pli_frontend_config* cfg, int32_t device, pli_ctx** out) {
  if (!cfg || !out) { g_err = "null argument"; return PLI_ERR_INVALID; }
  *out = nullptr;
  pli_status st = validate(*cfg);
  if (st != PLI_OK) return st;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    g_err = "no HIP device: this library has no CPU path";
    return PLI_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    g_err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return PLI_ERR_NO_DEVICE;
  }
  HIPCHK(hipSetDevice(device));
  std::unique_ptr<pli_ctx> c(new pli_ctx());
  c->cfg = *cfg;
  c->device = device;
  c->NI = 2 * cfg->max_frames;
  st = buildGeometry(c.get());
  if (st != PLI_OK) return st;
  buildLayout(c.get());
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  st = allocAll(c.get());
  if (st != PLI_OK) { pli_ctx_destroy(c.release()); return st; }
  *out = c.release();
  return PLI_OK;
}

void pli_ctx_destroy(pli_ctx* c) {
  if (!c) return;
  { CtxGuard wait__(c); }       // a call still running on another thread finishes first (destroying under it is the caller's bug)
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  for (void* p : c->allocs) hipFree(p);
  if (c->scratch) hipFree(c->scratch);
  for (hipEvent_t e : c->evPool) hipEventDestroy(e);
  if (c->ownStream && c->stream) hipStreamDestroy(c->stream);
  if (c->aux2) { hipStreamSynchronize(c->aux2); hipStreamDestroy(c->aux2); }
  if (c->evRectFork) { hipEventDestroy(c->evRectFork); hipEventDestroy(c->evRectDone); }
  if (c->aux) { hipStreamSynchronize(c->aux); hipStreamDestroy(c->aux); hipEventDestroy(c->evFork); hipEventDestroy(c->evJoin); hipEventDestroy(c->evLbdPre); }
  for (int e = 0; e < 2; ++e) if (c->rectMap[e]) hipFree(c->rectMap[e]);
  for (int s = 0; s < 2; ++s) {
    if (c->hs[s].busy) hipEventSynchronize(c->hs[s].d2h);
    if (c->hs[s].dimg) hipFree(c->hs[s].dimg);
    if (c->hs[s].dtab) hipFree(c->hs[s].dtab);
    if (c->hs[s].h2d) { hipEventDestroy(c->hs[s].h2d); hipEventDestroy(c->hs[s].kern); hipEventDestroy(c->hs[s].d2h); }
  }
  if (c->sH2D) { hipStreamDestroy(c->sH2D); hipStreamDestroy(c->sD2H); }
  if (c->rxSeen) { hipHostFree(c->rxSeen); hipEventDestroy(c->evRxSeen); }
  for (int e = 0; e < 2; ++e) if (c->pyrHost[e]) hipHostFree(c->pyrHost[e]);
  if (c->recPinned) hipHostFree(c->recPinned);
  delete c;
}

pli_status pli_ctx_layout(const pli_ctx* c, pli_table_layout* out) {
  CtxGuard guard__(c);
  if (!c || !out) return PLI_ERR_INVALID;
  *out = c->lay;
  return PLI_OK;
}

pli_status pli_ctx_set_stream(pli_ctx* c, void* s) {
  CtxGuard guard__(c);
  if (!c) return PLI_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (s) {
    if (c->ownStream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;
    c->ownStream = false;
  } else if (!c->ownStream) {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->ownStream = true;
  }
  return PLI_OK;
}

pli_status pli_set_rectify_maps(pli_ctx* c, int32_t eye, const float* mapx, const float* mapy) {
  CtxGuard guard__(c);
  if (!c || eye < 0 || eye > 1 || ((mapx == nullptr) != (mapy == nullptr))) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (!mapx) {
    if (c->rectMap[eye]) { hipFree(c->rectMap[eye]); c->rectMap[eye] = nullptr; }
    return PLI_OK;
  }
  const size_t n = (size_t)c->cfg.width * c->cfg.height;
  if (!c->rectMap[eye]) HIPCHK(hipMalloc(&c->rectMap[eye], n * sizeof(float2)));
  std::vector<float2> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = make_float2(mapx[i], mapy[i]);
  HIPCHK(hipMemcpy(c->rectMap[eye], h.data(), n * sizeof(float2), hipMemcpyHostToDevice));
  return PLI_OK;
}

pli_status pli_ctx_sync(pli_ctx* c) {
  CtxGuard guard__(c);
  if (!c) return PLI_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(c->stream));
  return PLI_OK;
}

pli_status pli_batch_run(pli_ctx* c, int32_t nframes, const uint8_t* dl, const uint8_t* dr, int64_t stride,
                         int64_t frameStride, uint32_t stages, void* table) {
  CtxGuard guard__(c);
  TraceRange range__("pli_batch_run");
  if (!c || !dl || !dr || !table) { g_err = "null argument"; return PLI_ERR_INVALID; }
  if (nframes < 1 || nframes > c->cfg.max_frames) { g_err = "nframes exceeds the context's max_frames"; return PLI_ERR_INVALID; }
  if (stride < c->cfg.width) { g_err = "stride < width"; return PLI_ERR_INVALID; }
  if (((uintptr_t)table & 15) != 0) { g_err = "the table must be 16-byte aligned"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  uint8_t* T = (uint8_t*)table;
  pli_status st;
  const int nimg = 2 * nframes;
  // a batch overwrites the pyramids and tables of images 0/1: whatever the per-call entry points cached about the last Frame
  // on this context is gone (pli_frame_extract, which calls this, sets its own state afterwards)
  c->frameFresh = c->pointsFresh = false;
  for (int e = 0; e < 2; ++e) { c->orbDone[e] = c->lineDone[e] = false; c->pyrHostValid[e] = false; }
  if ((st = runIngest(c, dl, dr, stride, frameStride, 0, nimg)) != PLI_OK) return st;
  bool stereoPointsDone = false;
  // The ORB chain runs beside the line chain (fork after the ingest, join before the stereo matchers) except under the
  // sequential grower above 2560 images.  Measured, 752x480, frames/s without -> with: tile relaxation 32 frames 2688 -> 2847,
  // 128 frames 3711 -> 3947, 256 frames 4054 -> 4245, 512 frames 4268 -> 4379, 768 frames 4303 -> 4420; sequential grower
  // 1024 frames 4213 -> 4544, 1280 frames 4830 -> 5266; from 1536 frames on the grower's blocks fill the CUs' LDS, the ORB
  // kernels wait for it and stretch the grower: 5170 -> 4287, 2048 frames 6067 -> 4942.
  static const int sideMax = DEVENV("PLI_SIDE_MAX") ? atoi(DEVENV("PLI_SIDE_MAX")) : INT_MAX;    // (dev: images below which the tile relaxation has the ORB chain beside it)
  static const int sideSeqMax = DEVENV("PLI_SIDE_SEQ_MAX") ? atoi(DEVENV("PLI_SIDE_SEQ_MAX")) : 2560;
  const bool seqGrower = c->lsdMode == 2 || (c->lsdMode == 0 && (nimg >= RX_AUTO_IMAGES || nimg > c->rxImages));
  if ((seqGrower ? nimg <= sideSeqMax : nimg < sideMax) && (stages & PLI_RUN_ORB) && (stages & PLI_RUN_LINES) && !c->syncDebug) {
    if (!c->aux) {
      // (the line chain is the longer one: the ORB chain beside it takes what the line kernels leave free)
      int prLow = 0, prHigh = 0;
      HIPCHK(hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
      if (DEVENV("PLI_SIDE_NOPRIO")) prLow = 0;            // dev switch: default priority
      HIPCHK(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, prLow));
      HIPCHK(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->evLbdPre, hipEventDisableTiming));
    }
    hipStream_t main = c->stream;
    static const bool sideStereo = DEVENV("PLI_SIDE_NOSTEREO") == nullptr;      // (dev switch)
    static const bool sideLbd = DEVENV("PLI_SIDE_NOLBD") == nullptr;            // (dev switch)
    static const bool sideDefer = DEVENV("PLI_SIDE_NODEFER") == nullptr;        // (dev switch)
    stereoPointsDone = (stages & PLI_RUN_STEREO_POINTS) && sideStereo;
    c->lbdPreOnSide = c->lsdF64 && sideLbd;
    // the side chain: forks from the line chain where it is enqueued.  Under the tile relaxation that is right after the
    // launch of round 1 (runLines calls it): round 1 keeps the chip to itself, and the later rounds, which leave most of it
    // idle, host the ORB kernels; otherwise before the line chain starts.
    auto sideChain = [c, nimg, nframes, T, stages, main]() -> pli_status {
      HIPCHK(hipEventRecord(c->evFork, main));
      HIPCHK(hipStreamWaitEvent(c->aux, c->evFork, 0));
      c->stream = c->aux;
      pli_status s2 = runOrb(c, 0, nimg, T);
      // (the stereo point matcher needs the ORB tables only: it stays on the side stream, off the line chain's path)
      if (s2 == PLI_OK && (stages & PLI_RUN_STEREO_POINTS) && sideStereo) s2 = runStereoPoints(c, nframes, T);
      if (s2 == PLI_OK && c->lbdPreOnSide) {
        s2 = runLbdPre(c, 0, nimg);
        if (s2 == PLI_OK) { hipError_t e_ = hipEventRecord(c->evLbdPre, c->aux); if (e_ != hipSuccess) s2 = PLI_ERR_HIP; }
      }
      c->stream = main;
      if (s2 != PLI_OK) return s2;
      HIPCHK(hipEventRecord(c->evJoin, c->aux));
      return PLI_OK;
    };
    // (measured: +1.3 % at 32 frames, +1.4 % on the 4K configuration (16 frames), +1 % on a single pair; neutral at 64 and 128
    // frames, -0.5 % at 256, -3 % on the 720p configuration with its 2000 keypoints: up to 64 images it is)
    // Round 5, with round 1 a tenth shorter when it has the chip to itself: 256 frames of 752 x 480 +1.9 % (47.4 -> 46.6 ms), 128 frames +2 %,
    // 512 frames +0.3 %, 1024 frames -0.4 %; 64 frames of 1280 x 720 (2000 keypoints: a longer ORB chain) still -1.3 %.  So: up to 64
    // images always, up to 1024 images of EuRoC-sized frames (< 0.5 M pixels).  Behind a LATER round (PLI_SIDE_FORK_ROUND = 3 / 4 / 5 / 7)
    // it is 47.4 / 47.5 / 47.9 / 48.4 ms on the synthetic stream; which round it forks behind: runLines (sideForkRound).
    const int sideDeferMax = DEVENV("PLI_SIDE_DEFER_MAX") ? atoi(DEVENV("PLI_SIDE_DEFER_MAX")) : -1;   // (dev: images up to which the ORB chain starts behind round 1)
    const bool deferSide = sideDeferMax >= 0 ? nimg <= sideDeferMax : (nimg <= 64 || (nimg <= 1024 && (int64_t)c->hp.W * c->hp.H < 500000));
    if (sideDefer && !seqGrower && c->lsdMode != 1 && deferSide) c->sideChain = sideChain;
    else if ((st = sideChain()) != PLI_OK) { c->lbdPreOnSide = false; return st; }
    st = runLines(c, 0, nimg, T);
    if (st == PLI_OK && c->sideChain) { auto f = std::move(c->sideChain); c->sideChain = nullptr; st = f(); }   // (not taken by runLines)
    c->sideChain = nullptr;
    c->lbdPreOnSide = false;
    if (st != PLI_OK) return st;
    HIPCHK(hipStreamWaitEvent(main, c->evJoin, 0));
  } else {
    if (stages & PLI_RUN_ORB) if ((st = runOrb(c, 0, nimg, T)) != PLI_OK) return st;
    if (stages & PLI_RUN_LINES) if ((st = runLines(c, 0, nimg, T)) != PLI_OK) return st;
  }
  if (stages & PLI_RUN_STEREO_LINES) if ((st = runStereoLines(c, nframes, T)) != PLI_OK) return st;
  if ((stages & PLI_RUN_STEREO_POINTS) && !stereoPointsDone) if ((st = runStereoPoints(c, nframes, T)) != PLI_OK) return st;
  return PLI_OK;
}

pli_status pli_batch_run_host(pli_ctx* c, int32_t nframes, const uint8_t* left, const uint8_t* right, int64_t stride,
                              int64_t frameStride, uint32_t stages, void* table) {
  CtxGuard guard__(c);
  TraceRange range__("pli_batch_run_host");
  if (!c || !left || !right || !table) { g_err = "null argument"; return PLI_ERR_INVALID; }
  if (nframes < 1 || nframes > c->cfg.max_frames) { g_err = "nframes exceeds the context's max_frames"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  const int W = c->cfg.width, H = c->cfg.height;
  const size_t imgBytes = (size_t)W * H;
  const size_t imgsBytes = alignUp(2 * imgBytes * nframes, 256);     // the table needs its natural alignment (odd image sizes!)
  pli_status st = ensureScratch(c, imgsBytes + (size_t)c->lay.record_bytes * nframes);
  if (st != PLI_OK) return st;
  uint8_t* dl = (uint8_t*)c->scratch;
  uint8_t* dr = dl + imgBytes * nframes;
  uint8_t* dt = dl + imgsBytes;
  for (int f = 0; f < nframes; ++f) {
    HIPCHK(hipMemcpy2DAsync(dl + imgBytes * f, W, left + (int64_t)f * frameStride, stride, W, H, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpy2DAsync(dr + imgBytes * f, W, right + (int64_t)f * frameStride, stride, W, H, hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(hipMemsetAsync(dt, 0, (size_t)c->lay.record_bytes * nframes, c->stream));
  st = pli_batch_run(c, nframes, dl, dr, W, (int64_t)imgBytes, stages, dt);
  if (st != PLI_OK) return st;
  HIPCHK(hipMemcpyAsync(table, dt, (size_t)c->lay.record_bytes * nframes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return PLI_OK;
}

pli_status pli_frame_extract(pli_ctx* c, const uint8_t* left, const uint8_t* right, int32_t w, int32_t h, int64_t strideLeft,
                             int64_t strideRight, void* record) {
  CtxGuard guard__(c);
  TraceRange range__("pli_frame_extract");
  if (!c || !record) { g_err = "null argument"; return PLI_ERR_INVALID; }
  pli_status st = checkImage(c, left, w, h, strideLeft);
  if (st != PLI_OK) return st;
  if ((st = checkImage(c, right, w, h, strideRight)) != PLI_OK) return st;
  HIPCHK(hipSetDevice(c->device));
  const int W = c->cfg.width, H = c->cfg.height;
  HIPCHK(hipMemcpy2DAsync(c->inStage[0], W, left, strideLeft, W, H, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpy2DAsync(c->inStage[1], W, right, strideRight, W, H, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(c->ownTable, 0, (size_t)c->lay.record_bytes, c->stream));
  c->frameFresh = false;
  if ((st = pli_batch_run(c, 1, c->inStage[0], c->inStage[1], W, 0, PLI_RUN_ALL, c->ownTable)) != PLI_OK) return st;
  if (!c->recPinned) HIPCHK(hipHostMalloc((void**)&c->recPinned, (size_t)c->lay.record_bytes, hipHostMallocDefault));
  HIPCHK(hipMemcpyAsync(c->recPinned, c->ownTable, (size_t)c->lay.record_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::memcpy(c->hostRec.data(), c->recPinned, (size_t)c->lay.record_bytes);
  std::memcpy(record, c->recPinned, (size_t)c->lay.record_bytes);
  const int32_t* counts = reinterpret_cast<const int32_t*>(c->hostRec.data() + c->lay.off_counts);
  for (int e = 0; e < 2; ++e) {
    c->orbDone[e] = c->lineDone[e] = true;
    c->orbCount[e] = c->monoCount[e] = counts[e];
    c->lineCount[e] = counts[2 + e];
    c->pyrHostValid[e] = false;
  }
  const uint8_t* flags = reinterpret_cast<const uint8_t*>(counts + 6);
  if (flags[0] || flags[1]) { g_err = "more segments pass the length cut than max_lines holds: raise pli_frontend_config.max_lines"; return PLI_ERR_CAPACITY; }
  if (flags[2] || flags[3]) { g_err = "more keypoints than kp_cap holds"; return PLI_ERR_CAPACITY; }
  c->frameFresh = true;
  c->pointsFresh = true;
  return PLI_OK;
}

pli_status pli_host_alloc(size_t bytes, void** out) {
  if (!out) { g_err = "null argument"; return PLI_ERR_INVALID; }
  *out = nullptr;
  HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return PLI_OK;
}
void pli_host_free(void* p) { if (p) (void)hipHostFree(p); }

static pli_status waitSlot(pli_ctx* c, int s) {
  pli_ctx::HostSlot& S = c->hs[s];
  if (S.busy) { HIPCHK(hipEventSynchronize(S.d2h)); S.busy = false; ++c->hsWaited; }
  return PLI_OK;
}

pli_status pli_batch_wait(pli_ctx* c, int32_t all) {
  CtxGuard guard__(c);
  if (!c) return PLI_ERR_INVALID;
  HIPCHK(hipSetDevice(c->device));
  pli_status st;
  while (c->hsWaited < c->hsSubmitted) {
    if ((st = waitSlot(c, (int)(c->hsWaited & 1))) != PLI_OK) return st;
    if (!all) break;
  }
  return PLI_OK;
}

pli_status pli_batch_submit_host(pli_ctx* c, int32_t nframes, const uint8_t* left, const uint8_t* right, int64_t stride,
                                 int64_t frameStride, uint32_t stages, void* table) {
  CtxGuard guard__(c);
  TraceRange range__("pli_batch_submit_host");
  if (!c || !left || !right || !table) { g_err = "null argument"; return PLI_ERR_INVALID; }
  if (nframes < 1 || nframes > c->cfg.max_frames) { g_err = "nframes exceeds the context's max_frames"; return PLI_ERR_INVALID; }
  if (stride < c->cfg.width) { g_err = "stride < width"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  const int W = c->cfg.width, H = c->cfg.height;
  const size_t imgBytes = (size_t)W * H, tabBytes = (size_t)c->lay.record_bytes * nframes;
  if (!c->sH2D) {
    HIPCHK(hipStreamCreateWithFlags(&c->sH2D, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->sD2H, hipStreamNonBlocking));
  }
  const int s = (int)(c->hsSubmitted & 1);
  pli_ctx::HostSlot& S = c->hs[s];
  pli_status st = waitSlot(c, s);                     // the slot's previous batch (two submits ago) must have left
  if (st != PLI_OK) return st;
  if (!S.h2d) {
    HIPCHK(hipEventCreateWithFlags(&S.h2d, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&S.kern, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&S.d2h, hipEventDisableTiming));
  }
  const size_t needImg = alignUp(2 * imgBytes * (size_t)c->cfg.max_frames, 256), needTab = (size_t)c->lay.record_bytes * c->cfg.max_frames;
  if (S.imgBytes < needImg) { if (S.dimg) hipFree(S.dimg); S.dimg = nullptr; HIPCHK(hipMalloc(&S.dimg, needImg)); S.imgBytes = needImg; }
  if (S.tabBytes < needTab) { if (S.dtab) hipFree(S.dtab); S.dtab = nullptr; HIPCHK(hipMalloc(&S.dtab, needTab)); S.tabBytes = needTab; }
  uint8_t* dl = S.dimg;
  uint8_t* dr = S.dimg + imgBytes * nframes;
  if (stride == W && frameStride == (int64_t)imgBytes) {
    HIPCHK(hipMemcpyAsync(dl, left, imgBytes * nframes, hipMemcpyHostToDevice, c->sH2D));
    HIPCHK(hipMemcpyAsync(dr, right, imgBytes * nframes, hipMemcpyHostToDevice, c->sH2D));
  } else {
    for (int f = 0; f < nframes; ++f) {
      HIPCHK(hipMemcpy2DAsync(dl + imgBytes * f, W, left + (int64_t)f * frameStride, stride, W, H, hipMemcpyHostToDevice, c->sH2D));
      HIPCHK(hipMemcpy2DAsync(dr + imgBytes * f, W, right + (int64_t)f * frameStride, stride, W, H, hipMemcpyHostToDevice, c->sH2D));
    }
  }
  HIPCHK(hipEventRecord(S.h2d, c->sH2D));
  HIPCHK(hipStreamWaitEvent(c->stream, S.h2d, 0));
  HIPCHK(hipMemsetAsync(S.dtab, 0, tabBytes, c->stream));
  st = pli_batch_run(c, nframes, dl, dr, W, (int64_t)imgBytes, stages, S.dtab);
  if (st != PLI_OK) return st;
  HIPCHK(hipEventRecord(S.kern, c->stream));
  HIPCHK(hipStreamWaitEvent(c->sD2H, S.kern, 0));
  HIPCHK(hipMemcpyAsync(table, S.dtab, tabBytes, hipMemcpyDeviceToHost, c->sD2H));
  HIPCHK(hipEventRecord(S.d2h, c->sD2H));
  S.busy = true;
  ++c->hsSubmitted;
  return PLI_OK;
}

pli_status pli_orb_extract(pli_ctx* c, int32_t eye, const uint8_t* img, int32_t w, int32_t h, int64_t stride,
                           pli_keypoint* kp, int32_t cap, uint8_t* desc, int32_t* n) {
  CtxGuard guard__(c);
  TraceRange range__("pli_orb_extract");
  if (!c || eye < 0 || eye > 1 || !n) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  *n = 0;
  pli_status st = checkImage(c, img, w, h, stride);
  if (st != PLI_OK) return st;
  HIPCHK(hipSetDevice(c->device));
  if ((st = stageImage(c, eye, img, w, h, stride)) != PLI_OK) return st;
  if ((st = runOrb(c, eye, 1, c->ownTable)) != PLI_OK) return st;
  const pli_table_layout& Y = c->lay;
  int counts[8];
  HIPCHK(hipMemcpyAsync(counts, c->ownTable + Y.off_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int N = counts[eye];
  c->orbDone[eye] = true;
  c->frameFresh = false;
  c->pyrHostValid[eye] = false;
  c->orbCount[eye] = N;
  c->monoCount[eye] = N;
  *n = N;
  if (reinterpret_cast<const uint8_t*>(counts + 6)[2 + eye]) { g_err = "more keypoints than kp_cap holds"; return PLI_ERR_CAPACITY; }
  if (N > cap) { g_err = "keypoint buffer too small"; return PLI_ERR_CAPACITY; }
  if (N > 0) {
    if (kp) HIPCHK(hipMemcpy(kp, c->ownTable + Y.off_kp[eye], (size_t)N * sizeof(pli_keypoint), hipMemcpyDeviceToHost));
    if (desc) HIPCHK(hipMemcpy(desc, c->ownTable + Y.off_desc[eye], (size_t)N * 32, hipMemcpyDeviceToHost));
  }
  return PLI_OK;
}

pli_status pli_orb_pyramid_level(pli_ctx* c, int32_t eye, int32_t level, uint8_t* dst, int64_t dstBytes, int32_t* w, int32_t* h) {
  CtxGuard guard__(c);
  if (!c || eye < 0 || eye > 1 || level < 0 || level >= c->hp.nlevels) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (!c->orbDone[eye]) { g_err = "pli_orb_extract has not run for this eye"; return PLI_ERR_STATE; }
  const LevelGeom& G = c->hp.lv[level];
  if (w) *w = G.w;
  if (h) *h = G.h;
  if (!dst) return PLI_OK;
  if (dstBytes < (int64_t)G.w * G.h) { g_err = "buffer too small"; return PLI_ERR_CAPACITY; }
  // One device-to-host copy of the eye's whole pyramid block into pinned memory per extraction (1.3 MB at 752x480), then the levels
  // are de-pitched on the host: eight pitched hipMemcpy2D calls into pageable memory cost 0.8 ms EACH (13 ms per extractor call of
  // the adapters, which fill the public member mvImagePyramid).
  if (!c->pyrHost[eye]) HIPCHK(hipHostMalloc((void**)&c->pyrHost[eye], (size_t)c->hp.pyrBlock, hipHostMallocDefault));
  if (!c->pyrHostValid[eye]) {
    HIPCHK(hipMemcpyAsync(c->pyrHost[eye], c->pyr + (int64_t)eye * c->hp.pyrBlock, (size_t)c->hp.pyrBlock, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->pyrHostValid[eye] = true;
  }
  const uint8_t* src = c->pyrHost[eye] + G.offset;
  for (int y = 0; y < G.h; ++y) std::memcpy(dst + (size_t)y * G.w, src + (size_t)y * G.pitch, (size_t)G.w);
  return PLI_OK;
}

pli_status pli_line_extract(pli_ctx* c, int32_t eye, const uint8_t* img, int32_t w, int32_t h, int64_t stride,
                            pli_keyline* kl, int32_t cap, uint8_t* desc, int32_t* n) {
  CtxGuard guard__(c);
  TraceRange range__("pli_line_extract");
  if (!c || eye < 0 || eye > 1 || !n) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  *n = 0;
  pli_status st = checkImage(c, img, w, h, stride);
  if (st != PLI_OK) return st;
  HIPCHK(hipSetDevice(c->device));
  if ((st = stageImage(c, eye, img, w, h, stride)) != PLI_OK) return st;
  if ((st = runLines(c, eye, 1, c->ownTable)) != PLI_OK) return st;
  const pli_table_layout& Y = c->lay;
  int counts[8];
  HIPCHK(hipMemcpyAsync(counts, c->ownTable + Y.off_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int N = counts[2 + eye];
  c->lineDone[eye] = true;
  c->frameFresh = false;
  c->lineCount[eye] = N;
  *n = N;
  if (reinterpret_cast<const uint8_t*>(counts + 6)[eye]) {
    g_err = "more segments pass the length cut than max_lines holds: raise pli_frontend_config.max_lines";
    return PLI_ERR_CAPACITY;
  }
  if (N > cap) { g_err = "keyline buffer too small"; return PLI_ERR_CAPACITY; }
  if (N > 0) {
    if (kl) HIPCHK(hipMemcpy(kl, c->ownTable + Y.off_kl[eye], (size_t)N * sizeof(pli_keyline), hipMemcpyDeviceToHost));
    if (desc) HIPCHK(hipMemcpy(desc, c->ownTable + Y.off_ldesc[eye], (size_t)N * 32, hipMemcpyDeviceToHost));
  }
  return PLI_OK;
}

pli_status pli_selftest_hot_trig(pli_ctx* c, double* max_abs_err) {
  CtxGuard guard__(c);
  if (!c || !max_abs_err) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  pli_status st = ensureScratch(c, 256);
  if (st != PLI_OK) return st;
  HIPCHK(hipMemsetAsync(c->scratch, 0, 8, c->stream));
  LAUNCH(c, "k_tx_hot_trig_err", k_tx_hot_trig_err, dim3(4096), dim3(256), 0, (unsigned long long*)c->scratch);
  unsigned long long bits = 0;
  HIPCHK(hipMemcpyAsync(&bits, c->scratch, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::memcpy(max_abs_err, &bits, 8);
  return PLI_OK;
}

pli_status pli_lsd_round_stats(pli_ctx* c, int32_t out[4]) {
  CtxGuard guard__(c);
  if (!c || !out) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->rxSeenImages > 0 && hipEventQuery(c->evRxSeen) == hipSuccess) out[1] = foldRoundStats(c);   // (as the next call would)
  else out[1] = c->rxLastRounds;
  out[0] = c->rxPlanned;
  out[2] = (int32_t)std::min<int64_t>(c->rxSlowImages, INT32_MAX);
  out[3] = c->rxLastRounds;
  return PLI_OK;
}

pli_status pli_lsd_arena_words(pli_ctx* c, int32_t* words_per_pixel) {
  CtxGuard guard__(c);
  if (!c || !words_per_pixel) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  *words_per_pixel = c->arenaFactor;
  return PLI_OK;
}

pli_status pli_set_stereo_camera(pli_ctx* c, float bf, float fx) {
  CtxGuard guard__(c);
  if (!c || !(bf > 0) || !(fx > 0)) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (bf == c->cfg.bf && fx == c->cfg.fx) return PLI_OK;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));           // kernels in flight read the old parameter block
  if (c->aux) HIPCHK(hipStreamSynchronize(c->aux));
  c->cfg.bf = bf; c->cfg.fx = fx;
  c->pointsFresh = false;                            // a fused Frame's cached mvuRight / mvDepth belong to the old rig (Frame.cc:1005-1008,
                                                     // 1131: maxD and bf / disparity); the line matcher does not read the rig
  c->hp.bf = bf;
  c->hp.maxD = c->cfg.stereo_maxd_inf ? std::numeric_limits<float>::infinity() : bf / (bf / fx);
  HIPCHK(hipMemcpy(c->dP, &c->hp, sizeof(DevParams), hipMemcpyHostToDevice));
  return PLI_OK;
}

pli_status pli_last_counts(pli_ctx* c, int32_t counts[4]) {
  CtxGuard guard__(c);
  if (!c || !counts) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  for (int e = 0; e < 2; ++e) {
    counts[e] = c->orbDone[e] ? c->orbCount[e] : -1;
    counts[2 + e] = c->lineDone[e] ? c->lineCount[e] : -1;
  }
  return PLI_OK;
}

pli_status pli_stereo_match_points(pli_ctx* c, float* uright, float* depth, int32_t cap) {
  CtxGuard guard__(c);
  TraceRange range__("pli_stereo_match_points");
  if (!c) return PLI_ERR_INVALID;
  if (!c->orbDone[0] || !c->orbDone[1]) { g_err = "pli_orb_extract must run for both eyes first"; return PLI_ERR_STATE; }
  if (c->frameFresh && c->pointsFresh) {                // (pli_frame_extract has matched already, with this rig)
    const pli_table_layout& Yf = c->lay;
    const int Nf = c->orbCount[0];
    if (Nf > cap) { g_err = "output buffer too small"; return PLI_ERR_CAPACITY; }
    if (uright) std::memcpy(uright, c->hostRec.data() + Yf.off_uright, (size_t)Nf * 4);
    if (depth) std::memcpy(depth, c->hostRec.data() + Yf.off_depth, (size_t)Nf * 4);
    return PLI_OK;
  }
  HIPCHK(hipSetDevice(c->device));
  pli_status st = runStereoPoints(c, 1, c->ownTable);
  if (st != PLI_OK) return st;
  const pli_table_layout& Y = c->lay;
  int counts[8];
  HIPCHK(hipMemcpyAsync(counts, c->ownTable + Y.off_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int N = counts[0];
  if (c->frameFresh && N > 0) {                         // the Frame's cached record follows the rig it was re-matched with
    HIPCHK(hipMemcpy(c->hostRec.data() + Y.off_uright, c->ownTable + Y.off_uright, (size_t)N * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c->hostRec.data() + Y.off_depth, c->ownTable + Y.off_depth, (size_t)N * 4, hipMemcpyDeviceToHost));
  }
  c->pointsFresh = c->frameFresh;
  if (N > cap) { g_err = "output buffer too small"; return PLI_ERR_CAPACITY; }
  if (N > 0) {
    if (uright) HIPCHK(hipMemcpy(uright, c->ownTable + Y.off_uright, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (depth) HIPCHK(hipMemcpy(depth, c->ownTable + Y.off_depth, (size_t)N * 4, hipMemcpyDeviceToHost));
  }
  return PLI_OK;
}

pli_status pli_stereo_from_depth(pli_ctx* c, const float* depth, int64_t strideFloats, float* uright, float* depthOut, int32_t cap) {
  CtxGuard guard__(c);
  if (!c || !depth || strideFloats < c->cfg.width) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (!c->orbDone[0]) { g_err = "pli_orb_extract must run for the left eye first"; return PLI_ERR_STATE; }
  HIPCHK(hipSetDevice(c->device));
  const int W = c->cfg.width, H = c->cfg.height;
  pli_status st = ensureScratch(c, (size_t)W * H * 4);
  if (st != PLI_OK) return st;
  HIPCHK(hipMemcpy2DAsync(c->scratch, (size_t)W * 4, depth, (size_t)strideFloats * 4, (size_t)W * 4, H, hipMemcpyHostToDevice, c->stream));
  const pli_table_layout& Y = c->lay;
  LAUNCH(c, "k_stereo_from_depth", k_stereo_from_depth, dim3((c->hp.kpCap + 255) / 256), dim3(256), 0, c->dP, (const float*)c->scratch,
         (int64_t)W, W, H, c->ownTable, Y.off_counts, Y.off_kp[0], Y.off_uright, Y.off_depth);
  int counts[8];
  HIPCHK(hipMemcpyAsync(counts, c->ownTable + Y.off_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int N = counts[0];
  if (N > cap) { g_err = "output buffer too small"; return PLI_ERR_CAPACITY; }
  if (N > 0) {
    if (uright) HIPCHK(hipMemcpy(uright, c->ownTable + Y.off_uright, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (depthOut) HIPCHK(hipMemcpy(depthOut, c->ownTable + Y.off_depth, (size_t)N * 4, hipMemcpyDeviceToHost));
  }
  return PLI_OK;
}

// ORBextractor::operator() with a lapping area (ORBextractor.cc:1135-1144): pli_orb_extract, then the device table of that eye
// is put into the reference's mono-first / lapping-from-the-back order (k_lapping_order), which is the order
// pli_stereo_fisheye expects; *n_mono = the reference's return value (monoIndex).
pli_status pli_orb_extract_lapping(pli_ctx* c, int32_t eye, const uint8_t* img, int32_t w, int32_t h, int64_t stride,
                                   int32_t lap0, int32_t lap1, pli_keypoint* kp, int32_t cap, uint8_t* desc, int32_t* n,
                                   int32_t* n_mono) {
  CtxGuard guard__(c);
  if (!n_mono) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  *n_mono = 0;
  pli_status st = pli_orb_extract(c, eye, img, w, h, stride, nullptr, INT_MAX, nullptr, n);
  if (st != PLI_OK) return st;
  const int N = *n;
  if (N > cap) { g_err = "keypoint buffer too small"; return PLI_ERR_CAPACITY; }
  if (N == 0) return PLI_OK;
  const pli_table_layout& Y = c->lay;
  const size_t bk = alignUp((size_t)N * sizeof(pli_keypoint), 256), bd = alignUp((size_t)N * 32, 256);
  if ((st = ensureScratch(c, bk + bd + 256)) != PLI_OK) return st;
  uint8_t* p = (uint8_t*)c->scratch;
  pli_keypoint* sk = (pli_keypoint*)p;
  uint8_t* sd = p + bk;
  int* dmono = (int*)(p + bk + bd);
  pli_keypoint* tk = (pli_keypoint*)(c->ownTable + Y.off_kp[eye]);
  uint8_t* td = c->ownTable + Y.off_desc[eye];
  HIPCHK(hipMemcpyAsync(sk, tk, (size_t)N * sizeof(pli_keypoint), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(sd, td, (size_t)N * 32, hipMemcpyDeviceToDevice, c->stream));
  LAUNCH(c, "k_lapping_order", k_lapping_order, dim3(1), dim3(1024), 0, (const pli_keypoint*)sk, (const uint8_t*)sd, N, (float)lap0,
         (float)lap1, tk, td, dmono);
  int mono = 0;
  HIPCHK(hipMemcpyAsync(&mono, dmono, 4, hipMemcpyDeviceToHost, c->stream));
  if (kp) HIPCHK(hipMemcpyAsync(kp, tk, (size_t)N * sizeof(pli_keypoint), hipMemcpyDeviceToHost, c->stream));
  if (desc) HIPCHK(hipMemcpyAsync(desc, td, (size_t)N * 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *n_mono = mono;
  c->monoCount[eye] = mono;
  return PLI_OK;
}

// Frame::ComputeStereoFishEyeMatches (Frame.cc:1577-1618): tables on the device (kL/dL/kR/dR), results to host buffers.
// `scratchOff`: bytes of c->scratch already in use by the caller.
static pli_status fisheyeCore(pli_ctx* c, const pli_keypoint* kL, const uint8_t* dL, int NL, int monoL, const pli_keypoint* kR,
                              const uint8_t* dR, int NR, int monoR, size_t scratchOff, const pli_kb8_camera* cam1,
                              const pli_kb8_camera* cam2, const float* Rlr, const float* tlr, int32_t* l2r, int32_t* r2l, float* depth,
                              float* p3d, int32_t* nmatches) {
  monoL = std::min(std::max(monoL, 0), NL);
  monoR = std::min(std::max(monoR, 0), NR);
  const int nq = NL - monoL, nt = NR - monoR;
  const size_t bi = alignUp((size_t)std::max(nq, 1) * 16, 256), bl = alignUp((size_t)std::max(NL, 1) * 4, 256),
               br = alignUp((size_t)std::max(NR, 1) * 4, 256), bp = alignUp((size_t)std::max(NL, 1) * 12, 256);
  uint8_t* p = (uint8_t*)c->scratch + scratchOff;
  int* kidx = (int*)p; int* kdst = kidx + 2 * std::max(nq, 1); p += bi;
  int* dl2r = (int*)p; p += bl;
  float* ddepth = (float*)p; p += bl;
  int* dr2l = (int*)p; p += br;
  float* dp3 = (float*)p; p += bp;
  float* dRt = (float*)p; p += 128;
  float* dsig = (float*)p; p += 128;
  int* dcnt = (int*)p;
  float hRt[12], hsig[MAX_LEVELS];
  for (int i = 0; i < 9; ++i) hRt[i] = Rlr[i];
  for (int i = 0; i < 3; ++i) hRt[9 + i] = tlr[i];
  for (int l = 0; l < c->hp.nlevels; ++l) hsig[l] = c->hp.lv[l].scale * c->hp.lv[l].scale;      // mvLevelSigma2, ORBextractor.cc:424
  HIPCHK(hipMemcpyAsync(dRt, hRt, sizeof(hRt), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dsig, hsig, sizeof(float) * c->hp.nlevels, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(dl2r, 0xFF, (size_t)std::max(NL, 1) * 4, c->stream));
  HIPCHK(hipMemsetAsync(dr2l, 0xFF, (size_t)std::max(NR, 1) * 4, c->stream));
  HIPCHK(hipMemsetAsync(dp3, 0, (size_t)std::max(NL, 1) * 12, c->stream));
  HIPCHK(hipMemsetAsync(dcnt, 0, 4, c->stream));
  LAUNCH(c, "k_fill_f32", k_fill_f32, dim3((std::max(NL, 1) + 255) / 256), dim3(256), 0, ddepth, std::max(NL, 1), -1.0f);
  if (nq > 0) {
    LAUNCH(c, "k_knn2", k_knn2, dim3(nq), dim3(64), 0, dL + (size_t)monoL * 32, nq, dR + (size_t)monoR * 32, nt, kidx, kdst);
    Kb8 c1, c2;
    memcpy(&c1, cam1, sizeof(c1));
    memcpy(&c2, cam2, sizeof(c2));
    LAUNCH(c, "k_fisheye_triangulate", k_fisheye_triangulate, dim3((nq + 63) / 64), dim3(64), 0, kL, kR, (const int*)kidx,
           (const int*)kdst, nq, nt, monoL, monoR, c1, c2, (const float*)dRt, (const float*)dsig, dl2r, dr2l, ddepth, dp3, dcnt);
  }
  int cnt = 0;
  if (l2r && NL) HIPCHK(hipMemcpyAsync(l2r, dl2r, (size_t)NL * 4, hipMemcpyDeviceToHost, c->stream));
  if (r2l && NR) HIPCHK(hipMemcpyAsync(r2l, dr2l, (size_t)NR * 4, hipMemcpyDeviceToHost, c->stream));
  if (depth && NL) HIPCHK(hipMemcpyAsync(depth, ddepth, (size_t)NL * 4, hipMemcpyDeviceToHost, c->stream));
  if (p3d && NL) HIPCHK(hipMemcpyAsync(p3d, dp3, (size_t)NL * 12, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(&cnt, dcnt, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (nmatches) *nmatches = cnt;
  return PLI_OK;
}
static size_t fisheyeScratch(int NL, int NR) {
  return alignUp((size_t)std::max(NL, 1) * 16, 256) + 2 * alignUp((size_t)std::max(NL, 1) * 4, 256) +
         alignUp((size_t)std::max(NR, 1) * 4, 256) + alignUp((size_t)std::max(NL, 1) * 12, 256) + 512;
}

// ... on the device tables of the last pli_orb_extract_lapping of both eyes
pli_status pli_stereo_fisheye(pli_ctx* c, const pli_kb8_camera* cam1, const pli_kb8_camera* cam2, const float* Rlr, const float* tlr,
                              int32_t* l2r, int32_t capL, int32_t* r2l, int32_t capR, float* depth, float* p3d, int32_t* nmatches) {
  CtxGuard guard__(c);
  if (!c || !cam1 || !cam2 || !Rlr || !tlr) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (nmatches) *nmatches = 0;
  if (!c->orbDone[0] || !c->orbDone[1]) { g_err = "pli_orb_extract_lapping must run for both eyes first"; return PLI_ERR_STATE; }
  HIPCHK(hipSetDevice(c->device));
  const pli_table_layout& Y = c->lay;
  int counts[8];
  HIPCHK(hipMemcpyAsync(counts, c->ownTable + Y.off_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int NL = counts[0], NR = counts[1];
  if (NL > capL || NR > capR) { g_err = "output buffer too small"; return PLI_ERR_CAPACITY; }
  pli_status st = ensureScratch(c, fisheyeScratch(NL, NR));
  if (st != PLI_OK) return st;
  return fisheyeCore(c, (const pli_keypoint*)(c->ownTable + Y.off_kp[0]), c->ownTable + Y.off_desc[0], NL, c->monoCount[0],
                     (const pli_keypoint*)(c->ownTable + Y.off_kp[1]), c->ownTable + Y.off_desc[1], NR, c->monoCount[1], 0, cam1, cam2,
                     Rlr, tlr, l2r, r2l, depth, p3d, nmatches);
}

// ... on caller tables (the Frame members mvKeys / mDescriptors / mvKeysRight / mDescriptorsRight, monoLeft / monoRight)
pli_status pli_stereo_fisheye_tables(pli_ctx* c, const pli_keypoint* kpL, const uint8_t* descL, int32_t nleft, int32_t monoLeft,
                                     const pli_keypoint* kpR, const uint8_t* descR, int32_t nright, int32_t monoRight,
                                     const pli_kb8_camera* cam1, const pli_kb8_camera* cam2, const float* Rlr, const float* tlr,
                                     int32_t* l2r, int32_t* r2l, float* depth, float* p3d, int32_t* nmatches) {
  CtxGuard guard__(c);
  if (!c || !cam1 || !cam2 || !Rlr || !tlr || nleft < 0 || nright < 0 || (nleft > 0 && (!kpL || !descL)) ||
      (nright > 0 && (!kpR || !descR))) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (nmatches) *nmatches = 0;
  HIPCHK(hipSetDevice(c->device));
  const size_t bkl = alignUp((size_t)std::max(nleft, 1) * sizeof(pli_keypoint), 256), bdl = alignUp((size_t)std::max(nleft, 1) * 32, 256);
  const size_t bkr = alignUp((size_t)std::max(nright, 1) * sizeof(pli_keypoint), 256), bdr = alignUp((size_t)std::max(nright, 1) * 32, 256);
  pli_status st = ensureScratch(c, bkl + bdl + bkr + bdr + fisheyeScratch(nleft, nright));
  if (st != PLI_OK) return st;
  uint8_t* p = (uint8_t*)c->scratch;
  pli_keypoint* kL = (pli_keypoint*)p; uint8_t* dL = p + bkl; pli_keypoint* kR = (pli_keypoint*)(p + bkl + bdl); uint8_t* dR = p + bkl + bdl + bkr;
  if (nleft) {
    HIPCHK(hipMemcpyAsync(kL, kpL, (size_t)nleft * sizeof(pli_keypoint), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dL, descL, (size_t)nleft * 32, hipMemcpyHostToDevice, c->stream));
  }
  if (nright) {
    HIPCHK(hipMemcpyAsync(kR, kpR, (size_t)nright * sizeof(pli_keypoint), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dR, descR, (size_t)nright * 32, hipMemcpyHostToDevice, c->stream));
  }
  return fisheyeCore(c, kL, dL, nleft, monoLeft, kR, dR, nright, monoRight, bkl + bdl + bkr + bdr, cam1, cam2, Rlr, tlr, l2r, r2l, depth,
                     p3d, nmatches);
}

pli_status pli_stereo_match_lines(pli_ctx* c, float* disp, double* le, int32_t cap) {
  CtxGuard guard__(c);
  TraceRange range__("pli_stereo_match_lines");
  if (!c) return PLI_ERR_INVALID;
  if (!c->lineDone[0] || !c->lineDone[1]) { g_err = "pli_line_extract must run for both eyes first"; return PLI_ERR_STATE; }
  if (c->frameFresh) {                                  // (pli_frame_extract has matched already)
    const pli_table_layout& Yf = c->lay;
    const int Nf = c->lineCount[0];
    if (Nf > cap) { g_err = "output buffer too small"; return PLI_ERR_CAPACITY; }
    if (disp) std::memcpy(disp, c->hostRec.data() + Yf.off_disp, (size_t)Nf * 8);
    if (le) std::memcpy(le, c->hostRec.data() + Yf.off_le, (size_t)Nf * 24);
    return PLI_OK;
  }
  HIPCHK(hipSetDevice(c->device));
  pli_status st = runStereoLines(c, 1, c->ownTable);
  if (st != PLI_OK) return st;
  const pli_table_layout& Y = c->lay;
  int counts[8];
  HIPCHK(hipMemcpyAsync(counts, c->ownTable + Y.off_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int N = counts[2];
  if (N > cap) { g_err = "output buffer too small"; return PLI_ERR_CAPACITY; }
  if (N > 0) {
    if (disp) HIPCHK(hipMemcpy(disp, c->ownTable + Y.off_disp, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (le) HIPCHK(hipMemcpy(le, c->ownTable + Y.off_le, (size_t)N * 24, hipMemcpyDeviceToHost));
  }
  return PLI_OK;
}

pli_status pli_descriptor_distance(pli_ctx* c, const uint8_t* a, const uint8_t* b, int32_t n, int32_t* dist) {
  CtxGuard guard__(c);
  if (!c || n < 0 || (n > 0 && (!a || !b || !dist))) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (n == 0) return PLI_OK;
  HIPCHK(hipSetDevice(c->device));
  const size_t db = (size_t)n * 32;
  pli_status st = ensureScratch(c, 2 * db + (size_t)n * 4);
  if (st != PLI_OK) return st;
  uint8_t* da = (uint8_t*)c->scratch;
  uint8_t* dbp = da + db;
  int* dd = (int*)(dbp + db);
  HIPCHK(hipMemcpyAsync(da, a, db, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dbp, b, db, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_distance", k_distance, dim3((n + 255) / 256), dim3(256), 0, da, dbp, n, dd);
  HIPCHK(hipMemcpyAsync(dist, dd, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return PLI_OK;
}

pli_status pli_hamming_knn2(pli_ctx* c, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t* idx, int32_t* dist) {
  CtxGuard guard__(c);
  if (!c || nq < 0 || nt < 0 || (nq > 0 && (!q || !idx || !dist)) || (nt > 0 && !t)) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (nq == 0) return PLI_OK;
  HIPCHK(hipSetDevice(c->device));
  const size_t qb = alignUp((size_t)nq * 32, 256), tb = alignUp((size_t)std::max(nt, 1) * 32, 256), ob = (size_t)nq * 8;
  pli_status st = ensureScratch(c, qb + tb + 2 * ob);
  if (st != PLI_OK) return st;
  uint8_t* dq = (uint8_t*)c->scratch;
  uint8_t* dt = dq + qb;
  int* di = (int*)(dt + tb);
  int* dd = di + 2 * nq;
  HIPCHK(hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
  if (nt > 0) HIPCHK(hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_knn2", k_knn2, dim3(nq), dim3(64), 0, dq, nq, dt, nt, di, dd);
  HIPCHK(hipMemcpyAsync(idx, di, ob, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(dist, dd, ob, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return PLI_OK;
}

static pli_status matchDescriptors(pli_ctx* c, const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2, float nnr,
                                   bool mutual, int32_t* m12, int32_t* nmatches) {
  if (!c || n1 < 0 || n2 < 0 || (n1 > 0 && (!d1 || !m12)) || (n2 > 0 && !d2)) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (nmatches) *nmatches = 0;
  if (n1 == 0) return PLI_OK;
  HIPCHK(hipSetDevice(c->device));
  const size_t b1 = alignUp((size_t)n1 * 32, 256), b2 = alignUp((size_t)std::max(n2, 1) * 32, 256);
  const size_t k1 = alignUp((size_t)n1 * 16, 256), k2 = alignUp((size_t)std::max(n2, 1) * 16, 256);
  const size_t mm1 = alignUp((size_t)n1 * 4, 256), mm2 = alignUp((size_t)std::max(n2, 1) * 4, 256);
  pli_status st = ensureScratch(c, b1 + b2 + k1 + k2 + mm1 + mm2 + 256);
  if (st != PLI_OK) return st;
  uint8_t* p = (uint8_t*)c->scratch;
  uint8_t* dd1 = p; p += b1;
  uint8_t* dd2 = p; p += b2;
  int* i1 = (int*)p; int* ds1 = i1 + 2 * n1; p += k1;
  int* i2 = (int*)p; int* ds2 = i2 + 2 * std::max(n2, 1); p += k2;
  int* dm12 = (int*)p; p += mm1;
  int* dm21 = (int*)p; p += mm2;
  int* dcount = (int*)p;
  HIPCHK(hipMemcpyAsync(dd1, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, c->stream));
  if (n2 > 0) HIPCHK(hipMemcpyAsync(dd2, d2, (size_t)n2 * 32, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(dcount, 0, 4, c->stream));
  LAUNCH(c, "k_knn2", k_knn2, dim3(n1), dim3(64), 0, dd1, n1, dd2, n2, i1, ds1);
  LAUNCH(c, "k_ratio", k_ratio, dim3((n1 + 255) / 256), dim3(256), 0, i1, ds1, n1, n2, nnr, dm12);
  const bool lr = mutual;
  if (lr && n2 > 0) {
    LAUNCH(c, "k_knn2", k_knn2, dim3(n2), dim3(64), 0, dd2, n2, dd1, n1, i2, ds2);
    LAUNCH(c, "k_ratio", k_ratio, dim3((n2 + 255) / 256), dim3(256), 0, i2, ds2, n2, n1, nnr, dm21);
  }
  LAUNCH(c, "k_mutual", k_mutual, dim3((n1 + 255) / 256), dim3(256), 0, dm12, (lr && n2 > 0) ? dm21 : (int*)nullptr, n1, dcount);
  int cnt = 0;
  HIPCHK(hipMemcpyAsync(m12, dm12, (size_t)n1 * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(&cnt, dcount, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (nmatches) *nmatches = cnt;
  return PLI_OK;
}

pli_status pli_match_lines(pli_ctx* c, const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2, float nnr,
                           int32_t* m12, int32_t* nmatches) {
  CtxGuard guard__(c);
  return matchDescriptors(c, d1, n1, d2, n2, nnr, c && c->cfg.best_lr_matches != 0, m12, nmatches);
}

pli_status pli_match_nnr(pli_ctx* c, const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2, float nnr,
                         int32_t* m12, int32_t* nmatches) {
  CtxGuard guard__(c);
  return matchDescriptors(c, d1, n1, d2, n2, nnr, false, m12, nmatches);
}

// Both projection searches: mode 0 frame-to-frame (ORBmatcher.cc:2179-2323), mode 1 local map (:44-143).
// Frames whose keypoints fit the LDS owner table take the two-phase form (candidates in parallel, then the ordered
// assignment); larger ones the single-wave kernels that scan the frame per query.
constexpr int PROJ_LDS_KEYPOINTS = 15360;
constexpr int PROJ_CAND = 64;                 // PROJ_K of match_kernels.hip

static pli_status projectionSearch(pli_ctx* c, int mode, const pli_proj_query* q, const uint8_t* qdesc, int32_t nq,
                                   const pli_keypoint* kp, const uint8_t* desc, const float* uright, const uint8_t* occupied,
                                   int32_t ncur, float minX, float maxX, float minY, float maxY, int32_t checkOri,
                                   float nnratio, int32_t* best, int32_t* nmatches, int32_t* raw = nullptr) {
  if (!c || nq < 0 || ncur < 0 || (nq > 0 && (!q || !qdesc || !best)) || (ncur > 0 && (!kp || !desc || !uright))) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (nmatches) *nmatches = 0;
  if (nq == 0) return PLI_OK;
  if (ncur >= (1 << 28)) { g_err = "too many keypoints"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  const int nc = std::max(ncur, 1);
  const bool twoPhase = ncur <= PROJ_LDS_KEYPOINTS;
  const size_t bq = alignUp((size_t)nq * sizeof(pli_proj_query), 256), bqd = alignUp((size_t)nq * 32, 256);
  const size_t bk = alignUp((size_t)nc * sizeof(pli_keypoint), 256), bd = alignUp((size_t)nc * 32, 256), bu = alignUp((size_t)nc * 4, 256);
  const size_t bo = alignUp((size_t)nc * 4, 256), bb = alignUp((size_t)nq * 4, 256), bc = alignUp((size_t)nc, 256);
  for (int i = 0; i < nq; ++i)
    if (mode == 0 && (q[i].valid & ~3)) { g_err = "pli_proj_query.valid: 0, 1 or 1 | PLI_PROJ_NO_OBSERVATIONS"; return PLI_ERR_INVALID; }
  const size_t bkeys = twoPhase ? alignUp((size_t)nq * PROJ_CAND * 8, 256) : 0, bcc = twoPhase ? alignUp((size_t)nq * 4, 256) : 0;
  pli_status st = ensureScratch(c, bq + bqd + bk + bd + bu + bo + 2 * bb + bc + bkeys + bcc + 256);
  if (st != PLI_OK) return st;
  uint8_t* p = (uint8_t*)c->scratch;
  pli_proj_query* dq = (pli_proj_query*)p; p += bq;
  uint8_t* dqd = p; p += bqd;
  pli_keypoint* dk = (pli_keypoint*)p; p += bk;
  uint8_t* ddsc = p; p += bd;
  float* du = (float*)p; p += bu;
  int* down = (int*)p; p += bo;
  int* dbest = (int*)p; p += bb;
  int* draw = (int*)p; p += bb;
  uint8_t* docc = p; p += bc;
  unsigned long long* dkeys = (unsigned long long*)p; p += bkeys;
  int* dcc = (int*)p; p += bcc;
  int* dcnt = (int*)p;
  HIPCHK(hipMemcpyAsync(dq, q, (size_t)nq * sizeof(pli_proj_query), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dqd, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
  if (ncur > 0) {
    HIPCHK(hipMemcpyAsync(dk, kp, (size_t)ncur * sizeof(pli_keypoint), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(ddsc, desc, (size_t)ncur * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(du, uright, (size_t)ncur * 4, hipMemcpyHostToDevice, c->stream));
    if (occupied) HIPCHK(hipMemcpyAsync(docc, occupied, (size_t)ncur, hipMemcpyHostToDevice, c->stream));
  }
  const uint8_t* occ = (occupied && ncur > 0) ? (const uint8_t*)docc : (const uint8_t*)nullptr;
  int* rawOut = (mode == 0 && raw) ? draw : (int*)nullptr;
  if (twoPhase) {
    // a second-best farther than 100 / nnratio can no longer reject a best of <= 100 (ORBmatcher.cc:124-126)
    int limit = 100;
    if (mode == 1) limit = (nnratio > 0.4f) ? std::min(255, (int)(100.0f / nnratio) + 2) : 255;
    LAUNCH(c, "k_proj_candidates", k_proj_candidates, dim3(nq), dim3(64), 0, dq, dqd, nq, dk, ddsc, du, ncur, minX, maxX, minY, maxY,
           mode == 0 ? 1 : 0, limit, dkeys, dcc);
    LAUNCH(c, "k_proj_assign", k_proj_assign, dim3(1), dim3(64), (size_t)nc * 4, dq, dqd, nq, dk, ddsc, du, occ, ncur, minX, maxX,
           minY, maxY, mode, checkOri, nnratio, dkeys, dcc, dbest, dcnt, rawOut);
  } else if (mode == 0) {
    LAUNCH(c, "k_search_by_projection", k_search_by_projection, dim3(1), dim3(64), 0, dq, dqd, nq, dk, ddsc, du, ncur, minX, maxX,
           minY, maxY, checkOri, down, dbest, dcnt, occ, rawOut);
  } else {
    LAUNCH(c, "k_search_local_map", k_search_local_map, dim3(1), dim3(64), 0, dq, dqd, nq, dk, ddsc, du, occ, ncur, minX, maxX,
           minY, maxY, nnratio, down, dbest, dcnt);
  }
  int cnt = 0;
  HIPCHK(hipMemcpyAsync(best, dbest, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
  if (rawOut) HIPCHK(hipMemcpyAsync(raw, draw, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(&cnt, dcnt, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (nmatches) *nmatches = cnt;
  return PLI_OK;
}

pli_status pli_search_by_projection(pli_ctx* c, const pli_proj_query* q, const uint8_t* qdesc, int32_t nq,
                                    const pli_keypoint* kp, const uint8_t* desc, const float* uright,
                                    const uint8_t* occupied, int32_t ncur,
                                    float minX, float maxX, float minY, float maxY, int32_t checkOri,
                                    int32_t* best, int32_t* raw, int32_t* nmatches) {
  CtxGuard guard__(c);
  return projectionSearch(c, 0, q, qdesc, nq, kp, desc, uright, occupied, ncur, minX, maxX, minY, maxY, checkOri, 0.0f, best, nmatches, raw);
}

pli_status pli_search_local_map(pli_ctx* c, const pli_proj_query* q, const uint8_t* qdesc, int32_t nq,
                                const pli_keypoint* kp, const uint8_t* desc, const float* uright,
                                const uint8_t* occupied, int32_t ncur, float minX, float maxX, float minY, float maxY,
                                float nnratio, int32_t* best, int32_t* nmatches) {
  CtxGuard guard__(c);
  return projectionSearch(c, 1, q, qdesc, nq, kp, desc, uright, occupied, ncur, minX, maxX, minY, maxY, 0, nnratio, best, nmatches);
}

pli_status pli_search_local_map_fisheye(pli_ctx* c, const pli_proj_query* qL, const pli_proj_query* qR, const uint8_t* qdesc,
                                        int32_t nq, const pli_keypoint* kpL, const uint8_t* descL, const uint8_t* occL,
                                        const int32_t* l2r, int32_t nL, const pli_keypoint* kpR, const uint8_t* descR,
                                        const uint8_t* occR, const int32_t* r2l, int32_t nR, float minX, float maxX, float minY,
                                        float maxY, float nnratio, int32_t* mpL, int32_t* mpR, int32_t* nmatches) {
  CtxGuard guard__(c);
  if (!c || nq < 0 || nL < 0 || nR < 0 || (nq > 0 && (!qL || !qR || !qdesc)) || (nL > 0 && (!kpL || !descL || !l2r || !mpL)) ||
      (nR > 0 && (!kpR || !descR || !r2l || !mpR))) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (nmatches) *nmatches = 0;
  for (int i = 0; i < nL; ++i) {
    mpL[i] = -1;
    if (l2r[i] < -1 || l2r[i] >= nR) { g_err = "left_to_right out of range"; return PLI_ERR_INVALID; }
  }
  for (int i = 0; i < nR; ++i) {
    mpR[i] = -1;
    if (r2l[i] < -1 || r2l[i] >= nL) { g_err = "right_to_left out of range"; return PLI_ERR_INVALID; }
  }
  if (nq == 0 || nL + nR == 0) return PLI_OK;
  if (nL + nR > PROJ_LDS_KEYPOINTS) { g_err = "too many keypoints for the LDS slot tables of the fisheye local-map search"; return PLI_ERR_CAPACITY; }
  HIPCHK(hipSetDevice(c->device));
  const int ncL = std::max(nL, 1), ncR = std::max(nR, 1), ncM = std::max(ncL, ncR);
  const size_t bq = alignUp((size_t)nq * sizeof(pli_proj_query), 256), bqd = alignUp((size_t)nq * 32, 256);
  const size_t bkL = alignUp((size_t)ncL * sizeof(pli_keypoint), 256), bdL = alignUp((size_t)ncL * 32, 256), boL = alignUp((size_t)ncL, 256), biL = alignUp((size_t)ncL * 4, 256);
  const size_t bkR = alignUp((size_t)ncR * sizeof(pli_keypoint), 256), bdR = alignUp((size_t)ncR * 32, 256), boR = alignUp((size_t)ncR, 256), biR = alignUp((size_t)ncR * 4, 256);
  const size_t bu = alignUp((size_t)ncM * 4, 256), bkeys = alignUp((size_t)nq * PROJ_CAND * 8, 256), bcc = alignUp((size_t)nq * 4, 256);
  pli_status st = ensureScratch(c, 2 * bq + bqd + bkL + bdL + boL + 2 * biL + bkR + bdR + boR + 2 * biR + bu + 2 * bkeys + 2 * bcc + 256);
  if (st != PLI_OK) return st;
  uint8_t* p = (uint8_t*)c->scratch;
  auto take = [&](size_t b) { uint8_t* r = p; p += b; return r; };
  pli_proj_query* dqL = (pli_proj_query*)take(bq); pli_proj_query* dqR = (pli_proj_query*)take(bq);
  uint8_t* dqd = take(bqd);
  pli_keypoint* dkL = (pli_keypoint*)take(bkL); uint8_t* ddL = take(bdL); uint8_t* doL = take(boL); int* dl2r = (int*)take(biL); int* dmpL = (int*)take(biL);
  pli_keypoint* dkR = (pli_keypoint*)take(bkR); uint8_t* ddR = take(bdR); uint8_t* doR = take(boR); int* dr2l = (int*)take(biR); int* dmpR = (int*)take(biR);
  float* du = (float*)take(bu);
  unsigned long long* dkeysL = (unsigned long long*)take(bkeys); unsigned long long* dkeysR = (unsigned long long*)take(bkeys);
  int* dccL = (int*)take(bcc); int* dccR = (int*)take(bcc);
  int* dcnt = (int*)p;
  HIPCHK(hipMemcpyAsync(dqL, qL, (size_t)nq * sizeof(pli_proj_query), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dqR, qR, (size_t)nq * sizeof(pli_proj_query), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dqd, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
  if (nL > 0) {
    HIPCHK(hipMemcpyAsync(dkL, kpL, (size_t)nL * sizeof(pli_keypoint), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(ddL, descL, (size_t)nL * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dl2r, l2r, (size_t)nL * 4, hipMemcpyHostToDevice, c->stream));
    if (occL) HIPCHK(hipMemcpyAsync(doL, occL, (size_t)nL, hipMemcpyHostToDevice, c->stream));
  }
  if (nR > 0) {
    HIPCHK(hipMemcpyAsync(dkR, kpR, (size_t)nR * sizeof(pli_keypoint), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(ddR, descR, (size_t)nR * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dr2l, r2l, (size_t)nR * 4, hipMemcpyHostToDevice, c->stream));
    if (occR) HIPCHK(hipMemcpyAsync(doR, occR, (size_t)nR, hipMemcpyHostToDevice, c->stream));
  }
  // (the fisheye branch has no mvuRight gate: a plane of -1 switches it off in the shared candidate kernel)
  LAUNCH(c, "k_fill_f32", k_fill_f32, dim3((ncM + 255) / 256), dim3(256), 0, du, ncM, -1.0f);
  const int limit = (nnratio > 0.4f) ? std::min(255, (int)(100.0f / nnratio) + 2) : 255;
  LAUNCH(c, "k_proj_candidates", k_proj_candidates, dim3(nq), dim3(64), 0, dqL, dqd, nq, dkL, ddL, du, nL, minX, maxX, minY, maxY, 0, limit,
         dkeysL, dccL);
  LAUNCH(c, "k_proj_candidates", k_proj_candidates, dim3(nq), dim3(64), 0, dqR, dqd, nq, dkR, ddR, du, nR, minX, maxX, minY, maxY, 0, limit,
         dkeysR, dccR);
  LAUNCH(c, "k_proj_assign_fisheye", k_proj_assign_fisheye, dim3(1), dim3(64), (size_t)(nL + nR) * 4, dqL, dqR, dqd, nq, dkL, ddL,
         (occL && nL > 0) ? (const uint8_t*)doL : (const uint8_t*)nullptr, dl2r, nL, dkR, ddR,
         (occR && nR > 0) ? (const uint8_t*)doR : (const uint8_t*)nullptr, dr2l, nR, du, minX, maxX, minY, maxY, nnratio, dkeysL, dccL,
         dkeysR, dccR, dmpL, dmpR, dcnt);
  int cnt = 0;
  if (nL > 0) HIPCHK(hipMemcpyAsync(mpL, dmpL, (size_t)nL * 4, hipMemcpyDeviceToHost, c->stream));
  if (nR > 0) HIPCHK(hipMemcpyAsync(mpR, dmpR, (size_t)nR * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(&cnt, dcnt, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (nmatches) *nmatches = cnt;
  return PLI_OK;
}

// ---- frame-to-frame track matching of a batch (match_kernels.hip: k_track_*) ------------------------
static void trackLayout(const pli_ctx* c, pli_track_layout& L) {
  int64_t o = 0;
  auto take = [&](int64_t bytes) { int64_t r = o; o = alignUp(o + bytes, 16); return r; };
  L.kp_cap = c->hp.kpCap; L.kl_cap = c->hp.klCap;
  L.off_counts = take(16);
  L.off_best = take((int64_t)L.kp_cap * 4);
  L.off_lines = take((int64_t)L.kl_cap * 4);
  L.record_bytes = alignUp(o, 256);
}

pli_status pli_track_layout_get(const pli_ctx* c, pli_track_layout* out) {
  CtxGuard guard__(c);
  if (!c || !out) return PLI_ERR_INVALID;
  trackLayout(c, *out);
  return PLI_OK;
}

pli_status pli_batch_track(pli_ctx* c, int32_t nframes, const void* table, const float* poses, const pli_track_params* tpar,
                           void* track) {
  CtxGuard guard__(c);
  TraceRange range__("pli_batch_track");
  if (!c || !table || !poses || !tpar || !track) { g_err = "null argument"; return PLI_ERR_INVALID; }
  if (nframes < 1 || nframes > c->cfg.max_frames) { g_err = "nframes exceeds the context's max_frames"; return PLI_ERR_INVALID; }
  if (!(tpar->max_x > tpar->min_x) || !(tpar->max_y > tpar->min_y) || !(tpar->fx > 0) || !(tpar->fy > 0)) { g_err = "bad track parameters"; return PLI_ERR_INVALID; }
  if (nframes < 2) return PLI_OK;
  HIPCHK(hipSetDevice(c->device));
  const DevParams& P = c->hp;
  if ((size_t)P.kpCap * 4 > 160 * 1024 - 1024) { g_err = "too many keypoints for the LDS owner table of the batch track matcher"; return PLI_ERR_INVALID; }
  pli_track_layout TL;
  trackLayout(c, TL);
  TrackParams tp;
  tp.fx = tpar->fx; tp.fy = tpar->fy; tp.cx = tpar->cx; tp.cy = tpar->cy; tp.bf = tpar->bf; tp.th = tpar->th;
  tp.minX = tpar->min_x; tp.maxX = tpar->max_x; tp.minY = tpar->min_y; tp.maxY = tpar->max_y;
  tp.mono = tpar->mono; tp.checkOri = tpar->check_orientation; tp.nnrLines = tpar->nnr_lines;
  tp.kpCap = P.kpCap; tp.klCap = P.klCap;
  const pli_table_layout& Y = c->lay;
  tp.recordBytes = Y.record_bytes; tp.offCounts = Y.off_counts; tp.offKp0 = Y.off_kp[0]; tp.offDesc0 = Y.off_desc[0];
  tp.offUr = Y.off_uright; tp.offDepth = Y.off_depth; tp.offLd0 = Y.off_ldesc[0];
  tp.trackBytes = TL.record_bytes; tp.toffCounts = TL.off_counts; tp.toffBest = TL.off_best; tp.toffLines = TL.off_lines;
  const size_t NF = (size_t)nframes;
  const size_t bq = alignUp(NF * P.kpCap * sizeof(pli_proj_query), 256), bk = alignUp(NF * P.kpCap * 64 * 8, 256),
               bc = alignUp(NF * P.kpCap * 4, 256), bl = alignUp(NF * std::max(P.klCap, 1) * 4, 256);
  pli_status st = ensureScratch(c, bq + bk + bc + bl);
  if (st != PLI_OK) return st;
  uint8_t* p = (uint8_t*)c->scratch;
  pli_proj_query* dq = (pli_proj_query*)p; p += bq;
  unsigned long long* dkeys = (unsigned long long*)p; p += bk;
  int* dcc = (int*)p; p += bc;
  int* dl = (int*)p;
  const uint8_t* T = (const uint8_t*)table;
  LAUNCH(c, "k_track_queries", k_track_queries, dim3((P.kpCap + 255) / 256, nframes - 1), dim3(256), 0, c->dP, T, poses, tp, dq);
  LAUNCH(c, "k_track_candidates", k_track_candidates, dim3(P.kpCap, nframes - 1), dim3(64), 0, T, tp, dq, dkeys, dcc);
  LAUNCH(c, "k_track_assign", k_track_assign, dim3(nframes - 1), dim3(64), (size_t)P.kpCap * 4, T, tp, dq, dkeys, dcc, (uint8_t*)track);
  LAUNCH(c, "k_track_lines", k_track_lines, dim3(nframes - 1), dim3(256), 0, T, tp, c->cfg.best_lr_matches != 0 ? 1 : 0, dl, (uint8_t*)track);
  return PLI_OK;
}

// ---- bag of words (DBoW2 vocabulary tree) -------------------------------------
struct pli_vocab {
  int device = 0, k = 0, L = 0, nnodes = 0, nwords = 0;     // nnodes counts the root
  int* childOff = nullptr; int* childList = nullptr; uint8_t* desc = nullptr; int* word = nullptr; double* weight = nullptr;
};

void pli_vocab_destroy(pli_vocab* v) {
  if (!v) return;
  hipSetDevice(v->device);
  hipFree(v->childOff); hipFree(v->childList); hipFree(v->desc); hipFree(v->word); hipFree(v->weight);
  delete v;
}

pli_status pli_vocab_create(pli_ctx* c, int32_t k, int32_t L, int32_t n, const int32_t* parent, const uint8_t* isLeaf,
                            const uint8_t* desc, const double* weight, pli_vocab** out) {
  CtxGuard guard__(c);
  if (!c || !out || n < 1 || !parent || !isLeaf || !desc || !weight || k < 1 || L < 1) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  *out = nullptr;
  const int N = n + 1;                                      // + root (TemplatedVocabulary.h:1387)
  std::vector<int> cnt(N + 1, 0), off(N + 1, 0), list(n), word(N, -1);
  std::vector<double> w(N, 0.0);
  std::vector<uint8_t> d((size_t)N * 32, 0);
  int nwords = 0;
  for (int i = 0; i < n; ++i) {
    const int nid = i + 1, pid = parent[i];
    if (pid < 0 || pid >= nid) { g_err = "vocabulary: a node's parent must precede it"; return PLI_ERR_INVALID; }
    cnt[pid]++;
    std::memcpy(&d[(size_t)nid * 32], desc + (size_t)i * 32, 32);
    w[nid] = weight[i];
    if (isLeaf[i]) word[nid] = nwords++;                   // word ids in file order (:1421-1427)
  }
  for (int i = 0; i < N; ++i) off[i + 1] = off[i] + cnt[i];
  std::vector<int> fill(off.begin(), off.end() - 1);
  for (int i = 0; i < n; ++i) list[fill[parent[i]]++] = i + 1;   // children in file order (:1402)
  for (int i = 1; i < N; ++i)
    if (cnt[i] == 0 && word[i] < 0) { g_err = "vocabulary: a node without children is not flagged as a word"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  std::unique_ptr<pli_vocab, void (*)(pli_vocab*)> v(new pli_vocab(), pli_vocab_destroy);
  v->device = c->device; v->k = k; v->L = L; v->nnodes = N; v->nwords = nwords;
  HIPCHK(hipMalloc(&v->childOff, (size_t)(N + 1) * 4));
  HIPCHK(hipMalloc(&v->childList, (size_t)n * 4));
  HIPCHK(hipMalloc(&v->desc, (size_t)N * 32));
  HIPCHK(hipMalloc(&v->word, (size_t)N * 4));
  HIPCHK(hipMalloc(&v->weight, (size_t)N * 8));
  HIPCHK(hipMemcpy(v->childOff, off.data(), (size_t)(N + 1) * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(v->childList, list.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(v->desc, d.data(), (size_t)N * 32, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(v->word, word.data(), (size_t)N * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(v->weight, w.data(), (size_t)N * 8, hipMemcpyHostToDevice));
  *out = v.release();
  return PLI_OK;
}

pli_status pli_bow_transform(pli_ctx* c, const pli_vocab* v, const uint8_t* desc, int32_t n, int32_t levelsup, int32_t* wordId,
                             double* weight, int32_t* nodeId) {
  CtxGuard guard__(c);
  if (!c || !v || n < 0 || (n > 0 && (!desc || !wordId || !weight || !nodeId))) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  if (n == 0) return PLI_OK;
  HIPCHK(hipSetDevice(c->device));
  const size_t bd = alignUp((size_t)n * 32, 256), bw = alignUp((size_t)n * 4, 256), bf = alignUp((size_t)n * 8, 256);
  pli_status st = ensureScratch(c, bd + 2 * bw + bf);
  if (st != PLI_OK) return st;
  uint8_t* p = (uint8_t*)c->scratch;
  uint8_t* dd = p; p += bd;
  double* dwt = (double*)p; p += bf;
  int* dword = (int*)p; p += bw;
  int* dnode = (int*)p;
  HIPCHK(hipMemcpyAsync(dd, desc, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_bow_descend", k_bow_descend, dim3(n), dim3(64), 0, dd, n, v->childOff, v->childList, v->desc, v->word, v->weight,
         v->L - levelsup, dword, dwt, dnode);
  HIPCHK(hipMemcpyAsync(wordId, dword, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(weight, dwt, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(nodeId, dnode, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return PLI_OK;
}

// ---- measurement -----------------------------------------------------------
int64_t pli_trace_ranges(void) { return (int64_t)g_roctx.pushed.load(std::memory_order_relaxed); }

pli_status pli_prof_enable(pli_ctx* c, int32_t on) {
  CtxGuard guard__(c);
  if (!c) return PLI_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->prof = on != 0;
  return PLI_OK;
}
pli_status pli_prof_reset(pli_ctx* c) {
  CtxGuard guard__(c);
  if (!c) return PLI_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->profLog.clear();
  c->evNext = 0;
  return PLI_OK;
}
pli_status pli_prof_report(pli_ctx* c, char* buf, int64_t bytes) {
  CtxGuard guard__(c);
  if (!c || !buf || bytes <= 0) return PLI_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(c->stream));
  std::map<std::string, std::pair<int, double>> acc;
  std::vector<std::string> orderNames;
  for (const ProfEntry& e : c->profLog) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) continue;
    auto it = acc.find(e.name);
    if (it == acc.end()) { acc[e.name] = std::make_pair(1, (double)ms); orderNames.push_back(e.name); }
    else { it->second.first++; it->second.second += ms; }
  }
  std::string s;
  char line[256];
  for (const std::string& nme : orderNames) {
    std::snprintf(line, sizeof(line), "%s %d %.6f\n", nme.c_str(), acc[nme].first, acc[nme].second);
    s += line;
  }
  if ((int64_t)s.size() + 1 > bytes) { g_err = "report buffer too small"; return PLI_ERR_CAPACITY; }
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return PLI_OK;
}

pli_status pli_debug_enable(pli_ctx* c, int32_t on) {
  CtxGuard guard__(c);
  if (!c) return PLI_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (on && !c->angDbg) {
    pli_status st;
    if ((st = c->dalloc(&c->angDbg, (size_t)c->hp.LW * c->hp.LH * c->NI)) != PLI_OK) return st;
    if ((st = c->dalloc(&c->lbdFloat, (size_t)c->NI * c->hp.klCap * 72)) != PLI_OK) return st;
    if (c->lsdF64 && (st = c->dalloc(&c->scaled64Dbg, (size_t)c->hp.LW * c->hp.LH * c->NI)) != PLI_OK) return st;
  }
  c->debug = on != 0;
  return PLI_OK;
}

pli_status pli_debug_fetch(pli_ctx* c, int32_t image, int32_t what, int32_t arg, void* dst, int64_t dstBytes, int64_t* outBytes) {
  CtxGuard guard__(c);
  if (!c || image < 0 || image >= c->NI || !outBytes) { g_err = "bad argument"; return PLI_ERR_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  const DevParams& P = c->hp;
  auto need = [&](int64_t n) -> bool { *outBytes = n; return dst && dstBytes >= n; };
  switch (what) {
    case PLI_DBG_PYRAMID_LEVEL:
    case PLI_DBG_BLUR_LEVEL: {
      if (arg < 0 || arg >= P.nlevels) return PLI_ERR_INVALID;
      const LevelGeom& G = P.lv[arg];
      if (!need((int64_t)G.w * G.h)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      const uint8_t* base = (what == PLI_DBG_PYRAMID_LEVEL ? c->pyr : c->blur) + (int64_t)image * P.pyrBlock + G.offset;
      HIPCHK(hipMemcpy2D(dst, G.w, base, G.pitch, G.w, G.h, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    case PLI_DBG_FAST_CANDIDATES:
    case PLI_DBG_LEVEL_KEYPOINTS: {
      if (arg < 0 || arg >= P.nlevels) return PLI_ERR_INVALID;
      const LevelGeom& G = P.lv[arg];
      int n = 0;
      const int* cntp = (what == PLI_DBG_FAST_CANDIDATES ? c->candCount : c->kpSelCount) + image * P.nlevels + arg;
      HIPCHK(hipMemcpy(&n, cntp, 4, hipMemcpyDeviceToHost));
      if (!need(4 + (int64_t)n * 12)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      std::vector<uint32_t> raw(std::max(n, 1));
      const uint32_t* src = what == PLI_DBG_FAST_CANDIDATES ? c->candAll + (int64_t)image * P.candPerImage + G.candBase
                                                            : c->kpSel + (int64_t)image * P.kpSlotsPerImage + G.kpBase;
      if (n) HIPCHK(hipMemcpy(raw.data(), src, (size_t)n * 4, hipMemcpyDeviceToHost));
      int* o = (int*)dst;
      o[0] = n;
      for (int i = 0; i < n; ++i) { o[1 + 3 * i] = (raw[i] >> 8) & 0xFFF; o[2 + 3 * i] = raw[i] >> 20; o[3 + 3 * i] = raw[i] & 0xFF; }
      return PLI_OK;
    }
    case PLI_DBG_LSD_SCALED: {
      if (c->lsdF64) {     // CV_64F pipeline: doubles (the debug copy: the plane itself is the growers' arena)
        if (!need((int64_t)P.LWt * P.LH * 8)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
        if (!c->scaled64Dbg) { g_err = "pli_debug_enable was not on during the run"; return PLI_ERR_STATE; }
        const double* src64 = c->scaled64Dbg;     // (rows of the true width, images pitch x height apart: runLines)
        HIPCHK(hipMemcpy(dst, src64 + (int64_t)image * P.LW * P.LH, (size_t)P.LWt * P.LH * 8, hipMemcpyDeviceToHost));
        return PLI_OK;
      }
      if (!need((int64_t)P.LWt * P.LH)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      if (c->cfg.lsd_scale != 1)
        HIPCHK(hipMemcpy2D(dst, P.LWt, c->lsdScaled + (int64_t)image * c->lsdStride, P.lpitch, P.LWt, P.LH, hipMemcpyDeviceToHost));
      else
        HIPCHK(hipMemcpy2D(dst, P.W, c->pyr + (int64_t)image * P.pyrBlock, P.lv[0].pitch, P.W, P.H, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    case PLI_DBG_LSD_ANGLE: {
      if (!c->angDbg) { g_err = "pli_debug_enable was not on during the run"; return PLI_ERR_STATE; }
      const int64_t n = (int64_t)P.LWt * P.LH * 4;   // (the debug plane has rows of the true width)
      if (!need(n)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      HIPCHK(hipMemcpy(dst, c->angDbg + (int64_t)image * P.LWt * P.LH, n, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    case PLI_DBG_LSD_SEGMENTS: {
      int n = 0;
      HIPCHK(hipMemcpy(&n, c->nSeg + image, 4, hipMemcpyDeviceToHost));
      if (!need(4 + (int64_t)n * 16)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      *(int*)dst = n;
      if (n) HIPCHK(hipMemcpy((char*)dst + 4, c->seg + (int64_t)image * c->maxSeg * 4, (size_t)n * 16, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    case PLI_DBG_LSD_ORDER: {
      int n = 0;
      HIPCHK(hipMemcpy(&n, c->nDefined + image, 4, hipMemcpyDeviceToHost));
      if (!need(4 + (int64_t)n * 4)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      *(int*)dst = n;
      if (n) HIPCHK(hipMemcpy((char*)dst + 4, c->order + (int64_t)image * P.LW * P.LH, (size_t)n * 4, hipMemcpyDeviceToHost));
      if (P.LW != P.LWt) {     // device pixel indices are y * pitch + x: the caller gets y * width + x
        int* o = (int*)dst + 1;
        for (int i = 0; i < n; ++i) o[i] = (o[i] / P.LW) * P.LWt + o[i] % P.LW;
      }
      return PLI_OK;
    }
    case PLI_DBG_LBD_DXDY: {
      const int64_t n = (int64_t)P.W * P.H * 2;
      if (!need(2 * n)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      // device layout: interleaved (dx, dy) pairs; the caller gets the dx plane followed by the dy plane
      std::vector<short> tmp((size_t)P.W * P.H * 2);
      HIPCHK(hipMemcpy(tmp.data(), c->dxy + (int64_t)image * P.W * P.H, 2 * n, hipMemcpyDeviceToHost));
      short* o = (short*)dst;
      for (int64_t i = 0; i < (int64_t)P.W * P.H; ++i) { o[i] = tmp[2 * i]; o[(int64_t)P.W * P.H + i] = tmp[2 * i + 1]; }
      return PLI_OK;
    }
    case PLI_DBG_LBD_FLOAT: {
      if (!c->lbdFloat) { g_err = "pli_debug_enable was not on during the run"; return PLI_ERR_STATE; }
      const int64_t n = (int64_t)P.klCap * 72 * 4;
      if (!need(n)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      HIPCHK(hipMemcpy(dst, c->lbdFloat + (int64_t)image * P.klCap * 72, n, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    case PLI_DBG_LSD_OWNER: {
      const int64_t np = (int64_t)P.LW * P.LH;
      if (!need(4 + (int64_t)P.LWt * P.LH * 4)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      RxCtl ctl;
      HIPCHK(hipMemcpy(&ctl, c->jrCtl + image, sizeof(ctl), hipMemcpyDeviceToHost));
      std::vector<int2> own(np);
      HIPCHK(hipMemcpy(own.data(), c->own + image * np, np * 8, hipMemcpyDeviceToHost));
      int* o = (int*)dst;
      o[0] = ctl.rounds;
      const int comp = (ctl.rounds - 1) & 1;          // owner_{t-1} of the round that detected the fixed point
      for (int y = 0; y < P.LH; ++y)
        for (int x = 0; x < P.LWt; ++x) { const int2 w = own[(int64_t)y * P.LW + x]; o[1 + (int64_t)y * P.LWt + x] = comp ? w.y : w.x; }
      return PLI_OK;
    }
    case 15: {   // debug: raw owner pairs
      const int64_t np = (int64_t)P.LW * P.LH;
      if (!need(np * 8)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      HIPCHK(hipMemcpy(dst, c->own + image * np, np * 8, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    case 14: {   // debug: captured queue of PLI_DBG_RANK (dev aid)
      if (!need(4096 * 4)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      HIPCHK(hipMemcpy(dst, c->regScratch, 4096 * 4, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    case PLI_DBG_LSD_SIZES: {
      const int64_t np = (int64_t)P.LW * P.LH;
      if (!need(np * 4)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      HIPCHK(hipMemcpy(dst, c->lastSize + image * np, np * 4, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    case PLI_DBG_STEREO_SAD: {
      const int frame = image >> 1;
      const int64_t n = (int64_t)P.kpCap * 4;
      if (!need(2 * n)) return dst ? PLI_ERR_CAPACITY : PLI_OK;
      HIPCHK(hipMemcpy(dst, c->sad + (int64_t)frame * P.kpCap, n, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy((char*)dst + n, c->bestIdx + (int64_t)frame * P.kpCap, n, hipMemcpyDeviceToHost));
      return PLI_OK;
    }
    default:
      g_err = "unknown debug item";
      return PLI_ERR_INVALID;
  }
}

}  // extern "C"
