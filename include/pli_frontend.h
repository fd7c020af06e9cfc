/*
 * pli_frontend.h — C ABI of the MI355X-native point+line front-end.
 *
 * This is the drop-in boundary underneath PLI-SLAM's per-frame front-end
 * (Frame::Frame stereo ctor, reference src/Frame.cc:98-228).  The reference has
 * no FFI of its own: its boundary is a set of C++ call sites.  Every entry
 * point below names the reference call it replaces (file:line under the
 * reference tree).  The C++ adapters in pli_slam_amd/adapters/ keep the
 * reference's functor signatures and call these functions; INTEGRATION.md
 * shows the binding a maintainer adds to Frame.cc / Tracking.cc.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary; every function
 *     returns a pli_status (0 = ok, negative = error).
 *   - caller owns all buffers; outputs are caller-allocated with a capacity and
 *     the function returns the produced count.
 *   - "host" pointers are ordinary memory; "dev" pointers are HIP device
 *     memory of the context's device.
 *   - a context owns its HIP streams and all scratch for `max_frames` stereo
 *     frames of `width x height`.  Every entry point that takes a context is
 *     THREAD-SAFE: it holds the context's own lock for the whole call, so
 *     concurrent calls on ONE context are legal and run one after the other
 *     (the reference drives its four extractors from four std::threads,
 *     Frame.cc:128-135, and the adapters put the four on one context — the
 *     stereo matchers need both eyes' tables on the device).  Calls on
 *     DIFFERENT contexts run concurrently.  What the lock does not order is
 *     the caller's own protocol: pli_stereo_match_* read the tables of the LAST
 *     extract calls, so a second Frame must not start extracting on a context
 *     while the first Frame's stereo matchers have not run yet.
 *     pli_ctx_destroy must not race with other calls on that context.
 *   - Sharing a device.  Several contexts of ONE process may share a device
 *     freely.  The late rounds of the LSD tile relaxation run in a persistent
 *     kernel with a grid barrier (k_tx_tail) whose grid is sized for a device
 *     it has to itself; the library starts such a kernel only when the
 *     previous one of this process on that device has ended.  Several
 *     PROCESSES on one device (ranks sharing a GPU) should set PLI_TX_TAIL=0
 *     in their environment (the planned-rounds schedule, no spinning kernel):
 *     without it two processes' tail kernels can each hold part of the
 *     compute units the other needs; every barrier wait is bounded (about a
 *     second), the kernels then give up and the images that had not settled
 *     are redone by the sequential grower — results stay exact, the call
 *     stalls, and pli_lsd_round_stats out[2] counts the images.  A process
 *     that sees this happen once (the control blocks of a call come back with
 *     unsettled images although the persistent kernel ran) switches itself to
 *     the planned-rounds schedule on that device for the rest of its life and
 *     says so once on stderr: the stall is paid at most twice, not per call.
 */
#ifndef PLI_FRONTEND_H
#define PLI_FRONTEND_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t pli_status;
enum {
  PLI_OK = 0,
  PLI_ERR_INVALID = -1,     /* bad argument (null pointer, size mismatch, ...)            */
  PLI_ERR_EMPTY_IMAGE = -2, /* ORBextractor::operator() returns -1, ORBextractor.cc:1072  */
  PLI_ERR_CAPACITY = -3,    /* caller buffer too small                                     */
  PLI_ERR_HIP = -4,         /* HIP runtime failure (see pli_last_error)                    */
  PLI_ERR_NO_DEVICE = -5,   /* no usable gfx950 device: the product path has no CPU fallback */
  PLI_ERR_STATE = -6        /* call order violated (e.g. stereo match before extract)      */
};

/* cv::KeyPoint fields that leave the extractor (ORBextractor.cc:868-877,1131):
 * pt, size, angle, response, octave.  class_id is always -1 there. 24 bytes. */
typedef struct pli_keypoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
} pli_keypoint;

/* cv::line_descriptor::KeyLine, field for field
 * (Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:105-144). 68 bytes. */
typedef struct pli_keyline {
  float angle;
  int32_t class_id;
  int32_t octave;
  float pt_x, pt_y;
  float response;
  float size;
  float startPointX, startPointY, endPointX, endPointY;
  float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength;
  int32_t numOfPixels;
} pli_keyline;

/* Parameters of the path.  Sources: ORBextractor ctor (ORBextractor.cc:408),
 * Lineextractor ctor (LineExtractor.h:44), Tracking.cc:75-98,683-736,
 * Config.cpp:26-160, Examples/Stereo/Config/EuRoC.yaml. */
typedef struct pli_frontend_config {
  int32_t width, height;      /* image size the context is built for                     */
  int32_t max_frames;         /* stereo frames per batch the context can hold (>=1)       */
  /* ORB */
  int32_t orb_nfeatures;      /* ORBextractor.nFeatures   (EuRoC.yaml:91)  1200            */
  float   orb_scale_factor;   /* ORBextractor.scaleFactor (EuRoC.yaml:94)  1.2             */
  int32_t orb_nlevels;        /* ORBextractor.nLevels     (EuRoC.yaml:97)  8               */
  int32_t orb_ini_th_fast;    /* ORBextractor.iniThFAST   (EuRoC.yaml:103) 20              */
  int32_t orb_min_th_fast;    /* ORBextractor.minThFAST   (EuRoC.yaml:104) 7               */
  /* LSD / LBD */
  int32_t lsd_nfeatures;      /* EuRoC.yaml:156 (500); 0 = keep all                        */
  int32_t lsd_refine;         /* only 0 (LSD_REFINE_NONE) is on the reference path         */
  int32_t lsd_n_bins;         /* 1024                                                      */
  int32_t max_lines;          /* capacity for detected segments before the top-N cut      */
  double  min_line_length;    /* 0.025 (x min(W,H))                                        */
  double  lsd_scale;          /* 1.2                                                       */
  double  lsd_sigma_scale;    /* 0.6                                                       */
  double  lsd_quant;          /* 2.0                                                       */
  double  lsd_ang_th;         /* 22.5                                                      */
  double  lsd_log_eps;        /* 1.0  (unused with refine 0)                               */
  double  lsd_density_th;     /* 0.6  (unused with refine 0)                               */
  /* stereo points (Frame.cc:976-1154) */
  float   bf;                 /* Camera.bf (EuRoC.yaml:28) 47.90639384423901               */
  float   fx;                 /* Camera.fx (EuRoC.yaml:9)  435.2046959714599               */
  int32_t stereo_maxd_inf;    /* 0: maxD = bf/(bf/fx) as intended; 1: maxD = +inf
                                 (what an uninitialised mb gives)       */
  /* stereo lines (Frame.cc:1156-1307, LineMatcher.cpp:317-396, Config.cpp) */
  int32_t matching_s_ws;      /* 10                                                        */
  int32_t best_lr_matches;    /* 1                                                         */
  double  line_sim_th;        /* 0.75                                                      */
  double  stereo_overlap_th;  /* 0.75                                                      */
  double  min_ratio_12_l;     /* 0.9                                                       */
  double  ls_min_disp_ratio;  /* 0.7                                                       */
  double  min_disp;           /* 1.0                                                       */
  double  line_horiz_th;      /* 0.1                                                       */
  /* execution strategy of the LSD region grower (results are identical):
     0 = auto (by batch size: 3 below 1280 frames per call, 2 from there on — pli_capi.hip: RX_AUTO_IMAGES, a measured crossover; a context created for 1280 frames or more keeps the buffers of mode 3 for 1023 frames only),
     1 = rank-ordered relaxation, one region per lane / lane group (lsd_relax.hip),
     2 = sequential, one wave per image (line_kernels.hip: k_lsd_grow),
     3 = tile-sequential relaxation, one wave per 64x64 tile (lsd_tile.hip) */
  int32_t lsd_mode;
  /* Arithmetic the reference leaves to its toolchain (PLI_PARITY_*, below): which libm function an unqualified
     cos(float) resolves to, and which OpenCV generation's LineSegmentDetector is linked.  Oracle and kernels honour the
     same flags; DESIGN.md "Oracle" says what is known about each. */
  int32_t parity_flags;
} pli_frontend_config;

enum {
  /* cos/sin of the FLOAT angle in computeOrbDescriptor (ORBextractor.cc:110-111; the file says `using namespace std;`,
     so the float overload std::cos(float) = cosf is selected): 1 = cosf/sinf as glibc >= 2.28 computes them,
     0 = correctly rounded ((float)cos((double)x)).  Default 1. */
  PLI_PARITY_TRIG_F32_ORB = 1,
  /* the same choice for `cos(float(angle))` in OpenCV's lsd.cpp region_grow (inside namespace cv, no using-directive:
     the overload depends on whether the C++ <math.h> wrapper is in scope).  Default 0. */
  PLI_PARITY_TRIG_F32_LSD = 2,
  /* ... and for `cos(direction)` in BinaryDescriptor::computeLBD (binary_descriptor_custom.cpp:1130).  Default 0. */
  PLI_PARITY_TRIG_F32_LBD = 4,
  /* LineSegmentDetector on the CV_64FC1 copy of the image (OpenCV 3.0 .. 3.4: double Gaussian blur, double bilinear
     resize, double gradient) instead of the CV_8UC1 pipeline of the detector re-added in 4.5.x.  Default 1: the reference
     pins OpenCV 3.3.1 (README.md:18-20). */
  PLI_PARITY_LSD_F64 = 8
};

/* Fill `cfg` with the values of Examples/Stereo/Config/EuRoC.yaml for a w x h image
 * (parity_flags = PLI_PARITY_TRIG_F32_ORB | PLI_PARITY_LSD_F64). */
void pli_config_default(pli_frontend_config* cfg, int32_t width, int32_t height);

/* Capacities derived from a config (sizes of the per-eye tables). */
int32_t pli_kp_capacity(const pli_frontend_config* cfg);   /* >= nfeatures + 3*nlevels    */
int32_t pli_kl_capacity(const pli_frontend_config* cfg);   /* lsd_nfeatures or max_lines  */

/* ------------------------------------------------------------------------ */
/* Result table: one fixed-stride record per stereo frame, written on the    */
/* device by pli_batch_run.  Layout (all offsets from the record start, in  */
/* bytes) is described by pli_table_layout so that the caller can slice it   */
/* from any language; eye 0 = left, eye 1 = right.                           */
/* ------------------------------------------------------------------------ */
typedef struct pli_table_layout {
  int64_t record_bytes;       /* stride between consecutive frames                         */
  int32_t kp_cap, kl_cap;
  int64_t off_counts;         /* int32[8]: n_kp[2], n_kl[2], n_stereo_pts, n_stereo_lines, truncation flags, 1 reserved.
                                 Flags = 4 bytes: [0],[1] lines of eye 0 / 1: more segments passed the length cut than
                                 max_lines holds (the top-N selection did not see all of them); [2],[3] keypoints of eye
                                 0 / 1 cut at kp_cap.  The per-call entry points return PLI_ERR_CAPACITY for them. */
  int64_t off_kp[2];          /* pli_keypoint[kp_cap]                                       */
  int64_t off_desc[2];        /* uint8[kp_cap][32]   (mDescriptors / mDescriptorsRight)     */
  int64_t off_uright;         /* float[kp_cap]       (mvuRight)                             */
  int64_t off_depth;          /* float[kp_cap]       (mvDepth)                              */
  int64_t off_kl[2];          /* pli_keyline[kl_cap]                                        */
  int64_t off_ldesc[2];       /* uint8[kl_cap][32]   (mDescriptors_Line / ..Right_Line)     */
  int64_t off_disp;           /* float[kl_cap][2]    (mvDisparity_l)                        */
  int64_t off_le;             /* double[kl_cap][3]   (mvle_l)                               */
} pli_table_layout;

typedef struct pli_ctx pli_ctx;

/* Create / destroy.  Replaces the construction of the four extractors in
 * Tracking.cc:87-98,743-749 (tables, rBRIEF pattern, LBD weights go to the device). */
pli_status pli_ctx_create(const pli_frontend_config* cfg, int32_t device, pli_ctx** out);
void       pli_ctx_destroy(pli_ctx* ctx);
const char* pli_last_error(void);                    /* thread-local message             */
pli_status pli_ctx_layout(const pli_ctx* ctx, pli_table_layout* out);
/* Use an existing hipStream_t (e.g. the caller framework's current stream). NULL = own stream. */
pli_status pli_ctx_set_stream(pli_ctx* ctx, void* hip_stream);
pli_status pli_ctx_sync(pli_ctx* ctx);

/* ------------------------------------------------------------------------ */
/* Batch (throughput) path: the whole Frame::Frame front-end for `nframes`   */
/* stereo pairs in one go — ExtractORB x2, ExtractLine x2 (Frame.cc:128-135),*/
/* ComputeStereoMatches_Lines (:160), ComputeStereoMatches (:163).           */
/* Images: device pointers, u8, row stride `stride`, consecutive frames      */
/* `frame_stride` bytes apart (no alignment needed).  Output: `dev_table` =  */
/* nframes records, 16-byte aligned (PLI_ERR_INVALID otherwise).             */
/* Asynchronous on the context stream; pli_ctx_sync() to wait.               */
/* Monocular / RGB-D colour streams (Frame.cc:231,334: one image per frame):  */
/* pass frame 0 as `left`, frame 1 as `right`, frame_stride = two frames,     */
/* nframes = half the stream, stages = PLI_RUN_ORB | PLI_RUN_LINES — frames   */
/* 2i and 2i+1 then fill the two eye slots of record i.                       */
/* ------------------------------------------------------------------------ */
enum {
  PLI_RUN_ORB = 1, PLI_RUN_LINES = 2, PLI_RUN_STEREO_POINTS = 4, PLI_RUN_STEREO_LINES = 8,
  PLI_RUN_ALL = 15
};
pli_status pli_batch_run(pli_ctx* ctx, int32_t nframes,
                         const uint8_t* dev_left, const uint8_t* dev_right,
                         int64_t stride, int64_t frame_stride,
                         uint32_t stages, void* dev_table);
/* Same with host images and a host table (does the H2D / D2H copies, synchronous). */
pli_status pli_batch_run_host(pli_ctx* ctx, int32_t nframes,
                              const uint8_t* left, const uint8_t* right,
                              int64_t stride, int64_t frame_stride,
                              uint32_t stages, void* table);

/* Pipelined host entry point (throughput with images in HOST memory): pli_batch_submit_host returns
 * after enqueueing the H2D copies (own copy stream), the kernels and the D2H copy of the table (second copy stream);
 * two submits may be in flight, so the copies of batch i+1 / i-1 overlap the kernels of batch i; a third submit first
 * waits for the oldest.  `left`, `right` and `table` should be pinned (pli_host_alloc, or memory the caller registered
 * with hipHostRegister) — pageable memory works but makes the copies synchronous.  pli_batch_wait(ctx, 0) waits for
 * the oldest outstanding submit (its table is then complete), pli_batch_wait(ctx, 1) for all of them.
 * Footprint: the first submit allocates, beside the context's own buffers, two device staging slots of max_frames frames each
 * (2 x width x height x max_frames bytes of images + max_frames table records per slot: 0.9 MB per frame at 752x480) and keeps
 * them until pli_ctx_destroy. */
pli_status pli_host_alloc(size_t bytes, void** out);
void       pli_host_free(void* p);
pli_status pli_batch_submit_host(pli_ctx* ctx, int32_t nframes,
                                 const uint8_t* left, const uint8_t* right,
                                 int64_t stride, int64_t frame_stride,
                                 uint32_t stages, void* table);
pli_status pli_batch_wait(pli_ctx* ctx, int32_t all);

/* ------------------------------------------------------------------------ */
/* Per-call drop-ins (host buffers, synchronous).                            */
/* ------------------------------------------------------------------------ */

/* The front-end of ONE Frame in one submission: what Frame::Frame (Frame.cc:128-163) does with four extractor calls on four
 * threads and two member calls — ExtractORB x2, ExtractLine x2, ComputeStereoMatches_Lines, ComputeStereoMatches — for a host
 * stereo pair.  `record` receives the frame's table record (pli_table_layout, record_bytes); the context is left as if
 * pli_orb_extract / pli_line_extract had run for both eyes (pli_last_counts, pli_orb_pyramid_level), and
 * pli_stereo_match_points / pli_stereo_match_lines hand out the matches of this submission without running again, until the next
 * per-call extraction.  The adapters funnel the four concurrent operator() calls of a Frame into one call of this
 * (orbslam_adapters.hpp: FrameFusion): 2.7 ms per Frame instead of 6.5 ms for the four calls one after the other. */
pli_status pli_frame_extract(pli_ctx* ctx, const uint8_t* left, const uint8_t* right, int32_t w, int32_t h,
                             int64_t stride_left, int64_t stride_right, void* record);

/* ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea)
 * ORBextractor.h:61-63 / ORBextractor.cc:1068-1150, with vLappingArea = {0,0}
 * as Frame::ExtractORB passes (Frame.cc:484-491).  `eye` selects which of the
 * two device-resident table sets (0 = left extractor, 1 = right extractor) is
 * filled; they feed pli_stereo_match_*.  Returns PLI_ERR_EMPTY_IMAGE for a
 * null/zero-sized image like the reference's -1. */
pli_status pli_orb_extract(pli_ctx* ctx, int32_t eye,
                           const uint8_t* img, int32_t w, int32_t h, int64_t stride,
                           pli_keypoint* kp, int32_t cap, uint8_t* desc /* cap x 32 */,
                           int32_t* n);

/* ORBextractor::mvImagePyramid[level] (public member, ORBextractor.h:87) copied
 * to `dst` (w*h bytes, no border).  Valid after pli_orb_extract on that eye. */
pli_status pli_orb_pyramid_level(pli_ctx* ctx, int32_t eye, int32_t level,
                                 uint8_t* dst, int64_t dst_bytes, int32_t* w, int32_t* h);

/* Lineextractor::operator()(image, mask, keylines, descriptors_line)
 * LineExtractor.h:49-51 / LineExtractor.cc:31-70. */
pli_status pli_line_extract(pli_ctx* ctx, int32_t eye,
                            const uint8_t* img, int32_t w, int32_t h, int64_t stride,
                            pli_keyline* kl, int32_t cap, uint8_t* desc /* cap x 32 */,
                            int32_t* n);

/* Diagnostics of the LSD relaxation schedule (lsd_mode 1 / 3; results never depend on it).  After the first call on a context
 * the relaxation rounds are launched WITHOUT a host look: as many as the slowest image of an earlier call needed plus a margin,
 * an image that is not at its fixed point after them is redone on the device by the sequential grower (exact, slow), and no
 * entry point drains the stream in the middle of a call.  (The tile relaxation needs no plan: its rounds from the third or fourth on
 * run in one persistent launch that ends when every image is at its fixed point.)  Synchronises the context.
 *   out[0] rounds the last call launched without looking; -1: the tile relaxation's persistent tail kernel ran the late rounds and
 *          stopped by itself at the fixed point (the default schedule of lsd_mode 3 / auto); 0: the host looked at the state
 *   out[1] rounds the slowest image of that call needed (-1: at least one image did not settle and took the slow path)
 *   out[2] images that took the slow path since the context was created
 *   out[3] the round count the next call plans from */
pli_status pli_lsd_round_stats(pli_ctx* ctx, int32_t out[4]);

/* Words of arena per scaled LSD pixel the context's relaxation got (region pixel lists and queue overflow blocks): 16 unless the
 * context is large or the device was short of free memory when it was created (then 3..15: results are unchanged, but batches of
 * long parallel structures may run out of it and take the slow device-side fallback, counted in pli_lsd_round_stats out[2]; the
 * library says so once on stderr); 0 for lsd_mode 2. */
pli_status pli_lsd_arena_words(pli_ctx* ctx, int32_t* words_per_pixel);

/* Self-test of the device the context runs on (not a reference function): the largest absolute error of the hardware cosine / sine
 * (v_cos_f32 / v_sin_f32 of angle / 360) against cos / sin of EVERY float angle in [0, 360] degrees.  Round 1 of the LSD tile
 * relaxation runs its vector filter on those (csrc/lsd_tile.hip "HOT RECORDS"); its error budget assumes 4e-6, and the GPU suite
 * asserts that this call reports less.  About a second of device time. */
pli_status pli_selftest_hot_trig(pli_ctx* ctx, double* max_abs_err);

/* The stereo rig of Frame::ComputeStereoMatches (Frame.cc:1005-1008: minZ = mb, maxD = mbf / minZ, depth = mbf / disparity):
 * replaces pli_frontend_config.bf / .fx of an existing context (the extractors' constructors, which create the context in
 * the adapters, do not know the camera; the Frame does: mbf and mK(0,0)).  A no-op when the values are the ones in use. */
pli_status pli_set_stereo_camera(pli_ctx* ctx, float bf, float fx);

/* Sizes of the tables the context holds from the last per-call extractions: counts[0], [1] = keypoints of eye 0 / 1
 * (mvKeys.size(), mvKeysRight.size()), counts[2], [3] = keylines of eye 0 / 1 (mvKeys_Line.size(),
 * mvKeysRight_Line.size()); -1 where that extractor has not run on this context.  The Frame-level stereo matchers of
 * the adapters check their vectors against these before they read the device tables (Frame.cc:976, :1156). */
pli_status pli_last_counts(pli_ctx* ctx, int32_t counts[4]);

/* Frame::ComputeStereoMatches() Frame.cc:976-1154 on the tables + pyramids left
 * on the device by the last pli_orb_extract(eye 0) / (eye 1).
 * uright/depth: N floats each (N = left keypoint count), -1 = no stereo. */
pli_status pli_stereo_match_points(pli_ctx* ctx, float* uright, float* depth, int32_t cap);

/* Frame::ComputeStereoMatches_Lines() Frame.cc:1156-1259 on the line tables of
 * the last pli_line_extract(eye 0) / (eye 1).
 * disp: N_l x 2 floats (mvDisparity_l, -1 = mono); le: N_l x 3 doubles (mvle_l). */
pli_status pli_stereo_match_lines(pli_ctx* ctx, float* disp, double* le, int32_t cap);

/* ORBmatcher::DescriptorDistance ORBmatcher.cc:2495-2511 / distance()
 * LineMatcher.cpp:231-247, batched: dist[i] = hamming(a[i], b[i]), 32-byte rows. */
pli_status pli_descriptor_distance(pli_ctx* ctx, const uint8_t* a, const uint8_t* b,
                                   int32_t n, int32_t* dist);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) as used by matchNNR
 * LineMatcher.cpp:139-159: per query the two smallest distances, ties to the
 * lower train index.  idx/dist: nq x 2 (idx -1 / dist INT32_MAX when nt < 2). */
pli_status pli_hamming_knn2(pli_ctx* ctx, const uint8_t* q, int32_t nq,
                            const uint8_t* t, int32_t nt, int32_t* idx, int32_t* dist);

/* int match(desc1, desc2, nnr, matches_12) LineMatcher.cpp:201-229 with
 * Config::bestLRMatches() = best_lr_matches of the context: ratio test both
 * ways + mutual check.  matches_12: n1 ints (-1 = none).  *nmatches = return value. */
pli_status pli_match_lines(pli_ctx* ctx, const uint8_t* desc1, int32_t n1,
                           const uint8_t* desc2, int32_t n2, float nnr,
                           int32_t* matches_12, int32_t* nmatches);

/* Core of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, match12)
 * ORBmatcher.cc:2179-2323: the caller projects LastFrame's map points with the
 * predicted pose (SLAM state stays on the host) and passes, per query i in
 * LastFrame order: projected (u,v), search radius, octave window
 * [min_level,max_level] (max_level < 0 = unbounded as GetFeaturesInArea
 * Frame.cc:774-843), the map point descriptor and the predicted right
 * coordinate ur = u - bf*invz (compared with mvuRight[i2] whenever that is > 0,
 * :2259-2265).  Queries with valid == 0 (no map point / outlier / behind the
 * camera) are skipped.  The current frame is described by its keypoints, descriptors,
 * mvuRight and the image bounds used by the 64x48 grid (mnMinX..mnMaxY).
 * "Not available" is the reference's test :2255-2257 — CurrentFrame.mvpMapPoints[i2] is set AND has
 * Observations() > 0 — in both of its uses: cur_occupied[i2] != 0 (may be NULL) marks the keypoints that hold
 * such a map point BEFORE the call, and a match made in this call takes its keypoint away from the queries
 * behind it unless the query says its own map point has no observations (valid = 1 | PLI_PROJ_NO_OBSERVATIONS:
 * the temporal points Tracking::UpdateLastFrame creates in localisation mode; such a keypoint can be matched
 * again and the last writer holds it).  TH_HIGH = 100, the 30-bin rotation histogram and
 * ComputeThreeMaxima (:2449-2490) run on the device.
 * best_idx2[i] = matched current keypoint or -1 after the rotation filter; raw_idx2 (may be NULL) = the same
 * before it (a maintainer replays :2280-2282 from raw_idx2 in query order and :2315-2317 for the queries with
 * raw_idx2[i] >= 0 > best_idx2[i], as the adapter does).  *nmatches as the reference. */
#define PLI_PROJ_NO_OBSERVATIONS 2
typedef struct pli_proj_query {
  float u, v, radius, ur;
  int32_t min_level, max_level;
  float angle;          /* LastFrame.mvKeysUn[i].angle */
  int32_t valid;
} pli_proj_query;
pli_status pli_search_by_projection(pli_ctx* ctx,
                                    const pli_proj_query* q, const uint8_t* qdesc, int32_t nq,
                                    const pli_keypoint* cur_kp, const uint8_t* cur_desc,
                                    const float* cur_uright, const uint8_t* cur_occupied, int32_t ncur,
                                    float min_x, float max_x, float min_y, float max_y,
                                    int32_t check_orientation,
                                    int32_t* best_idx2, int32_t* raw_idx2, int32_t* nmatches);

/* --- Frame-to-frame track matching of a batch (BASELINE config 3), on the device tables of pli_batch_run ---
 * Frame i (i >= 1) of the batch against frame i-1, as Tracking::TrackWithMotionModel does per frame:
 *   points: ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, match12) ORBmatcher.cc:2179-2323, the last
 *           frame's stereo points (mvDepth > 0) standing for its map points as Tracking::UpdateLastFrame creates them
 *           (Frame::UnprojectStereo, Frame.cc:1334-1350);
 *   lines:  match(LastFrame.mDescriptors_Line, CurrentFrame.mDescriptors_Line, nnr, matches_12) LineMatcher.cpp:201-229
 *           (Tracking.cc:3058), left eye.
 * dev_poses: nframes x 12 floats on the device, mTcw (world -> camera, row major 3x4) of every frame — the motion-model
 * prediction for the current frame, the optimised pose for the last.  dev_track: nframes records of pli_track_layout
 * (record 0 is left untouched): counts = {nq = last N, point matches, n1 = last line count, line matches},
 * best_idx2[j] = current keypoint matched to last keypoint j or -1, matches_12[i] = current line of last line i or -1.
 * Asynchronous on the context stream. */
typedef struct pli_track_params {
  float fx, fy, cx, cy, bf;          /* Camera.fx/fy/cx/cy/bf (EuRoC.yaml:9-28)                        */
  float th;                          /* search window: 15 stereo / 7 mono in TrackWithMotionModel       */
  float min_x, max_x, min_y, max_y;  /* mnMinX..mnMaxY of the frame grid                               */
  int32_t mono;                      /* bMono                                                          */
  int32_t check_orientation;         /* ORBmatcher::mbCheckOrientation                                 */
  float nnr_lines;                   /* minRatio12L                                                    */
  int32_t reserved;
} pli_track_params;
typedef struct pli_track_layout {
  int64_t record_bytes;
  int64_t off_counts;                /* int32[4]                                                       */
  int64_t off_best;                  /* int32[kp_cap]                                                  */
  int64_t off_lines;                 /* int32[kl_cap]                                                  */
  int32_t kp_cap, kl_cap;
} pli_track_layout;
pli_status pli_track_layout_get(const pli_ctx* ctx, pli_track_layout* out);
pli_status pli_batch_track(pli_ctx* ctx, int32_t nframes, const void* dev_table, const float* dev_poses,
                           const pli_track_params* params, void* dev_track);

/* --- the driver's rectification fused into the ingest ---
 * Replaces cv::remap(imLeft, imLeftRect, M1l, M2l, cv::INTER_LINEAR) / (imRight, ...) of
 * Examples/Stereo/stereo_euroc.cc:166-167 (maps from cv::initUndistortRectifyMap(..., CV_32F, M1, M2) :117-118).
 * mapx / mapy: width*height floats each (host), the CV_32FC1 maps of one eye; afterwards every image of that eye
 * handed to pli_batch_run / pli_batch_run_host / pli_orb_extract / pli_line_extract is the RAW image and level 0
 * of the pyramid is its rectification (bilinear, 1/32-px coordinates, constant-0 border, OpenCV 3.3.1 fixed point).
 * NULL, NULL removes the maps of that eye. */
pli_status pli_set_rectify_maps(pli_ctx* ctx, int32_t eye, const float* mapx, const float* mapy);

/* --- SURVEY.md §8(f) row 4 (RGB-D front-end): Frame::ComputeStereoFromRGBD(imDepth), Frame.cc:1309-1331 ---
 * depth: CV_32F depth image registered to the left image (already scaled by mDepthMapFactor, Tracking.cc), row
 * stride in floats.  For every left keypoint of the last pli_orb_extract(eye 0): d = depth(trunc(v), trunc(u));
 * d > 0 -> mvDepth = d, mvuRight = x - bf/d, else both -1 (the rectified pipeline: mvKeysUn == mvKeys). */
pli_status pli_stereo_from_depth(pli_ctx* ctx, const float* depth, int64_t stride_floats, float* uright, float* depth_out,
                                 int32_t cap);

/* --- SURVEY.md §8(f) row 4 (fisheye stereo front-end, Frame.cc:1484-1576) ---
 * ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea) with a real lapping area
 * (ORBextractor.cc:1135-1144; Frame::ExtractORB passes mpCamera->mvLappingArea, Frame.cc:488,490): keypoints whose
 * level-0 x lies in [lap0, lap1] fill the table from the back in visiting order, all others from the front.
 * *n = keypoints, *n_mono = the reference's return value (index of the first lapping-area keypoint = monoLeft /
 * monoRight, Frame.cc:1521-1522).  The device table of that eye keeps this order for pli_stereo_fisheye. */
pli_status pli_orb_extract_lapping(pli_ctx* ctx, int32_t eye,
                                   const uint8_t* img, int32_t w, int32_t h, int64_t stride,
                                   int32_t lap0, int32_t lap1,
                                   pli_keypoint* kp, int32_t cap, uint8_t* desc /* cap x 32 */,
                                   int32_t* n, int32_t* n_mono);

/* KannalaBrandt8::mvParameters (include/CameraModels/KannalaBrandt8.h): fx, fy, cx, cy, k0..k3. */
typedef struct pli_kb8_camera { float fx, fy, cx, cy, k0, k1, k2, k3; } pli_kb8_camera;

/* Frame::ComputeStereoFishEyeMatches() Frame.cc:1577-1618 on the tables left on the device by the last
 * pli_orb_extract_lapping(eye 0) / (eye 1): BFMatcher(NORM_HAMMING).knnMatch(k = 2) of the lapping-area descriptors,
 * Lowe ratio 0.7, KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:334-402: parallax
 * cos <= 0.9998, SVD triangulation, positive depth in both cameras, reprojection error <= 5.991 mvLevelSigma2[octave]
 * in both), depth > 0.0001.  Rlr: 3x3 row major (mRlr), tlr: 3 (mtlr) = mTlr of Frame.cc:1539-1540.
 * l2r: mvLeftToRightMatch (Nleft ints, -1 = none), r2l: mvRightToLeftMatch (Nright), depth: mvDepth (Nleft, -1),
 * p3d: mvStereo3Dpoints (Nleft x 3 floats, camera-1 coordinates, zeros where unset); *nmatches = nMatches. */
pli_status pli_stereo_fisheye(pli_ctx* ctx, const pli_kb8_camera* cam1, const pli_kb8_camera* cam2,
                              const float* Rlr, const float* tlr,
                              int32_t* l2r, int32_t cap_left, int32_t* r2l, int32_t cap_right,
                              float* depth, float* p3d, int32_t* nmatches);

/* The same on caller tables: kp_left / desc_left = mvKeys / mDescriptors (Nleft rows, lapping-area keypoints from row
 * mono_left on), kp_right / desc_right = mvKeysRight / mDescriptorsRight; octaves index the context's mvLevelSigma2. */
pli_status pli_stereo_fisheye_tables(pli_ctx* ctx, const pli_keypoint* kp_left, const uint8_t* desc_left, int32_t nleft,
                                     int32_t mono_left, const pli_keypoint* kp_right, const uint8_t* desc_right,
                                     int32_t nright, int32_t mono_right,
                                     const pli_kb8_camera* cam1, const pli_kb8_camera* cam2,
                                     const float* Rlr, const float* tlr,
                                     int32_t* l2r, int32_t* r2l, float* depth, float* p3d, int32_t* nmatches);

/* --- local-map tracking (Tracking::SearchLocalPointsAndLines, Tracking.cc:3854,3882) --- */

/* Core of ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, ...)
 * ORBmatcher.cc:44-143 (left / rectified-stereo branch, F.Nleft == -1).  Per map point in view the caller
 * passes (u,v) = mTrackProjX/Y, radius = r*mvScaleFactors[nPredictedLevel] (r from RadiusByViewingCos, times th),
 * ur = mTrackProjXR, min_level = nPredictedLevel-1, max_level = nPredictedLevel, valid = in view, not bad, not too far.
 * cur_occupied[i] != 0 marks current keypoints that already hold a map point with observations (may be NULL).
 * Best and second-best Hamming distance in GetFeaturesInArea order, TH_HIGH = 100, ratio test only when both are
 * on the same pyramid level (mfNNratio = nnratio); accepted keypoints become occupied for the following queries.
 * best_idx2[i] = matched keypoint or -1; *nmatches = return value. */
pli_status pli_search_local_map(pli_ctx* ctx, const pli_proj_query* q, const uint8_t* qdesc, int32_t nq,
                                const pli_keypoint* cur_kp, const uint8_t* cur_desc, const float* cur_uright,
                                const uint8_t* cur_occupied, int32_t ncur,
                                float min_x, float max_x, float min_y, float max_y, float nnratio,
                                int32_t* best_idx2, int32_t* nmatches);

/* The same function for a frame of two fisheye cameras (F.Nleft != -1), ORBmatcher.cc:44-214 in full.  Per map point i:
 * q_left[i] as above (valid = mbTrackInView etc.; no mvuRight gate in this branch), then — unless the left ratio test failed,
 * whose `continue` leaves the map point (:126) — q_right[i]: (u,v) = mTrackProjXR/YR, radius = RadiusByViewingCos(
 * mTrackViewCosR)*mvScaleFactors[mnTrackScaleLevelR] (no th, :148-151), levels mnTrackScaleLevelR-1..mnTrackScaleLevelR,
 * valid = mbTrackInViewR && mnTrackScaleLevelR != -1, searched in mGridRight / mvKeysRight.  F.mvpMapPoints is one array
 * of Nleft + Nright slots: occ_left / occ_right (may be NULL) mark the slots that hold a map point with observations,
 * left_to_right / right_to_left are mvLeftToRightMatch / mvRightToLeftMatch (-1 = none) — a match is also written to the
 * keypoint's stereo partner (:133-137, :201-205).  mp_left[k] / mp_right[k] = index of the map point this call left in the
 * slot, or -1; *nmatches = return value.  nleft + nright <= 15360. */
pli_status pli_search_local_map_fisheye(pli_ctx* ctx, const pli_proj_query* q_left, const pli_proj_query* q_right,
                                        const uint8_t* qdesc, int32_t nq,
                                        const pli_keypoint* kp_left, const uint8_t* desc_left, const uint8_t* occ_left,
                                        const int32_t* left_to_right, int32_t nleft,
                                        const pli_keypoint* kp_right, const uint8_t* desc_right, const uint8_t* occ_right,
                                        const int32_t* right_to_left, int32_t nright,
                                        float min_x, float max_x, float min_y, float max_y, float nnratio,
                                        int32_t* mp_left, int32_t* mp_right, int32_t* nmatches);

/* int match(const vector<MapLine*>&, Frame&, nnr, matches_12) LineMatcher.cpp:161-171: one-directional matchNNR
 * of the local map lines' descriptors against the frame's (the reference returns before its mutual check). */
pli_status pli_match_nnr(pli_ctx* ctx, const uint8_t* desc1, int32_t n1, const uint8_t* desc2, int32_t n2, float nnr,
                         int32_t* matches_12, int32_t* nmatches);

/* --- bag-of-words of a frame (Frame::ComputeBoW, Frame.cc:858-870) ---
 * DBoW2::TemplatedVocabulary<cv::Mat, FORB> (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h): a k-ary tree of 256-bit
 * descriptors.  pli_vocab_create takes the node list exactly as loadFromTextFile (:1350-1433) reads it from
 * ORBvoc.txt: node i+1 of the file (node 0 is the root) has parent[i], its 32 descriptor bytes, its weight and its
 * "is a word" flag; children keep file order; word ids are assigned to flagged nodes in file order. */
typedef struct pli_vocab pli_vocab;
pli_status pli_vocab_create(pli_ctx* ctx, int32_t k, int32_t L, int32_t nnodes, const int32_t* parent,
                            const uint8_t* is_leaf, const uint8_t* desc, const double* weight, pli_vocab** out);
void pli_vocab_destroy(pli_vocab* v);

/* The per-feature part of transform(features, BowVector&, FeatureVector&, levelsup) (:1139-1208): the descent
 * transform(feature, id, weight, &nid, levelsup) (:1230-1272) of every descriptor — at each level the child with the
 * smallest Hamming distance, the first one on ties — giving word_id[i], weight[i] (the word's idf weight; 0 =
 * stopped word) and node_id[i] (the ancestor at level L - levelsup, 0 if that is the root).  The accumulation into
 * the two std::maps and the L1 normalisation are done by the caller in feature order (adapters/pli_cpp.hpp does it
 * with the reference's own arithmetic). */
pli_status pli_bow_transform(pli_ctx* ctx, const pli_vocab* vocab, const uint8_t* desc, int32_t n, int32_t levelsup,
                             int32_t* word_id, double* weight, int32_t* node_id);

/* ------------------------------------------------------------------------ */
/* Measurement hooks (bench.py / tests only).                                */
/* ------------------------------------------------------------------------ */
/* Tracing: with PLI_ROCTX=1 in the environment when the library is loaded, every entry point, every stage of a call
 * (ingest, ORB chain, line chain, stereo matchers) and every kernel launch is bracketed by roctxRangePush / roctxRangePop
 * (rocprofiler-sdk's marker library, found at run time), so that `rocprofv3 --kernel-trace --marker-trace -- <program>` shows which
 * call and stage a kernel belongs to.  Without the switch nothing is loaded and nothing is pushed.  Returns the number of ranges
 * pushed so far by this process (0 when tracing is off). */
int64_t pli_trace_ranges(void);

/* When enabled every kernel launch of the context is bracketed by HIP events on
 * the context stream; pli_prof_report writes "name calls total_ms\n" lines. */
pli_status pli_prof_enable(pli_ctx* ctx, int32_t on);
pli_status pli_prof_reset(pli_ctx* ctx);
pli_status pli_prof_report(pli_ctx* ctx, char* buf, int64_t buf_bytes);

/* Intermediate products, for parity tests against the oracle stage by stage.
 * `what` is one of PLI_DBG_*; data are copied to `dst` (host).  *out_bytes =
 * bytes written.  image = frame*2 + eye. */
enum {
  PLI_DBG_PYRAMID_LEVEL = 1,   /* arg = level; u8 w*h                                       */
  PLI_DBG_BLUR_LEVEL = 2,      /* arg = level; u8 w*h (7x7 sigma 2)                         */
  PLI_DBG_FAST_CANDIDATES = 3, /* arg = level; int32 count then {int32 x,y,score} records, reference order */
  PLI_DBG_LEVEL_KEYPOINTS = 4, /* arg = level; int32 count then {int32 x,y,score} after the quadtree */
  PLI_DBG_LSD_SCALED = 5,      /* u8 W'*H' (blur 0.6 + x1.2 resize)                         */
  PLI_DBG_LSD_ANGLE = 6,       /* float W'*H' degrees, -1024 = NOTDEF                       */
  PLI_DBG_LSD_SEGMENTS = 7,    /* int32 count then float[4] x1,y1,x2,y2 in detection order  */
  PLI_DBG_LBD_DXDY = 8,        /* int16 dx[w*h] then int16 dy[w*h]                          */
  PLI_DBG_LSD_ORDER = 9,       /* int32 count then int32 pixel index of every seed in visiting order */
  PLI_DBG_LBD_FLOAT = 10,      /* float[kl_cap][72] LBD band vector before binarisation     */
  PLI_DBG_STEREO_SAD = 11,     /* int32 sad[kp_cap] (-1 = none) then int32 bestIdxR[kp_cap] of the frame */
  PLI_DBG_LSD_OWNER = 12,      /* int32 rounds, then int32 owner rank per scaled pixel (0x7fffffff = undefined) */
  PLI_DBG_LSD_SIZES = 13       /* int32 region size recorded by the last grower of every seed rank */
};
/* Keep the extra intermediates (LSD angle map, LBD float vectors, stereo best index) during runs. */
pli_status pli_debug_enable(pli_ctx* ctx, int32_t on);
pli_status pli_debug_fetch(pli_ctx* ctx, int32_t image, int32_t what, int32_t arg,
                           void* dst, int64_t dst_bytes, int64_t* out_bytes);

const char* pli_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PLI_FRONTEND_H */
