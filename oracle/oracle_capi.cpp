// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points (ctypes) over the CPU
// restatement of the reference front-end.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library; the product library
// (pli_slam_amd/csrc) never links or calls it.
#include "orb_oracle.hpp"
#include "line_oracle.hpp"
#include "match_oracle.hpp"
#include <memory>
#include <limits>

using namespace orc;

namespace {

struct Eye {
  std::unique_ptr<OrbExtractor> orb;
  std::vector<pli_keypoint> kps;
  std::vector<uint8_t> desc;
  std::vector<pli_keyline> kls;
  std::vector<uint8_t> ldesc;
  LineDebug ld;
  int imgW = 0, imgH = 0;
};

struct Frame {
  pli_frontend_config cfg;
  Eye eye[2];
  std::vector<float> uright, depth;
  std::vector<int> bestIdx, sad;
  std::vector<float> disp;
  std::vector<double> le;
  std::vector<int> lineMatches;
};

Img8 wrap(const uint8_t* img, int w, int h, int64_t stride) {
  Img8 I(w, h);
  for (int y = 0; y < h; ++y) std::memcpy(I.row(y), img + (size_t)y * stride, w);
  return I;
}

LineExtractorCfg lineCfg(const pli_frontend_config& c) {
  LineExtractorCfg L;
  L.lsd_nfeatures = c.lsd_nfeatures;
  L.min_line_length = c.min_line_length;
  L.lsd.refine = c.lsd_refine;
  L.lsd.scale = c.lsd_scale;
  L.lsd.sigma_scale = c.lsd_sigma_scale;
  L.lsd.quant = c.lsd_quant;
  L.lsd.ang_th = c.lsd_ang_th;
  L.lsd.log_eps = c.lsd_log_eps;
  L.lsd.density_th = c.lsd_density_th;
  L.lsd.n_bins = c.lsd_n_bins;
  L.lsd.input_f64 = (c.parity_flags & PLI_PARITY_LSD_F64) != 0;
  L.lsd.trig_f32 = (c.parity_flags & PLI_PARITY_TRIG_F32_LSD) != 0;
  L.lbd_trig_f32 = (c.parity_flags & PLI_PARITY_TRIG_F32_LBD) != 0;
  return L;
}

LineMatchCfg matchCfg(const pli_frontend_config& c) {
  LineMatchCfg M;
  M.matching_s_ws = c.matching_s_ws;
  M.best_lr_matches = c.best_lr_matches != 0;
  M.line_sim_th = c.line_sim_th;
  M.stereo_overlap_th = c.stereo_overlap_th;
  M.min_ratio_12_l = c.min_ratio_12_l;
  M.ls_min_disp_ratio = c.ls_min_disp_ratio;
  M.min_disp = c.min_disp;
  M.line_horiz_th = c.line_horiz_th;
  return M;
}

}  // namespace

extern "C" {

void* orc_frame_create(const pli_frontend_config* cfg) {
  Frame* f = new Frame();
  f->cfg = *cfg;
  for (int e = 0; e < 2; ++e)
    f->eye[e].orb.reset(new OrbExtractor(cfg->orb_nfeatures, cfg->orb_scale_factor, cfg->orb_nlevels,
                                         cfg->orb_ini_th_fast, cfg->orb_min_th_fast));
  for (int e = 0; e < 2; ++e) f->eye[e].orb->trigF32 = (cfg->parity_flags & PLI_PARITY_TRIG_F32_ORB) != 0;
  return f;
}
void orc_frame_destroy(void* h) { delete (Frame*)h; }

int orc_features_per_level(void* h, int* out) {
  Frame* f = (Frame*)h;
  for (int i = 0; i < f->cfg.orb_nlevels; ++i) out[i] = f->eye[0].orb->mnFeaturesPerLevel[i];
  return f->cfg.orb_nlevels;
}
int orc_umax(void* h, int* out) {
  Frame* f = (Frame*)h;
  for (int i = 0; i < 16; ++i) out[i] = f->eye[0].orb->umax[i];
  return 16;
}

int orc_orb_extract(void* h, int eye, const uint8_t* img, int w, int hgt, int64_t stride) {
  Frame* f = (Frame*)h;
  Eye& E = f->eye[eye];
  if (!img || w <= 0 || hgt <= 0) return -1;
  Img8 I = wrap(img, w, hgt, stride);
  E.imgW = w; E.imgH = hgt;
  return (*E.orb)(I, E.kps, E.desc);
}
int orc_get_keypoints(void* h, int eye, pli_keypoint* kp, uint8_t* desc) {
  Eye& E = ((Frame*)h)->eye[eye];
  if (kp) std::memcpy(kp, E.kps.data(), E.kps.size() * sizeof(pli_keypoint));
  if (desc) std::memcpy(desc, E.desc.data(), E.desc.size());
  return (int)E.kps.size();
}
int orc_get_pyramid(void* h, int eye, int level, int blurred, uint8_t* dst, int* w, int* hgt) {
  Eye& E = ((Frame*)h)->eye[eye];
  const Img8& I = blurred ? E.orb->mvBlurred[level] : E.orb->mvImagePyramid[level];
  *w = I.w; *hgt = I.h;
  if (dst) std::memcpy(dst, I.d.data(), I.d.size());
  return 0;
}
// which: 0 = FAST candidates (vToDistributeKeys), 1 = after DistributeOctTree. records of 3 ints.
int orc_get_level_points(void* h, int eye, int level, int which, int* dst, int cap) {
  Eye& E = ((Frame*)h)->eye[eye];
  const std::vector<OrbCand>& v = which ? E.orb->dbg[level].selected : E.orb->dbg[level].candidates;
  int n = (int)v.size();
  for (int i = 0; i < n && i < cap; ++i) { dst[3 * i] = v[i].x; dst[3 * i + 1] = v[i].y; dst[3 * i + 2] = v[i].score; }
  return n;
}

int orc_line_extract(void* h, int eye, const uint8_t* img, int w, int hgt, int64_t stride) {
  Frame* f = (Frame*)h;
  Eye& E = f->eye[eye];
  Img8 I = wrap(img, w, hgt, stride);
  E.imgW = w; E.imgH = hgt;
  lineExtract(I, lineCfg(f->cfg), E.kls, E.ldesc, E.ld);
  return (int)E.kls.size();
}
int orc_get_keylines(void* h, int eye, pli_keyline* kl, uint8_t* desc) {
  Eye& E = ((Frame*)h)->eye[eye];
  if (kl) std::memcpy(kl, E.kls.data(), E.kls.size() * sizeof(pli_keyline));
  if (desc) std::memcpy(desc, E.ldesc.data(), E.ldesc.size());
  return (int)E.kls.size();
}
int orc_get_lsd_dims(void* h, int eye, int* w, int* hgt) {
  Eye& E = ((Frame*)h)->eye[eye];
  *w = E.ld.lsd.W; *hgt = E.ld.lsd.H;
  return 0;
}
int orc_get_lsd_scaled(void* h, int eye, uint8_t* dst) {
  Eye& E = ((Frame*)h)->eye[eye];
  std::memcpy(dst, E.ld.lsd.scaled.d.data(), E.ld.lsd.scaled.d.size());
  return 0;
}
int orc_get_lsd_scaled64(void* h, int eye, double* dst) {
  Eye& E = ((Frame*)h)->eye[eye];
  std::memcpy(dst, E.ld.lsd.scaled64.data(), E.ld.lsd.scaled64.size() * 8);
  return 0;
}
int orc_get_lsd_angle(void* h, int eye, float* dst) {
  Eye& E = ((Frame*)h)->eye[eye];
  std::memcpy(dst, E.ld.lsd.angleDeg.data(), E.ld.lsd.angleDeg.size() * 4);
  return 0;
}
int orc_get_lsd_order(void* h, int eye, int* dst, int cap) {
  Eye& E = ((Frame*)h)->eye[eye];
  int n = (int)E.ld.lsd.order.size();
  if (dst) std::memcpy(dst, E.ld.lsd.order.data(), (size_t)std::min(n, cap) * 4);
  return n;
}
int orc_get_lsd_segments(void* h, int eye, float* dst, int cap) {
  Eye& E = ((Frame*)h)->eye[eye];
  int n = (int)E.ld.lsd.segments.size() / 4;
  if (dst) std::memcpy(dst, E.ld.lsd.segments.data(), (size_t)std::min(n, cap) * 16);
  return n;
}
int orc_get_lbd_dxdy(void* h, int eye, int16_t* dx, int16_t* dy) {
  Eye& E = ((Frame*)h)->eye[eye];
  std::memcpy(dx, E.ld.dx.data(), E.ld.dx.size() * 2);
  std::memcpy(dy, E.ld.dy.data(), E.ld.dy.size() * 2);
  return 0;
}
int orc_get_lbd_float(void* h, int eye, float* dst) {
  Eye& E = ((Frame*)h)->eye[eye];
  std::memcpy(dst, E.ld.lbdFloat.data(), E.ld.lbdFloat.size() * 4);
  return (int)E.ld.lbdFloat.size() / 72;
}

// Frame::ComputeStereoMatches on the last orb extracts of both eyes.
int orc_stereo_points(void* h, float* uright, float* depth, int* bestIdx, int* sad) {
  Frame* f = (Frame*)h;
  Eye& L = f->eye[0];
  Eye& R = f->eye[1];
  float maxD = f->cfg.stereo_maxd_inf ? std::numeric_limits<float>::infinity()
                                      : f->cfg.bf / (f->cfg.bf / f->cfg.fx);   // mbf/mb with mb = mbf/fx (Frame.cc:197)
  computeStereoMatches(L.kps, L.desc.data(), R.kps, R.desc.data(), L.orb->mvImagePyramid, R.orb->mvImagePyramid,
                       L.orb->mvScaleFactor, L.orb->mvInvScaleFactor, f->cfg.bf, maxD, f->uright, f->depth,
                       &f->bestIdx, &f->sad);
  int n = (int)L.kps.size();
  if (uright) std::memcpy(uright, f->uright.data(), n * 4);
  if (depth) std::memcpy(depth, f->depth.data(), n * 4);
  if (bestIdx) std::memcpy(bestIdx, f->bestIdx.data(), n * 4);
  if (sad) std::memcpy(sad, f->sad.data(), n * 4);
  return n;
}

int orc_stereo_lines(void* h, float* disp, double* le, int* matches) {
  Frame* f = (Frame*)h;
  Eye& L = f->eye[0];
  Eye& R = f->eye[1];
  computeStereoMatchesLines(L.kls, L.ldesc.data(), R.kls, R.ldesc.data(), L.imgW, R.imgH, matchCfg(f->cfg), f->disp,
                            f->le, &f->lineMatches);
  int n = (int)L.kls.size();
  if (disp) std::memcpy(disp, f->disp.data(), (size_t)n * 8);
  if (le) std::memcpy(le, f->le.data(), (size_t)n * 24);
  if (matches) std::memcpy(matches, f->lineMatches.data(), (size_t)n * 4);
  return n;
}

// Whole Frame::Frame front-end for one stereo pair (cpu_baseline leg of bench.py).
int orc_frame_run(void* h, const uint8_t* left, const uint8_t* right, int w, int hgt, int64_t stride) {
  int nl = orc_orb_extract(h, 0, left, w, hgt, stride);
  int nr = orc_orb_extract(h, 1, right, w, hgt, stride);
  int ll = orc_line_extract(h, 0, left, w, hgt, stride);
  int lr = orc_line_extract(h, 1, right, w, hgt, stride);
  if (nl <= 0 || ll <= 0) return 0;      // Frame.cc:146-149 early return
  (void)nr; (void)lr;
  orc_stereo_lines(h, nullptr, nullptr, nullptr);
  orc_stereo_points(h, nullptr, nullptr, nullptr, nullptr);
  return nl;
}

// ---- stateless helpers ------------------------------------------------------
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b, int n, int* dist) {
  for (int i = 0; i < n; ++i) dist[i] = descriptorDistance(a + (size_t)i * 32, b + (size_t)i * 32);
  return 0;
}
int orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist) {
  std::vector<int> I, D;
  knn2(q, nq, t, nt, I, D);
  std::memcpy(idx, I.data(), I.size() * 4);
  std::memcpy(dist, D.data(), D.size() * 4);
  return 0;
}
int orc_match_lines(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int bestLR, int* m12) {
  std::vector<int> M;
  int r = matchLines(d1, n1, d2, n2, nnr, bestLR != 0, M);
  std::memcpy(m12, M.data(), M.size() * 4);
  return r;
}
int orc_search_by_projection(const pli_proj_query* q, const uint8_t* qdesc, int nq, const pli_keypoint* kp,
                             const uint8_t* desc, const float* uright, int ncur, float minx, float maxx, float miny,
                             float maxy, int checkOri, int* best_idx2, const uint8_t* occupied, int* raw_idx2) {
  std::vector<int> B, Raw;
  int r = searchByProjection(q, qdesc, nq, kp, desc, uright, ncur, minx, maxx, miny, maxy, checkOri != 0, B, occupied, &Raw);
  std::memcpy(best_idx2, B.data(), B.size() * 4);
  if (raw_idx2) std::memcpy(raw_idx2, Raw.data(), Raw.size() * 4);
  return r;
}
void orc_track_queries(const pli_keypoint* lastKp, const float* lastDepth, int n, const float* Tlw, const float* Tcw, float fx,
                       float fy, float cx, float cy, float bf, float th, int mono, const float* scaleFactors, pli_proj_query* q) {
  trackQueries(lastKp, lastDepth, n, Tlw, Tcw, fx, fy, cx, cy, bf, th, mono != 0, scaleFactors, q);
}
int orc_stereo_from_depth(const pli_keypoint* kp, int n, const float* depth, int64_t stride, int w, int h, float bf, float* uright,
                          float* depthOut) {
  std::vector<float> U, D;
  stereoFromDepth(kp, n, depth, stride, w, h, bf, U, D);
  std::memcpy(uright, U.data(), (size_t)n * 4);
  std::memcpy(depthOut, D.data(), (size_t)n * 4);
  return n;
}
// DBoW2 vocabulary: per-feature descent and the normalised BowVector (words ascending)
void* orc_vocab_create(int k, int L, int n, const int* parent, const uint8_t* isLeaf, const uint8_t* desc, const double* weight) {
  BowVocabulary* v = new BowVocabulary();
  v->load(k, L, n, parent, isLeaf, desc, weight);
  return v;
}
void orc_vocab_destroy(void* v) { delete (BowVocabulary*)v; }
int orc_bow_descend(void* vv, const uint8_t* feat, int n, int levelsup, int* word, double* weight, int* node) {
  const BowVocabulary* v = (const BowVocabulary*)vv;
  for (int i = 0; i < n; ++i) v->transformFeature(feat + (size_t)i * 32, word[i], weight[i], &node[i], levelsup);
  return n;
}
int orc_bow_vector(void* vv, const uint8_t* feat, int n, int levelsup, unsigned* words, double* values, int cap) {
  std::map<unsigned, double> bow; std::map<unsigned, std::vector<unsigned>> fv;
  ((const BowVocabulary*)vv)->transform(feat, n, bow, fv, levelsup);
  int k = 0;
  for (auto& kv : bow) { if (k < cap) { words[k] = kv.first; values[k] = kv.second; } ++k; }
  return k;
}
// cv::remap INTER_LINEAR of one 8U image (stereo_euroc.cc:166)
int orc_remap_linear(const uint8_t* img, int w, int h, int64_t stride, const float* mapx, const float* mapy, uint8_t* dst) {
  Img8 I = wrap(img, w, h, stride), D;
  remapLinear8u(I, mapx, mapy, D, w, h);
  std::memcpy(dst, D.d.data(), D.d.size());
  return 0;
}
int orc_remap_weights(int fx, int fy, int* w) { remapBilinearWeights(fx, fy, w); return 0; }

int orc_search_local_map(const pli_proj_query* q, const uint8_t* qdesc, int nq, const pli_keypoint* kp,
                         const uint8_t* desc, const float* uright, const uint8_t* occupied, int ncur, float minx, float maxx,
                         float miny, float maxy, float nnratio, int* best_idx2) {
  std::vector<int> B;
  int r = searchLocalMap(q, qdesc, nq, kp, desc, uright, occupied, ncur, minx, maxx, miny, maxy, nnratio, B);
  std::memcpy(best_idx2, B.data(), B.size() * 4);
  return r;
}
int orc_search_local_map_fisheye(const pli_proj_query* ql, const pli_proj_query* qr, const uint8_t* qdesc, int nq,
                                 const pli_keypoint* kpl, const uint8_t* dl, const uint8_t* occl, const int* l2r, int nl,
                                 const pli_keypoint* kpr, const uint8_t* dr, const uint8_t* occr, const int* r2l, int nr, float minx,
                                 float maxx, float miny, float maxy, float nnratio, int* mpl, int* mpr) {
  std::vector<int> ML, MR;
  int r = searchLocalMapFisheye(ql, qr, qdesc, nq, kpl, dl, occl, l2r, nl, kpr, dr, occr, r2l, nr, minx, maxx, miny, maxy, nnratio, ML, MR);
  if (nl) std::memcpy(mpl, ML.data(), ML.size() * 4);
  if (nr) std::memcpy(mpr, MR.data(), MR.size() * 4);
  return r;
}
int orc_match_nnr(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int* m12) {
  std::vector<int> M;
  int r = matchNNR(d1, n1, d2, n2, nnr, M);
  std::memcpy(m12, M.data(), M.size() * 4);
  return r;
}
// Stereo line matching on caller tables (edge-case tests without running LSD).
int orc_stereo_lines_tables(const pli_frontend_config* cfg, const pli_keyline* kl, const uint8_t* dl, int n1,
                            const pli_keyline* kr, const uint8_t* dr, int n2, int w, int hgt, float* disp, double* le,
                            int* matches) {
  std::vector<pli_keyline> KL(kl, kl + n1), KR(kr, kr + n2);
  std::vector<float> D;
  std::vector<double> LE;
  std::vector<int> M;
  computeStereoMatchesLines(KL, dl, KR, dr, w, hgt, matchCfg(*cfg), D, LE, &M);
  if (disp) std::memcpy(disp, D.data(), D.size() * 4);
  if (le) std::memcpy(le, LE.data(), LE.size() * 8);
  if (matches) std::memcpy(matches, M.data(), M.size() * 4);
  return n1;
}
// Bresenham cell walk (pinned against oracle/_ref).
int orc_line_coords(double x1, double y1, double x2, double y2, int* out, int cap) {
  std::vector<std::pair<int, int>> lc;
  getLineCoords(x1, y1, x2, y2, lc);
  int n = (int)lc.size();
  for (int i = 0; i < n && i < cap; ++i) { out[2 * i] = lc[i].first; out[2 * i + 1] = lc[i].second; }
  return n;
}
// Grid fill + window query (pinned against oracle/_ref): segments -> grid; query returns sorted candidate ids.
int orc_grid_query(const double* segs, int nseg, int rows, int cols, int qx, int qy, int wl, int wr, int hu, int hd,
                   int* out, int cap) {
  Grid g(rows, cols);
  std::vector<std::pair<int, int>> lc;
  for (int i = 0; i < nseg; ++i) {
    getLineCoords(segs[4 * i], segs[4 * i + 1], segs[4 * i + 2], segs[4 * i + 3], lc);
    for (auto& p : lc) g.push(p.first, p.second, i);
  }
  std::set<int> c;
  g.get(qx, qy, wl, wr, hu, hd, c);
  int n = 0;
  for (int v : c) { if (n < cap) out[n] = v; ++n; }
  return n;
}
// OpenCV-primitive restatements, exposed for known-answer tests.
int orc_stereo_fisheye(const pli_keypoint* kpL, const uint8_t* descL, int nleft, int monoLeft, const pli_keypoint* kpR,
                       const uint8_t* descR, int nright, int monoRight, const float* cam1, const float* cam2, const float* Rlr,
                       const float* tlr, const float* sigma2, int* l2r, int* r2l, float* depth, float* p3d) {
  Kb8Camera c1, c2;
  memcpy(&c1, cam1, sizeof(c1));
  memcpy(&c2, cam2, sizeof(c2));
  return computeStereoFishEyeMatches(kpL, descL, nleft, monoLeft, kpR, descR, nright, monoRight, c1, c2, Rlr, tlr, sigma2, l2r, r2l,
                                     depth, p3d);
}
int orc_lapping_order(const pli_keypoint* kp, int n, int lap0, int lap1, int* order) {
  std::vector<int> o;
  const int mono = lappingOrder(kp, n, lap0, lap1, o);
  for (int i = 0; i < n; ++i) order[i] = o[i];
  return mono;
}
int orc_kb8_unproject(const float* cam, float u, float v, float* r) { Kb8Camera c; memcpy(&c, cam, sizeof(c)); kb8Unproject(c, u, v, r); return 0; }
int orc_kb8_project(const float* cam, const float* p, float* uv) { Kb8Camera c; memcpy(&c, cam, sizeof(c)); kb8Project(c, p, uv[0], uv[1]); return 0; }
int orc_level_sigma2(void* h, float* out) {
  Frame* f = (Frame*)h;
  const auto& v = f->eye[0].orb->mvLevelSigma2;
  for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
  return (int)v.size();
}
float orc_fast_atan2(float y, float x) { return fastAtan2(y, x); }
int orc_cv_round(double v) { return cvRound(v); }
int orc_gauss_kernel(int n, double sigma, int* out) {
  std::vector<int> k = gaussKernelFixed8(n, sigma);
  for (int i = 0; i < n; ++i) out[i] = k[i];
  return n;
}
int orc_gaussian_blur(const uint8_t* img, int w, int h, int n, double sigma, uint8_t* dst) {
  Img8 I = wrap(img, w, h, w), O;
  gaussianBlur8u(I, O, n, sigma);
  std::memcpy(dst, O.d.data(), O.d.size());
  return 0;
}
int orc_resize(const uint8_t* img, int w, int h, int dw, int dh, double sx, double sy, uint8_t* dst) {
  Img8 I = wrap(img, w, h, w), O;
  resizeLinear8u(I, O, dw, dh, sx, sy);
  std::memcpy(dst, O.d.data(), O.d.size());
  return 0;
}
int orc_fast_arc(const uint8_t* img, int w, int h, int x, int y) {
  (void)h;
  return fastArcValue(img + (size_t)y * w + x, w);
}
int orc_sobel(const uint8_t* img, int w, int h, int16_t* dx, int16_t* dy) {
  Img8 I = wrap(img, w, h, w);
  std::vector<int16_t> X, Y;
  sobel3x3_16s(I, X, Y);
  std::memcpy(dx, X.data(), X.size() * 2);
  std::memcpy(dy, Y.data(), Y.size() * 2);
  return 0;
}
int orc_lbd_weights(float* L21, float* G63) {
  LbdWeights W;
  std::memcpy(L21, W.gaussCoefL, sizeof(W.gaussCoefL));
  std::memcpy(G63, W.gaussCoefG, sizeof(W.gaussCoefG));
  return 0;
}
int orc_orb_descriptor(const uint8_t* img, int w, int h, int x, int y, float angle, uint8_t* desc, int trigF32) {
  Img8 I = wrap(img, w, h, w);
  OrbExtractor::computeOrbDescriptor(angle, x, y, I, desc, trigF32 != 0);
  return 0;
}

// cv::FAST(img, kps, th, true) on a whole image (the extractor calls it per cell, ORBextractor.cc:803-815): x, y, response triples
int orc_fast_image(const uint8_t* img, int w, int h, int th, float* out, int cap) {
  Img8 I = wrap(img, w, h, w);
  std::vector<OrbCand> c;
  OrbExtractor::fastCell(I, 0, 0, w, h, th, th, c);
  int n = 0;
  for (const OrbCand& k : c) {
    if (n < cap) { out[3 * n] = (float)k.x; out[3 * n + 1] = (float)k.y; out[3 * n + 2] = (float)k.score; }
    ++n;
  }
  return n;
}
float orc_glibc_cosf(float x) { return glibcCosf(x); }
float orc_glibc_sinf(float x) { return glibcSinf(x); }
// dense check of the sincosf restatement against the libm of this machine: returns the number of floats that differ
long orc_sincosf_selfcheck(uint32_t first, uint32_t last, uint32_t step) {
  long bad = 0;
  for (uint64_t b = first; b < last; b += step) {
    float x;
    const uint32_t u = (uint32_t)b;
    std::memcpy(&x, &u, 4);
    if (cosf(x) != glibcCosf(x) || sinf(x) != glibcSinf(x)) ++bad;
  }
  return bad;
}

}  // extern "C"
