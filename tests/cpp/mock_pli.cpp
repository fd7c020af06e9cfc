// TEST-ONLY stand-in for libpli_frontend.so on machines without a GPU: the C entry points the adapters call, with deterministic
// made-up results (functions of the image bytes), so that tests/cpp/dropin_harness.cpp — the adapters, their registry, the
// Frame-level matchers — can be built with -fsanitize=thread / address and RUN in the CPU suite (tests/test_cpp_host.py).
// It keeps the library's threading contract: every call on a context holds that context's lock (include/pli_frontend.h).
// Nothing here is product code and nothing is compared with the oracle.
#include "../../include/pli_frontend.h"
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

struct pli_ctx {
  std::recursive_mutex mu;
  pli_frontend_config cfg;
  pli_table_layout lay;
  int nkp[2] = {-1, -1}, nkl[2] = {-1, -1};
  std::vector<uint8_t> level0[2];
  uint32_t seed[2] = {0, 0};
};
static thread_local std::string g_err;
static uint32_t mix(uint32_t h, uint32_t v) { h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2); return h; }
static uint32_t hashImage(const uint8_t* img, int w, int h, int64_t stride) {
  uint32_t s = 2166136261u;
  for (int y = 0; y < h; y += 7) for (int x = 0; x < w; x += 5) s = mix(s, img[y * stride + x]);
  return s;
}
struct Lock { pli_ctx* c; explicit Lock(pli_ctx* c_) : c(c_) { c->mu.lock(); } ~Lock() { c->mu.unlock(); } };

extern "C" {
const char* pli_last_error(void) { return g_err.c_str(); }
void pli_config_default(pli_frontend_config* c, int32_t w, int32_t h) {
  std::memset(c, 0, sizeof(*c));
  c->width = w; c->height = h; c->max_frames = 1; c->orb_nfeatures = 1200; c->orb_scale_factor = 1.2f; c->orb_nlevels = 8;
  c->orb_ini_th_fast = 20; c->orb_min_th_fast = 7; c->lsd_nfeatures = 500; c->lsd_n_bins = 1024; c->max_lines = 4096;
  c->min_line_length = 0.025; c->lsd_scale = 1.2; c->lsd_sigma_scale = 0.6; c->lsd_quant = 2.0; c->lsd_ang_th = 22.5; c->lsd_log_eps = 1.0;
  c->lsd_density_th = 0.6; c->bf = 47.9f; c->fx = 435.2f;
}
int32_t pli_kp_capacity(const pli_frontend_config* c) { return c->orb_nfeatures + 3 * c->orb_nlevels; }
int32_t pli_kl_capacity(const pli_frontend_config* c) { return c->lsd_nfeatures ? c->lsd_nfeatures : c->max_lines; }
pli_status pli_ctx_create(const pli_frontend_config* cfg, int32_t, pli_ctx** out) {
  if (!cfg || !out) return PLI_ERR_INVALID;
  if (cfg->lsd_refine != 0) { g_err = "only lsd_refine = 0"; return PLI_ERR_INVALID; }
  pli_ctx* c = new pli_ctx();
  c->cfg = *cfg;
  std::memset(&c->lay, 0, sizeof(c->lay));
  pli_table_layout& L = c->lay;
  L.kp_cap = pli_kp_capacity(cfg); L.kl_cap = pli_kl_capacity(cfg);
  int64_t o = 0;
  auto take = [&](int64_t bytes) { const int64_t r = o; o = (o + bytes + 15) / 16 * 16; return r; };
  L.off_counts = take(32);
  for (int e = 0; e < 2; ++e) L.off_kp[e] = take((int64_t)L.kp_cap * sizeof(pli_keypoint));
  for (int e = 0; e < 2; ++e) L.off_desc[e] = take((int64_t)L.kp_cap * 32);
  L.off_uright = take((int64_t)L.kp_cap * 4); L.off_depth = take((int64_t)L.kp_cap * 4);
  for (int e = 0; e < 2; ++e) L.off_kl[e] = take((int64_t)L.kl_cap * sizeof(pli_keyline));
  for (int e = 0; e < 2; ++e) L.off_ldesc[e] = take((int64_t)L.kl_cap * 32);
  L.off_disp = take((int64_t)L.kl_cap * 8); L.off_le = take((int64_t)L.kl_cap * 24);
  L.record_bytes = o;
  *out = c;
  return PLI_OK;
}
void pli_ctx_destroy(pli_ctx* c) { if (c) { { Lock l(c); } delete c; } }
pli_status pli_ctx_layout(const pli_ctx* c, pli_table_layout* out) { *out = c->lay; return PLI_OK; }
pli_status pli_orb_extract(pli_ctx* c, int32_t eye, const uint8_t* img, int32_t w, int32_t h, int64_t stride, pli_keypoint* kp, int32_t cap,
                           uint8_t* desc, int32_t* n) {
  Lock l(c);
  if (!img || w <= 0 || h <= 0) return PLI_ERR_EMPTY_IMAGE;
  const uint32_t s = hashImage(img, w, h, stride);
  const int N = 200 + (int)(s % 100);
  if (N > cap) return PLI_ERR_CAPACITY;
  for (int i = 0; i < N; ++i) {
    uint32_t r = mix(s, (uint32_t)i);
    kp[i].x = 16.f + (float)(r % (uint32_t)(w - 32)); r = mix(r, 1);
    kp[i].y = 16.f + (float)(r % (uint32_t)(h - 32)); r = mix(r, 2);
    kp[i].size = 31.f; kp[i].angle = (float)(r % 360); kp[i].response = (float)(r % 255); kp[i].octave = (int)(r % 8u);
    for (int b = 0; b < 32; ++b) { r = mix(r, (uint32_t)b); desc[(size_t)i * 32 + b] = (uint8_t)r; }
  }
  c->nkp[eye] = N; c->seed[eye] = s;
  c->level0[eye].assign((size_t)w * h, 0);
  for (int y = 0; y < h; ++y) std::memcpy(&c->level0[eye][(size_t)y * w], img + y * stride, w);
  *n = N;
  return PLI_OK;
}
pli_status pli_orb_pyramid_level(pli_ctx* c, int32_t eye, int32_t level, uint8_t* dst, int64_t bytes, int32_t* w, int32_t* h) {
  Lock l(c);
  if (c->nkp[eye] < 0) return PLI_ERR_STATE;
  int lw = c->cfg.width, lh = c->cfg.height;
  for (int i = 0; i < level; ++i) { lw = (int)(lw / 1.2f + 0.5f); lh = (int)(lh / 1.2f + 0.5f); }
  if (w) *w = lw;
  if (h) *h = lh;
  if (!dst) return PLI_OK;
  if (bytes < (int64_t)lw * lh) return PLI_ERR_CAPACITY;
  for (int y = 0; y < lh; ++y) for (int x = 0; x < lw; ++x) dst[(size_t)y * lw + x] = c->level0[eye][(size_t)(y * c->cfg.height / lh) * c->cfg.width + x * c->cfg.width / lw];
  return PLI_OK;
}
pli_status pli_line_extract(pli_ctx* c, int32_t eye, const uint8_t* img, int32_t w, int32_t h, int64_t stride, pli_keyline* kl, int32_t cap,
                            uint8_t* desc, int32_t* n) {
  Lock l(c);
  const uint32_t s = hashImage(img, w, h, stride) ^ 0x55aa55aau;
  const int N = 20 + (int)(s % 10);
  if (N > cap) return PLI_ERR_CAPACITY;
  for (int i = 0; i < N; ++i) {
    uint32_t r = mix(s, (uint32_t)i);
    std::memset(&kl[i], 0, sizeof(kl[i]));
    kl[i].startPointX = (float)(r % (uint32_t)w); r = mix(r, 1); kl[i].startPointY = (float)(r % (uint32_t)h); r = mix(r, 2);
    kl[i].endPointX = (float)(r % (uint32_t)w); r = mix(r, 3); kl[i].endPointY = (float)(r % (uint32_t)h);
    kl[i].class_id = i; kl[i].lineLength = 30.f; kl[i].numOfPixels = 30; kl[i].response = 0.1f;
    for (int b = 0; b < 32; ++b) { r = mix(r, (uint32_t)b); desc[(size_t)i * 32 + b] = (uint8_t)r; }
  }
  c->nkl[eye] = N;
  *n = N;
  return PLI_OK;
}
pli_status pli_last_counts(pli_ctx* c, int32_t out[4]) { Lock l(c); out[0] = c->nkp[0]; out[1] = c->nkp[1]; out[2] = c->nkl[0]; out[3] = c->nkl[1]; return PLI_OK; }
pli_status pli_set_stereo_camera(pli_ctx* c, float bf, float fx) { Lock l(c); c->cfg.bf = bf; c->cfg.fx = fx; return PLI_OK; }
pli_status pli_stereo_match_points(pli_ctx* c, float* ur, float* depth, int32_t cap) {
  Lock l(c);
  if (c->nkp[0] < 0 || c->nkp[1] < 0) return PLI_ERR_STATE;
  if (c->nkp[0] > cap) return PLI_ERR_CAPACITY;
  for (int i = 0; i < c->nkp[0]; ++i) {
    const uint32_t r = mix(c->seed[0] ^ c->seed[1], (uint32_t)i);
    const bool m = (r & 3u) != 0u;
    depth[i] = m ? 1.f + (float)(r % 50u) * 0.2f : -1.f;
    ur[i] = m ? 100.f - c->cfg.bf / depth[i] : -1.f;
  }
  return PLI_OK;
}
pli_status pli_stereo_match_lines(pli_ctx* c, float* disp, double* le, int32_t cap) {
  Lock l(c);
  if (c->nkl[0] < 0 || c->nkl[1] < 0) return PLI_ERR_STATE;
  if (c->nkl[0] > cap) return PLI_ERR_CAPACITY;
  for (int i = 0; i < c->nkl[0]; ++i) {
    const bool m = (i & 1) == 0;
    disp[2 * i] = m ? 3.f + i : -1.f; disp[2 * i + 1] = m ? 4.f + i : -1.f;
    le[3 * i] = m ? 0.6 : 0.0; le[3 * i + 1] = m ? 0.8 : 0.0; le[3 * i + 2] = m ? -(double)i : 0.0;
  }
  return PLI_OK;
}
pli_status pli_frame_extract(pli_ctx* c, const uint8_t* left, const uint8_t* right, int32_t w, int32_t h, int64_t sl, int64_t sr, void* record) {
  Lock l(c);
  uint8_t* rec = (uint8_t*)record;
  std::memset(rec, 0, (size_t)c->lay.record_bytes);
  int32_t* counts = (int32_t*)(rec + c->lay.off_counts);
  const uint8_t* im[2] = {left, right};
  const int64_t st[2] = {sl, sr};
  for (int e = 0; e < 2; ++e) {
    pli_status s = pli_orb_extract(c, e, im[e], w, h, st[e], (pli_keypoint*)(rec + c->lay.off_kp[e]), c->lay.kp_cap, rec + c->lay.off_desc[e], &counts[e]);
    if (s != PLI_OK) return s;
    s = pli_line_extract(c, e, im[e], w, h, st[e], (pli_keyline*)(rec + c->lay.off_kl[e]), c->lay.kl_cap, rec + c->lay.off_ldesc[e], &counts[2 + e]);
    if (s != PLI_OK) return s;
  }
  pli_stereo_match_points(c, (float*)(rec + c->lay.off_uright), (float*)(rec + c->lay.off_depth), c->lay.kp_cap);
  pli_stereo_match_lines(c, (float*)(rec + c->lay.off_disp), (double*)(rec + c->lay.off_le), c->lay.kl_cap);
  return PLI_OK;
}
pli_status pli_match_lines(pli_ctx* c, const uint8_t*, int32_t n1, const uint8_t*, int32_t n2, float, int32_t* m12, int32_t* n) {
  Lock l(c);
  int k = 0;
  for (int i = 0; i < n1; ++i) { m12[i] = (i < n2 && (i % 3) == 0) ? i : -1; k += m12[i] >= 0; }
  *n = k;
  return PLI_OK;
}
pli_status pli_search_by_projection(pli_ctx* c, const pli_proj_query* q, const uint8_t*, int32_t nq, const pli_keypoint*, const uint8_t*,
                                    const float*, const uint8_t* occ, int32_t ncur, float, float, float, float, int32_t, int32_t* best,
                                    int32_t* raw, int32_t* n) {
  Lock l(c);
  int k = 0;
  for (int i = 0; i < nq; ++i) {
    best[i] = (q[i].valid && i < ncur && (i % 2) == 0 && !(occ && occ[i])) ? i : -1;
    if (raw) raw[i] = best[i];
    k += best[i] >= 0;
  }
  *n = k;
  return PLI_OK;
}
}  // extern "C"
