// The front of OpenCV 3.x's LineSegmentDetector on the CV_64FC1 copy of the image (PLI_PARITY_LSD_F64, the default: the
// reference pins OpenCV 3.3.1, whose lsd.cpp starts with `img.convertTo(image, CV_64FC1)`):
//   k_lsd_blur64    GaussianBlur(image, gaussian_img, ksize, sigma) on doubles — sepFilter2D with the filter engine's
//                   generic double filters: RowFilter (left to right), SymmColumnFilter (centre, then the +-k pairs)
//   k_lsd_resize64  resize(gaussian_img, scaled_image, Size(), SCALE, SCALE): INTER_LINEAR on doubles, float coefficients
//   k_lsd_grad64    ll_angle: 2x2 gradient in double, modgrad (double plane), level-line angle, max gradient
// (OpenCV-3.3.1-compatible by intent: see DESIGN.md "Oracle" for what is known about each primitive.)
#include "kernels.hpp"
#include "device_prims.hpp"

namespace pli {

constexpr float F64_NOTDEF = -1024.f;
constexpr double F64_DEG2RAD = 3.14159265358979323846 / 180;

// one 64 x 16 output tile per workgroup; u8 tile with halo and the row-filtered doubles staged in LDS
__global__ __launch_bounds__(256) void k_lsd_blur64(const uint8_t* __restrict__ pyr, int64_t pyrBlock, int W, int H, int pitch,
                                                    const double* __restrict__ kern, int radius, double* __restrict__ out,
                                                    int64_t outImgStride, int img0) {
  __shared__ uint8_t tile[22][72];
  __shared__ double rows[22][64];
  const int img = blockIdx.z + img0, tid = threadIdx.x;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
  const uint8_t* src = pyr + (int64_t)img * pyrBlock;
  const int r = radius, n = 2 * r + 1;
  const int tw = 64 + 2 * r, th = 16 + 2 * r;
  for (int i = tid; i < tw * th; i += 256) {
    const int ty = i / tw, tx = i - ty * tw;
    const int sx = reflect101(min(x0 + tx - r, W + r), W), sy = reflect101(min(y0 + ty - r, H + r), H);
    tile[ty][tx] = src[(int64_t)sy * pitch + sx];
  }
  __syncthreads();
  double k[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) k[i] = i < n ? kern[i] : 0.0;
  for (int i = tid; i < 64 * th; i += 256) {
    const int ty = i >> 6, tx = i & 63;
    double s = k[0] * (double)tile[ty][tx];
    for (int j = 1; j < n; ++j) s += k[j] * (double)tile[ty][tx + j];
    rows[ty][tx] = s;
  }
  __syncthreads();
  for (int i = tid; i < 64 * 16; i += 256) {
    const int ty = i >> 6, tx = i & 63;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) continue;
    double s = k[r] * rows[ty + r][tx];
    for (int j = 1; j <= r; ++j) s += k[r + j] * (rows[ty + r + j][tx] + rows[ty + r - j][tx]);
    out[(int64_t)img * outImgStride + (int64_t)y * W + x] = s;
  }
}

// tab: xofs[dw] | alpha[2*dw] (float bits) | yofs[dh] | beta[2*dh] (float bits)
__global__ __launch_bounds__(256) void k_lsd_resize64(const double* __restrict__ src, int sw, int sh, double* __restrict__ dst,
                                                      int dw, int dh, int64_t dstImgStride, const int* __restrict__ tab, int img0) {
  const int img = blockIdx.z + img0, dy = blockIdx.y, dx = blockIdx.x * 256 + threadIdx.x;
  if (dx >= dw) return;
  const int sx = tab[dx], sx1 = min(sx + 1, sw - 1);
  const double a0 = (double)__int_as_float(tab[dw + 2 * dx]), a1 = (double)__int_as_float(tab[dw + 2 * dx + 1]);
  const int sy = tab[3 * dw + dy];
  const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
  const double b0 = (double)__int_as_float(tab[3 * dw + dh + 2 * dy]), b1 = (double)__int_as_float(tab[3 * dw + dh + 2 * dy + 1]);
  const double* S0 = src + (int64_t)img * sw * sh + (int64_t)sy0 * sw;
  const double* S1 = src + (int64_t)img * sw * sh + (int64_t)sy1 * sw;
  const double t0 = S0[sx] * a0 + S0[sx1] * a1;
  const double t1 = S1[sx] * a0 + S1[sx1] * a1;
  dst[(int64_t)img * dstImgStride + (int64_t)dy * dw + dx] = t0 * b0 + t1 * b1;
}

// rec = { ang, c, s, 0 } (rec.w only carries the speculative grower's tags here), mg = modgrad (double)
// (W: the true width, the scaled plane's row length; WP >= W: the row pitch of the planes written — the pad columns are undefined pixels)
__global__ __launch_bounds__(256) void k_lsd_grad64(const double* __restrict__ scaled, int W, int H, int WP, int64_t scaledImgStride, double rho,
                                                    float4* __restrict__ rec, double* __restrict__ mg, int2* __restrict__ own,
                                                    unsigned long long* __restrict__ maxMg, float* __restrict__ angDbg, int img0,
                                                    int flags /* bit 0: PLI_PARITY_TRIG_F32_LSD; bit 1: rec.w = tx_unclaimed_norm_word */,
                                                    int2* __restrict__ hot /* or null: round 1's 8-byte hot records {angle, owner word} (lsd_tile.hip) */,
                                                    float2* __restrict__ cold /* or null: {cos, sin} beside the hot records when rec is not written */) {
  const int trigF32 = flags & 1;
  const bool packW = (flags & 2) != 0;
  __shared__ unsigned long long wmax[4];
  const int img = blockIdx.z + img0;
  const int x = blockIdx.x * 256 + threadIdx.x;
  unsigned long long m = 0ull;                            // bits of the largest norm (non-negative doubles order like their bits)
  if (x < WP) {
    const int yEnd = min((int)(blockIdx.y + 1) * 16, H);
    const double* S = scaled + (int64_t)img * scaledImgStride;
    for (int y = blockIdx.y * 16; y < yEnd; ++y) {
      double norm = 0.0;
      float a = F64_NOTDEF, cx = 0.f, sy = 0.f;
      if (x < W - 1 && y < H - 1) {
        const double* r0 = S + (int64_t)y * W;
        const double* r1 = r0 + W;
        const double DA = r1[x + 1] - r0[x];
        const double BC = r0[x + 1] - r1[x];
        const double gx = DA + BC, gy = DA - BC;
        norm = sqrt((gx * gx + gy * gy) / 4);
        if (!(norm <= rho)) {
          m = max(m, (unsigned long long)__double_as_longlong(norm));
          a = fast_atan2_deg((float)gx, (float)(-gy));
          sincos_of_float((float)((double)a * F64_DEG2RAD), trigF32 != 0, &sy, &cx);
        }
      }
      const int64_t o = (int64_t)img * WP * H + (int64_t)y * WP + x;
      if (rec) rec[o] = make_float4(a, cx, sy, (packW && !hot) ? tx_unclaimed_norm_word(norm) : 0.f);
      if (hot) hot[o] = make_int2(__float_as_int(a), packW ? __float_as_int(tx_unclaimed_norm_word(norm)) : 0);
      if (cold) cold[o] = make_float2(a == F64_NOTDEF ? F64_NOTDEF : cx, sy);      // (cos = NOTDEF: no level-line angle)
      mg[o] = norm;
      if (own) own[o] = make_int2(0x7FFFFFFF, 0x7FFFFFFF);
      if (angDbg && x < W) angDbg[(int64_t)img * W * H + (int64_t)y * W + x] = a;      // (the debug plane: rows of the true width)
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    if (m) atomicMax(&maxMg[img], m);
  }
}

// ---------------------------------------------------------------------------
// k_lsd_front64: the three kernels above in one pass per 64 x 16 tile of the SCALED image (lsd_scale != 1): the u8 source
// window of the tile (plus the blur radius) is staged in LDS, blurred there (rows, then columns: the same sums in the same
// order as k_lsd_blur64), resized from LDS (the expressions of k_lsd_resize64) into a 65 x 17 tile of the scaled image, and
// the gradient / level-line angle of the 64 x 16 pixels is written (k_lsd_grad64).  Neither the blurred nor the scaled
// double plane goes to HBM: per scaled pixel the pass writes 24 B (record + norm) and reads ~0.8 B, where the three
// kernels moved 8·(2·P0/P' + 2) + 24 = 51 B.  The host uses it when the source window of every tile fits FS_W x FS_H
// (any scale >= 1; otherwise the separate kernels run) and debugging is off (PLI_DBG_LSD_SCALED wants the plane).
// ---------------------------------------------------------------------------
constexpr int FT_W = 64, FT_H = 16;        // scaled pixels per tile
constexpr int FS_W = 64, FS_H = 20;        // blurred source window (columns, rows) a tile may need
constexpr int FR = 3;                      // largest blur radius

// (R > 0: the blur radius at compile time — the taps unrolled with the kernel in registers, the same products summed in the same order;
// R = 0: any radius up to FR, the taps looped.  OpenCV's default sigma_scale 0.6 gives radius 3 for every scale >= 1.)
template <int R>
__device__ __forceinline__ void lsd_front64_tile(const uint8_t* __restrict__ pyr, int64_t pyrBlock, int sw, int sh, int pitch,
                                                 const double* __restrict__ kern, int radius, const int* __restrict__ tab,
                                                 int dw, int dh, int dp, double rho, float4* __restrict__ rec, double* __restrict__ mg,
                                                 int2* __restrict__ own, unsigned long long* __restrict__ maxMg, int img0, int flags,
                                                 int2* __restrict__ hot, float2* __restrict__ cold, uint8_t (*tile)[FS_W + 2 * FR + 2], double (*rows)[FS_W], double (*blur)[FS_W],
                                                 unsigned long long* wmax) {
  const int trigF32 = flags & 1;
  const bool packW = (flags & 2) != 0;
  static_assert((FT_H + 1) * (FT_W + 1) <= (FS_H + 2 * FR) * FS_W, "the scaled tile reuses the row buffer");
  double (*scl)[FT_W + 1] = reinterpret_cast<double (*)[FT_W + 1]>(&rows[0][0]);
  const int img = blockIdx.z + img0, tid = threadIdx.x;
  const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
  const uint8_t* src = pyr + (int64_t)img * pyrBlock;
  const int r = R > 0 ? R : radius, n = 2 * r + 1;
  // scaled pixels of this tile (with the +1 halo of the 2x2 gradient), and the source window they read
  const int x1 = min(x0 + FT_W, dw - 1), y1 = min(y0 + FT_H, dh - 1);          // last scaled column / row needed
  const int sx0 = tab[x0], sx1 = min(tab[x1] + 1, sw - 1);
  const int sy0 = min(max(tab[3 * dw + y0], 0), sh - 1), sy1 = min(max(tab[3 * dw + y1] + 1, 0), sh - 1);
  const int cw = sx1 - sx0 + 1, ch = sy1 - sy0 + 1;                             // <= FS_W, FS_H (checked by the host)
  const int tw = cw + 2 * r, th = ch + 2 * r;
  // (thread = (column lx, row group ly): no division by the window width anywhere)
  const int lx = tid & 63, ly = tid >> 6;
  if (R > 0 && sw > 2 * R && sh > 2 * R) {
    // the window leaves the image by at most R < sw, sh pixels: ONE reflection, no loop — and a fixed trip count, so that the (up to
    // 14) byte loads of a thread are in flight together
    auto refl = [](int p, int len) { return p < 0 ? -p : (p >= len ? 2 * len - 2 - p : p); };
    static_assert((FS_H + 2 * FR + 3) / 4 == 7 && FS_W + 2 * FR <= 128, "rows / columns a thread stages");
    const int c0 = refl(sx0 + lx - R, sw), c1 = refl(sx0 + lx + 64 - R, sw);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int ty = ly + 4 * i;
      if (ty < th) {
        const uint8_t* row = src + (int64_t)refl(sy0 + ty - R, sh) * pitch;
        if (lx < tw) tile[ty][lx] = row[c0];
        if (lx + 64 < tw) tile[ty][lx + 64] = row[c1];
      }
    }
  } else {
    for (int ty = ly; ty < th; ty += 4) {
      const int sy = reflect101(sy0 + ty - r, sh);
      for (int tx = lx; tx < tw; tx += 64) tile[ty][tx] = src[(int64_t)sy * pitch + reflect101(sx0 + tx - r, sw)];
    }
  }
  __syncthreads();
  double k[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) k[i] = i < n ? kern[i] : 0.0;
  if (lx < cw)
    for (int ty = ly; ty < th; ty += 4) {
      double s = k[0] * (double)tile[ty][lx];
      if (R > 0) {
#pragma unroll
        for (int j = 1; j < 2 * R + 1; ++j) s += k[j] * (double)tile[ty][lx + j];
      } else {
        for (int j = 1; j < n; ++j) s += k[j] * (double)tile[ty][lx + j];
      }
      rows[ty][lx] = s;
    }
  __syncthreads();
  if (lx < cw)
    for (int ty = ly; ty < ch; ty += 4) {
      double s;
      if (R > 0) {
        s = k[R] * rows[ty + R][lx];
#pragma unroll
        for (int j = 1; j <= R; ++j) s += k[R + j] * (rows[ty + R + j][lx] + rows[ty + R - j][lx]);
      } else {
        s = k[r] * rows[ty + r][lx];
        for (int j = 1; j <= r; ++j) s += k[r + j] * (rows[ty + r + j][lx] + rows[ty + r - j][lx]);
      }
      blur[ty][lx] = s;
    }
  __syncthreads();
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
  // thread = (column lx, rows ly, ly + 4, ..); the 65th column (the +1 halo of the 2x2 gradient) is spread over the first ny threads,
  // one row each, instead of a second pass of the whole workgroup for one lane's worth of pixels
  auto resizeCol = [&](int tx, int tyFirst, int tyStep) {
    const int dx = x0 + tx;
    const int sx = tab[dx] - sx0, sxn = min(sx + sx0 + 1, sw - 1) - sx0;
    const double a0 = (double)__int_as_float(tab[dw + 2 * dx]), a1 = (double)__int_as_float(tab[dw + 2 * dx + 1]);
    for (int ty = tyFirst; ty < ny; ty += tyStep) {
      const int dy = y0 + ty;
      const int sy = tab[3 * dw + dy];
      const int sya = min(max(sy, 0), sh - 1) - sy0, syb = min(max(sy + 1, 0), sh - 1) - sy0;
      const double b0 = (double)__int_as_float(tab[3 * dw + dh + 2 * dy]), b1 = (double)__int_as_float(tab[3 * dw + dh + 2 * dy + 1]);
      const double t0 = blur[sya][sx] * a0 + blur[sya][sxn] * a1;
      const double t1 = blur[syb][sx] * a0 + blur[syb][sxn] * a1;
      scl[ty][tx] = t0 * b0 + t1 * b1;
    }
  };
  if (lx < nx) resizeCol(lx, ly, 4);
  if (nx > FT_W && tid < ny) resizeCol(FT_W, tid, FT_H + 1);
  __syncthreads();
  unsigned long long m = 0ull;
  for (int i = tid; i < FT_W * FT_H; i += 256) {
    const int ty = i >> 6, tx = i & 63;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= dp || y >= dh) continue;                      // (columns dw .. dp - 1: the pad of the planes' row pitch, undefined pixels)
    double norm = 0.0;
    float a = F64_NOTDEF, cx = 0.f, sy = 0.f;
    if (x < dw - 1 && y < dh - 1) {
      const double DA = scl[ty + 1][tx + 1] - scl[ty][tx];
      const double BC = scl[ty][tx + 1] - scl[ty + 1][tx];
      const double gx = DA + BC, gy = DA - BC;
      norm = sqrt((gx * gx + gy * gy) / 4);
      if (!(norm <= rho)) {
        m = max(m, (unsigned long long)__double_as_longlong(norm));
        a = fast_atan2_deg((float)gx, (float)(-gy));
        sincos_of_float((float)((double)a * F64_DEG2RAD), trigF32 != 0, &sy, &cx);
      }
    }
    const int64_t o = (int64_t)img * dp * dh + (int64_t)y * dp + x;
    if (rec) rec[o] = make_float4(a, cx, sy, (packW && !hot) ? tx_unclaimed_norm_word(norm) : 0.f);      // (null: every round runs on the hot records)
    if (hot) hot[o] = make_int2(__float_as_int(a), packW ? __float_as_int(tx_unclaimed_norm_word(norm)) : 0);
    if (cold) cold[o] = make_float2(a == F64_NOTDEF ? F64_NOTDEF : cx, sy);      // (cos = NOTDEF: no level-line angle)
    mg[o] = norm;
    if (own) own[o] = make_int2(0x7FFFFFFF, 0x7FFFFFFF);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    if (m) atomicMax(&maxMg[img], m);
  }
}

__global__ __launch_bounds__(256) void k_lsd_front64(const uint8_t* __restrict__ pyr, int64_t pyrBlock, int sw, int sh, int pitch,
                                                     const double* __restrict__ kern, int radius, const int* __restrict__ tab,
                                                     int dw, int dh, int dp /* row pitch of the planes written, >= dw */, double rho,
                                                     float4* __restrict__ rec, double* __restrict__ mg,
                                                     int2* __restrict__ own, unsigned long long* __restrict__ maxMg, int img0,
                                                     int flags /* as k_lsd_grad64 */, int2* __restrict__ hot /* as k_lsd_grad64 */,
                                                     float2* __restrict__ cold /* as k_lsd_grad64 */) {
  __shared__ uint8_t tile[FS_H + 2 * FR][FS_W + 2 * FR + 2];
  __shared__ double rows[FS_H + 2 * FR][FS_W];               // row-filtered window; afterwards the scaled tile (scl)
  __shared__ double blur[FS_H][FS_W];
  __shared__ unsigned long long wmax[4];
  if (radius == FR) lsd_front64_tile<FR>(pyr, pyrBlock, sw, sh, pitch, kern, radius, tab, dw, dh, dp, rho, rec, mg, own, maxMg, img0, flags, hot, cold, tile, rows, blur, wmax);
  else lsd_front64_tile<0>(pyr, pyrBlock, sw, sh, pitch, kern, radius, tab, dw, dh, dp, rho, rec, mg, own, maxMg, img0, flags, hot, cold, tile, rows, blur, wmax);
}

}  // namespace pli
