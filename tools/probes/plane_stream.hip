// Probe: how fast can the per-pixel bookkeeping passes of the tile relaxation (k_tx_round2 / k_tx_prep / k_tx_diffmark) read the
// owner plane (int2 per pixel) and the id plane (int per pixel) of 512 images of 602 x 384 pixels, by thread -> pixel mapping?
//   hipcc --offload-arch=gfx950 -O3 plane_stream.hip -o plane_stream && ./plane_stream
//   mode 0: 32 x 32 pixels per block of 256 threads, thread = column x of 4 rows 8 apart (the kernels' mapping in rounds 2..3)
//   mode 1: flat, one pixel per thread, 4 pixels 256 apart per thread
//   mode 2: flat, 4 consecutive pixels per thread (two 16-byte loads + one 16-byte load)
//   mode 3: mode 0 with 64-wide rows (64 x 16 per block)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const int2* __restrict__ own, const int* __restrict__ id, int W, int H, int* __restrict__ out) {
  const int64_t npix = (int64_t)W * H, base = (int64_t)blockIdx.z * npix;
  const int tid = threadIdx.x;
  int acc = 0;
  if (MODE == 0 || MODE == 3) {
    const int bw = MODE == 0 ? 32 : 64, rows = 256 / bw;
    const int x = blockIdx.x * bw + (tid % bw);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = blockIdx.y * (4 * rows) + i * rows + tid / bw;
      if (x < W && y < H) {
        const int2 o = own[base + y * W + x];
        acc += (o.x ^ o.y) + id[base + y * W + x];
      }
    }
  } else if (MODE == 1) {
    const int64_t p0 = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 1024 + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t p = p0 + i * 256;
      if (p < npix) {
        const int2 o = own[base + p];
        acc += (o.x ^ o.y) + id[base + p];
      }
    }
  } else {
    const int64_t p = (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid) * 4;
    if (p + 3 < npix) {
      const int4 a = *reinterpret_cast<const int4*>(own + base + p), b = *reinterpret_cast<const int4*>(own + base + p + 2);
      const int4 r = *reinterpret_cast<const int4*>(id + base + p);
      acc = (a.x ^ a.y) + (a.z ^ a.w) + (b.x ^ b.y) + (b.z ^ b.w) + r.x + r.y + r.z + r.w;
    }
  }
  if (acc == 0x12345678) out[0] = acc;
}

int main() {
  const int W = 602, H = 384, N = 512;
  const int64_t npix = (int64_t)W * H;
  int2* own; int* id; int* out;
  CHECK(hipMalloc(&own, npix * N * 8)); CHECK(hipMalloc(&id, npix * N * 4)); CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(own, 1, npix * N * 8)); CHECK(hipMemset(id, 2, npix * N * 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(a));
      const int nb = (int)((npix + 1023) / 1024);
      if (mode == 0) k<0><<<dim3((W + 31) / 32, (H + 31) / 32, N), 256>>>(own, id, W, H, out);
      if (mode == 1) k<1><<<dim3(nb, 1, N), 256>>>(own, id, W, H, out);
      if (mode == 2) k<2><<<dim3(nb, 1, N), 256>>>(own, id, W, H, out);
      if (mode == 3) k<3><<<dim3((W + 63) / 64, (H + 15) / 16, N), 256>>>(own, id, W, H, out);
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    printf("mode %d: %.3f ms  %.0f GB/s\n", mode, best, npix * N * 12.0 / best / 1e6);
  }
  return 0;
}
