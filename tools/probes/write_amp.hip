// Probe: how many bytes does WRITE_SIZE count for the store patterns of the LSD front pass and the bookkeeping passes, by pitch?
// (k_lsd_front64 writes 24 B per scaled pixel — a float4 record and a double norm — yet its WRITE_SIZE pass says 53 B: where from?)
//   hipcc --offload-arch=gfx950 -O3 write_amp.hip -o write_amp
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out -o s -- ./write_amp ;  python3 tools/rocprof_summary.py pmc out/*.db
// Each kernel name carries its pattern and pitch; the program prints the bytes every launch SHOULD have written.
//   tile<REC|MG, DW>   : the front pass's mapping — 64 x 16 pixels per workgroup of 256 threads, one row of 64 pixels per wave and step
//   linear<REC|MG, DW> : thread = pixel of the flat plane
//   word<DW>           : the fourth word of every record only (4 of 16 bytes, what k_tx_sort's packed form writes)
//   own32<DW>          : an int2 per pixel, 32 x 32 pixels per workgroup (k_tx_round2's mapping)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int REC = 1, MG = 2;

template <int WHAT, int DW>
__global__ __launch_bounds__(256) void tile(float4* __restrict__ rec, double* __restrict__ mg, int dh) {
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int x = x0 + (i & 63), y = y0 + (i >> 6);
    if (x >= DW || y >= dh) continue;
    const int64_t o = (int64_t)blockIdx.z * DW * dh + (int64_t)y * DW + x;
    if (WHAT & REC) rec[o] = make_float4((float)x, (float)y, 1.f, 2.f);
    if (WHAT & MG) mg[o] = (double)x;
  }
}

template <int WHAT, int DW>
__global__ __launch_bounds__(256) void linear(float4* __restrict__ rec, double* __restrict__ mg, int64_t n) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  if (WHAT & REC) rec[o] = make_float4((float)o, 0.f, 1.f, 2.f);
  if (WHAT & MG) mg[o] = (double)o;
}

template <int DW>
__global__ __launch_bounds__(256) void word(float4* __restrict__ rec, int64_t n) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o < n) reinterpret_cast<int*>(rec + o)[3] = (int)o;
}

template <int DW>
__global__ __launch_bounds__(256) void own32(int2* __restrict__ own, int dh) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = blockIdx.y * 32 + i * 8 + (threadIdx.x >> 5);
    if (x < DW && y < dh) own[(int64_t)blockIdx.z * DW * dh + (int64_t)y * DW + x] = make_int2(x, y);
  }
}

template <int DW>
int run(float4* rec, double* mg, int dh, int nimg) {
  const int64_t n = (int64_t)DW * dh * nimg;
  const dim3 gt((DW + 63) / 64, (dh + 15) / 16, nimg), g32((DW + 31) / 32, (dh + 31) / 32, nimg);
  const unsigned gl = (unsigned)((n + 255) / 256);
  printf("pitch %d: %lld pixels; rec %.1f MB, mg %.1f MB, word %.1f MB, own %.1f MB\n", DW, (long long)n, n * 16 / 1e6, n * 8 / 1e6, n * 4 / 1e6,
         n * 8 / 1e6);
  hipLaunchKernelGGL((tile<REC | MG, DW>), gt, dim3(256), 0, 0, rec, mg, dh);
  hipLaunchKernelGGL((tile<REC, DW>), gt, dim3(256), 0, 0, rec, mg, dh);
  hipLaunchKernelGGL((tile<MG, DW>), gt, dim3(256), 0, 0, rec, mg, dh);
  hipLaunchKernelGGL((linear<REC | MG, DW>), dim3(gl), dim3(256), 0, 0, rec, mg, n);
  hipLaunchKernelGGL((linear<REC, DW>), dim3(gl), dim3(256), 0, 0, rec, mg, n);
  hipLaunchKernelGGL((linear<MG, DW>), dim3(gl), dim3(256), 0, 0, rec, mg, n);
  hipLaunchKernelGGL((word<DW>), dim3(gl), dim3(256), 0, 0, rec, n);
  hipLaunchKernelGGL((own32<DW>), g32, dim3(256), 0, 0, reinterpret_cast<int2*>(mg), dh);
  CHECK(hipDeviceSynchronize());
  return 0;
}

int main() {
  const int dh = 384, nimg = 512;
  float4* rec; double* mg;
  const int64_t nmax = (int64_t)640 * dh * nimg;
  CHECK(hipMalloc(&rec, nmax * 16)); CHECK(hipMalloc(&mg, nmax * 8));
  CHECK(hipMemset(rec, 0, nmax * 16)); CHECK(hipMemset(mg, 0, nmax * 8));
  CHECK(hipDeviceSynchronize());
  if (run<602>(rec, mg, dh, nimg)) return 1;
  if (run<608>(rec, mg, dh, nimg)) return 1;
  if (run<640>(rec, mg, dh, nimg)) return 1;
  printf("done\n");
  return 0;
}
