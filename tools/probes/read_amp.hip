// Probe: which of the L2's memory-side read counters give the bytes a kernel really read?  (FETCH_SIZE on gfx950 tallies a 128-byte
// request as 64: right after doubling for wide streaming reads, uncalibrated for everything else — the growers' gathers, the
// bookkeeping passes' 4- and 8-byte streams.)  Known byte counts per pattern; run under
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -- ./read_amp
//   stream<B>  : thread = element of B bytes of a flat 1.9 GB plane (4, 8, 16)
//   tile16     : the front pass's mapping over float4 records, pitch 903 (64 x 16 pixels per workgroup)
//   gather16   : one random 16-byte record per thread (the growers' neighbour reads)
//   word       : the fourth word of every record (4 of 16 bytes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T> __device__ __forceinline__ int fold(const T& v);
template <> __device__ __forceinline__ int fold<int>(const int& v) { return v; }
template <> __device__ __forceinline__ int fold<int2>(const int2& v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ int fold<int4>(const int4& v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <typename T>
__global__ __launch_bounds__(256) void stream(const T* __restrict__ p, int64_t n, int* __restrict__ out) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o < n && fold(p[o]) == 0x12345678) out[0] = 1;
}

__global__ __launch_bounds__(256) void tile16(const int4* __restrict__ rec, int dw, int dh, int* __restrict__ out) {
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
  int acc = 0;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int x = x0 + (i & 63), y = y0 + (i >> 6);
    if (x >= dw || y >= dh) continue;
    acc ^= fold(rec[(int64_t)blockIdx.z * dw * dh + (int64_t)y * dw + x]);
  }
  if (acc == 0x12345678) out[0] = 1;
}

__global__ __launch_bounds__(256) void gather16(const int4* __restrict__ rec, int64_t n, int* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t h = (i * 0x9E3779B97F4A7C15ull) >> 20;
  if (fold(rec[h % (uint64_t)n]) == 0x12345678) out[0] = 1;
}

__global__ __launch_bounds__(256) void word(const int4* __restrict__ rec, int64_t n, int* __restrict__ out) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o < n && reinterpret_cast<const int*>(rec + o)[3] == 0x12345678) out[0] = 1;
}

int main() {
  const int dw = 903, dh = 576, nimg = 256;
  const int64_t n = (int64_t)dw * dh * nimg;          // records of 16 bytes: 2.13 GB
  int4* rec; int* out;
  CHECK(hipMalloc(&rec, n * 16)); CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(rec, 1, n * 16)); CHECK(hipMemset(out, 0, 4));
  CHECK(hipDeviceSynchronize());
  printf("%lld records; stream<4> %.1f MB, stream<8> %.1f MB, stream<16> / tile16 %.1f MB, gather16 %.1f MB in 16-byte pieces (%.1f MB of 64-byte lines), word %.1f MB (records: %.1f)\n",
         (long long)n, n * 4 / 1e6, n * 8 / 1e6, n * 16 / 1e6, n / 4 * 16 / 1e6, n / 4 * 64 / 1e6, n * 4 / 1e6, n * 16 / 1e6);
  const unsigned g = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(stream<int>, dim3(g), dim3(256), 0, 0, reinterpret_cast<const int*>(rec), n, out);
  hipLaunchKernelGGL(stream<int2>, dim3(g), dim3(256), 0, 0, reinterpret_cast<const int2*>(rec), n, out);
  hipLaunchKernelGGL(stream<int4>, dim3(g), dim3(256), 0, 0, rec, n, out);
  hipLaunchKernelGGL(tile16, dim3((dw + 63) / 64, (dh + 15) / 16, nimg), dim3(256), 0, 0, rec, dw, dh, out);
  hipLaunchKernelGGL(gather16, dim3(g / 4), dim3(256), 0, 0, rec, n, out);
  hipLaunchKernelGGL(word, dim3(g), dim3(256), 0, 0, rec, n, out);
  CHECK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
