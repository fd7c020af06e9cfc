// Dev probe (GPU box): what rate of random 64-byte sector requests HBM sustains — the access pattern of the owner-pair loads of
// the tile growers (8-byte elements, every lane its own sector, region far larger than the caches), as L1-bypassing agent-scope
// loads (the growers' form) and as plain loads, with 8 / 16 / 32 waves per CU and 1 or 4 independent loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int ILP>
__global__ __launch_bounds__(64) void k_gather(const unsigned long long* __restrict__ buf, uint64_t n, int steps, unsigned* out) {
  const unsigned wave = blockIdx.x, lane = threadIdx.x;
  uint64_t x = (uint64_t)wave * 0x9E3779B97F4A7C15ull + lane * 0xD1B54A32D192ED03ull + 12345u;
  unsigned acc = 0;
  for (int s = 0; s < steps; ++s) {
    unsigned long long v[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      const uint64_t i = (x >> 20) % n;
      if (MODE == 0) v[u] = __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v[u] = buf[i];
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) acc += (unsigned)v[u];
    x += acc & 1;                                        // the next addresses depend on the data: one round trip per step
  }
  if (lane == 0) out[wave] = acc;
}

template <int MODE, int ILP>
void run(int wavesPerCU, size_t bytes) {
  const uint64_t n = bytes / 8;
  unsigned long long* buf; unsigned* out;
  const int waves = 256 * wavesPerCU;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4 * waves));
  CK(hipMemset(buf, 0, bytes));
  const int steps = 4000 / ILP;
  size_t pad = wavesPerCU >= 32 ? 0 : (160 * 1024 / wavesPerCU) - 1024;   // LDS padding caps the occupancy
  if (pad > 65536 - 256) pad = 65536 - 256;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_gather<MODE, ILP>), dim3(waves), dim3(64), pad, 0, buf, n, 10, out);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k_gather<MODE, ILP>), dim3(waves), dim3(64), pad, 0, buf, n, steps, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double req = (double)waves * 64 * steps * ILP;
  printf("%-6s %d in flight, %2d waves/CU, %5.1f GB region: %6.2f G sector requests/s = %5.2f TB/s of 64-B sectors, %6.2f us per step\n",
         MODE == 0 ? "sc1" : "plain", ILP, wavesPerCU, bytes / 1e9, req / ms / 1e6, req * 64 / ms / 1e9, ms * 1e3 / steps);
  CK(hipFree(buf)); CK(hipFree(out));
}

int main() {
  const size_t big = (size_t)6 << 30;
  for (int w : {8, 16, 32}) { run<0, 1>(w, big); run<0, 4>(w, big); run<1, 1>(w, big); run<1, 4>(w, big); }
  run<0, 4>(32, (size_t)128 << 20); run<1, 4>(32, (size_t)128 << 20);      // a region the Infinity Cache holds
  return 0;
}
