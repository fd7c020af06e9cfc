// Dev probe (GPU box): time of one dependent "8 entries x 8 neighbours" gather step per wave, the access pattern of
// the sequential LSD grower, as a function of the number of waves in flight and the record size.
//   hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip && ./lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename REC>
__global__ __launch_bounds__(128) void k_chase(const REC* __restrict__ buf, int64_t perWave, int W, int H, int steps, unsigned* out,
                                               int spread) {
  const int wave = blockIdx.x * 2 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const REC* rec = buf + (int64_t)wave * perWave;
  unsigned x = wave * 2654435761u + 12345u;
  const int pi = lane >> 3, ni = (lane & 7) < 4 ? (lane & 7) : (lane & 7) + 1;
  unsigned acc = 0;
  for (int s = 0; s < steps; ++s) {
    // 8 entries near one random centre (a BFS frontier is local), or spread over the image
    unsigned h = x * 1664525u + 1013904223u;
    const int cx = 8 + (h >> 8) % (unsigned)(W - 16), cy = 8 + (h >> 20) % (unsigned)(H - 16);
    unsigned h2 = (h ^ (pi * 0x9E3779B9u)) * 2246822519u;
    int ex = cx + (int)((h2 >> 4) % 7u) - 3, ey = cy + (int)((h2 >> 12) % 7u) - 3;
    if (spread) { ex = 8 + (h2 >> 8) % (unsigned)(W - 16); ey = 8 + (h2 >> 20) % (unsigned)(H - 16); }
    const int nx = ex + ni % 3 - 1, ny = ey + ni / 3 - 1;
    const REC r = rec[ny * W + nx];
    const unsigned v = ((const unsigned*)&r)[0];
    acc += v;
    x = __builtin_amdgcn_readfirstlane(v) + h;      // next address depends on the loaded value
  }
  if (lane == 0) out[wave] = acc + x;
}

template <typename REC>
void run(const char* name, int waves, int steps, int spread) {
  const int W = 602, H = 384;
  const int64_t perWave = (int64_t)W * H;
  REC* buf; unsigned* out;
  CK(hipMalloc(&buf, sizeof(REC) * perWave * waves));
  CK(hipMalloc(&out, 4 * waves));
  CK(hipMemset(buf, 1, sizeof(REC) * perWave * waves));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_chase<REC>, dim3(waves / 2), dim3(128), 0, 0, buf, perWave, W, H, 100, out, spread);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k_chase<REC>, dim3(waves / 2), dim3(128), 0, 0, buf, perWave, W, H, steps, out, spread);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-8s waves %5d %s: %7.3f us per dependent step\n", name, waves, spread ? "spread" : "local ", ms * 1e3 / steps);
  CK(hipFree(buf)); CK(hipFree(out));
}

int main() {
  for (int spread = 0; spread < 2; ++spread)
    for (int waves : {64, 512, 1024, 2048, 4096}) {
      run<float4>("16 B", waves, 20000, spread);
      run<float>("4 B", waves, 20000, spread);
    }
  return 0;
}
