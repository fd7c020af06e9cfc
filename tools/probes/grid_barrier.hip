// Probe: cost of a software grid barrier on gfx950 (resident grid, one arrival counter), in the forms k_tx_tail could use.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__device__ __forceinline__ void barrier(unsigned* bar, unsigned target) {
  if (MODE == 0) {                 // every wave fences on both sides
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&bar[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    __threadfence();
  } else if (MODE == 1) {          // one release + one acquire per block, relaxed polls
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  } else if (MODE == 2) {          // no fences at all (lower bound: arrivals + polls)
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  } else {                         // two-level: one counter per 32 blocks, then the top counter
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned g = blockIdx.x >> 5, ng = (gridDim.x + 31) >> 5, round = target / gridDim.x;
      const unsigned members = min(32u, gridDim.x - g * 32u);
      const unsigned old = __hip_atomic_fetch_add(&bar[64 + 32 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == round * members) __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * ng) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* bar, int n, int* data) {
  for (int i = 1; i <= n; ++i) {
    data[(blockIdx.x * 256 + threadIdx.x) * 16] = i;     // a little traffic to fence
    barrier<MODE>(bar, (unsigned)i * gridDim.x);
  }
}

int main() {
  unsigned* bar;
  int* data;
  hipMalloc(&bar, 4096 * 4);
  hipMalloc(&data, 1024 * 256 * 16 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int n = 200;
  for (int blocks : {256, 512, 1024}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(bar, 0, 4096 * 4);
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, bar, n, data);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, bar, n, data);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, bar, n, data);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, bar, n, data);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      std::printf("blocks %4d mode %d: %.2f us per barrier\n", blocks, mode, best * 1000.f / n);
    }
  }
  return 0;
}
