// Cost of a grid barrier across 64 / 128 workgroups (one per CU) on gfx950, per variant:
//   0  counter, release / acquire (common.hpp grid_barrier: buffer_wbl2 + buffer_inv)
//   1  counter, relaxed agent-scope atomics (no write-back / invalidate)
//   2  flag slots: every workgroup stores its epoch into its own slot (sc1 store), wave 0 of
//      every workgroup polls the 64..128 slots with one sc1 load per lane
// build: hipcc --offload-arch=gfx950 -O3 gridbar.hip -o gridbar ; run: ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void bar0(unsigned* bar, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while ((int)(__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void bar1(unsigned* bar, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
__device__ __forceinline__ void bar2(unsigned* slots, unsigned epoch, int sleep) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(slots + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 64) {
    for (;;) {
      bool ok = true;
      for (unsigned s = threadIdx.x; s < gridDim.x; s += 64)
        ok = ok && (int)(__hip_atomic_load(slots + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
      if (__all(ok)) break;
      if (sleep) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
template <int V>
__global__ __launch_bounds__(512) void k(unsigned* bar, unsigned base, int n, float* data) {
  unsigned target = base;
  float acc = 0.f;
  for (int i = 0; i < n; ++i) {
    // a little data crossing, as the chain's chunk statistics do
    __hip_atomic_store(data + (size_t)blockIdx.x * 512 + threadIdx.x, (float)i, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    if (V == 0) bar0(bar, target += gridDim.x);
    else if (V == 1) bar1(bar, target += gridDim.x);
    else bar2(bar + 64, base + i + 1, V == 3);
    acc += __hip_atomic_load(data + (size_t)((blockIdx.x + 1) % gridDim.x) * 512 + threadIdx.x,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == -1.f) data[0] = acc;
}
int main() {
  unsigned* bar; float* data;
  hipMalloc(&bar, 4096); hipMemset(bar, 0, 4096);
  hipMalloc(&data, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 200;
  for (int wgs : {16, 64, 128}) {
    for (int v = 0; v < 4; ++v) {
      unsigned base = 0;
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(bar, 0, 4096);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (v == 0) k<0><<<wgs, 512, 65536>>>(bar, base, n, data);
        if (v == 1) k<1><<<wgs, 512, 65536>>>(bar, base, n, data);
        if (v == 2) k<2><<<wgs, 512, 65536>>>(bar, base, n, data);
        if (v == 3) k<3><<<wgs, 512, 65536>>>(bar, base, n, data);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("wgs %3d variant %d: %.2f us per barrier (+ one sc1 store and load)\n", wgs, v,
             best * 1e3f / n);
    }
  }
  return 0;
}
