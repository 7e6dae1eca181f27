// How much of the HBM read rate does a launch reach as a function of the waves per CU that issue
// loads?  2 GB streamed contiguously (16-byte loads, NL loads in flight per thread), by
// `wgs` workgroups of `threads` threads each (dynamic LDS padding pins the workgroups per CU).
// build: hipcc --offload-arch=gfx950 -O3 xread2.hip -o xread2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NL>
__global__ void k(const u32x4* __restrict__ X, size_t n16, unsigned* out) {
  extern __shared__ char pad[];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (; i + (NL - 1) * stride < n16; i += NL * stride) {
    u32x4 a[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) a[u] = X[i + u * stride];
#pragma unroll
    for (int u = 0; u < NL; ++u) { acc.x ^= a[u].x; acc.y += a[u].y; acc.z ^= a[u].z; acc.w += a[u].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) { out[0] = 1; pad[threadIdx.x] = 1; }
}
int main() {
  // (2 GB, eight times the infinity cache: a pass streams from HBM without any flush between the
  //  repetitions -- a flush by memset leaves a gigabyte of dirty lines draining under the read)
  const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
  u32x4* X; unsigned* out;
  (void)hipMalloc(&X, bytes); (void)hipMemset(X, 1, bytes);
  (void)hipMalloc(&out, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct Cfg { int wgs, threads, lds; const char* what; };
  const Cfg cfgs[] = {
      {256, 256, 140 * 1024, "1 wg/CU x 4 waves"},   {256, 512, 140 * 1024, "1 wg/CU x 8 waves"},
      {256, 1024, 140 * 1024, "1 wg/CU x 16 waves"}, {512, 512, 70 * 1024, "2 wg/CU x 8 waves"},
      {512, 1024, 70 * 1024, "2 wg/CU x 16 waves"},  {1024, 512, 36 * 1024, "4 wg/CU x 8 waves"},
      {2048, 256, 0, "8 wg/CU x 4 waves"},           {8192, 256, 0, "grid-stride, 8192 x 256"}};
  (void)hipDeviceSynchronize();
  for (const Cfg& c : cfgs) {
    for (int nl : {4, 16}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        if (nl == 4) {
          (void)hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
          k<4><<<c.wgs, c.threads, c.lds>>>(X, n16, out);
        } else {
          (void)hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
          k<16><<<c.wgs, c.threads, c.lds>>>(X, n16, out);
        }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%-26s %2d loads in flight per thread: %7.1f us  %5.2f TB/s\n", c.what, nl, best * 1e3f,
             bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
