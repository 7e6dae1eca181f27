// Dependent-chain latency vs independent issue rate of the small fp32 MFMAs (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k4(float* o, int n) {
  f4 c[CHAINS];
  for (int q = 0; q < CHAINS; ++q) c[q] = f4{0.f, 0.f, 0.f, 0.f};
  const float a = threadIdx.x * 1e-3f, b = 1.f;
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int q = 0; q < CHAINS; ++q) c[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[q], 0, 0, 0);
  float s = 0.f;
  for (int q = 0; q < CHAINS; ++q) s += c[q][0] + c[q][3];
  o[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
static void run(const char* name) {
  float* d; (void)hipMalloc(&d, 256 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int n = 1 << 16;
  hipLaunchKernelGGL(k4<CHAINS>, dim3(256), dim3(256), 0, 0, d, n);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k4<CHAINS>, dim3(256), dim3(256), 0, 0, d, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.1f cycles per MFMA at 2.4 GHz (%d independent chains)\n", name,
         ms * 1e-3 * 2.4e9 / ((double)n * CHAINS), CHAINS);
}
int main() {
  run<1>("v_mfma_f32_4x4x1"); run<2>("v_mfma_f32_4x4x1"); run<4>("v_mfma_f32_4x4x1"); run<8>("v_mfma_f32_4x4x1");
  return 0;
}
