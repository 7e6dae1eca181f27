// Issue rate (independent accumulator chains, one wave per SIMD) of the bf16 MFMAs of gfx950: the
// K = 32 / 16 forms against the K = 16 / 8 forms carried over from gfx90a ("_1k").  Question: does
// a half-depth product cost half the matrix-pipe time (then the padded tail of a contraction --
// H + 1 = 101 -> 112 instead of 128 -- is worth a K = 16 step)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void k(float* o, int n) {
  bf16x8 a8, b8;
  s16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 1e-3f); b8[i] = (__bf16)1.f; }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3f80 + threadIdx.x); b4[i] = 0x3f80; }
  float s = 0.f;
  if (KIND < 2) {
    f4 c[CHAINS];
    for (int q = 0; q < CHAINS; ++q) c[q] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int q = 0; q < CHAINS; ++q)
        if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[q]) : "v"(a8), "v"(b8));
        else asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(c[q]) : "v"(a4), "v"(b4));
    for (int q = 0; q < CHAINS; ++q) s += c[q][0] + c[q][3];
  } else {
    f16v c[CHAINS];
    for (int q = 0; q < CHAINS; ++q)
      for (int i = 0; i < 16; ++i) c[q][i] = 0.f;
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int q = 0; q < CHAINS; ++q)
        if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[q]) : "v"(a8), "v"(b8));
        else asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(c[q]) : "v"(a4), "v"(b4));
    for (int q = 0; q < CHAINS; ++q) s += c[q][0] + c[q][15];
  }
  o[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int CHAINS>
static void run(const char* name) {
  float* d; (void)hipMalloc(&d, 256 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int n = 1 << 15;
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(256), dim3(256), 0, 0, d, n);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(256), dim3(256), 0, 0, d, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %5.1f cycles per MFMA at 2.4 GHz (%d independent chains, one wave per SIMD)\n",
         name, ms * 1e-3 * 2.4e9 / ((double)n * CHAINS), CHAINS);
}
int main() {
  run<0, 4>("v_mfma_f32_16x16x32_bf16");
  run<1, 4>("v_mfma_f32_16x16x16_bf16 (_1k)");
  run<2, 4>("v_mfma_f32_32x32x16_bf16");
  run<3, 4>("v_mfma_f32_32x32x8_bf16 (_1k)");
  run<0, 1>("v_mfma_f32_16x16x32_bf16");
  run<1, 1>("v_mfma_f32_16x16x16_bf16 (_1k)");
  return 0;
}
