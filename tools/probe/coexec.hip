// Does a VALU wave make progress while another wave of the same SIMD keeps the fp32 MFMA pipe
// busy?  8 waves per workgroup (w % 4 -> SIMD): waves 0-3 run MFMA chains, waves 4-7 v_fma chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>  // 0: fp32 32x32x2, 1: bf16 32x32x16
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int mode) {
  const int w = threadIdx.x >> 6;
  float res = 0.f;
  if (w < 4) {
    if (mode & 1) {
      f32x16 a0, a1;
      for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
      float x = threadIdx.x * 1e-3f, y = 1.0f;
      bf16x8 xb, yb;
      for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)x; yb[i] = (__bf16)y; }
      for (int i = 0; i < n_mfma; i += 2) {
        if (KIND == 0) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        } else {
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yb, xb, a1, 0, 0, 0);
        }
      }
      for (int i = 0; i < 16; ++i) res += a0[i] + a1[i];
    }
  } else {
    if (mode & 2) {
      float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
      float m = 1.000001f, e = 1e-7f;
      for (int i = 0; i < n_valu; i += 4) {
        a = fmaf(a, m, e); b = fmaf(b, m, e); c = fmaf(c, m, e); d = fmaf(d, m, e);
      }
      res = a + b + c + d;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

// cross-wave again, with the MFMA waves pacing themselves: VAR 0: one dependent chain back to
// back; VAR 1: one chain with s_nop padding (48 cycles) after every MFMA; VAR 2: s_sleep 1
template <int VAR>
__global__ __launch_bounds__(512) void k3(float* out, int n_mfma, int n_valu, int mode) {
  const int w = threadIdx.x >> 6;
  float res = 0.f;
  if (w < 4) {
    if (mode & 1) {
      f32x16 a0;
      for (int i = 0; i < 16; ++i) a0[i] = 0.f;
      float x = threadIdx.x * 1e-3f, y = 1.0f;
      for (int i = 0; i < n_mfma; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        if (VAR == 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        if (VAR == 2) asm volatile("s_sleep 1" ::: "memory");
      }
      for (int i = 0; i < 16; ++i) res += a0[i];
    }
  } else {
    if (mode & 2) {
      float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
      float m = 1.000001f, e = 1e-7f;
      for (int i = 0; i < n_valu; i += 4) {
        a = fmaf(a, m, e); b = fmaf(b, m, e); c = fmaf(c, m, e); d = fmaf(d, m, e);
      }
      res = a + b + c + d;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int VAR>
static void run3() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int mode = 1; mode <= 3; ++mode) {
    // 256 blocks: one workgroup per CU -> exactly one MFMA wave and one VALU wave per SIMD
    hipLaunchKernelGGL(k3<VAR>, dim3(256), dim3(512), 0, 0, out, 8192, 8192 * 12, mode);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k3<VAR>, dim3(256), dim3(512), 0, 0, out, 8192, 8192 * 12, mode);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("cross-wave var %d, mode %d (%s): %.3f ms\n", VAR, mode, mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms / 5);
  }
}

// same-wave interleave: per MFMA (64 cycles) PER independent v_fma (4 cycles each); one wave per SIMD
template <int PER, int MODE>
__global__ __launch_bounds__(256) void k2(float* out, int n_mfma) {
  f32x16 a0;
  for (int i = 0; i < 16; ++i) a0[i] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const float m = 1.000001f, e = 1e-7f;
  for (int i = 0; i < n_mfma; ++i) {
    if (MODE & 1) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    if (MODE & 2) {
#pragma unroll
      for (int u = 0; u < PER; ++u) v[u & 7] = fmaf(v[u & 7], m, e);
    }
  }
  float res = 0.f;
  for (int i = 0; i < 16; ++i) res += a0[i];
  for (int i = 0; i < 8; ++i) res += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = res;
}
template <int PER, int MODE>
static void run2m() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k2<PER, MODE>), dim3(256), dim3(256), 0, 0, out, 8192);
  (void)hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k2<PER, MODE>), dim3(256), dim3(256), 0, 0, out, 8192);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("same-wave, %2d fma per MFMA, mode %d (%s): %.3f ms\n", PER, MODE, MODE == 1 ? "MFMA only" : MODE == 2 ? "VALU only" : "both", ms / 5);
}
template <int PER>
static void run2() { run2m<PER, 1>(); run2m<PER, 2>(); run2m<PER, 3>(); }

template <int KIND>
static void run(const char* name, int n_mfma, int n_valu) {
  float* out; (void)hipMalloc(&out, 1024 * 512 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(512), 0, 0, out, n_mfma, n_valu, mode);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(512), 0, 0, out, n_mfma, n_valu, mode);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%s mode %d (%s): %.3f ms\n", name, mode, mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms / 5);
  }
}
int main() {
  run<0>("fp32 32x32x2 ", 4096, 4096 * 16);
  run<1>("bf16 32x32x16", 8192, 4096 * 16);
  run3<0>();
  run3<1>();
  run3<2>();
  run2<8>();
  run2<12>();
  return 0;
}
