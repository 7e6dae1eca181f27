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
// Placement by the wave's actual SIMD (HW_REG_HW_ID bits 5:4), one workgroup of 8 waves per CU.
//   PLACE 0 ("same SIMD"):  on every SIMD the first wave to arrive issues MFMAs, the second v_fma
//   PLACE 1 ("cross SIMD"): waves on SIMD 0/1 issue MFMAs, waves on SIMD 2/3 v_fma -- the matrix
//                           pipes of SIMD 2/3 and the VALUs of SIMD 0/1 stay idle
// The loops are inline asm (no compiler scheduling, no inserted s_nop): 4 independent MFMA
// accumulator chains / 8 independent v_fma chains per wave.  Per wave the work is fixed, so
// "both" == max(MFMA only, VALU only) means co-execution, == sum means serialisation.
// out[block*8 + wave] = cycles (s_memtime) the wave spent in its loop.
template <int PLACE>
__global__ __launch_bounds__(512) void k4(unsigned long long* out, unsigned* simd_of, int n_mfma,
                                          int n_valu, int mode) {
  __shared__ unsigned arrived[4];
  if (threadIdx.x < 4) arrived[threadIdx.x] = 0;
  __syncthreads();
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  const unsigned simd = (id >> 4) & 3;
  unsigned order = 0;
  if ((threadIdx.x & 63) == 0) order = atomicAdd(&arrived[simd], 1u);
  order = __builtin_amdgcn_readfirstlane(order);
  const bool mfma_role = PLACE == 0 ? order == 0 : simd < 2;
  __syncthreads();
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const float m = 1.000001f, e = 1e-7f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  if (mfma_role) {
    if (mode & 1)
      for (int i = 0; i < n_mfma; i += 4)
        asm volatile(
            "v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\t"
            "v_mfma_f32_32x32x2_f32 %1, %5, %4, %1\n\t"
            "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\t"
            "v_mfma_f32_32x32x2_f32 %3, %5, %4, %3"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
  } else {
    if (mode & 2)
      for (int i = 0; i < n_valu; i += 8)
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\t"
            "v_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
            "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
            "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
            : "v"(m), "v"(e));
  }
  float res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) res += a0[i] + a1[i] + a2[i] + a3[i];
  asm volatile("s_nop 0" :: "v"(res));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    simd_of[blockIdx.x * 8 + (threadIdx.x >> 6)] = simd | (mfma_role ? 16u : 0u);
  }
  if (res == 12345.678f) out[0] = 0;
}
template <int PLACE>
static void run4(const char* name) {
  unsigned long long* out; unsigned* simd;
  (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&simd, 256 * 8 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int n_mfma = 8192, n_valu = 8192 * 16;   // 64 cycles x 8192 vs ~4 cycles x 131072
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k4<PLACE>, dim3(256), dim3(512), 0, 0, out, simd, n_mfma, n_valu, mode);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r)
      hipLaunchKernelGGL(k4<PLACE>, dim3(256), dim3(512), 0, 0, out, simd, n_mfma, n_valu, mode);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h[8]; unsigned hs[8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, simd, sizeof(hs), hipMemcpyDeviceToHost);
    printf("%s mode %d (%s): %.3f ms; block 0 waves (simd/role: memtime ticks):", name, mode,
           mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms / 5);
    for (int w = 0; w < 8; ++w)
      printf(" %u%c:%llu", hs[w] & 3, (hs[w] & 16) ? 'M' : 'V', h[w]);
    printf("\n");
  }
}

// Same-SIMD pairing again (first wave of a SIMD: MFMA chains, second: v_fma chains), with the two
// knobs that could let the VALU wave in while the matrix pipe is busy:
//   NOP  >= 0: `s_nop NOP` after every MFMA -- the MFMA wave is then not "ready" with its next
//              MFMA while the pipe still works on the last one (no head-of-line blocking of the
//              SIMD's VALU-class issue slot);  -1: back-to-back
//   PV / PM  : s_setprio of the VALU / MFMA wave
template <int NOP, int PV, int PM>
__global__ __launch_bounds__(512) void k5(unsigned long long* out, unsigned* simd_of, int n_mfma,
                                          int n_valu, int mode) {
  __shared__ unsigned arrived[4];
  if (threadIdx.x < 4) arrived[threadIdx.x] = 0;
  __syncthreads();
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  const unsigned simd = (id >> 4) & 3;
  unsigned order = 0;
  if ((threadIdx.x & 63) == 0) order = atomicAdd(&arrived[simd], 1u);
  order = __builtin_amdgcn_readfirstlane(order);
  const bool mfma_role = order == 0;
  __syncthreads();
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const float m = 1.000001f, e = 1e-7f;
  unsigned long long t0, t1;
  if (mfma_role) __builtin_amdgcn_s_setprio(PM); else __builtin_amdgcn_s_setprio(PV);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  if (mfma_role) {
    if (mode & 1)
      for (int i = 0; i < n_mfma; i += 4) {
        if (NOP >= 0)
          asm volatile(
              "v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\ts_nop %6\n\t"
              "v_mfma_f32_32x32x2_f32 %1, %5, %4, %1\n\ts_nop %6\n\t"
              "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\ts_nop %6\n\t"
              "v_mfma_f32_32x32x2_f32 %3, %5, %4, %3\n\ts_nop %6"
              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y), "n"(NOP >= 0 ? NOP : 0));
        else
          asm volatile(
              "v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\t"
              "v_mfma_f32_32x32x2_f32 %1, %5, %4, %1\n\t"
              "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\t"
              "v_mfma_f32_32x32x2_f32 %3, %5, %4, %3"
              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
      }
  } else {
    if (mode & 2)
      for (int i = 0; i < n_valu; i += 8)
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\t"
            "v_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
            "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
            "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
            : "v"(m), "v"(e));
  }
  float res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) res += a0[i] + a1[i] + a2[i] + a3[i];
  asm volatile("s_nop 0" :: "v"(res));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    simd_of[blockIdx.x * 8 + (threadIdx.x >> 6)] = simd | (mfma_role ? 16u : 0u);
  }
  if (res == 12345.678f) out[0] = 0;
}
template <int NOP, int PV, int PM>
static void run5() {
  unsigned long long* out; unsigned* simd;
  (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&simd, 256 * 8 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int n_mfma = 8192, n_valu = 8192 * 8;   // 64 cycles x 8192 vs 65536 v_fma
  printf("same SIMD, s_nop %2d, prio VALU %d / MFMA %d:", NOP, PV, PM);
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL((k5<NOP, PV, PM>), dim3(256), dim3(512), 0, 0, out, simd, n_mfma, n_valu, mode);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r)
      hipLaunchKernelGGL((k5<NOP, PV, PM>), dim3(256), dim3(512), 0, 0, out, simd, n_mfma, n_valu, mode);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h[8]; unsigned hs[8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, simd, sizeof(hs), hipMemcpyDeviceToHost);
    unsigned long long tm = 0, tv = 0;
    for (int w = 0; w < 8; ++w) { if (hs[w] & 16) { if (h[w] > tm) tm = h[w]; } else if (h[w] > tv) tv = h[w]; }
    printf("  %s %.3f ms (M %llu V %llu ticks)", mode == 1 ? "MFMA" : mode == 2 ? "VALU" : "both", ms / 5, tm, tv);
  }
  printf("\n");
}

int main() {
  run5<-1, 0, 0>();
  run5<-1, 3, 0>();
  run5<-1, 0, 3>();
  run5<7, 0, 0>();
  run5<10, 0, 0>();
  run5<12, 0, 0>();
  run5<13, 0, 0>();
  run5<14, 0, 0>();
  run5<15, 0, 0>();
  run5<13, 3, 0>();
  run5<13, 0, 3>();
  run4<0>("asm, same SIMD ");
  run4<1>("asm, cross SIMD");
  run<0>("fp32 32x32x2 ", 4096, 4096 * 16);
  run<1>("bf16 32x32x16", 8192, 4096 * 16);
  run3<0>();
  run3<1>();
  run3<2>();
  run2<8>();
  run2<12>();
  return 0;
}
