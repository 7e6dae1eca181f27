// Round 4: the co-execution question of coexec.hip asked again for the bf16 matrix instructions
// the decoder-head kernel runs on (v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16), with
// inline-asm loops (no compiler scheduling):
//   A. two waves of one SIMD, one issuing MFMAs (4 independent accumulators), one issuing v_fma
//      chains; NOP >= 0 pads each MFMA with `s_nop NOP`, PV / PM are the waves' s_setprio
//   B. ONE wave per SIMD interleaving PER v_fma after every MFMA
//   C. two waves of one SIMD, BOTH running the interleaved stream of B (the shape of a kernel
//      whose waves are all in the same phase)
// "both" == max(MFMA only, VALU only) means the matrix pipe and the VALU co-execute, == sum
// means they serialise.  Per wave the work is fixed.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define FMA8                                                                           \
  "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\t"                              \
  "v_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"                              \
  "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"                              \
  "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"

// SHAPE 0: 32x32x16 (f32x16 accumulators), 1: 16x16x32 (f32x4 accumulators)
template <int SHAPE, int NOP, int PV, int PM>
__global__ __launch_bounds__(512) void pair_kernel(unsigned long long* out, unsigned* role_of,
                                                   int n_mfma, int n_valu, int mode) {
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
  f32x4 c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
  for (int i = 0; i < 4; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
  bf16x8 xb, yb;
  for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(threadIdx.x * 1e-3f); yb[i] = (__bf16)1.0f; }
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const float m = 1.000001f, e = 1e-7f;
  unsigned long long t0, t1;
  if (mfma_role) __builtin_amdgcn_s_setprio(PM); else __builtin_amdgcn_s_setprio(PV);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  if (mfma_role) {
    if (mode & 1)
      for (int i = 0; i < n_mfma; i += 4) {
        if (SHAPE == 0) {
          if (NOP >= 0)
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\ts_nop %6\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %5, %4, %1\n\ts_nop %6\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\ts_nop %6\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %5, %4, %3\n\ts_nop %6"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                : "v"(xb), "v"(yb), "n"(NOP >= 0 ? NOP : 0));
          else
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %5, %4, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %5, %4, %3"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(xb), "v"(yb));
        } else {
          if (NOP >= 0)
            asm volatile(
                "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\ts_nop %6\n\t"
                "v_mfma_f32_16x16x32_bf16 %1, %5, %4, %1\n\ts_nop %6\n\t"
                "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\ts_nop %6\n\t"
                "v_mfma_f32_16x16x32_bf16 %3, %5, %4, %3\n\ts_nop %6"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                : "v"(xb), "v"(yb), "n"(NOP >= 0 ? NOP : 0));
          else
            asm volatile(
                "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t"
                "v_mfma_f32_16x16x32_bf16 %1, %5, %4, %1\n\t"
                "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\t"
                "v_mfma_f32_16x16x32_bf16 %3, %5, %4, %3"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(xb), "v"(yb));
        }
      }
  } else {
    if (mode & 2)
      for (int i = 0; i < n_valu; i += 8)
        asm volatile(FMA8
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),
                       "+v"(v7)
                     : "v"(m), "v"(e));
  }
  float res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) res += a0[i] + a1[i] + a2[i] + a3[i];
  for (int i = 0; i < 4; ++i) res += c0[i] + c1[i] + c2[i] + c3[i];
  asm volatile("s_nop 0" ::"v"(res));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    role_of[blockIdx.x * 8 + (threadIdx.x >> 6)] = simd | (mfma_role ? 16u : 0u);
  }
  if (res == 12345.678f) out[0] = 0;
}

template <int SHAPE, int NOP, int PV, int PM>
static void run_pair() {
  unsigned long long* out; unsigned* role;
  (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&role, 256 * 8 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int n_mfma = SHAPE == 0 ? 16384 : 32768, n_valu = 8192 * 16;
  printf("pair %s s_nop %2d prio V%d/M%d:", SHAPE == 0 ? "32x32x16" : "16x16x32", NOP, PV, PM);
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL((pair_kernel<SHAPE, NOP, PV, PM>), dim3(256), dim3(512), 0, 0, out, role,
                       n_mfma, n_valu, mode);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r)
      hipLaunchKernelGGL((pair_kernel<SHAPE, NOP, PV, PM>), dim3(256), dim3(512), 0, 0, out, role,
                         n_mfma, n_valu, mode);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h[8]; unsigned hs[8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, role, sizeof(hs), hipMemcpyDeviceToHost);
    unsigned long long tm = 0, tv = 0;
    for (int w = 0; w < 8; ++w) {
      if (hs[w] & 16) { if (h[w] > tm) tm = h[w]; } else if (h[w] > tv) tv = h[w];
    }
    printf("  %s %.3f ms (M %llu V %llu)", mode == 1 ? "MFMA" : mode == 2 ? "VALU" : "both",
           ms / 5, tm, tv);
  }
  printf("\n");
}

// B / C: every wave runs { MFMA ; PER x v_fma } x n; WAVES = 4 (one per SIMD) or 8 (two)
template <int SHAPE, int PER, int MODE>
__global__ __launch_bounds__(512) void inter_kernel(float* out, int n_mfma) {
  f32x16 a0, a1;
  f32x4 c0, c1;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
  for (int i = 0; i < 4; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  bf16x8 xb, yb;
  for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(threadIdx.x * 1e-3f); yb[i] = (__bf16)1.0f; }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const float m = 1.000001f, e = 1e-7f;
  for (int i = 0; i < n_mfma; i += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (MODE & 1) {
        if (SHAPE == 0) {
          if (half == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(xb), "v"(yb));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(yb), "v"(xb));
        } else {
          if (half == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(xb), "v"(yb));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(yb), "v"(xb));
        }
      }
      if (MODE & 2) {
#pragma unroll
        for (int u = 0; u < PER; ++u)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(m), "v"(e));
      }
    }
  }
  float res = 0.f;
  for (int i = 0; i < 16; ++i) res += a0[i] + a1[i];
  for (int i = 0; i < 4; ++i) res += c0[i] + c1[i];
  for (int i = 0; i < 8; ++i) res += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int SHAPE, int PER, int MODE>
static float run_inter1(int threads) {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int n = SHAPE == 0 ? 8192 : 16384;
  hipLaunchKernelGGL((inter_kernel<SHAPE, PER, MODE>), dim3(256), dim3(threads), 0, 0, out, n);
  (void)hipEventRecord(a);
  for (int r = 0; r < 5; ++r)
    hipLaunchKernelGGL((inter_kernel<SHAPE, PER, MODE>), dim3(256), dim3(threads), 0, 0, out, n);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipFree(out);
  return ms / 5;
}
template <int SHAPE, int PER>
static void run_inter() {
  for (int threads = 256; threads <= 512; threads += 256) {
    const float m1 = run_inter1<SHAPE, PER, 1>(threads), m2 = run_inter1<SHAPE, PER, 2>(threads),
                m3 = run_inter1<SHAPE, PER, 3>(threads);
    const int n = SHAPE == 0 ? 8192 : 16384;
    const double cyc = 2.4e6 / n / (threads / 256);     // ms -> cycles per MFMA slot per wave
    printf("interleave %s, %2d v_fma per MFMA, %d wave(s)/SIMD: MFMA %.3f  VALU %.3f  both %.3f ms"
           "  (per MFMA per wave: %.1f / %.1f / %.1f cycles at 2.4 GHz)\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", PER, threads / 256, m1, m2, m3, m1 * cyc,
           m2 * cyc, m3 * cyc);
  }
}

// pair test again with other VALU instruction kinds in the VALU wave (OP): 0 v_fma_f32,
// 1 v_exp_f32, 2 v_log_f32, 3 v_rcp_f32, 4 v_cmp + v_cndmask (through vcc), 5 v_perm_b32,
// 6 v_and + v_sub (the bf16 cut), 7 ds_bpermute-free cross-lane: v_mov_dpp row_shr
template <int OP>
__global__ __launch_bounds__(512) void pair_op_kernel(unsigned long long* out, unsigned* role_of,
                                                      int n_mfma, int n_valu, int mode) {
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
  bf16x8 xb, yb;
  for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(threadIdx.x * 1e-3f); yb[i] = (__bf16)1.0f; }
  float v0 = threadIdx.x + 1.f, v1 = 1.5f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const float m = 1.000001f, e = 1e-7f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  if (mfma_role) {
    if (mode & 1)
      for (int i = 0; i < n_mfma; i += 4)
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %1, %5, %4, %1\n\t"
            "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\t"
            "v_mfma_f32_32x32x16_bf16 %3, %5, %4, %3"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(xb), "v"(yb));
  } else {
    if (mode & 2)
      for (int i = 0; i < n_valu; i += 8) {
#define OP8(INS)                                                                           \
  asm volatile(INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t"     \
               INS " %4, %4\n\t" INS " %5, %5\n\t" INS " %6, %6\n\t" INS " %7, %7"         \
               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7))
        if (OP == 0)
          asm volatile(FMA8
                       : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),
                         "+v"(v7)
                       : "v"(m), "v"(e));
        else if (OP == 1) OP8("v_exp_f32");
        else if (OP == 2) OP8("v_log_f32");
        else if (OP == 3) OP8("v_rcp_f32");
        else if (OP == 4)
          asm volatile(
              "v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t"
              "v_cmp_gt_f32 vcc, %2, %3\n\tv_cndmask_b32 %2, %2, %3, vcc\n\t"
              "v_cmp_gt_f32 vcc, %4, %5\n\tv_cndmask_b32 %4, %4, %5, vcc\n\t"
              "v_cmp_gt_f32 vcc, %6, %7\n\tv_cndmask_b32 %6, %6, %7, vcc"
              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
              :: "vcc");
        else if (OP == 5)
          asm volatile(
              "v_perm_b32 %0, %0, %1, %8\n\tv_perm_b32 %1, %1, %2, %8\n\t"
              "v_perm_b32 %2, %2, %3, %8\n\tv_perm_b32 %3, %3, %4, %8\n\t"
              "v_perm_b32 %4, %4, %5, %8\n\tv_perm_b32 %5, %5, %6, %8\n\t"
              "v_perm_b32 %6, %6, %7, %8\n\tv_perm_b32 %7, %7, %0, %8"
              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
              : "s"(0x07060302u));
        else if (OP == 6)
          asm volatile(
              "v_and_b32 %0, 0xffff0000, %1\n\tv_sub_f32 %1, %1, %0\n\t"
              "v_and_b32 %2, 0xffff0000, %3\n\tv_sub_f32 %3, %3, %2\n\t"
              "v_and_b32 %4, 0xffff0000, %5\n\tv_sub_f32 %5, %5, %4\n\t"
              "v_and_b32 %6, 0xffff0000, %7\n\tv_sub_f32 %7, %7, %6"
              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        else
          asm volatile(
              "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
              "v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf"
              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
      }
  }
  float res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) res += a0[i] + a1[i] + a2[i] + a3[i];
  asm volatile("s_nop 0" ::"v"(res));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    role_of[blockIdx.x * 8 + (threadIdx.x >> 6)] = simd | (mfma_role ? 16u : 0u);
  }
  if (res == 12345.678f) out[0] = 0;
}
template <int OP>
static void run_pair_op(const char* name) {
  unsigned long long* out; unsigned* role;
  (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&role, 256 * 8 * 4);
  const int n_mfma = 16384, n_valu = 8192 * 8;
  printf("pair 32x32x16 vs %-22s:", name);
  for (int mode = 1; mode <= 3; ++mode) {
    for (int r = 0; r < 3; ++r)
      hipLaunchKernelGGL((pair_op_kernel<OP>), dim3(256), dim3(512), 0, 0, out, role, n_mfma,
                         n_valu, mode);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; unsigned hs[8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, role, sizeof(hs), hipMemcpyDeviceToHost);
    unsigned long long tm = 0, tv = 0;
    for (int w = 0; w < 8; ++w) {
      if (hs[w] & 16) { if (h[w] > tm) tm = h[w]; } else if (h[w] > tv) tv = h[w];
    }
    printf("  %s M %llu V %llu", mode == 1 ? "MFMA" : mode == 2 ? "VALU" : "both", tm, tv);
  }
  printf("\n");
}

// pair test with the MFMA wave on NACC accumulator chains (1: every MFMA depends on the one
// before) and LDSR ds_read_b128 between consecutive MFMAs (the shape of the kernel's GEMM3 / GEMM2)
template <int NACC, int LDSR>
__global__ __launch_bounds__(512) void pair_dep_kernel(unsigned long long* out, unsigned* role_of,
                                                       int n_mfma, int n_valu, int mode) {
  __shared__ unsigned arrived[4];
  __shared__ __attribute__((aligned(16))) float buf[4096];
  if (threadIdx.x < 4) arrived[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < 4096; i += 512) buf[i] = 1.f;
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
  bf16x8 xb, yb;
  for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(threadIdx.x * 1e-3f); yb[i] = (__bf16)1.0f; }
  float v0 = threadIdx.x + 1.f, v1 = 1.5f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const float m = 1.000001f, e = 1e-7f;
  const unsigned la = (unsigned)(size_t)(buf) + (threadIdx.x & 63) * 16;
  f32x4 l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  if (mfma_role) {
    if (mode & 1)
      for (int i = 0; i < n_mfma; i += 4) {
#define DEPM(ACC)                                                                          \
  do {                                                                                     \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(xb), "v"(yb)); \
    if (LDSR >= 1) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"(la));               \
    if (LDSR >= 2) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(l1) : "v"(la));   \
  } while (0)
        if (NACC == 1) { DEPM(a0); DEPM(a0); DEPM(a0); DEPM(a0); }
        else if (NACC == 2) { DEPM(a0); DEPM(a1); DEPM(a0); DEPM(a1); }
        else { DEPM(a0); DEPM(a1); DEPM(a2); DEPM(a3); }
        if (LDSR) asm volatile("s_waitcnt lgkmcnt(0)");
      }
  } else {
    if (mode & 2)
      for (int i = 0; i < n_valu; i += 8)
        asm volatile(FMA8
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),
                       "+v"(v7)
                     : "v"(m), "v"(e));
  }
  float res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + l0[0] + l1[0];
  for (int i = 0; i < 16; ++i) res += a0[i] + a1[i] + a2[i] + a3[i];
  asm volatile("s_nop 0" ::"v"(res));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    role_of[blockIdx.x * 8 + (threadIdx.x >> 6)] = simd | (mfma_role ? 16u : 0u);
  }
  if (res == 12345.678f) out[0] = 0;
}
template <int NACC, int LDSR>
static void run_pair_dep() {
  unsigned long long* out; unsigned* role;
  (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&role, 256 * 8 * 4);
  const int n_mfma = 16384, n_valu = 8192 * 8;
  printf("pair 32x32x16 on %d chain(s), %d ds_read_b128 per MFMA, vs v_fma:", NACC, LDSR);
  for (int mode = 1; mode <= 3; ++mode) {
    for (int r = 0; r < 3; ++r)
      hipLaunchKernelGGL((pair_dep_kernel<NACC, LDSR>), dim3(256), dim3(512), 0, 0, out, role,
                         n_mfma, n_valu, mode);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; unsigned hs[8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, role, sizeof(hs), hipMemcpyDeviceToHost);
    unsigned long long tm = 0, tv = 0;
    for (int w = 0; w < 8; ++w) {
      if (hs[w] & 16) { if (h[w] > tm) tm = h[w]; } else if (h[w] > tv) tv = h[w];
    }
    printf("  %s M %llu V %llu", mode == 1 ? "MFMA" : mode == 2 ? "VALU" : "both", tm, tv);
  }
  printf("\n");
}

// NV VALU waves + one MFMA wave per SIMD ((NV + 1) * 256 threads): does a VALU wave slow down
// more when it shares the SIMD with another VALU wave AND the MFMA wave?
template <int NV>
__global__ __launch_bounds__((NV + 1) * 256) void trio_kernel(unsigned long long* out,
                                                              unsigned* role_of, int n_mfma,
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
  const bool mfma_role = order == NV;      // the LAST wave to arrive on a SIMD issues the MFMAs
  __syncthreads();
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
  bf16x8 xb, yb;
  for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(threadIdx.x * 1e-3f); yb[i] = (__bf16)1.0f; }
  float v0 = threadIdx.x + 1.f, v1 = 1.5f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const float m = 1.000001f, e = 1e-7f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  if (mfma_role) {
    if (mode & 1)
      for (int i = 0; i < n_mfma; i += 4)
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %1, %5, %4, %1\n\t"
            "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\t"
            "v_mfma_f32_32x32x16_bf16 %3, %5, %4, %3"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(xb), "v"(yb));
  } else {
    if (mode & 2)
      for (int i = 0; i < n_valu; i += 8)
        asm volatile(FMA8
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),
                       "+v"(v7)
                     : "v"(m), "v"(e));
  }
  float res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) res += a0[i] + a1[i] + a2[i] + a3[i];
  asm volatile("s_nop 0" ::"v"(res));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    role_of[blockIdx.x * 16 + (threadIdx.x >> 6)] = simd | (mfma_role ? 16u : 0u);
  }
  if (res == 12345.678f) out[0] = 0;
}
template <int NV>
static void run_trio() {
  unsigned long long* out; unsigned* role;
  (void)hipMalloc(&out, 256 * 16 * 8); (void)hipMalloc(&role, 256 * 16 * 4);
  const int n_mfma = 16384, n_valu = 8192 * 8;
  printf("%d VALU wave(s) + 1 MFMA wave per SIMD:", NV);
  for (int mode = 1; mode <= 3; ++mode) {
    for (int r = 0; r < 3; ++r)
      hipLaunchKernelGGL((trio_kernel<NV>), dim3(256), dim3((NV + 1) * 256), 0, 0, out, role,
                         n_mfma, n_valu, mode);
    (void)hipDeviceSynchronize();
    unsigned long long h[16]; unsigned hs[16];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, role, sizeof(hs), hipMemcpyDeviceToHost);
    unsigned long long tm = 0, tv = 0;
    for (int w = 0; w < (NV + 1) * 4; ++w) {
      if (hs[w] & 16) { if (h[w] > tm) tm = h[w]; } else if (h[w] > tv) tv = h[w];
    }
    printf("  %s M %llu V %llu", mode == 1 ? "MFMA" : mode == 2 ? "VALU" : "both", tm, tv);
  }
  printf("\n");
}

// VALU-port occupancy of one MFMA: NV VALU waves (v_fma chains, enough to saturate the port) +
// one MFMA wave per SIMD.  SHAPE 0: 32x32x16, 1: 16x16x32; ACC 0: accumulators in VGPRs ("v"),
// 1: in AGPRs ("a").  With the port saturated by the VALU waves, every cycle an MFMA blocks the
// port shows up in the VALU waves' time: port cycles per MFMA = (V_both - V_alone) / n_mfma.
template <int NV, int SHAPE, int ACC>
__global__ __launch_bounds__((NV + 1) * 256) void port_kernel(unsigned long long* out,
                                                              unsigned* role_of, int n_mfma,
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
  const bool mfma_role = order == NV;
  __syncthreads();
  f32x16 a0, a1, a2, a3;
  f32x4 c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
  for (int i = 0; i < 4; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
  bf16x8 xb, yb;
  for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(threadIdx.x * 1e-3f); yb[i] = (__bf16)1.0f; }
  float v0 = threadIdx.x + 1.f, v1 = 1.5f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const float m = 1.000001f, e = 1e-7f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  if (mfma_role) {
    if (mode & 1)
      for (int i = 0; i < n_mfma; i += 4) {
#define PORT4(INS, R0, R1, R2, R3, CON)                                                      \
  asm volatile(INS " %0, %4, %5, %0\n\t" INS " %1, %5, %4, %1\n\t" INS " %2, %4, %5, %2\n\t"  \
               INS " %3, %5, %4, %3"                                                          \
               : CON(R0), CON(R1), CON(R2), CON(R3) : "v"(xb), "v"(yb))
#define CON_V(x) "+v"(x)
#define CON_A(x) "+a"(x)
        if (SHAPE == 0 && ACC == 0) PORT4("v_mfma_f32_32x32x16_bf16", a0, a1, a2, a3, CON_V);
        if (SHAPE == 0 && ACC == 1) PORT4("v_mfma_f32_32x32x16_bf16", a0, a1, a2, a3, CON_A);
        if (SHAPE == 1 && ACC == 0) PORT4("v_mfma_f32_16x16x32_bf16", c0, c1, c2, c3, CON_V);
        if (SHAPE == 1 && ACC == 1) PORT4("v_mfma_f32_16x16x32_bf16", c0, c1, c2, c3, CON_A);
      }
  } else {
    if (mode & 2)
      for (int i = 0; i < n_valu; i += 8)
        asm volatile(FMA8
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),
                       "+v"(v7)
                     : "v"(m), "v"(e));
  }
  float res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) res += a0[i] + a1[i] + a2[i] + a3[i];
  for (int i = 0; i < 4; ++i) res += c0[i] + c1[i] + c2[i] + c3[i];
  asm volatile("s_nop 0" ::"v"(res));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    role_of[blockIdx.x * 16 + (threadIdx.x >> 6)] = simd | (mfma_role ? 16u : 0u);
  }
  if (res == 12345.678f) out[0] = 0;
}
template <int NV, int SHAPE, int ACC>
static void run_port() {
  unsigned long long* out; unsigned* role;
  (void)hipMalloc(&out, 256 * 16 * 8); (void)hipMalloc(&role, 256 * 16 * 4);
  const int n_mfma = SHAPE == 0 ? 32768 : 65536, n_valu = 8192 * 8;
  printf("port: %d VALU waves + 1 MFMA wave (%s, acc in %s):", NV,
         SHAPE == 0 ? "32x32x16" : "16x16x32", ACC ? "AGPRs" : "VGPRs");
  unsigned long long tv_alone = 0;
  for (int mode = 1; mode <= 3; ++mode) {
    for (int r = 0; r < 3; ++r)
      hipLaunchKernelGGL((port_kernel<NV, SHAPE, ACC>), dim3(256), dim3((NV + 1) * 256), 0, 0, out,
                         role, n_mfma, n_valu, mode);
    (void)hipDeviceSynchronize();
    unsigned long long h[16]; unsigned hs[16];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, role, sizeof(hs), hipMemcpyDeviceToHost);
    unsigned long long tm = 0, tv = 0;
    for (int w = 0; w < (NV + 1) * 4; ++w) {
      if (hs[w] & 16) { if (h[w] > tm) tm = h[w]; } else if (h[w] > tv) tv = h[w];
    }
    if (mode == 2) tv_alone = tv;
    printf("  %s M %llu V %llu", mode == 1 ? "MFMA" : mode == 2 ? "VALU" : "both", tm, tv);
    if (mode == 3)
      printf("  -> %.1f port cycles per MFMA (MFMAs issued while the VALU waves ran: %.0f)",
             ((double)tv - (double)tv_alone) / ((double)n_mfma * tv / tm),
             (double)n_mfma * tv / tm);
  }
  printf("\n");
}

int main() {
  run_port<3, 0, 0>();
  run_port<3, 0, 1>();
  run_port<3, 1, 0>();
  run_port<3, 1, 1>();
  run_trio<1>();
  run_trio<2>();
  run_trio<3>();
  run_pair_dep<1, 0>();
  run_pair_dep<2, 0>();
  run_pair_dep<4, 0>();
  run_pair_dep<1, 1>();
  run_pair_dep<2, 1>();
  run_pair_dep<4, 1>();
  run_pair_dep<1, 2>();
  run_pair_dep<4, 2>();
  run_pair_op<0>("v_fma_f32");
  run_pair_op<1>("v_exp_f32");
  run_pair_op<2>("v_log_f32");
  run_pair_op<3>("v_rcp_f32");
  run_pair_op<4>("v_cmp + v_cndmask (vcc)");
  run_pair_op<5>("v_perm_b32");
  run_pair_op<6>("v_and + v_sub");
  run_pair_op<7>("v_mov_dpp row_shr");
  run_pair<0, -1, 0, 0>();
  run_pair<0, -1, 1, 0>();
  run_pair<0, -1, 0, 1>();
  run_pair<0, 3, 0, 0>();
  run_pair<0, 7, 0, 0>();
  run_pair<1, -1, 0, 0>();
  run_pair<1, -1, 1, 0>();
  run_pair<1, 1, 0, 0>();
  run_pair<1, 3, 0, 0>();
  run_inter<0, 2>();
  run_inter<0, 4>();
  run_inter<0, 6>();
  run_inter<0, 8>();
  run_inter<0, 12>();
  run_inter<1, 1>();
  run_inter<1, 2>();
  run_inter<1, 3>();
  run_inter<1, 4>();
  run_inter<1, 6>();
  return 0;
}
