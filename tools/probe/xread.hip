// What does the memory system give for the input layer's read pattern of x?  (round 5)
// x: uint16 [4096][32768] (268 MB).  Grid 16 row blocks x 16 k-splits of 512 threads, as
// count_gemm_fwd_kernel: workgroup (bx, by) reads rows 256 bx .. + 255, counts 2048 by .. + 2047.
//   A  per 32-count chunk: every row's 64 bytes (4 lanes x 16 B per row), chunk after chunk, two
//      chunks in flight -- the kernel's pattern;
//   B  the same 16-byte pieces, but the requests of EIGHT consecutive chunks of a row issued
//      back to back (512 contiguous bytes per row and burst), two bursts in flight;
//   C  32 lanes per row: an instruction reads two rows x 512 contiguous bytes;
//   D  each workgroup streams a contiguous 1 MB (no row structure): the ceiling.
// build: hipcc --offload-arch=gfx950 -O3 xread.hip -o xread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 32768, ROWS = 4096;
__device__ __forceinline__ void use(u32x4& acc, const u32x4& v) { acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
template <int V>
__global__ __launch_bounds__(512) void k(const uint16_t* __restrict__ X, unsigned* out) {
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * 256, k0 = blockIdx.y * 2048;
  u32x4 acc = {0u, 0u, 0u, 0u};
  if (V == 0) {
    const uint16_t* p0 = X + (size_t)(m0 + (tid >> 2)) * LD + k0 + (tid & 3) * 8;
    const uint16_t* p1 = p0 + (size_t)128 * LD;
    u32x4 a[2][2];
    a[0][0] = *(const u32x4*)(p0); a[0][1] = *(const u32x4*)(p1);
    a[1][0] = *(const u32x4*)(p0 + 32); a[1][1] = *(const u32x4*)(p1 + 32);
    for (int c = 0; c < 64; c += 2) {
      use(acc, a[0][0]); use(acc, a[0][1]);
      if (c + 2 < 64) { a[0][0] = *(const u32x4*)(p0 + (c + 2) * 32); a[0][1] = *(const u32x4*)(p1 + (c + 2) * 32); }
      __builtin_amdgcn_s_barrier();
      use(acc, a[1][0]); use(acc, a[1][1]);
      if (c + 3 < 64) { a[1][0] = *(const u32x4*)(p0 + (c + 3) * 32); a[1][1] = *(const u32x4*)(p1 + (c + 3) * 32); }
      __builtin_amdgcn_s_barrier();
    }
  } else if (V == 1) {
    const uint16_t* p0 = X + (size_t)(m0 + (tid >> 2)) * LD + k0 + (tid & 3) * 8;
    const uint16_t* p1 = p0 + (size_t)128 * LD;
    u32x4 a[2][16];
    auto burst = [&](int g, int s) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        a[s][2 * c] = *(const u32x4*)(p0 + (8 * g + c) * 32);
        a[s][2 * c + 1] = *(const u32x4*)(p1 + (8 * g + c) * 32);
      }
    };
    burst(0, 0); burst(1, 1);
    for (int g = 0; g < 8; g += 2) {
#pragma unroll
      for (int c = 0; c < 16; ++c) { use(acc, a[0][c]); if (c & 1) __builtin_amdgcn_s_barrier(); }
      if (g + 2 < 8) burst(g + 2, 0);
#pragma unroll
      for (int c = 0; c < 16; ++c) { use(acc, a[1][c]); if (c & 1) __builtin_amdgcn_s_barrier(); }
      if (g + 3 < 8) burst(g + 3, 1);
    }
  } else if (V == 2) {
    // 32 lanes per row: instruction i of a wave reads rows 2 i, 2 i + 1 of its 32 rows x 512 B
    const int w = tid >> 6, lane = tid & 63;
    const uint16_t* p = X + (size_t)(m0 + 32 * w + (lane >> 5)) * LD + k0 + (lane & 31) * 8;
    u32x4 a[2][16];
    auto burst = [&](int g, int s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[s][i] = *(const u32x4*)(p + (size_t)(2 * i) * LD + g * 256);
    };
    burst(0, 0); burst(1, 1);
    for (int g = 0; g < 8; g += 2) {
#pragma unroll
      for (int c = 0; c < 16; ++c) { use(acc, a[0][c]); if (c & 1) __builtin_amdgcn_s_barrier(); }
      if (g + 2 < 8) burst(g + 2, 0);
#pragma unroll
      for (int c = 0; c < 16; ++c) { use(acc, a[1][c]); if (c & 1) __builtin_amdgcn_s_barrier(); }
      if (g + 3 < 8) burst(g + 3, 1);
    }
  } else {
    const size_t wg = blockIdx.y * gridDim.x + blockIdx.x;
    const uint16_t* p = X + wg * (size_t)(256 * 2048) + tid * 8;
    u32x4 a[2][16];
    auto burst = [&](int g, int s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[s][i] = *(const u32x4*)(p + (size_t)(16 * g + i) * 4096);
    };
    burst(0, 0); burst(1, 1);
    for (int g = 0; g < 8; g += 2) {
#pragma unroll
      for (int c = 0; c < 16; ++c) use(acc, a[0][c]);
      if (g + 2 < 8) burst(g + 2, 0);
#pragma unroll
      for (int c = 0; c < 16; ++c) use(acc, a[1][c]);
      if (g + 3 < 8) burst(g + 3, 1);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
int main() {
  // eight matrices (2.1 GB, eight times the infinity cache) read one after the other by eight
  // launches: every launch streams from HBM, nothing is flushed in between (a flush by memset
  // leaves a gigabyte of dirty lines draining under the timed reads)
  constexpr int NM = 8;
  uint16_t* X; unsigned* out;
  const size_t elems = (size_t)ROWS * LD;
  (void)hipMalloc(&X, NM * elems * 2); (void)hipMemset(X, 1, NM * elems * 2);
  (void)hipMalloc(&out, 64);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[4] = {"A chunk by chunk (64 B per row and request)", "B bursts of 8 chunks (512 B per row)",
                          "C two rows x 512 B per instruction", "D contiguous 1 MB per workgroup"};
  for (int v = 0; v < 4; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      for (int m = 0; m < NM; ++m) {
        const uint16_t* x = X + m * elems;
        if (v == 0) k<0><<<dim3(16, 16), 512>>>(x, out);
        if (v == 1) k<1><<<dim3(16, 16), 512>>>(x, out);
        if (v == 2) k<2><<<dim3(16, 16), 512>>>(x, out);
        if (v == 3) k<3><<<dim3(16, 16), 512>>>(x, out);
      }
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms / NM < best) best = ms / NM;
    }
    printf("%-50s %7.1f us per 268 MB matrix  %5.2f TB/s\n", names[v], best * 1e3f,
           268.4e6 / (best * 1e-3) / 1e12);
  }
  return 0;
}
