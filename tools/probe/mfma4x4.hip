// Layout of v_mfma_f32_4x4x1_16b_f32: 16 blocks of 4x4 outer products.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* o) {
  const int l = threadIdx.x;
  f4 c = {0.f, 0.f, 0.f, 0.f};
  // A[blk][i] = 100*blk + i + 1 ; B[blk][j] = 10*(j+1) (+ blk/1000)
  const float a = 100.f * (l / 4) + (l % 4) + 1.f;
  const float b = 10.f * ((l % 4) + 1) + 0.001f * (l / 4);
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) o[l * 4 + i] = c[i];
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      // hypothesis: lane l, reg i = D[blk = l/4][row i][col j = l%4] = A[blk][i] * B[blk][j]
      const int blk = l / 4, j = l % 4;
      const float want = (100.f * blk + i + 1.f) * (10.f * (j + 1) + 0.001f * blk);
      if (fabsf(h[l * 4 + i] - want) > 1e-3f * fabsf(want)) ++bad;
    }
  printf("hypothesis D[lane l][reg i] = A[l/4][i] * B[l/4][l%%4]: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  for (int l = 0; l < 8; ++l) printf("lane %d: %.3f %.3f %.3f %.3f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
