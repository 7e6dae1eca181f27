// ds_read_b64_tr_b16 on gfx950: which 16-bit elements a lane receives, given the per-lane
// addresses.  Prints, for a few address patterns, out[lane][j] = index (in shorts) of the element
// delivered to lane `lane`, register half j.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode, int pitch) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, q = l >> 4;
  int off;                                 // in shorts
  if (mode == 0) off = 4 * l;              // lane-linear
  else off = (4 * q + (i >> 2)) * pitch + 4 * (i & 3);   // block of 4 rows x 16 cols per 16 lanes
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; (void)hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    const int pitch = mode == 0 ? 0 : 72;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode, pitch);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d pitch %d\n", mode, pitch);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]);
      if (mode == 1) {
        printf("   (row,col):");
        for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / pitch, h[l * 4 + j] % pitch);
      }
      printf("\n");
    }
  }
  return 0;
}
