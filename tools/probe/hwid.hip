#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 16 * 4);
  hipLaunchKernelGGL(probe, dim3(4), dim3(1024), 0, 0, d);
  unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 2; ++b) for (int w = 0; w < 16; ++w) {
    unsigned id = h[b * 16 + w];
    printf("block %d wave %2d: wave_id %u simd %u cu %u sh %u se %u\n", b, w, id & 15, (id >> 4) & 3, (id >> 8) & 15, (id >> 12) & 1, (id >> 13) & 7);
  }
  return 0;
}
