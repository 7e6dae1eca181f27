// Achievable HBM streaming rates on this box: read-only sum, write-only fill, copy and a
// "sum of K slabs" pattern (the dd reduce), for working sets inside and outside the 256 MB
// infinity cache.  hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_stream.hip -o tools/probe/hbm_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_sum(const float4* __restrict__ p, size_t n4, float* out) {
  float4 s = make_float4(0, 0, 0, 0);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; i < n4; i += stride) { float4 v = p[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void write_fill(float4* __restrict__ p, size_t n4, float v) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) b[i + u * stride] = v[u];
  }
  for (; i < n4; i += stride) b[i] = a[i];
}

// ---- "sum of K slabs": out[i] = sum_z part[z][i] (the dd reduce of the fused decoder kernel) ----
template <int INFLIGHT, bool STAGGER>
__global__ __launch_bounds__(256) void slab_sum(const float4* __restrict__ p, int slabs, size_t n4,
                                                float4* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 s = make_float4(0, 0, 0, 0);
    // STAGGER: every workgroup starts at another slab, so that the chip does not sweep the
    // slabs in lock step (the sum order per element then depends on the workgroup: probe only)
    const int z0 = STAGGER ? (int)((blockIdx.x * 37u) % (unsigned)slabs) : 0;
    for (int zz = 0; zz < slabs; zz += INFLIGHT) {
      float4 v[INFLIGHT];
#pragma unroll
      for (int u = 0; u < INFLIGHT; ++u) {
        int z = z0 + zz + u; if (z >= slabs) z -= slabs;
        v[u] = p[(size_t)z * n4 + i];
      }
#pragma unroll
      for (int u = 0; u < INFLIGHT; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    out[i] = s;
  }
}
// two columns per thread: 2 x INFLIGHT loads in flight without a longer dependent chain
template <int INFLIGHT>
__global__ __launch_bounds__(256) void slab_sum2(const float4* __restrict__ p, int slabs, size_t n4,
                                                 float4* __restrict__ out) {
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t half = n4 / 2;
  if (i0 >= half) return;
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
  for (int zz = 0; zz < slabs; zz += INFLIGHT) {
    float4 v[INFLIGHT], w[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) { v[u] = p[(size_t)(zz + u) * n4 + i0]; w[u] = p[(size_t)(zz + u) * n4 + half + i0]; }
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      s0.x += v[u].x; s0.y += v[u].y; s0.z += v[u].z; s0.w += v[u].w;
      s1.x += w[u].x; s1.y += w[u].y; s1.z += w[u].z; s1.w += w[u].w;
    }
  }
  out[i0] = s0; out[half + i0] = s1;
}

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t sizes_mb[] = {64, 192, 419, 1024, 2048};
  float* out; CHECK(hipMalloc(&out, 4));
  float4 *a, *b;
  CHECK(hipMalloc(&a, (size_t)2048 << 20));
  CHECK(hipMalloc(&b, (size_t)2048 << 20));
  CHECK(hipMemset(a, 0, (size_t)2048 << 20));
  CHECK(hipMemset(b, 0, (size_t)2048 << 20));
  printf("%8s %12s %12s %12s   (TB/s; copy counts read + write)\n", "MB", "read", "write", "copy");
  for (size_t mb : sizes_mb) {
    const size_t n4 = (mb << 20) / 16;
    for (int blocks : {2048, 8192}) {
      const float r = time_ms([&] { hipLaunchKernelGGL(read_sum, dim3(blocks), dim3(256), 0, 0, a, n4, out); }, 10);
      const float w = time_ms([&] { hipLaunchKernelGGL(write_fill, dim3(blocks), dim3(256), 0, 0, a, n4, 1.f); }, 10);
      const float c = time_ms([&] { hipLaunchKernelGGL(copy4, dim3(blocks), dim3(256), 0, 0, a, b, n4); }, 10);
      const double bytes = (double)(mb << 20);
      printf("%8zu %12.2f %12.2f %12.2f   blocks=%d\n", mb, bytes / r / 1e9, bytes / w / 1e9, 2 * bytes / c / 1e9, blocks);
    }
  }
  // write 419 MB with one kernel, then read it with another (the decoder -> dd_reduce hand-off)
  {
    const size_t n4 = ((size_t)419 << 20) / 16;
    const float t = time_ms([&] {
      hipLaunchKernelGGL(write_fill, dim3(8192), dim3(256), 0, 0, a, n4, 2.f);
      hipLaunchKernelGGL(read_sum, dim3(8192), dim3(256), 0, 0, a, n4, out);
    }, 10);
    printf("write 419 MB then read it back: %.1f us for both (%.2f TB/s over 838 MB)\n", t * 1e3,
           2.0 * ((size_t)419 << 20) / t / 1e9);
  }
  {
    const int slabs = 512; const size_t n4 = 4096 * 100 / 4;   // 512 x [4096, 100] floats = 839 MB
    const double bytes = (double)slabs * n4 * 16;
    auto report = [&](const char* name, float ms) { printf("%-44s %7.1f us  %.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9); };
    const int blocks = (int)((n4 + 255) / 256);
    report("slab_sum  8 in flight, 400 blocks", time_ms([&] { hipLaunchKernelGGL((slab_sum<8, false>), dim3(blocks), dim3(256), 0, 0, a, slabs, n4, b); }, 10));
    report("slab_sum 16 in flight", time_ms([&] { hipLaunchKernelGGL((slab_sum<16, false>), dim3(blocks), dim3(256), 0, 0, a, slabs, n4, b); }, 10));
    report("slab_sum 32 in flight", time_ms([&] { hipLaunchKernelGGL((slab_sum<32, false>), dim3(blocks), dim3(256), 0, 0, a, slabs, n4, b); }, 10));
    report("slab_sum  8 in flight, staggered start", time_ms([&] { hipLaunchKernelGGL((slab_sum<8, true>), dim3(blocks), dim3(256), 0, 0, a, slabs, n4, b); }, 10));
    report("slab_sum 16 in flight, staggered start", time_ms([&] { hipLaunchKernelGGL((slab_sum<16, true>), dim3(blocks), dim3(256), 0, 0, a, slabs, n4, b); }, 10));
    report("slab_sum2 2 x 8 in flight, 200 blocks", time_ms([&] { hipLaunchKernelGGL((slab_sum2<8>), dim3(blocks / 2), dim3(256), 0, 0, a, slabs, n4, b); }, 10));
  }
  return 0;
}
