// The hidden layers of a LARGE training minibatch with fewer launches (round 3).
//
// dense_layer with batch normalisation (mu:38-76: fully_connected -> batch_norm(center=True,
// scale=False) -> relu) is, launch by launch, GEMM | chunk statistics | merge | normalise going
// forward and statistics | merge | dA | dX GEMM | dW GEMM | split-K reduce going back: ten
// launches per layer of ~5 us each on [4096, 100] tensors, none of them bound by anything but its
// own dependent memory round trips.  The statistics of a layer need all rows, so the layer cannot
// be one kernel without a grid barrier -- but the work can be cut at the OTHER side of the
// statistics: a workgroup owns a 64-row tile through
//
//   forward  (tile_fwd_kernel):  merge the chunk statistics of the layer below (every workgroup
//            the same fixed-order merge: identical bits), normalise + relu its own tile of that
//            layer (written out once, as h, for the backward pass), product with this layer's
//            weights on the fp32 matrix cores, bias, the tile's own chunk statistics;
//   backward (tile_bwd_kernel):  merge the layer's two column sums, dA of the tile
//            (bn_input_gradient), dX = dA W^T, this tile's slab of dW = in^T dA, and the chunk
//            sums the layer BELOW needs for its own batch-norm backward.
//
// One launch per layer and direction (+ one fixed-order reduce of the dW slabs), kernel
// boundaries where the statistics need all rows, no atomics, no grid barrier.  The posterior heads
// (two weight matrices on the same input) ride in the same kernels.  Used by the VAE plan for
// training minibatches of more than 128 rows without dropout; everything else keeps the launch
// chain (plan.hip) or the two mid-chain kernels (midchain.hip).  Under a data-parallel hook
// (scvae_plan_set_sync) the statistics of a layer are those of the global minibatch: the rank's
// chunks are merged by a one-workgroup kernel, the hook merges the ranks, and the consuming tile
// kernel takes the result as given (TileBN::part == nullptr) -- two small launches and a
// collective per layer boundary instead of the launch chain's four and a collective.
#include "common.hpp"
#include "kernels.hpp"

namespace scvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TC_ROWS = 64;      // rows per tile
constexpr int TC_MAXN = 128;     // widest layer
constexpr int TC_LD = 129;       // LDS row stride (odd)
constexpr int TC_THREADS = 512;   // 8 waves, two per SIMD: at one per SIMD every LDS / memory latency is exposed
constexpr int TC_RG = TC_THREADS / 128;   // row groups: thread = (column tid & 127, row group tid >> 7)

size_t tile_chain_part_floats(int rows) {
  return (size_t)((rows + TC_ROWS - 1) / TC_ROWS) * 2 * TC_MAXN;
}
size_t tile_chain_slab_floats(int rows) {
  return (size_t)((rows + TC_ROWS - 1) / TC_ROWS) * (TC_MAXN + 1) * TC_MAXN;
}

// C[64, N] (+)= A[64, K] B[K, N] on the fp32 matrix cores; A in LDS [64][TC_LD], B in LDS
// [K][TC_LD] (k padded to even with zeros by the caller); wave w (of 8): row tile w & 1, column
// tile w >> 1.
__device__ __forceinline__ void tile_mma(const float* As, const float* Bs, int K2, int N, int w,
                                         int lane, f32x16& acc) {
  const int li = lane & 31, kh = lane >> 5;
  const int rt = w & 1, ct = w >> 1;
  if (32 * ct >= N) return;
  const float* ap = As + (32 * rt + li) * TC_LD + kh;
  const float* bp = Bs + kh * TC_LD + 32 * ct + li;
#pragma unroll 8
  for (int k = 0; k < K2; k += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k * TC_LD], acc, 0, 0, 0);
}
// the accumulator of tile_mma -> LDS tile [64][TC_LD]
__device__ __forceinline__ void tile_store(float* Os, int N, int w, int lane, const f32x16& acc) {
  const int li = lane & 31, kh = lane >> 5;
  const int rt = w & 1, ct = w >> 1;
  if (32 * ct >= N) return;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    Os[(32 * rt + (i & 3) + 8 * (i >> 2) + 4 * kh) * TC_LD + 32 * ct + li] = acc[i];
}

typedef float f32x4t __attribute__((ext_vector_type(4)));

// A contiguous block of n floats (a [rows, nc] tile whose pitch is nc: every tensor of the
// chain) on its way into an LDS tile [..][TC_LD]: `load` requests it with 16-byte loads, all of a
// thread's requests in flight together; `store` lands it later (element e -> row e / nc, column
// e % nc), so that several blocks travel at once and a kernel pays one memory round trip for its
// inputs instead of one per block.
template <int NV>
struct TilePre {
  float v[NV];
  // request rows rl, rl + TC_RG, ... (NV of them) of column c = tid & 127 of a [nr, nc] block of
  // pitch nc (n = nr * nc floats)
  __device__ __forceinline__ void load(const float* __restrict__ src, int n, int tid, int nc) {
    const int c = tid & 127, rl = tid >> 7;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = (rl + TC_RG * u) * nc + c;
      v[u] = (c < nc && e < n) ? src[e] : 0.f;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ dst, int cols_pad, int rows_pad,
                                        int tid) const {
    const int c = tid & 127, rl = tid >> 7;
    if (c >= cols_pad) return;
#pragma unroll
    for (int u = 0; u < NV; ++u)
      if (rl + TC_RG * u < rows_pad) dst[(rl + TC_RG * u) * TC_LD + c] = v[u];
  }
};
// a linear block of n floats (the chunk statistics): 16-byte loads
template <int NV>
struct LinePre {
  f32x4t v[NV];
  __device__ __forceinline__ void load(const float* __restrict__ src, int n, int tid) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = 4 * (tid + TC_THREADS * u);
      f32x4t x = {0.f, 0.f, 0.f, 0.f};
      if (e + 3 < n) {
        x = *reinterpret_cast<const f32x4t*>(src + e);
      } else if (e < n) {
        x.x = src[e];
        if (e + 1 < n) x.y = src[e + 1];
        if (e + 2 < n) x.z = src[e + 2];
      }
      v[u] = x;
    }
  }
  __device__ __forceinline__ void store_linear(float* __restrict__ dst, int n, int tid) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = 4 * (tid + TC_THREADS * u);
      if (e + 3 < n) {
        dst[e] = v[u].x; dst[e + 1] = v[u].y; dst[e + 2] = v[u].z; dst[e + 3] = v[u].w;
      } else if (e < n) {
        dst[e] = v[u].x;
        if (e + 1 < n) dst[e + 1] = v[u].y;
        if (e + 2 < n) dst[e + 2] = v[u].z;
      }
    }
  }
};
// What crosses workgroups INSIDE a launch of the resident chain (below) -- the chunk statistics
// and chunk sums of the batch norms, a few hundred bytes per tile -- is written and read with
// agent-scope accesses (global_store / global_load ... sc1: performed at the memory side, past
// the XCD's own L2), so that the grid barrier between two stages needs no L2 write-back and no
// invalidate.  The per-layer launches use the same accessors (same code, same bits).
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef SCVAE_TC_PROBE
// phase probe of the resident kernels (workgroup 1, thread 0): s_memtime ticks per (stage, phase),
// summed over the launches; slot 5 * stage + {0 run, 1 stores done, 2 issue, 3 wait, 4 partials},
// slot 63: launches
__device__ unsigned long long g_tc_probe[2][64];
extern "C" int scvae_debug_tc_probe(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tc_probe), sizeof(g_tc_probe));
}
#define TCP_RUN(slot) do { if (threadIdx.x == 0 && blockIdx.x == 1) { \
  const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  atomicAdd(&g_tc_probe[0][slot], t_ - trun); trun = t_; } } while (0)
#define TCP_RUN_BEGIN unsigned long long trun = __builtin_amdgcn_s_memtime()
#define TCP_BEGIN(dir) unsigned long long tlast = __builtin_amdgcn_s_memtime(); const int tdir = dir; \
  if (threadIdx.x == 0 && blockIdx.x == 1) atomicAdd(&g_tc_probe[tdir][63], 1ull)
#define TCP(slot) do { if (threadIdx.x == 0 && blockIdx.x == 1) { \
  const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  atomicAdd(&g_tc_probe[tdir][slot], t_ - tlast); tlast = t_; } } while (0)
#else
#define TCP_RUN(slot) do {} while (0)
#define TCP_RUN_BEGIN do {} while (0)
#define TCP_BEGIN(dir) do {} while (0)
#define TCP(slot) do {} while (0)
#endif
constexpr int TC_ZPT = 64 / TC_RG;    // chunks per thread in a block of 64 chunks

// One tile (workgroup g) of one forward stage, in three steps so that the resident chain can put
// its grid barrier between them: tile_fwd_issue requests what does not depend on other tiles (the
// first weight matrix, the tile's own input rows), tile_fwd_partials requests the first 64 chunk
// statistics of the layer below (every tile's: after the barrier), tile_fwd_run does the work.
// Thread (c = tid & 127, rl = tid >> 7) owns column c of rows rl, rl + 4, ... of the tile -- of
// the input tile (TilePre), through the normalisation, and of the output tile in the epilogue:
// normalisation, bias, the tile's statistics run on registers.
struct TileFwdRegs {
  TilePre<128 / TC_RG> pw;              // [K, N] weights (<= 128 x 128)
  TilePre<TC_ROWS / TC_RG> pa;          // [64, K] input tile
};
struct TileFwdPart {
  float pm[TC_ZPT], pv[TC_ZPT];         // chunk (mean, M2) of chunks rl, rl + 4, ... of a block
};
__device__ __forceinline__ void tile_fwd_issue(const TileFwdArgs& q, TileFwdRegs& R, const int g) {
  const int tid = threadIdx.x;
  const int r0 = g * TC_ROWS;
  const int nr = min(TC_ROWS, q.rows - r0);
  const int K = q.K;
  const int n_w0 = q.n_out > 0 ? K * q.o[0].N : 0;
  R.pw.load(q.n_out > 0 ? q.o[0].W : q.x, n_w0, tid, q.n_out > 0 ? q.o[0].N : 1);
  const float* src_a = q.bn.a ? q.bn.a + (size_t)r0 * K : q.x + (size_t)r0 * q.ldx;
  R.pa.load(src_a, nr * K, tid, K);
}
// chunks z0 + rl, z0 + rl + 4, ... (those below zn) of column c
__device__ __forceinline__ void tile_part_load(const float* __restrict__ part, int K, int zbase,
                                               int zn, float* pm, float* pv) {
  const int c = threadIdx.x & 127, rl = threadIdx.x >> 7;
#pragma unroll
  for (int j = 0; j < TC_ZPT; ++j) {
    const int z = rl + TC_RG * j;
    const bool on = z < zn && c < K;
    const float* pz = part + (size_t)(zbase + (on ? z : 0)) * 2 * K + (on ? c : 0);
    pm[j] = on ? ld_agent(pz) : 0.f;
    pv[j] = on ? ld_agent(pz + K) : 0.f;
  }
}
__device__ __forceinline__ void tile_fwd_partials(const TileFwdArgs& q, TileFwdPart& P, const int g) {
  if (!(q.bn.a && q.bn.part)) return;
  const int gt = q.bn.group_tiles;
  const int zfirst = gt ? (g / gt) * gt : 0;
  tile_part_load(q.bn.part, q.K, zfirst, min(64, gt ? gt : q.bn.chunks), P.pm, P.pv);
}
__device__ __forceinline__ void tile_fwd_run(const TileFwdArgs& q, TileFwdRegs& R, TileFwdPart& P,
                                             float* tsm, const int g) {
  float* As = tsm;                               // [64][TC_LD]
  float* Bs = As + TC_ROWS * TC_LD;              // [K2][TC_LD], then the output tile
  float* X = Bs + TC_MAXN * TC_LD;               // [TC_RG][128]: the row groups' partial sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = tid & 127, rl = tid >> 7;
  const int r0 = g * TC_ROWS;
  const int nr = min(TC_ROWS, q.rows - r0);
  const int K = q.K, K2 = (K + 1) & ~1;
  // the second weight matrix (the posterior heads) travels under the first product
  TCP_RUN_BEGIN;
  TilePre<128 / TC_RG> pw2;
  if (q.n_out > 1) pw2.load(q.o[1].W, K * q.o[1].N, tid, q.o[1].N);
  float bias[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) bias[o] = (o < q.n_out && c < q.o[o].N) ? q.o[o].b[c] : 0.f;
  // ---- the batch statistics of the layer below, column c ----
  float mean_c = 0.f, istd_c = 0.f, beta_c = 0.f;
  if (q.bn.a && !q.bn.part) {
    // given (merged over this rank's chunks by tile_stats_merge and over the ranks by the
    // caller's hook: data-parallel steps, scvae_plan_set_sync)
    if (c < K) {
      const int grp = q.bn.group_tiles ? g / q.bn.group_tiles : 0;    // (its group's: [groups][K])
      mean_c = q.bn.mean[grp * K + c];
      istd_c = rsqrtf(q.bn.var[grp * K + c] + BN_EPSILON);
      beta_c = q.bn.beta[c];
    }
  } else if (q.bn.a) {
    // merged here, by every workgroup alike: two passes -- mean, then M2 about it -- over the
    // chunks in a fixed order (thread (c, rl): chunks rl, rl + 4, ... of each block of 64, the
    // four row groups' sums added in group order), so every workgroup arrives at the same bits.
    // (groups: the statistics of this tile's group alone -- its group_tiles chunks of 64 rows)
    const int gt = q.bn.group_tiles;
    const int grp = gt ? g / gt : 0, zfirst = grp * gt;
    const int chunks = gt ? gt : q.bn.chunks, chunk = q.bn.chunk;
    const int rows_g = gt ? gt * TC_ROWS : q.rows;
    const float n_full = (float)chunk;
    const float n_last = (float)(rows_g - (chunks - 1) * chunk);
    if (c < K) beta_c = q.bn.beta[c];
    float mean = 0.f, m2 = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
      for (int z0 = 0; z0 < chunks; z0 += 64) {
        const int zn = min(64, chunks - z0);
        // (block 0 arrived with P; a single block stays in registers for the second pass)
        if (chunks > 64 && !(pass == 0 && z0 == 0))
          tile_part_load(q.bn.part, K, zfirst + z0, zn, P.pm, P.pv);
        float part_sum = 0.f;
#pragma unroll
        for (int j = 0; j < TC_ZPT; ++j) {
          const int z = rl + TC_RG * j;
          if (z < zn) {
            const float n = z0 + z == chunks - 1 ? n_last : n_full;
            part_sum = pass == 0 ? bn_merge_mean(part_sum, n, P.pm[j])
                                 : bn_merge_m2(part_sum, n, P.pm[j], P.pv[j], mean);
          }
        }
        lds_barrier();
        X[rl * TC_MAXN + c] = part_sum;
        lds_barrier();
        float total = X[c];
#pragma unroll
        for (int j = 1; j < TC_RG; ++j) total += X[j * TC_MAXN + c];
        if (pass == 0) mean += total; else m2 += total;
      }
      if (pass == 0) mean /= (float)rows_g;
    }
    const float var = m2 / (float)rows_g;
    mean_c = mean;
    istd_c = rsqrtf(var + BN_EPSILON);
    if (g == zfirst && rl == 0 && c < K) { q.bn.mean[grp * K + c] = mean; q.bn.var[grp * K + c] = var; }
  }
  TCP_RUN(40);    // statistics merged
  // ---- the input tile: normalise + relu in registers, h written out once ----
  if (q.bn.a && c < K) {
#pragma unroll
    for (int u = 0; u < TC_ROWS / TC_RG; ++u) {
      const int r = rl + TC_RG * u;
      if (r < nr) {
        const float v = fmaxf(bn_normalise(R.pa.v[u], mean_c, istd_c, beta_c), 0.f);
        R.pa.v[u] = v;
        q.bn.h[(size_t)(r0 + r) * K + c] = v;
      }
    }
  }
  if (q.n_out > 0) {
    lds_barrier();        // (As / Bs free: an earlier stage of a resident launch is done with them)
    R.pa.store(As, (K + 31) & ~31, TC_ROWS, tid);
  }
  TCP_RUN(41);    // normalised, h stored, tile -> LDS
  // ---- products ----
  for (int o = 0; o < q.n_out; ++o) {
    const TileFwdArgs::Out& out = q.o[o];
    const int N = out.N;
    if (o > 0) lds_barrier();     // (the previous output tile consumed)
    if (o == 0) R.pw.store(Bs, (N + 31) & ~31, K2, tid);
    else pw2.store(Bs, (N + 31) & ~31, K2, tid);
    lds_barrier();
    TCP_RUN(42);    // weights -> LDS
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    tile_mma(As, Bs, K2, N, w, lane, acc);
    lds_barrier();     // (all waves done with Bs)
    TCP_RUN(43);    // product
    tile_store(Bs, N, w, lane, acc);
    lds_barrier();
    TCP_RUN(44);    // accumulators -> LDS
    // bias, store, chunk statistics of the tile (two-pass, as bn_stats_partial_kernel) on the
    // thread's sixteen values of column c
    const float bv = o == 0 ? bias[0] : bias[1];
    float v[TC_ROWS / TC_RG];
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < TC_ROWS / TC_RG; ++u) {
      const int r = rl + TC_RG * u;
      const bool live = c < N && r < nr;
      v[u] = live ? Bs[r * TC_LD + c] + bv : 0.f;
      if (live) out.out[(size_t)(r0 + r) * N + c] = v[u];
      sum += v[u];
    }
    TCP_RUN(45);    // bias + output stored
    if (out.part) {
      lds_barrier();
      X[rl * TC_MAXN + c] = sum;
      lds_barrier();
      float tot = X[c];
#pragma unroll
      for (int j = 1; j < TC_RG; ++j) tot += X[j * TC_MAXN + c];
      const float mu = tot / (float)nr;
      float m2 = 0.f;
#pragma unroll
      for (int u = 0; u < TC_ROWS / TC_RG; ++u) {
        const float d = v[u] - mu;
        m2 = fmaf(d, (rl + TC_RG * u < nr) ? d : 0.f, m2);
      }
      lds_barrier();
      X[rl * TC_MAXN + c] = m2;
      lds_barrier();
      if (rl == 0 && c < N) {
        float t2 = X[c];
#pragma unroll
        for (int j = 1; j < TC_RG; ++j) t2 += X[j * TC_MAXN + c];
        st_agent(out.part + ((size_t)g * 2) * N + c, mu);
        st_agent(out.part + ((size_t)g * 2 + 1) * N + c, t2);
      }
      TCP_RUN(46);  // tile statistics
    }
  }
}
__global__ __launch_bounds__(TC_THREADS) void tile_fwd_kernel(TileFwdArgs q) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  TileFwdRegs R;
  TileFwdPart P;
  // (the chunk statistics of the layer below first: they are needed first, and a wave's loads
  //  return in order)
  tile_fwd_partials(q, P, blockIdx.x);
  tile_fwd_issue(q, R, blockIdx.x);
  tile_fwd_run(q, R, P, tsm, blockIdx.x);
}

static constexpr size_t TC_FWD_LDS =
    (size_t)(TC_ROWS * TC_LD + TC_MAXN * TC_LD + (1 + TC_RG) * TC_MAXN) * 4;

int tile_forward(hipStream_t s, const TileFwdArgs& q) {
  SCVAE_ARG(q.rows > 0 && q.K > 0 && q.K <= TC_MAXN && q.n_out >= 0 && q.n_out <= 2);
  SCVAE_ARG(q.bn.a || (q.x && q.ldx == q.K));      // (tiles are contiguous blocks)
  for (int o = 0; o < q.n_out; ++o) SCVAE_ARG(q.o[o].N > 0 && q.o[o].N <= TC_MAXN);
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(tile_fwd_kernel), (int)TC_FWD_LDS));
  hipLaunchKernelGGL(tile_fwd_kernel, dim3((q.rows + TC_ROWS - 1) / TC_ROWS), dim3(TC_THREADS),
                     TC_FWD_LDS, s, q);
  SCVAE_LAUNCH_CHECK("tile_fwd_kernel");
  return 0;
}

// ------------------------------- backward ---------------------------------------------------
__device__ __forceinline__ void tile_bwd_body(const TileBwdArgs& q, float* tsm, const int g) {
  float* As = tsm;                               // h tile, then the input tile, then d_in [64][TC_LD]
  float* Ds = As + TC_ROWS * TC_LD;              // dA tiles [n_up][64][TC_LD]
  float* Ws = Ds + 2 * TC_ROWS * TC_LD;          // a tile, then half of W: [64][TC_LD]
  float* st = Ws + TC_ROWS * TC_LD;              // [2 TC_RG][128]: mean, istd, s1, s2 / partial sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int r0 = g * TC_ROWS;
  const int nr = min(TC_ROWS, q.rows - r0);
  const int K = q.K;
  const bool bn = q.bn.a != nullptr;
  // ---- every input tile is requested up front (the chunk sums of this layer's batch norm
  //      first: they are needed first and loads return in order) ----
  float pm[TC_ZPT], pv[TC_ZPT];     // chunk (sum dxh, sum dxh xh) of chunks rl, rl + 4, ... of a block
  if (bn && q.bn.part) {
    const int gt0 = q.bn.group_tiles;
    const int zfirst0 = gt0 ? (g / gt0) * gt0 : 0;
    tile_part_load(q.bn.part, q.up[0].N, zfirst0, min(64, gt0 ? gt0 : q.bn.chunks), pm, pv);
  }
  TilePre<TC_ROWS / TC_RG> pg[2], ph, pa, pin, pw0;
#pragma unroll
  for (int u = 0; u < 2; ++u)
    pg[u].load(u < q.n_up ? q.up[u].g + (size_t)r0 * q.up[u].N : q.up[0].g,
               u < q.n_up ? nr * q.up[u].N : 0, tid, u < q.n_up ? q.up[u].N : 1);
  {
    const int N0 = q.up[0].N;
    ph.load(bn ? q.bn.h + (size_t)r0 * N0 : q.up[0].g, bn ? nr * N0 : 0, tid, N0);
    pa.load(bn ? q.bn.a + (size_t)r0 * N0 : q.up[0].g, bn ? nr * N0 : 0, tid, N0);
    pin.load(q.in ? q.in + (size_t)r0 * K : q.up[0].g, q.in ? nr * K : 0, tid, q.in ? K : 1);
    pw0.load(q.up[0].W, q.d_in ? min(64, K) * N0 : 0, tid, N0);
  }
  TilePre<TC_ROWS / TC_RG> pb;   // the pre-normalisation tile of the layer below (its chunk sums)
  pb.load(q.below.a ? q.below.a + (size_t)r0 * K : q.up[0].g, q.below.a ? nr * K : 0, tid,
          q.below.a ? K : 1);
  // ---- this layer's batch norm: merged sums (fixed order), dbeta, moving averages ----
  // (groups: see TileBN -- this tile's group, its first chunk, the chunks merged here)
  const int gt_bn = q.bn.group_tiles;
  const int grp = gt_bn ? g / gt_bn : (q.below.group_tiles ? g / q.below.group_tiles : 0);
  if (bn && !q.bn.part) {
    // the sums are given (tile_sums_merge + the caller's all-reduce, as in the forward kernel;
    // dbeta and the moving averages were written there)
    const int N = q.up[0].N;
    if (tid < N) {
      st[tid] = q.bn.mean[grp * N + tid];
      st[TC_MAXN + tid] = rsqrtf(q.bn.var[grp * N + tid] + BN_EPSILON);
      st[2 * TC_MAXN + tid] = q.bn.s1[grp * N + tid];
      st[3 * TC_MAXN + tid] = q.bn.s2[grp * N + tid];
    }
    lds_barrier();
  } else if (bn) {
    const int N = q.up[0].N;
    const int zfirst = gt_bn ? grp * gt_bn : 0;
    const int nchunks = gt_bn ? gt_bn : q.bn.chunks;
    float t1 = 0.f, t2 = 0.f;
    for (int z0 = 0; z0 < nchunks; z0 += 64) {     // (from registers, as in the forward kernel)
      const int zn = min(64, nchunks - z0);
      if (z0 > 0) tile_part_load(q.bn.part, N, zfirst + z0, zn, pm, pv);
      {
        // thread (column c, group zg): chunks z = zg mod TC_RG; groups combined in a fixed order
        const int c = tid & 127, zg = tid >> 7;
        float p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int j = 0; j < TC_ZPT; ++j) {
          const bool on = zg + TC_RG * j < zn && c < N;
          p1 += on ? pm[j] : 0.f;
          p2 += on ? pv[j] : 0.f;
        }
        lds_barrier();
        st[zg * TC_MAXN + c] = p1;
        st[(TC_RG + zg) * TC_MAXN + c] = p2;
        lds_barrier();
        if (tid < N) {
#pragma unroll
          for (int j = 0; j < TC_RG; ++j) {
            t1 += st[j * TC_MAXN + tid];
            t2 += st[(TC_RG + j) * TC_MAXN + tid];
          }
        }
      }
    }
    lds_barrier();
    if (tid < N) {
      const float mean = q.bn.mean[grp * N + tid], var = q.bn.var[grp * N + tid];
      st[tid] = mean;
      st[TC_MAXN + tid] = rsqrtf(var + BN_EPSILON);
      st[2 * TC_MAXN + tid] = t1;
      st[3 * TC_MAXN + tid] = t2;
      if (g == zfirst) {
        q.bn.s1[grp * N + tid] = t1;
        q.bn.s2[grp * N + tid] = t2;
      }
      if (g == 0 && !gt_bn) {
        q.bn.dbeta[tid] = t1;
        // UPDATE_OPS (va:2763-2768): moving <- moving - (moving - batch) * rate, Bessel-corrected
        q.bn.mov_mean[tid] = bn_moving_update(q.bn.mov_mean[tid], mean);
        q.bn.mov_var[tid] = bn_moving_update(q.bn.mov_var[tid], var * q.bessel);
      }
    }
    lds_barrier();
    if (g == 0 && gt_bn) {
      // the layer ran once per group on shared variables: dbeta is the sum over the groups of
      // their s1 (all the chunks, in chunk order), the moving averages are updated group after
      // group (gm:2859-2922: K executions of the UPDATE_OPS in pass order)
      float tall = 0.f;
      const int c = tid & 127, zg = tid >> 7;
      for (int z0 = 0; z0 < q.bn.chunks; z0 += 64) {
        const int zn = min(64, q.bn.chunks - z0);
        LinePre<8> pp;
        pp.load(q.bn.part + (size_t)z0 * 2 * N, zn * 2 * N, tid);
        lds_barrier();
        pp.store_linear(Ds, zn * 2 * N, tid);
        lds_barrier();
        float p1 = 0.f;
        if (c < N) {
#pragma unroll 4
          for (int z = zg; z < 64; z += TC_RG) p1 += z < zn ? Ds[z * 2 * N + c] : 0.f;
        }
        lds_barrier();
        st[(4 + zg) * TC_MAXN + c] = p1;
        lds_barrier();
        if (tid < N) {
#pragma unroll
          for (int j = 0; j < TC_RG; ++j) tall += st[(4 + j) * TC_MAXN + tid];
        }
      }
      if (tid < N) {
        q.bn.dbeta[tid] = tall;
        float mm = q.bn.mov_mean[tid], mv = q.bn.mov_var[tid];
        for (int k = 0; k < q.bn.groups; ++k) {
          mm = bn_moving_update(mm, q.bn.mean[k * N + tid]);
          mv = bn_moving_update(mv, q.bn.var[k * N + tid] * q.bessel);
        }
        q.bn.mov_mean[tid] = mm;
        q.bn.mov_var[tid] = mv;
      }
      lds_barrier();
    }
  }
  // ---- dA tiles: formed on registers (thread (c, rl) holds rows rl, rl + 4, ... of column c of
  //      the gradient, the normalised output and the pre-normalisation tile alike), parked once ----
  {
    const int c = tid & 127, rl = tid >> 7;
    if (bn) {
      const int N = q.up[0].N;
      if (c < N) {        // (st: behind the barrier that closes the merge above)
        const float mean_c = st[c], istd_c = st[TC_MAXN + c];
        const float s1_c = st[2 * TC_MAXN + c], s2_c = st[3 * TC_MAXN + c];
#pragma unroll
        for (int u = 0; u < TC_ROWS / TC_RG; ++u) {
          if (rl + TC_RG * u < nr) {
            float v = pg[0].v[u];
            if (!(ph.v[u] > 0.f)) v = 0.f;
            const float xh = (pa.v[u] - mean_c) * istd_c;
            pg[0].v[u] = bn_input_gradient(v, xh, s1_c, s2_c, q.inv_count, istd_c);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u < q.n_up) {
        const int N = q.up[u].N;
        if (q.up[u].dA_out && c < N) {
#pragma unroll
          for (int uu = 0; uu < TC_ROWS / TC_RG; ++uu)
            if (rl + TC_RG * uu < nr)
              q.up[u].dA_out[(size_t)(r0 + rl + TC_RG * uu) * N + c] = pg[u].v[uu];
        }
        pg[u].store(Ds + u * TC_ROWS * TC_LD, (N + 31) & ~31, TC_ROWS, tid);
      }
    }
  }
  if (!q.in) return;      // (uniform: the layer that sees x stops here)
  pin.store(As, (K + 31) & ~31, TC_ROWS, tid);
  lds_barrier();
  // ---- dW slabs: dW_u[k, n] (this tile) = sum_row in[row, k] dA_u[row, n]; bias: column sums ----
  for (int u = 0; u < q.n_up; ++u) {
    const TileBwdArgs::Up& up = q.up[u];
    const int N = up.N;
    const float* D = Ds + u * TC_ROWS * TC_LD;
    float* slab = up.dW_slab + (size_t)g * K * N;
    const int kts = (K + 31) / 32, nts = (N + 31) / 32;
    for (int t = w; t < kts * nts; t += TC_THREADS / 64) {
      const int kt = t / nts, nt = t % nts;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = As + kh * TC_LD + 32 * kt + li;     // A[i = k][kk = row]
      const float* bp = D + kh * TC_LD + 32 * nt + li;      // B[kk = row][n]
#pragma unroll 8
      for (int r = 0; r < TC_ROWS; r += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[r * TC_LD], bp[r * TC_LD], acc, 0, 0, 0);
      const int n = 32 * nt + li;
      if (n < N) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * kh;
          if (k < K) slab[(size_t)k * N + n] = acc[i];
        }
      }
    }
    if (up.db_slab && tid < N) {
      float sum = 0.f;
#pragma unroll 8
      for (int r = 0; r < TC_ROWS; ++r) sum += D[r * TC_LD + tid];   // (rows >= nr hold zeros)
      up.db_slab[(size_t)g * N + tid] = sum;
    }
  }
  if (!q.d_in) return;
  // ---- d_in[row, k] = sum_u sum_n dA_u[row, n] W_u[k, n]; the weights pass through LDS in
  //      halves of 64 input units (column tiles 0-1, then 2-3) ----
  f32x16 acc;               // wave w: row tile w & 1, column tile w >> 1 (of the input width)
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int u = 0; u < q.n_up; ++u) {
    const TileBwdArgs::Up& up = q.up[u];
    const int N = up.N, N2 = (N + 1) & ~1;
    const float* D = Ds + u * TC_ROWS * TC_LD;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      if (64 * v >= K) continue;
      const int kn = min(64, K - 64 * v);
      if (u > 0 || v > 0) pw0.load(up.W + (size_t)64 * v * N, kn * N, tid, N);
      lds_barrier();      // (Ws free)
      pw0.store(Ws, N2, 64, tid);
      lds_barrier();
      const int rt = w & 1, kt = w >> 1;
      if ((kt >> 1) == v && 32 * kt < K) {
        const float* ap = D + (32 * rt + li) * TC_LD + kh;              // A[i = row][kk = n]
        const float* bp = Ws + (32 * (kt & 1) + li) * TC_LD + kh;       // B[kk = n][j = k] = W[k][n]
#pragma unroll 8
        for (int n = 0; n < N2; n += 2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[n], bp[n], acc, 0, 0, 0);
      }
    }
  }
  lds_barrier();
  float* Xs = Ds;                         // the d_in tile and the layer below's a tile take the
  float* Ab = Ds + TC_ROWS * TC_LD;       // place of the dA tiles; As still holds `in` (= its h)
  tile_store(Xs, K, w, lane, acc);
  if (q.below.a) pb.store(Ab, K, TC_ROWS, tid);
  lds_barrier();
  const int c = tid & 127, rl = tid >> 7;
  if (c < K) {
#pragma unroll 4
    for (int r = rl; r < nr; r += TC_RG) q.d_in[(size_t)(r0 + r) * K + c] = Xs[r * TC_LD + c];
  }
  // ---- the chunk sums the layer below needs for its own batch-norm backward:
  //      dxh = d_in * (h > 0), xh = (a - mean) * istd; two threads per column ----
  if (q.below.a) {
    float a1 = 0.f, a2 = 0.f;
    if (c < K) {
      const int gb = q.below.group_tiles ? g / q.below.group_tiles : 0;
      const float mu = q.below.mean[gb * K + c];
      const float istd = rsqrtf(q.below.var[gb * K + c] + BN_EPSILON);
#pragma unroll 8
      for (int r = rl; r < TC_ROWS; r += TC_RG) {
        float d = Xs[r * TC_LD + c];
        if (!(r < nr && As[r * TC_LD + c] > 0.f)) d = 0.f;
        const float xh = (Ab[r * TC_LD + c] - mu) * istd;
        a1 += d;
        a2 = fmaf(d, r < nr ? xh : 0.f, a2);
      }
    }
    lds_barrier();
    st[rl * TC_MAXN + c] = a1;
    st[(TC_RG + rl) * TC_MAXN + c] = a2;
    lds_barrier();
    if (rl == 0 && c < K) {
      float u1 = 0.f, u2 = 0.f;
#pragma unroll
      for (int j = 0; j < TC_RG; ++j) {
        u1 += st[j * TC_MAXN + c];
        u2 += st[(TC_RG + j) * TC_MAXN + c];
      }
      q.below.part_out[((size_t)g * 2) * K + c] = u1;
      q.below.part_out[((size_t)g * 2 + 1) * K + c] = u2;
    }
  }
}
__global__ __launch_bounds__(TC_THREADS) void tile_bwd_kernel(TileBwdArgs q) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  tile_bwd_body(q, tsm, blockIdx.x);
}

static constexpr size_t TC_BWD_LDS =
    (size_t)(4 * TC_ROWS * TC_LD + 2 * TC_RG * TC_MAXN) * 4;

int tile_backward(hipStream_t s, const TileBwdArgs& q) {
  SCVAE_ARG(q.rows > 0 && q.n_up >= 1 && q.n_up <= 2 && q.K >= 0 && q.K <= TC_MAXN);
  SCVAE_ARG(!q.bn.a || q.n_up == 1);
  for (int u = 0; u < q.n_up; ++u) SCVAE_ARG(q.up[u].N > 0 && q.up[u].N <= TC_MAXN);
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(tile_bwd_kernel), (int)TC_BWD_LDS));
  hipLaunchKernelGGL(tile_bwd_kernel, dim3((q.rows + TC_ROWS - 1) / TC_ROWS), dim3(TC_THREADS),
                     TC_BWD_LDS, s, q);
  SCVAE_LAUNCH_CHECK("tile_bwd_kernel");
  return 0;
}

// chunk sums of the TOP batch-normalised layer of a chain from its output gradient dh (the first
// 256 threads of the workgroup work; red: 512 floats of LDS)
__device__ __forceinline__ void tile_bwd_stats_body(const float* __restrict__ dh, const TileBN& bn,
                                                    int rows, int N, float* red, const int g) {
  const int r0 = g * TC_ROWS;
  const int nr = min(TC_ROWS, rows - r0);
  const int c = threadIdx.x & 127, rl = (threadIdx.x >> 7) & 1;
  float a1 = 0.f, a2 = 0.f;
  if (c < N && threadIdx.x < 256) {
    const int gb = bn.group_tiles ? g / bn.group_tiles : 0;
    const float mu = bn.mean[gb * N + c];
    const float istd = rsqrtf(bn.var[gb * N + c] + BN_EPSILON);
    // (eight rows' loads in flight: a load-use-per-iteration loop pays a round trip per row)
    for (int r = rl; r < TC_ROWS; r += 16) {
      float dv[8], hv[8], av[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = r + 2 * u;
        const size_t e = (size_t)(r0 + (rr < nr ? rr : 0)) * N + c;
        dv[u] = dh[e]; hv[u] = bn.h[e]; av[u] = bn.a[e];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = r + 2 * u;
        float d = dv[u];
        if (!(rr < nr && hv[u] > 0.f)) d = 0.f;
        const float xh = (av[u] - mu) * istd;
        a1 += d;
        a2 = fmaf(d, xh, a2);
      }
    }
  }
  if (threadIdx.x < 256) {
    red[threadIdx.x] = a1;
    red[256 + threadIdx.x] = a2;
  }
  lds_barrier();
  if (threadIdx.x < 128 && c < N) {
    bn.part_out[((size_t)g * 2) * N + c] = red[c] + red[128 + c];
    bn.part_out[((size_t)g * 2 + 1) * N + c] = red[256 + c] + red[256 + 128 + c];
  }
}
__global__ __launch_bounds__(256) void tile_bwd_stats_kernel(const float* __restrict__ dh,
                                                                    TileBN bn, int rows, int N) {
  __shared__ float red[512];
  tile_bwd_stats_body(dh, bn, rows, N, red, blockIdx.x);
}

int tile_backward_stats(hipStream_t s, const float* dh, const TileBN& bn, int rows, int N) {
  SCVAE_ARG(dh && bn.a && bn.h && bn.mean && bn.var && bn.part_out && N > 0 && N <= TC_MAXN);
  hipLaunchKernelGGL(tile_bwd_stats_kernel, dim3((rows + TC_ROWS - 1) / TC_ROWS), dim3(256),
                     0, s, dh, bn, rows, N);
  SCVAE_LAUNCH_CHECK("tile_bwd_stats_kernel");
  return 0;
}

// ---- data-parallel steps (a sync hook between a layer's statistics and their use): this rank's
//      chunk statistics / chunk sums merged by ONE workgroup into the layer's buffers, which the
//      caller's hook then merges over the ranks; the consuming tile kernel takes them as given
//      (TileBN::part == nullptr).  One thread per column, the chunks in order. ----
// (groups: block k merges the chunks of group k alone -- its group_tiles tiles of 64 rows -- into
//  mean[k][N], var[k][N]: the GMVAE's K passes, gm:2859-2922)
__global__ __launch_bounds__(TC_MAXN) void tile_stats_merge_kernel(const float* __restrict__ part,
                                                                   int chunks, int chunk, int rows,
                                                                   int N, float* __restrict__ mean,
                                                                   float* __restrict__ var) {
  const int c = threadIdx.x, k = blockIdx.x;
  if (c >= N) return;
  part += (size_t)k * chunks * 2 * N;
  const float n_last = (float)(rows - (chunks - 1) * chunk);
  float sum = 0.f;
  for (int z = 0; z < chunks; ++z)
    sum = bn_merge_mean(sum, z == chunks - 1 ? n_last : (float)chunk, part[(size_t)z * 2 * N + c]);
  const float mu = sum / (float)rows;
  float m2 = 0.f;
  for (int z = 0; z < chunks; ++z)
    m2 = bn_merge_m2(m2, z == chunks - 1 ? n_last : (float)chunk, part[(size_t)z * 2 * N + c],
                     part[((size_t)z * 2 + 1) * N + c], mu);
  mean[k * N + c] = mu;
  var[k * N + c] = m2 / (float)rows;
}
__global__ __launch_bounds__(TC_MAXN) void tile_sums_merge_kernel(const float* __restrict__ part,
                                                                  int chunks, int N, TileBN bn,
                                                                  float bessel, int groups) {
  const int c = threadIdx.x;
  if (c >= N) return;
  // this rank's sums per group (s1, s2: [groups][N]); dbeta: all the rank's chunks in chunk order
  // (the gradient all-reduce sums the ranks); the moving averages from the batch statistics of
  // the forward pass -- those of the global minibatch -- group after group (K executions of the
  // UPDATE_OPS in pass order)
  float tall = 0.f;
  float mm = bn.mov_mean[c], mv = bn.mov_var[c];
  for (int k = 0; k < groups; ++k) {
    float t1 = 0.f, t2 = 0.f;
    for (int z = k * chunks; z < (k + 1) * chunks; ++z) {
      const float p1 = part[(size_t)z * 2 * N + c];
      t1 += p1;
      t2 += part[((size_t)z * 2 + 1) * N + c];
      tall += p1;
    }
    bn.s1[k * N + c] = t1;
    bn.s2[k * N + c] = t2;
    mm = bn_moving_update(mm, bn.mean[k * N + c]);
    mv = bn_moving_update(mv, bn.var[k * N + c] * bessel);
  }
  bn.dbeta[c] = tall;
  bn.mov_mean[c] = mm;
  bn.mov_var[c] = mv;
}
// chunks: per group; rows: per group
int tile_stats_merge(hipStream_t s, const float* part, int chunks, int chunk, int rows, int N,
                     float* mean, float* var, int groups) {
  SCVAE_ARG(part && chunks > 0 && chunk > 0 && N > 0 && N <= TC_MAXN && mean && var && groups >= 1);
  hipLaunchKernelGGL(tile_stats_merge_kernel, dim3(groups), dim3(TC_MAXN), 0, s, part, chunks,
                     chunk, rows, N, mean, var);
  SCVAE_LAUNCH_CHECK("tile_stats_merge_kernel");
  return 0;
}
int tile_sums_merge(hipStream_t s, const float* part, int chunks, int N, const TileBN& bn,
                    float bessel, int groups) {
  SCVAE_ARG(part && chunks > 0 && N > 0 && N <= TC_MAXN && bn.s1 && bn.s2 && bn.dbeta &&
            bn.mov_mean && bn.mov_var && bn.mean && bn.var && groups >= 1);
  hipLaunchKernelGGL(tile_sums_merge_kernel, dim3(1), dim3(TC_MAXN), 0, s, part, chunks, N, bn,
                     bessel, groups);
  SCVAE_LAUNCH_CHECK("tile_sums_merge_kernel");
  return 0;
}

// one job: out[i] = sum_g slabs[g][i], elements first, first + stride, ...
__device__ __forceinline__ void tile_slab_reduce_job(const SlabJobs::Job& jb, int first,
                                                     int stride) {
  for (int i = first; i < jb.n; i += stride) {
    float s0 = 0.f;
    int g = 0;
    for (; g + 8 <= jb.G; g += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = jb.slabs[(size_t)(g + u) * jb.n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s0 += v[u];
    }
    for (; g < jb.G; ++g) s0 += jb.slabs[(size_t)g * jb.n + i];
    jb.out[i] = s0;
  }
}
__global__ __launch_bounds__(256) void tile_slab_reduce_kernel(SlabJobs q) {
  tile_slab_reduce_job(q.job[blockIdx.y], blockIdx.x * blockDim.x + threadIdx.x,
                       gridDim.x * blockDim.x);
}
int tile_slab_reduce(hipStream_t s, const SlabJobs& q) {
  SCVAE_ARG(q.n_jobs >= 1 && q.n_jobs <= TC_MAX_JOBS);
  int n_max = 0;
  for (int j = 0; j < q.n_jobs; ++j) {
    SCVAE_ARG(q.job[j].G >= 1 && q.job[j].slabs && q.job[j].out);
    n_max = q.job[j].n > n_max ? q.job[j].n : n_max;
  }
  hipLaunchKernelGGL(tile_slab_reduce_kernel, dim3((n_max + 255) / 256, q.n_jobs), dim3(256), 0, s,
                     q);
  SCVAE_LAUNCH_CHECK("tile_slab_reduce_kernel");
  return 0;
}

// ------------------------------- the resident chain -----------------------------------------
// The stages of a whole forward (backward) pass of the hidden layers in ONE launch: the same tile
// bodies as above -- the same arithmetic in the same order, hence the same bits as the chain of
// per-layer launches -- run by workgroups that stay resident, with a grid barrier (common.hpp)
// where the next stage needs every tile's chunk statistics and a workgroup barrier where it only
// needs the tile's own rows (posterior heads -> latent stage -> first decoder layer, one sample
// per cell).  What goes away: nine of ten launches, their cold starts (instruction cache, the
// first round trip of each kernel) and the gaps between them.  Single process only: a data-
// parallel hook needs the host between the stages.
//
// The latent stage (va:2266-2289, 2346-2369, 2624-2656: clip, z = mu + sigma eps, analytic KL)
// is gauss_latent_fwd_kernel / gauss_latent_bwd_kernel of elementwise.hip restated per tile:
// a wave per cell, the lanes over the latent units, the same wave sum.
__device__ __forceinline__ void tile_latent_fwd_body(const TileLatent& q, const int g) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = g * TC_ROWS;
  const int nr = min(TC_ROWS, q.B - r0);
  const int nw = (q.L + 63) >> 6;
  for (int r = w; r < nr; r += TC_THREADS / 64) {
    const int b = r0 + r;
    float total = 0.f;
    for (int part = 0; part < nw; ++part) {
      const int l = 64 * part + lane;
      const bool live = l < q.L;
      const size_t i = (size_t)b * q.L + (live ? l : 0);
      const float mu = fminf(fmaxf(q.mu_pre[i], -F32_MAX_HALF), F32_MAX_HALF);
      const float ls = fminf(fmaxf(q.ls_pre[i], -3.f), 3.f);
      const float sigma = __expf(ls);
      float kl = 0.f;
      if (live) {
        for (int s = 0; s < q.S; ++s) {
          const size_t o = ((size_t)s * q.B + b) * q.L + l;
          q.z[o] = fmaf(sigma, q.eps[o], mu);
        }
        kl = gauss_kl_elem(mu, sigma, ls);
        q.kl_elem[i] = kl;
      }
      total += wave_sum(kl);
    }
    if (lane == 0) q.kl_cell[b] = total;
  }
}
__device__ __forceinline__ void tile_latent_bwd_body(const TileLatent& q, const int g) {
  const int r0 = g * TC_ROWS;
  const int nr = min(TC_ROWS, q.B - r0);
  const size_t n = (size_t)q.B * q.L;
  for (int e = threadIdx.x; e < nr * q.L; e += TC_THREADS) {
    const size_t i = (size_t)r0 * q.L + e;
    const float mp = q.mu_pre[i], lp = q.ls_pre[i];
    const float mu = fminf(fmaxf(mp, -F32_MAX_HALF), F32_MAX_HALF);
    const float ls = fminf(fmaxf(lp, -3.f), 3.f);
    const float sigma = __expf(ls);
    float gz = 0.f, gze = 0.f;
    for (int s = 0; s < q.S; ++s) {
      const size_t o = (size_t)s * n + i;
      const float ev = q.eps[o];
      const float d = q.dz[o];
      gz += d;
      gze = fmaf(d, ev, gze);
    }
    const float gmu = gauss_kl_dmu(gz, q.kl_coeff, mu);
    const float gls = gauss_kl_dls(gze, sigma, q.kl_coeff);
    q.dmu[i] = (mp >= -F32_MAX_HALF && mp <= F32_MAX_HALF) ? gmu : 0.f;
    q.dls[i] = (lp >= -3.f && lp <= 3.f) ? gls : 0.f;
  }
}

// Between two stages.  sync 1: this workgroup alone (the tile's own rows went through global
// memory: stores complete, then the barrier).  2: every workgroup of the launch, for stages that
// only exchange chunk statistics / chunk sums (ld_agent / st_agent, above): arrival and wait on the
// counter with relaxed agent-scope atomics, no L2 write-back, no invalidate.  3: every workgroup,
// full release / acquire (grid_barrier, common.hpp): whole rows or slabs of other tiles are read
// next.  The barrier is cut in two so that a stage's own inputs travel while the workgroup waits.
__device__ __forceinline__ void tile_chain_stores_done() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
__device__ __forceinline__ void tile_chain_wait(int how, unsigned* bar, unsigned& target) {
  if (how == 3) {
    grid_barrier(bar, target += gridDim.x);
  } else if (how == 2) {
    target += gridDim.x;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins == (1u << 28)) __builtin_trap();
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(TC_THREADS) void tile_chain_fwd_kernel(TileChainFwdArgs q) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  const int g = blockIdx.x;
  unsigned target = q.bar_base;
  TileFwdRegs R;
  TileFwdPart P;
  TCP_BEGIN(0);
  // (stage i as an index into q.f, -1: not a tile stage or no tile g in it; the arguments are
  //  read in place, in the kernel-argument segment: a pointer to one of them would send the whole
  //  table to scratch memory)
  auto tile_of = [&](int i) -> int {
    if (q.kind[i] != TCS_TILE) return -1;
    return g * TC_ROWS < q.f[q.idx[i]].rows ? q.idx[i] : -1;
  };
  int cur = tile_of(0);
  if (cur >= 0) {
    tile_fwd_partials(q.f[cur], P, g);
    tile_fwd_issue(q.f[cur], R, g);
  }
  for (int i = 0; i < q.n; ++i) {
    if (q.kind[i] == TCS_LATENT) {
      if (g * TC_ROWS < q.lat.B) tile_latent_fwd_body(q.lat, g);
    } else if (cur >= 0) {
      tile_fwd_run(q.f[cur], R, P, tsm, g);
    }
    TCP(5 * i);
    if (i + 1 == q.n) break;
    // the next stage: its own rows and weights are requested before the wait for the other
    // workgroups, the chunk statistics of all tiles after it
    tile_chain_stores_done();
    TCP(5 * i + 1);
    cur = tile_of(i + 1);
    if (q.sync[i] == 3) tile_chain_wait(3, q.bar, target);
    if (cur >= 0) tile_fwd_issue(q.f[cur], R, g);
    TCP(5 * i + 2);
    if (q.sync[i] == 2) tile_chain_wait(2, q.bar, target);
    TCP(5 * i + 3);
    if (cur >= 0) tile_fwd_partials(q.f[cur], P, g);
    TCP(5 * i + 4);
  }
}

__global__ __launch_bounds__(TC_THREADS) void tile_chain_bwd_kernel(TileChainBwdArgs q) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  const int g = blockIdx.x;
  unsigned target = q.bar_base;
  for (int i = 0; i < q.n; ++i) {
    const int kind = q.kind[i];
    if (kind == TCS_LATENT) {
      if (g * TC_ROWS < q.lat.B) tile_latent_bwd_body(q.lat, g);
    } else if (kind == TCS_STATS) {
      if (g * TC_ROWS < q.stats_rows)
        tile_bwd_stats_body(q.stats_dh, q.stats_bn, q.stats_rows, q.stats_N, tsm, g);
    } else if (kind == TCS_REDUCE) {
      // the dW / db slabs of the pass, summed in slab order: all workgroups, job after job
      for (int j = 0; j < q.jobs.n_jobs; ++j)
        tile_slab_reduce_job(q.jobs.job[j], g * TC_THREADS + (int)threadIdx.x,
                             (int)gridDim.x * TC_THREADS);
    } else {
      const TileBwdArgs& b = q.b[q.idx[i]];
      if (g * TC_ROWS < b.rows) tile_bwd_body(b, tsm, g);
    }
    if (i + 1 == q.n) break;
    tile_chain_stores_done();
    tile_chain_wait(q.sync[i], q.bar, target);
  }
}

static int tile_chain_setup() {
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(tile_chain_fwd_kernel), (int)TC_FWD_LDS));
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(tile_chain_bwd_kernel), (int)TC_BWD_LDS));
  return 0;
}
// how many workgroups of the resident kernels the current device holds at once (0: unknown);
// the hand-rolled grid barrier needs the whole launch co-resident (cached per device)
int tile_chain_resident_capacity() {
  static thread_local int cached_device = -1;
  static thread_local int cached = 0;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return 0;
  if (device == cached_device) return cached;
  int cap = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && tile_chain_setup() == 0) {
    int fwd = 0, bwd = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fwd, tile_chain_fwd_kernel, TC_THREADS,
                                                     TC_FWD_LDS) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&bwd, tile_chain_bwd_kernel, TC_THREADS,
                                                     TC_BWD_LDS) == hipSuccess)
      cap = (fwd < bwd ? fwd : bwd) * prop.multiProcessorCount;
  }
  cached_device = device;
  cached = cap;
  return cap;
}
static int tile_chain_check(int n, const int* kind, const int* idx, const int* sync, int n_items) {
  SCVAE_ARG(n >= 1 && n <= TCR_MAX_STAGES);
  for (int i = 0; i < n; ++i) {
    SCVAE_ARG(kind[i] >= TCS_TILE && kind[i] <= TCS_REDUCE);
    SCVAE_ARG(kind[i] != TCS_TILE || (idx[i] >= 0 && idx[i] < n_items));
    SCVAE_ARG(i + 1 == n || (sync[i] >= 1 && sync[i] <= 3));
  }
  return 0;
}
// grid barriers of a launch (the host advances the counter's base by barriers * workgroups)
static int tile_chain_barriers(int n, const int* sync) {
  int b = 0;
  for (int i = 0; i + 1 < n; ++i) b += sync[i] >= 2 ? 1 : 0;
  return b;
}
int tile_chain_forward(hipStream_t s, const TileChainFwdArgs& q, int tiles, unsigned* advance) {
  SCVAE_ARG(q.bar && tiles >= 1 && advance);
  if (int rc = tile_chain_check(q.n, q.kind, q.idx, q.sync, TCR_MAX_TILES)) return rc;
  for (int i = 0; i < q.n; ++i) SCVAE_ARG(q.kind[i] == TCS_TILE || q.kind[i] == TCS_LATENT);
  // (a launch with grid barriers needs all its workgroups resident at once)
  SCVAE_ARG(tile_chain_barriers(q.n, q.sync) == 0 || tiles <= tile_chain_resident_capacity());
  if (int rc = tile_chain_setup()) return rc;
  hipLaunchKernelGGL(tile_chain_fwd_kernel, dim3(tiles), dim3(TC_THREADS), TC_FWD_LDS, s, q);
  SCVAE_LAUNCH_CHECK("tile_chain_fwd_kernel");
  *advance = (unsigned)(tile_chain_barriers(q.n, q.sync) * tiles);
  return 0;
}
int tile_chain_backward(hipStream_t s, const TileChainBwdArgs& q, int tiles, unsigned* advance) {
  SCVAE_ARG(q.bar && tiles >= 1 && advance);
  if (int rc = tile_chain_check(q.n, q.kind, q.idx, q.sync, TCR_MAX_TILES)) return rc;
  SCVAE_ARG(tile_chain_barriers(q.n, q.sync) == 0 || tiles <= tile_chain_resident_capacity());
  if (int rc = tile_chain_setup()) return rc;
  hipLaunchKernelGGL(tile_chain_bwd_kernel, dim3(tiles), dim3(TC_THREADS), TC_BWD_LDS, s, q);
  SCVAE_LAUNCH_CHECK("tile_chain_bwd_kernel");
  *advance = (unsigned)(tile_chain_barriers(q.n, q.sync) * tiles);
  return 0;
}

// ---- evaluation steps: the hidden stack in one launch (EvalMlpArgs, kernels.hpp) ----
// A workgroup of eight waves owns 16 cells; wave w owns output columns 16 w .. 16 w + 15 of every
// layer (widths <= 128).  The cells' activations never leave LDS between the layers ([16][130]
// fp32: the MFMA's A operand without bank conflicts); a layer's weights come through registers
// into LDS ([128][144]: the B operand likewise), requested while the layer before is multiplied;
// v_mfma_f32_16x16x4_f32 (fp32 products, as gemm_kernel); bias, moving-statistics normalisation
// and relu on the accumulators.  The two posterior heads multiply the same input; the latent
// stage (gauss_latent_fwd's formulas, a wave per cell) writes z as the decoder's input.  What the
// rest of the step reads -- the normalised layers, mu / log sigma, z, the KL terms -- goes to
// memory as the launch chain leaves it.
constexpr int EM_ROWS = 16;
constexpr int EM_THREADS = 512;
constexpr int EM_PA = 130;          // As pitch: 130 = 2 (mod 32): row i, k -> bank 2 i + k
constexpr int EM_PW = 144;          // Ws pitch: 144 = 16 (mod 32): k, column j -> bank 16 k + j
constexpr size_t EM_LDS = (size_t)(3 * EM_ROWS * EM_PA + 128 * EM_PW) * sizeof(float);
typedef float f32x4a __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(EM_THREADS) void eval_mlp_kernel(EvalMlpArgs q) {
  extern __shared__ __attribute__((aligned(16))) float em[];
  float* As = em;                                   // [16][EM_PA] the layer's input
  float* Ws = As + EM_ROWS * EM_PA;                 // [128][EM_PW] the layer's weights
  float* Mu = Ws + 128 * EM_PW;                     // [16][EM_PA] mu, log sigma pre-activations
  float* Ls = Mu + EM_ROWS * EM_PA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = tid & 127, rl = tid >> 7;           // staging: column c, rows / k rl, rl + 4, ...
  const int r0 = blockIdx.x * EM_ROWS;
  const int nr = min(EM_ROWS, q.rows - r0);
  const int li = lane & 15, kq = lane >> 4;         // MFMA: row / column li, k (or row group) kq
  const int col = 16 * w + li;                      // this lane's output column

  // the weights of a layer: thread (c, rl) holds W[rl + 4 u][c], zero beyond [K, N]
  float wreg[32];
  auto load_w = [&](int o) {
    const float* W = q.op[o].W;
    const int K = q.op[o].K, N = q.op[o].N;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int k = rl + 4 * u;
      wreg[u] = (c < N && k < K) ? W[(size_t)k * N + c] : 0.f;
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int u = 0; u < 32; ++u) Ws[(rl + 4 * u) * EM_PW + c] = wreg[u];
  };
  load_w(0);
  // ---- the input layer's output: normalise + relu -> As (and h0) ----
  {
    const int K0 = q.K0;
    const bool on = c < K0;
    const float mean = on ? q.mean0[c] : 0.f;
    const float istd = on ? rsqrtf(q.var0[c] + BN_EPSILON) : 0.f;
    const float beta = on ? q.beta0[c] : 0.f;
#pragma unroll
    for (int u = 0; u < EM_ROWS / 4; ++u) {
      const int r = rl + 4 * u;
      float v = 0.f;
      if (on && r < nr) {
        v = fmaxf(bn_normalise(q.a0[(size_t)(r0 + r) * K0 + c], mean, istd, beta), 0.f);
        q.h0[(size_t)(r0 + r) * K0 + c] = v;
      }
      As[r * EM_PA + c] = v;                          // (columns up to 127: zero beyond K0)
    }
  }
  store_w();
  __syncthreads();

  for (int o = 0; o < q.n_ops; ++o) {
    const int K = q.op[o].K, N = q.op[o].N, kind = q.op[o].kind;
    // the next layer's weights travel under this layer's product
    if (o + 1 < q.n_ops) load_w(o + 1);
    const bool live = col < N;
    const float bias = live ? q.op[o].b[col] : 0.f;
    float mean = 0.f, istd = 0.f, beta = 0.f;
    if (kind == EM_HIDDEN && live) {
      mean = q.op[o].mean[col];
      istd = rsqrtf(q.op[o].var[col] + BN_EPSILON);
      beta = q.op[o].beta[col];
    }
    f32x4a acc = {0.f, 0.f, 0.f, 0.f};
    if (16 * w < N) {
      const float* ap = As + li * EM_PA + kq;
      const float* bp = Ws + kq * EM_PW + 16 * w + li;
      const int K4 = (K + 3) & ~3;
#pragma unroll 4
      for (int k = 0; k < K4; k += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k], bp[k * EM_PW], acc, 0, 0, 0);
    }
    __syncthreads();                                // (every wave done with As and Ws)
    // accumulator element v: row 4 kq + v, column col
    float* dst = kind == EM_HIDDEN ? As : kind == EM_MU ? Mu : Ls;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = 4 * kq + v;
      const float pre = acc[v] + bias;
      float val = 0.f;
      if (live && r < nr) {
        val = kind == EM_HIDDEN ? fmaxf(bn_normalise(pre, mean, istd, beta), 0.f) : pre;
        if (q.op[o].pre) q.op[o].pre[(size_t)(r0 + r) * N + col] = pre;
        if (kind == EM_HIDDEN && q.op[o].out) q.op[o].out[(size_t)(r0 + r) * N + col] = val;
      }
      dst[r * EM_PA + col] = val;                     // (columns up to 127: zero beyond N)
    }
    if (o + 1 < q.n_ops) store_w();
    __syncthreads();
    if (kind == EM_LOG_SIGMA) {
      // ---- the latent stage: z = mu + sigma eps -> As; KL per unit and per cell ----
      const int L = q.L;
      for (int r = w; r < EM_ROWS; r += EM_THREADS / 64) {
        float total = 0.f;
        for (int part = 0; part < 2; ++part) {
          const int l = 64 * part + lane;
          const bool on = l < L && r < nr;
          const float mu = fminf(fmaxf(Mu[r * EM_PA + l], -F32_MAX_HALF), F32_MAX_HALF);
          const float ls = fminf(fmaxf(Ls[r * EM_PA + l], -3.f), 3.f);
          const float sigma = __expf(ls);
          float zv = 0.f, kl = 0.f;
          if (on) {
            const size_t i = (size_t)(r0 + r) * L + l;
            zv = q.eps ? fmaf(sigma, q.eps[i], mu) : mu;      // (no noise: the deterministic z)
            kl = gauss_kl_elem(mu, sigma, ls);
            q.z[i] = zv;
            q.kl_elem[i] = kl;
          }
          As[r * EM_PA + l] = zv;
          total += wave_sum(kl);
        }
        if (lane == 0 && r < nr) q.kl_cell[r0 + r] = total;
      }
      __syncthreads();
    }
  }
}

int eval_mlp(hipStream_t s, const EvalMlpArgs& q) {
  SCVAE_ARG(q.rows > 0 && q.n_ops >= 1 && q.n_ops <= EM_MAX_OPS);
  SCVAE_ARG(q.a0 && q.h0 && q.mean0 && q.var0 && q.beta0 && q.K0 >= 1 && q.K0 <= 128);
  int K = q.K0, n_ls = 0;
  for (int o = 0; o < q.n_ops; ++o) {
    const EvalMlpArgs::Op& op = q.op[o];
    SCVAE_ARG(op.W && op.b && op.N >= 1 && op.N <= 128 && op.K == K);
    SCVAE_ARG(op.kind >= EM_HIDDEN && op.kind <= EM_LOG_SIGMA);
    if (op.kind == EM_HIDDEN) { SCVAE_ARG(op.mean && op.var && op.beta); K = op.N; }
    else SCVAE_ARG(op.pre);
    if (op.kind == EM_MU) SCVAE_ARG(o + 1 < q.n_ops && q.op[o + 1].kind == EM_LOG_SIGMA);
    if (op.kind == EM_LOG_SIGMA) {
      SCVAE_ARG(o >= 1 && q.op[o - 1].kind == EM_MU && q.op[o - 1].N == op.N && op.N == q.L);
      SCVAE_ARG(q.z && q.kl_elem && q.kl_cell);
      K = q.L; ++n_ls;
    }
  }
  SCVAE_ARG(n_ls <= 1);
  static const int prepared = [] {
    return (int)max_dynamic_lds(reinterpret_cast<const void*>(eval_mlp_kernel), (int)EM_LDS);
  }();
  SCVAE_HIP((hipError_t)prepared);
  hipLaunchKernelGGL(eval_mlp_kernel, dim3((q.rows + EM_ROWS - 1) / EM_ROWS), dim3(EM_THREADS),
                     EM_LDS, s, q);
  SCVAE_LAUNCH_CHECK("eval_mlp_kernel");
  return 0;
}

}  // namespace scvae
