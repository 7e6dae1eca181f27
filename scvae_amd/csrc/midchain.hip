// The middle of a VAE step for SMALL minibatches, in two launches.
//
// Between the encoder's input layer and the likelihood heads a training step of the reference
// (mu:38-126, va:2221-2457, 2624-2656) is a chain of small dense layers on [rows, <=128] tensors:
// batch norm, ReLU, [rows,100] x [100,100] products, the posterior heads, the reparameterised
// sample, and all of it again backwards.  At the reference's default minibatch of 100 cells these
// were 27 dependent launches; a dependent launch costs >= ~4.5 us on this part whatever it does
// (profiles/r02_b100_kernel_stats.txt), and a [100,100]^2 product about 4 us on top, so the chain
// took ~200 of the step's 417 us.
//
// Here the chain is one launch forward and one backward, each of MC_WGS = 16 workgroups that walk
// the layers together with a grid barrier (one atomic counter, agent-scope release / acquire)
// between layers.  The work is split by COLUMN STRIPS of the layer being produced: workgroup g
// owns output units [8g, 8g + 8) of every layer for all rows.  With that ownership
//   * a layer's product, its bias, its batch-norm statistics (column sums over rows the
//     workgroup already holds), normalisation and ReLU are one stage with no exchange inside;
//   * backwards, the gradient of a layer's output strip (dA_next W_next^T restricted to the
//     strip), the batch-norm backward sums, dA and the strip of dW = in^T dA are one stage.
// So a layer costs one grid barrier each way (8 barriers for the 2 + 2 layer model) instead of
// 3 - 4 launches.  A strip product is [rows <= 128] x [K <= 128] x [8]: thread (row, group of 4
// columns) runs K fused multiply-adds on 4 accumulators from LDS (the A operand staged once per
// stage with every load in flight, the 8-column B strip broadcast) -- the matrix cores would
// spend 4x the cycles on a 32-wide tile that is 8 wide.  (A single workgroup walking the same
// chain, the first version of this file, was bound by one CU's fp32 MFMA rate: 5.4 us per
// [100,100]^2 product, 182 us for the chain.)
//
// Same formulas as the stand-alone kernels (gemm.hip, elementwise.hip); the stages communicate
// through the plan's workspace exactly like the launches they replace, so the rest of the step
// (likelihood heads before, the input layer's weight gradient after) is unchanged.  All sums run
// in a fixed order: the step stays bitwise repeatable.
//
// Eligibility is decided by the plan (plan.hip: mid_chain_ok): VAE with batch norm and hidden
// layers on both sides, analytic KL, no dropout, no decoder extras, single process, rows and
// widths <= 128.  Everything else keeps the launch chain.
#include "common.hpp"
#include "kernels.hpp"

namespace scvae {

namespace {

constexpr int MC_WGS = 16;          // column strips
constexpr int MC_STRIP = 8;         // columns per strip: MC_WGS * MC_STRIP = 128 >= any width
constexpr int MC_THREADS = 256;     // thread = (row = tid & 127, column group cg = tid >> 7)
constexpr int MC_LD = 129;          // As[m][k] pitch: conflict-free for row-per-lane reads
constexpr int MC_AS = 128 * MC_LD + 64;          // + a dump slot for masked stores
constexpr int MC_SB = 128 * MC_STRIP;            // one strip buffer [k][8]
constexpr size_t MC_LDS_BYTES = (MC_AS + 3 * MC_SB + 64) * sizeof(float);

struct Ctx {
  float *As, *S0, *S1, *S2, *wred;
  int tid, row, cg, wave, lane;
  int c0;                            // first column of this workgroup's strip
};

__device__ __forceinline__ Ctx make_ctx(float* smem) {
  Ctx x;
  x.As = smem;
  x.S0 = smem + MC_AS;
  x.S1 = x.S0 + MC_SB;
  x.S2 = x.S1 + MC_SB;
  x.wred = x.S2 + MC_SB;
  x.tid = threadIdx.x;
  x.row = x.tid & 127;
  x.cg = x.tid >> 7;
  x.wave = x.tid >> 6;
  x.lane = x.tid & 63;
  x.c0 = blockIdx.x * MC_STRIP;
  return x;
}

// All workgroups of the launch arrive; writes made before are visible to every workgroup after.
// `target` = the counter value once all of them have arrived (the counter is never reset: the
// host advances the base by MC_WGS x barriers per launch; the comparison is wrap-safe).
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // (sixteen workgroups are always co-resident on this part; should the others never arrive
    //  -- tens of seconds -- abort the launch loudly rather than hang the queue)
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins == (1u << 28)) __builtin_trap();
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// The memory matrix Mx[nr][nc] (row pitch ld) into LDS, every load of the stage in flight:
// As[r][c] (transpose = false) or As[c][r] (transpose = true).  VEC = 4 reads float4 (nc, ld
// multiples of 4, 16-byte aligned base); VEC = 1 anything.
template <int VEC>
__device__ __forceinline__ void stage_matrix_v(const Ctx& x, const float* Mx, int ld, int nr,
                                               int nc, bool transpose) {
  const int ncv = nc / VEC;                       // column vectors per row
  const int sh = ncv <= 8 ? 3 : ncv <= 16 ? 4 : ncv <= 32 ? 5 : ncv <= 64 ? 6 : 7;
  const int cv = x.tid & ((1 << sh) - 1), r0 = x.tid >> sh, rstep = MC_THREADS >> sh;
  const int sr = transpose ? 1 : MC_LD, sc = transpose ? MC_LD : 1;
  const int dump = 128 * MC_LD + (x.tid & 63);
  for (int rb = r0; rb < nr; rb += 16 * rstep) {
    float v[16][VEC];
#pragma unroll
    for (int i = 0; i < 16; ++i) {                 // (clamped addresses: no branches)
      const int r = min(rb + i * rstep, nr - 1), c = min(cv, ncv - 1);
      const float* src = Mx + (size_t)r * ld + c * VEC;
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      } else {
        v[i][0] = *src;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = rb + i * rstep;
      const bool ok = r < nr && cv < ncv;
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        x.As[ok ? r * sr + (cv * VEC + j) * sc : dump] = v[i][j];
    }
  }
}

__device__ __forceinline__ void stage_matrix(const Ctx& x, const float* Mx, int ld, int nr, int nc,
                                             bool transpose) {
  __syncthreads();                                 // (readers of the previous contents)
  const bool vec = (nc & 3) == 0 && (ld & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(Mx) & 15) == 0;
  if (vec) stage_matrix_v<4>(x, Mx, ld, nr, nc, transpose);
  else stage_matrix_v<1>(x, Mx, ld, nr, nc, transpose);
}

// Strip buffer S[k][j] = base[k sk + j sj] for k < K, j < 8 (zero for j >= nv).  K <= 128.
__device__ __forceinline__ void load_strip(const Ctx& x, float* S, const float* base, int sk,
                                           int sj, int K, int nv) {
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = x.tid + MC_THREADS * i, k = e >> 3, j = e & 7;
    v[i] = base[(size_t)min(k, K - 1) * sk + (size_t)min(j, nv - 1) * sj];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = x.tid + MC_THREADS * i, j = e & 7;
    S[e] = j < nv ? v[i] : 0.f;                    // (rows k >= K hold clamped copies: unread)
  }
}

// acc[j] += sum_k As[row][k] S[k][4 cg + j]
__device__ __forceinline__ void strip_product(const Ctx& x, const float* S, int K, float acc[4]) {
  const float* ap = x.As + x.row * MC_LD;
  const float4* bp = reinterpret_cast<const float4*>(S + 4 * x.cg);
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    const float a = ap[k];
    const float4 b = bp[2 * k];
    acc[0] = fmaf(a, b.x, acc[0]);
    acc[1] = fmaf(a, b.y, acc[1]);
    acc[2] = fmaf(a, b.z, acc[2]);
    acc[3] = fmaf(a, b.w, acc[3]);
  }
}

// Column sums over the 128 rows of N per-thread values (the caller zeroes rows that do not
// exist): a 64-lane butterfly, then the two waves of a column group through LDS.  Fixed order.
template <int N>
__device__ __forceinline__ void column_sums(const Ctx& x, float* v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += __shfl_xor(v[j], m, 64);
  __syncthreads();                                 // (previous use of wred)
  if (x.lane == 0)
#pragma unroll
    for (int j = 0; j < N; ++j) x.wred[x.wave * 8 + j] = v[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = x.wred[(2 * x.cg) * 8 + j] + x.wred[(2 * x.cg + 1) * 8 + j];
}

// ---- batch norm + ReLU on the strip (bn_fwd_* of elementwise.hip): a[4] = the thread's
//      pre-activations ----
__device__ __forceinline__ void strip_bn_forward(const Ctx& x, const MidLayer& d, int rows,
                                                 int training, const float* a) {
  const int N = d.n_out;
  float mu[4], var[4];
  if (training) {
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (x.row < rows) ? a[j] : 0.f;
    column_sums<4>(x, s);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = s[j] / (float)rows;
      const float c = a[j] - mu[j];
      s[j] = (x.row < rows) ? c * c : 0.f;
    }
    column_sums<4>(x, s);
#pragma unroll
    for (int j = 0; j < 4; ++j) var[j] = s[j] / (float)rows;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = min(x.c0 + 4 * x.cg + j, N - 1);
      mu[j] = d.mov_mean[col];
      var[j] = d.mov_var[col];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = x.c0 + 4 * x.cg + j;
    if (col < N) {
      if (training && x.row == 0) { d.stats[col] = mu[j]; d.stats[N + col] = var[j]; }
      if (x.row < rows)
        d.h[(size_t)x.row * N + col] =
            fmaxf((a[j] - mu[j]) * rsqrtf(var[j] + BN_EPSILON) + d.beta[col], 0.f);
    }
  }
}

// One forward layer: a = in W + b on the strip, statistics, h.  `in` = nullptr: d.a already
// holds the pre-activation (the input layer's product ran outside).
__device__ void forward_layer(const Ctx& x, const MidLayer& d, const float* in, int rows,
                              int training) {
  const int N = d.n_out;
  if (x.c0 >= N) return;                           // (uniform per workgroup)
  const int nv = min(MC_STRIP, N - x.c0);
  float a[4];
  if (in != nullptr) {
    stage_matrix(x, in, d.n_in, rows, d.n_in, false);
    load_strip(x, x.S0, d.W + x.c0, N, 1, d.n_in, nv);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = d.b[min(x.c0 + 4 * x.cg + j, N - 1)];
    __syncthreads();
    strip_product(x, x.S0, d.n_in, a);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = x.c0 + 4 * x.cg + j;
      if (x.row < rows && col < N) d.a[(size_t)x.row * N + col] = a[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      a[j] = d.a[(size_t)min(x.row, rows - 1) * N + min(x.c0 + 4 * x.cg + j, N - 1)];
  }
  strip_bn_forward(x, d, rows, training, a);
}

// ---- batch-norm backward on the strip (bn_bwd_* of elementwise.hip, one group): from the
// gradient of the layer output dh[4] to dA[4] (zero where the row or column does not exist);
// dbeta, moving statistics, [s1 | s2] on the way.
__device__ __forceinline__ void strip_bn_backward(const Ctx& x, const MidLayer& d, int rows,
                                                  const float* dh, float* da) {
  const int N = d.n_out;
  float g[4], xh[4], mu[4], var[4], istd[4], s[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = min(x.c0 + 4 * x.cg + j, N - 1);
    const size_t o = (size_t)min(x.row, rows - 1) * N + col;
    const float hv = d.h[o], av = d.a[o];
    mu[j] = d.stats[col];
    var[j] = d.stats[N + col];
    istd[j] = rsqrtf(var[j] + BN_EPSILON);
    const bool in = x.row < rows && x.c0 + 4 * x.cg + j < N;
    g[j] = (in && hv > 0.f) ? dh[j] : 0.f;
    xh[j] = in ? (av - mu[j]) * istd[j] : 0.f;
    s[j] = g[j];
    s[4 + j] = g[j] * xh[j];
  }
  column_sums<8>(x, s);
  const float inv = 1.f / (float)rows;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = x.c0 + 4 * x.cg + j;
    da[j] = 0.f;
    if (col < N) {
      if (x.row == 0) {
        d.stats[2 * N + col] = s[j];
        d.stats[3 * N + col] = s[4 + j];
        d.dbeta[col] = s[j];
        const float bessel = (float)rows / (float)(rows > 1 ? rows - 1 : 1);
        d.mov_mean[col] -= (d.mov_mean[col] - mu[j]) * BN_UPDATE_RATE;
        d.mov_var[col] -= (d.mov_var[col] - var[j] * bessel) * BN_UPDATE_RATE;
      }
      if (x.row < rows) da[j] = istd[j] * (g[j] - s[j] * inv - xh[j] * (s[4 + j] * inv));
    }
  }
}

// The strip of dW = in^T dA; the dA strip in S (rows x 8, zero padded): thread row = input unit.
__device__ __forceinline__ void strip_weight_gradient(const Ctx& x, const float* in, int n_in,
                                                      int rows, const float* S, float* dW, int N) {
  stage_matrix(x, in, n_in, rows, n_in, true);     // As[unit][row]
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  strip_product(x, S, rows, acc);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = x.c0 + 4 * x.cg + j;
    if (x.row < n_in && col < N) dW[(size_t)x.row * N + col] = acc[j];
  }
}

// The tail of a backward stage: batch-norm backward of the strip, dA to memory (for the stage
// below) and, when the layer's input is part of the chain, the strip of its weight gradient.
__device__ __forceinline__ void backward_tail(const Ctx& x, const MidLayer& d, const float* in,
                                              int rows, const float* dh, float* da_out) {
  const int N = d.n_out;
  float da[4];
  strip_bn_backward(x, d, rows, dh, da);
  *reinterpret_cast<float4*>(x.S1 + x.row * MC_STRIP + 4 * x.cg) =
      make_float4(da[0], da[1], da[2], da[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = x.c0 + 4 * x.cg + j;
    if (x.row < rows && col < N) da_out[(size_t)x.row * N + col] = da[j];
  }
  if (in != nullptr) strip_weight_gradient(x, in, d.n_in, rows, x.S1, d.dW, N);
}

// One backward layer.  The gradient of the layer's output strip comes from `dh_full` (a
// [rows, N] matrix in memory) or from the layer above: dA_next [rows, N_next] W_next^T.
__device__ void backward_layer(const Ctx& x, const MidLayer& d, const float* in, int rows,
                               const float* dh_full, const float* da_next, const MidLayer* next,
                               float* da_out) {
  const int N = d.n_out;
  if (x.c0 >= N) return;
  const int nv = min(MC_STRIP, N - x.c0);
  float dh[4] = {0.f, 0.f, 0.f, 0.f};
  if (dh_full != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      dh[j] = dh_full[(size_t)min(x.row, rows - 1) * N + min(x.c0 + 4 * x.cg + j, N - 1)];
  } else {
    const int Kn = next->n_out;                    // W_next: [N, Kn]
    stage_matrix(x, da_next, Kn, rows, Kn, false);
    load_strip(x, x.S0, next->W + (size_t)x.c0 * Kn, 1, Kn, Kn, nv);
    __syncthreads();
    strip_product(x, x.S0, Kn, dh);
  }
  backward_tail(x, d, in, rows, dh, da_out);
}

}  // namespace

// ---- forward ----
__global__ __launch_bounds__(MC_THREADS) void vae_mid_forward_kernel(MidChainArgs q) {
  extern __shared__ __attribute__((aligned(16))) float mc_smem[];
  const Ctx x = make_ctx(mc_smem);
  const int B = q.cells, S = q.samples, R = B * S, L = q.latent;
  unsigned target = q.bar_base;
  // encoder: layer 0 arrives as its pre-activation (the input-layer product ran outside)
  for (int i = 0; i < q.n_enc; ++i) {
    forward_layer(x, q.enc[i], i > 0 ? q.enc[i - 1].h : nullptr, B, q.training);
    grid_barrier(q.bar, target += MC_WGS);
  }
  // posterior heads, reparameterised sample, analytic KL (gauss_latent_fwd_kernel) on the strip
  if (x.c0 < L) {
    const MidLayer& last = q.enc[q.n_enc - 1];
    const int H = last.n_out, nv = min(MC_STRIP, L - x.c0);
    stage_matrix(x, last.h, H, B, H, false);
    load_strip(x, x.S0, q.mu.W + x.c0, L, 1, H, nv);
    load_strip(x, x.S1, q.ls.W + x.c0, L, 1, H, nv);
    float am[4], al[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = min(x.c0 + 4 * x.cg + j, L - 1);
      am[j] = q.mu.b[col];
      al[j] = q.ls.b[col];
    }
    __syncthreads();
    strip_product(x, x.S0, H, am);
    strip_product(x, x.S1, H, al);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = x.c0 + 4 * x.cg + j;
      if (x.row < B && col < L) {
        const size_t i = (size_t)x.row * L + col;
        q.mu_pre[i] = am[j];
        q.ls_pre[i] = al[j];
        const float mu = fminf(fmaxf(am[j], -F32_MAX_HALF), F32_MAX_HALF);
        const float ls = fminf(fmaxf(al[j], -3.f), 3.f);
        const float sigma = __expf(ls);
        if (q.deterministic) {
          q.z[i] = mu;
        } else {
          for (int s = 0; s < S; ++s) {
            const size_t o = (size_t)s * B * L + i;
            q.z[o] = fmaf(sigma, q.eps[o], mu);
          }
        }
        q.kl_elem[i] = 0.5f * (mu * mu + sigma * sigma - 1.f) - ls;
      }
    }
  }
  grid_barrier(q.bar, target += MC_WGS);
  // kl_cell = sum over the latent units (the last workgroup: its strip is the emptiest)
  if (blockIdx.x == MC_WGS - 1)
    for (int b = x.tid; b < B; b += MC_THREADS) {
      float s = 0.f;
      for (int l = 0; l < L; ++l) s += q.kl_elem[(size_t)b * L + l];
      q.kl_cell[b] = s;
    }
  // decoder
  for (int j = 0; j < q.n_dec; ++j) {
    forward_layer(x, q.dec[j], j > 0 ? q.dec[j - 1].h : q.z, R, q.training);
    if (j + 1 < q.n_dec) grid_barrier(q.bar, target += MC_WGS);
  }
}

// ---- backward: from dd = d(-ELBO_w) / d(decoder output) (q.buf[0]) down to the gradient with
//      respect to the input layer's pre-activation (q.da0), all parameter gradients of the
//      chain on the way ----
__global__ __launch_bounds__(MC_THREADS) void vae_mid_backward_kernel(MidChainArgs q) {
  extern __shared__ __attribute__((aligned(16))) float mc_smem[];
  const Ctx x = make_ctx(mc_smem);
  const int B = q.cells, S = q.samples, R = B * S, L = q.latent;
  unsigned target = q.bar_base;
  // dA of the layer just done, for the stage below: buf[1] / buf[0] in turn (buf[0]'s dd is
  // read, strip by strip, only in the first stage; buf[2] = da0 is the chain's result)
  const float* da_prev = nullptr;
  int turn = 1;
  for (int j = q.n_dec - 1; j >= 0; --j) {
    float* out = q.buf[turn];
    backward_layer(x, q.dec[j], j > 0 ? q.dec[j - 1].h : q.z, R,
                   j == q.n_dec - 1 ? q.buf[0] : nullptr, da_prev,
                   j == q.n_dec - 1 ? nullptr : &q.dec[j + 1], out);
    da_prev = out;
    turn ^= 1;
    grid_barrier(q.bar, target += MC_WGS);
  }
  // latent stage on the strip of latent units: dz = dA_dec0 W_dec0^T, gauss_latent_bwd_kernel
  // (analytic KL: d(-ELBO_w)/dKL_cell = kl_coeff), the heads' dW = h^T dpre and db
  const MidLayer& last = q.enc[q.n_enc - 1];
  if (x.c0 < L) {
    const MidLayer& d0 = q.dec[0];
    const int Kn = d0.n_out, nv = min(MC_STRIP, L - x.c0), H = last.n_out;
    stage_matrix(x, da_prev, Kn, R, Kn, false);
    load_strip(x, x.S0, d0.W + (size_t)x.c0 * Kn, 1, Kn, Kn, nv);
    __syncthreads();
    float dz[4] = {0.f, 0.f, 0.f, 0.f};
    strip_product(x, x.S0, Kn, dz);
    *reinterpret_cast<float4*>(x.S2 + x.row * MC_STRIP + 4 * x.cg) =
        make_float4(dz[0], dz[1], dz[2], dz[3]);      // (rows >= R: unread)
    __syncthreads();
    float dm[4], dl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = x.c0 + 4 * x.cg + j;
      dm[j] = 0.f; dl[j] = 0.f;
      if (x.row < B && col < L) {
        const size_t i = (size_t)x.row * L + col;
        const float mp = q.mu_pre[i], lp = q.ls_pre[i];
        const float mu = fminf(fmaxf(mp, -F32_MAX_HALF), F32_MAX_HALF);
        const float ls = fminf(fmaxf(lp, -3.f), 3.f);
        const float sigma = __expf(ls);
        float gz = 0.f, gze = 0.f;
        for (int s = 0; s < S; ++s) {
          const float dzv = x.S2[(s * B + x.row) * MC_STRIP + 4 * x.cg + j];
          gz += dzv;
          gze += dzv * q.eps[(size_t)s * B * L + i];
        }
        const float gmu = gz + q.kl_coeff * mu;
        const float gls = gze * sigma + q.kl_coeff * (sigma * sigma - 1.f);
        dm[j] = (mp >= -F32_MAX_HALF && mp <= F32_MAX_HALF) ? gmu : 0.f;
        dl[j] = (lp >= -3.f && lp <= 3.f) ? gls : 0.f;
        q.dmu[i] = dm[j];
        q.dls[i] = dl[j];
      }
    }
    __syncthreads();                                 // (S2 read above, S0 by the product)
    *reinterpret_cast<float4*>(x.S0 + x.row * MC_STRIP + 4 * x.cg) =
        make_float4(dm[0], dm[1], dm[2], dm[3]);
    *reinterpret_cast<float4*>(x.S1 + x.row * MC_STRIP + 4 * x.cg) =
        make_float4(dl[0], dl[1], dl[2], dl[3]);
    float s[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] = dm[j]; s[4 + j] = dl[j]; }
    column_sums<8>(x, s);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = x.c0 + 4 * x.cg + j;
      if (x.row == 0 && col < L) { q.mu.db[col] = s[j]; q.ls.db[col] = s[4 + j]; }
    }
    stage_matrix(x, last.h, H, B, H, true);          // As[unit][cell]
    __syncthreads();
    float wm[4] = {0.f, 0.f, 0.f, 0.f}, wl[4] = {0.f, 0.f, 0.f, 0.f};
    strip_product(x, x.S0, B, wm);
    strip_product(x, x.S1, B, wl);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = x.c0 + 4 * x.cg + j;
      if (x.row < H && col < L) {
        q.mu.dW[(size_t)x.row * L + col] = wm[j];
        q.ls.dW[(size_t)x.row * L + col] = wl[j];
      }
    }
  }
  grid_barrier(q.bar, target += MC_WGS);
  // encoder layers, last to first; layer 0 stops at its pre-activation gradient (q.da0)
  for (int i = q.n_enc - 1; i >= 0; --i) {
    const MidLayer& d = q.enc[i];
    const float* in = i > 0 ? q.enc[i - 1].h : nullptr;
    float* out = i > 0 ? q.buf[turn] : q.da0;
    if (i == q.n_enc - 1) {
      // dh = dmu Wmu^T + dls Wls^T on the strip, then the layer's own backward
      const int N = d.n_out;
      if (x.c0 < N) {
        const int nv = min(MC_STRIP, N - x.c0);
        float dh[4] = {0.f, 0.f, 0.f, 0.f};
        stage_matrix(x, q.dmu, L, B, L, false);
        load_strip(x, x.S0, q.mu.W + (size_t)x.c0 * L, 1, L, L, nv);
        load_strip(x, x.S2, q.ls.W + (size_t)x.c0 * L, 1, L, L, nv);
        __syncthreads();
        strip_product(x, x.S0, L, dh);
        stage_matrix(x, q.dls, L, B, L, false);
        __syncthreads();
        strip_product(x, x.S2, L, dh);
        backward_tail(x, d, in, B, dh, out);
      }
    } else {
      backward_layer(x, d, in, B, nullptr, da_prev, &q.enc[i + 1], out);
    }
    da_prev = out;
    turn ^= 1;
    if (i > 0) grid_barrier(q.bar, target += MC_WGS);
  }
}

// counter advance of one launch: MC_WGS arrivals per grid barrier
unsigned vae_mid_barrier_advance(const MidChainArgs& q, bool backward) {
  const int barriers = backward ? q.n_dec + 1 + (q.n_enc - 1) : q.n_enc + 1 + (q.n_dec - 1);
  return (unsigned)(MC_WGS * barriers);
}

static int mid_launch_setup() {
  static bool done = false;
  if (done) return 0;
  SCVAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vae_mid_forward_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)MC_LDS_BYTES));
  SCVAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vae_mid_backward_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)MC_LDS_BYTES));
  done = true;
  return 0;
}

int vae_mid_forward(hipStream_t stream, const MidChainArgs& args) {
  if (int rc = mid_launch_setup()) return rc;
  hipLaunchKernelGGL(vae_mid_forward_kernel, dim3(MC_WGS), dim3(MC_THREADS), MC_LDS_BYTES, stream,
                     args);
  SCVAE_LAUNCH_CHECK("vae_mid_forward_kernel");
  return 0;
}

int vae_mid_backward(hipStream_t stream, const MidChainArgs& args) {
  if (int rc = mid_launch_setup()) return rc;
  hipLaunchKernelGGL(vae_mid_backward_kernel, dim3(MC_WGS), dim3(MC_THREADS), MC_LDS_BYTES, stream,
                     args);
  SCVAE_LAUNCH_CHECK("vae_mid_backward_kernel");
  return 0;
}

}  // namespace scvae
