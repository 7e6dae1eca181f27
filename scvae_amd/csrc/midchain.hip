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
// columns, half of k) runs K / 2 fused multiply-adds on 4 accumulators from LDS (the A operand
// staged once per stage with every load in flight and read back four k at a time, the 8-column B
// strip broadcast; the two k-halves are added through LDS) -- the matrix cores would spend 4x
// the cycles on a 32-wide tile that is 8 wide.  What a stage needs that does not depend on the
// stage before (weight strips, the forward pass's activations and statistics, the layer's
// input for dW) is requested BEFORE the grid barrier and lands while the workgroup waits; the one
// dependent matrix is loaded after it.  (A single workgroup walking the same chain, the first
// version of this file, was bound by one CU's fp32 MFMA rate: 5.4 us per [100,100]^2 product,
// 182 us for the chain.)  Measured at B = 100: forward 36 us, backward 60 us.
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
constexpr int MC_KS = 2;            // k-split: threads per (row, column group)
constexpr int MC_THREADS = 256 * MC_KS;   // thread = (row, column group cg of 4 columns, k-part kq):
                                    // 2 waves per SIMD -- a lone wave issues an instruction every ~8 cycles,
                                    // two share the SIMD at one per 4; 256 VGPRs each
constexpr int MC_LD = 132;          // As[r][c] pitch: 16-byte rows, 33 float4 (odd): b128 row reads are conflict-free
constexpr int MC_AS = 128 * MC_LD + 64;          // + dump slots for masked stores
constexpr int MC_SB = 128 * MC_STRIP;            // one strip buffer [k][8]
constexpr int MC_PB = MC_KS * MC_SB;             // the k-parts' partial products
constexpr size_t MC_LDS_BYTES = (MC_AS + 3 * MC_SB + MC_PB + 128) * sizeof(float);

struct Ctx {
  float *As, *S0, *S1, *S2, *P, *wred;
  int tid, row, cg, kq, wave, lane;
  int c0;                            // first column of this workgroup's strip
};

__device__ __forceinline__ Ctx make_ctx(float* smem) {
  Ctx x;
  x.As = smem;
  x.S0 = smem + MC_AS;
  x.S1 = x.S0 + MC_SB;
  x.S2 = x.S1 + MC_SB;
  x.P = x.S2 + MC_SB;
  x.wred = x.P + MC_PB;
  x.tid = threadIdx.x;
  x.row = x.tid & 127;
  x.cg = (x.tid >> 7) & 1;
  x.kq = x.tid >> 8;
  x.wave = x.tid >> 6;
  x.lane = x.tid & 63;
  x.c0 = blockIdx.x * MC_STRIP;
  return x;
}

// The memory matrix Mx[nr][nc] (row pitch ld) into LDS as As[r][c].  Two halves so that a stage can put its loads in flight early
// (before the grid barrier when the data does not depend on the stage before): mat_issue starts
// the loads of the first pass (the whole matrix unless it is wider than 32 columns AND not
// float4-loadable), mat_commit writes them to LDS (and runs any further pass).
constexpr int MC_MR = 16 / MC_KS;   // loads per thread and pass: MC_MR x MC_THREADS float4 = 128 x 32
struct MatRegs { float4 v[MC_MR]; };

struct MatShape {
  int ncv, sh, cv, r0, rstep;
  bool vec;
};
__device__ __forceinline__ MatShape mat_shape(const Ctx& x, const float* Mx, int ld, int nc) {
  MatShape m;
  m.vec = (nc & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(Mx) & 15) == 0;
  m.ncv = m.vec ? nc >> 2 : nc;                   // column vectors per row
  const int span = m.vec ? m.ncv : (nc + 3) & ~3;  // (+ the zeroed padding columns)
  m.sh = span <= 8 ? 3 : span <= 16 ? 4 : span <= 32 ? 5 : span <= 64 ? 6 : 7;
  m.cv = x.tid & ((1 << m.sh) - 1);
  m.r0 = x.tid >> m.sh;
  m.rstep = MC_THREADS >> m.sh;
  return m;
}

__device__ __forceinline__ void mat_issue(const Ctx& x, MatRegs& m, const float* Mx, int ld, int nr,
                                          int nc) {
  const MatShape g = mat_shape(x, Mx, ld, nc);
  const int c = min(g.cv, g.ncv - 1);
  if (g.vec) {                                     // (clamped addresses: no branches inside)
#pragma unroll
    for (int i = 0; i < MC_MR; ++i)
      m.v[i] = *reinterpret_cast<const float4*>(
          Mx + (size_t)min(g.r0 + i * g.rstep, nr - 1) * ld + 4 * c);
  } else {
#pragma unroll
    for (int i = 0; i < MC_MR; ++i)
      m.v[i] = make_float4(Mx[(size_t)min(g.r0 + i * g.rstep, nr - 1) * ld + c], 0.f, 0.f, 0.f);
  }
}

__device__ __forceinline__ void mat_commit(const Ctx& x, const MatRegs& m, const float* Mx, int ld,
                                           int nr, int nc) {
  __syncthreads();                                 // (readers of the previous contents)
  const MatShape g = mat_shape(x, Mx, ld, nc);
  const int dump = 128 * MC_LD + 4 * (x.tid & 15);
  if (g.vec) {
#pragma unroll
    for (int i = 0; i < MC_MR; ++i) {
      const int r = g.r0 + i * g.rstep;
      *reinterpret_cast<float4*>(x.As + ((r < nr && g.cv < g.ncv) ? r * MC_LD + 4 * g.cv : dump)) =
          m.v[i];
    }
  } else {
    // columns nc .. roundup4(nc) - 1 are zeroed: the row-wise product reads whole float4
    const int nc4 = (nc + 3) & ~3;
    const bool pad = g.cv >= nc && g.cv < nc4;
#pragma unroll
    for (int i = 0; i < MC_MR; ++i) {
      const int r = g.r0 + i * g.rstep;
      x.As[(r < nr && g.cv < nc4) ? r * MC_LD + g.cv : dump] = pad ? 0.f : m.v[i].x;
    }
    const int c = min(g.cv, g.ncv - 1);
    for (int rb = g.r0 + MC_MR * g.rstep; rb < nr; rb += MC_MR * g.rstep) {   // (wide and unaligned)
      float v[MC_MR];
#pragma unroll
      for (int i = 0; i < MC_MR; ++i) v[i] = Mx[(size_t)min(rb + i * g.rstep, nr - 1) * ld + c];
#pragma unroll
      for (int i = 0; i < MC_MR; ++i) {
        const int r = rb + i * g.rstep;
        x.As[(r < nr && g.cv < nc4) ? r * MC_LD + g.cv : dump] = pad ? 0.f : v[i];
      }
    }
  }
}

// Strip buffer S[k][j] = base[k sk + j sj] for k < K, j < 8 (zero for j >= nv).  K <= 128.
constexpr int MC_SR = 4 / MC_KS;
struct StripRegs { float v[MC_SR]; };
__device__ __forceinline__ void strip_issue(const Ctx& x, StripRegs& s, const float* base, int sk,
                                            int sj, int K, int nv) {
#pragma unroll
  for (int i = 0; i < MC_SR; ++i) {
    const int e = x.tid + MC_THREADS * i, k = e >> 3, j = e & 7;
    s.v[i] = base[(size_t)min(k, K - 1) * sk + (size_t)min(j, nv - 1) * sj];
  }
}
__device__ __forceinline__ void strip_commit(const Ctx& x, const StripRegs& s, float* S, int nv) {
#pragma unroll
  for (int i = 0; i < MC_SR; ++i) {
    const int e = x.tid + MC_THREADS * i;
    S[e] = (e & 7) < nv ? s.v[i] : 0.f;            // (rows k >= K hold clamped copies: finite)
  }
}

// The k-parts of a product: partial sums through LDS, added in a fixed order; every thread of
// (row, cg) ends with the full sum.
__device__ __forceinline__ void combine_parts(const Ctx& x, float acc[4]) {
  __syncthreads();                                 // (readers of the previous partials)
  *reinterpret_cast<float4*>(x.P + (x.kq * 128 + x.row) * MC_STRIP + 4 * x.cg) =
      make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  const float4* p = reinterpret_cast<const float4*>(x.P + x.row * MC_STRIP + 4 * x.cg);
  float4 t = p[0];
#pragma unroll
  for (int i = 1; i < MC_KS; ++i) {
    const float4 u = p[i * (MC_SB / 4)];
    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
  }
  acc[0] = t.x; acc[1] = t.y; acc[2] = t.z; acc[3] = t.w;
}

// acc[j] = init[j] + sum_k As[row][k] S[k][4 cg + j]: the A row read four k at a time (the
// commit zeroes the columns up to the next multiple of 4; strip rows past K hold finite copies);
// each k-part of threads takes its share of the float4 groups
__device__ __forceinline__ void strip_product(const Ctx& x, const float* S, int K, float acc[4]) {
  const float4* ap = reinterpret_cast<const float4*>(x.As + x.row * MC_LD);
  const float4* bp = reinterpret_cast<const float4*>(S + 4 * x.cg);
  const int K4 = (K + 3) >> 2, per = (K4 + MC_KS - 1) / MC_KS;
  const int lo = x.kq * per, hi = min(lo + per, K4);
  if (x.kq != 0) { acc[0] = 0.f; acc[1] = 0.f; acc[2] = 0.f; acc[3] = 0.f; }
#pragma unroll 2
  for (int k4 = lo; k4 < hi; ++k4) {
    const float4 a = ap[k4];
    const float4 b0 = bp[8 * k4], b1 = bp[8 * k4 + 2], b2 = bp[8 * k4 + 4], b3 = bp[8 * k4 + 6];
    acc[0] = fmaf(a.x, b0.x, acc[0]); acc[1] = fmaf(a.x, b0.y, acc[1]);
    acc[2] = fmaf(a.x, b0.z, acc[2]); acc[3] = fmaf(a.x, b0.w, acc[3]);
    acc[0] = fmaf(a.y, b1.x, acc[0]); acc[1] = fmaf(a.y, b1.y, acc[1]);
    acc[2] = fmaf(a.y, b1.z, acc[2]); acc[3] = fmaf(a.y, b1.w, acc[3]);
    acc[0] = fmaf(a.z, b2.x, acc[0]); acc[1] = fmaf(a.z, b2.y, acc[1]);
    acc[2] = fmaf(a.z, b2.z, acc[2]); acc[3] = fmaf(a.z, b2.w, acc[3]);
    acc[0] = fmaf(a.w, b3.x, acc[0]); acc[1] = fmaf(a.w, b3.y, acc[1]);
    acc[2] = fmaf(a.w, b3.z, acc[2]); acc[3] = fmaf(a.w, b3.w, acc[3]);
  }
  combine_parts(x, acc);
}

// acc[j] += sum_k As[k][row] S[k][4 cg + j]: the transposed product (dW = in^T dA) reads the same
// staged matrix column-wise -- lanes are consecutive units, consecutive banks
__device__ __forceinline__ void strip_product_t(const Ctx& x, const float* S, int K, float acc[4]) {
  const float* ap = x.As + x.row;
  const float4* bp = reinterpret_cast<const float4*>(S + 4 * x.cg);
  const int per = (K + MC_KS - 1) / MC_KS;
  const int lo = x.kq * per, hi = min(lo + per, K);
  if (x.kq != 0) { acc[0] = 0.f; acc[1] = 0.f; acc[2] = 0.f; acc[3] = 0.f; }
#pragma unroll 8
  for (int k = lo; k < hi; ++k) {
    const float a = ap[k * MC_LD];
    const float4 b = bp[2 * k];
    acc[0] = fmaf(a, b.x, acc[0]);
    acc[1] = fmaf(a, b.y, acc[1]);
    acc[2] = fmaf(a, b.z, acc[2]);
    acc[3] = fmaf(a, b.w, acc[3]);
  }
  combine_parts(x, acc);
}

// Column sums over the 128 rows of N per-thread values (the caller zeroes rows that do not
// exist).  Within a wave: four data-parallel steps sum each row of 16 lanes, the four row totals
// are read back and added in lane order; then the two waves of a column group through LDS.
// Fixed order.
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return ((r0 + r1) + r2) + r3;
}

template <int N>
__device__ __forceinline__ void column_sums(const Ctx& x, float* v) {
  // (every k-part holds the same values: part 0 does the sums, all read the result)
  if (x.kq == 0) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = wave_sum(v[j]);
  }
  __syncthreads();                                 // (previous use of wred)
  if (x.kq == 0 && x.lane == 0)
#pragma unroll
    for (int j = 0; j < N; ++j) x.wred[x.wave * 8 + j] = v[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = x.wred[(2 * x.cg) * 8 + j] + x.wred[(2 * x.cg + 1) * 8 + j];
}

// the thread's four columns
#define MC_COL(j) (x.c0 + 4 * x.cg + (j))
// stores of a (row, cg)'s four columns are shared out among its k-parts
#define MC_MINE(j) (((j) % MC_KS) == x.kq)

// ---- one forward layer: a = in W + b on the strip, batch-norm statistics, h = relu(bn(a))
//      (bn_fwd_* of elementwise.hip).  `in` = nullptr: d.a already holds the pre-activation (the
//      input layer's product ran outside). ----
struct FwdPre {
  StripRegs w;
  float b[4], beta[4], mean[4], var[4];
};

// what does not depend on the stage before: weights, bias, beta, moving statistics
__device__ __forceinline__ void forward_prefetch(const Ctx& x, const MidLayer& d, bool product,
                                                 int training, FwdPre& p) {
  const int N = d.n_out;
  if (x.c0 >= N) return;                           // (uniform per workgroup)
  if (product) strip_issue(x, p.w, d.W + x.c0, N, 1, d.n_in, min(MC_STRIP, N - x.c0));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = min(MC_COL(j), N - 1);
    p.b[j] = product ? d.b[col] : 0.f;
    p.beta[j] = d.beta[col];
    if (!training) { p.mean[j] = d.mov_mean[col]; p.var[j] = d.mov_var[col]; }
  }
}

__device__ __forceinline__ void forward_run(const Ctx& x, const MidLayer& d, const float* in,
                                            int rows, int training, FwdPre& p) {
  const int N = d.n_out;
  if (x.c0 >= N) return;
  float a[4];
  if (in != nullptr) {
    MatRegs m;
    mat_issue(x, m, in, d.n_in, rows, d.n_in);
    mat_commit(x, m, in, d.n_in, rows, d.n_in);
    strip_commit(x, p.w, x.S0, min(MC_STRIP, N - x.c0));
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = p.b[j];
    __syncthreads();
    strip_product(x, x.S0, d.n_in, a);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (MC_MINE(j) && x.row < rows && MC_COL(j) < N) d.a[(size_t)x.row * N + MC_COL(j)] = a[j];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      a[j] = d.a[(size_t)min(x.row, rows - 1) * N + min(MC_COL(j), N - 1)];
  }
  if (training) {
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (x.row < rows) ? a[j] : 0.f;
    column_sums<4>(x, s);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      p.mean[j] = s[j] / (float)rows;
      const float c = a[j] - p.mean[j];
      s[j] = (x.row < rows) ? c * c : 0.f;
    }
    column_sums<4>(x, s);
#pragma unroll
    for (int j = 0; j < 4; ++j) p.var[j] = s[j] / (float)rows;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = MC_COL(j);
    if (col < N) {
      if (training && x.row == 0 && x.kq == 0) { d.stats[col] = p.mean[j]; d.stats[N + col] = p.var[j]; }
      if (MC_MINE(j) && x.row < rows)
        d.h[(size_t)x.row * N + col] =
            fmaxf((a[j] - p.mean[j]) * rsqrtf(p.var[j] + BN_EPSILON) + p.beta[j], 0.f);
    }
  }
}

// ---- one backward layer: the gradient of the layer's output strip (from `dh_full`, a [rows, N]
//      matrix in memory, or from the layer above as dA_next [rows, N_next] W_next^T), batch-norm
//      backward (bn_bwd_* of elementwise.hip, one group: dbeta, moving statistics, [s1 | s2], dA),
//      dA to memory for the stage below and the strip of dW = in^T dA. ----
struct BwdPre {
  StripRegs w;                 // W_next^T strip
  float h[4], a[4], mean[4], var[4];
  MatRegs in_t;                // the layer's input (for dW)
};

__device__ __forceinline__ void bn_backward_prefetch(const Ctx& x, const MidLayer& d,
                                                     const float* in, int rows, BwdPre& p) {
  const int N = d.n_out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = min(MC_COL(j), N - 1);
    const size_t o = (size_t)min(x.row, rows - 1) * N + col;
    p.h[j] = d.h[o];
    p.a[j] = d.a[o];
    p.mean[j] = d.stats[col];
    p.var[j] = d.stats[N + col];
  }
  if (in != nullptr) mat_issue(x, p.in_t, in, d.n_in, rows, d.n_in);
}

__device__ __forceinline__ void backward_prefetch(const Ctx& x, const MidLayer& d, const float* in,
                                                  int rows, const MidLayer* next, BwdPre& p) {
  const int N = d.n_out;
  if (x.c0 >= N) return;
  if (next != nullptr)
    strip_issue(x, p.w, next->W + (size_t)x.c0 * next->n_out, 1, next->n_out, next->n_out,
                min(MC_STRIP, N - x.c0));
  bn_backward_prefetch(x, d, in, rows, p);
}

// from dh[4] on: batch-norm backward, dA out, weight-gradient strip
__device__ __forceinline__ void backward_tail(const Ctx& x, const MidLayer& d, const float* in,
                                              int rows, const float* dh, float* da_out,
                                              const BwdPre& p) {
  const int N = d.n_out;
  float g[4], xh[4], istd[4], s[8], da[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    istd[j] = rsqrtf(p.var[j] + BN_EPSILON);
    const bool in_range = x.row < rows && MC_COL(j) < N;
    g[j] = (in_range && p.h[j] > 0.f) ? dh[j] : 0.f;
    xh[j] = in_range ? (p.a[j] - p.mean[j]) * istd[j] : 0.f;
    s[j] = g[j];
    s[4 + j] = g[j] * xh[j];
  }
  column_sums<8>(x, s);
  const float inv = 1.f / (float)rows;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = MC_COL(j);
    da[j] = 0.f;
    if (col < N) {
      if (x.row == 0 && x.kq == 0) {
        d.stats[2 * N + col] = s[j];
        d.stats[3 * N + col] = s[4 + j];
        d.dbeta[col] = s[j];
        const float bessel = (float)rows / (float)(rows > 1 ? rows - 1 : 1);
        d.mov_mean[col] -= (d.mov_mean[col] - p.mean[j]) * BN_UPDATE_RATE;
        d.mov_var[col] -= (d.mov_var[col] - p.var[j] * bessel) * BN_UPDATE_RATE;
      }
      if (x.row < rows) {
        da[j] = istd[j] * (g[j] - s[j] * inv - xh[j] * (s[4 + j] * inv));
        if (MC_MINE(j)) da_out[(size_t)x.row * N + col] = da[j];
      }
    }
  }
  if (in == nullptr) return;
  // dW strip = in^T dA: thread row = input unit, K = the rows
  if (x.kq == 0)
    *reinterpret_cast<float4*>(x.S1 + x.row * MC_STRIP + 4 * x.cg) =
        make_float4(da[0], da[1], da[2], da[3]);
  mat_commit(x, p.in_t, in, d.n_in, rows, d.n_in);           // As[row][unit]
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  strip_product_t(x, x.S1, rows, acc);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (MC_MINE(j) && x.row < d.n_in && MC_COL(j) < N) d.dW[(size_t)x.row * N + MC_COL(j)] = acc[j];
}

__device__ __forceinline__ void backward_run(const Ctx& x, const MidLayer& d, const float* in,
                                             int rows, const float* dh_full, const float* da_next,
                                             const MidLayer* next, float* da_out, const BwdPre& p) {
  const int N = d.n_out;
  if (x.c0 >= N) return;
  float dh[4] = {0.f, 0.f, 0.f, 0.f};
  if (dh_full != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      dh[j] = dh_full[(size_t)min(x.row, rows - 1) * N + min(MC_COL(j), N - 1)];
  } else {
    const int Kn = next->n_out;                    // W_next: [N, Kn]
    MatRegs m;
    mat_issue(x, m, da_next, Kn, rows, Kn);
    mat_commit(x, m, da_next, Kn, rows, Kn);
    strip_commit(x, p.w, x.S0, min(MC_STRIP, N - x.c0));
    __syncthreads();
    strip_product(x, x.S0, Kn, dh);
  }
  backward_tail(x, d, in, rows, dh, da_out, p);
}

}  // namespace

// ---- forward ----
__global__ __launch_bounds__(MC_THREADS) void vae_mid_forward_kernel(MidChainArgs) {
  // (the argument block is read in place: indexing a by-value copy with a loop variable would
  //  put the layer table in scratch)
  const MidChainArgs& q = *(const MidChainArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) float mc_smem[];
  const Ctx x = make_ctx(mc_smem);
  const int B = q.cells, S = q.samples, R = B * S, L = q.latent;
  unsigned target = q.bar_base;
  const MidLayer& last = q.enc[q.n_enc - 1];
  const int H = last.n_out;
  FwdPre fp;
  // posterior-head operands (independent of the encoder's result)
  StripRegs wm, wl;
  float bm[4], bl[4], e0[4];
  // encoder: layer 0 arrives as its pre-activation (the input-layer product ran outside).
  // Before each grid barrier the next stage's independent loads are put in flight.
  forward_prefetch(x, q.enc[0], false, q.training, fp);
  for (int i = 0; i < q.n_enc; ++i) {
    forward_run(x, q.enc[i], i > 0 ? q.enc[i - 1].h : nullptr, B, q.training, fp);
    if (i + 1 < q.n_enc) {
      forward_prefetch(x, q.enc[i + 1], true, q.training, fp);
    } else if (x.c0 < L) {
      const int nv = min(MC_STRIP, L - x.c0);
      strip_issue(x, wm, q.mu.W + x.c0, L, 1, H, nv);
      strip_issue(x, wl, q.ls.W + x.c0, L, 1, H, nv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = min(MC_COL(j), L - 1);
        bm[j] = q.mu.b[col];
        bl[j] = q.ls.b[col];
        e0[j] = q.deterministic ? 0.f : q.eps[(size_t)min(x.row, B - 1) * L + col];
      }
    }
    grid_barrier(q.bar, target += MC_WGS);
  }
  // posterior heads, reparameterised sample, analytic KL (gauss_latent_fwd_kernel) on the strip
  if (x.c0 < L) {
    const int nv = min(MC_STRIP, L - x.c0);
    MatRegs m;
    mat_issue(x, m, last.h, H, B, H);
    mat_commit(x, m, last.h, H, B, H);
    strip_commit(x, wm, x.S0, nv);
    strip_commit(x, wl, x.S1, nv);
    __syncthreads();
    strip_product(x, x.S0, H, bm);
    strip_product(x, x.S1, H, bl);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = MC_COL(j);
      if (MC_MINE(j) && x.row < B && col < L) {
        const size_t i = (size_t)x.row * L + col;
        q.mu_pre[i] = bm[j];
        q.ls_pre[i] = bl[j];
        const float mu = fminf(fmaxf(bm[j], -F32_MAX_HALF), F32_MAX_HALF);
        const float ls = fminf(fmaxf(bl[j], -3.f), 3.f);
        const float sigma = __expf(ls);
        if (q.deterministic) {
          q.z[i] = mu;
        } else {
          q.z[i] = fmaf(sigma, e0[j], mu);
          for (int s = 1; s < S; ++s) {
            const size_t o = (size_t)s * B * L + i;
            q.z[o] = fmaf(sigma, q.eps[o], mu);
          }
        }
        q.kl_elem[i] = gauss_kl_elem(mu, sigma, ls);
      }
    }
  }
  forward_prefetch(x, q.dec[0], true, q.training, fp);
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
    forward_run(x, q.dec[j], j > 0 ? q.dec[j - 1].h : q.z, R, q.training, fp);
    if (j + 1 < q.n_dec) {
      forward_prefetch(x, q.dec[j + 1], true, q.training, fp);
      grid_barrier(q.bar, target += MC_WGS);
    }
  }
}

// ---- backward: from dd = d(-ELBO_w) / d(decoder output) (q.buf[0]) down to the gradient with
//      respect to the input layer's pre-activation (q.da0), all parameter gradients of the
//      chain on the way ----
__global__ __launch_bounds__(MC_THREADS) void vae_mid_backward_kernel(MidChainArgs) {
  const MidChainArgs& q = *(const MidChainArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) float mc_smem[];
  const Ctx x = make_ctx(mc_smem);
  const int B = q.cells, S = q.samples, R = B * S, L = q.latent;
  unsigned target = q.bar_base;
  const MidLayer& last = q.enc[q.n_enc - 1];
  const MidLayer& d0 = q.dec[0];
  const int H = last.n_out, K0 = d0.n_out;
  BwdPre bp;
  // latent-stage operands
  StripRegs w0;
  MatRegs ht;
  float mp[4], lp[4], e0[4];
  // dA of the layer just done, for the stage below: buf[1] / buf[0] in turn (buf[0]'s dd is
  // read, strip by strip, only in the first stage; buf[2] = da0 is the chain's result)
  const float* da_prev = nullptr;
  int turn = 1;
  backward_prefetch(x, q.dec[q.n_dec - 1], q.n_dec > 1 ? q.dec[q.n_dec - 2].h : q.z, R, nullptr, bp);
  for (int j = q.n_dec - 1; j >= 0; --j) {
    float* out = q.buf[turn];
    backward_run(x, q.dec[j], j > 0 ? q.dec[j - 1].h : q.z, R,
                 j == q.n_dec - 1 ? q.buf[0] : nullptr, da_prev,
                 j == q.n_dec - 1 ? nullptr : &q.dec[j + 1], out, bp);
    da_prev = out;
    turn ^= 1;
    if (j > 0) {
      backward_prefetch(x, q.dec[j - 1], j > 1 ? q.dec[j - 2].h : q.z, R, &q.dec[j], bp);
    } else if (x.c0 < L) {
      strip_issue(x, w0, d0.W + (size_t)x.c0 * K0, 1, K0, K0, min(MC_STRIP, L - x.c0));
      mat_issue(x, ht, last.h, H, B, H);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const size_t i = (size_t)min(x.row, B - 1) * L + min(MC_COL(jj), L - 1);
        mp[jj] = q.mu_pre[i];
        lp[jj] = q.ls_pre[i];
        e0[jj] = q.deterministic ? 0.f : q.eps[i];
      }
    }
    grid_barrier(q.bar, target += MC_WGS);
  }
  // latent stage on the strip of latent units: dz = dA_dec0 W_dec0^T, gauss_latent_bwd_kernel
  // (analytic KL: d(-ELBO_w)/dKL_cell = kl_coeff), the heads' dW = h^T dpre and db
  if (x.c0 < L) {
    const int nv = min(MC_STRIP, L - x.c0);
    MatRegs m;
    mat_issue(x, m, da_prev, K0, R, K0);
    mat_commit(x, m, da_prev, K0, R, K0);
    strip_commit(x, w0, x.S0, nv);
    __syncthreads();
    float dz[4] = {0.f, 0.f, 0.f, 0.f};
    strip_product(x, x.S0, K0, dz);
    if (x.kq == 0)
      *reinterpret_cast<float4*>(x.S2 + x.row * MC_STRIP + 4 * x.cg) =
          make_float4(dz[0], dz[1], dz[2], dz[3]);    // (rows >= R: unread)
    __syncthreads();
    float dm[4], dl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = MC_COL(j);
      dm[j] = 0.f; dl[j] = 0.f;
      if (x.row < B && col < L) {
        const size_t i = (size_t)x.row * L + col;
        const float mu = fminf(fmaxf(mp[j], -F32_MAX_HALF), F32_MAX_HALF);
        const float ls = fminf(fmaxf(lp[j], -3.f), 3.f);
        const float sigma = __expf(ls);
        float gz = dz[j], gze = dz[j] * e0[j];
        for (int s = 1; s < S; ++s) {
          const float dzv = x.S2[(s * B + x.row) * MC_STRIP + 4 * x.cg + j];
          gz += dzv;
          gze += dzv * q.eps[(size_t)s * B * L + i];
        }
        const float gmu = gauss_kl_dmu(gz, q.kl_coeff, mu);
        const float gls = gauss_kl_dls(gze, sigma, q.kl_coeff);
        dm[j] = (mp[j] >= -F32_MAX_HALF && mp[j] <= F32_MAX_HALF) ? gmu : 0.f;
        dl[j] = (lp[j] >= -3.f && lp[j] <= 3.f) ? gls : 0.f;
        if (MC_MINE(j)) { q.dmu[i] = dm[j]; q.dls[i] = dl[j]; }
      }
    }
    __syncthreads();                                 // (S2 read above, S0 by the product)
    if (x.kq == 0) {
      *reinterpret_cast<float4*>(x.S0 + x.row * MC_STRIP + 4 * x.cg) =
          make_float4(dm[0], dm[1], dm[2], dm[3]);
      *reinterpret_cast<float4*>(x.S1 + x.row * MC_STRIP + 4 * x.cg) =
          make_float4(dl[0], dl[1], dl[2], dl[3]);
    }
    float s[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] = dm[j]; s[4 + j] = dl[j]; }
    column_sums<8>(x, s);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x.row == 0 && x.kq == 0 && MC_COL(j) < L) {
        q.mu.db[MC_COL(j)] = s[j];
        q.ls.db[MC_COL(j)] = s[4 + j];
      }
    mat_commit(x, ht, last.h, H, B, H);              // As[cell][unit]
    __syncthreads();
    float wm[4] = {0.f, 0.f, 0.f, 0.f}, wl[4] = {0.f, 0.f, 0.f, 0.f};
    strip_product_t(x, x.S0, B, wm);
    strip_product_t(x, x.S1, B, wl);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = MC_COL(j);
      if (MC_MINE(j) && x.row < H && col < L) {
        q.mu.dW[(size_t)x.row * L + col] = wm[j];
        q.ls.dW[(size_t)x.row * L + col] = wl[j];
      }
    }
  }
  // the last encoder layer takes dh = dmu Wmu^T + dls Wls^T: both head strips in flight
  StripRegs um, ul;
  if (x.c0 < H) {
    const int nv = min(MC_STRIP, H - x.c0);
    strip_issue(x, um, q.mu.W + (size_t)x.c0 * L, 1, L, L, nv);
    strip_issue(x, ul, q.ls.W + (size_t)x.c0 * L, 1, L, L, nv);
    bn_backward_prefetch(x, last, q.n_enc > 1 ? q.enc[q.n_enc - 2].h : nullptr, B, bp);
  }
  grid_barrier(q.bar, target += MC_WGS);
  // encoder layers, last to first; layer 0 stops at its pre-activation gradient (q.da0)
  for (int i = q.n_enc - 1; i >= 0; --i) {
    const MidLayer& d = q.enc[i];
    const float* in = i > 0 ? q.enc[i - 1].h : nullptr;
    float* out = i > 0 ? q.buf[turn] : q.da0;
    if (i == q.n_enc - 1) {
      if (x.c0 < H) {
        const int nv = min(MC_STRIP, H - x.c0);
        float dh[4] = {0.f, 0.f, 0.f, 0.f};
        MatRegs m1, m2;
        mat_issue(x, m1, q.dmu, L, B, L);
        mat_issue(x, m2, q.dls, L, B, L);
        mat_commit(x, m1, q.dmu, L, B, L);
        strip_commit(x, um, x.S0, nv);
        strip_commit(x, ul, x.S2, nv);
        __syncthreads();
        strip_product(x, x.S0, L, dh);
        mat_commit(x, m2, q.dls, L, B, L);
        __syncthreads();
        strip_product(x, x.S2, L, dh);
        backward_tail(x, d, in, B, dh, out, bp);
      }
    } else {
      backward_run(x, d, in, B, nullptr, da_prev, &q.enc[i + 1], out, bp);
    }
    da_prev = out;
    turn ^= 1;
    if (i > 0) {
      backward_prefetch(x, q.enc[i - 1], i > 1 ? q.enc[i - 2].h : nullptr, B, &q.enc[i], bp);
      grid_barrier(q.bar, target += MC_WGS);
    }
  }
}

// counter advance of one launch: MC_WGS arrivals per grid barrier
unsigned vae_mid_barrier_advance(const MidChainArgs& q, bool backward) {
  const int barriers = backward ? q.n_dec + 1 + (q.n_enc - 1) : q.n_enc + 1 + (q.n_dec - 1);
  return (unsigned)(MC_WGS * barriers);
}

// Per launch: the dynamic-LDS limit is an attribute of the function ON THE CURRENT DEVICE (a
// process that builds engines on several GPUs must set it on each), and the hand-rolled grid
// barrier needs every workgroup of the launch resident at once: checked against the occupancy
// query (cached per device) instead of being assumed.
static int mid_launch_setup() {
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(vae_mid_forward_kernel), (int)MC_LDS_BYTES));
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(vae_mid_backward_kernel), (int)MC_LDS_BYTES));
  return 0;
}

// 1 if all MC_WGS workgroups of both kernels can be co-resident on the current device (one
// workgroup per CU suffices: MC_WGS <= number of CUs and at least one block fits a CU).  The
// plan disables the mid chain otherwise (scvae_plan_bind).  Note that the barrier counter's base
// is a launch argument advanced by the host: a step that uses these kernels cannot be replayed
// from a captured HIP graph.
bool vae_mid_chain_resident() {
  static thread_local int cached_device = -1;
  static thread_local bool cached = false;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return false;
  if (device == cached_device) return cached;
  bool ok = false;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && mid_launch_setup() == 0) {
    int fwd = 0, bwd = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fwd, vae_mid_forward_kernel, MC_THREADS,
                                                     MC_LDS_BYTES) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&bwd, vae_mid_backward_kernel, MC_THREADS,
                                                     MC_LDS_BYTES) == hipSuccess)
      ok = (long)fwd * prop.multiProcessorCount >= MC_WGS &&
           (long)bwd * prop.multiProcessorCount >= MC_WGS;
  }
  cached_device = device;
  cached = ok;
  return ok;
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
