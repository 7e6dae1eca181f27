// Fused decoder output layer of the scVAE step (the dominant cost):
//
//   pre_j = d W_j + b_j                         X_TILDE/<PARAM> heads, va:2466-2489
//   theta_j = clip(act_j(pre_j)) ; log p(t|z)   du:206-305, zero_inflated.py:194-199, va:2583
//   ll[r]  = sum_f log p(t[r,f] | theta[r,f])   va:2584-2590
//   G_j    = gw[r] * d log p / d pre_j          (backward of the above)
//   dW_j   = d^T G_j ; db_j = colsum(G_j) ; dd = sum_j G_j W_j^T
//
// in ONE kernel: the [rows, P*F] pre-activations and their gradients never touch HBM.
//
// Decomposition: one workgroup (4 waves) owns a strip of BN = 64 columns (genes) for the whole
// launch and walks over the rows in tiles of BM = 64:
//   * the strip's weights W_j[:, strip] (P x H x 64 fp32) are loaded into LDS once;
//   * GEMM1 (K = H): each wave computes one 32x32 tile of every head with
//     v_mfma_f32_32x32x2_f32 and runs the likelihood epilogue on its accumulators in
//     registers (t is prefetched from HBM in the accumulator layout);
//   * the G_j tiles go through LDS to become MFMA operands of
//     GEMM2 dW_j[H, strip] += d^T G_j   (accumulated in registers over all row tiles, written once)
//     GEMM3 dd_part[rows, H] = sum_j G_j W_j^T  (this strip's contribution; summed over the
//     strips by dd_reduce_kernel in a fixed order: deterministic, no atomics);
//   * per-row log-likelihood partial sums likewise (ll_reduce_kernel).
// Algorithmic HBM traffic per cell: 4F B (t) + dd slabs; MFMA work 2*P*F*(H + 2*128) flop
// (the H dimension of GEMM2/GEMM3 is padded to the 32-wide MFMA tile).
#include "common.hpp"
#include "kernels.hpp"
#include "likelihood.hpp"

namespace scvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DF_THREADS = 512;  // 8 waves: 2 per SIMD
constexpr int DF_BN = 64;        // columns (genes) per workgroup
constexpr int DF_LD = DF_BN + 1; // LDS row stride of the [.., 64] tiles (odd: conflict-free)

__host__ __device__ inline int df_ldd(int H) { return H | 1; }   // H even -> H + 1
__host__ __device__ inline int df_bm(int P) { return P >= 3 ? 32 : 64; }

constexpr int DF_QUEUE = 1024;   // capacity of the per-tile queue of t > 0 elements

size_t decoder_fused_lds_bytes(int P, int H, bool train) {
  const int BM = df_bm(P);
  size_t floats = (size_t)P * H * DF_LD          // Ws
                  + (size_t)BM * df_ldd(H) + 32  // dsh (+ slack for the padded h tile)
                  + (size_t)P * BM * DF_LD       // Gs: pre_j, then G_j in place
                  + (size_t)BM * DF_LD           // ts
                  + BM + 3 * DF_BN               // gws, bias
                  + 3 * DF_QUEUE + 4;            // sparse-correction queue
  // over-reads of the padded h tiles (h up to 127) must stay inside the allocation
  const size_t need = (size_t)((P - 1) * H + 128) * DF_LD + 64;
  if (floats < need) floats = need;
  (void)train;
  return floats * sizeof(float);
}

// BM rows x 64 columns per step; see the file header for the phases.
template <int KIND, bool TRAIN, int BM>
__global__ __launch_bounds__(DF_THREADS, 2) void decoder_head_kernel(
    const float* __restrict__ d, int R, int H, unsigned magic_h, HeadParams hp, int F,
    const float* __restrict__ t, int B, const float* __restrict__ gw, int inline_lgamma,
    float* __restrict__ ll_part, float* __restrict__ dd_part) {
  using Traits = LikelihoodTraits<KIND>;
  constexpr int P = Traits::P;
  constexpr int BN = DF_BN, LD = DF_LD, NT = DF_THREADS;
  constexpr int MT = BM / 32;                 // 32-row tiles per step
  constexpr int TPR = NT / BM;                // epilogue threads per row (8 or 16)
  constexpr int EPT = BN / TPR;               // epilogue elements per thread (8 or 4)
  constexpr int TLOADS = BM * BN / NT;        // t elements staged per thread
  constexpr int DLOADS = (BM * 126 + NT - 1) / NT;   // upper bound of d elements per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDD = df_ldd(H);
  float* Ws = smem;                                   // [P][H][LD]
  float* dsh = Ws + (size_t)P * H * LD;               // [BM][LDD] (+32 slack); column H = 1
  float* Gs = dsh + (size_t)BM * LDD + 32;            // [P][BM][LD]
  float* ts = Gs + (size_t)P * BM * LD;               // [BM][LD]
  float* gws = ts + (size_t)BM * LD;                  // [BM]
  float* bs = gws + BM;                               // [3][BN]
  int* qidx = reinterpret_cast<int*>(bs + 3 * BN);    // [Q] element index row*64 + col
  float* qr = reinterpret_cast<float*>(qidx + DF_QUEUE);   // [Q] total_count r
  float* qs = qr + DF_QUEUE;                          // [Q] upstream * clip gate
  int* qcount = reinterpret_cast<int*>(qs + DF_QUEUE);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kh = lane >> 5, li = lane & 31;
  const int c0 = blockIdx.x * BN;

  // ---- strip weights and biases -> LDS (once) ----
  for (int i = tid; i < P * H * BN; i += NT) {
    const int c = i & (BN - 1);
    const int jh = i >> 6;  // j*H + h
    const int j = __umulhi((unsigned)jh, magic_h), h = jh - j * H;
    float v = 0.f;
    if (c0 + c < F) v = hp.W[j][(size_t)h * F + c0 + c];
    Ws[(size_t)jh * LD + c] = v;
  }
  if (tid < P * BN) {
    const int j = tid / BN, c = tid % BN;
    bs[tid] = (c0 + c < F) ? hp.b[j][c0 + c] : 0.f;
  }
  if (tid == 0) *qcount = 0;

  // dW tile of this wave: rows h0..h0+31 (incl. the ones-row h == H -> db), columns n0..n0+31
  const int g2_h0 = (w >> 1) * 32, g2_n0 = (w & 1) * 32;
  f32x16 accW[P];
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) accW[j][i] = 0.f;

  // epilogue element ownership: row er, columns ec0 .. ec0+EPT-1
  const int er = tid / TPR, ec0 = (tid % TPR) * EPT;
  const int n_d = BM * H;                     // d-tile elements (contiguous in HBM)

  // register prefetch of the next tile (d: flat contiguous; t: one 256-byte row per wave)
  float dv[DLOADS], tv[TLOADS];
  auto prefetch = [&](int m0) {
    const float* dbase = d + (size_t)m0 * H;
    const int n_valid = min(R - m0, BM) * H;
#pragma unroll
    for (int i = 0; i < DLOADS; ++i) {
      const int e = i * NT + tid;
      dv[i] = (e < n_valid) ? dbase[e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TLOADS; ++i) {
      const int r = i * (NT / BN) + (tid >> 6), c = tid & 63;
      const int grow = m0 + r;
      tv[i] = (grow < R && c0 + c < F) ? t[(size_t)(grow % B) * F + c0 + c] : 0.f;
    }
  };
  prefetch(0);

  for (int m0 = 0; m0 < R; m0 += BM) {
    // ---- registers -> LDS: d tile (+ ones column), t tile, upstream weights ----
#pragma unroll
    for (int i = 0; i < DLOADS; ++i) {
      const int e = i * NT + tid;
      if (e < n_d) {
        const int r = __umulhi((unsigned)e, magic_h), h = e - r * H;
        dsh[(size_t)r * LDD + h] = dv[i];
      }
    }
#pragma unroll
    for (int i = 0; i < TLOADS; ++i)
      ts[(size_t)(i * (NT / BN) + (tid >> 6)) * LD + (tid & 63)] = tv[i];
    if (tid < BM) {
      dsh[(size_t)tid * LDD + H] = (m0 + tid < R) ? 1.f : 0.f;
      if (TRAIN) gws[tid] = (m0 + tid < R) ? gw[m0 + tid] : 0.f;
    }
    __syncthreads();

    // ---- GEMM1: pre_j = d W_j + b_j, 32x32 tiles spread over the waves -> LDS ----
    for (int x = w; x < P * MT * 2; x += NT / 64) {
      const int j = x / (MT * 2), mt = (x / 2) % MT, nt = x & 1;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* arow = dsh + (size_t)(mt * 32 + li) * LDD + kh;      // A[i=row][k=h]
      const float* bcol = Ws + ((size_t)j * H + kh) * LD + nt * 32 + li; // B[k=h][n=col]
#pragma unroll 10
      for (int k = 0; k < H; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k], bcol[(size_t)k * LD], acc, 0, 0, 0);
      const float bv = bs[j * BN + nt * 32 + li];
      float* out = Gs + ((size_t)j * BM + mt * 32 + 4 * kh) * LD + nt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2)) * LD] = acc[r] + bv;
    }
    __syncthreads();

    // next tile's HBM loads fly under the epilogue and GEMM2/GEMM3
    if (m0 + BM < R) prefetch(m0 + BM);

    // ---- likelihood epilogue, dense part: EPT elements of one row per thread; G_j in place.
    //      Elements with t > 0 of the negative-binomial kinds are queued for the correction
    //      lgamma(r+t)-lgamma(r) / digamma(r+t)-digamma(r) (5 % of a count matrix). ----
    float lsum = 0.f;
    const bool row_ok = m0 + er < R;
    {
      const float up = TRAIN ? gws[er] : 0.f;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int c = ec0 + e;
        const bool ok = row_ok && (c0 + c < F);
        float a[P], g[P], lp, r, rgate;
#pragma unroll
        for (int j = 0; j < P; ++j) a[j] = Gs[((size_t)j * BM + er) * LD + c];
        const float tval = ts[(size_t)er * LD + c];
        lik_dense<KIND, TRAIN>(tval, a, lp, g, r, rgate);
        float corr = 0.f;   // value left in the t tile: summed into the row's log-likelihood
        if (ok && tval > 0.f) {
          bool queued = false;
          if (Traits::HAS_R) {
            const int slot = atomicAdd(qcount, 1);
            if (slot < DF_QUEUE) {
              qidx[slot] = er * BN + c;
              qr[slot] = r;
              qs[slot] = up * rgate;
              queued = true;
              corr = tval;   // the queue pass replaces it by the correction
            }
          }
          if (!queued) {
            if (Traits::HAS_R) {   // queue full (dense data): correct in place
              float A, D;
              lgamma_digamma_diff<TRAIN>(r, tval, A, D);
              lp += A;
              if (TRAIN) g[P - 1] += rgate * r * D;
            }
            if (inline_lgamma) lp -= lgamma1p(tval);
          }
        }
        lsum += ok ? lp : 0.f;
        ts[(size_t)er * LD + c] = corr;
        if (TRAIN) {
#pragma unroll
          for (int j = 0; j < P; ++j) Gs[((size_t)j * BM + er) * LD + c] = ok ? up * g[j] : 0.f;
        }
      }
    }
    lds_barrier();
    if (Traits::HAS_R) {
      const int n_q = min(*qcount, DF_QUEUE);
      for (int s = tid; s < n_q; s += NT) {
        const int idx = qidx[s];
        const int row = idx >> 6, c = idx & 63;
        const float tval = ts[(size_t)row * LD + c];
        const float r = qr[s];
        float A, D;
        lgamma_digamma_diff<TRAIN>(r, tval, A, D);
        if (inline_lgamma) A -= lgamma1p(tval);
        ts[(size_t)row * LD + c] = A;
        if (TRAIN) Gs[((size_t)(P - 1) * BM + row) * LD + c] += qs[s] * r * D;
      }
      lds_barrier();
      if (tid == 0) *qcount = 0;
    }
    // ---- per-row partial log-likelihood of this strip (dense part + corrections) ----
    {
#pragma unroll
      for (int e = 0; e < EPT; ++e) lsum += ts[(size_t)er * LD + ec0 + e];
#pragma unroll
      for (int off = 1; off < TPR; off <<= 1) lsum += __shfl_xor(lsum, off, WAVE);
      if ((tid % TPR) == 0 && row_ok) ll_part[(size_t)blockIdx.x * R + m0 + er] = lsum;
    }
    if (!TRAIN) {
      lds_barrier();
      continue;
    }

    // ---- GEMM2: dW_j[h, col] += sum_row d[row, h] G_j[row, col]   (M = h, N = col, K = row);
    //      row h == H of d^T is all ones, so that row of the result is db_j ----
    if (g2_h0 <= H) {
      const float* ap = dsh + (size_t)kh * LDD + g2_h0 + li;        // A[i=h][k=row] = d[row][h]
      const float* bp = Gs + (size_t)kh * LD + g2_n0 + li;          // B[k=row][n=col]
#pragma unroll 8
      for (int k = 0; k < BM; k += 2) {
        const float a = ap[(size_t)k * LDD];
#pragma unroll
        for (int j = 0; j < P; ++j)
          accW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[((size_t)j * BM + k) * LD], accW[j],
                                                         0, 0, 0);
      }
    }
    // ---- GEMM3: dd[row, h] = sum_j sum_col G_j[row, col] W_j[h, col]   (M = row, N = h, K = col)
    {
      const int mt = w % MT, h0 = (w / MT) * 32;
      if (h0 < H && w < MT * 4) {
        f32x16 accD;
#pragma unroll
        for (int i = 0; i < 16; ++i) accD[i] = 0.f;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const float* ap = Gs + ((size_t)j * BM + mt * 32 + li) * LD + kh;   // A[i=row][k=col]
          const float* bp = Ws + ((size_t)j * H + h0 + li) * LD + kh;         // B[k=col][n=h]
#pragma unroll 8
          for (int k = 0; k < BN; k += 2)
            accD = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], accD, 0, 0, 0);
        }
        const int h = h0 + li;
        if (h < H) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int grow = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (grow < R) dd_part[((size_t)blockIdx.x * R + grow) * H + h] = accD[r];
          }
        }
      }
    }
    lds_barrier();
  }

  if (!TRAIN) return;
  // ---- write dW_j[:, strip] (rows h < H) and db_j (row h == H) ----
  if (g2_h0 <= H) {
    const int c = c0 + g2_n0 + li;
    if (c < F) {
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int h = g2_h0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (h < H) hp.dW[j][(size_t)h * F + c] = accW[j][r];
          else if (h == H) hp.db[j][c] = accW[j][r];
        }
    }
  }
}

// ll[r] = sum_strips ll_part[strip][r] - row_const[r % B]
// workgroup = 64 rows x 16 strip lanes (fixed summation order: deterministic)
__global__ __launch_bounds__(1024) void ll_reduce_kernel(const float* __restrict__ ll_part,
                                                         int strips, int R,
                                                         const float* __restrict__ row_const, int B,
                                                         float* __restrict__ ll) {
  __shared__ float red[16][64];
  const int rl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + rl;
  float s = 0.f;
  if (r < R)
    for (int z = g; z < strips; z += 16) s += ll_part[(size_t)z * R + r];
  red[g][rl] = s;
  __syncthreads();
  if (g == 0 && r < R) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][rl];
    ll[r] = t - (row_const ? row_const[r % B] : 0.f);
  }
}

// dd[i] = sum_strips dd_part[strip][i], i over rows*H (float4 where possible)
__global__ __launch_bounds__(256) void dd_reduce_kernel(const float* __restrict__ dd_part,
                                                        int strips, size_t n,
                                                        float* __restrict__ dd) {
  const size_t n4 = n / 4;
  const float4* p4 = reinterpret_cast<const float4*>(dd_part);
  float4* o4 = reinterpret_cast<float4*>(dd);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 8 <= strips; z += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p4[(size_t)(z + u) * n4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < strips; ++z) {
      const float4 v = p4[(size_t)z * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    o4[i] = s;
  }
}

__global__ __launch_bounds__(256) void dd_reduce_scalar_kernel(const float* __restrict__ dd_part,
                                                               int strips, size_t n,
                                                               float* __restrict__ dd) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < strips; ++z) s += dd_part[(size_t)z * n + i];
    dd[i] = s;
  }
}

bool decoder_fused_supported(int H) { return H >= 2 && H <= 126 && (H % 2) == 0; }

size_t decoder_fused_workspace_floats(int rows, int H, int F, bool train) {
  const size_t strips = (size_t)(F + DF_BN - 1) / DF_BN;
  size_t n = strips * rows;                       // ll_part
  if (train) n += strips * (size_t)rows * H + 64; // dd_part (16-byte aligned start)
  return n + 64;
}

template <bool TRAIN>
static int launch_decoder(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                          int F, const float* t, int B, const float* gw, int inline_lgamma,
                          float* ll_part, float* dd_part) {
  const int P = likelihood_heads(kind);
  const size_t lds = decoder_fused_lds_bytes(P, H, TRAIN);
  const int strips = (F + DF_BN - 1) / DF_BN;
  // e / H == umulhi(e, magic_h) for every e < 2^16 used here (H <= 126)
  const unsigned magic_h = (unsigned)(0x100000000ull / (unsigned)H) + 1u;
#define SCVAE_DF(K_)                                                                              \
  do {                                                                                            \
    auto kfn = decoder_head_kernel<K_, TRAIN, (K_ == LK_ZINB ? 32 : 64)>;                         \
    SCVAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
    hipLaunchKernelGGL(kfn, dim3(strips), dim3(DF_THREADS), lds, s, d, rows, H, magic_h, hp, F, t, \
                       B, gw, inline_lgamma, ll_part, dd_part);                                   \
  } while (0)
  switch (kind) {
    case LK_POISSON: SCVAE_DF(LK_POISSON); break;
    case LK_NB: SCVAE_DF(LK_NB); break;
    case LK_ZIP: SCVAE_DF(LK_ZIP); break;
    case LK_ZINB: SCVAE_DF(LK_ZINB); break;
    default: set_error("unknown likelihood kind %d", kind); return -1;
  }
#undef SCVAE_DF
  SCVAE_LAUNCH_CHECK("decoder_head_kernel");
  return 0;
}

// Forward only (is_training=False / importance-weight pass): ll[rows]
int decoder_fused_forward(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                          int F, const float* t, int B, const float* row_const, float* ll,
                          float* workspace) {
  SCVAE_ARG(d && t && ll && workspace && decoder_fused_supported(H));
  if (rows == 0) return 0;
  const int strips = (F + DF_BN - 1) / DF_BN;
  float* ll_part = workspace;
  int rc = launch_decoder<false>(s, kind, d, rows, H, hp, F, t, B, nullptr, row_const ? 0 : 1, ll_part,
                                 nullptr);
  if (rc) return rc;
  hipLaunchKernelGGL(ll_reduce_kernel, dim3((rows + 63) / 64), dim3(1024), 0, s, ll_part, strips,
                     rows, row_const, B, ll);
  SCVAE_LAUNCH_CHECK("ll_reduce_kernel");
  return 0;
}

// Forward + backward: ll[rows], dW_j, db_j (in hp), dd[rows, H]
int decoder_fused_train(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                        int F, const float* t, int B, const float* gw, const float* row_const,
                        float* ll, float* dd, float* workspace, bool kernel_only) {
  SCVAE_ARG(d && t && gw && ll && dd && workspace && decoder_fused_supported(H));
  if (rows == 0) return 0;
  const int strips = (F + DF_BN - 1) / DF_BN;
  float* ll_part = workspace;
  size_t off = ((size_t)strips * rows + 63) / 64 * 64;
  float* dd_part = workspace + off;
  int rc = launch_decoder<true>(s, kind, d, rows, H, hp, F, t, B, gw, row_const ? 0 : 1, ll_part,
                                dd_part);
  if (rc) return rc;
  if (kernel_only) return 0;  // profiling aid: leave the per-strip partials unreduced
  hipLaunchKernelGGL(ll_reduce_kernel, dim3((rows + 63) / 64), dim3(1024), 0, s, ll_part,
                     strips, rows, row_const, B, ll);
  SCVAE_LAUNCH_CHECK("ll_reduce_kernel");
  const size_t n = (size_t)rows * H;
  if (n % 4 == 0) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dd_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dd_part, strips,
                       n, dd);
  } else {
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dd_reduce_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dd_part,
                       strips, n, dd);
  }
  SCVAE_LAUNCH_CHECK("dd_reduce_kernel");
  return 0;
}

}  // namespace scvae
