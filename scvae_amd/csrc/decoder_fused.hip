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
// Decomposition: one workgroup (8 waves, 2 per SIMD) owns a strip of BN = 64 columns (genes) for
// the whole launch and walks over the rows in tiles of BM = 64 (32 for three heads):
//   * the strip's weights W_j[:, strip] (P x H x 64 fp32) are loaded into LDS once;
//   * GEMM1 (K = H): 32x32 v_mfma_f32_32x32x2_f32 tiles spread over the waves, pre_j + b_j -> LDS;
//   * likelihood epilogue in registers: a thread owns rows r, r+16, .. x columns c, c+32 of the
//     tile (t is loaded from HBM into the owner's registers one tile ahead), writes G_j in place
//     of pre_j; the t > 0 corrections of the negative-binomial kinds are compacted per wave
//     (ballot) into a wave-private LDS queue;
//   * the G_j tiles are MFMA operands of
//     GEMM2 dW_j[H, strip] += d^T G_j   (accumulated in registers over all row tiles, written once;
//                                        an appended ones-column of d makes db_j fall out of it)
//     GEMM3 dd_part[rows, H] = sum_j G_j W_j^T  (this strip's contribution; summed over the
//     strips by dd_reduce_kernel in a fixed order: deterministic, no atomics);
//   * per-row log-likelihood partial sums likewise (ll_reduce_kernel).
// Three workgroup barriers per tile; the d tile is double buffered when LDS allows.
// Algorithmic HBM traffic per cell: 4F B (t) + dd slabs; MFMA work 2*P*F*(H + 2*128) flop
// (the H dimension of GEMM2/GEMM3 is padded to the 32-wide MFMA tile).
// decoder_fused2.hip holds a second schedule of the same phases (two pipelined half workgroups),
// which the dispatcher below prefers where it fits.
#include <cstring>
#include <type_traits>

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

constexpr int DF_QCAP = 128;     // wave-private queue of t > 0 elements (512 / 256 elements per wave)

__device__ __forceinline__ int df_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ void df_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

static size_t df_lds_floats(int P, int H, int d_buffers) {
  const int BM = df_bm(P);
  size_t floats = (size_t)P * H * DF_LD              // Ws
                  + d_buffers * ((size_t)BM * df_ldd(H) + 32)  // dsh (+ slack for the padded h tile)
                  + (size_t)P * BM * DF_LD           // Gs: pre_j, then G_j in place
                  + 3 * DF_BN                        // bias
                  + 8 * 4 * DF_QCAP + 4;             // 8 wave-private queues x 4 fields
  // over-reads of the padded h tiles (h up to 127) must stay inside the allocation
  const size_t need = (size_t)((P - 1) * H + 128) * DF_LD + 64;
  if (floats < need) floats = need;
  return floats;
}
constexpr size_t DF_LDS_LIMIT = 160 * 1024;
// the d tile is double buffered when that fits into the 160 KB of LDS (H <= 110 for two heads)
static int df_d_buffers(int P, int H) {
  return df_lds_floats(P, H, 2) * sizeof(float) <= DF_LDS_LIMIT ? 2 : 1;
}
size_t decoder_fused_lds_bytes(int P, int H, bool train) {
  (void)train;
  return df_lds_floats(P, H, df_d_buffers(P, H)) * sizeof(float);
}

// BM rows x 64 columns per step.  Per step: GEMM1 | epilogue (+ next d tile -> LDS) | GEMM2, GEMM3,
// three workgroup barriers; the d tile is double buffered, t goes from HBM straight into the
// registers of the thread that owns the element.
template <int KIND, bool TRAIN, int BM>
__global__ __launch_bounds__(DF_THREADS, 2) void decoder_head_kernel(
    const float* __restrict__ d, int R, int H, unsigned magic_h, HeadParams hp, int F,
    Targets tg, int B, const float* __restrict__ gw, int inline_lgamma,
    float* __restrict__ ll_part, float* __restrict__ dd_part, int d_buffers) {
  using Traits = LikelihoodTraits<KIND>;
  constexpr int P = Traits::P;
  constexpr int BN = DF_BN, LD = DF_LD, NT = DF_THREADS;
  constexpr int MT = BM / 32;                 // 32-row tiles per step
  constexpr int RI = BM / 16;                 // epilogue rows per thread (rows er0 + 16 i)
  constexpr int EPT = 2 * RI;                 // epilogue elements per thread (columns ec, ec + 32)
  constexpr int DLOADS = (BM * 126 + NT - 1) / NT;   // upper bound of d elements per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDD = df_ldd(H);
  const size_t DBUF = (size_t)BM * LDD + 32;          // one d tile (+32 slack)
  float* Ws = smem;                                   // [P][H][LD]
  float* dsh = Ws + (size_t)P * H * LD;               // [d_buffers][BM][LDD]; column H = 1
  float* Gs = dsh + d_buffers * DBUF;                 // [P][BM][LD]
  float* bs = Gs + (size_t)P * BM * LD;               // [3][BN]
  float* qbase = bs + 3 * BN;                         // [8 waves][4][QCAP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = lane >> 5, li = lane & 31;
  const int c0 = blockIdx.x * BN;

  // ---- strip weights and biases -> LDS (once): eight HBM loads in flight per thread ----
  {
    const int c = tid & (BN - 1);
    const bool col_ok = c0 + c < F;
    constexpr int RPP = NT / 64;                        // rows of 64 genes per pass
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const float* Wj = hp.W[j] + c0 + c;
      float* dst = Ws + (size_t)j * H * LD + c;
      int h = tid >> 6;
      for (; h + 7 * RPP < H; h += 8 * RPP) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col_ok ? Wj[(size_t)(h + u * RPP) * F] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[(size_t)(h + u * RPP) * LD] = v[u];
      }
      for (; h < H; h += RPP) dst[(size_t)h * LD] = col_ok ? Wj[(size_t)h * F] : 0.f;
    }
  }
  if (tid < P * BN) {
    const int j = tid / BN, c = tid % BN;
    bs[tid] = (c0 + c < F) ? hp.b[j][c0 + c] : 0.f;
  }

  // dW tile of this wave: rows h0..h0+31 (incl. the ones-row h == H -> db), columns n0..n0+31
  const int g2_h0 = (w >> 1) * 32, g2_n0 = (w & 1) * 32;
  f32x16 accW[P];
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) accW[j][i] = 0.f;

  float dv[DLOADS];          // next d tile: HBM -> registers (under GEMM1) -> LDS (after the epilogue)
  float tv[EPT], up[RI];     // t / upstream weights of the element owner (loaded a step ahead)
#pragma unroll
  for (int i = 0; i < EPT; ++i) tv[i] = 0.f;
#pragma unroll
  for (int i = 0; i < RI; ++i) up[i] = 0.f;

  // `tq` is an opaque copy of the thread index: per-thread addresses are re-derived in every
  // phase instead of living in registers across the whole loop
  auto load_d = [&](int m0, int tq) {
    const float* dbase = d + (size_t)m0 * H + tq;
    const int n_valid = (m0 < R) ? min(R - m0, BM) * H : 0;
#pragma unroll
    for (int i = 0; i < DLOADS; ++i) dv[i] = (i * NT + tq < n_valid) ? dbase[i * NT] : 0.f;
  };
  auto store_d = [&](int m0, int buf, int tq) {
    float* dst = dsh + (size_t)buf * DBUF;
    const int n_d = BM * H;
#pragma unroll
    for (int i = 0; i < DLOADS; ++i) {
      const int e = i * NT + tq;
      if (e < n_d) {
        const int r = __umulhi((unsigned)e, magic_h), h = e - r * H;
        dst[r * LDD + h] = dv[i];
      }
    }
    if (tq < BM) dst[tq * LDD + H] = (m0 + tq < R) ? 1.f : 0.f;
  };
  // element owner: rows er0 + 16 i, columns ec and ec + 32 (one 128-byte row segment per half
  // wave: coalesced HBM loads, conflict-free LDS accesses with the odd stride)
  auto load_t_from = [&](auto* t, int m0, int tq) {
    const int ec = tq & 31, er0 = tq >> 5;
    const int ldt = tg.ld;
    if (m0 + BM <= R && c0 + BN <= F && R == B) {   // full tile, no row wrap: no predicates
      auto* tp = t + (size_t)(m0 + er0) * ldt + c0 + ec;
#pragma unroll
      for (int ri = 0; ri < RI; ++ri) {
        tv[2 * ri] = target_raw(tp[(size_t)(16 * ri) * ldt]);
        tv[2 * ri + 1] = target_raw(tp[(size_t)(16 * ri) * ldt + 32]);
        if (TRAIN) up[ri] = gw[m0 + er0 + 16 * ri];
      }
    } else {
#pragma unroll
      for (int ri = 0; ri < RI; ++ri) {
        const int grow = m0 + er0 + 16 * ri;
        const bool rok = grow < R;
        up[ri] = (TRAIN && rok) ? gw[grow] : 0.f;
        auto* trow = t + (size_t)(rok ? grow % B : 0) * ldt + c0;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
          const int c = ec + 32 * ci;
          tv[2 * ri + ci] = (rok && c0 + c < F) ? target_raw(trow[c]) : 0.f;
        }
      }
    }
  };
  // (fp32 batch or the uint16 minibatch: one uniform branch around the loads; the values stay
  //  raw until they are used)
  auto load_t = [&](int m0, int tq) {
    if (tg.u16) load_t_from(static_cast<const uint16_t*>(tg.p), m0, tq);
    else load_t_from(static_cast<const float*>(tg.p), m0, tq);
  };

  load_d(0, tid);
  store_d(0, 0, tid);
  load_t(0, tid);
  __syncthreads();

  // with one d buffer the next tile can only be stored once GEMM2 has read the current one
  const bool store_early = !TRAIN || d_buffers == 2;
  int buf = 0;
  for (int m0 = 0; m0 < R; m0 += BM, buf ^= (d_buffers - 1)) {
    const float* dcur = dsh + (size_t)buf * DBUF;
    {
      // ---- GEMM1: pre_j = d W_j + b_j, 32x32 tiles spread over the waves -> LDS ----
      const int tq = df_opaque(tid);
      load_d(m0 + BM, tq);                    // next d tile -> registers
      for (int x = w; x < P * MT * 2; x += NT / 64) {
        const int j = x / (MT * 2), mt = (x / 2) % MT, nt = x & 1;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* arow = dcur + (mt * 32 + li) * LDD + kh;        // A[i=row][k=h]
        const float* bcol = Ws + (j * H + kh) * LD + nt * 32 + li;   // B[k=h][n=col]
        int kk = 0;
        for (; kk + 20 <= H; kk += 20) {
#pragma unroll
          for (int u = 0; u < 20; u += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[kk + u], bcol[(kk + u) * LD], acc, 0, 0,
                                                       0);
        }
        for (; kk < H; kk += 2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[kk], bcol[kk * LD], acc, 0, 0, 0);
        const float bv = bs[j * BN + nt * 32 + li];
        float* out = Gs + (j * BM + mt * 32 + 4 * kh) * LD + nt * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2)) * LD] = acc[r] + bv;
      }
    }
    lds_barrier();
    {
      // ---- likelihood epilogue in registers; G_j in place of pre_j.  Elements with t > 0 of
      //      the negative-binomial kinds are compacted per wave (ballots) into a wave-private
      //      queue for the correction lgamma(r+t)-lgamma(r) / digamma(r+t)-digamma(r), so that
      //      the expensive code runs on dense lanes (5 % of a count matrix is non-zero) ----
      // (full tiles take the copy without the row / column bound checks)
      auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int tq = df_opaque(tid);
        const int ln = tq & 63;
        const int ec = tq & 31, er0 = tq >> 5;
        float* q0 = qbase + (tq >> 6) * 4 * DF_QCAP;                    // r -> A
        float* q1 = q0 + DF_QCAP;                                       // t
        float* q2 = q1 + DF_QCAP;                                       // upstream * gate
        int* q3 = reinterpret_cast<int*>(q2 + DF_QCAP);                 // LDS offset row*LD + col
        float lsum[RI];
        int slot[EPT];
        int q_n = 0;
  #pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const int ri = e >> 1, ci = e & 1;
          const int row = er0 + 16 * ri, c = ec + 32 * ci;
          const bool ok = FULL || ((m0 + row < R) && (c0 + c < F));
          const float tval = target_value(tv[e], tg.u16);
          float a[P], g[P], lp, r, rgate;
  #pragma unroll
          for (int j = 0; j < P; ++j) a[j] = Gs[(j * BM + row) * LD + c];
          lik_dense<KIND, TRAIN>(tval, a, lp, g, r, rgate);
          const bool nz = ok && tval > 0.f;
          slot[e] = -1;
          if (Traits::HAS_R) {
            const unsigned long long mask = __ballot(nz);
            if (nz) {
              const int s = q_n + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
              if (s < DF_QCAP) {
                slot[e] = s;
                q0[s] = r;
                q1[s] = tval;
                q2[s] = up[ri] * rgate;
                q3[s] = row * LD + c;
              } else {   // queue full (dense data): correct in place
                float A, D;
                lgamma_digamma_diff<TRAIN>(r, tval, A, D);
                lp += A;
                if (TRAIN) g[P - 1] += rgate * r * D;
                if (inline_lgamma) lp -= lgamma1p(tval);
              }
            }
            q_n += __popcll(mask);
          } else if (nz && inline_lgamma) {
            lp -= lgamma1p(tval);
          }
          if (ci == 0) lsum[ri] = ok ? lp : 0.f;
          else lsum[ri] += ok ? lp : 0.f;
          if (TRAIN) {
  #pragma unroll
            for (int j = 0; j < P; ++j) Gs[(j * BM + row) * LD + c] = ok ? up[ri] * g[j] : 0.f;
          }
          asm volatile("" ::: "memory");   // one element at a time: keeps the register peak low
        }
        if (Traits::HAS_R) {
          df_wave_fence();
          const int n_q = min(q_n, DF_QCAP);
          for (int s = ln; s < n_q; s += 64) {
            const float r = q0[s], tval = q1[s];
            float A, D;
            lgamma_digamma_diff<TRAIN>(r, tval, A, D);
            if (inline_lgamma) A -= lgamma1p(tval);
            q0[s] = A;
            if (TRAIN) Gs[(P - 1) * BM * LD + q3[s]] += q2[s] * r * D;
          }
          df_wave_fence();
  #pragma unroll
          for (int e = 0; e < EPT; ++e)
            if (slot[e] >= 0) lsum[e >> 1] += q0[slot[e]];
          df_wave_fence();   // the queue is reused in the next step
        }
        // ---- per-row partial log-likelihood of this strip ----
  #pragma unroll
        for (int ri = 0; ri < RI; ++ri) {
          float sum = lsum[ri];
  #pragma unroll
          for (int off = 1; off < 32; off <<= 1) sum += __shfl_xor(sum, off, WAVE);
          const int grow = m0 + er0 + 16 * ri;
          if (ec == 0 && (FULL || grow < R)) ll_part[(size_t)blockIdx.x * R + grow] = sum;
        }
      };
      if (m0 + BM <= R && c0 + BN <= F) epilogue(std::true_type{});
      else epilogue(std::false_type{});
      const int tq = df_opaque(tid);
      // next step's operands: d tile -> the other LDS buffer, t / upstream -> registers
      if (store_early) store_d(m0 + BM, buf ^ (d_buffers - 1), tq);
      if (m0 + BM < R) load_t(m0 + BM, tq);
    }
    lds_barrier();
    if (TRAIN) {
      // ---- GEMM2: dW_j[h, col] += sum_row d[row, h] G_j[row, col]   (M = h, N = col, K = row);
      //      row h == H of d^T is all ones, so that row of the result is db_j ----
      if (g2_h0 <= H) {
        const float* ap = dcur + kh * LDD + g2_h0 + li;                // A[i=h][k=row] = d[row][h]
        const float* bp = Gs + kh * LD + g2_n0 + li;                   // B[k=row][n=col]
#pragma unroll 8
        for (int k = 0; k < BM; k += 2) {
          const float a = ap[k * LDD];
#pragma unroll
          for (int j = 0; j < P; ++j)
            accW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[(j * BM + k) * LD], accW[j], 0, 0, 0);
        }
      }
      // ---- GEMM3: dd[row, h] = sum_j sum_col G_j[row, col] W_j[h, col]   (M = row, N = h, K = col)
      const int mt = w % MT, h0 = (w / MT) * 32;
      if (h0 < H && w < MT * 4) {
        f32x16 accD;
#pragma unroll
        for (int i = 0; i < 16; ++i) accD[i] = 0.f;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const float* ap = Gs + (j * BM + mt * 32 + li) * LD + kh;    // A[i=row][k=col]
          const float* bp = Ws + (j * H + h0 + li) * LD + kh;          // B[k=col][n=h]
#pragma unroll 8
          for (int k = 0; k < BN; k += 2)
            accD = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], accD, 0, 0, 0);
        }
        const int h = h0 + li;
        if (h < H) {
          const int r0 = m0 + mt * 32 + 4 * kh;
          float* dst = dd_part + ((size_t)blockIdx.x * R + r0) * H + h;
          if (m0 + BM <= R) {
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(accD[r], dst + ((r & 3) + 8 * (r >> 2)) * H);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ro = (r & 3) + 8 * (r >> 2);
              if (r0 + ro < R) __builtin_nontemporal_store(accD[r], dst + ro * H);
            }
          }
        }
      }
      lds_barrier();
      if (!store_early) {
        store_d(m0 + BM, 0, df_opaque(tid));
        lds_barrier();
      }
    }
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
// workgroup = 16 rows x 64 strip lanes (fixed summation order: deterministic); rows/16 workgroups
// keep every CU busy on this 8 MB read
__global__ __launch_bounds__(1024) void ll_reduce_kernel(const float* __restrict__ ll_part,
                                                         int strips, int R,
                                                         const float* __restrict__ row_const, int B,
                                                         float* __restrict__ ll) {
  __shared__ float red[64][17];
  const int rl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int r = blockIdx.x * 16 + rl;
  float s = 0.f;
  if (r < R) {
    int z = g;
    for (; z + 7 * 64 < strips; z += 8 * 64) {     // (eight strips' loads in flight, same order)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ll_part[(size_t)(z + 64 * u) * R + r];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < strips; z += 64) s += ll_part[(size_t)z * R + r];
  }
  red[g][rl] = s;
  __syncthreads();
  if (g == 0 && r < R) {
    float t = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) t += red[i][rl];
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
      for (int u = 0; u < 8; ++u) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 q = __builtin_nontemporal_load(
            reinterpret_cast<const f32x4*>(p4 + (size_t)(z + u) * n4 + i));
        v[u] = make_float4(q.x, q.y, q.z, q.w);
      }
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

// The same sum for small batches (rows*H/4 below DD_SPLIT_MAX float4 columns): one thread per
// column leaves a B = 100 step with 10 workgroups walking 512 slabs one after the other (42 us of
// latency).  Here a workgroup owns 16 columns and its 16 thread groups each sum every 16th slab;
// the groups are combined through LDS in a fixed order (deterministic, no atomics).
constexpr size_t DD_SPLIT_MAX = 32768;
__global__ __launch_bounds__(256) void dd_reduce_split_kernel(const float* __restrict__ dd_part,
                                                              int strips, size_t n,
                                                              float* __restrict__ dd) {
  __shared__ float4 red[16][16];
  const size_t n4 = n / 4;
  const float4* p4 = reinterpret_cast<const float4*>(dd_part);
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + cl;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    int z = g;
    for (; z + 7 * 16 < strips; z += 8 * 16) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p4[(size_t)(z + u * 16) * n4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < strips; z += 16) {
      const float4 v = p4[(size_t)z * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[g][cl] = s;
  __syncthreads();
  if (g == 0 && i < n4) {
    float4 t = red[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = red[k][cl];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(dd)[i] = t;
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

// The same sums for the slabs of decoder_head3_kernel, dd_part[z][H / 4][R][4] (four consecutive
// h of a row are one 16-byte piece): a thread owns one piece and writes it to the row-major
// dd[r][h]; `split`: 16 thread groups per 16 pieces, every 16th slab each (small batches, as
// dd_reduce_split_kernel).
template <bool SPLIT>
__global__ __launch_bounds__(256) void dd_reduce_q_kernel(const float* __restrict__ dd_part,
                                                          int strips, int R, int H,
                                                          float* __restrict__ dd) {
  __shared__ float4 red[16][16];
  const size_t n4 = (size_t)((H + 3) >> 2) * R;
  const float4* p4 = reinterpret_cast<const float4*>(dd_part);
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  auto emit = [&](size_t i, float4 t) {
    const int hq = (int)(i / R), r = (int)(i % R);
    float* out = dd + (size_t)r * H + 4 * hq;
    if ((H & 3) == 0) {
      *reinterpret_cast<float4*>(out) = t;
    } else {
      out[0] = t.x;
      if (4 * hq + 1 < H) out[1] = t.y;
      if (4 * hq + 2 < H) out[2] = t.z;
      if (4 * hq + 3 < H) out[3] = t.w;
    }
  };
  if (SPLIT) {
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + cl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      int z = g;
      for (; z + 7 * 16 < strips; z += 8 * 16) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p4[(size_t)(z + u * 16) * n4 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; z < strips; z += 16) {
        const float4 v = p4[(size_t)z * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
      float4 t = red[0][cl];
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        const float4 v = red[k][cl];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      emit(i, t);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int z = 0;
      for (; z + 8 <= strips; z += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const f32x4 q = __builtin_nontemporal_load(
              reinterpret_cast<const f32x4*>(p4 + (size_t)(z + u) * n4 + i));
          v[u] = make_float4(q.x, q.y, q.z, q.w);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; z < strips; ++z) {
        const float4 v = p4[(size_t)z * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      emit(i, s);
    }
  }
}

// dd[r][h] = sum over the eight XCD-local accumulators acc[x][h][r] of decoder_head4_kernel's
// atomic store (fixed order: the rounding of the SUM is repeatable, that of the accumulators is
// not).  One thread = one row and four consecutive h.
__global__ __launch_bounds__(256) void dd_reduce_xcd_kernel(const float* __restrict__ acc, int R,
                                                            int H, float* __restrict__ dd) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int HQ = (H + 3) >> 2;
  if (i >= (size_t)HQ * R) return;
  const int hq = (int)(i / R), r = (int)(i % R);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int h = min(4 * hq + e, H - 1);
      v[e] += acc[((size_t)x * H + h) * R + r];
    }
  float* out = dd + (size_t)r * H + 4 * hq;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (4 * hq + e < H) out[e] = v[e];
}

bool decoder_fused_supported(int H) { return H >= 2 && H <= 126 && (H % 2) == 0; }

// the per-strip slabs of dd [strips][H / 4][rows][4] -- or, with SCVAE_HEADS_DD_ATOMICS, the eight
// XCD-local accumulators [8][H][rows] (more than the slabs where F < 8 strips)
static size_t dd_part_floats(size_t strips, int rows, int H) {
  return (strips > 8 ? strips : 8) * (size_t)rows * ((H + 3) / 4 * 4);
}

bool decoder_fused_train_supported(int P, int H, int arith) {
  return decoder_fused_supported(H) || (arith >= 1 && decoder_fused4_supported(P, H));
}

size_t decoder_fused_workspace_floats(int rows, int H, int F, bool train) {
  // (the bf16x9 kernels take 32-gene strips for three heads, and for two beyond H = 110: size for
  //  the narrowest strip)
  const int bn = train ? 32 : DF_BN;
  const size_t strips = (size_t)(F + bn - 1) / bn;
  size_t n = strips * rows;                       // ll_part
  if (train) n += dd_part_floats(strips, rows, H) + 64; // dd_part (16-byte aligned start; H in quads)
  if (train) n += decoder_fused3_workspace_floats(rows, H) + 64;   // bf16 planes of d (bf16x9 kernel)
  // (constrained Poisson passes: a second [strips][rows] array, lse[rows], S[rows])
  if (train) n += strips * (size_t)rows + 2 * (size_t)rows + 192;
  if (train) n += decoder_fused3_rg_slab_floats(H, F) + 64;       // row groups' dW / db slabs
  return n + 64;
}

// 2 (default): two pipelined halves per workgroup (decoder_fused2.hip) where its LDS budget and
// head count allow; 1: one 8-wave workgroup per strip (this file).  SCVAE_DECODER_VARIANT=1
// forces the latter (A/B measurements).
static int g_decoder_variant = 0;
static int decoder_variant() {
  if (g_decoder_variant == 0) {
    const char* e = getenv("SCVAE_DECODER_VARIANT");
    g_decoder_variant = (e && e[0] == '1') ? 1 : 2;
  }
  return g_decoder_variant;
}

int decoder_fused_variant(int P, int H) {
  return (decoder_variant() == 2 && decoder_fused2_supported(P, H)) ? 2 : 1;
}

// Arithmetic of the three products of the fused head kernels (`arith` of every entry below):
// 0 = fp32 MFMA (decoder_fused.hip / decoder_fused2.hip), 1 = the exact nine-term bf16 split
// (decoder_fused3.hip) where that kernel applies (its LDS budget), 2 = the same kernels with the
// three smallest of the nine terms (a2 b3, a3 b2, a3 b3: <= 2^-26 of a product on the rounded split) left out in the
// producer / consumer training kernel -- six matrix instructions per product instead of nine;
// every other launch under 2 runs as under 1.  A plan carries its own
// (scvae_plan_set_head_arith); the default of a new plan and of the stand-alone entry is 1, the
// exact form (every bf16 product exact, fp32 accumulation: an fp32 sum of exact products), or
// what SCVAE_HEAD_ARITH=fp32 / bf16x6 says -- read once, never written again: no mutable process
// state.  2 is an opt-in (its error is that of 1 to the digits tests/test_gpu_as_benched.py
// prints, a third of the matrix instructions gone); bench.py reports it beside the headline.
int default_head_arith() {
  static const int v = [] {
    const char* e = getenv("SCVAE_HEAD_ARITH");
    if (e && (e[0] == 'f' || e[0] == '0')) return 0;
    if (e && (strstr(e, "x6") || e[0] == '2')) return 2;
    return 1;   // default: bf16x9, exact
  }();
  return v;
}
// Accumulation of the decoder gradient dd over the gene strips in the producer / consumer kernel:
// 1 (default) XCD-local fp32 atomics, 0 per-strip slabs + fixed-order reduce (bit-repeatable).
// SCVAE_DD_ACCUMULATION=slabs, read once; a plan carries its own (scvae_plan_set_dd_atomics).
int default_dd_atomics() {
  static const int v = [] {
    const char* e = getenv("SCVAE_DD_ACCUMULATION");
    return (e && (e[0] == 's' || e[0] == '0')) ? 0 : 1;
  }();
  return v;
}
int decoder_train_kernel(int P, int H, int arith) {
  if (arith >= 1 && (decoder_fused3_supported(P, H) || decoder_fused4_supported(P, H))) return 3;
  return decoder_fused_supported(H) ? decoder_fused_variant(P, H) : 0;
}

// Measurement aid (scvae_plan_probe_heads): a pair of HIP events to record around the training
// kernel proper -- after the pre-pass that cuts d into planes, before the reductions -- of the
// next decoder_fused_train call on this host thread.
static thread_local hipEvent_t* g_stage_events = nullptr;
static thread_local unsigned* g_stage_recorded = nullptr;
void stage_probe_arm(hipEvent_t* events, unsigned* recorded) {
  g_stage_events = events;
  g_stage_recorded = recorded;
}
void stage_probe(int stage, int which, hipStream_t s) {
  if (!g_stage_events || stage < 0 || stage >= PS_COUNT) return;
  // (a stage launched more than once per step -- the GMVAE's products with x -- times its first)
  if (g_stage_recorded && (*g_stage_recorded >> (2 * stage + which)) & 1u) return;
  if (hipEventRecord(g_stage_events[2 * stage + which], s) == hipSuccess && g_stage_recorded)
    *g_stage_recorded |= 1u << (2 * stage + which);
}
static thread_local hipEvent_t g_probe[2] = {nullptr, nullptr};
static thread_local bool g_probe_recorded = false;
void decoder_fused_set_probe(hipEvent_t before, hipEvent_t after) {
  g_probe[0] = before;
  g_probe[1] = after;
  g_probe_recorded = false;
}
hipEvent_t decoder_fused_probe(int which) {
  if (which == 1 && g_probe[1]) g_probe_recorded = true;
  return g_probe[which];
}
bool decoder_fused_probe_recorded() { return g_probe_recorded; }

template <bool TRAIN>
static int launch_decoder(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                          int F, Targets t, int B, const float* gw, int inline_lgamma,
                          float* ll_part, float* dd_part, int arith, float* planes = nullptr,
                          const HeadDropout* drop = nullptr, int dd_mode = 0,
                          float* rg_slab = nullptr) {
  const int P = likelihood_heads(kind);
  if (TRAIN && planes && decoder_train_kernel(P, H, arith) == 3) {
    static const int dbg = [] { const char* e = getenv("SCVAE_D3_DEBUG"); return e ? atoi(e) : 0; }();
    return decoder_fused3_launch(s, true, kind, d, rows, H, hp, F, t, B, gw,
                                 inline_lgamma | (dbg << 8), ll_part, dd_part, planes, drop, 0,
                                 nullptr, (dd_mode & 1) | (arith == 2 ? 2 : 0) | (dd_mode & 4),
                                 rg_slab);
  }
  if (drop) {
    set_error("head dropout inside the fused kernel needs the bf16x9 head kernel");
    return -1;
  }
  if (decoder_fused_variant(P, H) == 2)
    return decoder_fused2_launch(s, TRAIN, kind, d, rows, H, hp, F, t, B, gw, inline_lgamma, ll_part,
                                 dd_part);
  const size_t lds = decoder_fused_lds_bytes(P, H, TRAIN);
  const int strips = (F + DF_BN - 1) / DF_BN;
  // e / H == umulhi(e, magic_h) for every e < 2^16 used here (H <= 126)
  const unsigned magic_h = (unsigned)(0x100000000ull / (unsigned)H) + 1u;
#define SCVAE_DF(K_)                                                                              \
  do {                                                                                            \
    auto kfn = decoder_head_kernel<K_, TRAIN, (K_ == LK_ZINB ? 32 : 64)>;                         \
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), \
                                  (int)lds));         \
    hipLaunchKernelGGL(kfn, dim3(strips), dim3(DF_THREADS), lds, s, d, rows, H, magic_h, hp, F, t, \
                       B, gw, inline_lgamma, ll_part, dd_part, df_d_buffers(P, H));               \
  } while (0)
  switch (kind) {
    case LK_POISSON: SCVAE_DF(LK_POISSON); break;
    case LK_NB: SCVAE_DF(LK_NB); break;
    case LK_ZIP: SCVAE_DF(LK_ZIP); break;
    case LK_ZINB: SCVAE_DF(LK_ZINB); break;
    case LK_BERNOULLI: SCVAE_DF(LK_BERNOULLI); break;   // du:194-204; targets binarised by the caller
    default: set_error("unknown likelihood kind %d", kind); return -1;
  }
#undef SCVAE_DF
  SCVAE_LAUNCH_CHECK("decoder_head_kernel");
  return 0;
}

// Forward only (is_training=False / importance-weight pass): ll[rows]
int decoder_fused_forward(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                          int F, Targets t, int B, const float* row_const, float* ll,
                          float* workspace, int arith) {
  const int heads = likelihood_heads(kind);
  // (odd widths and widths beyond 126: the forward half of the producer / consumer kernel)
  const bool wide = !decoder_fused_supported(H);
  SCVAE_ARG(d && t.p && ll && workspace &&
            (!wide || (arith >= 1 && decoder_fused4_supported(heads, H))));
  if (rows == 0) return 0;
  int strips = (F + DF_BN - 1) / DF_BN;
  float* ll_part = workspace;
  // (the data-only term lgamma(1 + t) of the count likelihoods: the caller's row constant, or
  //  evaluated inline; the Bernoulli likelihood has none)
  const int inline_lgamma = (row_const || kind == LK_BERNOULLI) ? 0 : 1;
  // Which kernel (SCVAE_DECODER_FORWARD overrides, for A/B runs):
  //   3  the forward instantiation of the bf16x9 training kernel (decoder_fused3.hip) -- default
  //      for one- and two-head likelihoods under the bf16x9 head arithmetic;
  //   1  the register-resident fp32 forward kernel (decoder_forward.hip) -- default otherwise,
  //      where its LDS budget allows;
  //   0  the forward instantiation of the fp32 training kernels;
  //   4  (odd widths, widths beyond 126, bf16x9 arithmetic) decoder_head4_kernel<.., FWD = true>.
  // The workspace is the one decoder_fused_workspace_floats(.., train = true) sizes (the plans
  // and the C ABI size no other): the bf16 planes of d go behind ll_part.
  static const int forced = [] {
    const char* e = getenv("SCVAE_DECODER_FORWARD");
    return (e && e[0] >= '0' && e[0] <= '4') ? e[0] - '0' : -1;
  }();
  int which = decoder_forward_supported(heads, H) ? 1 : 0;
  if (arith >= 1 && heads <= 2 && decoder_fused3_supported(heads, H)) which = 3;
  if (forced == 0) which = 0;
  if (forced == 1 && decoder_forward_supported(heads, H)) which = 1;
  if (wide) which = 4;
  if (forced == 4 && arith >= 1 && decoder_fused4_supported(heads, H)) which = 4;
  int rc;
  if (which == 4) {
    const int bn = d4_strip_genes(heads, H);
    strips = (F + bn - 1) / bn;
    float* planes = workspace + ((size_t)strips * rows + 63) / 64 * 64;
    rc = decoder_fused3_launch(s, false, kind, d, rows, H, hp, F, t, B, nullptr, inline_lgamma,
                               ll_part, nullptr, planes, nullptr, 0, nullptr, 8);
  } else if (which == 3) {
    const int bn = decoder_fused3_strip_genes(heads);
    strips = (F + bn - 1) / bn;
    float* planes = workspace + ((size_t)strips * rows + 63) / 64 * 64;
    rc = decoder_fused3_launch(s, false, kind, d, rows, H, hp, F, t, B, nullptr, inline_lgamma,
                               ll_part, nullptr, planes);
  } else if (which == 1) {
    rc = decoder_forward_launch(s, kind, d, rows, H, hp, F, t, B, inline_lgamma, ll_part);
  } else {
    rc = launch_decoder<false>(s, kind, d, rows, H, hp, F, t, B, nullptr, inline_lgamma, ll_part,
                               nullptr, arith);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(ll_reduce_kernel, dim3((rows + 15) / 16), dim3(1024), 0, s, ll_part, strips,
                     rows, row_const, B, ll);
  SCVAE_LAUNCH_CHECK("ll_reduce_kernel");
  return 0;
}

// lse[r] = M + log(sum_strips se[strip][r] * exp(m[strip][r] - M)), M = max_strips m[strip][r]
// (fixed order; a strip without a valid gene carries m = -inf, se = 0)
__global__ __launch_bounds__(256) void lse_reduce_kernel(const float* __restrict__ m_part,
                                                         const float* __restrict__ se_part,
                                                         int strips, int R,
                                                         float* __restrict__ lse) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float M = -INFINITY;
  for (int z0 = 0; z0 < strips; z0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = m_part[(size_t)min(z0 + u, strips - 1) * R + r];
#pragma unroll
    for (int u = 0; u < 8; ++u) M = fmaxf(M, v[u]);
  }
  float se = 0.f;
  for (int z0 = 0; z0 < strips; z0 += 8) {
    float v[8], e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = (size_t)min(z0 + u, strips - 1) * R + r;
      v[u] = m_part[i];
      e[u] = se_part[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (z0 + u < strips && v[u] > -INFINITY) se += e[u] * __expf(v[u] - M);
  }
  lse[r] = M + __logf(se);
}

bool decoder_fused_cpoisson_supported(int H, int arith) {
  return decoder_fused_supported(H) && arith >= 1 && decoder_fused3_supported(1, H);
}

// Constrained Poisson through the bf16x9 head kernel: three passes over the strip grid (an
// element's likelihood needs the row's log-sum-exp, its gradient the row sum S; the formulas:
// cpoisson_rows_kernel, elementwise.hip)
int decoder_fused_cpoisson(hipStream_t s, bool train, const float* d, int rows, int H,
                           HeadParams hp, int F, Targets t, int B, const float* gw,
                           const float* count_sum, const float* row_const, float* ll, float* dd,
                           float* workspace) {
  SCVAE_ARG(d && t.p && ll && workspace && count_sum && decoder_fused_cpoisson_supported(H, 1));
  SCVAE_ARG(!train || (gw && dd));
  if (rows == 0) return 0;
  const int strips = (F + DF_BN - 1) / DF_BN;     // (one head: 64-gene strips)
  const size_t n_part = ((size_t)strips * rows + 63) / 64 * 64;
  float* ll_part = workspace;
  float* dd_part = ll_part + n_part;
  float* planes = dd_part + (dd_part_floats(strips, rows, H) + 63) / 64 * 64;
  float* part2 = planes + (decoder_fused3_workspace_floats(rows, H) + 63) / 64 * 64;
  float* lse = part2 + n_part;
  float* S = lse + ((size_t)rows + 63) / 64 * 64;
  CpRows cp;
  cp.count_sum = count_sum;
  cp.out2 = part2;
  const int inline_lgamma = row_const ? 0 : 1;
  int rc;
  // pass 1: row maxima and sums of exponentials per strip -> log-sum-exp of every row
  if ((rc = decoder_fused3_launch(s, false, LK_CPOISSON, d, rows, H, hp, F, t, B, nullptr, 0,
                                  ll_part, nullptr, planes, nullptr, 1, &cp)))
    return rc;
  hipLaunchKernelGGL(lse_reduce_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, ll_part, part2,
                     strips, rows, lse);
  SCVAE_LAUNCH_CHECK("lse_reduce_kernel");
  // pass 2: log-likelihood and S
  cp.lse = lse;
  if ((rc = decoder_fused3_launch(s, false, LK_CPOISSON, d, rows, H, hp, F, t, B, nullptr,
                                  inline_lgamma, ll_part, nullptr, planes, nullptr, 2, &cp)))
    return rc;
  hipLaunchKernelGGL(ll_reduce_kernel, dim3((rows + 15) / 16), dim3(1024), 0, s, ll_part, strips,
                     rows, row_const, B, ll);
  SCVAE_LAUNCH_CHECK("ll_reduce_kernel");
  if (!train) return 0;
  hipLaunchKernelGGL(ll_reduce_kernel, dim3((rows + 15) / 16), dim3(1024), 0, s, part2, strips,
                     rows, (const float*)nullptr, B, S);
  SCVAE_LAUNCH_CHECK("ll_reduce_kernel");
  // pass 3: gradients
  cp.S = S;
  if ((rc = decoder_fused3_launch(s, true, LK_CPOISSON, d, rows, H, hp, F, t, B, gw, 0, ll_part,
                                  dd_part, planes, nullptr, 3, &cp)))
    return rc;
  const size_t n4 = (size_t)((H + 3) / 4) * rows;
  if (n4 <= DD_SPLIT_MAX) {
    hipLaunchKernelGGL(dd_reduce_q_kernel<true>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s,
                       dd_part, strips, rows, H, dd);
  } else {
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dd_reduce_q_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, dd_part,
                       strips, rows, H, dd);
  }
  SCVAE_LAUNCH_CHECK("dd_reduce_q_kernel");
  return 0;
}

// Forward + backward: ll[rows], dW_j, db_j (in hp), dd[rows, H]
int decoder_fused_train(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                        int F, Targets t, int B, const float* gw, const float* row_const,
                        float* ll, float* dd, float* workspace, int arith, bool kernel_only,
                        const HeadDropout* drop, int dd_mode) {
  const int heads = likelihood_heads(kind);
  SCVAE_ARG(d && t.p && gw && ll && dd && workspace &&
            decoder_fused_train_supported(heads, H, arith));
  SCVAE_ARG(!drop || decoder_fused3_supported(heads, H));
  if (rows == 0) return 0;
  const bool head3 = (dd_mode & 4) != 0;     // (forced: decoder_fused_train_cat)
  SCVAE_ARG(!head3 || (decoder_train_kernel(heads, H, arith) == 3 && decoder_fused_supported(H)));
  if (head3) dd_mode &= ~1;                   // (only the producer / consumer kernel has the atomics)
  const int bn = head3 ? decoder_fused3_strip_genes(heads)
                 : decoder_train_kernel(heads, H, arith) == 3
                     ? decoder_fused3_train_strip_genes(heads, H, rows, drop != nullptr, 0)
                     : DF_BN;
  const int strips = (F + bn - 1) / bn;
  float* ll_part = workspace;
  size_t off = ((size_t)strips * rows + 63) / 64 * 64;
  float* dd_part = workspace + off;
  float* planes = dd_part + (dd_part_floats(strips, rows, H) + 63) / 64 * 64;
  // (the last region of the workspace: behind everything the constrained-Poisson passes carve)
  float* rg_slab = workspace + decoder_fused_workspace_floats(rows, H, F, true) -
                   (decoder_fused3_rg_slab_floats(H, F) + 64);
  // (the data-only term lgamma(1 + t): the caller's row constant, or inline; the Bernoulli and
  //  the categorical kinds have none)
  const bool no_lgamma = kind == LK_BERNOULLI || kind == LK_CAT2 || kind == LK_CAT3;
  int rc = launch_decoder<true>(s, kind, d, rows, H, hp, F, t, B, gw,
                                (row_const || no_lgamma) ? 0 : 1, ll_part, dd_part,
                                arith, planes, drop, dd_mode, rg_slab);
  if (rc) return rc;
  if (kernel_only) return 0;  // profiling aid: leave the per-strip partials unreduced
  hipLaunchKernelGGL(ll_reduce_kernel, dim3((rows + 15) / 16), dim3(1024), 0, s, ll_part,
                     strips, rows, row_const, B, ll);
  SCVAE_LAUNCH_CHECK("ll_reduce_kernel");
  const size_t n = (size_t)rows * H;
  stage_probe(PS_DD_REDUCE, 0, s);
  struct EndProbe { hipStream_t s; ~EndProbe() { stage_probe(PS_DD_REDUCE, 1, s); } } end_probe{s};
  if (decoder_train_kernel(likelihood_heads(kind), H, arith) == 3 &&
      decoder_fused3_dd_atomics(kind, H, rows, drop != nullptr, 0, dd_mode)) {
    hipLaunchKernelGGL(dd_reduce_xcd_kernel, dim3((unsigned)(((size_t)((H + 3) / 4) * rows + 255) / 256)),
                       dim3(256), 0, s, dd_part, rows, H, dd);
    SCVAE_LAUNCH_CHECK("dd_reduce_xcd_kernel");
    return 0;
  }
  if (decoder_train_kernel(likelihood_heads(kind), H, arith) == 3) {
    // quad slabs of decoder_head3_kernel
    const size_t n4 = (size_t)((H + 3) / 4) * rows;
    if (n4 <= DD_SPLIT_MAX) {
      hipLaunchKernelGGL(dd_reduce_q_kernel<true>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0,
                         s, dd_part, strips, rows, H, dd);
    } else {
      size_t blocks = (n4 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(dd_reduce_q_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s,
                         dd_part, strips, rows, H, dd);
    }
    SCVAE_LAUNCH_CHECK("dd_reduce_q_kernel");
    return 0;
  }
  if (n % 4 == 0 && n / 4 <= DD_SPLIT_MAX) {
    hipLaunchKernelGGL(dd_reduce_split_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0,
                       s, dd_part, strips, n, dd);
  } else if (n % 4 == 0) {
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

// ---- the piecewise categorical likelihood `-k` on the fused kernels ----
// log p(t) = log softmax(logits)[min(t, k)] + [t >= k] log p_count(t - k)
// (distributions/categorised.py:255-263, va:2507-2532) is a SUM, so the step is two launches of
// decoder_head3_kernel over the same d and the same gw: the count distribution's heads on shifted,
// masked targets (Targets::shift), and the k + 1 class logits of every gene -- columns c,
// c + (k + 1), ... of the P_K head's [H, F (k + 1)] matrix -- as k + 1 heads of a categorical
// kind (LK_CAT2 / LK_CAT3: k = 1, 2; more classes than the kernels have heads stay unfused).
// ll and dd of the two are added.  Nothing [rows, (P + k + 1) F]-sized touches HBM.
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a,
                                                          const float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    a[i] += b[i];
}
bool decoder_fused_cat_supported(int kind, int k_max, int H, int arith) {
  // (the forward half as well: the first pass of an importance-weighted step)
  return (kind == LK_POISSON || kind == LK_NB) && (k_max == 1 || k_max == 2) && arith >= 1 &&
         decoder_fused_supported(H) && decoder_fused3_supported(3, H) &&
         decoder_forward_supported(3, H);
}
size_t decoder_fused_cat_scratch_floats(int rows, int H) {
  return ((size_t)rows + 63) / 64 * 64 + (size_t)rows * H;
}
int decoder_fused_train_cat(hipStream_t s, int kind, int k_max, const float* d, int rows, int H,
                            HeadParams hp, const float* Wk, const float* bk, float* dWk,
                            float* dbk, int F, const float* t, int B, const float* gw, float* ll,
                            float* dd, float* workspace, int arith, float* scratch) {
  SCVAE_ARG(decoder_fused_cat_supported(kind, k_max, H, arith) && Wk && bk && dWk && dbk && scratch);
  if (rows == 0) return 0;
  Targets shifted = targets_f32(t, F);
  shifted.shift = (float)k_max;
  int rc = decoder_fused_train(s, kind, d, rows, H, hp, F, shifted, B, gw, nullptr, ll, dd,
                               workspace, arith, false, nullptr, 4);
  if (rc) return rc;
  HeadParams hc;
  for (int c = 0; c < 3; ++c) {
    const bool on = c <= k_max;
    hc.W[c] = on ? Wk + c : nullptr;
    hc.b[c] = on ? bk + c : nullptr;
    hc.dW[c] = on ? dWk + c : nullptr;
    hc.db[c] = on ? dbk + c : nullptr;
  }
  hc.gene_stride = k_max + 1;
  hc.row_pitch = F * (k_max + 1);
  float* ll2 = scratch;
  float* dd2 = scratch + ((size_t)rows + 63) / 64 * 64;
  rc = decoder_fused_train(s, k_max == 1 ? LK_CAT2 : LK_CAT3, d, rows, H, hc, F,
                           targets_f32(t, F), B, gw, nullptr, ll2, dd2, workspace, arith, false,
                           nullptr, 4);
  if (rc) return rc;
  const size_t n = (size_t)rows * H;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, ll,
                     ll2, (size_t)rows);
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)),
                     dim3(256), 0, s, dd, dd2, n);
  SCVAE_LAUNCH_CHECK("add_inplace_kernel");
  return 0;
}

// ... and its forward half (evaluation passes, the first pass of an importance-weighted step):
// two launches of decoder_forward_kernel (fp32 matrix cores, any head count up to three) -- the
// count heads on shifted, masked targets, the class logits as strided heads -- ll added.
bool decoder_fused_forward_cat_supported(int kind, int k_max, int H) {
  return (kind == LK_POISSON || kind == LK_NB) && (k_max == 1 || k_max == 2) &&
         decoder_forward_supported(3, H);
}
int decoder_fused_forward_cat(hipStream_t s, int kind, int k_max, const float* d, int rows, int H,
                              HeadParams hp, const float* Wk, const float* bk, int F,
                              const float* t, int B, float* ll, float* workspace, float* scratch) {
  SCVAE_ARG(decoder_fused_forward_cat_supported(kind, k_max, H) && Wk && bk && scratch);
  if (rows == 0) return 0;
  const int strips = (F + DF_BN - 1) / DF_BN;
  float* ll_part = workspace;
  Targets shifted = targets_f32(t, F);
  shifted.shift = (float)k_max;
  int rc = decoder_forward_launch(s, kind, d, rows, H, hp, F, shifted, B, 1, ll_part);
  if (rc) return rc;
  hipLaunchKernelGGL(ll_reduce_kernel, dim3((rows + 15) / 16), dim3(1024), 0, s, ll_part, strips,
                     rows, nullptr, B, ll);
  HeadParams hc;
  for (int c = 0; c < 3; ++c) {
    const bool on = c <= k_max;
    hc.W[c] = on ? Wk + c : nullptr;
    hc.b[c] = on ? bk + c : nullptr;
    hc.dW[c] = nullptr;
    hc.db[c] = nullptr;
  }
  hc.gene_stride = k_max + 1;
  hc.row_pitch = F * (k_max + 1);
  rc = decoder_forward_launch(s, k_max == 1 ? LK_CAT2 : LK_CAT3, d, rows, H, hc, F,
                              targets_f32(t, F), B, 0, ll_part);
  if (rc) return rc;
  hipLaunchKernelGGL(ll_reduce_kernel, dim3((rows + 15) / 16), dim3(1024), 0, s, ll_part, strips,
                     rows, nullptr, B, scratch);
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, ll,
                     scratch, (size_t)rows);
  SCVAE_LAUNCH_CHECK("decoder_fused_forward_cat");
  return 0;
}

}  // namespace scvae
