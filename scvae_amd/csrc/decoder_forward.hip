// Forward-only decoder output layer: X_TILDE heads + count likelihood + row sum in one kernel,
// for the graph executions with is_training = False (the epoch-end evaluations of train(),
// va:1092-1150 / 1251-1304, evaluate(), va:1969-2055) and the first pass of an importance-
// weighted training step.  Same inputs and per-strip partial buffer as the training kernels
// (decoder_fused.hip), a different organisation, because nothing has to go back to the matrix
// cores after the likelihood:
//
//   * a workgroup owns a 64-gene strip (weights and biases in LDS for the whole kernel); its waves
//     walk over their own 32-row tiles independently -- no workgroup barrier in the loop, and no
//     LDS staging of d: in the k-slot layout chosen below a lane's share of the d tile is one
//     contiguous run of its row, read from L2 with 16-byte loads into the registers that feed the
//     MFMAs (the next tile's run is requested as soon as the last MFMA of the tile is issued);
//   * a wave computes the TRANSPOSED head tile pre_j^T[gene, row] = W_j^T d^T with
//     v_mfma_f32_32x32x2: in the result layout a lane then holds 16 genes of ONE row, so the
//     likelihood of its 16 elements sums into one register, and the row sum of the tile is that
//     register plus the one of lane ^ 32.  The pre-activations never leave the registers;
//   * the bias rides along as an extra contraction step (d gets a ones column, W_j the bias row);
//   * t is read from HBM straight into that register layout (16-byte loads), in flight under
//     the MFMAs;
//   * the t > 0 corrections of the negative-binomial kinds, lgamma(r+t) - lgamma(r), are the
//     expensive part of an element but needed for 5 % of them: each lane walks over its own
//     non-zero elements (bit mask + find-first-set), so the cost follows the largest count of a
//     lane (about 3 of 16) instead of 16.
//
// MFMA work 2 * P * (H + 2) flop per element; algorithmic HBM traffic 4 B per element (t).
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"
#include "likelihood.hpp"

namespace scvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FW_BN = 64;         // genes per workgroup (= the strip width of ll_part)
constexpr int FW_LD = FW_BN + 1;  // odd LDS row stride of the weight strip
constexpr int FW_BM = 32;         // rows per wave tile
constexpr int FW_HMAX = 126;

// contraction length: H, the ones column (bias), zero padding up to a multiple of 8, so that each
// of the two k-slots covers a multiple of 4 positions.  The number of MFMA steps, HK / 2, is a
// template parameter of the kernel (in units of 4 steps): the d operands live in registers, which
// only works with compile-time indices, and a guard per step makes the compiler copy the
// accumulators at every merge point.
__host__ __device__ inline int fw_hk(int H) { return (H + 8) / 8 * 8; }

static size_t fw_lds_bytes(int P, int H) { return (size_t)P * fw_hk(H) * FW_LD * sizeof(float); }
bool decoder_forward_supported(int P, int H) {
  return H >= 2 && H <= FW_HMAX && (H % 2) == 0 && fw_lds_bytes(P, H) <= 160 * 1024;
}

// One of 16 registers by a per-lane index: a binary tree of 15 v_cndmask under the four lane masks
// of the index bits.  (Written with inline asm: as C++ selects the compiler folds the tree into a
// dynamically indexed array, which for a per-lane index means scratch memory.)
struct IndexMasks { unsigned long long m[4]; };
__device__ __forceinline__ IndexMasks index_masks(int idx) {
  IndexMasks k;
#pragma unroll
  for (int b = 0; b < 4; ++b) k.m[b] = __builtin_amdgcn_ballot_w64((idx >> b) & 1);
  return k;
}
__device__ __forceinline__ float cnd(float lo, float hi, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(hi), "s"(mask));
  return r;
}
__device__ __forceinline__ float select16(const float (&v)[16], const IndexMasks& k) {
  float a[8], b[4], c[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = cnd(v[2 * i], v[2 * i + 1], k.m[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = cnd(a[2 * i], a[2 * i + 1], k.m[1]);
#pragma unroll
  for (int i = 0; i < 2; ++i) c[i] = cnd(b[2 * i], b[2 * i + 1], k.m[2]);
  return cnd(c[0], c[1], k.m[3]);
}

// U16 (compile time): the targets are the uint16 minibatch (as a run-time flag the two load paths
// met in a branch and every target load was waited for inside it, before the products started)
template <int KIND, int KS4, bool U16>
__global__ __launch_bounds__(512) void decoder_forward_kernel(
    const float* __restrict__ d, int R, int H, HeadParams hp, int F, Targets tg,
    int B, int inline_lgamma, float* __restrict__ ll_part) {
  using Traits = LikelihoodTraits<KIND>;
  constexpr int P = Traits::P;
  constexpr int NW = 8;
  constexpr int BN = FW_BN, LD = FW_LD, BM = FW_BM;
  extern __shared__ __attribute__((aligned(16))) float Ws[];   // [P][HK][LD]; row H = bias
  constexpr int HS = 4 * KS4;         // MFMA steps; k-slot kh covers positions [kh*HS, kh*HS+HS)
  constexpr int HK = 2 * HS;          // == fw_hk(H)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int c0 = blockIdx.x * BN;

  // ---- strip weights (rows 0..H-1), biases (row H), zeros (rows H+1..HK-1) -> LDS, once ----
  {
    // (all of a thread's loads in flight together, from clamped -- always valid -- addresses: a
    //  loop with one load per trip pays a global-memory round trip per weight row)
    const int c = tid & (BN - 1), p0 = tid >> 6;
    const bool col_ok = c0 + c < F;
    const int cc = min(c0 + c, F - 1);
    constexpr int NV = HK / NW;                        // rows p0 + NW u < HK of this thread
    float v[P][NV];
    // (the plain [H, F] layout, or -- the class logits of the P_K head of `-k` -- genes
    //  gene_stride apart in rows of row_pitch elements: kernels.hpp, HeadParams)
    const size_t gs = hp.gene_stride ? hp.gene_stride : 1;
    const size_t rp = hp.row_pitch ? hp.row_pitch : F;
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int pos = p0 + u * NW;
        v[j][u] = pos < H ? hp.W[j][(size_t)pos * rp + cc * gs] : hp.b[j][cc * gs];
      }
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int pos = p0 + u * NW;
        Ws[((size_t)j * HK + pos) * LD + c] = (col_ok && pos <= H) ? v[j][u] : 0.f;
      }
  }
  __syncthreads();

  // ---- this lane's share of a d tile: row li, positions [kh*HS, kh*HS + HS), as the B operand
  //      of step s in dB[s].  H is even and HS is even: pairs never straddle H. ----
  float dB[HS];
  auto load_d = [&](int m0) {
    const int row = min(m0 + li, R - 1);             // (rows beyond R: results are not stored)
    const float* src = d + (size_t)row * H + kh * HS;
    const int p0 = kh * HS;
    // HS is a multiple of 4 and H is even: a group of 4 positions is inside the row, or holds
    // its last two elements, or starts at the ones column / in the zero padding
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
    for (int q = 0; q < HS / 4; ++q) {
      const int pos = p0 + 4 * q;
      f32x4u v = {0.f, 0.f, 0.f, 0.f};
      if (pos + 3 < H) {
        v = *reinterpret_cast<const f32x4u*>(src + 4 * q);
      } else if (pos + 1 < H) {
        const f32x2u lo = *reinterpret_cast<const f32x2u*>(src + 4 * q);
        v.x = lo.x; v.y = lo.y; v.z = 1.f;       // pos + 2 == H: the ones column
      } else if (pos == H) {
        v.x = 1.f;
      }
      dB[4 * q] = v.x; dB[4 * q + 1] = v.y; dB[4 * q + 2] = v.z; dB[4 * q + 3] = v.w;
    }
  };
  const int n_tiles = (R + BM - 1) / BM;

  load_d(w * BM);
  for (int tile = w; tile < n_tiles; tile += NW) {
    const int m0 = tile * BM;
    const int row = m0 + li;
    const bool row_ok = row < R;
    const size_t trow_off = (size_t)((row_ok ? row : R - 1) % B) * tg.ld;
    float lane_sum = 0.f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      // ---- t of this lane's 16 elements: genes cbase + 8*(i>>2) + (i&3), row li ----
      const int cbase = c0 + cb * 32 + 4 * kh;
      float tv[16];
      if (U16) {           // the uint16 minibatch: four counts per 8-byte load (the pitch covers
                           // whole 64-gene strips, pad columns zero: no bound, no branch, and
                           // nothing touches the loaded value before its use -- a select on it
                           // would end the load's flight)
        const uint16_t* trow = static_cast<const uint16_t*>(tg.p) + trow_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = cbase + 8 * g;
          typedef unsigned u32x2u __attribute__((ext_vector_type(2), aligned(4)));
          const u32x2u v = *reinterpret_cast<const u32x2u*>(trow + c);
          tv[4 * g] = (float)(v.x & 0xFFFFu); tv[4 * g + 1] = (float)(v.x >> 16);
          tv[4 * g + 2] = (float)(v.y & 0xFFFFu); tv[4 * g + 3] = (float)(v.y >> 16);
        }
      } else {
        const float* trow = static_cast<const float*>(tg.p) + trow_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = cbase + 8 * g;
          if (c + 3 < F) {   // one 16-byte load (global loads need only 4-byte alignment)
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            const f32x4u v = *reinterpret_cast<const f32x4u*>(trow + c);
            tv[4 * g] = v.x; tv[4 * g + 1] = v.y; tv[4 * g + 2] = v.z; tv[4 * g + 3] = v.w;
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) tv[4 * g + u] = (c + u < F) ? trow[c + u] : 0.f;
          }
        }
      }
      // ---- pre_j^T[gene, row] = sum_pos Ws_j[pos, gene] * d[row, pos] ----
      f32x16 acc[P];
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
      const float* ap = Ws + (size_t)(kh * HS) * LD + cb * 32 + li;   // A[m = gene][k-slot kh]
#pragma unroll
      for (int s = 0; s < HS; ++s)
#pragma unroll
        for (int j = 0; j < P; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(j * HK + s) * LD], dB[s], acc[j], 0, 0,
                                                        0);
      // the d registers are free again: request this wave's next tile, it arrives during the
      // likelihood math below
      if (cb == 1) load_d(m0 + NW * BM);
      // ---- likelihood of the 16 elements ----
      unsigned nz = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float a[P], lp, g[P], r, rgate;
#pragma unroll
        for (int j = 0; j < P; ++j) a[j] = acc[j][i];
        // (tg.shift > 0 -- the count part of the piecewise categorical likelihood: the
        //  distribution sees t - shift where t >= shift, nothing elsewhere; 0: every element)
        const bool live = tv[i] >= tg.shift;
        tv[i] = live ? tv[i] - tg.shift : 0.f;
        lik_dense<KIND, false>(tv[i], a, lp, g, r, rgate);
        const int c = cbase + 8 * (i >> 2) + (i & 3);
        lane_sum += (live && c < F) ? lp : 0.f;
        nz |= (live && tv[i] > 0.f) ? (1u << i) : 0u;    // (t of a gene beyond F was loaded as 0)
        // four elements at a time: the 16 are independent, interleaving all of them only
        // costs registers
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      // ---- t > 0: + lgamma(r+t) - lgamma(r)  [- lgamma(1+t) unless the caller adds it] ----
      if (Traits::HAS_R || inline_lgamma) {
        while (__builtin_amdgcn_ballot_w64(nz != 0) != 0) {
          const bool on = nz != 0;
          const int idx = on ? __builtin_ctz(nz) : 0;
          nz &= nz - 1;
          const IndexMasks km = index_masks(idx);
          const float tt = select16(tv, km);
          float corr = 0.f;
          if (Traits::HAS_R) {
            // total_count = exp(clip(log_r pre-activation)), the last head (du:266-305)
            float lr[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) lr[i] = acc[P - 1][i];
            const float r = __expf(fminf(fmaxf(select16(lr, km), -10.f), 10.f));
            // one form for the whole wave: the product recurrence if every pending count is a
            // small integer, else the general form (exact for those as well)
            const bool small = !on || (tt <= 8.f && tt == __builtin_rintf(tt));
            float A, D;
            if (__builtin_amdgcn_ballot_w64(!small) == 0)
              lgamma_digamma_diff_small<false>(r, on ? tt : 0.f, A, D);
            else
              lgamma_digamma_diff_general<false>(r, on ? tt : 1.f, A, D);
            corr = A;
          }
          if (inline_lgamma) corr -= lgamma1p(tt);
          lane_sum += on ? corr : 0.f;
        }
      }
    }
    // row sum of the strip: this lane's 32 genes + the 32 of lane ^ 32
    lane_sum += __shfl_xor(lane_sum, 32, 64);
    if (kh == 0 && row_ok) ll_part[(size_t)blockIdx.x * R + row] = lane_sum;
  }
}

int decoder_forward_launch(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                           int F, Targets t, int B, int inline_lgamma, float* ll_part) {
  const int P = likelihood_heads(kind);
  const size_t lds = fw_lds_bytes(P, H);
  const int strips = (F + FW_BN - 1) / FW_BN;
#define SCVAE_FW(K_, KS_)                                                                       \
  case KS_: {                                                                                   \
    auto kfn = t.u16 ? decoder_forward_kernel<K_, KS_, true>                                    \
                     : decoder_forward_kernel<K_, KS_, false>;                                  \
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), \
                                  (int)lds));       \
    hipLaunchKernelGGL(kfn, dim3(strips), dim3(512), lds, s, d, rows, H, hp, F, t, B,           \
                       inline_lgamma, ll_part);                                                 \
  } break
#define SCVAE_FWK(K_)                                                                           \
  switch (fw_hk(H) / 8) {                                                                       \
    SCVAE_FW(K_, 1); SCVAE_FW(K_, 2); SCVAE_FW(K_, 3); SCVAE_FW(K_, 4); SCVAE_FW(K_, 5);        \
    SCVAE_FW(K_, 6); SCVAE_FW(K_, 7); SCVAE_FW(K_, 8); SCVAE_FW(K_, 9); SCVAE_FW(K_, 10);       \
    SCVAE_FW(K_, 11); SCVAE_FW(K_, 12); SCVAE_FW(K_, 13); SCVAE_FW(K_, 14); SCVAE_FW(K_, 15);   \
    SCVAE_FW(K_, 16);                                                                           \
    default: set_error("decoder_forward: hidden size %d", H); return -1;                        \
  }
  switch (kind) {
    case LK_POISSON: SCVAE_FWK(LK_POISSON); break;
    case LK_NB: SCVAE_FWK(LK_NB); break;
    case LK_ZIP: SCVAE_FWK(LK_ZIP); break;
    case LK_ZINB: SCVAE_FWK(LK_ZINB); break;
    case LK_BERNOULLI: SCVAE_FWK(LK_BERNOULLI); break;   // du:194-204; targets binarised by the caller
    case LK_CAT2: SCVAE_FWK(LK_CAT2); break;   // the class logits of -k (decoder_fused_forward_cat)
    case LK_CAT3: SCVAE_FWK(LK_CAT3); break;
    default: set_error("unknown likelihood kind %d", kind); return -1;
  }
#undef SCVAE_FWK
#undef SCVAE_FW
  SCVAE_LAUNCH_CHECK("decoder_forward_kernel");
  return 0;
}

}  // namespace scvae
