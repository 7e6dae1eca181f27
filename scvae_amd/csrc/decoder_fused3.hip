// Fused decoder output layer on the bf16 matrix cores in the exact-product nine-term form
// ("bf16x9"): the same maths, inputs, outputs and per-strip partial buffers as decoder_fused.hip /
// decoder_fused2.hip (X_TILDE heads va:2466-2489, activation + clip + TFP log_prob du:206-305,
// sum over the genes va:2583-2590, and their backward), with the three products
//
//     GEMM1  pre_j[row, gene] = sum_h d[row, h] W_j[h, gene] + b_j[gene]
//     GEMM2  dW_j[h, gene]   += sum_row d[row, h] G_j[row, gene]         (row h == H: db_j)
//     GEMM3  dd[row, h]       = sum_j sum_gene G_j[row, gene] W_j[h, gene]
//
// evaluated as follows.  An fp32 value is cut EXACTLY into three bf16 terms x = x1 + x2 + x3
// (8 + 8 + 8 significant bits, by truncation: x1 = the upper 16 bits of x, x2 = those of x - x1,
// x3 = the rest); a product x y is the sum of the nine products x_a y_b, each of which is exact
// in fp32 (8 x 8 bits), and the matrix core accumulates them in fp32: the result is an fp32 sum
// of exact products -- the error class of an fp32 FMA chain, not of a bf16 GEMM.  Nine
// v_mfma_f32_*_bf16 per 16 k cost 9 x 32 cycles against 8 x 64 for v_mfma_f32_32x32x2_f32:
// 0.56 of the matrix time of the fp32 kernels.  Reported by bench.py as
// "decoder_head_arith": "bf16x9-exact" with the roofline priced at 2500 / 9 TFLOP/s.
//
// Organisation (one workgroup = 8 waves = one 64-gene strip, walking over 64-row tiles):
//   * d reaches the kernel already cut into bf16 planes, in both operand orientations
//     (split3_hidden_kernel: dA [3][rows][128] for GEMM1, dT [3][128][rows] for GEMM2; column /
//     row H holds ones, so b_j and db_j fall out of the same MFMAs); every operand fragment of d
//     is ONE 16-byte load from L2 straight into the registers that feed the MFMAs -- d never
//     passes through LDS;
//   * the strip's weights are cut once per workgroup and stay in LDS as [head][plane][h][gene]
//     bf16 (144-byte rows: conflict-free ds_read_b128 fragments for GEMM3); GEMM1 reads the same
//     image through ds_read_b64_tr_b16, the hardware transpose read, which hands a lane four
//     consecutive h of its gene;
//   * phase A (all waves): wave (gene block of 16, row half of 32) computes the TRANSPOSED head
//     tile pre_j^T[gene, row] with v_mfma_f32_16x16x32_bf16: a lane then holds four consecutive
//     genes of one row for every head, the likelihood and its gradient run on the accumulator
//     registers (no LDS round trip of the pre-activations), the t > 0 corrections of the
//     negative-binomial kinds as a per-lane walk over the lane's non-zeros; G_j is cut into
//     planes and stored to LDS once, row-major;
//   * phase B (all waves): GEMM3 from LDS (ds_read_b128 of G and W) while the dT fragments of
//     GEMM2 are in flight, then GEMM2 (G through the transpose read); dW accumulators persist
//     in registers over the whole launch;
//   * two workgroup barriers per 64 rows.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"
#include "likelihood.hpp"

namespace scvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4m __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// A/B switch (tools/ab_variants.sh): 1 = the global fragment loads of d stay where the source
// issues them -- one k-step ahead -- (a scheduling barrier that only vector-memory reads may not
// cross); 0 = the compiler sinks them to just before their use to save registers, and the wave
// then waits for an L2 round trip in every k-step
#ifndef D3_PIN_LOADS
#define D3_PIN_LOADS 1
#endif
// how many k-steps ahead the fragments of d are requested (1, or 2: 24 more VGPRs and no faster)
#ifndef D3_AHEAD
#define D3_AHEAD 1
#endif
constexpr int D3_NB = D3_AHEAD + 1;      // fragment buffers
__device__ __forceinline__ void d3_pin_loads() {
  if (D3_PIN_LOADS) __builtin_amdgcn_sched_barrier(0x7C6);   // VMEM reads stay above, MFMAs below; the rest may cross
}

constexpr int D3_THREADS = 512;
constexpr int D3_BM = 64;           // rows per tile
constexpr int D3_KP = 128;          // padded hidden width of the planes of d
// genes per strip (= per workgroup): 64 for one / two heads; 32 for three heads, whose weight
// planes (9 x 112 rows) would not fit LDS next to the G tile at 64
__host__ __device__ constexpr int d3_bn(int P) { return P >= 3 ? 32 : 64; }
// bytes per LDS row: the strip's genes as bf16 + 16 bytes of padding (144 / 80: odd multiples of
// 16 bytes, conflict-free ds_read_b128 fragments)
__host__ __device__ constexpr int d3_rowb(int P) { return 2 * d3_bn(P) + 16; }

__host__ __device__ inline int d3_hp1(int H) { return (H + 1 + 15) / 16 * 16; }

int decoder_fused3_strip_genes(int P) { return d3_bn(P); }
size_t decoder_fused3_lds_bytes(int P, int H) {
  return (size_t)P * 3 * d3_hp1(H) * d3_rowb(P) + (size_t)P * 3 * D3_BM * d3_rowb(P) +
         2 * D3_BM * sizeof(float);
}
bool decoder_fused3_supported(int P, int H) {
  return P <= 3 && H >= 2 && H <= 126 && decoder_fused3_lds_bytes(P, H) <= 160 * 1024;
}
// padded hidden width of the planes of d: [d | 1 | 0 ...] in whole 32-wide contraction steps;
// 128 for every H the all-in-one-phase kernel takes, up to 288 for the producer / consumer kernel
__host__ __device__ inline int d3_kp(int H) { return H + 1 <= D3_KP ? D3_KP : (H + 1 + 31) / 32 * 32; }
// one plane set of d: dA [3][Rpad][KP] then dT [3][KP][Rpad], bf16
__host__ __device__ inline size_t d3_set_elems(int Rpad, int KP = D3_KP) {
  return (size_t)2 * 3 * Rpad * KP;
}
// three plane sets (head dropout: one dropped-out copy of d per head) + the heads' mask words
// [3][Rpad][4]; beyond H = 126 (no dropout instantiation there) one set
size_t decoder_fused3_workspace_floats(int rows, int H) {
  const size_t rpad = (size_t)(rows + D3_BM - 1) / D3_BM * D3_BM;
  const int kp = d3_kp(H);
  if (kp > D3_KP) return d3_set_elems((int)rpad, kp) * sizeof(uint16_t) / sizeof(float) + 64;
  return 3 * d3_set_elems((int)rpad) * sizeof(uint16_t) / sizeof(float) + 3 * rpad * 4 + 64;
}

// x = b1 + b2 + b3 exactly, each term's upper 16 bits a bf16 value (lower 16 bits zero).
// The terms are cut by ROUNDING to nearest (v_cvt_pk_bf16_f32), not by truncation: x - bf16(x)
// has at most 16 significant bits and is exact in fp32, the next residual at most 8 -- the
// three terms still add up to x exactly (the nine-term products stay exact), but they are
// smaller, |b2| <= 2^-9 |x| and |b3| <= 2^-18 |x|, and of either sign: the three products the
// six-term arithmetic leaves out (b2 c3, b3 c2, b3 c3) are then <= 2^-26 |x c| together and
// unbiased -- below the rounding of an fp32 multiply-add -- where truncated terms (all of the
// sign of x, up to 2^-8 and 2^-16) left a one-sided 2^-22 (measured against fp64 at 4096 rows:
// 8e-6 of the largest element of dd, ten times the fp32 matrix cores' error).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {   // RN, lo in bits 0-15
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{lo, hi}, bf16x2));
}
__device__ __forceinline__ void split3_rn(float x, unsigned& b1, unsigned& b2, unsigned& b3) {
  b1 = cvt_pk_bf16(0.f, x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(b1);
  b2 = cvt_pk_bf16(0.f, r1) & 0xFFFF0000u;
  b3 = __float_as_uint(r1 - __uint_as_float(b2));
}
// (upper half of hi) << 16 | upper half of lo
__device__ __forceinline__ unsigned pack_hi16(unsigned lo, unsigned hi) {
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
// the same for two values at once, packed as the planes hold them (x0 in bits 0-15)
__device__ __forceinline__ void split3_rn_pair(float x0, float x1, unsigned& p1, unsigned& p2,
                                               unsigned& p3) {
  p1 = cvt_pk_bf16(x0, x1);
  float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xFFFF0000u);
  p2 = cvt_pk_bf16(r0, r1);
  r0 -= __uint_as_float(p2 << 16);
  r1 -= __uint_as_float(p2 & 0xFFFF0000u);
  p3 = pack_hi16(__float_as_uint(r0), __float_as_uint(r1));      // (8 bits each: exact)
}

// d [R, H] fp32 -> the bf16 planes of [d | 1 | 0...] (column H = 1: bias / db; zero beyond and
// for rows >= R), laid out FRAGMENT-MAJOR, so that the operand fragment of a wave is one
// contiguous KiB (64 lanes x 16 bytes, eight full cache lines):
//   dA[pl][rb][ks][lane][8]   GEMM1's B operand (16x16x32): row 16 rb + (lane & 15),
//                             k = 32 ks + 8 (lane >> 4) + e    (rb < Rpad / 16, ks < KP / 32)
//   dT[pl][ht][kg][lane][8]   GEMM2's A operand (32x32x16): h = 32 ht + (lane & 31),
//                             row 16 kg + 8 (lane >> 5) + e                  (kg < Rpad / 16)
// One thread = one lane slot of a fragment, all three planes.  blockIdx.y: 0 = dA, 1 = dT.
// (zero, zero16: the XCD-local accumulators of dd the training kernel is about to add into, as
//  16-byte pieces -- cleared by this launch's threads instead of a memset launch of its own)
__global__ __launch_bounds__(256) void split3_hidden_kernel(const float* __restrict__ d, int R,
                                                            int H, int Rpad, int KP,
                                                            uint16_t* __restrict__ dA,
                                                            uint16_t* __restrict__ dT,
                                                            u32x4* __restrict__ zero,
                                                            size_t zero16) {
  const int slot = blockIdx.x * 256 + threadIdx.x;       // < Rpad * KP / 8
  for (size_t i = (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero16;
       i += (size_t)gridDim.x * gridDim.y * 256)
    zero[i] = u32x4{0u, 0u, 0u, 0u};
  if (slot >= Rpad * (KP / 8)) return;
  const int ksp = KP / 32;                               // contraction steps per 16-row block
  const int lane = slot & 63, frag = slot >> 6;
  const int nb = Rpad / 16;
  unsigned t1[8], t2[8], t3[8];
  auto value = [&](int r, int k) {
    if (r >= R) return 0.f;
    return k < H ? d[(size_t)r * H + k] : (k == H ? 1.f : 0.f);
  };
  if (blockIdx.y == 0) {
    const int rb = frag / ksp, ks = frag % ksp;
    const int row = 16 * rb + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) split3_rn(value(row, k0 + e), t1[e], t2[e], t3[e]);
  } else {
    const int ht = frag / nb, kg = frag % nb;
    const int h = 32 * ht + (lane & 31), r0 = 16 * kg + 8 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 8; ++e) split3_rn(value(r0 + e, h), t1[e], t2[e], t3[e]);
  }
  auto pack = [](const unsigned* t) {
    u32x4 v;
    v.x = pack_hi16(t[0], t[1]); v.y = pack_hi16(t[2], t[3]);
    v.z = pack_hi16(t[4], t[5]); v.w = pack_hi16(t[6], t[7]);
    return v;
  };
  const size_t plane = (size_t)KP * Rpad;
  uint16_t* dst = (blockIdx.y == 0 ? dA : dT) + (size_t)slot * 8;
  *reinterpret_cast<u32x4*>(dst) = pack(t1);
  *reinterpret_cast<u32x4*>(dst + plane) = pack(t2);
  *reinterpret_cast<u32x4*>(dst + 2 * plane) = pack(t3);
}

// ---- LDS / global operand fragments ----
__device__ __forceinline__ bf16x8 lds_b128(const char* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
// transpose read: the 16 lanes of a group pass the addresses of a [4 rows][16 columns] block
// (lane i: row i >> 2, columns 4 (i & 3) ..) and lane i receives column i of the four rows
__device__ __forceinline__ s16x4 lds_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(p));
}
template <int ROWB>
__device__ __forceinline__ bf16x8 lds_tr8(const char* p) {    // rows 0-3 and rows 4-7
  const s16x4 lo = lds_tr(p), hi = lds_tr(p + 4 * ROWB);
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 global_b128(const uint16_t* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}

// one of N registers by a per-lane index (lane masks of the index bits; inline asm: written as
// C++ selects the compiler turns the tree into a dynamically indexed array = scratch)
struct IndexMasks3 { unsigned long long m[3]; };
__device__ __forceinline__ IndexMasks3 index_masks3(int idx) {
  IndexMasks3 k;
#pragma unroll
  for (int b = 0; b < 3; ++b) k.m[b] = __builtin_amdgcn_ballot_w64((idx >> b) & 1);
  return k;
}
__device__ __forceinline__ float cnd3(float lo, float hi, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(hi), "s"(mask));
  return r;
}
__device__ __forceinline__ float select_n(const float (&v)[8], const IndexMasks3& k) {
  float a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = cnd3(v[2 * i], v[2 * i + 1], k.m[0]);
#pragma unroll
  for (int i = 0; i < 2; ++i) b[i] = cnd3(a[2 * i], a[2 * i + 1], k.m[1]);
  return cnd3(b[0], b[1], k.m[2]);
}
__device__ __forceinline__ float select_n(const float (&v)[4], const IndexMasks3& k) {
  return cnd3(cnd3(v[0], v[1], k.m[0]), cnd3(v[2], v[3], k.m[0]), k.m[1]);
}

// U16 (compile time): the targets are the uint16 minibatch.  As a run-time flag the two load
// paths met in a branch, the compiler waited for the load INSIDE each arm, and the request for the
// next tile's targets -- meant to travel under a whole phase B -- cost a full memory round trip
// (vmcnt(0): everything in flight) per tile.
//
// TRAIN = false: the forward half alone (evaluation passes, the importance-weight pass): GEMM1 +
// likelihood + row sums; no G, no phase B, one barrier per tile (the row-sum buffer alternates
// between two places), the next tile's operands requested under the last k-step of this one.
//
// DROP = true (training only): dropout of the heads' input connections (mu:45-50 inside every
// X_TILDE dense_layer, va:2475-2488): head j reads its OWN dropped-out copy of d -- plane set j,
// d3_set_elems(Rpad) apart -- in GEMM1 and GEMM2 (contraction steps (k-step, head) instead of
// k-steps, the fragments of d two such steps ahead), and its part of dd passes the head's mask
// (drop_bits[j][row][4]: bit h % 32 of word h / 32) times 1 / keep before the heads are summed.
//
// CP > 0 (KIND = LK_CPOISSON only): the constrained Poisson likelihood (du:218-228: lambda =
// clip(softmax over ALL genes), rate = lambda N) in three passes over the strip grid, because an
// element's likelihood needs the row's log-sum-exp and its gradient the row sum S = sum_f gate_f
// (t_f - N lambda_f) (cpoisson_rows_kernel has the formulas):
//   CP = 1 (forward): per strip and row the maximum of the logits -> ll_part, sum exp(a - max)
//          -> cp.out2;
//   CP = 2 (forward, given cp.lse): the strip's part of sum_f log p(t_f) -> ll_part, of S -> cp.out2;
//   CP = 3 (training, given cp.lse and cp.S): G = gw (gate (t - N lambda) - lambda S), phase B.
template <int KIND, int KS1, bool U16, bool TRAIN, bool DROP, int CP>
__global__ __launch_bounds__(D3_THREADS) void decoder_head3_kernel(
    const uint16_t* __restrict__ dA, const uint16_t* __restrict__ dT, int R, int Rpad, int H,
    HeadParams hp, int F, Targets tg, int B, const float* __restrict__ gw, int inline_lgamma,
    float* __restrict__ ll_part, float* __restrict__ dd_part,
    const uint32_t* __restrict__ drop_bits, float inv_keep, CpRows cp) {
  static_assert(TRAIN || !DROP, "dropout is a training-time operation");
  static_assert((CP > 0) == (KIND == LK_CPOISSON), "CP selects the passes of LK_CPOISSON");
  static_assert(CP == 0 || ((CP == 3) == TRAIN && !DROP), "CP 1 / 2 forward, CP 3 training");
  using Traits = LikelihoodTraits<KIND>;
  constexpr int P = Traits::P;
  constexpr int BN = d3_bn(P), ROWB = d3_rowb(P);
  constexpr int GPLANE = D3_BM * ROWB;      // bytes of one [64 rows][BN genes] plane of G
  constexpr int NSB = BN / 32;              // 16-gene blocks of a wave in phase A (2 / 1)
  constexpr int NE = 4 * NSB;               // elements of a lane
  constexpr bool KSPLIT = P >= 3;           // GEMM2 splits the tile's rows between wave pairs
  constexpr int KS2 = KSPLIT ? 2 : 4;       // 16-row k-steps of GEMM2 per wave
  constexpr int KS3 = BN / 16;              // 16-gene k-steps of GEMM3 per head
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HP1 = d3_hp1(H);
  const int WPLANE = HP1 * ROWB;                    // bytes of one [HP1][BN] plane of W
  char* Wl = smem;                                  // [P][3][HP1][BN + 8] bf16
  char* Gl = smem + (size_t)P * 3 * WPLANE;         // [P][3][64][BN + 8] bf16
  float* llbuf = reinterpret_cast<float*>(Gl + (size_t)P * 3 * GPLANE);   // [2][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, i16 = lane & 15, li = lane & 31, kh = lane >> 5;
  const int c0 = blockIdx.x * BN;

  // ---- LDS: zero everything (padding and over-read regions must hold finite values), then the
  //      strip's weights and biases, cut into planes.  ALL of a thread's weight loads -- its rows
  //      of every head -- are requested first, from clamped, always valid addresses, and land
  //      under the zero fill: as a plain loop this fill was one dependent global-memory round trip
  //      per weight row, 12 us per workgroup -- a third of the kernel at a 100-cell minibatch ----
  constexpr int HSTEP = D3_THREADS / BN;
  constexpr int NV = (126 + HSTEP) / HSTEP;            // rows 0 .. H <= 126 of a thread
  {
    const int g = tid & (BN - 1), h0 = tid / BN;
    const bool col_ok = c0 + g < F;
    const int gc = min(c0 + g, F - 1);
    float v[P][NV];
    // (the plain [H, F] layout, or -- the class logits of the P_K head -- genes gene_stride apart
    //  in rows of row_pitch elements)
    const size_t gs = hp.gene_stride ? hp.gene_stride : 1;
    const size_t rp = hp.row_pitch ? hp.row_pitch : F;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const float* wj = hp.W[j] + gc * gs;
      const float* bj = hp.b[j] + gc * gs;
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int h = h0 + u * HSTEP;
        const float* src = h < H ? wj + (size_t)h * rp : bj;
        v[j][u] = *src;
      }
    }
    {
      const int n16 = (int)(((size_t)P * 3 * WPLANE + (size_t)P * 3 * GPLANE + 2 * D3_BM * 4) / 16);
      u32x4* z = reinterpret_cast<u32x4*>(smem);
      for (int i = tid; i < n16; i += D3_THREADS) z[i] = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int h = h0 + u * HSTEP;
        if (h <= H) {
          unsigned b1, b2, b3;
          split3_rn(col_ok ? v[j][u] : 0.f, b1, b2, b3);
          char* dst = Wl + (size_t)(j * 3) * WPLANE + h * ROWB + 2 * g;
          *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(b1 >> 16);
          *reinterpret_cast<uint16_t*>(dst + WPLANE) = (uint16_t)(b2 >> 16);
          *reinterpret_cast<uint16_t*>(dst + 2 * WPLANE) = (uint16_t)(b3 >> 16);
        }
      }
  }
  __syncthreads();

  // ---- wave roles ----
  const int gp = w & 1, rq = w >> 1;        // phase A: genes 16 NSB gp .., rows 16 rq .. of the tile
  const int ht = w & 3, hi2 = w >> 2;       // phase B: h tile; GEMM3: row tile hi2; GEMM2: gene
                                            // tile hi2 (all 64 rows), or -- three heads -- rows
                                            // 32 hi2 .. (all 32 genes)
  const int n_ht3 = (H + 31) / 32, n_ht2 = (H + 1 + 31) / 32;
  const int nb16 = Rpad / 16;               // 16-row blocks of the planes of d

  // per-lane byte offsets
  const int gbase = 16 * NSB * gp;
  const int trw = (8 * q + (i16 >> 2)) * ROWB + 2 * (gbase + 4 * (i16 & 3));        // W, GEMM1
  const int gst = (16 * rq + i16) * ROWB + 2 * (gbase + 4 * q);                     // G store
  const int g3a = (32 * hi2 + li) * ROWB + 16 * kh;                                 // G, GEMM3 A
  const int g3b = (32 * ht + li) * ROWB + 16 * kh;                                  // W, GEMM3 B
  const int g2b = ((KSPLIT ? 32 * hi2 : 0) + 8 * (q >> 1) + (i16 >> 2)) * ROWB +
                  2 * ((KSPLIT ? 0 : 32 * hi2) + 16 * (q & 1) + 4 * (i16 & 3));     // G, GEMM2 B

  f32x16 accW[P];                           // dW tile (h tile ht [, gene tile hi2]) of every head
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) accW[j][i] = 0.f;

  const int n_tiles = (R + D3_BM - 1) / D3_BM;
  const size_t dplane = (size_t)Rpad * D3_KP;

  // targets / upstream of a tile, in flight from the previous phase B (returned by value: an
  // array written through a reference capture ends up in scratch memory)
  struct TileIn { f32x4m t[NSB]; float up0; float cpn, cpl, cps; };
  auto load_t = [&](int m0) {
    TileIn in;
    const int row = m0 + 16 * rq + i16;
    const bool rok = row < R;
    in.up0 = (TRAIN && rok) ? gw[row] : 0.f;
    const int rc = rok ? row : R - 1;
    const int cell = R == B ? rc : rc % B;
    in.cpn = in.cpl = in.cps = 0.f;
    if (CP > 0) in.cpn = cp.count_sum[cell];
    if (CP >= 2) in.cpl = cp.lse[rc];
    if (CP == 3) in.cps = cp.S[rc];
    const size_t trow = (size_t)cell * tg.ld;
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      const int c = c0 + gbase + 16 * sb + 4 * q;
      f32x4m v = {0.f, 0.f, 0.f, 0.f};
      if (U16) {        // pitch % 8 == 0, padding columns zero: one 8-byte load
        const uint16_t* tp = static_cast<const uint16_t*>(tg.p) + trow + c;
        const u32x2 u = *reinterpret_cast<const u32x2*>(tp);
        v.x = __uint_as_float(u.x); v.y = __uint_as_float(u.y);
      } else {
        const float* tp = static_cast<const float*>(tg.p) + trow + c;
        if (c + 3 < F) {
          const f32x4u u = *reinterpret_cast<const f32x4u*>(tp);
          v.x = u.x; v.y = u.y; v.z = u.z; v.w = u.w;
        } else {
          v.x = (c < F) ? tp[0] : 0.f;
          v.y = (c + 1 < F) ? tp[1] : 0.f;
          v.z = (c + 2 < F) ? tp[2] : 0.f;
        }
      }
      in.t[sb] = v;
    }
    return in;
  };
  TileIn nxt = load_t(0);
  // d fragments of GEMM1 (B[k = h][n = row]): 3 planes per k-step, one contiguous KiB each,
  // requested one k-step ahead of the MFMAs that use them (k-step 0 of a tile during the
  // previous phase B)
  const size_t dset = DROP ? d3_set_elems(Rpad) : 0;     // plane set of head j: + j * dset
  auto load_d1 = [&](int m0, int ks, bf16x8 (&dst)[3], int j = 0) {
    const uint16_t* dbase = dA + j * dset + ((size_t)(m0 / 16 + rq) * 4 + ks) * 512 + lane * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) dst[pl] = global_b128(dbase + pl * dplane);
  };
  // (DROP: contraction steps st = k-step * P + head; steps 0 and 1 of a tile travel under the
  //  previous phase B)
  constexpr int NST1 = KS1 * P;
  bf16x8 bfr0[3], bfr1[3];
  load_d1(0, 0, bfr0);
  if (DROP) { if (NST1 > 1) load_d1(0, 1 / P, bfr1, 1 % P); }
  else if (D3_AHEAD > 1 && KS1 > 1) load_d1(0, 1, bfr1);

  for (int tile = 0; tile < n_tiles; ++tile) {
    const int m0 = tile * D3_BM;
    const TileIn cur = nxt;
    const float up = cur.up0;
    // row sums of the tile (forward only: alternating with the unused G area)
    // (not at the start of the G area: GEMM1's last k-step reads up to 16 rows past the last
    //  weight plane -- times zero columns of d, but a row sum's low half can be a bf16 NaN)
    float* lb = (!TRAIN && (tile & 1)) ? reinterpret_cast<float*>(Gl + 4096) : llbuf;
    float* lb2 = reinterpret_cast<float*>(Gl + ((tile & 1) ? 12288 : 8192));   // (CP 1 / 2)
    // =================== phase A: GEMM1 (transposed) + likelihood + G -> LDS ===================
    f32x4m acc1[P][NSB];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) acc1[j][sb] = f32x4m{0.f, 0.f, 0.f, 0.f};
    if constexpr (DROP) {
      // one head per step: its W fragments one step ahead, its d fragments two
      bf16x8 afr[2][NSB][3], bfr[3][3];
      auto load_wj = [&](int st, bf16x8 (&dst)[NSB][3]) {
        const int ks = st / P, j = st % P;
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            dst[sb][pl] = lds_tr8<ROWB>(Wl + (size_t)(j * 3 + pl) * WPLANE + trw + 32 * sb +
                                        32 * ks * ROWB);
      };
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) { bfr[0][pl] = bfr0[pl]; bfr[1][pl] = bfr1[pl]; }
      load_wj(0, afr[0]);
#pragma unroll
      for (int st = 0; st < NST1; ++st) {
        if (st + 2 < NST1) {
          load_d1(m0, (st + 2) / P, bfr[(st + 2) % 3], (st + 2) % P);
          d3_pin_loads();
        }
        if (st + 1 < NST1) load_wj(st + 1, afr[(st + 1) & 1]);
        const int j = st % P;
#pragma unroll
        for (int a = 2; a >= 0; --a)
#pragma unroll
          for (int b = 2; b >= 0; --b)
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
              acc1[j][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  afr[st & 1][sb][a], bfr[st % 3][b], acc1[j][sb], 0, 0, 0);
      }
    } else {
      // W fragments (transpose reads) and d fragments one k-step ahead of the MFMAs
      bf16x8 afr[2][P][NSB][3], bfr[D3_NB][3];
      auto load_w = [&](int ks, bf16x8 (&dst)[P][NSB][3]) {
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
          for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              dst[j][sb][pl] = lds_tr8<ROWB>(Wl + (size_t)(j * 3 + pl) * WPLANE + trw + 32 * sb +
                                             32 * ks * ROWB);
      };
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bfr[0][pl] = bfr0[pl];
      if (D3_AHEAD > 1 && KS1 > 1) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bfr[1][pl] = bfr1[pl];
      }
      load_w(0, afr[0]);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        if (ks + D3_AHEAD < KS1) {
          load_d1(m0, ks + D3_AHEAD, bfr[(ks + D3_AHEAD) % D3_NB]);
          d3_pin_loads();
        }
        if (ks + 1 < KS1) load_w(ks + 1, afr[(ks + 1) & 1]);
        if (!TRAIN && ks == KS1 - 1) {
          // (forward only) the next tile's targets and first d fragments: under this k-step
          // and the likelihood.  Unconditional -- the last tile requests itself again: under
          // a branch the compiler waits for the loads where the arms meet
          const int mn = min(m0 + D3_BM, Rpad - D3_BM);
          nxt = load_t(mn);
          load_d1(mn, 0, bfr0);
          if (D3_AHEAD > 1 && KS1 > 1) load_d1(mn, 1, bfr1);
          d3_pin_loads();
        }
        // small terms first; the accumulators (head x gene block) are independent chains
#pragma unroll
        for (int a = 2; a >= 0; --a)
#pragma unroll
          for (int b = 2; b >= 0; --b)
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
              for (int sb = 0; sb < NSB; ++sb)
                acc1[j][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    afr[ks & 1][j][sb][a], bfr[ks % D3_NB][b], acc1[j][sb], 0, 0, 0);
      }
    }
    // ---- likelihood of this lane's NSB x 4 elements: row 16 rq + i16, genes
    //      16 NSB gp + 16 sb + 4 q + e ----
    float G[P][NE], tval[NE];
    float lsum = 0.f, lsum2 = 0.f;
    unsigned nz = 0;
    if constexpr (CP > 0) {
      // ---- constrained Poisson: this lane's NE logits of ONE row ----
      const float cpn = cur.cpn, cpl = cur.cpl, cps = cur.cps;
      float av[NE];
      bool okv[NE];
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        if (U16) {
          const unsigned v0 = __float_as_uint(cur.t[sb][0]), v1 = __float_as_uint(cur.t[sb][1]);
          tval[4 * sb] = (float)(v0 & 0xFFFFu); tval[4 * sb + 1] = (float)(v0 >> 16);
          tval[4 * sb + 2] = (float)(v1 & 0xFFFFu); tval[4 * sb + 3] = (float)(v1 >> 16);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) tval[4 * sb + e] = cur.t[sb][e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          av[4 * sb + e] = acc1[0][sb][e];
          okv[4 * sb + e] = c0 + gbase + 16 * sb + 4 * q + e < F;
        }
      }
      if (CP == 1) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NE; ++i) mx = fmaxf(mx, okv[i] ? av[i] : -INFINITY);
        float se = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) se += okv[i] ? __expf(av[i] - mx) : 0.f;
        float m2 = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
        m2 = fmaxf(m2, __shfl_xor(m2, 32, WAVE));
        se = mx > -INFINITY ? se * __expf(mx - m2) : 0.f;
        lsum = m2;
        lsum2 = se;     // (summed over the wave's gene groups below)
      } else {
        const float log_n = __logf(fmaxf(cpn, F32_TINY));
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          const float tv = tval[i];
          const float log_lam = av[i] - cpl;
          const float lam = __expf(log_lam);
          const bool gate = lam >= F32_TINY;
          const float own = gate ? tv - cpn * lam : 0.f;
          if (CP == 2) {
            const float lam_c = gate ? lam : F32_TINY;
            const float log_rate = (gate ? log_lam : LOG_F32_TINY) + log_n;
            lsum += okv[i] ? (tv > 0.f ? tv * log_rate : 0.f) - lam_c * cpn : 0.f;
            lsum2 += okv[i] ? own : 0.f;
            nz |= (okv[i] && tv > 0.f) ? (1u << i) : 0u;
          } else {
            G[0][i] = up * (own - lam * cps);
          }
        }
      }
    } else {
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      if (U16) {
        const unsigned v0 = __float_as_uint(cur.t[sb][0]), v1 = __float_as_uint(cur.t[sb][1]);
        tval[4 * sb] = (float)(v0 & 0xFFFFu); tval[4 * sb + 1] = (float)(v0 >> 16);
        tval[4 * sb + 2] = (float)(v1 & 0xFFFFu); tval[4 * sb + 3] = (float)(v1 >> 16);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) tval[4 * sb + e] = cur.t[sb][e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a[P], g[P], lp, r, rgate;
#pragma unroll
        for (int j = 0; j < P; ++j) a[j] = acc1[j][sb][e];
        // (tg.shift > 0 -- the count part of the piecewise categorical likelihood: the
        //  distribution sees t - shift where t >= shift, nothing elsewhere; 0: every element)
        const bool live = tval[4 * sb + e] >= tg.shift;
        tval[4 * sb + e] = live ? tval[4 * sb + e] - tg.shift : 0.f;
        lik_dense<KIND, TRAIN>(tval[4 * sb + e], a, lp, g, r, rgate);
        const bool ok = live && c0 + gbase + 16 * sb + 4 * q + e < F;
        lsum += ok ? lp : 0.f;
        if (TRAIN) {
#pragma unroll
          for (int j = 0; j < P; ++j) G[j][4 * sb + e] = live ? up * g[j] : 0.f;
        }
        nz |= (ok && tval[4 * sb + e] > 0.f) ? (1u << (4 * sb + e)) : 0u;
      }
    }
    }
    // ---- t > 0: + lgamma(r+t) - lgamma(r) [- lgamma(1+t)], and the digamma term of dlog r:
    //      a per-lane walk over the lane's non-zero elements ----
    if (Traits::HAS_R || (inline_lgamma && (CP == 0 || CP == 2))) {
      float lr[NE];
      if (Traits::HAS_R) {
#pragma unroll
        for (int i = 0; i < NE; ++i) lr[i] = acc1[P - 1][i >> 2][i & 3];
      }
      while (__builtin_amdgcn_ballot_w64(nz != 0) != 0) {
        const bool on = nz != 0;
        const int idx = on ? __builtin_ctz(nz) : 0;
        nz &= nz - 1;
        const IndexMasks3 km = index_masks3(idx);
        const float tt = select_n(tval, km);
        float corr = 0.f;
        if (Traits::HAS_R) {
          const float lrv = select_n(lr, km);
          const float r = __expf(fminf(fmaxf(lrv, -10.f), 10.f));
          const float rgate = (lrv >= -10.f && lrv <= 10.f) ? 1.f : 0.f;
          const bool small = !on || (tt <= 8.f && tt == __builtin_rintf(tt));
          float A, D;
          if (__builtin_amdgcn_ballot_w64(!small) == 0)
            lgamma_digamma_diff_small_wave<TRAIN>(r, on ? tt : 0.f, A, D);
          else
            lgamma_digamma_diff_general<TRAIN>(r, on ? tt : 1.f, A, D);
          corr = A;
          // (zero-inflated: at t > 0 the gradient of the base distribution passes unscaled,
          //  zero_inflated.py:194-199 -- the same insertion)
          if (TRAIN) {
            const float delta = on ? up * rgate * r * D : 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) G[P - 1][e] += (idx == e) ? delta : 0.f;
          }
        }
        if (inline_lgamma) corr -= lgamma1p(tt);
        lsum += on ? corr : 0.f;
      }
    }
    // ---- row sums over this wave's genes -> llbuf[gp][row] ----
    if (CP == 1) {
      // (maximum already common to the wave's gene groups; the sums of exponentials refer to it)
      float se = lsum2;
      se += __shfl_xor(se, 16, WAVE);
      se += __shfl_xor(se, 32, WAVE);
      if (q == 0) {
        lb[gp * D3_BM + 16 * rq + i16] = lsum;
        lb2[gp * D3_BM + 16 * rq + i16] = se;
      }
    } else if (CP != 3) {
      float sm = lsum;
      sm += __shfl_xor(sm, 16, WAVE);
      sm += __shfl_xor(sm, 32, WAVE);
      if (q == 0) lb[gp * D3_BM + 16 * rq + i16] = sm;
      if (CP == 2) {
        float s2 = lsum2;
        s2 += __shfl_xor(s2, 16, WAVE);
        s2 += __shfl_xor(s2, 32, WAVE);
        if (q == 0) lb2[gp * D3_BM + 16 * rq + i16] = s2;
      }
    }
    // ---- G_j -> three bf16 planes, row-major [row][gene], 8 bytes (4 genes) per store ----
    if (TRAIN) {
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        unsigned p1[2], p2[2], p3[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
          split3_rn_pair(G[j][4 * sb + 2 * e], G[j][4 * sb + 2 * e + 1], p1[e], p2[e], p3[e]);
        char* dst = Gl + (size_t)(j * 3) * GPLANE + gst + 32 * sb;
        *reinterpret_cast<u32x2*>(dst) = u32x2{p1[0], p1[1]};
        *reinterpret_cast<u32x2*>(dst + GPLANE) = u32x2{p2[0], p2[1]};
        *reinterpret_cast<u32x2*>(dst + 2 * GPLANE) = u32x2{p3[0], p3[1]};
      }
    }
    lds_barrier();

    // =================== phase B: GEMM3 (LDS operands), then GEMM2 ===================
    // per-row log-likelihood of the strip: the two gene blocks summed in a fixed order
    if (CP == 1) {
      // the two gene halves: common maximum, sums of exponentials rescaled to it
      if (tid < D3_BM && m0 + tid < R) {
        const float ma = lb[tid], mb = lb[D3_BM + tid];
        const float m = fmaxf(ma, mb);
        const float se = (ma > -INFINITY ? lb2[tid] * __expf(ma - m) : 0.f) +
                         (mb > -INFINITY ? lb2[D3_BM + tid] * __expf(mb - m) : 0.f);
        ll_part[(size_t)blockIdx.x * R + m0 + tid] = m;
        cp.out2[(size_t)blockIdx.x * R + m0 + tid] = se;
      }
    } else if (CP != 3) {
      if (tid < D3_BM && m0 + tid < R) {
        ll_part[(size_t)blockIdx.x * R + m0 + tid] = lb[tid] + lb[D3_BM + tid];
        if (CP == 2) cp.out2[(size_t)blockIdx.x * R + m0 + tid] = lb2[tid] + lb2[D3_BM + tid];
      }
    }
    if (!TRAIN) continue;   // (the next tile writes the other row-sum buffer: no second barrier)
    // GEMM2's d fragments (A[i = h][k = row]) come from L2 one k-step ahead; k-step 0 is
    // requested here and lands under GEMM3
    auto load_a2 = [&](int ks, bf16x8 (&dst)[3], int j = 0) {
      const uint16_t* tb = dT + j * dset +
          ((size_t)ht * nb16 + m0 / 16 + (KSPLIT ? 2 * hi2 : 0) + ks) * 512 + lane * 8;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[pl] = global_b128(tb + pl * dplane);
    };
    constexpr int NST2 = KS2 * P;              // (DROP: steps (k-step, head), two ahead)
    bf16x8 a2[DROP ? 3 : D3_NB][3];
    if (ht < n_ht2) {
      load_a2(0, a2[0]);
      if (DROP) { if (NST2 > 1) load_a2(1 / P, a2[1], 1 % P); }
      else if (D3_AHEAD > 1 && KS2 > 1) load_a2(1, a2[1]);
    }
    // (DROP) the heads' mask words of this lane's row and h tile, shifted to its four-h groups
    uint32_t mw[P];
    if (DROP) {
#pragma unroll
      for (int j = 0; j < P; ++j)
        mw[j] = ht < n_ht3
                    ? drop_bits[((size_t)j * Rpad + m0 + 32 * hi2 + li) * 4 + ht] >> (4 * kh)
                    : 0u;
    }
    // next tile's targets
    if (tile + 1 < n_tiles) nxt = load_t(m0 + D3_BM);
    if (ht < n_ht3) {
      // ---- GEMM3: dd[row, h] = sum_j sum_gene G_j[row, gene] W_j[h, gene] ----
      f32x16 acc3, accS;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc3[i] = 0.f; accS[i] = 0.f; }
      bf16x8 af[2][3], bf[2][3];
      auto load_3 = [&](int st, bf16x8 (&a)[3], bf16x8 (&b)[3]) {   // step = head * KS3 + k-step
        const int j = st / KS3, ks = st % KS3;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          a[pl] = lds_b128(Gl + (size_t)(j * 3 + pl) * GPLANE + g3a + 32 * ks);
          b[pl] = lds_b128(Wl + (size_t)(j * 3 + pl) * WPLANE + g3b + 32 * ks);
        }
      };
      load_3(0, af[0], bf[0]);
#pragma unroll
      for (int st = 0; st < KS3 * P; ++st) {
        if (st + 1 < KS3 * P) load_3(st + 1, af[(st + 1) & 1], bf[(st + 1) & 1]);
#pragma unroll
        for (int a = 2; a >= 0; --a)
#pragma unroll
          for (int b = 2; b >= 0; --b)
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[st & 1][b], af[st & 1][a], acc3, 0,
                                                           0, 0);
        if (DROP && st % KS3 == KS3 - 1) {
          // head st / KS3 is complete: through its mask (element i = 4 c + e <-> h = 32 ht +
          // 8 c + 4 kh + e), times 1 / keep, into the sum over the heads
          const uint32_t m = mw[st / KS3];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            accS[i] += ((m >> (8 * (i >> 2) + (i & 3))) & 1u) ? acc3[i] * inv_keep : 0.f;
            acc3[i] = 0.f;
          }
        }
      }
      if (DROP) acc3 = accS;
      // (computed transposed, dd^T[h, row]: a lane holds, for each of four groups, FOUR consecutive
      //  h of one row.  The per-strip partial goes to a slab [strip][H / 4][R][4]: one 16-byte
      //  store per group and lane, 32 consecutive rows of an h quad = 512 contiguous bytes per
      //  half wave -- whole lines, a quarter of the store instructions of an [H][R] slab, which
      //  in turn beat the row-major slab with its 100-float rows in partial lines.
      //  Non-temporal: the slabs are read exactly once, by dd_reduce_q_kernel)
      const int row = m0 + 32 * hi2 + li;
      if (row < R) {
        const int HQ = (H + 3) >> 2;
        f32x4m* dst = reinterpret_cast<f32x4m*>(dd_part) +
                      ((size_t)blockIdx.x * HQ + 8 * ht + kh) * R + row;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (4 * (8 * ht + 2 * c + kh) < H)
            __builtin_nontemporal_store(
                f32x4m{acc3[4 * c], acc3[4 * c + 1], acc3[4 * c + 2], acc3[4 * c + 3]},
                dst + (size_t)2 * c * R);
        }
      }
    }
    // the next tile's d fragments of GEMM1: in flight under GEMM2 and the barrier
    if (tile + 1 < n_tiles) {
      load_d1(m0 + D3_BM, 0, bfr0);
      if (DROP) { if (NST1 > 1) load_d1(m0 + D3_BM, 1 / P, bfr1, 1 % P); }
      else if (D3_AHEAD > 1 && KS1 > 1) load_d1(m0 + D3_BM, 1, bfr1);
    }
    if (ht < n_ht2) {
      // ---- GEMM2: dW_j[h, gene] += sum_row d[row, h] G_j[row, gene] ----
      bf16x8 bf[2][3];
      auto load_2 = [&](int st, bf16x8 (&b)[3]) {                    // step = k-step * P + head
        const int ks = st / P, j = st % P;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          b[pl] = lds_tr8<ROWB>(Gl + (size_t)(j * 3 + pl) * GPLANE + g2b + 16 * ks * ROWB);
      };
      load_2(0, bf[0]);
#pragma unroll
      for (int st = 0; st < KS2 * P; ++st) {
        if (st + 1 < KS2 * P) load_2(st + 1, bf[(st + 1) & 1]);
        if (DROP) {
          if (st + 2 < NST2) {
            load_a2((st + 2) / P, a2[(st + 2) % 3], (st + 2) % P);
            d3_pin_loads();
          }
        } else if (st % P == 0 && st / P + D3_AHEAD < KS2) {
          load_a2(st / P + D3_AHEAD, a2[(st / P + D3_AHEAD) % D3_NB]);
          d3_pin_loads();
        }
        const int ks = st / P, j = st % P;
#pragma unroll
        for (int a = 2; a >= 0; --a)
#pragma unroll
          for (int b = 2; b >= 0; --b)
            accW[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[DROP ? st % 3 : ks % D3_NB][a],
                                                              bf[st & 1][b], accW[j], 0, 0, 0);
      }
    }
    lds_barrier();
  }

  if (!TRAIN) return;
  // ---- dW / db of the strip ----
  if (KSPLIT) {
    // the two waves of an h tile hold partial sums over the two row halves: waves 4-7 park
    // theirs in LDS (the weights are no longer needed), waves 0-3 add and write
    float* park = reinterpret_cast<float*>(smem) + (size_t)ht * P * 16 * 64;
    if (hi2 == 1 && ht < n_ht2) {
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) park[(j * 16 + i) * 64 + lane] = accW[j][i];
    }
    __syncthreads();
    if (hi2 == 0 && ht < n_ht2) {
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) accW[j][i] += park[(j * 16 + i) * 64 + lane];
    }
  }
  if (ht < n_ht2 && (!KSPLIT || hi2 == 0)) {
    const int c = c0 + (KSPLIT ? 0 : 32 * hi2) + li;
    if (c < F) {
      const size_t gs_out = hp.gene_stride ? hp.gene_stride : 1;
      const size_t rp_out = hp.row_pitch ? hp.row_pitch : F;
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int h = 32 * ht + (i & 3) + 8 * (i >> 2) + 4 * kh;
          if (h < H) hp.dW[j][(size_t)h * rp_out + c * gs_out] = accW[j][i];
          else if (h == H) hp.db[j][c * gs_out] = accW[j][i];
        }
    }
  }
}


// =================================================================================================
// Round 4: the training step above (TRAIN, no dropout, no constrained-Poisson pass) with the two
// waves of every SIMD in DIFFERENT phases.  tools/probe/coexec_bf16.hip: a wave that keeps the
// bf16 matrix pipe of a SIMD saturated does not slow a VALU wave of the same SIMD (and vice
// versa: both == max), while ONE instruction stream hides only ~4 VALU instructions per
// 32x32x16 MFMA and pays the sum beyond -- and in decoder_head3_kernel both waves of a SIMD run
// the same stream between the same workgroup barriers (GEMM1, then the likelihood's VALU stretch,
// then GEMM3 / GEMM2): matrix busy 45 %, everything else serialised behind it.
//
// Here the workgroup's eight waves are four PRODUCERS (waves 0-3, one per SIMD) and four CONSUMERS
// (waves 4-7, the other wave of each SIMD) on 32-row tiles:
//   producer, tile t:      GEMM1 (wave = 32 genes x 16 rows, or 16 x 16 for three heads) ->
//                          likelihood + gradient on the accumulators -> non-zero walk -> row sums
//                          -> G_j cut into planes -> LDS buffer t & 1
//   consumer, tile t - 1:  GEMM3 dd^T (h tile of the wave x the tile's 32 rows) and GEMM2 dW
//                          (h tile x all the strip's genes x every head: the accumulators persist)
//                          from LDS buffer (t - 1) & 1
// one workgroup barrier per 32 rows (the old schedule: two per 64).  The consumer's matrix work
// (two of the three products) runs under the producer's VALU stretch; the producers carry a
// raised priority, since their GEMM1 + likelihood chain is the longer of the two.
// Same inputs, outputs, slab layouts and arithmetic as decoder_head3_kernel: the two are
// interchangeable launch by launch (SCVAE_D3_SCHEDULE=3 selects the old one; A/B + tests).
constexpr int D4_BM = 32;           // rows per tile

#ifndef D4_COMPACT
#define D4_COMPACT 1
#endif
// the non-zeros' corrections through a dense queue (below) up to four contraction steps
// (H <= 126); the wide geometries, whose weight planes leave no LDS for the queues, keep the
// per-lane walk
__host__ __device__ constexpr bool d4_compact(int ks1) { return D4_COMPACT && ks1 <= 4; }
// NPW producer waves (4 or 8) + four consumers per workgroup
__host__ __device__ constexpr int d4_threads(int npw) { return (npw + 4) * 64; }
size_t decoder_fused4_lds_bytes(int P, int H, int npw, int bn = 0) {
  const int rowb = 2 * (bn ? bn : d3_bn(P)) + 16;
  return (size_t)P * 3 * d3_hp1(H) * rowb + (size_t)2 * P * 3 * D4_BM * rowb +
         (size_t)2 * (npw / 2) * 4 * D4_BM * sizeof(float) +
         (d4_compact((H + 1 + 31) / 32) ? npw * 512 : 0);   // (the producers' queues)
}

#ifndef D4_PROF
#define D4_PROF 0
#endif
// s_setprio of the two kinds of waves (A/B: scvae_amd/csrc/build_prof.sh with EXTRA=-D...)
#ifndef D4_PRIO_PRODUCER
#define D4_PRIO_PRODUCER 1
#endif
#ifndef D4_PRIO_CONSUMER
#define D4_PRIO_CONSUMER 0
#endif
#if D4_PROF
// probe build: cycles (s_memtime) per section, summed over the tiles, of the waves of block 0
__device__ unsigned long long d4_prof[12 * 8];
#define D4_STAMP(k)                                            \
  do {                                                         \
    __builtin_amdgcn_sched_barrier(0);                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    pacc[k] += t_ - plast;                                     \
    plast = t_;                                                \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)
#define D4_PROF_BEGIN unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = __builtin_amdgcn_s_memtime()
#define D4_PROF_END                                                                  \
  do {                                                                               \
    if (blockIdx.x == 0 && lane == 0)                                                \
      for (int k_ = 0; k_ < 8; ++k_) d4_prof[w * 8 + k_] = pacc[k_];                 \
  } while (0)
#else
#define D4_STAMP(k) do {} while (0)
#define D4_PROF_BEGIN do {} while (0)
#define D4_PROF_END do {} while (0)
#endif

// G1: where the producers' GEMM1 sits relative to the workgroup barrier -- 1: at the END of an
// iteration (for the tile after the one whose likelihood the iteration evaluates), 0: at its
// start, 2 (eight producers): the first producer wave of every SIMD at the start, the second at
// the end.  Measured (4096 x 32 738, kernel + reduces, dd atomics): with four producer waves the
// end is better than the start (ZINB 2.35 -> 2.23 ms, Poisson 0.82 -> 0.81); with eight the start
// beats the end (NB 1.42 against 1.50) and the staggered order beats both: with both GEMM1s at
// the start they and the consumer's GEMM2 all want the matrix pipe in the first half of a tile
// (the section probes: GEMM2 6.0 k cycles for 2.3 k of pipe work) and leave it to GEMM3 alone
// and then idle under the atomic adds in the second.
#ifndef D4_G1_EIGHT
#define D4_G1_EIGHT 0
#endif
// KS1 = ceil((H + 1) / 32) contraction steps of GEMM1 = 32-wide h tiles of GEMM2, up to 9
// (H <= 256): beyond four, a consumer wave owns (KS1 + 3) / 4 h tiles (ht, ht + 4, ht + 8) and the
// producers refill their four fragment slots of d inside the loop.  BN_: genes per strip (0: the
// default of the head count; 32 for two heads once the weight planes of 64 genes no longer fit).
// DBP (H a multiple of 32, from 128): the bias gradients db_j = column sums of G_j are summed by
// the PRODUCERS on their registers (one add per element and tile, a cross-lane reduce at the end)
// instead of falling out of GEMM2's ones row -- which at these widths would be an h tile of its
// own (the fifth at H = 128, the ninth at 256) holding nothing but that row.
// FWD: the forward half alone (is_training = False, the first pass of an importance-weighted
// step) for the decoder widths only this kernel takes -- odd ones and everything beyond 126: the
// producers' GEMM1 + likelihood + row sums; the gradient planes are not formed (nothing reads G:
// the compiler drops its arithmetic), the consumers only add up the row sums.
template <int KIND, int KS1, bool U16, int NPW, int BN_ = 0, bool DBP = false,
          int G1 = (NPW == 4 ? 1 : D4_G1_EIGHT), int TERMS = 9, bool FWD = false>
__global__ __launch_bounds__(d4_threads(NPW)) void decoder_head4_kernel(
    const uint16_t* __restrict__ dA, const uint16_t* __restrict__ dT, int R, int Rpad, int H,
    HeadParams hp, int F, Targets tg, int B, const float* __restrict__ gw, int inline_lgamma,
    float* __restrict__ ll_part, float* __restrict__ dd_part, int dd_atomic, int rg_tiles,
    float* __restrict__ rg_slab) {
  using Traits = LikelihoodTraits<KIND>;
  constexpr int P = Traits::P;
  constexpr int NT = d4_threads(NPW);
  constexpr int BN = BN_ ? BN_ : d3_bn(P), ROWB = 2 * BN + 16;
  constexpr int NT2 = DBP ? KS1 - 1 : KS1;  // 32-wide h tiles of GEMM2
  constexpr int NHT = (NT2 + 3) / 4;        // h tiles of a consumer wave
  static_assert(!DBP || (NPW == 4 && KS1 >= 2), "DBP: four producers");
  constexpr int DF = KS1 < 4 ? KS1 : 4;     // fragment slots of d a producer holds
  constexpr int GPLANE = D4_BM * ROWB;      // bytes of one [32 rows][BN genes] plane of G
  constexpr int GBUF = P * 3 * GPLANE;      // one tile's G: [P][3][32][BN + 8] bf16
  constexpr int NGP = NPW / 2;              // producer waves side by side over the strip's genes
  constexpr int NSB = BN / (16 * NGP);      // 16-gene blocks of a producer wave
  static_assert(NSB >= 1 && NSB * 16 * NGP == BN, "producer waves tile the strip");
  constexpr int NE = 4 * NSB;               // elements of a producer lane
  constexpr int KS3 = BN / 16;              // 16-gene k-steps of GEMM3 per head
  constexpr int NGT = BN / 32;              // 32-gene tiles of GEMM2
  constexpr int LLN = NGP * 4 * D4_BM;      // row-sum partials of a tile: [gene group][q][row]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HP1 = d3_hp1(H);
  const int WPLANE = HP1 * ROWB;                    // bytes of one [HP1][BN] plane of W
  char* Wl = smem;                                  // [P][3][HP1][BN + 8] bf16
  char* Gl = smem + (size_t)P * 3 * WPLANE;         // [2][P][3][32][BN + 8] bf16
  float* llbuf = reinterpret_cast<float*>(Gl + 2 * GBUF);   // [2][LLN]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, i16 = lane & 15, li = lane & 31, kh = lane >> 5;
  const int c0 = blockIdx.x * BN;
  // (probe build only: ablation flags travel in the upper bits of inline_lgamma -- 1 consumers
  //  idle, 2 producers idle, 4 no non-zero walk, 8 no dd stores; tools/d4_probe.sh, d4_prof.py)
  const int dbg = D4_PROF ? inline_lgamma >> 8 : 0;
  inline_lgamma &= 0xFF;

  // ---- LDS: zero fill, then the strip's weights and biases cut into planes (as above) ----
  constexpr int HSTEP = NT / BN;
  constexpr int NV = (32 * KS1 + HSTEP - 1) / HSTEP;    // rows 0 .. H < 32 KS1 of a thread
  {
    const int g = tid & (BN - 1), h0 = tid / BN;
    const bool col_ok = c0 + g < F;
    const int gc = min(c0 + g, F - 1);
    float v[P][NV];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const float* wj = hp.W[j] + gc;
      const float* bj = hp.b[j] + gc;
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int h = min(h0 + u * HSTEP, H);
        const float* src = h < H ? wj + (size_t)h * F : bj;
        v[j][u] = *src;
      }
    }
    {
      const int n16 = (int)(((size_t)P * 3 * WPLANE + 2 * GBUF + 2 * LLN * 4) / 16);
      u32x4* z = reinterpret_cast<u32x4*>(smem);
      for (int i = tid; i < n16; i += NT) z[i] = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int h = h0 + u * HSTEP;
        if (h <= H) {
          unsigned b1, b2, b3;
          split3_rn(col_ok ? v[j][u] : 0.f, b1, b2, b3);
          char* dst = Wl + (size_t)(j * 3) * WPLANE + h * ROWB + 2 * g;
          *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(b1 >> 16);
          *reinterpret_cast<uint16_t*>(dst + WPLANE) = (uint16_t)(b2 >> 16);
          *reinterpret_cast<uint16_t*>(dst + 2 * WPLANE) = (uint16_t)(b3 >> 16);
        }
      }
  }
  __syncthreads();

  // (strip x ROW GROUP: workgroup (x, y) takes the 32-row tiles y * rg_tiles .. of strip x, so
  //  that the launch fills whole rounds of the CUs whatever the gene count -- d4_row_groups.  ll
  //  and dd are per row; the strip's dW / db of row group 0 go to the gradient buffers, those of
  //  the groups behind it to rg_slab [group - 1][P][H + 1][F], summed by d4_rg_combine_kernel)
  const int tile0 = blockIdx.y * rg_tiles;
  const int n_tiles = min((R + D4_BM - 1) / D4_BM, tile0 + rg_tiles);
  const int mfirst = tile0 * D4_BM;
  auto grad_row = [&](int j, int h) -> float* {       // dW_j[h, :] (h < H) or db_j (h == H)
    if (blockIdx.y == 0) return h < H ? hp.dW[j] + (size_t)h * F : hp.db[j];
    return rg_slab + (((size_t)(blockIdx.y - 1) * P + j) * (H + 1) + h) * F;
  };
  const int KP = d3_kp(H), ksp = KP / 32;   // padded width of the planes of d, in elements / steps
  const size_t dplane = (size_t)Rpad * KP;
  const int nb16 = Rpad / 16;

  if (w < NPW) {
    // =========================== producers: GEMM1 + likelihood + G ===========================
    // (their GEMM1 + likelihood chain is the longer of the two; measured: which producers win the
    //  arbitration changes who waits at the barrier, not the tile time)
    __builtin_amdgcn_s_setprio(D4_PRIO_PRODUCER);
    const int gp = w % NGP, rq = w / NGP;     // genes 16 NSB gp .., rows 16 rq .. of the tile
    const int gbase = 16 * NSB * gp;
    const int trw = (8 * q + (i16 >> 2)) * ROWB + 2 * (gbase + 4 * (i16 & 3));        // W, GEMM1
    const int gst = (16 * rq + i16) * ROWB + 2 * (gbase + 4 * q);                     // G store
    struct TileIn { f32x4m t[NSB]; float up0; };
    auto load_t = [&](int m0) {
      TileIn in;
      const int row = m0 + 16 * rq + i16;
      const bool rok = row < R;
      in.up0 = (rok && !FWD) ? gw[row] : 0.f;
      const int rc = rok ? row : R - 1;
      const int cell = R == B ? rc : rc % B;
      const size_t trow = (size_t)cell * tg.ld;
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        const int c = c0 + gbase + 16 * sb + 4 * q;
        f32x4m v = {0.f, 0.f, 0.f, 0.f};
        if (U16) {        // pitch % 8 == 0, padding columns zero: one 8-byte load
          const uint16_t* tp = static_cast<const uint16_t*>(tg.p) + trow + c;
          const u32x2 u = *reinterpret_cast<const u32x2*>(tp);
          v.x = __uint_as_float(u.x); v.y = __uint_as_float(u.y);
        } else {
          const float* tp = static_cast<const float*>(tg.p) + trow + c;
          if (c + 3 < F) {
            const f32x4u u = *reinterpret_cast<const f32x4u*>(tp);
            v.x = u.x; v.y = u.y; v.z = u.z; v.w = u.w;
          } else {
            v.x = (c < F) ? tp[0] : 0.f;
            v.y = (c + 1 < F) ? tp[1] : 0.f;
            v.z = (c + 2 < F) ? tp[2] : 0.f;
          }
        }
        in.t[sb] = v;
      }
      return in;
    };
    // d fragments of GEMM1 (B[k = h][n = row]): 3 planes per k-step, one contiguous KiB each; a
    // whole tile's worth is requested at once, behind the previous tile's GEMM1, and lands under
    // that tile's likelihood
    bf16x8 dfr[DF][3];
    auto load_dk = [&](int m0, int ks, bf16x8 (&dst)[3]) {
      const uint16_t* dbase = dA + ((size_t)(m0 / 16 + rq) * ksp + ks) * 512 + lane * 8;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[pl] = global_b128(dbase + pl * dplane);
    };
    auto load_d = [&](int m0) {       // the first DF steps of a tile
#pragma unroll
      for (int ks = 0; ks < DF; ++ks) load_dk(m0, ks, dfr[ks]);
    };
    // GEMM1 of a tile from the fragments of d in dfr: pre_j^T[gene, row] on the accumulators
    f32x4m acc1[P][NSB];
    auto gemm1 = [&](int mt) {      // (mt: the tile's first row -- steps beyond DF are requested here)
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) acc1[j][sb] = f32x4m{0.f, 0.f, 0.f, 0.f};
      bf16x8 afr[2][P][NSB][3];
      auto load_w = [&](int ks, bf16x8 (&dst)[P][NSB][3]) {
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
          for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              dst[j][sb][pl] = lds_tr8<ROWB>(Wl + (size_t)(j * 3 + pl) * WPLANE + trw + 32 * sb +
                                             32 * ks * ROWB);
      };
      load_w(0, afr[0]);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        if (ks + 1 < KS1) load_w(ks + 1, afr[(ks + 1) & 1]);
        // small terms first; the accumulators (head x gene block) are independent chains
#pragma unroll
        for (int a = 2; a >= 0; --a)
#pragma unroll
          for (int b = 2; b >= 0; --b)
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
              for (int sb = 0; sb < NSB; ++sb)
                if (TERMS == 9 || a + b < 3) acc1[j][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    afr[ks & 1][j][sb][a], dfr[ks % DF][b], acc1[j][sb], 0, 0, 0);
        if (ks + DF < KS1) {        // this step's slot is free: step ks + DF of the same tile
          load_dk(mt, ks + DF, dfr[ks % DF]);
          d3_pin_loads();
        }
      }
    };
    // (g1last) the barrier sits between GEMM1 of a tile and its likelihood: when it releases, the
    // producers are in their VALU stretch and the consumers' GEMM2 finds the matrix pipe free;
    // the producers' GEMM1 of the NEXT tile runs at the end of the iteration, under the
    // consumers' stores (or atomic adds) of dd, which issue no matrix instructions.
    float dbacc[P][NE];                 // (DBP) this lane's part of db_j: its genes, its rows
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int e = 0; e < NE; ++e) dbacc[j][e] = 0.f;
    TileIn nxt = load_t(mfirst);
    load_d(mfirst);
    const bool g1last = G1 == 1 || (G1 == 2 && w >= NPW / 2);    // (wave-uniform)
    if (g1last) {
      gemm1(mfirst);
      load_d(min(mfirst + D4_BM, Rpad - D4_BM));
      d3_pin_loads();
    }
    D4_PROF_BEGIN;
    for (int tile = tile0; tile < n_tiles; ++tile) {
      const int m0 = tile * D4_BM;
      if (dbg & 2) { lds_barrier(); continue; }
      const TileIn cur = nxt;
      const float up = cur.up0;
      char* Gb = Gl + ((tile - tile0) & 1) * GBUF;
      float* lb = llbuf + ((tile - tile0) & 1) * LLN;
      if (!g1last) gemm1(m0);
      {
        // the next tile's targets (and, GEMM1 first, its fragments of d): under the likelihood.
        // Unconditional (the last tile requests a valid tile again): under a branch the compiler
        // waits for the loads where the arms meet
        nxt = load_t(min(m0 + D4_BM, Rpad - D4_BM));
        if (!g1last) load_d(min(m0 + D4_BM, Rpad - D4_BM));
        d3_pin_loads();
      }
      // ---- likelihood of this lane's NSB x 4 elements: row 16 rq + i16, genes
      //      16 NSB gp + 16 sb + 4 q + e ----
      float G[P][NE], tval[NE];
      float lsum = 0.f;
      unsigned nz = 0;
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        if (U16) {
          const unsigned v0 = __float_as_uint(cur.t[sb][0]), v1 = __float_as_uint(cur.t[sb][1]);
          tval[4 * sb] = (float)(v0 & 0xFFFFu); tval[4 * sb + 1] = (float)(v0 >> 16);
          tval[4 * sb + 2] = (float)(v1 & 0xFFFFu); tval[4 * sb + 3] = (float)(v1 >> 16);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) tval[4 * sb + e] = cur.t[sb][e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a[P], g[P], lp, r, rgate;
#pragma unroll
          for (int j = 0; j < P; ++j) a[j] = acc1[j][sb][e];
          lik_dense<KIND, true>(tval[4 * sb + e], a, lp, g, r, rgate);
          const bool ok = c0 + gbase + 16 * sb + 4 * q + e < F;
          lsum += ok ? lp : 0.f;
#pragma unroll
          for (int j = 0; j < P; ++j) G[j][4 * sb + e] = up * g[j];
          nz |= (ok && tval[4 * sb + e] > 0.f) ? (1u << (4 * sb + e)) : 0u;
        }
      }
      D4_STAMP(1);
      if constexpr (d4_compact(KS1)) {
      // ---- t > 0: + lgamma(r+t) - lgamma(r) [- lgamma(1+t)], and the digamma term of dlog r.
      //      5 % of the elements: instead of a per-lane walk (as many passes as the fullest lane
      //      holds non-zeros -- 1.6 on average with a fifth of the lanes busy, and the VALU
      //      instructions of these waves are what the tile time is made of), the wave's non-zeros
      //      are queued densely -- (t, log r) at position [elements e' < e of all lanes][lanes
      //      below] from one ballot per element slot -- corrected in ONE pass of full lanes, and
      //      read back by their owners.  The queue is 512 bytes of LDS of the wave's own (64
      //      entries: one pass per 64 non-zeros).  (Kept in the wave's corner of the G buffer
      //      the tile is about to fill, the kernels whose GEMM1 sits at the end of the
      //      iteration were not repeatable from run to run -- 26-40 of 40 launches differed,
      //      in sporadic elements whose log r came out of GEMM1 wrong -- although no other
      //      wave touches that corner between the two barriers; with the queue in LDS of
      //      its own: 0 of 40.  Not strict aliasing (-fno-strict-aliasing: the same), rarer
      //      with dd through slabs (0-2 of 30), and gone with the queue in the buffer's LAST
      //      plane instead of its first.  Not understood; tools/time_head.py TIME_HEAD_STRESS.) ----
      if ((Traits::HAS_R || inline_lgamma) && !(dbg & 4)) {
        int pos[NE];
        int total = 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(((nz >> e) & 1u) != 0u);
          pos[e] = total + (int)__builtin_amdgcn_mbcnt_hi(
                               (unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          total += __builtin_popcountll(bal);
        }
        char* qb = reinterpret_cast<char*>(llbuf + 2 * LLN) + w * 512;
        auto qaddr = [&](int k) { return qb + 8 * k; };
        for (int q0 = 0; q0 < total; q0 += 64) {
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int k = pos[e] - q0;
            if (((nz >> e) & 1u) && (unsigned)k < 64u) {
              const float lrv = Traits::HAS_R ? acc1[P - 1][e >> 2][e & 3] : 0.f;
              *reinterpret_cast<f32x2*>(qaddr(k)) = f32x2{tval[e], lrv};
            }
          }
          __builtin_amdgcn_wave_barrier();
          const bool on = lane < total - q0;
          f32x2 in = *reinterpret_cast<const f32x2*>(qaddr(lane));
          const float tt = on ? in.x : 1.f;
          float corr = 0.f, rd = 0.f;
          if (Traits::HAS_R) {
            const float lrv = on ? in.y : 0.f;
            const float r = __expf(fminf(fmaxf(lrv, -10.f), 10.f));
            const float rgate = (lrv >= -10.f && lrv <= 10.f) ? 1.f : 0.f;
            const bool small = tt <= 8.f && (U16 || tt == __builtin_rintf(tt));
            float A, D;
            if (__builtin_amdgcn_ballot_w64(!small) == 0)
              lgamma_digamma_diff_small_wave<true>(r, tt, A, D);
            else
              lgamma_digamma_diff_general<true>(r, tt, A, D);
            corr = A;
            rd = rgate * r * D;
          }
          if (inline_lgamma) corr -= lgamma1p(tt);
          if (on) *reinterpret_cast<f32x2*>(qaddr(lane)) = f32x2{corr, rd};
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int k = pos[e] - q0;
            if (((nz >> e) & 1u) && (unsigned)k < 64u) {
              const f32x2 o = *reinterpret_cast<const f32x2*>(qaddr(k));
              lsum += o.x;
              if (Traits::HAS_R) G[P - 1][e] = fmaf(up, o.y, G[P - 1][e]);
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      } else {
      // ---- t > 0: + lgamma(r+t) - lgamma(r) [- lgamma(1+t)], and the digamma term of dlog r:
      //      a per-lane walk over the lane's non-zero elements ----
      if ((Traits::HAS_R || inline_lgamma) && !(dbg & 4)) {
        float lr[NE];
        if (Traits::HAS_R) {
#pragma unroll
          for (int i = 0; i < NE; ++i) lr[i] = acc1[P - 1][i >> 2][i & 3];
        }
        while (__builtin_amdgcn_ballot_w64(nz != 0) != 0) {
          const bool on = nz != 0;
          const int idx = on ? __builtin_ctz(nz) : 0;
          nz &= nz - 1;
          const IndexMasks3 km = index_masks3(idx);
          const float tt = select_n(tval, km);
          float corr = 0.f;
          if (Traits::HAS_R) {
            const float lrv = select_n(lr, km);
            const float r = __expf(fminf(fmaxf(lrv, -10.f), 10.f));
            const float rgate = (lrv >= -10.f && lrv <= 10.f) ? 1.f : 0.f;
            const bool small = !on || (tt <= 8.f && tt == __builtin_rintf(tt));
            float A, D;
            if (__builtin_amdgcn_ballot_w64(!small) == 0)
              lgamma_digamma_diff_small_wave<true>(r, on ? tt : 0.f, A, D);
            else
              lgamma_digamma_diff_general<true>(r, on ? tt : 1.f, A, D);
            corr = A;
            const float delta = on ? up * rgate * r * D : 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) G[P - 1][e] += (idx == e) ? delta : 0.f;
          }
          if (inline_lgamma) corr -= lgamma1p(tt);
          lsum += on ? corr : 0.f;
        }
      }
      }
      D4_STAMP(2);
      if (DBP && !FWD) {
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
          for (int e = 0; e < NE; ++e) dbacc[j][e] += G[j][e];
      }
      // ---- this lane's part of the row sum -> lb[gp][q][row]: the consumers add the parts ----
      lb[(gp * 4 + q) * D4_BM + 16 * rq + i16] = lsum;
      // ---- G_j -> three bf16 planes, row-major [row][gene], 8 bytes (4 genes) per store ----
      if constexpr (!FWD)
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
          unsigned p1[2], p2[2], p3[2];
#pragma unroll
          for (int e = 0; e < 2; ++e)
            split3_rn_pair(G[j][4 * sb + 2 * e], G[j][4 * sb + 2 * e + 1], p1[e], p2[e], p3[e]);
          char* dst = Gb + (size_t)(j * 3) * GPLANE + gst + 32 * sb;
          *reinterpret_cast<u32x2*>(dst) = u32x2{p1[0], p1[1]};
          *reinterpret_cast<u32x2*>(dst + GPLANE) = u32x2{p2[0], p2[1]};
          *reinterpret_cast<u32x2*>(dst + 2 * GPLANE) = u32x2{p3[0], p3[1]};
        }
      D4_STAMP(3);
      // GEMM1 of the next tile (the last iteration: a valid tile again, unused), then the request
      // for the fragments of the tile after it
      if (g1last) {
        gemm1(min(m0 + D4_BM, Rpad - D4_BM));
        load_d(min(m0 + 2 * D4_BM, Rpad - D4_BM));
        d3_pin_loads();
      }
      D4_STAMP(0);
      lds_barrier();
      D4_STAMP(4);
    }
    lds_barrier();     // (the consumers' pass over the last tile)
    D4_PROF_END;
    if (DBP && !FWD) {
      // db_j[gene] = sum over the rows: over the 16 lanes of a q group (the tile's rows of this
      // wave), then over the row blocks rq through LDS (the G tiles are free now), fixed order
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          float v = dbacc[j][e];
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, WAVE);
          dbacc[j][e] = v;
        }
      float* park = reinterpret_cast<float*>(Gl);        // [gp][j][e][q]
      if (rq == 1 && i16 == 0) {
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
          for (int e = 0; e < NE; ++e) park[((gp * P + j) * NE + e) * 4 + q] = dbacc[j][e];
      }
      lds_barrier();     // (every wave of the workgroup: the consumers pass it before their dW)
      if (rq == 0 && i16 == 0) {
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int c = c0 + gbase + 16 * (e >> 2) + 4 * q + (e & 3);
            if (c < F) grad_row(j, H)[c] = dbacc[j][e] + park[((gp * P + j) * NE + e) * 4 + q];
          }
      }
    }
    return;
  }

  // =========================== consumers: GEMM3 (dd) and GEMM2 (dW) ===========================
  __builtin_amdgcn_s_setprio(D4_PRIO_CONSUMER);
  if constexpr (FWD) {
    // forward only: the strip's per-row log-likelihood, the producers' parts in a fixed order
    lds_barrier();       // (the producers' first tile)
    for (int tile = tile0; tile < n_tiles; ++tile) {
      const int m0 = tile * D4_BM;
      const float* lb = llbuf + ((tile - tile0) & 1) * LLN;
      if (w == NPW && lane < D4_BM && m0 + lane < R) {
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < NGP * 4; ++u) sm += lb[u * D4_BM + lane];
        ll_part[(size_t)blockIdx.x * R + m0 + lane] = sm;
      }
      lds_barrier();
    }
    return;
  }
  const int ht = w - NPW;                         // h tile of this wave
  // (dd_atomic) the accumulator copy of the XCD this workgroup actually runs on: its adds are
  // then performed in that XCD's own L2, the only L2 that ever holds lines of that copy --
  // correct whatever the dispatcher's block -> XCD placement is
  unsigned xcc = 0;
  if (dd_atomic) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    // (probe build, timing only -- the sums are wrong: the workgroups of an XCD spread over the
    //  eight copies, a quarter of the adds per line at a time)
    if (dbg & 16) xcc = (blockIdx.x >> 3) & 7u;
  }
  const int n_ht3 = (H + 31) / 32, n_ht2 = DBP ? H / 32 : (H + 1 + 31) / 32;
  // this wave's h tiles: ht, ht + 4, ... (NHT of them; one for H <= 126)
  const int g3a = li * ROWB + 16 * kh;                                               // G, GEMM3
  const int g3b = (32 * ht + li) * ROWB + 16 * kh;                                   // W, GEMM3
  const int g2b = (8 * (q >> 1) + (i16 >> 2)) * ROWB + 2 * (16 * (q & 1) + 4 * (i16 & 3));  // G, GEMM2
  f32x16 accW[NHT][P][NGT];                 // dW tiles (h tile x gene tile) of every head
#pragma unroll
  for (int t = 0; t < NHT; ++t)
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int gt = 0; gt < NGT; ++gt)
#pragma unroll
        for (int i = 0; i < 16; ++i) accW[t][j][gt][i] = 0.f;
  // GEMM2's d fragments (A[i = h][k = row]): one contiguous KiB per plane and 16-row k-step; the
  // two k-steps of (row tile, h tile) are requested while the wave works on the pair before
  constexpr int NA2 = NHT > 1 ? 2 : 1;
  bf16x8 a2[NA2][2][3];
  auto load_a2k = [&](int m0, int t, int ks, bf16x8 (&dst)[3]) {
    // (h tile clamped to the planes' last: a wave without a tile t requests a valid one, unused
    //  -- no branch around the loads, at whose end the compiler would wait for them)
    const int htt = min(ht + 4 * t, ksp - 1);
    const uint16_t* tb = dT + ((size_t)htt * nb16 + m0 / 16 + ks) * 512 + lane * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) dst[pl] = global_b128(tb + pl * dplane);
  };
  auto load_a2 = [&](int m0, int t, bf16x8 (&dst)[2][3]) {
    load_a2k(m0, t, 0, dst[0]);
    load_a2k(m0, t, 1, dst[1]);
  };
  // (across the barrier -- and across GEMM3, where the wave's register need peaks -- only the
  //  first k-step of the next row tile travels; the second is requested when its GEMM2 starts,
  //  half a GEMM2 ahead of its use)
  if (ht < n_ht2) load_a2k(mfirst, 0, 0, a2[0][0]);
  lds_barrier();       // (the producers' first tile)
  D4_PROF_BEGIN;
  for (int tile = tile0; tile < n_tiles; ++tile) {
    const int m0 = tile * D4_BM;
    const char* Gb = Gl + ((tile - tile0) & 1) * GBUF;
    const float* lb = llbuf + ((tile - tile0) & 1) * LLN;
    // per-row log-likelihood of the strip: the producers' parts summed in a fixed order
    if (w == NPW && lane < D4_BM && m0 + lane < R) {
      float sm = 0.f;
#pragma unroll
      for (int u = 0; u < NGP * 4; ++u) sm += lb[u * D4_BM + lane];
      ll_part[(size_t)blockIdx.x * R + m0 + lane] = sm;
    }
    D4_STAMP(0);
    // (GEMM2 first: GEMM3's stores -- or atomic adds -- of this tile's part of dd then sit
    //  right before the barrier and drain under the wait and the next tile's GEMM2)
#pragma unroll
    for (int t = 0; t < NHT; ++t) {
      if (ht + 4 * t < n_ht2 && !(dbg & 1)) {
        // ---- GEMM2: dW_j[h, gene] += sum_row d[row, h] G_j[row, gene] ----
        bf16x8 (&a2t)[2][3] = a2[t % NA2];
        if (t == 0) {
          load_a2k(m0, 0, 1, a2[0][1]);
          d3_pin_loads();
        }
        if (t + 1 < NHT) {
          load_a2(m0, t + 1, a2[(t + 1) % NA2]);
          d3_pin_loads();
        }
        constexpr int NST = 2 * P * NGT;             // step = (k-step * P + head) * NGT + gene tile
        bf16x8 bf[2][3];
        auto load_2 = [&](int st, bf16x8 (&b)[3]) {
          const int gt = st % NGT, j = (st / NGT) % P, ks = st / (NGT * P);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[pl] = lds_tr8<ROWB>(Gb + (size_t)(j * 3 + pl) * GPLANE + g2b + 64 * gt +
                                  16 * ks * ROWB);
        };
        load_2(0, bf[0]);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          if (st + 1 < NST) load_2(st + 1, bf[(st + 1) & 1]);
          const int gt = st % NGT, j = (st / NGT) % P, ks = st / (NGT * P);
#pragma unroll
          for (int a = 2; a >= 0; --a)
#pragma unroll
            for (int b = 2; b >= 0; --b)
              if (TERMS == 9 || a + b < 3) accW[t][j][gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2t[ks][a], bf[st & 1][b],
                                                                       accW[t][j][gt], 0, 0, 0);
        }
      }
    }
    if (ht < n_ht2 && !(dbg & 1)) {
      // the next row tile's fragments of this wave's first h tile (the last tile: its own
      // again), in flight over the barrier
      load_a2k(min(m0 + D4_BM, Rpad - D4_BM), 0, 0, a2[0][0]);
      d3_pin_loads();
    }
    D4_STAMP(2);
#pragma unroll
    for (int t = 0; t < NHT; ++t) {
      if (ht + 4 * t < n_ht3 && !(dbg & 1)) {
        // ---- GEMM3: dd^T[h, row] = sum_j sum_gene W_j[h, gene] G_j[row, gene] ----
        const int htt = ht + 4 * t;
        f32x16 acc3;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc3[i] = 0.f;
        bf16x8 af[2][3], bf[2][3];
        auto load_3 = [&](int st, bf16x8 (&a)[3], bf16x8 (&b)[3]) {   // step = head * KS3 + k-step
          const int j = st / KS3, ks = st % KS3;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            a[pl] = lds_b128(Gb + (size_t)(j * 3 + pl) * GPLANE + g3a + 32 * ks);
            b[pl] = lds_b128(Wl + (size_t)(j * 3 + pl) * WPLANE + g3b + 128 * t * ROWB + 32 * ks);
          }
        };
        load_3(0, af[0], bf[0]);
#pragma unroll
        for (int st = 0; st < KS3 * P; ++st) {
          if (st + 1 < KS3 * P) load_3(st + 1, af[(st + 1) & 1], bf[(st + 1) & 1]);
#pragma unroll
          for (int a = 2; a >= 0; --a)
#pragma unroll
            for (int b = 2; b >= 0; --b)
              if (TERMS == 9 || a + b < 3) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[st & 1][b], af[st & 1][a], acc3,
                                                             0, 0, 0);
        }
        D4_STAMP(1);
        const int row = (dbg & 8) ? R : m0 + li;
        if (dd_atomic) {
          // no-return fp32 adds into this XCD's [H][R] accumulator (h-major: the 32 lanes of a
          // half wave add to 128 contiguous bytes); dd_reduce_xcd_kernel sums the eight copies
          if (row < R) {
            typedef __attribute__((address_space(1))) float gfloat;
            float* base = dd_part + ((size_t)xcc * H + 32 * htt + 4 * kh) * R + row;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (32 * htt + 8 * c + 4 * kh + e < H)
                  __builtin_amdgcn_global_atomic_fadd_f32(
                      (gfloat*)(base + (size_t)(8 * c + e) * R), acc3[4 * c + e]);
          }
        } else if (row < R) {
          // slab [strip][H / 4][R][4] (see decoder_head3_kernel): one 16-byte store per h quad
          const int HQ = (H + 3) >> 2;
          f32x4m* dst = reinterpret_cast<f32x4m*>(dd_part) +
                        ((size_t)blockIdx.x * HQ + 8 * htt + kh) * R + row;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (4 * (8 * htt + 2 * c + kh) < H)
              __builtin_nontemporal_store(
                  f32x4m{acc3[4 * c], acc3[4 * c + 1], acc3[4 * c + 2], acc3[4 * c + 3]},
                  dst + (size_t)2 * c * R);
          }
        }
      }
    }
    D4_STAMP(3);
    lds_barrier();
    D4_STAMP(4);
  }
  D4_PROF_END;
  if (DBP) lds_barrier();     // (the producers' exchange of their db parts)
  // ---- dW / db of the strip ----
#pragma unroll
  for (int t = 0; t < NHT; ++t) {
    const int htt = ht + 4 * t;
    if (htt < n_ht2) {
#pragma unroll
      for (int gt = 0; gt < NGT; ++gt) {
        const int c = c0 + 32 * gt + li;
        if (c < F) {
#pragma unroll
          for (int j = 0; j < P; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int h = 32 * htt + (i & 3) + 8 * (i >> 2) + 4 * kh;
              if (h <= H) grad_row(j, h)[c] = accW[t][j][gt][i];
            }
        }
      }
    }
  }
}

#if D4_PROF
extern "C" int scvae_d4_prof_dump(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(d4_prof), sizeof(unsigned long long) * 96);
}
#endif

// which schedule runs a plain training launch: 4 producer / consumer waves (decoder_head4_kernel,
// the default), 3 all waves in one phase (decoder_head3_kernel); read once
// (SCVAE_D3_SCHEDULE=3 / 4 forces one for A/B runs and tests.  Measured at 4096 x 32 738, kernel +
//  reduces: NB 1.49 against 1.58 ms, Poisson 0.90-0.92 against 0.90, ZINB 2.60-2.63 against 2.67)
static int d3_schedule_env() {
  static const int v = [] {
    const char* e = getenv("SCVAE_D3_SCHEDULE");
    return (e && (e[0] == '3' || e[0] == '4')) ? e[0] - '0' : 0;
  }();
  return v;
}
// producer waves per workgroup: eight (16 x 16 blocks, three waves per SIMD) for two heads, four
// for one head (measured, 4096 x 32 738: Poisson 0.81-0.83 ms with four, 0.87-0.88 with eight;
// NB 1.49-1.51 with eight, 1.58-1.61 with four); SCVAE_D4_PRODUCERS=4 / 8 overrides (A/B).
// 32-gene strips (four 16 x 16 blocks per tile) have four.
static int d4_producers_env() {
  static const int v = [] {
    const char* e = getenv("SCVAE_D4_PRODUCERS");
    return (e && (e[0] == '4' || e[0] == '8')) ? e[0] - '0' : 0;
  }();
  return v;
}
// The producer / consumer kernel's geometry for P heads and decoder width H: genes per strip,
// producer waves, contraction steps; ok = it fits (LDS: the strip's weight planes + two G tiles)
struct D4Config { int bn, npw, ks1; size_t lds; bool ok; bool dbp; };
static D4Config d4_config(int P, int H) {
  D4Config c;
  c.ks1 = (H + 1 + 31) / 32;
  c.bn = d3_bn(P);
  c.npw = P >= 3 ? 4 : (d4_producers_env() ? d4_producers_env() : (P == 1 ? 4 : 8));
  c.lds = decoder_fused4_lds_bytes(P, H, c.npw, c.bn);
  if (c.lds > 160 * 1024 && c.bn == 64 && P == 2) {
    // two heads beyond H = 110: the planes of 64 genes no longer fit -- 32-gene strips
    c.bn = 32; c.npw = 4;
    c.lds = decoder_fused4_lds_bytes(P, H, c.npw, c.bn);
  }
  if (c.ks1 > 4 && c.npw == 8) {   // (wide decoders: four producers -- the consumers own 2-3 h tiles)
    c.npw = 4;
    c.lds = decoder_fused4_lds_bytes(P, H, c.npw, c.bn);
  }
  // (H = 128, 160, .. 256: the bias gradient by the producers, GEMM2 without the ones row's tile)
  c.dbp = H >= 128 && H % 32 == 0;
  c.ok = P >= 1 && P <= 3 && H >= 2 && c.ks1 <= (c.dbp ? 9 : 8) && c.lds <= 160 * 1024;
  return c;
}
bool decoder_fused4_supported(int P, int H) { return d4_config(P, H).ok; }
// which schedule runs a plain training launch: 4 producer / consumer waves (decoder_head4_kernel,
// the default and the only one beyond H = 126), 3 all waves in one phase (decoder_head3_kernel)
// Up to 128 rows (the reference's default minibatch of 100) the all-in-one-phase kernel is the
// faster one (4 tiles: 69 against 80 us at 100 x 32 738; 512 rows: the other way round).
static int d3_schedule(int P, int H, int rows) {
  if (!decoder_fused3_supported(P, H)) return 4;
  if (!decoder_fused4_supported(P, H)) return 3;
  return d3_schedule_env() ? d3_schedule_env() : (rows <= 128 ? 3 : 4);
}
int d4_strip_genes(int P, int H) { return d4_config(P, H).bn; }   // the producer / consumer kernel's
// genes per workgroup (= per slab of ll_part / dd_part) of a TRAINING launch
int decoder_fused3_train_strip_genes(int P, int H, int rows, bool drop, int cp_pass) {
  if (!drop && cp_pass == 0 && d3_schedule(P, H, rows) == 4) return d4_config(P, H).bn;
  return d3_bn(P);
}

// whether a training launch with these options accumulates dd with XCD-local atomics (the caller
// then reduces eight [H][rows] copies instead of the per-strip slabs): only the producer /
// consumer kernel has that store
bool decoder_fused3_dd_atomics(int kind, int H, int rows, bool drop, int cp_pass, int dd_mode) {
  const int P = likelihood_heads(kind);
  return (dd_mode & 1) && !(dd_mode & 4) && !drop && cp_pass == 0 && d3_schedule(P, H, rows) == 4;
}

// the training instantiation a plain launch (no dropout, no constrained-Poisson pass) takes, as
// rocprofv3 prints it (bench.py matches its HIP-event timing against the kernel trace by name)
int decoder_fused3_train_kernel_name(int kind, int H, int rows, bool u16, char* out, size_t n,
                                     int terms) {
  const int P = likelihood_heads(kind);
  if (d3_schedule(P, H, rows) == 4) {
    const D4Config c = d4_config(P, H);
    const bool six = terms == 6 && c.ks1 <= 4 && c.bn == d3_bn(P) && !c.dbp;
    return snprintf(out, n, "decoder_head4_kernel<%d, %d, %s, %d, %d, %s, %d, %d, false>", kind, c.ks1,
                    u16 ? "true" : "false", c.npw, c.bn == d3_bn(P) ? 0 : c.bn,
                    c.dbp ? "true" : "false", c.npw == 4 ? 1 : D4_G1_EIGHT, six ? 6 : 9);
  }
  return snprintf(out, n, "decoder_head3_kernel<%d, %d, %s, true, false, 0>", kind,
                  (d3_hp1(H) + 31) / 32, u16 ? "true" : "false");
}

// ---- row groups of the producer / consumer kernel ----
// One workgroup per CU (its LDS), so a launch of `strips` workgroups runs in ceil(strips / CUs)
// rounds and the last round may be nearly empty (27 998 genes on 32-gene strips: 875 workgroups
// on 256 CUs, the fourth round on 107 of them -- a seventh of the launch).  Cutting the rows of
// every strip into n groups multiplies the workgroups; n is chosen for the fullest rounds, a
// group keeps at least 256 rows, and every group beyond the first costs a pass over
// [P][H + 1][F] floats (its dW / db slab), so one more group has to buy at least four percent.
constexpr int D4_MAX_ROW_GROUPS = 4;
size_t decoder_fused3_rg_slab_floats(int H, int F) {
  return (size_t)(D4_MAX_ROW_GROUPS - 1) * 3 * (size_t)(H + 1) * F + 64;
}
static int d4_cu_count() {
  static thread_local int cached_device = -1, cached = 0;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return 256;
  if (device != cached_device) {
    hipDeviceProp_t prop;
    cached = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
    cached_device = device;
  }
  return cached > 0 ? cached : 256;
}
static int d4_row_groups(int strips, int rows) {
  static const int forced = [] {
    const char* e = getenv("SCVAE_D4_ROW_GROUPS");
    return e ? atoi(e) : 0;
  }();
  const int tiles = (rows + D4_BM - 1) / D4_BM;
  int most = tiles / 8;                            // >= 256 rows per group
  if (most > D4_MAX_ROW_GROUPS) most = D4_MAX_ROW_GROUPS;
  if (most < 1) most = 1;
  if (forced >= 1) return forced < most ? forced : most;
  const int cus = d4_cu_count();
  int best = 1;
  double best_score = 0.0;
  for (int n = 1; n <= most; ++n) {
    const long wgs = (long)strips * n;
    const long rounds = (wgs + cus - 1) / cus;
    const double score = (double)wgs / (double)(rounds * cus) - 0.04 * (n - 1);
    if (score > best_score + 1e-9) { best = n; best_score = score; }
  }
  return best;
}
// gradient rows (dW_j[h, :], h < H; db_j, h == H) += the row groups' slabs, in group order
__global__ __launch_bounds__(256) void d4_rg_combine_kernel(HeadParams hp, int P, int H, int F,
                                                            const float* __restrict__ slab,
                                                            int groups) {
  const int jh = blockIdx.y;                       // (head, row)
  const int j = jh / (H + 1), h = jh % (H + 1);
  float* dst = h < H ? hp.dW[j] + (size_t)h * F : hp.db[j];
  const size_t gstride = (size_t)P * (H + 1) * F;
  const float* src = slab + ((size_t)j * (H + 1) + h) * F;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < F; c += gridDim.x * blockDim.x) {
    float v = dst[c];
    for (int g = 0; g < groups; ++g) v += src[(size_t)g * gstride + c];
    dst[c] = v;
  }
}

// ---- launch of the producer / consumer kernel: the instantiation for (kind, steps, strip, waves) ----
struct D4Launch {
  hipStream_t s; const uint16_t* dA; const uint16_t* dT; int rows, Rpad, H; HeadParams hp; int F;
  Targets t; int B; const float* gw; int inline_lgamma; float* ll_part; float* dd_part;
  int dd_atomic; int strips; size_t lds; int terms; int row_groups; float* rg_slab;
  bool fwd = false;      // the forward half alone (decoder_head4_kernel<..., FWD = true>)
};
template <int KIND, int KS1, int NPW, int BN_, bool DBP = false, int TERMS = 9>
static int d4_launch_one(const D4Launch& a) {
  constexpr int G1 = NPW == 4 ? 1 : D4_G1_EIGHT;
  // (six-term products: instantiated for the default strips up to four contraction steps, i.e.
  //  H <= 126; the wider geometries keep all nine terms)
  if constexpr (TERMS == 9 && KS1 <= 4 && BN_ == 0 && !DBP) {
    if (a.terms == 6) return d4_launch_one<KIND, KS1, NPW, BN_, DBP, 6>(a);
  }
  const int tiles = (a.rows + D4_BM - 1) / D4_BM;
  const int rg_tiles = (tiles + a.row_groups - 1) / a.row_groups;
  const int groups = (tiles + rg_tiles - 1) / rg_tiles;     // (no empty group)
  if constexpr (TERMS == 9) {
    if (a.fwd) {
      auto ffn = a.t.u16 ? decoder_head4_kernel<KIND, KS1, true, NPW, BN_, DBP, G1, 9, true>
                         : decoder_head4_kernel<KIND, KS1, false, NPW, BN_, DBP, G1, 9, true>;
      SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(ffn), (int)a.lds));
      hipLaunchKernelGGL(ffn, dim3(a.strips, groups), dim3(d4_threads(NPW)), a.lds, a.s, a.dA,
                         a.dT, a.rows, a.Rpad, a.H, a.hp, a.F, a.t, a.B, a.gw, a.inline_lgamma,
                         a.ll_part, a.dd_part, 0, rg_tiles, nullptr);
      return 0;
    }
  }
  auto kfn = a.t.u16 ? decoder_head4_kernel<KIND, KS1, true, NPW, BN_, DBP, G1, TERMS>
                     : decoder_head4_kernel<KIND, KS1, false, NPW, BN_, DBP, G1, TERMS>;
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)a.lds));
  hipLaunchKernelGGL(kfn, dim3(a.strips, groups), dim3(d4_threads(NPW)), a.lds, a.s, a.dA, a.dT,
                     a.rows, a.Rpad, a.H, a.hp, a.F, a.t, a.B, a.gw, a.inline_lgamma, a.ll_part,
                     a.dd_part, a.dd_atomic, rg_tiles, a.rg_slab);
  if (groups > 1) {
    constexpr int P = likelihood_heads(KIND);
    hipLaunchKernelGGL(d4_rg_combine_kernel, dim3((a.F + 1023) / 1024, P * (a.H + 1)), dim3(256), 0,
                       a.s, a.hp, P, a.H, a.F, a.rg_slab, groups - 1);
  }
  return 0;
}
template <int KIND, int KS1>
static int d4_launch_steps(const D4Launch& a, const D4Config& c) {
  constexpr int P = likelihood_heads(KIND);
  // (the combinations d4_config can return for P heads and KS1 steps)
  if constexpr (KS1 >= 5) {
    if (c.dbp) {
      if constexpr (P == 1) return d4_launch_one<KIND, KS1, 4, 0, true>(a);
      else if constexpr (P == 2) return d4_launch_one<KIND, KS1, 4, 32, true>(a);
      else if constexpr (KS1 == 5) return d4_launch_one<KIND, KS1, 4, 0, true>(a);
    }
  }
  if constexpr (KS1 == 9) {
    set_error("decoder_head4_kernel: nine contraction steps only at H = 256");
    return -1;
  } else if constexpr (P == 1) {
    if constexpr (KS1 <= 4) { if (c.npw == 8) return d4_launch_one<KIND, KS1, 8, 0>(a); }
    return d4_launch_one<KIND, KS1, 4, 0>(a);
  } else if constexpr (P == 2) {
    if constexpr (KS1 <= 4) {
      if (c.bn == 64) return c.npw == 8 ? d4_launch_one<KIND, KS1, 8, 0>(a)
                                        : d4_launch_one<KIND, KS1, 4, 0>(a);
    }
    if constexpr (KS1 >= 4) return d4_launch_one<KIND, KS1, 4, 32>(a);
    set_error("decoder_head4_kernel: no instantiation for %d steps on %d-gene strips", KS1, c.bn);
    return -1;
  } else {
    if constexpr (KS1 <= 5) return d4_launch_one<KIND, KS1, 4, 0>(a);
    set_error("decoder_head4_kernel: three heads beyond H = 159");
    return -1;
  }
}
template <int KIND>
static int d4_launch_kind(const D4Launch& a, const D4Config& c) {
  switch (c.ks1) {
    case 1: return d4_launch_steps<KIND, 1>(a, c);
    case 2: return d4_launch_steps<KIND, 2>(a, c);
    case 3: return d4_launch_steps<KIND, 3>(a, c);
    case 4: return d4_launch_steps<KIND, 4>(a, c);
    case 5: return d4_launch_steps<KIND, 5>(a, c);
    case 6: return d4_launch_steps<KIND, 6>(a, c);
    case 7: return d4_launch_steps<KIND, 7>(a, c);
    case 8: return d4_launch_steps<KIND, 8>(a, c);
    case 9: return d4_launch_steps<KIND, 9>(a, c);
    default: set_error("decoder_head4_kernel: %d contraction steps", c.ks1); return -1;
  }
}

int decoder_fused3_launch(hipStream_t s, bool train, int kind, const float* d, int rows, int H,
                          HeadParams hp, int F, Targets t, int B, const float* gw,
                          int inline_lgamma, float* ll_part, float* dd_part, float* planes,
                          const HeadDropout* drop, int cp_pass, const CpRows* cp, int dd_mode,
                          float* rg_slab) {
  const int P = likelihood_heads(kind);
  // (head4 alone: beyond the all-in-one-phase kernel's LDS budget; forward-only calls also the
  //  widths and head counts that kernel's forward instantiation does not take -- odd widths, three
  //  heads: decoder_fused_forward sends it exactly those)
  // (dd_mode & 8: a forward-only call asks for the producer / consumer kernel's forward half at a
  //  width the all-in-one-phase kernel would take too)
  const bool wide = !decoder_fused3_supported(P, H) ||
                    (!train && cp_pass == 0 &&
                     (P > 2 || !decoder_fused_supported(H) || (dd_mode & 8)));
  // (dd_mode & 4: the all-in-one-phase kernel whatever the row count -- the two launches of the
  //  piecewise categorical likelihood, whose strided heads and shifted targets only it takes)
  SCVAE_ARG(!(dd_mode & 4) || (train && !wide && !drop && cp_pass == 0));
  SCVAE_ARG((hp.gene_stride == 0 && t.shift == 0.f) || (dd_mode & 4));
  SCVAE_ARG(planes && (!wide || (!drop && cp_pass == 0 && decoder_fused4_supported(P, H))));
  SCVAE_ARG(train || !drop);
  SCVAE_ARG((kind == LK_CPOISSON) == (cp_pass >= 1 && cp_pass <= 3 && cp && cp->count_sum));
  SCVAE_ARG(cp_pass == 0 || ((cp_pass == 3) == train && !drop));
  const CpRows cpr = cp ? *cp : CpRows();
  // (forward only: one- and two-head likelihoods; the three-head one, on 32-gene strips, is
  //  no faster than decoder_forward_kernel: 0.92 vs 0.94 ms at 4096 x 32 738, the same step)
  const int Rpad = (rows + D3_BM - 1) / D3_BM * D3_BM;
  const int KP = d3_kp(H);
  uint16_t* dA = reinterpret_cast<uint16_t*>(planes);
  uint16_t* dT = dA + (size_t)3 * Rpad * KP;
  // (head dropout: one plane set per head, cut from that head's dropped-out copy of d, and the
  //  heads' masks as bits behind the three sets)
  uint32_t* bits = reinterpret_cast<uint32_t*>(dA + 3 * d3_set_elems(Rpad));
  // (the producer / consumer kernel with dd through atomics: its eight accumulators [8][H][rows]
  //  are cleared by the plane-cutting launch)
  const bool head4_train = train && !(dd_mode & 4) && !drop && cp_pass == 0 &&
                           d3_schedule(P, H, rows) == 4;
  const size_t acc_bytes = (size_t)8 * H * rows * sizeof(float);
  const bool clear_here = head4_train && (dd_mode & 1) && (acc_bytes & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(dd_part) & 15) == 0;
  for (int j = 0; j < (drop ? P : 1); ++j) {
    hipLaunchKernelGGL(split3_hidden_kernel, dim3((Rpad * (KP / 8) + 255) / 256, train ? 2 : 1),
                       dim3(256), 0, s, drop ? drop->d[j] : d, rows, H, Rpad, KP,
                       dA + j * d3_set_elems(Rpad), dT + j * d3_set_elems(Rpad),
                       clear_here ? reinterpret_cast<u32x4*>(dd_part) : nullptr,
                       clear_here ? acc_bytes / 16 : (size_t)0);
    SCVAE_LAUNCH_CHECK("split3_hidden_kernel");
    if (drop) {
      const int rc = dropout_mask_words(s, bits + (size_t)j * Rpad * 4, rows, Rpad, H, drop->keep,
                                        drop->seed, drop->site[j], drop->map);
      if (rc) return rc;
    }
  }
  const float inv_keep = drop ? 1.f / drop->keep : 1.f;
  const size_t lds = decoder_fused3_lds_bytes(P, H);
  const int strips = (F + d3_bn(P) - 1) / d3_bn(P);
  const int ks1 = (d3_hp1(H) + 31) / 32;
#define SCVAE_D3KC(K_, KS_, T_, D_, C_)                                                          \
  do {                                                                                            \
    auto kfn = t.u16 ? decoder_head3_kernel<K_, KS_, true, T_, D_, C_>                            \
                     : decoder_head3_kernel<K_, KS_, false, T_, D_, C_>;                          \
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), \
                                  (int)lds));         \
    hipLaunchKernelGGL(kfn, dim3(strips), dim3(D3_THREADS), lds, s, dA, dT, rows, Rpad, H, hp, F, \
                       t, B, gw, inline_lgamma, ll_part, dd_part, bits, inv_keep, cpr);           \
  } while (0)
#define SCVAE_D3K(K_, KS_, T_, D_) SCVAE_D3KC(K_, KS_, T_, D_, 0)
#define SCVAE_D3C(T_, C_)                                                                         \
  switch (ks1) {                                                                                  \
    case 1: SCVAE_D3KC(LK_CPOISSON, 1, T_, false, C_); break;                                     \
    case 2: SCVAE_D3KC(LK_CPOISSON, 2, T_, false, C_); break;                                     \
    case 3: SCVAE_D3KC(LK_CPOISSON, 3, T_, false, C_); break;                                     \
    default: SCVAE_D3KC(LK_CPOISSON, 4, T_, false, C_); break;                                    \
  }
#define SCVAE_D3(K_, T_, D_)                                                                      \
  switch (ks1) {                                                                                  \
    case 1: SCVAE_D3K(K_, 1, T_, D_); break;                                                      \
    case 2: SCVAE_D3K(K_, 2, T_, D_); break;                                                      \
    case 3: SCVAE_D3K(K_, 3, T_, D_); break;                                                      \
    default: SCVAE_D3K(K_, 4, T_, D_); break;                                                     \
  }
  if (train && decoder_fused_probe(0)) SCVAE_HIP(hipEventRecord(decoder_fused_probe(0), s));
  if (cp_pass == 1) {
    SCVAE_D3C(false, 1);
  } else if (cp_pass == 2) {
    SCVAE_D3C(false, 2);
  } else if (cp_pass == 3) {
    SCVAE_D3C(true, 3);
  } else if (train && drop) {
    switch (kind) {
      case LK_POISSON: SCVAE_D3(LK_POISSON, true, true); break;
      case LK_NB: SCVAE_D3(LK_NB, true, true); break;
      case LK_ZIP: SCVAE_D3(LK_ZIP, true, true); break;
      case LK_ZINB: SCVAE_D3(LK_ZINB, true, true); break;
      case LK_BERNOULLI: SCVAE_D3(LK_BERNOULLI, true, true); break;
      default: set_error("decoder_head3_kernel: likelihood kind %d", kind); return -1;
    }
  } else if ((train && !(dd_mode & 4) && d3_schedule(P, H, rows) == 4) || (!train && wide)) {
    const D4Config c = d4_config(P, H);
    D4Launch a{s, dA, dT, rows, Rpad, H, hp, F, t, B, gw, inline_lgamma, ll_part, dd_part,
               (dd_mode & 1) ? 1 : 0, (F + c.bn - 1) / c.bn, c.lds, (dd_mode & 2) ? 6 : 9, 1,
               rg_slab};
    a.fwd = !train;
    if (a.fwd) { a.dd_atomic = 0; a.terms = 9; }
    // (no slab from the caller: one group -- the stand-alone forward-only / probe entries; the
    //  forward half leaves no dW and needs none)
    if (rg_slab || a.fwd) a.row_groups = d4_row_groups(a.strips, rows);
    if (a.dd_atomic && !clear_here)   // eight XCD-local accumulators [8][H][rows], cleared for this launch
      SCVAE_HIP(hipMemsetAsync(dd_part, 0, (size_t)8 * H * rows * sizeof(float), s));
    int rc;
    switch (kind) {
      case LK_POISSON: rc = d4_launch_kind<LK_POISSON>(a, c); break;
      case LK_NB: rc = d4_launch_kind<LK_NB>(a, c); break;
      case LK_ZIP: rc = d4_launch_kind<LK_ZIP>(a, c); break;
      case LK_ZINB: rc = d4_launch_kind<LK_ZINB>(a, c); break;
      case LK_BERNOULLI: rc = d4_launch_kind<LK_BERNOULLI>(a, c); break;   // du:194-204; targets binarised by the caller
      default: set_error("decoder_head4_kernel: likelihood kind %d", kind); return -1;
    }
    if (rc) return rc;
  } else if (train) {
    switch (kind) {
      case LK_POISSON: SCVAE_D3(LK_POISSON, true, false); break;
      case LK_NB: SCVAE_D3(LK_NB, true, false); break;
      case LK_ZIP: SCVAE_D3(LK_ZIP, true, false); break;
      case LK_ZINB: SCVAE_D3(LK_ZINB, true, false); break;
      case LK_BERNOULLI: SCVAE_D3(LK_BERNOULLI, true, false); break;   // du:194-204; targets binarised by the caller
      case LK_CAT2: SCVAE_D3(LK_CAT2, true, false); break;   // the class logits of -k (decoder_fused_train_cat)
      case LK_CAT3: SCVAE_D3(LK_CAT3, true, false); break;
      default: set_error("decoder_head3_kernel: likelihood kind %d", kind); return -1;
    }
  } else {
    switch (kind) {
      case LK_POISSON: SCVAE_D3(LK_POISSON, false, false); break;
      case LK_NB: SCVAE_D3(LK_NB, false, false); break;
      case LK_ZIP: SCVAE_D3(LK_ZIP, false, false); break;
      case LK_BERNOULLI: SCVAE_D3(LK_BERNOULLI, false, false); break;
      default: set_error("decoder_head3_kernel (forward): likelihood kind %d", kind); return -1;
    }
  }
#undef SCVAE_D3
#undef SCVAE_D3C
#undef SCVAE_D3KC
#undef SCVAE_D3K
  SCVAE_LAUNCH_CHECK("decoder_head3_kernel");
  if (train && decoder_fused_probe(1)) SCVAE_HIP(hipEventRecord(decoder_fused_probe(1), s));
  return 0;
}

}  // namespace scvae
