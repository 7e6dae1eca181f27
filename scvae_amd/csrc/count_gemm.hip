// The two large products of the encoder's input layer when x is a COUNT matrix
// (scvae/models/utilities.py:53-59 on `x_train[idx].toarray()`, va:997-998):
//
//     forward   a[B,N]  = x[B,F] W[F,N] + b          (MODE 0; contraction over the genes)
//     backward  dW[F,N] = x[B,F]^T dA[B,N]           (MODE 1; contraction over the cells)
//
// x holds integers below 65 536 (UMI counts), i.e. at most 16 significant bits: x = hi + lo with
// hi = the fp32 bit pattern truncated to its upper 16 bits and lo = x - hi, BOTH exactly
// representable in bf16 (lo == 0 for every count below 256, the bulk of a count matrix).  The
// fp32 operand on the other side (W or dA) is split exactly into three bf16 terms
// w = w1 + w2 + w3 (3 x 8 = 24 significant bits, round-to-nearest remainders).  Every product
// hi*w_i / lo*w_i is then exact in the matrix core's fp32 accumulator, so the result is the fp32
// sum of the exact products -- the quality of the fp32 MFMA path -- at 3 (or 6, where a tile
// holds counts >= 256) bf16 MFMAs per 16 k, 16x the fp32 matrix rate: these two GEMMs become
// HBM-bound reads of x instead of MFMA-bound.  Arithmetic type of the path stays fp32 ("dtype"
// f32; bench.py reports "encoder_input_arith": "bf16x3-exact").
//
// The caller guarantees the precondition (DeviceCSR verifies integrality and range once, at
// upload: scvae_csr_check_counts); anything else takes the fp32 MFMA kernels of gemm.hip.
//
// Structure: the x-side operand of both products is cut into hi / lo in registers (two or three
// VALU instructions per pair); the other operand is split and transposed to k-contiguous bf16
// once per launch (split3_transpose_kernel) and reaches the MFMAs through LDS rows padded by 16
// bytes (conflict-free ds_read_b128 fragments).  A wave owns 64 x-side rows (two 32-row tiles of
// v_mfma_f32_32x32x16_bf16) times up to four 32-column tiles.  Forward: the [256, 32] tile of x is
// read with coalesced 16-byte loads and parked in LDS as bf16 (count_gemm_fwd_kernel); weight
// gradient: a lane reads its gene down the cells straight into the operand layout
// (count_gemm_dw_kernel).  Both request x two chunks ahead.  Split-K over the grid, slabs summed
// in a fixed order (deterministic); the K % chunk leftover terms are added there in fp32.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace scvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int CG_NP = 128;            // columns, padded
constexpr int CG_BK = 32;             // k per chunk (two MFMA k-steps)
constexpr int CG_ROW = 80;            // LDS bytes per (term, column) row: 64 + 16 pad
constexpr int CG_TM = 64;             // x-side rows per wave (two 32-row tiles)
constexpr int CG_KPAD = 64;           // the split operand is zero-padded to a multiple of this

// The count matrix arrives as fp32 (any caller) or as uint16 (the minibatch densified for these
// kernels: half the bytes of the HBM-bound reads); a count converts exactly, the cut into hi / lo
// is the same arithmetic on the same fp32 value.
template <typename XT> __device__ __forceinline__ float count_to_f32(XT v) { return (float)v; }

// A/B switch (tools/ab_cg_nt.sh, round 5: refuted): the requests for x marked non-temporal (bit 0:
// forward product, bit 1: weight gradient).  Forward: + 15 us stand-alone, + 30 us in the step (a
// chunk's request takes half of a 128-byte line, the next chunk the other half: the line no longer
// waits in the L2 for it); weight gradient: +- 0.
#ifndef SCVAE_CG_NT
#define SCVAE_CG_NT 0
#endif
template <bool NT, typename T> __device__ __forceinline__ T cg_load(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

static inline int cg_kpad(int K) { return (K + CG_KPAD - 1) / CG_KPAD * CG_KPAD; }
static inline int cg_bk(int mode) { return mode == 0 ? 32 : 16; }   // chunk of the mode's kernel

__device__ __forceinline__ unsigned bf16_rne_bits(float v) {
  // round-to-nearest-even to bf16, returned in the low 16 bits (finite inputs)
  const unsigned u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// S[R][C] (row-major, pitch ld) -> T[Rpad / BK][3][CG_NP][BK] bf16 with T[r / BK][t][c][r % BK] =
// term t of S[r][c]; zero for c >= C or r >= R.  Chunk-major: the operand of one chunk of the
// contraction (BK = 32 forward, 16 weight gradient) is ONE contiguous run of 3 * 128 * BK * 2
// bytes, so the kernels stream it with coalesced 16-byte loads.  (Round 6: with the planes
// [3][128][Rpad] a chunk was 384 pieces of 64 / 32 bytes, each Rpad * 2 bytes from the next --
// 384 DRAM pages per chunk -- and every count kernel, dense or tiles, waited ~2 k cycles per
// chunk on exactly those loads.)  One thread: one column, 8 consecutive rows (one 16-byte store
// per term).
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ S, int R,
                                                               int C, int ld,
                                                               uint16_t* __restrict__ T,
                                                               int Rpad, int BK) {
  const int c = threadIdx.x & (CG_NP - 1);
  const int r0 = (blockIdx.x * 2 + (threadIdx.x >> 7)) * 8;
  if (r0 >= Rpad) return;
  unsigned t1[8], t2[8], t3[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = r0 + j;
    const float w = (c < C && r < R) ? S[(size_t)r * ld + c] : 0.f;
    const unsigned b1 = bf16_rne_bits(w);
    const float r1 = w - __uint_as_float(b1 << 16);          // exact
    const unsigned b2 = bf16_rne_bits(r1);
    const float r2 = r1 - __uint_as_float(b2 << 16);         // exact, <= 8 significant bits
    const unsigned b3 = bf16_rne_bits(r2);
    t1[j] = b1; t2[j] = b2; t3[j] = b3;
  }
  const size_t plane = (size_t)CG_NP * BK;
  uint16_t* dst = T + ((size_t)(r0 / BK) * 3 * CG_NP + c) * BK + r0 % BK;
  auto pack = [](const unsigned* t) {
    u32x4 v;
    v.x = t[0] | (t[1] << 16); v.y = t[2] | (t[3] << 16);
    v.z = t[4] | (t[5] << 16); v.w = t[6] | (t[7] << 16);
    return v;
  };
  *reinterpret_cast<u32x4*>(dst) = pack(t1);
  *reinterpret_cast<u32x4*>(dst + plane) = pack(t2);
  *reinterpret_cast<u32x4*>(dst + 2 * plane) = pack(t3);
}

// offset (elements) of the 8-element piece `part` of row (term * 128 + column) in chunk kc / BK
template <int BK> __device__ __forceinline__ size_t cg_piece(int kc, int row, int part) {
  return ((size_t)(kc / BK) * 3 * CG_NP + row) * BK + part * 8;
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// ---- weight gradient (MODE 1): dW[F, N] = x^T dA ----
// A lane owns one gene and reads it down the cells of the chunk: every load instruction covers
// two 128-byte row segments of x (fully coalesced; 5.9 TB/s on its own).  256-thread workgroups
// (4 waves x 64 genes), two per CU, so that one workgroup's wait for HBM is the other's MFMA
// phase; chunks of 16 cells (one MFMA k-step), x requested two chunks ahead (static register
// slots), the split dA operand through LDS (48-byte rows: conflict-free ds_read_b128).
constexpr int CD_BK = 16;
constexpr int CD_ROW = 48;            // LDS bytes per (term, column) row: 32 + 16 pad

// NW: waves per workgroup -- 4 (256 genes, two workgroups per CU) or 8 (512 genes, one per CU: the
// dA planes of a chunk are fetched and parked once for twice the genes)
template <int NT, typename XT, bool PAIR = false, bool USE_STEADY = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void count_gemm_dw_kernel(
    const XT* __restrict__ X, int ldx, int M, int K, const uint16_t* __restrict__ T, int Kpad,
    int N, int k_chunk, float* __restrict__ out, int ldo) {
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][3 * CG_NP * CD_ROW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kg = lane >> 5;
  constexpr int NTHR = 64 * NW, NPC = (768 + NTHR - 1) / NTHR;   // pieces of dA per thread
  const int m_w = blockIdx.x * (NW * CG_TM) + w * CG_TM;       // first gene of this wave
  const int k_begin = blockIdx.y * k_chunk;
  const int k_end = min(K, k_begin + k_chunk);

  f32x16 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][q][i] = 0.f;

  // raw[slot][t][j]  <->  cell kc + 8 kg + j of gene tile t; uniform row pointer + per-lane
  // 32-bit element offset (gene + 8 kg rows)
  // PAIR (uint16 counts, M even): a lane reads genes 2 li and 2 li + 1 of the wave's 64 with one
  // 4-byte load -- gene tile 0 takes the even genes, tile 1 the odd ones -- half the load
  // instructions of the lane-per-gene pattern for the same bytes.
  static_assert(!PAIR || sizeof(XT) == 2, "gene pairs: uint16 counts");
  XT raw[2][2][PAIR ? 1 : 8];
  unsigned rawp[2][PAIR ? 8 : 1];
  unsigned xoff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    xoff[t] = (unsigned)(8 * kg) * (unsigned)ldx +
              (PAIR ? (unsigned)min(m_w + 2 * li, M - 2)
                    : (unsigned)min(m_w + 32 * t + li, M - 1));
  // the count of gene tile t, cell j of the slot, as fp32
  auto value = [&](auto slot_tag, int t, int j) -> float {
    constexpr int SLOT = decltype(slot_tag)::value;
    if constexpr (PAIR) return (float)(t == 0 ? (rawp[SLOT][j] & 0xFFFFu) : (rawp[SLOT][j] >> 16));
    else return count_to_f32(raw[SLOT][t][j]);
  };
  u32x4 breg[USE_STEADY ? 2 : 1][NPC];
  auto load_x = [&](int kc, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const XT* srow = X + (size_t)(kc + j) * ldx;               // uniform: scalar base
      if constexpr (PAIR) {
        rawp[SLOT][j] = cg_load<(SCVAE_CG_NT & 2) != 0>(
            reinterpret_cast<const unsigned*>(srow + xoff[0]));
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          raw[SLOT][t][j] = cg_load<(SCVAE_CG_NT & 2) != 0>(srow + xoff[t]);
      }
    }
  };
  auto load_b = [&](int kc, int slot = 0) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int p = tid + NTHR * i;                // 768 pieces: (term, column, half)
      const int row = p >> 1, part = p & 1;        // row = term * 128 + column
      // (columns beyond N: the last live column's piece again -- a line this wave requests
      //  anyway, no branch around the load; what they multiply into is never stored)
      const int col = row & (CG_NP - 1), rowl = col < N ? row : row - col + (N - 1);
      if (p < 768 && col < NT * 32)
        breg[slot][i] = *reinterpret_cast<const u32x4*>(T + cg_piece<CD_BK>(kc, rowl, part));
    }
  };
  auto store_b = [&](int buf, int slot = 0) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int p = tid + NTHR * i;
      const int row = p >> 1, part = p & 1;
      if (p < 768 && (row & (CG_NP - 1)) < NT * 32)
        *reinterpret_cast<u32x4*>(&Bs[buf][row * CD_ROW + part * 16]) = breg[slot][i];
    }
  };
  // (USE_STEADY: every request of the loop is unconditional -- a chunk index beyond the split's
  //  last chunk is clamped to it, its data never used -- so that the compiler can count the
  //  loads in flight; dA travels TWO chunks ahead, like x)
  const int k_last = k_end - CD_BK;
  auto clampk = [&](int k) { return min(k, k_last); };

  if (k_begin < k_end) {
    load_x(k_begin, std::integral_constant<int, 0>{});
    if (USE_STEADY) load_x(clampk(k_begin + CD_BK), std::integral_constant<int, 1>{});
    else if (k_begin + CD_BK < k_end) load_x(k_begin + CD_BK, std::integral_constant<int, 1>{});
    load_b(k_begin);
    store_b(0);
    if (USE_STEADY) load_b(clampk(k_begin + CD_BK), 1);
  }
  __syncthreads();

  const int frag_off = li * CD_ROW + 16 * kg;
  // STEADY (compile time): chunks j + 1 and j + 2 exist -- requests and hand-over unconditional
  // (see count_gemm_fwd_kernel).  Measured on this kernel the unconditional loop is SLOWER (178 vs
  // 155 us at 4096 x 32 738: both x chunks then really stay in flight, the kernel sits at 256
  // VGPRs and the deeper queue does not pay), so USE_STEADY defaults to off here.
  auto chunk = [&](int kc, auto buf_tag, auto steady_tag) {
    constexpr int BUF = decltype(buf_tag)::value;        // LDS buffer and x slot of this chunk
    constexpr bool STEADY = decltype(steady_tag)::value;
    // ---- cut the counts of this chunk into hi / lo bf16 fragments ----
    u32x4 ahi[2], alo[2];
    unsigned low_bits = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned h[4];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const unsigned u0 = __float_as_uint(value(buf_tag, t, 2 * pr));
        const unsigned u1 = __float_as_uint(value(buf_tag, t, 2 * pr + 1));
        low_bits |= u0 | u1;
        h[pr] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);          // upper halves
      }
      ahi[t] = u32x4{h[0], h[1], h[2], h[3]};
    }
    const bool need_lo =
        __builtin_amdgcn_readfirstlane(__any((int)((low_bits & 0xFFFFu) != 0u)));
    if (need_lo) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned l[4];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const float x0 = value(buf_tag, t, 2 * pr);
          const float x1 = value(buf_tag, t, 2 * pr + 1);
          const float l0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xFFFF0000u);
          const float l1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xFFFF0000u);
          l[pr] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
        }
        alo[t] = u32x4{l[0], l[1], l[2], l[3]};
      }
    }
    // ---- requests: dA one chunk ahead, x two chunks ahead (the slot just converted).  dA first:
    //      in the unconditional loop (STEADY) the wait for dA at the end of the chunk (store_b)
    //      is then vmcnt(8) and leaves the eight younger x loads in flight across the barrier;
    //      with x first it was vmcnt(0) -- the counter retires in issue order -- and in the
    //      conditional loop it still is (the compiler cannot count loads under a branch): every
    //      x request lands within the chunk that issued it.  Measured (round 5, SCVAE_CD_STEADY,
    //      tools/ab_cd_steady.sh): with the x requests really in flight the kernel is SLOWER,
    //      138.8-142.2 against 134.7-137.2 us stand-alone, + 7 us in the step -- as round 2
    //      found with the other order; the default stays the conditional loop ----
    const bool has_next = STEADY || kc + CD_BK < k_end;
    if (STEADY) {
      load_b(clampk(kc + 2 * CD_BK), BUF);
      __builtin_amdgcn_sched_barrier(0);
      load_x(clampk(kc + 2 * CD_BK), buf_tag);
    } else {
      if (has_next) load_b(kc + CD_BK);
      __builtin_amdgcn_sched_barrier(0);
      if (kc + 2 * CD_BK < k_end) load_x(kc + 2 * CD_BK, buf_tag);
    }
    __builtin_amdgcn_sched_barrier(0);

    const unsigned char* bcur = Bs[BUF] + frag_off;
#pragma unroll
    for (int term = 2; term >= 0; --term) {              // smallest term first
      bf16x8 fr[NT];
#pragma unroll
      for (int q = 0; q < NT; ++q)
        fr[q] = as_bf16x8(*reinterpret_cast<const u32x4*>(
            bcur + (term * CG_NP + q * 32) * CD_ROW));
      if (need_lo) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(alo[0]), fr[q], acc[0][q],
                                                              0, 0, 0);
          acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(alo[1]), fr[q], acc[1][q],
                                                              0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ahi[0]), fr[q], acc[0][q], 0,
                                                            0, 0);
        acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ahi[1]), fr[q], acc[1][q], 0,
                                                            0, 0);
      }
    }
    if (STEADY) store_b(BUF ^ 1, BUF ^ 1);     // (dA of chunk kc + 1: requested a chunk ago)
    else if (has_next) store_b(BUF ^ 1);
    lds_barrier();       // (LDS only: the x requests stay in flight across it)
  };
  {
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int kc = k_begin;
    if (USE_STEADY) {
      for (; kc + CD_BK < k_end; kc += 2 * CD_BK) {     // pairs of chunks
        chunk(kc, B0{}, std::true_type{});
        chunk(kc + CD_BK, B1{}, std::true_type{});
      }
      if (kc < k_end) chunk(kc, B0{}, std::true_type{});   // an odd last one
    } else
    for (; kc < k_end; kc += 2 * CD_BK) {
      chunk(kc, B0{}, std::false_type{});
      if (kc + CD_BK < k_end) chunk(kc + CD_BK, B1{}, std::false_type{});
    }
  }

  float* dst = out + (size_t)blockIdx.y * M * ldo;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int col = q * 32 + li;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * kg;         // row of the gene tile
        const int m = PAIR ? m_w + 2 * i + t : m_w + 32 * t + i;
        if (m < M) dst[(size_t)m * ldo + col] = acc[t][q][r];
      }
    }
}

// ---- forward (MODE 0) with the x tile staged through LDS ----
// Reading x straight into the operand layout (a lane = one cell, 64 contiguous bytes) touches 32
// cache lines per load instruction and a quarter of each: measured 3.5 TB/s on its own.  Here a
// workgroup (8 waves, 256 cells) reads its [256, 32] tile of x with fully coalesced 16-byte
// loads (8 lanes per 128-byte row segment), cuts it into hi / lo bf16 in registers and parks it
// in LDS (80-byte rows, the layout of the split operand); the MFMA fragments of both operands
// are then conflict-free ds_read_b128.  Wave w: cells 64 (w & 3) .. +63 (two 32-row tiles),
// column tiles NQ (w >> 2) .. + NQ - 1.  The lo plane is written only by waves whose cells need
// it (more than 8 significant bits) or whose earlier lo data has to be cleared; it is read when
// any wave of the workgroup flagged the chunk.
constexpr int CF_BM = 256;
constexpr int CF_A_BYTES = CF_BM * CG_ROW;            // one plane of one buffer

static size_t cf_lds_bytes(int NQ) {
  return 4 * (size_t)CF_A_BYTES + 2 * (size_t)(3 * 64 * NQ * CG_ROW) + 2 * 8 * sizeof(int);
}

template <int NQ, typename XT>
__global__ __launch_bounds__(512) void count_gemm_fwd_kernel(
    const XT* __restrict__ X, int ldx, int M, int K, const uint16_t* __restrict__ T, int Kpad,
    int N, int k_chunk, float* __restrict__ out, int ldo, const float* __restrict__ bias,
    int act, int direct) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
  constexpr int NCOL = 64 * NQ;                         // columns staged per term
  constexpr int B_BYTES = 3 * NCOL * CG_ROW;
  unsigned char* Ahi = cf_smem;                         // [2][256][80]
  unsigned char* Alo = Ahi + 2 * CF_A_BYTES;            // [2][256][80]
  unsigned char* Bsm = Alo + 2 * CF_A_BYTES;            // [2][3][NCOL][80]
  int* lo_flag = reinterpret_cast<int*>(Bsm + 2 * B_BYTES);   // [2][8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kg = lane >> 5;
  const int rg = w & 3, q0 = (w >> 2) * NQ;
  const int m0 = blockIdx.x * CF_BM;
  const int k_begin = blockIdx.y * k_chunk;
  const int k_end = min(K, k_begin + k_chunk);

  f32x16 acc[2][NQ];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][q][i] = 0.f;

  // zero both lo planes once (a wave only ever rewrites its own rows)
  for (int i = tid; i < 2 * CF_A_BYTES / 16; i += 512)
    reinterpret_cast<u32x4*>(Alo)[i] = u32x4{0u, 0u, 0u, 0u};

  // ---- staging: 16-byte pieces of the [256, 32] tile; fp32: 8 per row, thread -> 4 pieces (row
  // (tid >> 3) + 64 i, floats 4 (tid & 7) .. + 3); uint16: 4 per row, thread -> 2 pieces (row
  // (tid >> 2) + 128 i, counts 8 (tid & 3) .. + 7) ----
  constexpr int EPP = 16 / (int)sizeof(XT);             // elements per piece
  constexpr int PPR = CG_BK / EPP;                      // pieces per row
  constexpr int RPP = 512 / PPR;                        // rows per pass
  constexpr int PCS = CF_BM / RPP;                      // pieces per thread
  const int part = tid & (PPR - 1);
  const XT* xsrc[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    const int m = min(m0 + tid / PPR + RPP * i, M - 1);
    xsrc[i] = X + (size_t)m * ldx + EPP * part;
  }
  const int a_off = (tid / PPR) * CG_ROW + part * (2 * EPP);   // + i * RPP rows
  // two chunks of staging registers: chunk c + 2 is requested while chunk c is multiplied and
  // chunk c + 1 (requested one iteration earlier) is converted and parked -- a full iteration
  // plus the MFMA phase of latency tolerance with a single workgroup per CU
  f32x4u raw[2][PCS];
  u32x4 breg[2][3];
  auto load_tiles = [&](int kc, int slot) {
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
      const f32x4u* src = reinterpret_cast<const f32x4u*>(xsrc[i] + kc);
      if constexpr ((SCVAE_CG_NT & 1) != 0) raw[slot][i] = __builtin_nontemporal_load(src);
      else raw[slot][i] = *src;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p = tid + 512 * i;                      // (term, column, quarter)
      const int row = p >> 2, prt = p & 3;              // row = term * 128 + column
      // (columns beyond N: the last live column's piece again, as in count_gemm_dw_kernel)
      const int col = row & (CG_NP - 1), rowl = col < N ? row : row - col + (N - 1);
      if (col < NCOL)
        breg[slot][i] = *reinterpret_cast<const u32x4*>(T + cg_piece<CG_BK>(kc, rowl, prt));
    }
  };
  bool dirty0 = false, dirty1 = false;                  // this wave's lo rows of buffer b are set
  auto store_tiles = [&](int buf, int slot) {
    unsigned low = 0u;
    // the piece's counts as fp32 bit patterns (uint16: two per loaded dword)
    auto bits_of = [&](int i, unsigned* u) {
      if constexpr (sizeof(XT) == 4) {
        u[0] = __float_as_uint(raw[slot][i].x); u[1] = __float_as_uint(raw[slot][i].y);
        u[2] = __float_as_uint(raw[slot][i].z); u[3] = __float_as_uint(raw[slot][i].w);
      } else {
        const unsigned w[4] = {__float_as_uint(raw[slot][i].x), __float_as_uint(raw[slot][i].y),
                               __float_as_uint(raw[slot][i].z), __float_as_uint(raw[slot][i].w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u[2 * j] = __float_as_uint((float)(w[j] & 0xFFFFu));
          u[2 * j + 1] = __float_as_uint((float)(w[j] >> 16));
        }
      }
    };
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
      unsigned u[EPP], h[EPP / 2];
      bits_of(i, u);
#pragma unroll
      for (int j = 0; j < EPP / 2; ++j) {
        low |= u[2 * j] | u[2 * j + 1];
        h[j] = __builtin_amdgcn_perm(u[2 * j + 1], u[2 * j], 0x07060302u);     // upper halves
      }
      unsigned char* dst = Ahi + buf * CF_A_BYTES + a_off + i * RPP * CG_ROW;
      if constexpr (EPP == 4) *reinterpret_cast<uint2*>(dst) = uint2{h[0], h[1]};
      else *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
    }
    const bool need = __builtin_amdgcn_readfirstlane(__any((int)((low & 0xFFFFu) != 0u)));
    if (need || (buf ? dirty1 : dirty0)) {
#pragma unroll
      for (int i = 0; i < PCS; ++i) {
        unsigned u[EPP], l[EPP / 2];
        bits_of(i, u);
#pragma unroll
        for (int j = 0; j < EPP / 2; ++j) {
          const float l0 = __uint_as_float(u[2 * j]) - __uint_as_float(u[2 * j] & 0xFFFF0000u);
          const float l1 = __uint_as_float(u[2 * j + 1]) - __uint_as_float(u[2 * j + 1] & 0xFFFF0000u);
          l[j] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
        }
        unsigned char* dst = Alo + buf * CF_A_BYTES + a_off + i * RPP * CG_ROW;
        if constexpr (EPP == 4) *reinterpret_cast<uint2*>(dst) = uint2{l[0], l[1]};
        else *reinterpret_cast<u32x4*>(dst) = u32x4{l[0], l[1], l[2], l[3]};
      }
    }
    if (buf) dirty1 = need; else dirty0 = need;
    if (lane == 0) lo_flag[buf * 8 + w] = need ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p = tid + 512 * i;
      const int row = p >> 2, prt = p & 3;
      const int term = row >> 7, col = row & (CG_NP - 1);
      if (col < NCOL)
        *reinterpret_cast<u32x4*>(Bsm + buf * B_BYTES + (term * NCOL + col) * CG_ROW + prt * 16) =
            breg[slot][i];
    }
  };

  __syncthreads();                                      // lo planes zeroed
  if (k_begin < k_end) {
    load_tiles(k_begin, 0);
    if (k_begin + CG_BK < k_end) load_tiles(k_begin + CG_BK, 1);
    store_tiles(0, 0);
  }
  __syncthreads();

  const int a_frag = (64 * rg + li) * CG_ROW + 32 * kg;      // + 32 rows * t, + 16 s
  const int b_frag = (q0 * 32 + li) * CG_ROW + 32 * kg;      // + term * NCOL rows, + 32 rows * q
  // one chunk; BUF (compile time: the staging registers are indexed statically) = LDS buffer and
  // staging slot of chunk j = j & 1
  // STEADY (compile time): chunks j + 1 and j + 2 exist, so the request and the hand-over are
  // unconditional -- with conditions the compiler cannot pair them up and waits for every
  // outstanding load at the loop header, which cancels the second chunk of latency tolerance
  auto chunk = [&](int kc, auto buf_tag, auto steady_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
    constexpr bool STEADY = decltype(steady_tag)::value;
    // chunk j + 2 -> staging slot BUF (chunk j left it for LDS before this iteration)
    if (STEADY || kc + 2 * CG_BK < k_end) load_tiles(kc + 2 * CG_BK, BUF);
    __builtin_amdgcn_sched_barrier(0);

    const bool need_lo =
        __builtin_amdgcn_readfirstlane(__any(lo_flag[BUF * 8 + (lane & 7)]));
    const unsigned char* ah = Ahi + BUF * CF_A_BYTES + a_frag;
    const unsigned char* al = Alo + BUF * CF_A_BYTES + a_frag;
    const unsigned char* bb = Bsm + BUF * B_BYTES + b_frag;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fh[2], fl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
        fh[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(ah + t * 32 * CG_ROW + 16 * s));
      if (need_lo) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          fl[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(al + t * 32 * CG_ROW + 16 * s));
      }
#pragma unroll
      for (int term = 2; term >= 0; --term) {           // smallest term first
        bf16x8 fb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          fb[q] = as_bf16x8(*reinterpret_cast<const u32x4*>(
              bb + (term * NCOL + q * 32) * CG_ROW + 16 * s));
        if (need_lo) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[0], fb[q], acc[0][q], 0, 0, 0);
            acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[1], fb[q], acc[1][q], 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[0], fb[q], acc[0][q], 0, 0, 0);
          acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[1], fb[q], acc[1][q], 0, 0, 0);
        }
      }
    }
    // chunk j + 1 (requested one iteration ago, staging slot BUF ^ 1) -> LDS buffer BUF ^ 1
    // (scheduling fence: the conversion and its wait for the loads stay below the MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    if (STEADY || kc + CG_BK < k_end) store_tiles(BUF ^ 1, BUF ^ 1);
    lds_barrier();       // (LDS only: the requests for chunk j + 2 stay in flight across it)
  };
  {
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int kc = k_begin;
    for (; kc + 3 * CG_BK < k_end; kc += 2 * CG_BK) {   // chunks j, j + 1 with j + 3 in range
      chunk(kc, B0{}, std::true_type{});
      chunk(kc + CG_BK, B1{}, std::true_type{});
    }
    for (; kc < k_end; kc += 2 * CG_BK) {               // the last one to three chunks
      chunk(kc, B0{}, std::false_type{});
      if (kc + CG_BK < k_end) chunk(kc + CG_BK, B1{}, std::false_type{});
    }
  }

  float* dst = direct ? out : out + (size_t)blockIdx.y * M * ldo;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int col = (q0 + q) * 32 + li;
      if (col >= N) continue;
      const float bv = (direct && bias != nullptr) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 64 * rg + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (m < M) {
          float v = acc[t][q][r] + bv;
          if (direct && act == ACT_RELU) v = fmaxf(v, 0.f);
          dst[(size_t)m * ldo + col] = v;
        }
      }
    }
}

// fixed-order sum of the split-K slabs, + the K % 32 leftover terms of the contraction (plain
// fp32 fma on the original operands: k_main <= k < K), + bias, activation
template <typename XT>
__global__ __launch_bounds__(256) void count_gemm_reduce_kernel(
    const float* __restrict__ slabs, const float* __restrict__ bias, float* __restrict__ C, int M,
    int N, int ldc, int splits, int act, int mode, const XT* __restrict__ X, int ldx,
    const float* __restrict__ other, int ld_other, int k_main, int K) {
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / N), col = (int)(i % N);
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {      // (eight slabs' loads in flight, summed in slab order)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(z + u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; ++z) s += slabs[(size_t)z * total + i];
    for (int k = k_main; k < K; ++k) {
      const float xv = count_to_f32(mode == 0 ? X[(size_t)row * ldx + k] : X[(size_t)k * ldx + row]);
      s = fmaf(xv, other[(size_t)k * ld_other + col], s);
    }
    if (bias) s += bias[col];
    if (act == ACT_RELU) s = fmaxf(s, 0.f);
    C[(size_t)row * ldc + col] = s;
  }
}

// (A/B: SCVAE_CD_STEADY=0 / 1 -- the weight-gradient kernel's unconditional main loop)
static bool cd_steady() {
  static const bool on = [] { const char* e = getenv("SCVAE_CD_STEADY"); return e ? e[0] == '1' : false; }();
  return on;
}

// Waves per workgroup of the weight-gradient kernel (uint16 gene pairs): 8 -- 512 genes, one
// workgroup per CU, the dA planes of a chunk fetched and parked once for twice the genes -- where
// such workgroups still fill the chip, else 4 (256 genes, two per CU).  The kernel is bound by
// what it moves through the L2s, x + one copy of the dA planes per workgroup: 268 + 403 MB at
// 4096 x 32 738 with 512 workgroups, 268 + 201 MB with 256 (round 5: 146-155 -> 131-136 us
// stand-alone, -17 to -25 us in the step; tools/ab_cd_waves.sh, SCVAE_CD_WAVES=4 / 8 forces one).
static int cd_waves(int M, int K) {
  static const int forced = [] { const char* e = getenv("SCVAE_CD_WAVES"); return e ? atoi(e) : 0; }();
  if (forced == 4 || forced == 8) return forced;
  const long blocks = (M + 8 * CG_TM - 1) / (8 * CG_TM);
  const long want = (256 + blocks - 1) / blocks, max_by_k = K / 256 > 0 ? K / 256 : 1;
  return blocks * (want < max_by_k ? want : max_by_k) >= 224 ? 8 : 4;
}

static int cg_splits(int mode, int M, int K) {
  const int nw = mode == 0 ? 0 : cd_waves(M, K);
  const int bm = mode == 0 ? CF_BM : nw * CG_TM;
  const long blocks = (M + bm - 1) / bm;
  const long target = mode == 0 || nw == 8 ? 256 : 512;   // workgroups per CU: forward 1, dW 2 (or 1)
  long want = (target + blocks - 1) / blocks;
  const long max_by_k = K / 256 > 0 ? K / 256 : 1;   // at least 8 chunks per split
  long s = want < max_by_k ? want : max_by_k;
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  return (int)s;
}

bool count_gemm_supported(int N) { return N >= 1 && N <= CG_NP; }

// mode 0: x [rows, cols] (pitch ldx) times other [cols, N]   -> C [rows, N]
// mode 1: x^T                        times other [rows, N]   -> C [cols, N]
size_t count_gemm_workspace_bytes(int mode, int rows, int cols, int N) {
  if (!count_gemm_supported(N)) return 0;
  const int M = mode == 0 ? rows : cols, K = mode == 0 ? cols : rows;
  const size_t t_bytes = (size_t)3 * CG_NP * cg_kpad(K) * sizeof(uint16_t);
  const int splits = cg_splits(mode, M, K / cg_bk(mode) * cg_bk(mode));
  return (t_bytes + 255) / 256 * 256 + (size_t)splits * M * N * sizeof(float);
}

// the two-role forward kernel (defined with the count tiles below; TILES = false: x from the dense
// uint16 batch) and its LDS
template <int NQ, bool TILES>
__global__ __launch_bounds__(512) void count_fwd2_kernel(const uint32_t* ent, const uint32_t* tptr, int ntp, int n_groups,
                                  const uint16_t* X, int ldx, int M, int K, const uint16_t* T,
                                  int Kpad, int N, int k_chunk, float* out, int ldo,
                                  const float* bias, int act, int direct);
static size_t ctf_lds_bytes(int NQ);
// SCVAE_CG_FWD2=1: the two-role kernel for uint16 batches instead of count_gemm_fwd_kernel.
// (Round 6, 4096 x 32 738 x 100: 119 us against 125 us alone, bit-identical, no difference in the
// training step's time -- so the kernel of five rounds of tests stays the default.)
static bool cf_two_roles() {
  static const bool on = [] {
    const char* e = getenv("SCVAE_CG_FWD2");
    return e && e[0] == '1';
  }();
  return on;
}

template <typename XT>
static int count_gemm_impl(hipStream_t stream, int mode, const XT* x, int ldx, int rows, int cols,
                           const float* other, int ld_other, int N, const float* bias, int act,
                           float* C, int ldc, void* workspace, size_t workspace_bytes) {
  SCVAE_ARG(x && other && C && workspace);
  SCVAE_ARG(mode == 0 || mode == 1);
  SCVAE_ARG(count_gemm_supported(N) && ld_other >= N && ldc >= N && ldx >= cols);
  // 16-byte loads of a row need 4-byte aligned rows
  SCVAE_ARG(sizeof(XT) == 4 || ((ldx & 1) == 0 && ((uintptr_t)x & 3) == 0));
  if (rows == 0 || cols == 0) return 0;
  SCVAE_ARG(workspace_bytes >= count_gemm_workspace_bytes(mode, rows, cols, N));
  SCVAE_ARG(((uintptr_t)workspace & 15) == 0);
  const int M = mode == 0 ? rows : cols, K = mode == 0 ? cols : rows;
  const int bk = cg_bk(mode);
  const int k_main = K / bk * bk;            // whole chunks: matrix cores; the rest: reduction
  const int Kpad = cg_kpad(K);
  uint16_t* T = reinterpret_cast<uint16_t*>(workspace);
  const size_t t_bytes = ((size_t)3 * CG_NP * Kpad * sizeof(uint16_t) + 255) / 256 * 256;
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + t_bytes);
  int splits = 0;
  if (k_main > 0) {
    hipLaunchKernelGGL(split3_transpose_kernel, dim3((Kpad / 8 + 1) / 2), dim3(256), 0, stream,
                       other, K, N, ld_other, T, Kpad, bk);
    SCVAE_LAUNCH_CHECK("split3_transpose_kernel");
    splits = cg_splits(mode, M, k_main);
    int k_chunk = k_main;
    if (splits > 1) {
      k_chunk = (k_main + splits - 1) / splits;
      k_chunk = (k_chunk + bk - 1) / bk * bk;
      splits = (k_main + k_chunk - 1) / k_chunk;
    }
    // a single split with no leftover terms writes C directly (bias and activation included)
    const bool direct = mode == 0 && splits == 1 && k_main == K;
    float* dst = direct ? C : slabs;
    const int ldo = direct ? ldc : N;
    const int NT = (N + 31) / 32;
    const float* kbias = direct ? bias : nullptr;
    const int kact = direct ? act : (int)ACT_NONE, kdirect = direct ? 1 : 0;
    if (mode == 0) {
      const dim3 grid((M + CF_BM - 1) / CF_BM, splits);
      const int NQ = NT > 2 ? 2 : 1;
      const size_t lds = cf_lds_bytes(NQ);
#define SCVAE_CF(NQ_)                                                                             \
  do {                                                                                            \
    auto kfn = count_gemm_fwd_kernel<NQ_, XT>;                                                    \
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), \
                                  (int)lds));         \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, stream, x, ldx, M, k_main, T, Kpad, N, k_chunk, \
                       dst, ldo, kbias, kact, kdirect);                                           \
  } while (0)
#define SCVAE_CF2(NQ_)                                                                            \
  do {                                                                                            \
    auto kfn = count_fwd2_kernel<NQ_, false>;                                                     \
    const size_t lds2 = ctf_lds_bytes(NQ_);                                                       \
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)lds2));                    \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds2, stream, (const uint32_t*)nullptr,              \
                       (const uint32_t*)nullptr, 0, 0, reinterpret_cast<const uint16_t*>(x), ldx, \
                       M, k_main, T, Kpad, N, k_chunk, dst, ldo, kbias, kact, kdirect);           \
  } while (0)
      if (sizeof(XT) == 2 && cf_two_roles()) {
        if (NQ == 2) SCVAE_CF2(2); else SCVAE_CF2(1);
      } else if (NQ == 2) SCVAE_CF(2); else SCVAE_CF(1);
#undef SCVAE_CF2
#undef SCVAE_CF
    } else {
      const int nw = (sizeof(XT) == 2 && (M & 1) == 0) ? cd_waves(M, k_main) : 4;   // (8: the pair kernel)
      const dim3 grid((M + nw * CG_TM - 1) / (nw * CG_TM), splits);
#define SCVAE_CD(NT_)                                                                             \
  do {                                                                                            \
    if constexpr (sizeof(XT) == 2) {                                                              \
      if ((M & 1) == 0) {   /* gene pairs per lane (4-byte loads) */                               \
        if (nw == 8)                                                                              \
          hipLaunchKernelGGL((count_gemm_dw_kernel<NT_, XT, true, false, 8>), grid, dim3(512), 0, \
                             stream, x, ldx, M, k_main, T, Kpad, N, k_chunk, dst, ldo);           \
        else if (cd_steady())                                                                     \
          hipLaunchKernelGGL((count_gemm_dw_kernel<NT_, XT, true, true>), grid, dim3(256), 0,     \
                             stream, x, ldx, M, k_main, T, Kpad, N, k_chunk, dst, ldo);           \
        else                                                                                      \
          hipLaunchKernelGGL((count_gemm_dw_kernel<NT_, XT, true>), grid, dim3(256), 0, stream,   \
                             x, ldx, M, k_main, T, Kpad, N, k_chunk, dst, ldo);                   \
        break;                                                                                    \
      }                                                                                           \
    }                                                                                             \
    hipLaunchKernelGGL((count_gemm_dw_kernel<NT_, XT, false>), grid, dim3(256), 0, stream, x,     \
                       ldx, M, k_main, T, Kpad, N, k_chunk, dst, ldo);                            \
  } while (0)
      switch (NT) { case 1: SCVAE_CD(1); break; case 2: SCVAE_CD(2); break;
                    case 3: SCVAE_CD(3); break; default: SCVAE_CD(4); }
#undef SCVAE_CD
    }
    SCVAE_LAUNCH_CHECK("count_gemm_kernel");
    if (direct) return 0;
  }
  const size_t total = (size_t)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(count_gemm_reduce_kernel<XT>, dim3(blocks), dim3(256), 0, stream, slabs, bias,
                     C, M, N, ldc, splits, act, mode, x, ldx, other, ld_other, k_main, K);
  SCVAE_LAUNCH_CHECK("count_gemm_reduce_kernel");
  return 0;
}

int count_gemm(hipStream_t stream, int mode, const float* x, int ldx, int rows, int cols,
               const float* other, int ld_other, int N, const float* bias, int act, float* C,
               int ldc, void* workspace, size_t workspace_bytes) {
  return count_gemm_impl<float>(stream, mode, x, ldx, rows, cols, other, ld_other, N, bias, act, C,
                                ldc, workspace, workspace_bytes);
}

int count_gemm_u16(hipStream_t stream, int mode, const uint16_t* x, int ldx, int rows, int cols,
                   const float* other, int ld_other, int N, const float* bias, int act, float* C,
                   int ldc, void* workspace, size_t workspace_bytes) {
  return count_gemm_impl<uint16_t>(stream, mode, x, ldx, rows, cols, other, ld_other, N, bias, act,
                                   C, ldc, workspace, workspace_bytes);
}

// ====================== the minibatch as tile-indexed non-zeros ======================
// A count minibatch is ~5 % non-zeros; densified to uint16 it is 268 MB at 4096 x 32 738 of which
// the two products above use 27 MB.  CountTiles is the same minibatch as a list of its non-zeros
// grouped by (16 cells, 32 genes): for group g of 16 consecutive minibatch rows the entries of
// gene tile t are entries[tile_ptr[g][t] .. tile_ptr[g][t + 1]) -- in no particular order within
// the bucket --, tiles ascending, so that 16 tiles (a block of 512 genes) are one contiguous run
// block_ptr[g][b] .. block_ptr[g][b + 1].  An entry is
//     bf16(value) << 16 | lo << 13 | (tile & 15) << 9 | row << 5 | gene & 31
// with value the count cut as the dense kernels cut it (and carried as its bf16 bit pattern): a count of up to 8 significant bits is
// one entry; a larger one is two, its upper 8 significant bits (lo = 0) and the remainder
// (lo = 1), both exact in bf16.  Bit 31 of a pointer says that the bucket (the block) holds lo
// entries.  The sparse kernels below scatter a bucket into the zeroed LDS tile the dense kernels
// fill from the uint16 batch and run the same MFMAs on it in the same order: bit-identical
// results, a tenth of the bytes.
constexpr int CT_ROWS = 16, CT_GENES = 32, CT_BLOCK = 16;
constexpr unsigned CT_LO = 0x80000000u, CT_MASK = 0x7FFFFFFFu;
constexpr int CT_MAX_TILES = 2048;            // F <= 65 536

int count_tiles_padded(int F) { return (F + 511) / 512 * CT_BLOCK; }
bool count_tiles_supported(int F) { return F > 0 && F <= CT_MAX_TILES * CT_GENES; }

// hi / lo cut of an integer count below 65 536 (the fp32 bit pattern truncated to 16 bits)
__device__ __forceinline__ void ct_cut(unsigned v, unsigned& hi, unsigned& lo) {
  const int nb = 32 - __builtin_clz(v | 1u);
  const int sh = nb > 8 ? nb - 8 : 0;
  hi = (v >> sh) << sh;
  lo = v - hi;
}

// bf16 bits of an integer of at most 8 significant bits (exact)
__device__ __forceinline__ unsigned ct_bits(unsigned v) { return __float_as_uint((float)v) >> 16; }

// One workgroup per group of 16 rows (a wave per row): count the entries of every tile, scan,
// place.  Rows of one group land in the group's own region [g * cap, (g + 1) * cap) of `ent`, so
// no workgroup waits for another.
__global__ __launch_bounds__(1024) void csr_count_tiles_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ values, const int64_t* __restrict__ rows, int n, int F, int ntp,
    uint32_t* __restrict__ ent, long cap, uint32_t* __restrict__ tptr, uint32_t* __restrict__ gptr,
    int* __restrict__ status) {
  __shared__ unsigned cnt[CT_MAX_TILES], flag[CT_MAX_TILES], off[CT_MAX_TILES + 1], wtot[16];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < ntp; i += 1024) { cnt[i] = 0u; flag[i] = 0u; }
  __syncthreads();
  const int b = g * CT_ROWS + w;
  int64_t lo_j = 0, hi_j = 0;
  if (b < n) { const int64_t r = rows[b]; lo_j = indptr[r]; hi_j = indptr[r + 1]; }
  // (four strides of the row per trip: eight loads in flight per lane)
  for (int64_t j0 = lo_j + lane; j0 < hi_j; j0 += 256) {
    int c[4]; float fv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = j0 + 64 * u;
      c[u] = j < hi_j ? indices[j] : -1;
      fv[u] = j < hi_j ? values[j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned v = (unsigned)(int)fv[u];
      if (c[u] < 0 || c[u] >= F || v == 0u) continue;
      unsigned h, l;
      ct_cut(v & 0xFFFFu, h, l);
      atomicAdd(&cnt[c[u] >> 5], l ? 2u : 1u);
      if (l) flag[c[u] >> 5] = 1u;
    }
  }
  __syncthreads();
  {   // exclusive scan of cnt[0 .. ntp): two tiles per thread
    const unsigned a = 2 * tid < ntp ? cnt[2 * tid] : 0u;
    const unsigned c2 = 2 * tid + 1 < ntp ? cnt[2 * tid + 1] : 0u;
    unsigned s = a + c2, incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    unsigned base = 0u;
    for (int i = 0; i < w; ++i) base += wtot[i];
    const unsigned ex = base + incl - s;
    if (2 * tid < ntp) off[2 * tid] = ex;
    if (2 * tid + 1 < ntp) off[2 * tid + 1] = ex + a;
    if (tid == 1023) off[ntp] = base + incl;      // (tiles beyond ntp count zero)
  }
  __syncthreads();
  const unsigned total = off[ntp];
  const bool fits = (long)total <= cap;
  if (!fits && tid == 0 && status) atomicOr(status, 1);
  const unsigned gbase = (unsigned)((long)g * cap);
  const int ngb = ntp / CT_BLOCK;
  for (int i = tid; i <= ntp; i += 1024) {
    const unsigned o = fits ? off[i] : 0u;       // (an overflowing group: empty, flagged)
    tptr[(size_t)g * (ntp + 1) + i] = (gbase + o) | ((i < ntp && fits && flag[i]) ? CT_LO : 0u);
  }
  for (int i = tid; i <= ngb; i += 1024) {
    unsigned f = 0u;
    if (i < ngb && fits)
      for (int k = 0; k < CT_BLOCK; ++k) f |= flag[i * CT_BLOCK + k];
    gptr[(size_t)g * (ngb + 1) + i] = (gbase + (fits ? off[i * CT_BLOCK] : 0u)) | (f ? CT_LO : 0u);
  }
  if (!fits) return;
  __syncthreads();
  for (int i = tid; i < ntp; i += 1024) cnt[i] = 0u;
  __syncthreads();
  for (int64_t j0 = lo_j + lane; j0 < hi_j; j0 += 256) {
    int c[4]; float fv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = j0 + 64 * u;
      c[u] = j < hi_j ? indices[j] : -1;
      fv[u] = j < hi_j ? values[j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned v = (unsigned)(int)fv[u];
      if (c[u] < 0 || c[u] >= F || v == 0u) continue;
      unsigned h, l;
      ct_cut(v & 0xFFFFu, h, l);
      const int t = c[u] >> 5;
      const unsigned slot = atomicAdd(&cnt[t], l ? 2u : 1u);
      const unsigned key = ((unsigned)(t & 15) << 9) | ((unsigned)w << 5) | (unsigned)(c[u] & 31);
      uint32_t* dst = ent + gbase + off[t] + slot;
      dst[0] = (ct_bits(h) << 16) | key;
      if (l) dst[1] = (ct_bits(l) << 16) | (1u << 13) | key;
    }
  }
}

// per row: entries it contributes (non-zeros + one more per count above 8 significant bits whose
// remainder is not zero) -- the caller sizes CountTiles.cap as 16 x the maximum over the matrix
__global__ __launch_bounds__(256) void csr_row_entries_kernel(const int64_t* __restrict__ indptr,
                                                             const float* __restrict__ values,
                                                             int64_t n_rows,
                                                             int32_t* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int lane = threadIdx.x & 63;
  int s = 0;
  for (int64_t j = indptr[r] + lane; j < indptr[r + 1]; j += 64) {
    const unsigned v = (unsigned)(int)values[j] & 0xFFFFu;
    unsigned h, l;
    ct_cut(v, h, l);
    s += v ? (l ? 2 : 1) : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[r] = s;
}

int csr_row_entries(hipStream_t stream, const int64_t* indptr, const float* values, int64_t n_rows,
                    int32_t* out) {
  SCVAE_ARG(indptr && values && out);
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(csr_row_entries_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0,
                     stream, indptr, values, n_rows, out);
  SCVAE_LAUNCH_CHECK("csr_row_entries_kernel");
  return 0;
}

int csr_count_tiles(hipStream_t stream, const int64_t* indptr, const int32_t* indices,
                    const float* values, const int64_t* rows, int B, int F, CountTiles t) {
  SCVAE_ARG(indptr && indices && values && rows && t.ent && t.tptr && t.gptr);
  SCVAE_ARG(count_tiles_supported(F) && t.cap > 0 && B >= 0);
  const long groups = (B + CT_ROWS - 1) / CT_ROWS;
  SCVAE_ARG(groups * t.cap < (1L << 31));
  if (B == 0) return 0;
  hipLaunchKernelGGL(csr_count_tiles_kernel, dim3((unsigned)groups), dim3(1024), 0, stream, indptr,
                     indices, values, rows, B, F, count_tiles_padded(F), t.ent, (long)t.cap,
                     t.tptr, t.gptr, t.status);
  SCVAE_LAUNCH_CHECK("csr_count_tiles_kernel");
  return 0;
}

// Probe build (-DCT_PROF=1, scvae_amd/csrc/build_ctprof.sh): s_memtime sums per section of a chunk
// for the eight waves of workgroup (0, 0) of count_fwd2_kernel (scvae_ct_prof_dump,
// tools/ct_prof.py).
#ifndef CT_PROF
#define CT_PROF 0
#endif
#ifndef CT_PRIO
#define CT_PRIO 0
#endif
#ifndef CT_EXP
#define CT_EXP 0      // probe experiments (wrong results): 2 = the entries of chunk 0 always;
                      // 4 = no MFMAs, 5 = no staging, 6 = neither, 7 = 5 and no W planes in the multiplying waves
#endif
#if CT_PROF
__device__ unsigned long long ct_prof[8 * 8];
#define CT_STAMP(k)                                                         \
  do {                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                      \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
    pacc[k] += now_ - plast; plast = now_;                                  \
    __builtin_amdgcn_sched_barrier(0);                                      \
  } while (0)
#else
#define CT_STAMP(k) do {} while (0)
#endif
#define CT_FENCE __builtin_amdgcn_sched_barrier(0)

// ---- forward from tiles: count_gemm_fwd_kernel with the [256, 32] tile of x scattered into LDS
// from its non-zeros (16 buckets: one per group of 16 rows) instead of copied from the dense
// batch.  Same MFMAs on the same operands in the same order per accumulator: bit-identical.
//
// What bounds these kernels (round 6, s_memtime per section, tools/ct_prof.py): neither bytes nor
// instruction count but what the two waves of a SIMD do beside each other.
//   * With all eight waves running the same program -- stage, barrier, multiply -- both waves of
//     a SIMD multiply at the same time (8 MFMAs: 250 cycles, the pipe's rate) and stage at the
//     same time (the pipe idle): 3.8 k cycles per chunk for 1.5 k of matrix work, whatever the
//     barriers, the buffer count, the order of the pieces or the instruction count.
//   * With two roles (waves 0-3 multiply, waves 4-7 stage) the chunk still took 3.8 k: alone the
//     multiplying waves need 2.1 k and the staging waves 2.0 k, together the SUM -- the staging
//     wave's 14 global loads per chunk take ~140 cycles EACH to issue beside a wave that streams
//     MFMAs (1.9 k of its 3.6 k; 0.6 k alone), whatever the priorities, and whichever of its
//     streams is made cache-hot.
// Hence: few vector-memory instructions per wave and chunk, spread over all eight waves.
//   waves 0-3 (one per SIMD) multiply chunk j from buffers j & 1 (48 MFMAs each; a
//     [128 rows, 64 columns] tile per wave at NQ = 2: 20 fragment reads per chunk);
//   waves 4-7 stage the x side of chunk j + 1 into the other buffers meanwhile: a wave owns 64
//     rows = four buckets, a quarter-wave per bucket; per chunk and wave ONE load (a bucket's
//     first 64 entries: an aligned 16 bytes per lane), the tile pointers read 16 chunks at a time
//     and handed around by ds_bpermute, and no zeroing pass: a lane stores zeros where it wrote
//     two chunks ago before it scatters -- LDS operations of one wave complete in order;
//   all eight waves move the W planes of chunk j + 1 (three 16-byte pieces per thread: loaded a
//     chunk ahead, stored, the next requested);
//   one barrier per chunk.
// Entries carry their bf16 value bits (the fetch converts once): an entry becomes an LDS address
// (bit field extract, multiply-add) and a 16-bit store of its upper half; no branches in the
// common path (a lane without an entry -- or with a lo entry in the hi pass -- stores to a dummy
// slot of its own).  The lo plane (counts above 8 significant bits: 0.02 % of the entries, 8 % of
// the chunks) is zeroed by its owner only after it was written.
constexpr int CTF_DUMMY = 256 * 2;      // bytes: a 2-byte slot per staging thread
constexpr int CTF_WINDOW = 15;          // chunks served by one read of the tile pointers

static size_t ctf_lds_bytes(int NQ) {
  return 4 * (size_t)CF_A_BYTES + 2 * (size_t)(3 * 64 * NQ * CG_ROW) + 2 * 4 * sizeof(int) +
         CTF_DUMMY;
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int NQ, bool TILES>
__global__ __launch_bounds__(512) void count_fwd2_kernel(
    const uint32_t* __restrict__ ent, const uint32_t* __restrict__ tptr, int ntp, int n_groups,
    const uint16_t* __restrict__ X, int ldx, int M, int K, const uint16_t* __restrict__ T,
    int Kpad, int N, int k_chunk, float* __restrict__ out, int ldo,
    const float* __restrict__ bias, int act, int direct) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
  constexpr int NCOL = 64 * NQ;
  constexpr int B_BYTES = 3 * NCOL * CG_ROW;
  constexpr int AHI = 0, ALO = 2 * CF_A_BYTES, BSM = 4 * CF_A_BYTES, FLG = BSM + 2 * B_BYTES,
                DUM = FLG + 2 * 4 * 4;
  int* lo_flag = reinterpret_cast<int*>(cf_smem + FLG);   // [2][4]: buffer, staging wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * CF_BM;
  const int k_begin = blockIdx.y * k_chunk;
  const int k_end = min(K, k_begin + k_chunk);
  const int k_last = k_end - CG_BK;
  const int nch = k_begin < k_end ? (k_end - k_begin + CG_BK - 1) / CG_BK : 0;

  // all four x planes start clean: a hi plane is kept clean by un-scattering, a lo plane is
  // zeroed by its owners after use
  for (int i = tid; i < (4 * CF_A_BYTES) / 16; i += 512)
    reinterpret_cast<u32x4*>(cf_smem)[i] = u32x4{0u, 0u, 0u, 0u};
  if (tid < 8) lo_flag[tid] = 0;

  // ---- the W planes: every thread moves NPB 16-byte pieces per chunk ----
  constexpr int NPIECE = 3 * NCOL * 4;
  constexpr int NPB = (NPIECE + 511) / 512;
  unsigned boff[NPB], bst[NPB];
  bool bon[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int p2 = tid + 512 * i;                       // (term, column, quarter)
    bon[i] = p2 < NPIECE;
    const int row = (bon[i] ? p2 : 0) >> 2, prt = p2 & 3;
    const int term = row / NCOL, col = row % NCOL;
    const int rowl = term * CG_NP + (col < N ? col : N - 1);
    boff[i] = ((unsigned)rowl * CG_BK + prt * 8) * 2u;                          // bytes
    bst[i] = BSM + (term * NCOL + col) * CG_ROW + prt * 16;
  }
  u32x4 breg[NPB];
  auto load_b = [&](int kc) {     // (clamped to the split's last chunk: unconditional loads)
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(T) +
                              (size_t)(min(kc, k_last) / CG_BK) * (3 * CG_NP * CG_BK * 2);   // uniform
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      if (NPIECE % 512 == 0 || bon[i]) breg[i] = *reinterpret_cast<const u32x4*>(tb + boff[i]);
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      if (NPIECE % 512 == 0 || bon[i])
        *reinterpret_cast<u32x4*>(cf_smem + bst[i] + buf * B_BYTES) = breg[i];
  };
  if (nch > 0) load_b(k_begin);
  __syncthreads();                                      // planes zeroed
  if (nch > 0) { store_b(0); load_b(k_begin + CG_BK); }
#if CT_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = __builtin_amdgcn_s_memtime();
#endif

  if (w < 4) {
    // ================= the multiplying waves =================
    constexpr int TT = 2 * NQ;                          // 32-row tiles of this wave
    const int li = lane & 31, kg = lane >> 5;
    const int row0 = NQ == 2 ? 128 * (w & 1) : 64 * w;
    const int qb = NQ == 2 ? 2 * (w >> 1) : 0;          // first of its two 32-column tiles
    f32x16 acc[TT][2];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][q][i] = 0.f;
    const int a_frag = (row0 + li) * CG_ROW + 32 * kg;         // + 32 rows * t, + 16 s
    const int b_frag = (qb * 32 + li) * CG_ROW + 32 * kg;      // + term * NCOL rows, + 32 rows * q
    // The last eight MFMAs of a chunk (s = 1, term 0) wait in registers across the barrier and
    // run behind the next chunk's first fragment reads: the pipe works through the reads'
    // latency.  (Zero operands before the first chunk: acc + 0 * 0.)
    bf16x8 th[TT], tb[2];
#pragma unroll
    for (int t = 0; t < TT; ++t) th[t] = as_bf16x8(u32x4{0u, 0u, 0u, 0u});
#pragma unroll
    for (int q = 0; q < 2; ++q) tb[q] = as_bf16x8(u32x4{0u, 0u, 0u, 0u});
    auto mma = [&](const bf16x8* a, const bf16x8* b2) {
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b2[q], acc[t][q], 0, 0, 0);
    };
    lds_barrier();                                      // chunk 0 staged
    for (int j = 0; j < nch; ++j) {
      const int buf = j & 1;
      const unsigned char* ah = cf_smem + AHI + buf * CF_A_BYTES + a_frag;
      const unsigned char* al = cf_smem + ALO + buf * CF_A_BYTES + a_frag;
      const unsigned char* bb = cf_smem + BSM + buf * B_BYTES + b_frag;
      auto frags = [&](const unsigned char* a, bf16x8* f, int s_) {
#pragma unroll
        for (int t = 0; t < TT; ++t)
          f[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(a + t * 32 * CG_ROW + 16 * s_));
      };
      auto frags_b = [&](bf16x8 (*f)[2], int s_) {
#pragma unroll
        for (int term = 2; term >= 0; --term)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            f[term][q] = as_bf16x8(*reinterpret_cast<const u32x4*>(
                bb + (term * NCOL + q * 32) * CG_ROW + 16 * s_));
      };
      bf16x8 fh0[TT], fb0[3][2], fh1[TT], fb1[3][2], fl[TT];
      const int flag = lo_flag[buf * 4 + (lane & 3)];  // (first: its wait is the shortest)
      CT_FENCE;
      frags(ah, fh0, 0);
      frags_b(fb0, 0);
      CT_FENCE;
#if CT_EXP == 4 || CT_EXP == 6
      if (M < 0) {
#endif
      mma(th, tb);                                      // the previous chunk's last eight
      CT_FENCE;
      const bool need_lo = __builtin_amdgcn_readfirstlane(__any(flag));
      if (need_lo) frags(al, fl, 0);
      if (need_lo) mma(fl, fb0[2]);
      mma(fh0, fb0[2]);
      // its share of the W planes of chunk j + 1 (requested a chunk ago); chunk j + 2's
#if CT_EXP != 7
      store_b(buf ^ 1);
      load_b(k_begin + (j + 2) * CG_BK);
#endif
      if (need_lo) mma(fl, fb0[1]);
      mma(fh0, fb0[1]);
      CT_FENCE;
      frags(ah, fh1, 1);
      frags_b(fb1, 1);
      CT_FENCE;
      if (need_lo) mma(fl, fb0[0]);
      mma(fh0, fb0[0]);
      if (need_lo) frags(al, fl, 1);
      CT_FENCE;
      if (need_lo) mma(fl, fb1[2]);
      mma(fh1, fb1[2]);
      if (need_lo) mma(fl, fb1[1]);
      mma(fh1, fb1[1]);
      if (need_lo) mma(fl, fb1[0]);
#pragma unroll
      for (int t = 0; t < TT; ++t) th[t] = fh1[t];
#pragma unroll
      for (int q = 0; q < 2; ++q) tb[q] = fb1[0][q];
#if CT_EXP == 4 || CT_EXP == 6
      }
#endif
      CT_STAMP(0);
      lds_barrier();
      CT_STAMP(1);
    }
    mma(th, tb);
#if CT_PROF
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
      for (int k_ = 0; k_ < 8; ++k_) ct_prof[w * 8 + k_] = pacc[k_];
#endif
    float* dst = direct ? out : out + (size_t)blockIdx.y * M * ldo;
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int col = (qb + q) * 32 + li;
        if (col >= N) continue;
        const float bv = (direct && bias != nullptr) ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + row0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg;
          if (m < M) {
            float v = acc[t][q][r] + bv;
            if (direct && act == ACT_RELU) v = fmaxf(v, 0.f);
            dst[(size_t)m * ldo + col] = v;
          }
        }
      }
    return;
  }

  // ================= the staging waves =================
  __builtin_amdgcn_s_setprio(CT_PRIO);
  const int p = w - 4, pt = tid - 256;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if constexpr (!TILES) {
    // ---- x from the dense uint16 batch: thread = 16 bytes (8 counts) of rows r, r + 64, ... of
    //      the [256, 32] tile (4 lanes per 64-byte row segment); cut into hi / lo bf16 exactly as
    //      count_gemm_fwd_kernel does; a wave only ever writes its own rows of the planes ----
    const int prt = pt & 3, r = pt >> 2;
    const uint16_t* xsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xsrc[i] = X + (size_t)min(m0 + r + 64 * i, M - 1) * ldx + 8 * prt;
    const unsigned a_off = r * CG_ROW + prt * 16;         // + 64 i rows
    f32x4u raw[4];
    auto load_x = [&](int kc) {     // (clamped to the split's last chunk: unconditional loads)
      const int at = min(kc, k_last);
#pragma unroll
      for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const f32x4u*>(xsrc[i] + at);
    };
    bool lo_dirty[2] = {false, false};      // this wave's rows of lo plane 0 / 1 hold entries
    auto fill = [&](auto buf_tag) {
      constexpr int BUF = decltype(buf_tag)::value;
      const unsigned hi = AHI + BUF * CF_A_BYTES, lo = ALO + BUF * CF_A_BYTES;
      CT_STAMP(2);
      unsigned big = 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned wv[4] = {__float_as_uint(raw[i].x), __float_as_uint(raw[i].y),
                                __float_as_uint(raw[i].z), __float_as_uint(raw[i].w)};
        unsigned h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned u0 = __float_as_uint((float)(wv[j] & 0xFFFFu));
          const unsigned u1 = __float_as_uint((float)(wv[j] >> 16));
          h[j] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                  // upper halves
          big |= wv[j];
        }
        *reinterpret_cast<u32x4*>(cf_smem + hi + a_off + i * 64 * CG_ROW) =
            u32x4{h[0], h[1], h[2], h[3]};
      }
      CT_STAMP(3);
      // a count below 256 has no lo part; beyond that the exact test (any bit below the bf16 cut)
      const bool maybe = __builtin_amdgcn_readfirstlane(__any((int)((big & 0xFF00FF00u) != 0u)));
      bool need = false;
      if (maybe || lo_dirty[BUF]) {
        unsigned low = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned wv[4] = {__float_as_uint(raw[i].x), __float_as_uint(raw[i].y),
                                  __float_as_uint(raw[i].z), __float_as_uint(raw[i].w)};
          unsigned l[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float f0 = (float)(wv[j] & 0xFFFFu), f1 = (float)(wv[j] >> 16);
            const unsigned u0 = __float_as_uint(f0), u1 = __float_as_uint(f1);
            low |= u0 | u1;
            const float l0 = f0 - __uint_as_float(u0 & 0xFFFF0000u);
            const float l1 = f1 - __uint_as_float(u1 & 0xFFFF0000u);
            l[j] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
          }
          *reinterpret_cast<u32x4*>(cf_smem + lo + a_off + i * 64 * CG_ROW) =
              u32x4{l[0], l[1], l[2], l[3]};
        }
        need = __builtin_amdgcn_readfirstlane(__any((int)((low & 0xFFFFu) != 0u)));
        lo_dirty[BUF] = need;
      }
      if (lane == 0) lo_flag[BUF * 4 + p] = need ? 1 : 0;
      CT_STAMP(5);
    };
    if (nch > 0) {
      load_x(k_begin);
      fill(S0{});
      load_x(k_begin + CG_BK);
    }
    lds_barrier();
    auto step = [&](int j, auto buf_tag) {
      constexpr int BUF = decltype(buf_tag)::value;       // buffers of chunk j + 1
#if CT_EXP == 5 || CT_EXP == 6 || CT_EXP == 7
      if (M < 0) {
#endif
      fill(buf_tag);
      load_x(k_begin + (j + 2) * CG_BK);
      CT_STAMP(6);
      store_b(BUF);
      load_b(k_begin + (j + 2) * CG_BK);
#if CT_EXP == 5 || CT_EXP == 6 || CT_EXP == 7
      }
#endif
      CT_STAMP(0);
      lds_barrier();
      CT_STAMP(1);
    };
    for (int j = 0; j < nch; j += 2) {
      step(j, S1{});
      if (j + 1 < nch) step(j + 1, S0{});
    }
  } else {
  const unsigned l16 = lane & 15, quarter = lane & 48;
  // a quarter-wave per bucket
  const int bucket = 4 * p + (lane >> 4);
  const int cc = blockIdx.x * (CF_BM / CT_ROWS) + bucket;
  const bool live = cc < n_groups;
  const unsigned tp_off = (unsigned)(live ? cc : 0) * (unsigned)(ntp + 1);      // elements
  const unsigned a_base = (unsigned)(CT_ROWS * bucket) * CG_ROW;
  const unsigned dummy = DUM + 2u * pt;
  // tile pointers: lane l of a quarter-wave holds pointer c0 + l of its bucket
  int c0 = 0;
  unsigned win = 0u;
  auto load_window = [&](int tile) {
    c0 = tile;
    win = tptr[tp_off + (unsigned)min(tile + (int)l16, ntp)];
  };
  u32x4 E;                              // the lane's four of the chunk's entries
  unsigned Es = 0u, En = 0u;            // the bucket's first entry, its count | CT_LO
  // pointers of the chunk at kc -> (Es, En), its entries requested into E: an aligned 16 bytes
  // per lane, entries 4 (Es / 4 + l16) .. + 3
  auto load_entries = [&](int kc) {
    const int tile = min(kc, k_last) / CG_BK;
    if (tile - c0 >= CTF_WINDOW) load_window(tile);
    const int at = (int)((quarter + (unsigned)(tile - c0)) * 4u);
    const unsigned ps = (unsigned)__builtin_amdgcn_ds_bpermute(at, (int)win);
    const unsigned pe = (unsigned)__builtin_amdgcn_ds_bpermute(at + 4, (int)win);
    const unsigned s0 = ps & CT_MASK;
    const unsigned n = (live && kc < k_end) ? (pe & CT_MASK) - s0 : 0u;
    Es = s0; En = n | (((ps & CT_LO) && n) ? CT_LO : 0u);
    const unsigned first = ((s0 >> 2) + l16) << 2;
#if CT_EXP == 2
    E = *reinterpret_cast<const u32x4*>(ent + 4 * l16);
#else
    E = *reinterpret_cast<const u32x4*>(ent + (first < s0 + n ? first : 0u));
#endif
  };
  auto where = [&](unsigned e) {            // entry -> byte offset within its bucket's rows
    return __builtin_amdgcn_ubfe(e, 5, 4) * (unsigned)CG_ROW + ((e & 31u) << 1);
  };
  auto put16 = [&](unsigned addr, unsigned e) {
    *reinterpret_cast<uint16_t*>(cf_smem + addr) = (uint16_t)(e >> 16);
  };
  auto zero_rows = [&](unsigned plane) {    // this wave's 64 rows of a plane
    u32x4* z = reinterpret_cast<u32x4*>(cf_smem + plane + p * (64 * CG_ROW));
#pragma unroll
    for (int i = 0; i < 64 * CG_ROW / 16 / 64; ++i) z[lane + 64 * i] = u32x4{0u, 0u, 0u, 0u};
  };
  // where this lane wrote in hi buffer 0 / 1 (to take it out again)
  unsigned held[2][4];
  bool spilled[2] = {false, false};         // ... and the wave wrote more than that
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int c = 0; c < 4; ++c) held[b][c] = dummy;
  bool lo_dirty[2] = {false, false};        // this wave's rows of lo plane 0 / 1 hold entries
  // the chunk in (E, Es, En) -> buffers BUF
  auto fill = [&](auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
    const unsigned hi = AHI + BUF * CF_A_BYTES, lo = ALO + BUF * CF_A_BYTES;
    CT_STAMP(2);
    if (spilled[BUF]) zero_rows(hi);
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c) put16(held[BUF][c], 0u);
    }
    CT_STAMP(3);
    const unsigned n = En & CT_MASK;
    const unsigned d = (((Es >> 2) + l16) << 2) - Es;      // index of E.x in the bucket (may be < 0)
    const unsigned base = hi + a_base;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned e = E[c];
      const bool ok = (d + c) < n && !(e & (1u << 13));
      const unsigned at = ok ? base + where(e) : dummy;
      put16(at, e);
      held[BUF][c] = at;
    }
    const unsigned covered = 64u - (Es & 3u);
    const bool more = n > covered;
    if (more)                      // (a bucket beyond four per lane: rare, a direct loop)
      for (unsigned idx = covered + l16; idx < n; idx += 16u) {
        const unsigned e = ent[Es + idx];
        if (!(e & (1u << 13))) put16(base + where(e), e);
      }
    spilled[BUF] = (bool)__builtin_amdgcn_readfirstlane(__any((int)more));
    CT_STAMP(4);
    const bool need = (bool)__builtin_amdgcn_readfirstlane(__any((int)((En & CT_LO) != 0u)));
    if (lo_dirty[BUF]) { zero_rows(lo); lo_dirty[BUF] = false; }
    if (need) {
      for (unsigned idx = l16; idx < n; idx += 16u) {
        const unsigned e = ent[Es + idx];
        if (e & (1u << 13)) put16(lo + a_base + where(e), e);
      }
      lo_dirty[BUF] = true;
    }
    if (lane == 0) lo_flag[BUF * 4 + p] = need ? 1 : 0;
    CT_STAMP(5);
  };

  if (nch > 0) {
    load_window(k_begin / CG_BK);
    load_entries(k_begin);
    fill(S0{});
    load_entries(k_begin + CG_BK);
  }
  lds_barrier();
  // during chunk j: the x side of chunk j + 1 (requested a chunk ago) into the other buffers,
  // chunk j + 2 requested; the same for this thread's pieces of the W planes
  auto step = [&](int j, auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;       // buffers of chunk j + 1
#if CT_EXP == 5 || CT_EXP == 6 || CT_EXP == 7
    if (M < 0) {
#endif
    fill(buf_tag);
    load_entries(k_begin + (j + 2) * CG_BK);
    CT_STAMP(6);
    store_b(BUF);
    load_b(k_begin + (j + 2) * CG_BK);
#if CT_EXP == 5 || CT_EXP == 6 || CT_EXP == 7
    }
#endif
    CT_STAMP(0);
    lds_barrier();
    CT_STAMP(1);
  };
  for (int j = 0; j < nch; j += 2) {
    step(j, S1{});
    if (j + 1 < nch) step(j + 1, S0{});
  }
  }
#if CT_PROF
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
    for (int k_ = 0; k_ < 8; ++k_) ct_prof[w * 8 + k_] = pacc[k_];
#endif
}

// ---- weight gradient from tiles: the [512 genes, 16 cells] operand of a chunk scattered into
// LDS from one contiguous run of entries (the 16 buckets of gene block x group), eight waves of
// 64 genes; dA planes as in count_gemm_dw_kernel.  Built like the forward kernel above: three hi
// buffers, one lo buffer, one barrier per chunk, branch-free scatter, 32-bit offsets from uniform
// bases, the MFMAs in groups with the staging between them.
constexpr int CTD_NE = 2;
constexpr int CTD_A_BYTES = 512 * CD_ROW;
constexpr int CTD_DUMMY = 512 * 2;

template <int NT>
__global__ __launch_bounds__(512, 1) void count_tiles_dw_kernel(
    const uint32_t* __restrict__ ent, const uint32_t* __restrict__ gptr, int ngb, int M, int K,
    const uint16_t* __restrict__ T, int Kpad, int N, int k_chunk, float* __restrict__ out,
    int ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cd_smem[];
  constexpr int B_BYTES = 3 * CG_NP * CD_ROW;
  constexpr int AHI = 0, ALO = 3 * CTD_A_BYTES, BSM = 4 * CTD_A_BYTES, DUM = BSM + 2 * B_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kg = lane >> 5;
  const int gb = blockIdx.x;
  const int m_w = gb * 512 + w * CG_TM;
  const int k_begin = blockIdx.y * k_chunk;
  const int k_end = min(K, k_begin + k_chunk);
  const int k_last = k_end - CD_BK;

  f32x16 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][q][i] = 0.f;

  for (int i = tid; i < 4 * CTD_A_BYTES / 16; i += 512)
    reinterpret_cast<u32x4*>(cd_smem)[i] = u32x4{0u, 0u, 0u, 0u};
  auto zero_plane = [&](unsigned plane) {
    u32x4* z = reinterpret_cast<u32x4*>(cd_smem + plane);
#pragma unroll
    for (int i = 0; i < CTD_A_BYTES / 16 / 512; ++i) z[tid + 512 * i] = u32x4{0u, 0u, 0u, 0u};
  };
  const unsigned dummy = DUM + 2u * tid;
  constexpr int NPC = 2;                                // 768 pieces of dA over 512 threads
  unsigned boff[NPC], bst[NPC];
  bool bon[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int p2 = tid + 512 * i;                // 768 pieces: (term, column, half)
    const int row = p2 >> 1, part = p2 & 1;
    const int col = row & (CG_NP - 1), rowl = col < N ? row : row - col + (N - 1);
    bon[i] = p2 < 768 && col < NT * 32;
    boff[i] = ((unsigned)(bon[i] ? rowl : 0) * CD_BK + part * 8) * 2u;
    bst[i] = BSM + row * CD_ROW + part * 16;
  }
  struct Ptr { unsigned s, e; };
  auto load_ptr = [&](int kc) {
    const uint32_t* gp = gptr + (size_t)(min(kc, k_last) / CD_BK) * (ngb + 1) + gb;   // uniform
    Ptr p; p.s = gp[0]; p.e = gp[1];
    return p;
  };
  unsigned E[2][CTD_NE], Es[2], En[2];
  u32x4 breg[2][NPC];
  auto load_chunk = [&](int kc, Ptr p, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const unsigned s0 = p.s & CT_MASK;
    const unsigned n = kc < k_end ? (p.e & CT_MASK) - s0 : 0u;
    Es[SLOT] = s0; En[SLOT] = n | (((p.s & CT_LO) && n) ? CT_LO : 0u);
#pragma unroll
    for (int k = 0; k < CTD_NE; ++k) {
      const unsigned i = tid + 512u * k;
      E[SLOT][k] = ent[i < n ? s0 + i : 0u];
    }
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(T) +
                              (size_t)(min(kc, k_last) / CD_BK) * (3 * CG_NP * CD_BK * 2);   // uniform
#pragma unroll
    for (int i = 0; i < NPC; ++i)
      if (bon[i]) breg[SLOT][i] = *reinterpret_cast<const u32x4*>(tb + boff[i]);
  };
  // entry -> byte offset of (gene of the block, cell of the chunk) within a plane
  auto where = [&](unsigned e) {
    const unsigned gene = ((e >> 4) & 0x1E0u) | (e & 31u);          // (tile & 15) * 32 + gene
    return gene * (unsigned)CD_ROW + ((e >> 4) & 0x1Eu);            // + 2 * row
  };
  auto put16 = [&](unsigned addr, unsigned e) {
    *reinterpret_cast<uint16_t*>(cd_smem + addr) = (uint16_t)(e >> 16);
  };
  auto scatter_hi = [&](unsigned plane, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const unsigned n = En[SLOT] & CT_MASK;
#pragma unroll
    for (int k = 0; k < CTD_NE; ++k) {
      const unsigned e = E[SLOT][k];
      const bool ok = (tid + 512u * k < n) && !(e & (1u << 13));
      put16(ok ? plane + where(e) : dummy, e);
    }
    if (n > 512u * CTD_NE)
      for (unsigned i = tid + 512u * CTD_NE; i < n; i += 512u) {
        const unsigned e = ent[Es[SLOT] + i];
        if (!(e & (1u << 13))) put16(plane + where(e), e);
      }
  };
  auto scatter_lo = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const unsigned n = En[SLOT] & CT_MASK;
#pragma unroll
    for (int k = 0; k < CTD_NE; ++k) {
      const unsigned e = E[SLOT][k];
      if ((tid + 512u * k < n) && (e & (1u << 13))) put16(ALO + where(e), e);
    }
    if (n > 512u * CTD_NE)
      for (unsigned i = tid + 512u * CTD_NE; i < n; i += 512u) {
        const unsigned e = ent[Es[SLOT] + i];
        if (e & (1u << 13)) put16(ALO + where(e), e);
      }
  };
  auto store_b = [&](int buf, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
    for (int i = 0; i < NPC; ++i)
      if (bon[i])
        *reinterpret_cast<u32x4*>(cd_smem + bst[i] + buf * B_BYTES) = breg[SLOT][i];
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  Ptr pn = Ptr{0u, 0u};
  __syncthreads();
  if (k_begin < k_end) {
    load_chunk(k_begin, load_ptr(k_begin), S0{});
    load_chunk(k_begin + CD_BK, load_ptr(k_begin + CD_BK), S1{});
    pn = load_ptr(k_begin + 2 * CD_BK);
    scatter_hi(AHI, S0{});
    store_b(0, S0{});
  }
  __syncthreads();

  const int a_frag = (64 * w + li) * CD_ROW + 16 * kg;       // + 32 genes * t
  const int b_frag = li * CD_ROW + 16 * kg;
  bool lo_dirty = false;
  int ab = 0;
  auto chunk = [&](int kc, auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
    using Other = std::integral_constant<int, BUF ^ 1>;
    const int ab1 = ab == 2 ? 0 : ab + 1, ab2 = ab1 == 2 ? 0 : ab1 + 1;
    const bool need_lo = (En[BUF] & CT_LO) != 0u;        // (uniform: the block pointer's flag)
    if (lo_dirty || need_lo) {
      if (lo_dirty) { zero_plane(ALO); lds_barrier(); }
      if (need_lo) { scatter_lo(buf_tag); lds_barrier(); }
      lo_dirty = need_lo;
    }
    const unsigned char* ah = cd_smem + AHI + ab * CTD_A_BYTES + a_frag;
    const unsigned char* al = cd_smem + ALO + a_frag;
    const unsigned char* bcur = cd_smem + BSM + BUF * B_BYTES + b_frag;
    bf16x8 fh[2], fl[2], fr[3][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
      fh[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(ah + t * 32 * CD_ROW));
#pragma unroll
    for (int term = 2; term >= 0; --term)
#pragma unroll
      for (int q = 0; q < NT; ++q)
        fr[term][q] = as_bf16x8(*reinterpret_cast<const u32x4*>(
            bcur + (term * CG_NP + q * 32) * CD_ROW));
    if (need_lo) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
        fl[t] = as_bf16x8(*reinterpret_cast<const u32x4*>(al + t * 32 * CD_ROW));
    }
    auto mm = [&](int term) {
      if (need_lo) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[0], fr[term][q], acc[0][q], 0, 0, 0);
          acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[1], fr[term][q], acc[1][q], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[0], fr[term][q], acc[0][q], 0, 0, 0);
        acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[1], fr[term][q], acc[1][q], 0, 0, 0);
      }
    };
    zero_plane(AHI + ab2 * CTD_A_BYTES);
    {
      const Ptr p = pn;
      pn = load_ptr(kc + 3 * CD_BK);
      load_chunk(kc + 2 * CD_BK, p, buf_tag);      // (overwrites En[BUF]: need_lo was read above)
    }
    CT_FENCE;
    mm(2);
    CT_FENCE;
    scatter_hi(AHI + ab1 * CTD_A_BYTES, Other{});
    CT_FENCE;
    mm(1);
    CT_FENCE;
    store_b(BUF ^ 1, Other{});
    CT_FENCE;
    mm(0);
    CT_FENCE;
    lds_barrier();
    ab = ab1;
  };
  for (int kc = k_begin; kc < k_end; kc += 2 * CD_BK) {
    chunk(kc, S0{});
    if (kc + CD_BK < k_end) chunk(kc + CD_BK, S1{});
  }

  float* dst = out + (size_t)blockIdx.y * M * ldo;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int col = q * 32 + li;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_w + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (m < M) dst[(size_t)m * ldo + col] = acc[t][q][r];
      }
    }
}

#if CT_PROF
extern "C" int scvae_ct_prof_dump(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ct_prof), sizeof(unsigned long long) * 64);
}
#endif
static size_t ctd_lds_bytes() {
  return 4 * (size_t)CTD_A_BYTES + 2 * (size_t)(3 * CG_NP * CD_ROW) + CTD_DUMMY;
}

// count_gemm_u16 with the large part of the contraction read from `tiles` (the same minibatch:
// csr_count_tiles of the rows x was densified from).  x (the uint16 batch) is only read for the
// K % chunk leftover terms of the reduce kernel; the split, the k ranges of the slabs and their
// fixed-order sum are those of count_gemm_u16: bit-identical results.
int count_gemm_tiles(hipStream_t stream, int mode, CountTiles tiles, const uint16_t* x, int ldx,
                     int rows, int cols, const float* other, int ld_other, int N,
                     const float* bias, int act, float* C, int ldc, void* workspace,
                     size_t workspace_bytes) {
  SCVAE_ARG(tiles.ent && tiles.tptr && tiles.gptr && other && C && workspace);
  SCVAE_ARG(mode == 0 || mode == 1);
  SCVAE_ARG(count_gemm_supported(N) && ld_other >= N && ldc >= N && count_tiles_supported(cols));
  if (rows == 0 || cols == 0) return 0;
  SCVAE_ARG(workspace_bytes >= count_gemm_workspace_bytes(mode, rows, cols, N));
  SCVAE_ARG(((uintptr_t)workspace & 15) == 0);
  const int M = mode == 0 ? rows : cols, K = mode == 0 ? cols : rows;
  const int bk = cg_bk(mode);
  const int k_main = K / bk * bk;
  SCVAE_ARG(k_main == K || (x && ldx >= cols));      // the leftover terms come from the dense batch
  const int Kpad = cg_kpad(K);
  uint16_t* T = reinterpret_cast<uint16_t*>(workspace);
  const size_t t_bytes = ((size_t)3 * CG_NP * Kpad * sizeof(uint16_t) + 255) / 256 * 256;
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + t_bytes);
  const int ntp = count_tiles_padded(cols), ngb = ntp / CT_BLOCK;
  const int n_groups = (rows + CT_ROWS - 1) / CT_ROWS;
  int splits = 0;
  if (k_main > 0) {
    hipLaunchKernelGGL(split3_transpose_kernel, dim3((Kpad / 8 + 1) / 2), dim3(256), 0, stream,
                       other, K, N, ld_other, T, Kpad, bk);
    SCVAE_LAUNCH_CHECK("split3_transpose_kernel");
    splits = cg_splits(mode, M, k_main);
    int k_chunk = k_main;
    if (splits > 1) {
      k_chunk = (k_main + splits - 1) / splits;
      k_chunk = (k_chunk + bk - 1) / bk * bk;
      splits = (k_main + k_chunk - 1) / k_chunk;
    }
    const bool direct = mode == 0 && splits == 1 && k_main == K;
    float* dst = direct ? C : slabs;
    const int ldo = direct ? ldc : N;
    const int NT = (N + 31) / 32;
    const float* kbias = direct ? bias : nullptr;
    const int kact = direct ? act : (int)ACT_NONE, kdirect = direct ? 1 : 0;
    if (mode == 0) {
      const dim3 grid((M + CF_BM - 1) / CF_BM, splits);
      const int NQ = NT > 2 ? 2 : 1;
      const size_t lds = ctf_lds_bytes(NQ);
#define SCVAE_CTF(NQ_)                                                                           \
  do {                                                                                           \
    auto kfn = count_fwd2_kernel<NQ_, true>;                                                     \
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)lds));                    \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, stream, tiles.ent, tiles.tptr, ntp, n_groups,  \
                       (const uint16_t*)nullptr, 0, M, k_main, T, Kpad, N, k_chunk, dst, ldo,    \
                       kbias, kact, kdirect);                                                    \
  } while (0)
      if (NQ == 2) SCVAE_CTF(2); else SCVAE_CTF(1);
#undef SCVAE_CTF
    } else {
      const dim3 grid((M + 511) / 512, splits);
      const size_t lds = ctd_lds_bytes();
#define SCVAE_CTD(NT_)                                                                           \
  do {                                                                                           \
    auto kfn = count_tiles_dw_kernel<NT_>;                                                       \
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)lds));                    \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, stream, tiles.ent, tiles.gptr, ngb, M, k_main, \
                       T, Kpad, N, k_chunk, dst, ldo);                                           \
  } while (0)
      switch (NT) { case 1: SCVAE_CTD(1); break; case 2: SCVAE_CTD(2); break;
                    case 3: SCVAE_CTD(3); break; default: SCVAE_CTD(4); }
#undef SCVAE_CTD
    }
    SCVAE_LAUNCH_CHECK("count_tiles_kernel");
    if (direct) return 0;
  }
  const size_t total = (size_t)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(count_gemm_reduce_kernel<uint16_t>, dim3(blocks), dim3(256), 0, stream, slabs,
                     bias, C, M, N, ldc, splits, act, mode, x, ldx, other, ld_other, k_main, K);
  SCVAE_LAUNCH_CHECK("count_gemm_reduce_kernel");
  return 0;
}

// ---- precondition check: every value an integer in [0, 65536) ----
__global__ __launch_bounds__(256) void check_counts_kernel(const float* __restrict__ v, size_t n,
                                                           int* __restrict__ bad) {
  int local = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float x = v[i];
    if (!(x >= 0.f && x < 65536.f && x == __builtin_truncf(x))) local = 1;
  }
  if (__any(local) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

int check_counts(hipStream_t stream, const float* values, size_t n, int* bad) {
  SCVAE_ARG(bad && (values || n == 0));
  SCVAE_HIP(hipMemsetAsync(bad, 0, sizeof(int), stream));
  if (n == 0) return 0;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(check_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, values, n,
                     bad);
  SCVAE_LAUNCH_CHECK("check_counts_kernel");
  return 0;
}

}  // namespace scvae
