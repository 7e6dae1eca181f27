// Two-half software-pipelined variant of the fused decoder output layer (see decoder_fused.hip
// for the maths and the reference lines).  Same inputs, outputs and per-strip partial buffers.
//
// One workgroup = 16 waves = two halves of 8 waves that share the strip's weights in LDS and walk
// over alternating 32-row tiles.  A half alternates between
//     an MFMA slot:  GEMM2 dW_j += d^T G_j (8 waves), then GEMM3 dd = sum_j G_j W_j^T (waves 0-3)
//                    of tile k-1 next to GEMM1 pre_j = d W_j of tile k (waves 4-7, result kept in
//                    registers); the HBM loads of t (tile k) and d (tile k+1) are issued here;
//     a hand-over:   pre_j + b_j and the d tile k+1 go to LDS (short);
//     a VALU slot:   likelihood epilogue of tile k (all 8 waves): G_j -> LDS, ll partials;
// and half B runs two slots behind half A, so that while one half feeds the MFMA pipe (two MFMA
// waves per SIMD) the other half does the VALU / LDS work of the epilogue (two VALU waves per
// SIMD): the matrix cores do not wait for the likelihood math any more.  The only synchronisation
// is one workgroup barrier per slot; the d tile is double buffered.
//
// Two job tables for the matrix work of a half (see the kernel): the default one, and -- REM
// instantiations, chosen by the launcher for 96 < H <= 111 from 512 rows on -- one in which the h
// remainder of the two backward products runs on 16-wide MFMA tiles and the jobs are dealt
// evenly over the SIMDs.
//
// The t > 0 corrections of the negative-binomial kinds (lgamma / digamma differences; 5 % of a
// count matrix) are compacted per wave with ballots into a wave-private LDS queue, so that the
// expensive code runs once per wave on dense lanes instead of four times on sparse lanes; no
// LDS atomics, summation order fixed (deterministic).
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"
#include "likelihood.hpp"

namespace scvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4m __attribute__((ext_vector_type(4)));

constexpr int D2_THREADS = 1024;  // 16 waves: 4 per SIMD
constexpr int D2_HALF = 512;
constexpr int D2_BN = 64;         // columns (genes) per workgroup
constexpr int D2_LD = D2_BN + 1;  // odd LDS row stride
constexpr int D2_BM = 32;         // rows per tile (per half)
constexpr int D2_QCAP = 64;       // wave-private queue capacity (t > 0 elements per 256)

__host__ __device__ inline int d2_ldd(int H) { return H | 1; }

static size_t d2_half_floats(int P, int H) {
  return 2 * ((size_t)D2_BM * d2_ldd(H) + 32)  // dsh x 2 (+ slack for the padded h tile)
         + (size_t)P * D2_BM * D2_LD           // Gs
         + 8 * 4 * D2_QCAP;                    // 8 wave-private queues x 4 fields
}

size_t decoder_fused2_lds_bytes(int P, int H) {
  size_t floats = (size_t)P * H * D2_LD + 3 * D2_BN + 2 * d2_half_floats(P, H) + 64;
  // over-reads of the padded h tiles (h up to 127) must stay inside the allocation
  const size_t need = (size_t)((P - 1) * H + 128) * D2_LD + 64;
  if (floats < need) floats = need;
  // the final dW combine parks half B's accumulators: P x 8 waves x 16 x 64 floats
  const size_t park = (size_t)P * 8 * 16 * 64;
  if (floats < park) floats = park;
  return floats * sizeof(float);
}

bool decoder_fused2_supported(int P, int H) {
  return P <= 2 && H >= 2 && H <= 126 && (H % 2) == 0 &&
         decoder_fused2_lds_bytes(P, H) <= 160 * 1024;
}

__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ void lds_wave_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int KIND, bool TRAIN, bool REM = false>
__global__ __launch_bounds__(D2_THREADS) void decoder_head2_kernel(
    const float* __restrict__ d, int R, int H, unsigned magic_h, HeadParams hp, int F,
    Targets tg, int B, const float* __restrict__ gw, int inline_lgamma,
    float* __restrict__ ll_part, float* __restrict__ dd_part) {
  using Traits = LikelihoodTraits<KIND>;
  constexpr int P = Traits::P;
  constexpr int BN = D2_BN, LD = D2_LD, BM = D2_BM, NH = D2_HALF;
  constexpr int DLOADS = (BM * 126 + NH - 1) / NH;   // upper bound of d elements per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDD = d2_ldd(H);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches
  const int half = w >> 3, hw = w & 7, th = tid & (NH - 1);
  const int kh = lane >> 5, li = lane & 31;
  const int c0 = blockIdx.x * BN;

  float* Ws = smem;                                   // [P][H][LD]
  float* bs = Ws + (size_t)P * H * LD;                // [3][BN]
  const size_t DBUF = (size_t)BM * LDD + 32;          // one d tile (+32 slack)
  const size_t half_floats = 2 * DBUF + (size_t)P * BM * LD + 8 * 4 * D2_QCAP;
  float* hbase = bs + 3 * BN + (size_t)half * half_floats;
  float* dsh = hbase;                                 // [2][BM][LDD]; column H = 1
  float* Gs = dsh + 2 * DBUF;                         // [P][BM][LD]: pre_j, then G_j

  // ---- strip weights and biases -> LDS (once): eight HBM loads in flight per thread ----
  {
    const int c = tid & (BN - 1);
    const bool col_ok = c0 + c < F;
    constexpr int RPP = D2_THREADS / 64;                        // rows of 64 genes per pass
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
  // ---- which matrix jobs this wave runs (hw = wave within its half; SIMD = hw & 3) ----
  // Default: GEMM2 tile (h-tile hw >> 1, column tile hw & 1) on every wave, GEMM3 h-tile hw on
  // waves 0-3, GEMM1 job hw - 4 on waves 4-7: 178 MFMA-equivalents per SIMD and tile when H pads
  // to 128.  When the h remainder (rows 96 .. H of dW incl. the ones-row, columns 96 .. H - 1 of
  // dd) fits a 16-wide tile (96 < H <= 111, e.g. the default 100) it runs on
  // v_mfma_f32_16x16x4_f32 tiles instead of a mostly empty 32-wide one, and the jobs are dealt
  // so that every SIMD carries 160-164 instead of 178 (32x32x2 equivalents per tile, P = 2):
  //   SIMD 0: w0 GEMM3 h0 + GEMM2 (0,0)    w4 GEMM3 remainder + GEMM2 (0,1)
  //   SIMD 1: w1 GEMM3 h1 + GEMM2 (1,0)    w5 GEMM1 job 0    + GEMM2 remainder, columns 0-31
  //   SIMD 2: w2 GEMM3 h2 + GEMM2 (1,1)    w6 GEMM1 job 1    + GEMM2 remainder, columns 32-63
  //   SIMD 3: w3 GEMM1 job 2 + GEMM2 (2,0) w7 GEMM1 job 3    + GEMM2 (2,1)
  // REM is chosen by the launcher: 96 < H <= 111, at most two heads, from a few hundred rows on
  // (with the four tiles of a 100-cell minibatch the dealt schedule is 9 us slower: 104 against
  // 95 us).  In a REM kernel EVERY GEMM2 tile runs as 16x16x4 sub-tiles (same MFMA time and LDS
  // reads as 32x32x2), so that the dW accumulators have one shape and one definition site: with
  // two MFMA shapes writing them the register allocator kept a second copy of all 32 across the
  // tile loop.
  constexpr bool rem16 = REM;
  static_assert(!REM || (TRAIN && P <= 2), "remainder schedule: training kernels, two heads");
  int g2_h0 = (hw >> 1) * 32, g2_n0 = (hw & 1) * 32;
  bool g2_rem = false;
  int g3_h0 = hw * 32;
  bool g3_full = hw < 4 && hw * 32 < H, g3_rem = false;
  int g1_job = hw - 4;                       // GEMM1: head g1_job >> 1, column tile g1_job & 1
  if (rem16) {
    const int ht = (0x20002110 >> (4 * hw)) & 15;     // h-tile of the full GEMM2 tile (w5, w6: -)
    const int nt = (0xD4 >> hw) & 1;                  // column tile: w2, w4, w6, w7 -> 1
    g2_rem = hw == 5 || hw == 6;
    g2_h0 = g2_rem ? 96 : ht * 32;
    g2_n0 = nt * 32;
    g3_full = hw < 3;
    g3_rem = hw == 4;
    g1_job = hw == 5 ? 0 : hw == 6 ? 1 : hw == 3 ? 2 : hw == 7 ? 3 : -1;
  }
  f32x16 accW[P];            // default schedule: one 32x32 tile per head
  f32x4m accQ[P][4];         // REM: its four 16x16 sub-tiles [2 hi + cj] (remainder job: hi = 0)
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) accW[j][i] = 0.f;
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) accQ[j][q] = f32x4m{0.f, 0.f, 0.f, 0.f};

  // next d tile, in flight from the MFMA slot to the hand-over.  H % 4 == 0 (the usual case: rows
  // of d are 16-byte multiples, a 32-row tile is one contiguous block): two 16-byte loads per
  // thread and one row / column split per load instead of seven scalar loads with one each.
  constexpr int DREGS = DLOADS > 8 ? DLOADS : 8;
  float dv[DREGS];
  float tv[4] = {0.f, 0.f, 0.f, 0.f}, up[2] = {0.f, 0.f};   // in flight until the VALU slot
  const bool h4 = (H & 3) == 0;

  // `tq` is an opaque copy of the thread's index within its half: the per-thread addresses are
  // re-derived in every slot instead of living in (spilled) registers across the whole loop.
  auto load_d = [&](int m0, int tq) {
    const int n_valid = (m0 < R) ? min(R - m0, BM) * H : 0;
    if (h4) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const float* dbase = d + (size_t)m0 * H + 4 * tq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (4 * (i * NH + tq) < n_valid) v = *reinterpret_cast<const f32x4*>(dbase + 4 * i * NH);
        dv[4 * i] = v.x; dv[4 * i + 1] = v.y; dv[4 * i + 2] = v.z; dv[4 * i + 3] = v.w;
      }
    } else {
      const float* dbase = d + (size_t)m0 * H + tq;
#pragma unroll
      for (int i = 0; i < DLOADS; ++i) dv[i] = (i * NH + tq < n_valid) ? dbase[i * NH] : 0.f;
    }
  };
  auto store_d = [&](int m0, int buf, int tq) {
    float* dst = dsh + (size_t)buf * DBUF;
    const int n_d = BM * H;
    if (h4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = 4 * (i * NH + tq);
        if (e < n_d) {
          const int r = __umulhi((unsigned)e, magic_h), h = e - r * H;
          float* q = dst + r * LDD + h;
          q[0] = dv[4 * i]; q[1] = dv[4 * i + 1]; q[2] = dv[4 * i + 2]; q[3] = dv[4 * i + 3];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < DLOADS; ++i) {
        const int e = i * NH + tq;
        if (e < n_d) {
          const int r = __umulhi((unsigned)e, magic_h), h = e - r * H;
          dst[r * LDD + h] = dv[i];
        }
      }
    }
    if (tq < BM) dst[tq * LDD + H] = (m0 + tq < R) ? 1.f : 0.f;
  };

  const int n_tiles = (R + BM - 1) / BM;
  const int n_a = (n_tiles + 1) / 2;         // tiles of half A (half B: the odd ones)

  // both halves stage their first tile (tile index = half) into d buffer 0
  load_d(half * BM, th);
  store_d(half * BM, 0, th);
  __syncthreads();

  const bool g1_wave = g1_job >= 0 && g1_job < 2 * P;

  // Slot schedule (one workgroup barrier after every slot; the slot order is static so that the
  // register allocator sees the short live ranges of accX / dv / tv):
  //   A:  M(0) H(0) V(0)  -   M(1) H(1) V(1)  -   ...
  //   B:   -    -   M(0) H(0) V(0)  -   M(1) H(1) V(1) ...
  if (half) {
    lds_barrier();
    lds_barrier();
  }
  const int n_iter = n_a + 1;                // the last iteration drains GEMM2/GEMM3 of the last tile
  for (int k = 0; k < n_iter; ++k) {
    const int m0 = (2 * k + half) * BM;      // tile k of this half
    const int mp = m0 - 2 * BM;              // tile k-1
    const bool live = m0 < R;
    const bool live_prev = (k >= 1) && (mp < R);
    // GEMM3 / GEMM1 accumulator of this tile, held until the hand-over (declared per tile: a
    // value carried round the loop would be copied through every path that does not set it)
    f32x16 accX;
    {
      // ============ MFMA slot: GEMM2 + GEMM3 of tile k-1, GEMM1 of tile k ============
      __builtin_amdgcn_s_setprio(3);   // MFMA waves first; the other half's VALU work fills the gaps
      const int tq = opaque(th);
      const int li = tq & 31, kh = (tq >> 5) & 1;
      const float* dprev = dsh + (size_t)((k - 1) & 1) * DBUF;
      const float* dcur = dsh + (size_t)(k & 1) * DBUF;
      // ---- GEMM2: dW_j[h, col] += sum_row d[row, h] G_j[row, col]; row h == H gives db_j ----
      if constexpr (REM) {
        // every job as 16x16 sub-tiles (hi, cj) of its h rows x 32 columns, four rows per step:
        // a full job has two h sub-tiles, the remainder job (rows 96 .. 111, 96 .. H live) one.
        // (The four k-lanes of the 16x16x4 operands may take any four rows as long as A and B
        //  agree: lane group k4 walks rows 8 k4 .. 8 k4 + 7, which spreads the banks.)
        if (live_prev) {
          const int i16 = tq & 15, k4 = (tq >> 4) & 3;
          const float* ap = dprev + 8 * k4 * LDD + g2_h0 + i16;       // A[i=h][k=row]
          const float* bp = Gs + 8 * k4 * LD + g2_n0 + i16;           // B[k=row][n=col]
          // h sub-tile 0 on every wave, then h sub-tile 1 where the job has one: each
          // accumulator is written in exactly one place (the B operands are read twice)
#pragma unroll
          for (int kk = 0; kk < BM / 4; ++kk) {
            const float a0 = ap[kk * LDD];
#pragma unroll
            for (int j = 0; j < P; ++j) {
              const float b0 = bp[(j * BM + kk) * LD], b1 = bp[(j * BM + kk) * LD + 16];
              accQ[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, accQ[j][0], 0, 0, 0);
              accQ[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, accQ[j][1], 0, 0, 0);
            }
          }
          if (!g2_rem) {
#pragma unroll
            for (int kk = 0; kk < BM / 4; ++kk) {
              const float a1 = ap[kk * LDD + 16];
#pragma unroll
              for (int j = 0; j < P; ++j) {
                const float b0 = bp[(j * BM + kk) * LD], b1 = bp[(j * BM + kk) * LD + 16];
                accQ[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, accQ[j][2], 0, 0, 0);
                accQ[j][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, accQ[j][3], 0, 0, 0);
              }
            }
          }
        }
      } else if (TRAIN && live_prev && g2_h0 <= H) {
        const float* ap = dprev + kh * LDD + g2_h0 + li;              // A[i=h][k=row]
        const float* bp = Gs + kh * LD + g2_n0 + li;                  // B[k=row][n=col]
#pragma unroll 8
        for (int kk = 0; kk < BM; kk += 2) {
          const float a = ap[kk * LDD];
#pragma unroll
          for (int j = 0; j < P; ++j)
            accW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[(j * BM + kk) * LD], accW[j], 0, 0,
                                                           0);
        }
      }
      // ---- HBM loads: d tile k+1 (stored in the hand-over), t / upstream of tile k (VALU slot);
      //      they fly under GEMM3 / GEMM1 ----
      load_d(m0 + 2 * BM, tq);
      if (live) {
        const int ec = tq & 31, er0 = tq >> 5;
        // (fp32 batch or the uint16 minibatch: one uniform branch around the loads)
        auto load_t = [&](auto* t) {
        const int ldt = tg.ld;
        if (m0 + BM <= R && c0 + BN <= F && R == B) {   // full tile, no row wrap: no predicates
          auto* tp = t + (size_t)(m0 + er0) * ldt + c0 + ec;
          tv[0] = target_raw(tp[0]);
          tv[1] = target_raw(tp[32]);
          tv[2] = target_raw(tp[(size_t)16 * ldt]);
          tv[3] = target_raw(tp[(size_t)16 * ldt + 32]);
          if (TRAIN) {
            up[0] = gw[m0 + er0];
            up[1] = gw[m0 + er0 + 16];
          }
        } else {
#pragma unroll
          for (int ri = 0; ri < 2; ++ri) {
            const int grow = m0 + er0 + 16 * ri;
            const bool rok = grow < R;
            up[ri] = (TRAIN && rok) ? gw[grow] : 0.f;
            auto* trow = t + (size_t)(rok ? grow % B : 0) * ldt + c0;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
              const int c = ec + 32 * ci;
              tv[ri * 2 + ci] = (rok && c0 + c < F) ? target_raw(trow[c]) : 0.f;
            }
          }
        }
        };
        if (tg.u16) load_t(static_cast<const uint16_t*>(tg.p));
        else load_t(static_cast<const float*>(tg.p));
      }
      if (g3_rem) {
        // ---- GEMM3, columns h = 96 .. 111 (96 .. H - 1 live): two 16-row tiles, four gene
        //      columns per step ----
        if (TRAIN && live_prev) {
          const int i16 = tq & 15, k4 = (tq >> 4) & 3;
          f32x4m x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < P; ++j) {
            // (lane group k4 walks gene columns 16 k4 .. 16 k4 + 15: conflict-free with the
            //  odd row stride)
            const float* ap = Gs + (j * BM + i16) * LD + 16 * k4;     // A[i=row][k=col]
            const float* bp = Ws + (j * H + 96 + i16) * LD + 16 * k4; // B[k=col][n=h]
#pragma unroll 8
            for (int kk = 0; kk < BN / 4; ++kk) {
              const float b = bp[kk];
              x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk], b, x0, 0, 0, 0);
              x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[16 * LD + kk], b, x1, 0, 0, 0);
            }
          }
          {   // accX[0..3] = x0, accX[4..7] = x1, the rest unused on this wave
            const f32x16 t0 = __builtin_shufflevector(x0, x0, 0, 1, 2, 3, -1, -1, -1, -1, -1, -1, -1,
                                                      -1, -1, -1, -1, -1);
            const f32x16 t1 = __builtin_shufflevector(x1, x1, 0, 1, 2, 3, -1, -1, -1, -1, -1, -1, -1,
                                                      -1, -1, -1, -1, -1);
            accX = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 16, 17, 18, 19, -1, -1, -1, -1, -1,
                                           -1, -1, -1);
          }
        }
      } else if (g1_job < 0) {
        // ---- GEMM3: dd[row, h] = sum_j sum_col G_j[row, col] W_j[h, col] ----
        if (TRAIN && live_prev && g3_full) {
          const int h0 = g3_h0;
#pragma unroll
          for (int i = 0; i < 16; ++i) accX[i] = 0.f;
#pragma unroll
          for (int j = 0; j < P; ++j) {
            const float* ap = Gs + (j * BM + li) * LD + kh;           // A[i=row][k=col]
            const float* bp = Ws + (j * H + h0 + li) * LD + kh;       // B[k=col][n=h]
#pragma unroll 8
            for (int kk = 0; kk < BN; kk += 2)
              accX = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], accX, 0, 0, 0);
          }
        }
      } else if (g1_wave && live) {
        // ---- GEMM1: pre_j = d W_j (+ b_j in the hand-over), one 32x32 tile per wave ----
        const int j = g1_job >> 1, nt = g1_job & 1;
#pragma unroll
        for (int i = 0; i < 16; ++i) accX[i] = 0.f;
        const float* arow = dcur + li * LDD + kh;                     // A[i=row][k=h]
        const float* bcol = Ws + (j * H + kh) * LD + nt * 32 + li;    // B[k=h][n=col]
        int kk = 0;
        for (; kk + 20 <= H; kk += 20) {
#pragma unroll
          for (int u = 0; u < 20; u += 2)
            accX = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[kk + u], bcol[(kk + u) * LD], accX, 0,
                                                        0, 0);
        }
        for (; kk < H; kk += 2)
          accX = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[kk], bcol[kk * LD], accX, 0, 0, 0);
      }
      // the d loads have landed by now: consume the wait here, before the dd stores are issued,
      // so that the hand-over does not wait for the stores
#pragma unroll
      for (int i = 0; i < DREGS; ++i) asm volatile("" : "+v"(dv[i]));
      // (non-temporal stores: 839 MB of dd slabs per launch that are read exactly once, by
      //  dd_reduce_kernel -- written through, they do not linger as dirty lines in the 256 MB
      //  infinity cache and get evicted in the middle of that reduce: 151 -> 123 us)
      if (TRAIN && live_prev && g3_rem) {
        const int i16 = tq & 15, k4 = (tq >> 4) & 3;
        const int h = 96 + i16;
        if (h < H) {
          float* dst = dd_part + ((size_t)blockIdx.x * R + mp + 4 * k4) * H + h;
          if (mp + BM <= R) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                __builtin_nontemporal_store(accX[4 * rt + r], dst + (16 * rt + r) * H);
          } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (mp + 16 * rt + 4 * k4 + r < R)
                  __builtin_nontemporal_store(accX[4 * rt + r], dst + (16 * rt + r) * H);
          }
        }
      } else if (TRAIN && live_prev && g3_full) {
        const int h = g3_h0 + li;
        if (h < H) {
          float* dst = dd_part + ((size_t)blockIdx.x * R + mp + 4 * kh) * H + h;
          if (mp + BM <= R) {
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(accX[r], dst + ((r & 3) + 8 * (r >> 2)) * H);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ro = (r & 3) + 8 * (r >> 2);
              if (mp + 4 * kh + ro < R) __builtin_nontemporal_store(accX[r], dst + ro * H);
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    lds_barrier();
    {
      // ============ hand-over (short): d tile k+1 and pre_j of tile k -> LDS ============
      const int tq = opaque(th);
      const int li = tq & 31, kh = (tq >> 5) & 1;
      store_d(m0 + 2 * BM, (k + 1) & 1, tq);
      if (g1_wave && live) {
        const int j = g1_job >> 1, nt = g1_job & 1;
        const float bv = bs[j * BN + nt * 32 + li];
        float* out = Gs + (j * BM + 4 * kh) * LD + nt * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2)) * LD] = accX[r] + bv;
      }
    }
    lds_barrier();
    // ============ VALU slot: likelihood epilogue of tile k ============
    // 2 rows x 2 columns per thread (one 128-byte row segment per half wave: coalesced t loads,
    // conflict-free LDS accesses with the odd stride); G_j written in place of pre_j
    // (full tiles take the copy without the row / column bound checks)
    auto epilogue = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      const int tq = opaque(th);
      const int lane = tq & 63;
      const int ec = tq & 31, er0 = tq >> 5;
      float* q0 = Gs + P * BM * LD + ((tq >> 6) & 7) * 4 * D2_QCAP;   // wave-private queue: r -> A
      float* q1 = q0 + D2_QCAP;                                        // t
      float* q2 = q1 + D2_QCAP;                                        // upstream * gate
      int* q3 = reinterpret_cast<int*>(q2 + D2_QCAP);                  // LDS offset row*LD + col
      float lsum[2] = {0.f, 0.f};
      int slot[4];
      int q_base = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
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
            const int s = q_base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (s < D2_QCAP) {
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
          q_base += __popcll(mask);
        } else if (nz && inline_lgamma) {
          lp -= lgamma1p(tval);
        }
        lsum[ri] += ok ? lp : 0.f;
        if (TRAIN) {
#pragma unroll
          for (int j = 0; j < P; ++j) Gs[(j * BM + row) * LD + c] = ok ? up[ri] * g[j] : 0.f;
        }
        asm volatile("" ::: "memory");   // one element at a time: keeps the register peak low
      }
      if (Traits::HAS_R) {
        // ---- queue pass: one queued element per lane ----
        lds_wave_fence();
        const int n_q = min(q_base, D2_QCAP);
        if (lane < n_q) {
          const float r = q0[lane], tval = q1[lane];
          float A, D;
          lgamma_digamma_diff<TRAIN>(r, tval, A, D);
          if (inline_lgamma) A -= lgamma1p(tval);
          q0[lane] = A;
          if (TRAIN) Gs[(P - 1) * BM * LD + q3[lane]] += q2[lane] * r * D;
        }
        lds_wave_fence();
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (slot[e] >= 0) lsum[e >> 1] += q0[slot[e]];
        lds_wave_fence();   // the queue is reused by this wave's next tile
      }
      // ---- per-row partial log-likelihood of this strip ----
#pragma unroll
      for (int ri = 0; ri < 2; ++ri) {
        float s = lsum[ri];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) s += __shfl_xor(s, off, WAVE);
        const int grow = m0 + er0 + 16 * ri;
        if (ec == 0 && (FULL || grow < R)) ll_part[(size_t)blockIdx.x * R + grow] = s;
      }
        };
    if (live) {
      if (m0 + BM <= R && c0 + BN <= F) epilogue(std::true_type{});
      else epilogue(std::false_type{});
    }
    lds_barrier();
    lds_barrier();
  }
  if (!half) {
    lds_barrier();
    lds_barrier();
  }

  if (!TRAIN) return;
  // ---- combine the two halves' dW accumulators (B parks in LDS, A adds and writes) ----
  __syncthreads();
  float* park = smem + (size_t)hw * P * 16 * 64;
  if (half == 1) {
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        park[(j * 16 + r) * 64 + lane] = REM ? accQ[j][r >> 2][r & 3] : accW[j][r];
  }
  __syncthreads();
  if (REM && half == 0) {
    const int i16 = lane & 15, k4 = lane >> 4;
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (g2_rem && q >= 2) continue;          // (the remainder job has one h sub-tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = accQ[j][q][r] + park[(j * 16 + 4 * q + r) * 64 + lane];
          const int h = g2_h0 + 16 * (q >> 1) + 4 * k4 + r, c = c0 + g2_n0 + 16 * (q & 1) + i16;
          if (c < F) {
            if (h < H) hp.dW[j][(size_t)h * F + c] = v;
            else if (h == H) hp.db[j][c] = v;
          }
        }
      }
  } else if (half == 0 && g2_h0 <= H) {
    const int c = c0 + g2_n0 + li;
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = accW[j][r] + park[(j * 16 + r) * 64 + lane];
        const int h = g2_h0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (c < F) {
          if (h < H) hp.dW[j][(size_t)h * F + c] = v;
          else if (h == H) hp.db[j][c] = v;
        }
      }
  }
}

template <bool TRAIN>
static int launch_decoder2(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                           int F, Targets t, int B, const float* gw, int inline_lgamma,
                           float* ll_part, float* dd_part) {
  const int P = likelihood_heads(kind);
  const size_t lds = decoder_fused2_lds_bytes(P, H);
  const int strips = (F + D2_BN - 1) / D2_BN;
  const unsigned magic_h = (unsigned)(0x100000000ull / (unsigned)H) + 1u;
#define SCVAE_D2(K_)                                                                              \
  do {                                                                                            \
    void (*kfn)(const float*, int, int, unsigned, HeadParams, int, Targets, int, const float*,    \
                int, float*, float*) = decoder_head2_kernel<K_, TRAIN, false>;                    \
    if constexpr (TRAIN && LikelihoodTraits<K_>::P <= 2) {                                        \
      if (rem) kfn = decoder_head2_kernel<K_, TRAIN, true>;                                       \
    }                                                                                             \
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), \
                                  (int)lds));         \
    hipLaunchKernelGGL(kfn, dim3(strips), dim3(D2_THREADS), lds, s, d, rows, H, magic_h, hp, F,   \
                       t, B, gw, inline_lgamma, ll_part, dd_part);                                \
  } while (0)
  // the schedule with the 16-wide h remainder (see the kernel): where it applies and pays
  const bool rem = TRAIN && P <= 2 && H > 96 && H <= 111 && rows >= 512;
  switch (kind) {
    case LK_POISSON: SCVAE_D2(LK_POISSON); break;
    case LK_NB: SCVAE_D2(LK_NB); break;
    case LK_ZIP: SCVAE_D2(LK_ZIP); break;
    case LK_ZINB: SCVAE_D2(LK_ZINB); break;
    case LK_BERNOULLI: SCVAE_D2(LK_BERNOULLI); break;   // du:194-204; targets binarised by the caller
    default: set_error("unknown likelihood kind %d", kind); return -1;
  }
#undef SCVAE_D2
  SCVAE_LAUNCH_CHECK("decoder_head2_kernel");
  return 0;
}

int decoder_fused2_launch(hipStream_t s, bool train, int kind, const float* d, int rows, int H,
                          HeadParams hp, int F, Targets t, int B, const float* gw,
                          int inline_lgamma, float* ll_part, float* dd_part) {
  return train ? launch_decoder2<true>(s, kind, d, rows, H, hp, F, t, B, gw, inline_lgamma, ll_part,
                                       dd_part)
               : launch_decoder2<false>(s, kind, d, rows, H, hp, F, t, B, gw, inline_lgamma,
                                        ll_part, dd_part);
}

}  // namespace scvae
