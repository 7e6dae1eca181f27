// Generic fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32):
//   C[M,N] (+)= act(op(A)[M,K] * op(B)[K,N] + bias[N])
// This is the dense layer of scvae/models/utilities.py:38-76
// (tf.contrib.layers.fully_connected: x*W + b, optional ReLU) and its two
// backward contractions (dX = dY*W^T, dW = X^T*dY).  fp32 in, fp32 accumulate:
// the MFMA result is bit-identical to a k-ordered fmaf chain.
//
// Tile: 64x64 per 256-thread workgroup (4 waves, one 32x32 accumulator each),
// BK = 16, operands staged k-major in LDS so that the MFMA A/B fragment reads
// (lane l: A[i=l&31][k=l>>5]) are bank-conflict free.  Optional split-K writes
// fp32 partial slabs that a second kernel reduces in a fixed order
// (deterministic; no atomics).
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace scvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GT = 64;   // tile edge
constexpr int GBK = 16;  // k-step staged per barrier
constexpr int GLD = GT + 4;

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(
    const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int act, int accumulate,
    int k_chunk, float* __restrict__ slabs) {
  __shared__ float As[2][GBK][GLD];
  __shared__ float Bs[2][GBK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int kz = blockIdx.z;
  const int k_begin = kz * k_chunk;
  const int k_end = min(K, k_begin + k_chunk);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  float ra[4], rb[4];
  auto load_tiles = [&](int kt) {
    // A tile: 64 (m) x 16 (k)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int m, k;
      if (TA) { m = tid & 63; k = (tid >> 6) + 4 * p; }
      else    { k = tid & 15; m = (tid >> 4) + 16 * p; }
      const int gm = m0 + m, gk = kt + k;
      float v = 0.f;
      if (gm < M && gk < k_end) v = TA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk];
      ra[p] = v;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int n, k;
      if (TB) { k = tid & 15; n = (tid >> 4) + 16 * p; }
      else    { n = tid & 63; k = (tid >> 6) + 4 * p; }
      const int gn = n0 + n, gk = kt + k;
      float v = 0.f;
      if (gn < N && gk < k_end) v = TB ? B[(size_t)gn * ldb + gk] : B[(size_t)gk * ldb + gn];
      rb[p] = v;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int m, k;
      if (TA) { m = tid & 63; k = (tid >> 6) + 4 * p; }
      else    { k = tid & 15; m = (tid >> 4) + 16 * p; }
      As[buf][k][m] = ra[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int n, k;
      if (TB) { k = tid & 15; n = (tid >> 4) + 16 * p; }
      else    { n = tid & 63; k = (tid >> 6) + 4 * p; }
      Bs[buf][k][n] = rb[p];
    }
  };

  int buf = 0;
  if (k_begin < k_end) {
    load_tiles(k_begin);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = k_begin; kt < k_end; kt += GBK) {
    const bool has_next = kt + GBK < k_end;
    if (has_next) load_tiles(kt + GBK);  // global loads in flight under the MFMAs
    const int kh = lane >> 5, li = lane & 31;
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const float a = As[buf][kk + kh][wm * 32 + li];
      const float b = Bs[buf][kk + kh][wn * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (has_next) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int col = n0 + wn * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = (bias != nullptr && slabs == nullptr) ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row >= M) continue;
    if (slabs != nullptr) {
      slabs[((size_t)kz * M + row) * N + col] = acc[r];
    } else {
      float v = acc[r] + bv;
      if (act == ACT_RELU) v = fmaxf(v, 0.f);
      float* c = C + (size_t)row * ldc + col;
      *c = accumulate ? (*c + v) : v;
    }
  }
}

// Large-shape variant (encoder input layer: X[B,F] W[F,H] and its dW = X^T dY):
// 128x128 tile per 512-thread workgroup (8 waves, two 32x32 accumulators each sharing the A
// fragment), BK = 32, B operand row-major [K,N].  TA: A is stored [K,M] (i.e. C = A^T B).
constexpr int BT = 128;
constexpr int BBK = 32;
constexpr int BLD = BT + 1;   // odd: the transposing store of the A[m][k] tile is conflict free

template <bool TA>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(
    const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int act, int accumulate,
    int k_chunk, float* __restrict__ slabs) {
  __shared__ float As[2][BBK][BLD];
  __shared__ float Bs[2][BBK][BLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mt = w >> 1, ntp = (w & 1) * 2;     // row tile, first of two column tiles
  const int kh = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
  const int kz = blockIdx.z;
  const int k_begin = kz * k_chunk;
  const int k_end = min(K, k_begin + k_chunk);

  f32x16 acc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;

  // per k-step every thread moves 8 A and 8 B elements
  float ra[8], rb[8];
  auto load_tiles = [&](int kt) {
    if (TA) {
      // A[k][m]: 32 rows of 128 consecutive m
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int k = (tid >> 7) + 4 * p, m = tid & 127;
        const int gk = kt + k, gm = m0 + m;
        ra[p] = (gk < k_end && gm < M) ? A[(size_t)gk * lda + gm] : 0.f;
      }
    } else {
      // A[m][k]: 128 rows of 32 consecutive k
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int m = (tid >> 5) + 16 * p, k = tid & 31;
        const int gk = kt + k, gm = m0 + m;
        ra[p] = (gk < k_end && gm < M) ? A[(size_t)gm * lda + gk] : 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int k = (tid >> 7) + 4 * p, n = tid & 127;
      const int gk = kt + k, gn = n0 + n;
      rb[p] = (gk < k_end && gn < N) ? B[(size_t)gk * ldb + gn] : 0.f;
    }
  };
  auto store_tiles = [&](int buf) {
    if (TA) {
#pragma unroll
      for (int p = 0; p < 8; ++p) As[buf][(tid >> 7) + 4 * p][tid & 127] = ra[p];
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) As[buf][tid & 31][(tid >> 5) + 16 * p] = ra[p];
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) Bs[buf][(tid >> 7) + 4 * p][tid & 127] = rb[p];
  };

  int buf = 0;
  if (k_begin < k_end) {
    load_tiles(k_begin);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = k_begin; kt < k_end; kt += BBK) {
    const bool has_next = kt + BBK < k_end;
    if (has_next) load_tiles(kt + BBK);
#pragma unroll
    for (int kk = 0; kk < BBK; kk += 2) {
      const float a = As[buf][kk + kh][mt * 32 + li];
      const float b0 = Bs[buf][kk + kh][ntp * 32 + li];
      const float b1 = Bs[buf][kk + kh][ntp * 32 + 32 + li];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
    }
    if (has_next) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int col = n0 + (ntp + q) * 32 + li;
    if (col >= N) continue;
    const float bv = (bias != nullptr && slabs == nullptr) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row >= M) continue;
      if (slabs != nullptr) {
        slabs[((size_t)kz * M + row) * N + col] = acc[q][r];
      } else {
        float v = acc[q][r] + bv;
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        float* c = C + (size_t)row * ldc + col;
        *c = accumulate ? (*c + v) : v;
      }
    }
  }
}

// Narrow-N variant for the same two products when N is not a multiple of 32 (the reference's
// default hidden size is 100): 256 rows x N columns per 512-thread workgroup, every wave one
// 32-row tile with all NT = N / 32 full 32-wide column tiles (A fragment shared), and the
// N % 32 <= 8 remainder columns through v_mfma_f32_4x4x1 (64 rows x 4 columns per instruction)
// instead of a padded fourth tile: 100 columns cost 3.06 tiles of MFMA work instead of 4.
constexpr int NBM = 256;          // rows per workgroup
constexpr int NBK = 16;           // k-step per barrier
constexpr int NLDA = NBM + 4;     // k-stride of 4 banks: the A[m][k] store (4 k-groups x 16 rows
                                  // per wave) then puts exactly two lanes on every bank
constexpr int NBN = 104;          // at most 96 + 8 columns
constexpr int NLDB = NBN + 1;

template <bool TA>
__global__ __launch_bounds__(512, 2) void gemm_narrow_kernel(
    const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int act, int accumulate,
    int k_chunk, float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) float As[2][NBK][NLDA];
  __shared__ float Bs[2][NBK][NLDB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kh = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.y * NBM;
  const int kz = blockIdx.z;
  const int k_begin = kz * k_chunk;
  const int k_end = min(K, k_begin + k_chunk);
  const int NT = N / 32;                 // full column tiles (1..3)
  const int rem0 = NT * 32;              // first remainder column
  const int groups = (N - rem0 + 3) / 4; // 4-column remainder groups (0..2)
  // remainder job of this wave: 64-row block (w & 3), column group (w >> 2)
  const bool rem_wave = (w >> 2) < groups;

  f32x16 acc[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
  // the 4x4x1 MFMA has a ~40-cycle dependent latency at an 8-12-cycle issue rate
  // (tools/probe/mfma_latency.hip): two independent accumulator chains (k parity), summed at
  // the end; four would push the kernel over 128 VGPRs (one workgroup per CU instead of two)
  f32x4 accR[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) accR[q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: A 256 x 16 (8 per thread), B 16 x 104 (<= 4 per thread).  A[m][k] (row-major
  // input, any pitch): two 16-byte loads of 4 consecutive k per thread (global loads need only
  // 4-byte alignment); A[k][m]: 8 x 4 bytes, 256 consecutive m per instruction (16-byte loads
  // were slower there: 285 -> 305 us).
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  float ra[8], rb[4];
  auto load_tiles = [&](int kt) {
    if (TA) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int m = tid & 255, k = (tid >> 8) + 2 * p;      // A[k][m]: 256 consecutive m
        const int gm = m0 + m, gk = kt + k;
        ra[p] = (gm < M && gk < k_end) ? A[(size_t)gk * lda + gm] : 0.f;
      }
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int k = (tid & 3) * 4, m = (tid >> 2) + 128 * p;
        const int gm = m0 + m, gk = kt + k;
        const float* src = A + (size_t)gm * lda + gk;
        if (gm < M && gk + 3 < k_end) {
          const f32x4u v = *reinterpret_cast<const f32x4u*>(src);
          ra[4 * p] = v.x; ra[4 * p + 1] = v.y; ra[4 * p + 2] = v.z; ra[4 * p + 3] = v.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) ra[4 * p + u] = (gm < M && gk + u < k_end) ? src[u] : 0.f;
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = p * 512 + tid;           // 16 x 104 = 1664 elements
      const int k = e / NBN, n = e - k * NBN;
      const int gk = kt + k;
      float v = 0.f;
      if (k < NBK && n < N && gk < k_end) v = B[(size_t)gk * ldb + n];
      rb[p] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    if (TA) {
#pragma unroll
      for (int p = 0; p < 8; ++p) As[buf][(tid >> 8) + 2 * p][tid & 255] = ra[p];
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p)
        As[buf][(tid & 3) * 4 + (p & 3)][(tid >> 2) + 128 * (p >> 2)] = ra[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = p * 512 + tid;
      const int k = e / NBN, n = e - k * NBN;
      if (k < NBK) Bs[buf][k][n] = rb[p];
    }
  };

  int buf = 0;
  if (k_begin < k_end) {
    load_tiles(k_begin);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = k_begin; kt < k_end; kt += NBK) {
    const bool has_next = kt + NBK < k_end;
    if (has_next) load_tiles(kt + NBK);
#pragma unroll
    for (int kk = 0; kk < NBK; kk += 2) {
      const float a = As[buf][kk + kh][w * 32 + li];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (q < NT)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[buf][kk + kh][q * 32 + li], acc[q], 0,
                                                        0, 0);
    }
    if (rem_wave) {
      // 16 blocks x (4 rows x 4 columns): lane l gives A = row 64*(w&3) + l, B = column
      // rem0 + 4*(w>>2) + (l & 3)
      const int rr = (w & 3) * 64 + lane, cc = rem0 + (w >> 2) * 4 + (lane & 3);
#pragma unroll
      for (int kk = 0; kk < NBK; ++kk)
        accR[kk & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(As[buf][kk][rr], Bs[buf][kk][cc],
                                                          accR[kk & 1], 0, 0, 0);
    }
    if (has_next) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  auto emit = [&](int row, int col, float v) {
    if (row >= M || col >= N) return;
    if (slabs != nullptr) {
      slabs[((size_t)kz * M + row) * N + col] = v;
    } else {
      if (bias != nullptr) v += bias[col];
      if (act == ACT_RELU) v = fmaxf(v, 0.f);
      float* c = C + (size_t)row * ldc + col;
      *c = accumulate ? (*c + v) : v;
    }
  };
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (q >= NT) break;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      emit(m0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, q * 32 + li, acc[q][r]);
  }
  if (rem_wave) {
    // lane l, register i: row 64*(w&3) + 4*(l>>2) + i, column rem0 + 4*(w>>2) + (l & 3)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      emit(m0 + (w & 3) * 64 + 4 * (lane >> 2) + i, rem0 + (w >> 2) * 4 + (lane & 3),
           accR[0][i] + accR[1][i]);
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(
    const float* __restrict__ slabs, const float* __restrict__ bias, float* __restrict__ C, int M,
    int N, int ldc, int splits, int act, int accumulate) {
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / N), col = (int)(i % N);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(size_t)z * total + i];
    if (bias) s += bias[col];
    if (act == ACT_RELU) s = fmaxf(s, 0.f);
    float* c = C + (size_t)row * ldc + col;
    *c = accumulate ? (*c + s) : s;
  }
}

// Many splits of a small output: 16 elements x 16 split lanes per workgroup; a lane sums splits
// l, l+16, ..., the lanes are then combined in sequence (fixed order: deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(
    const float* __restrict__ slabs, const float* __restrict__ bias, float* __restrict__ C, int M,
    int N, int ldc, int splits, int act, int accumulate) {
  __shared__ float red[16][17];
  const size_t total = (size_t)M * N;
  const int il = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + il;
  float s = 0.f;
  if (i < total) {
    int z = zl;
    for (; z + 7 * 16 < splits; z += 8 * 16) {     // (eight slabs' loads in flight, same order)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(z + 16 * u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; z += 16) s += slabs[(size_t)z * total + i];
  }
  red[zl][il] = s;
  __syncthreads();
  if (zl == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][il];
    const int row = (int)(i / N), col = (int)(i % N);
    if (bias) t += bias[col];
    if (act == ACT_RELU) t = fmaxf(t, 0.f);
    float* c = C + (size_t)row * ldc + col;
    *c = accumulate ? (*c + t) : t;
  }
}

static bool gemm_use_big(bool tb, int M, int N, int K) {
  return !tb && (double)M * N * K >= 4.0e9 && N >= 64;
}

static int gemm_big_splits(int M, int N, int K) {
  const long tiles = (long)((M + BT - 1) / BT) * ((N + BT - 1) / BT);
  if (tiles >= 192) return 1;
  long want = (512 + tiles - 1) / tiles;
  long max_by_k = K / 512;
  long s = want < max_by_k ? want : max_by_k;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

size_t gemm_workspace_bytes(int M, int N, int K) {
  int splits = gemm_choose_splits(M, N, K);
  const int big = gemm_big_splits(M, N, K);
  if (big > splits) splits = big;
  {   // narrow-N kernel (see gemm())
    const long tiles = (M + NBM - 1) / NBM;
    if (tiles < 384) {
      long want = (512 + tiles - 1) / tiles, max_by_k = K / 256;
      long sn = want < max_by_k ? want : max_by_k;
      if (sn > 64) sn = 64;
      if (sn > splits) splits = (int)sn;
    }
  }
  return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

int gemm_choose_splits(int M, int N, int K) {
  const long tiles = (long)((M + GT - 1) / GT) * ((N + GT - 1) / GT);
  // (the k loop is a chain of dependent global round trips, ~0.9 us per 16-k step on one
  //  workgroup: K = 512 on a 2 x 2 grid took 28 us, the weight gradients of the GMVAE's q(y|x)
  //  layers; split from K = 256 on -- SCVAE_GEMM_SPLIT_MIN_K=1024 restores the old threshold)
  static const int min_k = [] { const char* e = getenv("SCVAE_GEMM_SPLIT_MIN_K"); return e ? atoi(e) : 256; }();
  if (tiles >= 256 || K < min_k) return 1;
  long want = (512 + tiles - 1) / tiles;          // aim for ~2 workgroups per CU
  long max_by_k = K / 64;                         // keep >= 64 of K (4 k-steps) per split
  long s = want < max_by_k ? want : max_by_k;
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  return (int)s;
}

int gemm(hipStream_t stream, bool ta, bool tb, const float* A, const float* B, const float* bias,
         float* C, int M, int N, int K, int lda, int ldb, int ldc, int act, bool accumulate,
         float* workspace, size_t workspace_bytes) {
  SCVAE_ARG(A && B && C);
  SCVAE_ARG(M >= 0 && N >= 0 && K >= 0);
  if (M == 0 || N == 0) return 0;
  const bool big = gemm_use_big(tb, M, N, K);
  // N = 32 q + r with r <= 8 (e.g. the default hidden size 100): no padding of N
  const bool narrow = big && N >= 32 && N <= NBN && (N % 32) != 0 && (N % 32) <= 8 && ldb >= N;
  if (narrow) {
    int splits = 1;
    const long tiles = (M + NBM - 1) / NBM;
    if (tiles < 384) {
      // two workgroups per CU in total (a 1.5-wave grid costs more than it gains)
      long want = (512 + tiles - 1) / tiles, max_by_k = K / 256;
      splits = (int)(want < max_by_k ? want : max_by_k);
      if (splits < 1) splits = 1;
      if (splits > 64) splits = 64;
    }
    if (splits > 1 && (workspace == nullptr ||
                       workspace_bytes < (size_t)splits * M * N * sizeof(float)))
      splits = 1;
    int k_chunk = K;
    if (splits > 1) {
      k_chunk = (K + splits - 1) / splits;
      k_chunk = (k_chunk + NBK - 1) / NBK * NBK;
      splits = (K + k_chunk - 1) / k_chunk;
    }
    dim3 grid(1, (M + NBM - 1) / NBM, splits);
    float* slabs = splits > 1 ? workspace : nullptr;
    const int acc = accumulate ? 1 : 0;
    if (ta)
      hipLaunchKernelGGL((gemm_narrow_kernel<true>), grid, dim3(512), 0, stream, A, B, bias, C, M,
                         N, K, lda, ldb, ldc, act, acc, k_chunk, slabs);
    else
      hipLaunchKernelGGL((gemm_narrow_kernel<false>), grid, dim3(512), 0, stream, A, B, bias, C, M,
                         N, K, lda, ldb, ldc, act, acc, k_chunk, slabs);
    SCVAE_LAUNCH_CHECK("gemm_narrow_kernel");
    if (splits > 1) {
      const size_t total = (size_t)M * N;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slabs, bias, C,
                         M, N, ldc, splits, act, acc);
      SCVAE_LAUNCH_CHECK("splitk_reduce_kernel");
    }
    return 0;
  }
  int splits = big ? gemm_big_splits(M, N, K) : gemm_choose_splits(M, N, K);
  if (splits > 1 && (workspace == nullptr ||
                     workspace_bytes < (size_t)splits * M * N * sizeof(float)))
    splits = 1;
  const int kstep = big ? BBK : GBK;
  const int tile = big ? BT : GT;
  int k_chunk = K;
  if (splits > 1) {
    k_chunk = (K + splits - 1) / splits;
    k_chunk = (k_chunk + kstep - 1) / kstep * kstep;
    splits = (K + k_chunk - 1) / k_chunk;
  }
  dim3 grid((N + tile - 1) / tile, (M + tile - 1) / tile, splits);
  float* slabs = splits > 1 ? workspace : nullptr;
  const int acc = accumulate ? 1 : 0;
  if (big) {
    if (ta)
      hipLaunchKernelGGL((gemm_big_kernel<true>), grid, dim3(512), 0, stream, A, B, bias, C, M, N, K,
                         lda, ldb, ldc, act, acc, k_chunk, slabs);
    else
      hipLaunchKernelGGL((gemm_big_kernel<false>), grid, dim3(512), 0, stream, A, B, bias, C, M, N,
                         K, lda, ldb, ldc, act, acc, k_chunk, slabs);
    SCVAE_LAUNCH_CHECK("gemm_big_kernel");
  } else {
#define SCVAE_GEMM_LAUNCH(TA_, TB_)                                                              \
  hipLaunchKernelGGL((gemm_kernel<TA_, TB_>), grid, dim3(256), 0, stream, A, B, bias, C, M, N, K, \
                     lda, ldb, ldc, act, acc, k_chunk, slabs)
  if (ta) { if (tb) SCVAE_GEMM_LAUNCH(true, true); else SCVAE_GEMM_LAUNCH(true, false); }
  else    { if (tb) SCVAE_GEMM_LAUNCH(false, true); else SCVAE_GEMM_LAUNCH(false, false); }
#undef SCVAE_GEMM_LAUNCH
    SCVAE_LAUNCH_CHECK("gemm_kernel");
  }
  if (splits > 1) {
    const size_t total = (size_t)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (splits >= 16 && total <= (size_t)1 << 16) {
      hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256),
                         0, stream, slabs, bias, C, M, N, ldc, splits, act, acc);
    } else {
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slabs, bias, C,
                         M, N, ldc, splits, act, acc);
    }
    SCVAE_LAUNCH_CHECK("splitk_reduce_kernel");
  }
  return 0;
}

}  // namespace scvae
