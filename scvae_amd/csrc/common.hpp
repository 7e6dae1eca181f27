// Shared device/host helpers for the scVAE gfx950 kernels.
// Wave = 64 lanes everywhere in this file (CDNA4); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace scvae {

constexpr int WAVE = 64;
constexpr float F32_TINY = 1.1754943508222875e-38f;
constexpr float LOGIT_OF_TINY = -87.33654475055310898657f;  // log(float32.tiny)
constexpr float LOG_F32_TINY = LOGIT_OF_TINY;
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;
constexpr float F32_MAX_HALF = 1.7014117331926443e38f;
constexpr float BN_EPSILON = 1e-3f;
constexpr float BN_DECAY = 0.999f;
// assign_moving_average casts the Python double (1.0 - decay) to fp32
constexpr float BN_UPDATE_RATE = 1e-3f;

// ---- error reporting (per-thread string, returned by scvae_last_error) ----
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
#define SCVAE_HIP(call)                                            \
  do {                                                             \
    int _rc = ::scvae::check_hip((call), #call);                   \
    if (_rc) return _rc;                                           \
  } while (0)
// hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), issued only when a
// kernel needs more than any earlier launch of it in this process on this device (a driver
// call per launch is microseconds on paths tuned to the microsecond)
hipError_t max_dynamic_lds(const void* fn, int bytes);
#define SCVAE_LAUNCH_CHECK(name)                                   \
  do {                                                             \
    int _rc = ::scvae::check_hip(hipGetLastError(), name);         \
    if (_rc) return _rc;                                           \
  } while (0)
#define SCVAE_ARG(cond)                                            \
  do {                                                             \
    if (!(cond)) {                                                 \
      ::scvae::set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
      return -1;                                                   \
    }                                                              \
  } while (0)

// likelihood kinds (order of heads follows the DISTRIBUTIONS registry,
// scvae/distributions/utilities.py:206-305)
enum Likelihood : int {
  LK_POISSON = 0,  // heads: log_lambda
  LK_NB = 1,       // heads: p, log_r
  LK_ZIP = 2,      // heads: pi, log_lambda
  LK_ZINB = 3,     // heads: pi, p, log_r
  LK_CPOISSON = 4, // constrained Poisson (du:218-228): head lambda = softmax over the genes,
                   // rate = lambda * N with N the count sum of the cell; unfused path only
  LK_BERNOULLI = 5, // heads: logits (du:194-204; binarised targets); unfused path only
  // (internal to the fused kernels, never a model's likelihood) the categorical part of the
  // piecewise categorical likelihood `-k` (distributions/categorised.py:255-263, va:2507-2532):
  // k + 1 classes of one gene as k + 1 "heads", log softmax(logits)[min(t, k)]
  LK_CAT2 = 6,      // k = 1: classes 0, >= 1
  LK_CAT3 = 7       // k = 2: classes 0, 1, >= 2
};
__host__ __device__ constexpr int likelihood_heads(int kind) {
  return (kind == LK_POISSON || kind == LK_CPOISSON || kind == LK_BERNOULLI)
             ? 1
             : ((kind == LK_ZINB || kind == LK_CAT3) ? 3 : 2);
}

#ifdef __HIPCC__
// ---- wave / block reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, WAVE));
  return v;
}
// Sum over a block of NT threads (NT multiple of 64, <= 1024). `red` is an LDS
// array of NT/64 floats. Result valid in every thread.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) s += red[i];
  return s;
}

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations
// (lgkmcnt) but NOT for its outstanding global loads/stores (vmcnt), so register prefetches
// issued before the barrier stay in flight across it.  (__syncthreads() drains vmcnt too.)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- scalar math used by the likelihood kernels ----
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// natural log of a normal positive float: v_log_f32 (1 ulp) times ln 2.  (__logf expands to a
// dozen instructions: denormal range handling and a hi/lo multiply by ln 2, neither needed here)
__device__ __forceinline__ float fast_log(float x) {
  return __builtin_amdgcn_logf(x) * 0.6931471805599453f;
}
// log(1 + x), x > -1: log(u) * x / (u - 1) with u = fl(1 + x) removes the rounding of 1 + x
__device__ __forceinline__ float fast_log1p(float x) {
  const float u = 1.f + x;
  const float d = u - 1.f;
  return d == 0.f ? x : fast_log(u) * (x * fast_rcp(d));
}
// log(sigmoid(a)), log(sigmoid(-a)), sigmoid(a) and sigmoid(-a) from one exp and one log.
// sigmoid(-a) is formed on its own, not as 1 - sigmoid(a): for a >~ 8 that difference loses
// most of its bits (p = 0.99995 at a = 10: 1.f - p carries a relative error of 1e-3), and the
// gradient of a saturated head is t (1 - p) - r p (tests/test_gpu_head_edges.py)
__device__ __forceinline__ void log_sigmoid_pair(float a, float& ls_pos, float& ls_neg,
                                                 float& sig, float& sig_neg) {
  const float e = __expf(-fabsf(a));
  const float l = fast_log1p(e);
  ls_pos = fminf(a, 0.f) - l;
  ls_neg = fminf(-a, 0.f) - l;
  const float s = fast_rcp(1.f + e);
  const float es = e * s;
  sig = a >= 0.f ? s : es;
  sig_neg = a >= 0.f ? es : s;
}
__device__ __forceinline__ float log_sigmoid(float a) {
  return fminf(a, 0.f) - fast_log1p(__expf(-fabsf(a)));
}
__device__ __forceinline__ float softplusf(float a) {
  return fmaxf(a, 0.f) + fast_log1p(__expf(-fabsf(a)));
}
__device__ __forceinline__ float sigmoidf(float a) {
  const float e = __expf(-fabsf(a));
  const float s = fast_rcp(1.f + e);
  return a >= 0.f ? s : e * s;
}

// Stirling tail s(y) = 1/(12y) - 1/(360y^3) + 1/(1260y^5) - 1/(1680y^7), y >= 4 (iy = 1/y):
// the first term left out is 1/(1188 y^9) <= 3.2e-9
__device__ __forceinline__ float stirling_tail_r(float iy) {
  const float iy2 = iy * iy;
  return iy * (8.3333333333e-2f +
               iy2 * (-2.7777777778e-3f + iy2 * (7.9365079365e-4f + iy2 * -5.9523809524e-4f)));
}
// digamma tail u(y) = 1/(2y) + 1/(12y^2) - 1/(120y^4) + 1/(252y^6) - 1/(240y^8), psi(y) = log y
// - u(y), y >= 4 (left out: 1/(132 y^10) <= 7.3e-9)
__device__ __forceinline__ float digamma_tail_r(float iy) {
  const float iy2 = iy * iy;
  return 0.5f * iy +
         iy2 * (8.3333333333e-2f +
                iy2 * (-8.3333333333e-3f + iy2 * (3.9682539683e-3f + iy2 * -4.1666666667e-3f)));
}

// A = lgamma(r+t) - lgamma(r) and D = digamma(r+t) - digamma(r) for r > 0,
// t >= 0 (t need not be an integer).  Both arguments are shifted up by 4 with
// the recurrence, and the shifted difference is taken analytically
//   lgamma(b+t)-lgamma(b) = t*log(b+t) + (b-1/2)*log1p(t/b) - t + s(b+t)-s(b)
// so there is no cancellation of two large lgamma values.  Exactly 0 at t == 0.
// Small integer counts (the bulk of a count matrix): lgamma(r+t)-lgamma(r) = log prod_{i<t}(r+i)
// and digamma(r+t)-digamma(r) = sum_{i<t} 1/(r+i) = P'/P by the product recurrence -- one log and
// one reciprocal.  r <= e^10, t <= 8: the product stays below 6e34.
template <bool WITH_D>
__device__ __forceinline__ void lgamma_digamma_diff_small(float r, float t, float& A, float& D) {
  float P = 1.f, Q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float f = r + (float)i;
    const bool on = (float)i < t;
    if (WITH_D) Q = on ? fmaf(Q, f, P) : Q;
    P = on ? P * f : P;
  }
  A = fast_log(P);
  D = WITH_D ? Q * fast_rcp(P) : 0.f;
}

// The same for a whole wave inside a non-zero walk: the product runs only as far as the largest t
// among the lanes (the step's own comparison is the loop condition: no extra vector instruction),
// which for count data is 2-5 steps instead of 8.  Identical bits to lgamma_digamma_diff_small
// (a lane's skipped steps are the ones that leave its P and Q unchanged).
template <bool WITH_D>
__device__ __forceinline__ void lgamma_digamma_diff_small_wave(float r, float t, float& A,
                                                               float& D) {
  float P = 1.f, Q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool on = (float)i < t;
    if (__builtin_amdgcn_ballot_w64(on) == 0) break;
    const float f = r + (float)i;
    if (WITH_D) Q = on ? fmaf(Q, f, P) : Q;
    P = on ? P * f : P;
  }
  A = fast_log(P);
  D = WITH_D ? Q * fast_rcp(P) : 0.f;
}

// (round 4: a shift by 4 with one more term in each tail instead of a shift by 8 -- half the
//  product work and 8 transcendentals instead of 11; against scipy in fp32 emulation over
//  r in [e^-10, e^10], t in [1, 65535]: |A - A*| <= 2.1e-7 (t log(r + t) + |log r| + 1),
//  |D - D*| <= 1.2e-6 D*, both a little below the shift by 8.)  The four shift factors as
//  y (y + 3) = u and (y + 1)(y + 2) = u + 2: product u (u + 2), derivative (2 y + 3)(2 u + 2).
template <bool WITH_D>
__device__ __forceinline__ void lgamma_digamma_diff_general(float r, float t, float& A, float& D) {
  const float x = r + t;
  const float b = r + 4.f, a = x + 4.f;
  const float ib = fast_rcp(b), ia = fast_rcp(a);
  const float l1 = fast_log1p(t * ib);
  const float A4 = t * fast_log(a) + (b - 0.5f) * l1 - t + (stirling_tail_r(ia) - stirling_tail_r(ib));
  const float ux = x * (x + 3.f), ur = r * (r + 3.f);
  const float n = ux * (ux + 2.f), d = ur * (ur + 2.f);      // no overflow for t < 1e7
  const float in = fast_rcp(n), id = fast_rcp(d);
  A = A4 - fast_log(n * id);              // the ratio is >= 1 and finite
  if (WITH_D) {
    const float D4 = l1 - (digamma_tail_r(ia) - digamma_tail_r(ib));
    const float np = fmaf(2.f, x, 3.f) * fmaf(2.f, ux, 2.f);
    const float dp = fmaf(2.f, r, 3.f) * fmaf(2.f, ur, 2.f);
    // sum_{i<4} 1/(r+i) - 1/(x+i)
    D = D4 + (dp * id - np * in);
  } else {
    D = 0.f;
  }
}

template <bool WITH_D>
__device__ __forceinline__ void lgamma_digamma_diff(float r, float t, float& A, float& D) {
  if (t <= 8.f && t == __builtin_rintf(t)) lgamma_digamma_diff_small<WITH_D>(r, t, A, D);
  else lgamma_digamma_diff_general<WITH_D>(r, t, A, D);
}

// lgamma(1+t) for t >= 0 (data-only term of the count likelihoods)
__device__ __forceinline__ float lgamma1p(float t) {
  return t == 0.f ? 0.f : lgammaf(1.f + t);
}
// ---- batch-norm arithmetic of the chunked kernels (elementwise.hip).
// Written out operation by operation (no contraction left to the compiler) so that two kernels
// using the same helper give the same bits whatever surrounds the call.
#pragma clang fp contract(off)
__device__ __forceinline__ float bn_normalise(float a, float mu, float istd, float beta) {
  return fmaf(a - mu, istd, beta);
}
// merge of chunk statistics: sum of n * mean, then sum of M2 + n * (mean - mu)^2
__device__ __forceinline__ float bn_merge_mean(float acc, float n, float chunk_mean) {
  return fmaf(n, chunk_mean, acc);
}
__device__ __forceinline__ float bn_merge_m2(float acc, float n, float chunk_mean, float chunk_m2,
                                             float mu) {
  const float d = chunk_mean - mu;
  return acc + fmaf(n * d, d, chunk_m2);
}
// dA = istd (g - s1 / count - xhat s2 / count)
__device__ __forceinline__ float bn_input_gradient(float g, float xh, float s1, float s2,
                                                   float inv_count, float istd) {
  const float t = g - s1 * inv_count;
  return istd * (t - (xh * s2) * inv_count);
}
// moving <- moving - (moving - batch) * rate
// The analytic Gaussian KL term and its gradients (va:2624-2656), shared by gauss_latent_*
// (elementwise.hip) and the tile chain's latent stage (tilechain.hip); the fused multiply-adds
// are spelled out so that every caller rounds alike (the compiler's own contraction differs from
// one kernel to the next).
__device__ __forceinline__ float gauss_kl_elem(float mu, float sigma, float ls) {
  return fmaf(0.5f, fmaf(mu, mu, fmaf(sigma, sigma, -1.f)), -ls);
}
__device__ __forceinline__ float gauss_kl_dmu(float gz, float kl_coeff, float mu) {
  return fmaf(kl_coeff, mu, gz);
}
__device__ __forceinline__ float gauss_kl_dls(float gze, float sigma, float kl_coeff) {
  return fmaf(gze, sigma, kl_coeff * fmaf(sigma, sigma, -1.f));
}
__device__ __forceinline__ float bn_moving_update(float moving, float batch) {
  const float d = moving - batch;
  return moving - d * BN_UPDATE_RATE;
}
#pragma clang fp contract(fast)

// All workgroups of a launch arrive; writes made before are visible to every workgroup after.
// `bar` is a counter that is never reset; `target` = its value once all workgroups of this
// barrier have arrived (the host advances the base by workgroups x barriers per launch; the
// comparison is wrap-safe).  Only for launches whose workgroups are all co-resident (<= one per
// CU); should the others never arrive -- tens of seconds -- the launch aborts loudly rather than
// hang the queue.
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins == (1u << 28)) __builtin_trap();
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
#endif  // __HIPCC__

}  // namespace scvae
