// Coalesced elementwise / row-reduction kernels of the scVAE step (HBM-bound):
// count log-likelihoods, Gaussian reparameterisation + KL, ELBO, batch norm,
// clip + Adam, CSR densify, Philox normals.  One wavefront = 64 lanes.
#include "common.hpp"
#include "kernels.hpp"
#include "likelihood.hpp"

namespace scvae {

// ============================ log-likelihood ==============================

// One workgroup per row r of the [rows, F] head pre-activations; target row r % B.
// GRAD: overwrite pre_j with gw[r] * dlp/dpre_j.
template <int KIND, bool GRAD>
__global__ __launch_bounds__(256) void loglik_rows_kernel(const float* __restrict__ t, int ldt,
                                                          HeadPtrs pre, int ldp,
                                                          const float* __restrict__ gw,
                                                          const float* __restrict__ row_const,
                                                          float* __restrict__ ll, int B, int F) {
  constexpr int P = likelihood_heads(KIND);
  __shared__ float red[4];
  const int r = blockIdx.x;
  const int b = r % B;
  const float* trow = t + (size_t)b * ldt;
  const float g_up = GRAD ? gw[r] : 0.f;
  float acc = 0.f;
  for (int f = threadIdx.x; f < F; f += 256) {
    const float tv = trow[f];
    float a[P], g[P], lp;
#pragma unroll
    for (int j = 0; j < P; ++j) a[j] = pre.p[j][(size_t)r * ldp + f];
    lik_elem<KIND, GRAD>(tv, a, lp, g);
    acc += lp;
    if (row_const == nullptr) acc -= lgamma1p(tv);
    if (GRAD) {
#pragma unroll
      for (int j = 0; j < P; ++j) pre.p[j][(size_t)r * ldp + f] = g_up * g[j];
    }
  }
  acc = block_sum<256>(acc, red);
  if (threadIdx.x == 0 && ll != nullptr) ll[r] = acc - (row_const ? row_const[b] : 0.f);
}

template <bool GRAD>
static int launch_loglik(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre,
                         int ldp, const float* gw, const float* row_const, float* ll, int rows,
                         int B, int F) {
  SCVAE_ARG(t && pre.p[0] && rows >= 0 && B > 0 && F > 0);
  if (rows == 0) return 0;
  dim3 grid(rows), block(256);
  switch (kind) {
    case LK_POISSON:
      hipLaunchKernelGGL((loglik_rows_kernel<LK_POISSON, GRAD>), grid, block, 0, stream, t, ldt,
                         pre, ldp, gw, row_const, ll, B, F);
      break;
    case LK_NB:
      hipLaunchKernelGGL((loglik_rows_kernel<LK_NB, GRAD>), grid, block, 0, stream, t, ldt, pre,
                         ldp, gw, row_const, ll, B, F);
      break;
    case LK_ZIP:
      hipLaunchKernelGGL((loglik_rows_kernel<LK_ZIP, GRAD>), grid, block, 0, stream, t, ldt, pre,
                         ldp, gw, row_const, ll, B, F);
      break;
    case LK_ZINB:
      hipLaunchKernelGGL((loglik_rows_kernel<LK_ZINB, GRAD>), grid, block, 0, stream, t, ldt, pre,
                         ldp, gw, row_const, ll, B, F);
      break;
    case LK_BERNOULLI:
      hipLaunchKernelGGL((loglik_rows_kernel<LK_BERNOULLI, GRAD>), grid, block, 0, stream, t, ldt,
                         pre, ldp, gw, row_const, ll, B, F);
      break;
    default:
      set_error("unknown likelihood kind %d", kind);
      return -1;
  }
  SCVAE_LAUNCH_CHECK("loglik_rows_kernel");
  return 0;
}

int loglik_fwd(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre, int ldp,
               const float* row_const, float* ll, int rows, int B, int F) {
  SCVAE_ARG(ll);
  return launch_loglik<false>(stream, kind, t, ldt, pre, ldp, nullptr, row_const, ll, rows, B, F);
}

int loglik_bwd(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre, int ldp,
               const float* gw, const float* row_const, float* ll, int rows, int B, int F) {
  SCVAE_ARG(gw);
  return launch_loglik<true>(stream, kind, t, ldt, pre, ldp, gw, row_const, ll, rows, B, F);
}

// ---- constrained Poisson (du:218-228; N = count sum of the cell, va:2400-2405, 2490-2496) ----
// lambda = clip(softmax_F(pre), tiny, 1), rate = lambda * N,
//   sum_f log p(t_f) = sum_f t_f * log(rate_f) - rate_f - lgamma(1 + t_f).
// One workgroup per row; the row of logits is staged in LDS (F <= 38 000) and walked four times:
// max, sum of exponentials, log-probability (+ the sum S = sum_f gate_f (t_f - N lambda_f) the
// gradient needs), and, if GRAD, pre_g <- gw * (gate_g (t_g - N lambda_g) - lambda_g S).
// gate_f = [lambda_f >= tiny]: the clip's pass-through (always 1 unless a logit is 87 below the
// row maximum).  RATE: overwrite pre with the rate instead (evaluate-time statistics).
template <bool GRAD, bool RATE>
__global__ __launch_bounds__(1024) void cpoisson_rows_kernel(
    const float* __restrict__ t, int ldt, float* __restrict__ pre, int ldp,
    const float* __restrict__ gw, const float* __restrict__ count_sum,
    const float* __restrict__ row_const, float* __restrict__ ll, int B, int F) {
  extern __shared__ float row[];
  __shared__ float red[16];
  const int r = blockIdx.x, b = r % B;
  float* prow = pre + (size_t)r * ldp;
  const float* trow = t + (size_t)b * ldt;
  const float N = count_sum[b];
  float mx = -INFINITY;
  for (int f = threadIdx.x; f < F; f += 1024) {
    const float v = prow[f];
    row[f] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
  }
  float se = 0.f;
  for (int f = threadIdx.x; f < F; f += 1024) se += __expf(row[f] - mx);
  se = block_sum<1024>(se, red);
  const float lse = mx + __logf(se);
  const float log_n = __logf(fmaxf(N, F32_TINY));
  if (RATE) {
    for (int f = threadIdx.x; f < F; f += 1024)
      prow[f] = fmaxf(__expf(row[f] - lse), F32_TINY) * N;
    return;
  }
  float lp = 0.f, S = 0.f;
  for (int f = threadIdx.x; f < F; f += 1024) {
    const float tv = trow[f];
    const float log_lam = row[f] - lse;
    const float lam = __expf(log_lam);
    const bool gate = lam >= F32_TINY;
    const float lam_c = gate ? lam : F32_TINY;
    const float log_rate = (gate ? log_lam : LOG_F32_TINY) + log_n;
    lp += (tv > 0.f ? tv * log_rate : 0.f) - lam_c * N;
    if (row_const == nullptr) lp -= lgamma1p(tv);
    if (GRAD && gate) S += tv - N * lam;
  }
  lp = block_sum<1024>(lp, red);
  if (GRAD) {
    S = block_sum<1024>(S, red);
    const float g_up = gw[r];
    for (int f = threadIdx.x; f < F; f += 1024) {
      const float lam = __expf(row[f] - lse);
      const float own = lam >= F32_TINY ? trow[f] - N * lam : 0.f;
      prow[f] = g_up * (own - lam * S);
    }
  }
  if (threadIdx.x == 0 && ll != nullptr) ll[r] = lp - (row_const ? row_const[b] : 0.f);
}

template <bool GRAD, bool RATE>
static int launch_cpoisson(hipStream_t stream, const float* t, int ldt, float* pre, int ldp,
                           const float* gw, const float* count_sum, const float* row_const,
                           float* ll, int rows, int B, int F) {
  SCVAE_ARG(pre && count_sum && rows >= 0 && B > 0 && F > 0);
  SCVAE_ARG(RATE || t);
  if (rows == 0) return 0;
  const size_t lds = (size_t)F * sizeof(float);
  if (lds > 152 * 1024) {
    set_error("constrained Poisson: %d genes do not fit into LDS", F);
    return -1;
  }
  auto kfn = cpoisson_rows_kernel<GRAD, RATE>;
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)lds));
  hipLaunchKernelGGL(kfn, dim3(rows), dim3(1024), lds, stream, t, ldt, pre, ldp, gw, count_sum,
                     row_const, ll, B, F);
  SCVAE_LAUNCH_CHECK("cpoisson_rows_kernel");
  return 0;
}

int cpoisson_fwd(hipStream_t stream, const float* t, int ldt, float* pre, int ldp,
                 const float* count_sum, const float* row_const, float* ll, int rows, int B,
                 int F) {
  SCVAE_ARG(ll);
  return launch_cpoisson<false, false>(stream, t, ldt, pre, ldp, nullptr, count_sum, row_const, ll,
                                       rows, B, F);
}
int cpoisson_bwd(hipStream_t stream, const float* t, int ldt, float* pre, int ldp, const float* gw,
                 const float* count_sum, const float* row_const, float* ll, int rows, int B,
                 int F) {
  SCVAE_ARG(gw);
  return launch_cpoisson<true, false>(stream, t, ldt, pre, ldp, gw, count_sum, row_const, ll, rows,
                                      B, F);
}
int cpoisson_rate(hipStream_t stream, float* pre, int ldp, const float* count_sum, int rows, int B,
                  int F) {
  return launch_cpoisson<false, true>(stream, nullptr, 0, pre, ldp, nullptr, count_sum, nullptr,
                                      nullptr, rows, B, F);
}

// ---- piecewise categorical likelihood (-k) ----
// log-softmax over the K+1 logits of one (row, feature); returns the log-normaliser
__device__ __forceinline__ float cat_log_normaliser(const float* __restrict__ l, int C) {
  float m = l[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += __expf(l[c] - m);
  return m + __logf(sum);
}

template <int KIND, bool GRAD>
__global__ __launch_bounds__(256) void loglik_cat_rows_kernel(
    const float* __restrict__ t, int ldt, HeadPtrs pre, int ldp, float* __restrict__ logits, int K,
    const float* __restrict__ gw, float* __restrict__ ll, int B, int F) {
  constexpr int P = (KIND == LK_POISSON) ? 1 : 2;
  __shared__ float red[4];
  const int r = blockIdx.x;
  const int C = K + 1;
  const float* trow = t + (size_t)(r % B) * ldt;
  float* lrow = logits + (size_t)r * F * C;
  const float g_up = GRAD ? gw[r] : 0.f;
  float acc = 0.f;
  for (int f = threadIdx.x; f < F; f += 256) {
    const float tv = trow[f];
    float* l = lrow + (size_t)f * C;
    const float lse = cat_log_normaliser(l, C);
    // cast(clip(x, 0, K), int32), categorised.py:257-258
    const int cls = (int)fminf(fmaxf(tv, 0.f), (float)K);
    float lp = l[cls] - lse;
    float a[P], g[P];
#pragma unroll
    for (int j = 0; j < P; ++j) { a[j] = pre.p[j][(size_t)r * ldp + f]; g[j] = 0.f; }
    if (tv >= (float)K) {
      float lpd;
      lik_elem<KIND, GRAD>(tv - (float)K, a, lpd, g);
      lp += lpd - lgamma1p(tv - (float)K);
    }
    acc += lp;
    if (GRAD) {
#pragma unroll
      for (int j = 0; j < P; ++j) pre.p[j][(size_t)r * ldp + f] = g_up * g[j];
      for (int c = 0; c < C; ++c) {
        const float soft = __expf(l[c] - lse);
        l[c] = g_up * ((c == cls ? 1.f : 0.f) - soft);
      }
    }
  }
  acc = block_sum<256>(acc, red);
  if (threadIdx.x == 0 && ll != nullptr) ll[r] = acc;
}

template <bool GRAD>
static int launch_loglik_cat(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre,
                             int ldp, float* logits, int K, const float* gw, float* ll, int rows,
                             int B, int F) {
  SCVAE_ARG(t && pre.p[0] && logits && K > 0 && K <= 64 && rows >= 0 && B > 0 && F > 0);
  if (rows == 0) return 0;
  dim3 grid(rows), block(256);
  switch (kind) {
    case LK_POISSON:
      hipLaunchKernelGGL((loglik_cat_rows_kernel<LK_POISSON, GRAD>), grid, block, 0, stream, t, ldt,
                         pre, ldp, logits, K, gw, ll, B, F);
      break;
    case LK_NB:
      hipLaunchKernelGGL((loglik_cat_rows_kernel<LK_NB, GRAD>), grid, block, 0, stream, t, ldt, pre,
                         ldp, logits, K, gw, ll, B, F);
      break;
    default:
      set_error("the piecewise categorical likelihood wraps Poisson or negative binomial only");
      return -1;
  }
  SCVAE_LAUNCH_CHECK("loglik_cat_rows_kernel");
  return 0;
}
int loglik_cat_fwd(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre, int ldp,
                   float* logits, int K, float* ll, int rows, int B, int F) {
  SCVAE_ARG(ll);
  return launch_loglik_cat<false>(stream, kind, t, ldt, pre, ldp, logits, K, nullptr, ll, rows, B,
                                  F);
}
int loglik_cat_bwd(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre, int ldp,
                   float* logits, int K, const float* gw, float* ll, int rows, int B, int F) {
  SCVAE_ARG(gw);
  return launch_loglik_cat<true>(stream, kind, t, ldt, pre, ldp, logits, K, gw, ll, rows, B, F);
}

// mean and variance of Categorised(dist, cat) (categorised.py:210-253)
template <int KIND>
__device__ __forceinline__ void cat_mean_var(const float* a, const float* __restrict__ l, int K,
                                             float& mean, float& var) {
  const int C = K + 1;
  const float lse = cat_log_normaliser(l, C);
  float m1 = 0.f, m2 = 0.f;
  for (int c = 0; c < K; ++c) {
    const float pi = __expf(l[c] - lse);
    m1 += (float)c * pi;
    m2 += (float)(c * c) * pi;
  }
  const float tail = __expf(l[K] - lse);
  float dm, dv;
  lik_mean_var<KIND>(a, dm, dv);
  const float k = (float)K;
  mean = m1 + tail * (dm + k);
  var = m2 + tail * (2.f * k * dm + dv + dm * dm + k * k) - mean * mean;
}

template <int KIND>
__global__ __launch_bounds__(256) void px_statistics_cat_kernel(
    HeadPtrs pre, int ldp, const float* __restrict__ logits, int K, int S, int B, int F,
    const float* __restrict__ weight, int ldw, int accumulate, float* __restrict__ p_x_mean,
    float* __restrict__ mean_of_var, float* __restrict__ var_of_mean) {
  constexpr int P = (KIND == LK_POISSON) ? 1 : 2;
  const int b = blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const int C = K + 1;
  const float wgt = weight ? weight[(size_t)b * ldw] : 1.f;
  const float inv_s = 1.f / (float)S;
  float ms = 0.f, vs = 0.f;
  for (int s = 0; s < S; ++s) {
    const size_t row = (size_t)s * B + b;
    float a[P], m, v;
#pragma unroll
    for (int j = 0; j < P; ++j) a[j] = pre.p[j][row * ldp + f];
    cat_mean_var<KIND>(a, logits + (row * F + f) * C, K, m, v);
    ms += m; vs += v;
  }
  const float pm = ms * inv_s * wgt;
  float vom = 0.f;
  for (int s = 0; s < S; ++s) {
    const size_t row = (size_t)s * B + b;
    float a[P], m, v;
#pragma unroll
    for (int j = 0; j < P; ++j) a[j] = pre.p[j][row * ldp + f];
    cat_mean_var<KIND>(a, logits + (row * F + f) * C, K, m, v);
    vom += (m - pm) * (m - pm);
  }
  vom *= inv_s * wgt;
  const float mov = vs * inv_s * wgt;
  const size_t o = (size_t)b * F + f;
  if (accumulate) { p_x_mean[o] += pm; mean_of_var[o] += mov; var_of_mean[o] += vom; }
  else { p_x_mean[o] = pm; mean_of_var[o] = mov; var_of_mean[o] = vom; }
}

int px_statistics_cat(hipStream_t stream, int kind, HeadPtrs pre, int ldp, const float* logits,
                      int K, int S, int B, int F, const float* weight, int ldw, int accumulate,
                      float* p_x_mean, float* mean_of_var, float* var_of_mean) {
  SCVAE_ARG(pre.p[0] && logits && K > 0 && p_x_mean && mean_of_var && var_of_mean && S > 0 && F > 0);
  if (B == 0) return 0;
  dim3 grid((F + 255) / 256, B), block(256);
  switch (kind) {
    case LK_POISSON:
      hipLaunchKernelGGL((px_statistics_cat_kernel<LK_POISSON>), grid, block, 0, stream, pre, ldp,
                         logits, K, S, B, F, weight, ldw, accumulate, p_x_mean, mean_of_var,
                         var_of_mean);
      break;
    case LK_NB:
      hipLaunchKernelGGL((px_statistics_cat_kernel<LK_NB>), grid, block, 0, stream, pre, ldp, logits,
                         K, S, B, F, weight, ldw, accumulate, p_x_mean, mean_of_var, var_of_mean);
      break;
    default:
      set_error("the piecewise categorical likelihood wraps Poisson or negative binomial only");
      return -1;
  }
  SCVAE_LAUNCH_CHECK("px_statistics_cat_kernel");
  return 0;
}

template <int KIND>
__global__ __launch_bounds__(256) void px_statistics_kernel(HeadPtrs pre, int ldp, int S, int B,
                                                            int F, const float* __restrict__ weight,
                                                            int ldw, int accumulate,
                                                            float* __restrict__ p_x_mean,
                                                            float* __restrict__ mean_of_var,
                                                            float* __restrict__ var_of_mean) {
  constexpr int P = likelihood_heads(KIND);
  const int b = blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const float wgt = weight ? weight[(size_t)b * ldw] : 1.f;
  const float inv_s = 1.f / (float)S;
  float ms = 0.f, vs = 0.f;
  for (int s = 0; s < S; ++s) {
    float a[P], m, v;
#pragma unroll
    for (int j = 0; j < P; ++j) a[j] = pre.p[j][((size_t)s * B + b) * ldp + f];
    lik_mean_var<KIND>(a, m, v);
    ms += m; vs += v;
  }
  const float pm = ms * inv_s * wgt;
  float vom = 0.f;
  for (int s = 0; s < S; ++s) {
    float a[P], m, v;
#pragma unroll
    for (int j = 0; j < P; ++j) a[j] = pre.p[j][((size_t)s * B + b) * ldp + f];
    lik_mean_var<KIND>(a, m, v);
    vom += (m - pm) * (m - pm);
  }
  vom *= inv_s * wgt;
  const float mov = vs * inv_s * wgt;
  const size_t o = (size_t)b * F + f;
  if (accumulate) { p_x_mean[o] += pm; mean_of_var[o] += mov; var_of_mean[o] += vom; }
  else { p_x_mean[o] = pm; mean_of_var[o] = mov; var_of_mean[o] = vom; }
}

int px_statistics(hipStream_t stream, int kind, HeadPtrs pre, int ldp, int S, int B, int F,
                  const float* weight, int ldw, int accumulate, float* p_x_mean,
                  float* mean_of_var, float* var_of_mean) {
  SCVAE_ARG(pre.p[0] && p_x_mean && mean_of_var && var_of_mean && S > 0 && F > 0);
  if (B == 0) return 0;
  dim3 grid((F + 255) / 256, B), block(256);
#define SCVAE_PX(K_)                                                                             \
  hipLaunchKernelGGL((px_statistics_kernel<K_>), grid, block, 0, stream, pre, ldp, S, B, F, weight, \
                     ldw, accumulate, p_x_mean, mean_of_var, var_of_mean)
  switch (kind) {
    case LK_POISSON: SCVAE_PX(LK_POISSON); break;
    case LK_NB: SCVAE_PX(LK_NB); break;
    case LK_ZIP: SCVAE_PX(LK_ZIP); break;
    case LK_ZINB: SCVAE_PX(LK_ZINB); break;
    case LK_CPOISSON: SCVAE_PX(LK_CPOISSON); break;
    case LK_BERNOULLI: SCVAE_PX(LK_BERNOULLI); break;
    default: set_error("unknown likelihood kind %d", kind); return -1;
  }
#undef SCVAE_PX
  SCVAE_LAUNCH_CHECK("px_statistics_kernel");
  return 0;
}

// element-wise log p(t | theta) (including -lgamma(1+t)) and mean/variance, for the
// DISTRIBUTIONS plugin surface (.log_prob / .mean / .variance of the registry classes)
template <int KIND>
__global__ __launch_bounds__(256) void loglik_elementwise_kernel(const float* __restrict__ t,
                                                                 HeadPtrs pre,
                                                                 float* __restrict__ out,
                                                                 float* __restrict__ mean,
                                                                 float* __restrict__ var, size_t n) {
  constexpr int P = likelihood_heads(KIND);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float a[P], g[P];
#pragma unroll
    for (int j = 0; j < P; ++j) a[j] = pre.p[j][i];
    if (out != nullptr) {
      const float tv = t[i];
      float lp;
      lik_elem<KIND, false>(tv, a, lp, g);
      out[i] = lp - lgamma1p(tv);
    }
    if (mean != nullptr) {
      float m, v;
      lik_mean_var<KIND>(a, m, v);
      mean[i] = m;
      var[i] = v;
    }
  }
}

int loglik_elementwise(hipStream_t stream, int kind, const float* t, HeadPtrs pre, float* out,
                       float* mean, float* var, size_t n) {
  SCVAE_ARG(pre.p[0] && (out || mean));
  SCVAE_ARG(!out || t);
  SCVAE_ARG(!mean || var);
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
#define SCVAE_LE(K_)                                                                            \
  hipLaunchKernelGGL((loglik_elementwise_kernel<K_>), dim3(blocks), dim3(256), 0, stream, t, pre, \
                     out, mean, var, n)
  switch (kind) {
    case LK_POISSON: SCVAE_LE(LK_POISSON); break;
    case LK_NB: SCVAE_LE(LK_NB); break;
    case LK_ZIP: SCVAE_LE(LK_ZIP); break;
    case LK_ZINB: SCVAE_LE(LK_ZINB); break;
    case LK_BERNOULLI: SCVAE_LE(LK_BERNOULLI); break;
    default: set_error("unknown likelihood kind %d", kind); return -1;
  }
#undef SCVAE_LE
  SCVAE_LAUNCH_CHECK("loglik_elementwise_kernel");
  return 0;
}

__global__ void sqrt_sum_kernel(const float* a, const float* b, float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = sqrtf(a[i] + (b ? b[i] : 0.f));
}
int sqrt_sum(hipStream_t stream, const float* a, const float* b, float* out, size_t n) {
  SCVAE_ARG(a && out);
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sqrt_sum_kernel, dim3(blocks), dim3(256), 0, stream, a, b, out, n);
  SCVAE_LAUNCH_CHECK("sqrt_sum_kernel");
  return 0;
}

// ============================ Gaussian latent =============================

// one workgroup (64*ceil(L/64) threads) per cell b.  ls_pre == nullptr: the posterior's
// log_sigma is the constant 0 ("unit-variance gaussian", du:323-337).  kl_sample != nullptr:
// Monte-Carlo KL (va:2633-2640), log q(z|x) - log p(z) at every sample,
//   kl[s,b,l] = (z^2 - eps^2) / 2 - log sigma,   kl_sample[s*B + b] = sum_l kl[s,b,l],
// and kl_elem[b,l] = mean_s kl[s,b,l] (so that its column mean is kl_divergence_neurons).
__global__ void gauss_latent_fwd_kernel(const float* __restrict__ mu_pre,
                                        const float* __restrict__ ls_pre,
                                        const float* __restrict__ eps, float* __restrict__ z,
                                        float* __restrict__ kl_elem, float* __restrict__ kl_cell,
                                        float* __restrict__ kl_sample, int S, int B, int L,
                                        int deterministic) {
  __shared__ float red[16];
  const int b = blockIdx.x, l = threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const bool live = l < L;
  const size_t i = (size_t)b * L + (live ? l : 0);
  const float mu = fminf(fmaxf(mu_pre[i], -F32_MAX_HALF), F32_MAX_HALF);
  const float ls = ls_pre ? fminf(fmaxf(ls_pre[i], -3.f), 3.f) : 0.f;
  const float sigma = ls_pre ? __expf(ls) : 1.f;
  if (kl_sample == nullptr) {
    float kl = 0.f;
    if (live) {
      if (deterministic) {
        z[i] = mu;
      } else {
        for (int s = 0; s < S; ++s) {
          const size_t o = ((size_t)s * B + b) * L + l;
          z[o] = fmaf(sigma, eps[o], mu);
        }
      }
      kl = gauss_kl_elem(mu, sigma, ls);
      kl_elem[i] = kl;
    }
    // block reduction over L (blockDim.x is a multiple of 64, <= 1024)
    kl = wave_sum(kl);
    if (lane == 0) red[w] = kl;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int q = 0; q < nw; ++q) s += red[q];
      kl_cell[b] = s;
    }
    return;
  }
  float mean_kl = 0.f;
  for (int s = 0; s < S; ++s) {
    float kl = 0.f;
    if (live) {
      const size_t o = ((size_t)s * B + b) * L + l;
      const float e = deterministic ? 0.f : eps[o];
      const float zz = fmaf(sigma, e, mu);
      z[o] = zz;
      kl = 0.5f * (zz * zz - e * e) - ls;
      mean_kl += kl;
    }
    kl = wave_sum(kl);
    if (nw > 1) {
      __syncthreads();
      if (lane == 0) red[w] = kl;
      __syncthreads();
      kl = 0.f;
      for (int q = 0; q < nw; ++q) kl += red[q];
    }
    if (threadIdx.x == 0) kl_sample[(size_t)s * B + b] = kl;
  }
  if (live) kl_elem[i] = mean_kl / (float)S;
}

int gauss_latent_fwd(hipStream_t stream, const float* mu_pre, const float* ls_pre,
                     const float* eps, float* z, float* kl_elem, float* kl_cell,
                     float* kl_sample, int S, int B, int L, int deterministic) {
  SCVAE_ARG(mu_pre && z && kl_elem && (kl_cell || kl_sample));
  SCVAE_ARG(deterministic || eps);
  SCVAE_ARG(L > 0 && L <= 1024 && S > 0);
  if (B == 0) return 0;
  const int threads = (L + 63) / 64 * 64;
  hipLaunchKernelGGL(gauss_latent_fwd_kernel, dim3(B), dim3(threads), 0, stream, mu_pre, ls_pre,
                     eps, z, kl_elem, kl_cell, kl_sample, S, B, L, deterministic);
  SCVAE_LAUNCH_CHECK("gauss_latent_fwd_kernel");
  return 0;
}

// Analytic KL (kl_gw == nullptr): d loss / d KL_cell = kl_coeff for every cell.  Monte-Carlo KL:
// d loss / d kl[s,b] = c_sb = -kl_coeff * kl_gw[s*B + b] (kl_gw = d loss / d log p(x|z) of the
// row, kl_coeff = the KL weight), and with dz' = dz + c_sb * z
//   d mu = sum_s dz',   d log_sigma = sigma * sum_s dz' * eps - sum_s c_sb.
__global__ void gauss_latent_bwd_kernel(const float* __restrict__ mu_pre,
                                        const float* __restrict__ ls_pre,
                                        const float* __restrict__ eps,
                                        const float* __restrict__ dz, float kl_coeff,
                                        const float* __restrict__ kl_gw,
                                        float* __restrict__ dmu_pre, float* __restrict__ dls_pre,
                                        int S, int B, int L) {
  const size_t n = (size_t)B * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float mp = mu_pre[i], lp = ls_pre ? ls_pre[i] : 0.f;
    const float mu = fminf(fmaxf(mp, -F32_MAX_HALF), F32_MAX_HALF);
    const float ls = fminf(fmaxf(lp, -3.f), 3.f);
    const float sigma = ls_pre ? __expf(ls) : 1.f;
    const size_t b = i / (size_t)L;
    float gz = 0.f, gze = 0.f, csum = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t o = (size_t)s * n + i;
      const float e = eps[o];
      float d = dz[o];
      if (kl_gw) {
        const float c = -kl_coeff * kl_gw[(size_t)s * B + b];
        d = fmaf(c, fmaf(sigma, e, mu), d);
        csum += c;
      }
      gz += d;
      gze = fmaf(d, e, gze);
    }
    float gmu, gls;
    if (kl_gw) {
      gmu = gz;
      gls = gze * sigma - csum;
    } else {
      gmu = gauss_kl_dmu(gz, kl_coeff, mu);
      gls = gauss_kl_dls(gze, sigma, kl_coeff);
    }
    dmu_pre[i] = (mp >= -F32_MAX_HALF && mp <= F32_MAX_HALF) ? gmu : 0.f;
    if (dls_pre) dls_pre[i] = (lp >= -3.f && lp <= 3.f) ? gls : 0.f;
  }
}

int gauss_latent_bwd(hipStream_t stream, const float* mu_pre, const float* ls_pre,
                     const float* eps, const float* dz, float kl_coeff, const float* kl_gw,
                     float* dmu_pre, float* dls_pre, int S, int B, int L) {
  SCVAE_ARG(mu_pre && eps && dz && dmu_pre);
  SCVAE_ARG((ls_pre == nullptr) == (dls_pre == nullptr));
  if (B == 0) return 0;
  const size_t n = (size_t)B * L;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gauss_latent_bwd_kernel, dim3(blocks), dim3(256), 0, stream, mu_pre, ls_pre,
                     eps, dz, kl_coeff, kl_gw, dmu_pre, dls_pre, S, B, L);
  SCVAE_LAUNCH_CHECK("gauss_latent_bwd_kernel");
  return 0;
}

// ================================= ELBO ===================================

// single workgroup; ll index = (r*n_mc + m)*B + b.  row_scale = 1/(n_mc*B_global)
// lets a data-parallel rank emit its share of the global means (summed by all-reduce).
__global__ __launch_bounds__(1024) void vae_elbo_kernel(const float* __restrict__ ll,
                                                       const float* __restrict__ kl_cell,
                                                       int kl_per_sample, int n_iw, int n_mc,
                                                       int B, float w, float row_scale,
                                                       float* __restrict__ scalars,
                                                       float* __restrict__ gw) {
  __shared__ float red[16];
  float lb = 0.f, lbw = 0.f, rec = 0.f, klsum = 0.f;
  const int pairs = n_mc * B;
  if (n_iw == 1 && !kl_per_sample && gw == nullptr) {
    // the training step's case: one sample row per pair, log-mean-exp of one term (= the term:
    // the general path below computes log(exp(0) * 1) + x, the same bits).  Eight pairs of loads
    // in flight per thread instead of a dependent round trip per pair.
    for (int i0 = threadIdx.x; i0 < pairs; i0 += 8 * 1024) {
      float v[8], kl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(i0 + 1024 * u, pairs - 1);
        v[u] = ll[i];
        kl[u] = kl_cell[i % B];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 1024 * u;
        if (i < pairs) {
          if (i < B) klsum += kl[u];
          rec += v[u];
          lb += __logf(1.f) + (v[u] - kl[u]);
          lbw += __logf(1.f) + (v[u] - w * kl[u]);
        }
      }
    }
    lb = block_sum<1024>(lb, red);
    lbw = block_sum<1024>(lbw, red);
    rec = block_sum<1024>(rec, red);
    klsum = block_sum<1024>(klsum, red);
    if (threadIdx.x == 0) {
      scalars[0] = lb * row_scale;
      scalars[1] = lbw * row_scale;
      scalars[2] = rec * row_scale;
      scalars[3] = klsum * row_scale * (float)n_mc;
      if (!isfinite(lb)) scalars[7] += 1.f;
    }
    return;
  }
  // (m, b) advance without a division per element: 1024 = q*B + rem
  const int step_m = 1024 / B, step_b = 1024 % B;
  int m = threadIdx.x / B, b = threadIdx.x % B;
  for (int i = threadIdx.x; i < pairs; i += 1024) {
    // analytic KL: one value per cell; Monte-Carlo KL: one per sample row (va:2656)
    float kl = kl_cell[b];
    if (!kl_per_sample && m == 0) klsum += kl;
    float mx = -INFINITY, mxw = -INFINITY;
    for (int r = 0; r < n_iw; ++r) {
      const size_t o = ((size_t)r * n_mc + m) * B + b;
      const float v = ll[o];
      if (kl_per_sample) { kl = kl_cell[o]; klsum += kl; }
      rec += v;
      mx = fmaxf(mx, v - kl);
      mxw = fmaxf(mxw, v - w * kl);
    }
    float se = 0.f, sew = 0.f;
    for (int r = 0; r < n_iw; ++r) {
      const size_t o = ((size_t)r * n_mc + m) * B + b;
      const float v = ll[o];
      if (kl_per_sample) kl = kl_cell[o];
      se += __expf(v - kl - mx);
      sew += __expf(v - w * kl - mxw);
    }
    const float inv = 1.f / (float)n_iw;
    lb += __logf(se * inv) + mx;
    lbw += __logf(sew * inv) + mxw;
    if (gw != nullptr) {
      for (int r = 0; r < n_iw; ++r) {
        const size_t o = ((size_t)r * n_mc + m) * B + b;
        if (kl_per_sample) kl = kl_cell[o];
        gw[o] = -__expf(ll[o] - w * kl - mxw) / sew * row_scale;
      }
    }
    m += step_m; b += step_b;
    if (b >= B) { b -= B; ++m; }
  }
  lb = block_sum<1024>(lb, red);
  lbw = block_sum<1024>(lbw, red);
  rec = block_sum<1024>(rec, red);
  klsum = block_sum<1024>(klsum, red);
  if (threadIdx.x == 0) {
    scalars[0] = lb * row_scale;
    scalars[1] = lbw * row_scale;
    scalars[2] = rec * row_scale / (float)n_iw;
    scalars[3] = kl_per_sample ? klsum * row_scale / (float)n_iw : klsum * row_scale * (float)n_mc;
    // sticky: executions since the caller last zeroed the buffer whose ELBO was not finite
    // (the reference tests the loss at the steps it prints, va:1034-1044: polled there)
    if (!isfinite(lb)) scalars[7] += 1.f;
  }
}

int vae_elbo(hipStream_t stream, const float* ll, const float* kl_cell, int kl_per_sample,
             int n_iw, int n_mc, int B, float kl_weight_total, float row_scale, float* scalars,
             float* gw) {
  SCVAE_ARG(ll && kl_cell && scalars && n_iw > 0 && n_mc > 0 && B > 0);
  hipLaunchKernelGGL(vae_elbo_kernel, dim3(1), dim3(1024), 0, stream, ll, kl_cell, kl_per_sample,
                     n_iw, n_mc, B, kl_weight_total, row_scale, scalars, gw);
  SCVAE_LAUNCH_CHECK("vae_elbo_kernel");
  return 0;
}

// ============================ batch normalisation =========================

// Row-chunked statistics: grid (ceil(N/64), groups, chunks), 64 columns x 16 row lanes per
// workgroup.  Each workgroup computes the two-pass (centred) mean / M2 of its row chunk
// (second pass re-reads the chunk from cache); bn_stats_finalize_kernel merges the chunks
// with the parallel-variance formula in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(1024) void bn_stats_partial_kernel(const float* __restrict__ a,
                                                                int lda, int R, int N, int chunk,
                                                                float* __restrict__ partial) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + cl;
  const int rl = threadIdx.x >> 6;
  const int g = blockIdx.y, z = blockIdx.z, Z = gridDim.z, G = gridDim.y;
  const int r0 = z * chunk, r1 = min(R, r0 + chunk);
  const int n = r1 - r0;
  const float* base = a + (size_t)g * R * lda;
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 16) s += base[(size_t)r * lda + c];
  red[rl][cl] = s;
  __syncthreads();
  float mu = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mu += red[i][cl];
  mu = n > 0 ? mu / (float)n : 0.f;
  __syncthreads();
  float q = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 16) {
      const float d = base[(size_t)r * lda + c] - mu;
      q = fmaf(d, d, q);
    }
  red[rl][cl] = q;
  __syncthreads();
  if (rl == 0 && c < N) {
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m2 += red[i][cl];
    float* out = partial + (((size_t)z * G + g) * 2) * N;
    out[c] = mu;
    out[N + c] = m2;
  }
  (void)Z;
}

// 64 columns x 16 chunk lanes per workgroup; the chunk lanes are combined in a fixed order
// (lane l takes chunks l, l+16, ...; then lanes 0..15 in sequence): deterministic.
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(
    const float* __restrict__ partial, int R, int N, int chunk, int chunks, int G,
    float* __restrict__ mean, float* __restrict__ var) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int g = blockIdx.y;
  float acc = 0.f;
  if (c < N)
    for (int z = zl; z < chunks; z += 16) {
      const int n = min(R, (z + 1) * chunk) - z * chunk;
      acc = bn_merge_mean(acc, (float)n, partial[(((size_t)z * G + g) * 2) * N + c]);
    }
  red[zl][cl] = acc;
  __syncthreads();
  float mu = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mu += red[i][cl];
  mu /= (float)R;
  __syncthreads();
  acc = 0.f;
  if (c < N)
    for (int z = zl; z < chunks; z += 16) {
      const int n = min(R, (z + 1) * chunk) - z * chunk;
      const float* pz = partial + (((size_t)z * G + g) * 2) * N;
      acc = bn_merge_m2(acc, (float)n, pz[c], pz[N + c], mu);
    }
  red[zl][cl] = acc;
  __syncthreads();
  if (zl == 0 && c < N) {
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m2 += red[i][cl];
    mean[(size_t)g * N + c] = mu;
    var[(size_t)g * N + c] = m2 / (float)R;
  }
}

static inline int bn_chunks(int R, int* chunk) {
  int chunks = (R + 63) / 64;   // 64-row chunks: 4 rows per thread, enough workgroups to fill the chip
  if (chunks > BN_MAX_CHUNKS) chunks = BN_MAX_CHUNKS;
  if (chunks < 1) chunks = 1;
  *chunk = (R + chunks - 1) / chunks;
  return (R + *chunk - 1) / *chunk;
}

size_t bn_partial_floats(int groups, int N) { return (size_t)BN_MAX_CHUNKS * groups * 2 * N; }

int bn_stats(hipStream_t stream, const float* a, int lda, int rows_per_group, int groups, int N,
             float* mean, float* var, float* partial) {
  SCVAE_ARG(a && mean && var && partial && rows_per_group > 0 && groups > 0 && N > 0);
  int chunk;
  const int chunks = bn_chunks(rows_per_group, &chunk);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((N + 63) / 64, groups, chunks), dim3(1024), 0,
                     stream, a, lda, rows_per_group, N, chunk, partial);
  SCVAE_LAUNCH_CHECK("bn_stats_partial_kernel");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((N + 63) / 64, groups), dim3(1024), 0, stream,
                     partial, rows_per_group, N, chunk, chunks, groups, mean, var);
  SCVAE_LAUNCH_CHECK("bn_stats_finalize_kernel");
  return 0;
}

// the chunk statistics alone (one group): merged by the consumer (tilechain.hip)
int bn_stats_partial(hipStream_t stream, const float* a, int lda, int rows, int N, float* partial,
                     int* chunk_out, int* chunks_out) {
  SCVAE_ARG(a && partial && rows > 0 && N > 0 && chunk_out && chunks_out);
  int chunk;
  const int chunks = bn_chunks(rows, &chunk);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((N + 63) / 64, 1, chunks), dim3(1024), 0, stream,
                     a, lda, rows, N, chunk, partial);
  SCVAE_LAUNCH_CHECK("bn_stats_partial_kernel");
  *chunk_out = chunk;
  *chunks_out = chunks;
  return 0;
}

// h = [relu]((a - mean) * rsqrt(var + eps) + beta); stat_stride = N for per-group batch
// statistics, 0 for the moving statistics shared by all groups (is_training=False).
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ a, int lda,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ var,
                                                       int stat_stride,
                                                       const float* __restrict__ beta,
                                                       float* __restrict__ h, int ldh, int R,
                                                       int groups, int N, int relu) {
  const size_t total = (size_t)R * groups * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % N);
    const size_t row = i / N;
    const int g = (int)(row / R);
    const float mu = mean[(size_t)g * stat_stride + c];
    const float istd = rsqrtf(var[(size_t)g * stat_stride + c] + BN_EPSILON);
    float v = bn_normalise(a[row * lda + c], mu, istd, beta[c]);
    if (relu) v = fmaxf(v, 0.f);
    h[row * ldh + c] = v;
  }
}

int bn_apply(hipStream_t stream, const float* a, int lda, const float* mean, const float* var,
             int stat_stride, const float* beta, float* h, int ldh, int rows_per_group, int groups,
             int N, int relu) {
  SCVAE_ARG(a && mean && var && beta && h);
  const size_t total = (size_t)rows_per_group * groups * N;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, stream, a, lda, mean, var,
                     stat_stride, beta, h, ldh, rows_per_group, groups, N, relu);
  SCVAE_LAUNCH_CHECK("bn_apply_kernel");
  return 0;
}

// s1[g,c] = sum_r dA, s2[g,c] = sum_r dA * xhat with dA = dh * (h > 0) [relu];
// row-chunked partial sums + fixed-order finalize
__global__ __launch_bounds__(1024) void bn_bwd_stats_partial_kernel(
    const float* __restrict__ dh, int lddh, const float* __restrict__ h, int ldh,
    const float* __restrict__ a, int lda, const float* __restrict__ mean,
    const float* __restrict__ var, int R, int N, int relu, int chunk,
    float* __restrict__ partial) {
  __shared__ float red1[16][64];
  __shared__ float red2[16][64];
  const int cl = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + cl;
  const int rl = threadIdx.x >> 6;
  const int g = blockIdx.y, z = blockIdx.z, G = gridDim.y;
  const int r0 = z * chunk, r1 = min(R, r0 + chunk);
  float a1 = 0.f, a2 = 0.f;
  if (c < N) {
    const float mu = mean[(size_t)g * N + c];
    const float istd = rsqrtf(var[(size_t)g * N + c] + BN_EPSILON);
    for (int r = r0 + rl; r < r1; r += 16) {
      const size_t row = (size_t)g * R + r;
      float d = dh[row * lddh + c];
      if (relu && !(h[row * ldh + c] > 0.f)) d = 0.f;
      const float xh = (a[row * lda + c] - mu) * istd;
      a1 += d;
      a2 = fmaf(d, xh, a2);
    }
  }
  red1[rl][cl] = a1;
  red2[rl][cl] = a2;
  __syncthreads();
  if (rl == 0 && c < N) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { t1 += red1[i][cl]; t2 += red2[i][cl]; }
    float* out = partial + (((size_t)z * G + g) * 2) * N;
    out[c] = t1;
    out[N + c] = t2;
  }
}

// Fixed-order merge of the row-chunk partials.  The group-0 workgroups also do the two small
// per-layer jobs that need nothing else: dbeta = sum over groups of s1 (this rank's rows, before
// any data-parallel exchange of s1) and the moving-average update of the layer's batch statistics
// (UPDATE_OPS, va:2763-2768; group after group, Bessel-corrected variance).
__global__ __launch_bounds__(1024) void bn_bwd_stats_finalize_kernel(
    const float* __restrict__ partial, int N, int chunks, int G, float* __restrict__ s1,
    float* __restrict__ s2, float* __restrict__ dbeta, const float* __restrict__ mean,
    const float* __restrict__ var, float* __restrict__ moving_mean,
    float* __restrict__ moving_var, float bessel) {
  __shared__ float red1[16][64];
  __shared__ float red2[16][64];
  const int cl = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int g = blockIdx.y;
  // chunk lane zl sums chunks zl, zl+16, ...; lanes combined in sequence (fixed order)
  auto group_sums = [&](int q, float& t1, float& t2) {
    float a1 = 0.f, a2 = 0.f;
    if (c < N)
      for (int z = zl; z < chunks; z += 16) {
        const float* pz = partial + (((size_t)z * G + q) * 2) * N;
        a1 += pz[c];
        a2 += pz[N + c];
      }
    __syncthreads();   // (previous use of the buffers)
    red1[zl][cl] = a1;
    red2[zl][cl] = a2;
    __syncthreads();
    t1 = 0.f; t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { t1 += red1[i][cl]; t2 += red2[i][cl]; }
  };
  float t1, t2;
  group_sums(g, t1, t2);
  const bool writer = zl == 0 && c < N;
  if (writer) {
    s1[(size_t)g * N + c] = t1;
    s2[(size_t)g * N + c] = t2;
  }
  if (g != 0) return;   // (uniform per workgroup)
  // (the loads of eight groups in flight at a time, from clamped addresses; the sums keep the
  //  order of group_sums -- chunk lanes in sequence, groups in sequence: the same bits.  As a loop
  //  over group_sums this tail was a dependent global-memory round trip per group: 18 us for the
  //  GMVAE's 20 passes)
  if (dbeta != nullptr) {
    float total = t1;
    const int cc = min(c, N - 1);
    for (int q0 = 1; q0 < G; q0 += 8) {
      float a1q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = min(q0 + u, G - 1);
        float a1 = 0.f;
        for (int z = zl; z < chunks; z += 16) a1 += partial[(((size_t)z * G + q) * 2) * N + cc];
        a1q[u] = a1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (q0 + u < G) {          // (uniform)
          __syncthreads();
          red1[zl][cl] = c < N ? a1q[u] : 0.f;
          __syncthreads();
          float u1 = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) u1 += red1[i][cl];
          total += u1;
        }
      }
    }
    if (writer) dbeta[c] = total;
  }
  if (moving_mean != nullptr && writer) {
    float mm = moving_mean[c], mv = moving_var[c];
    for (int q0 = 0; q0 < G; q0 += 8) {
      float mq[8], vq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = min(q0 + u, G - 1);
        mq[u] = mean[(size_t)q * N + c];
        vq[u] = var[(size_t)q * N + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (q0 + u < G) {
          mm = bn_moving_update(mm, mq[u]);
          mv = bn_moving_update(mv, vq[u] * bessel);
        }
      }
    }
    moving_mean[c] = mm;
    moving_var[c] = mv;
  }
}

int bn_bwd_stats(hipStream_t stream, const float* dh, int lddh, const float* h, int ldh,
                 const float* a, int lda, const float* mean, const float* var, int rows_per_group,
                 int groups, int N, int relu, float* s1, float* s2, float* partial, float* dbeta,
                 float* moving_mean, float* moving_var, int64_t global_rows_per_group) {
  SCVAE_ARG(dh && h && a && mean && var && s1 && s2 && partial);
  SCVAE_ARG((moving_mean == nullptr) == (moving_var == nullptr));
  int chunk;
  const int chunks = bn_chunks(rows_per_group, &chunk);
  hipLaunchKernelGGL(bn_bwd_stats_partial_kernel, dim3((N + 63) / 64, groups, chunks), dim3(1024),
                     0, stream, dh, lddh, h, ldh, a, lda, mean, var, rows_per_group, N, relu, chunk,
                     partial);
  SCVAE_LAUNCH_CHECK("bn_bwd_stats_partial_kernel");
  const int64_t R = global_rows_per_group;
  const float bessel = (float)R / (float)(R > 1 ? R - 1 : 1);
  hipLaunchKernelGGL(bn_bwd_stats_finalize_kernel, dim3((N + 63) / 64, groups), dim3(1024), 0,
                     stream, partial, N, chunks, groups, s1, s2, dbeta, mean, var, moving_mean,
                     moving_var, bessel);
  SCVAE_LAUNCH_CHECK("bn_bwd_stats_finalize_kernel");
  return 0;
}

// da = istd * (dA - s1*inv_count - xhat * s2*inv_count); inv_count = 1 / (rows of the
// whole (global) minibatch in the group), s1/s2 the matching global sums
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dh, int lddh, const float* __restrict__ h, int ldh,
    const float* __restrict__ a, int lda, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ s1, const float* __restrict__ s2,
    int R, int groups, int N, int relu, float inv_count, float* __restrict__ da, int ldda) {
  const size_t total = (size_t)R * groups * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % N);
    const size_t row = i / N;
    const int g = (int)(row / R);
    const float mu = mean[(size_t)g * N + c];
    const float istd = rsqrtf(var[(size_t)g * N + c] + BN_EPSILON);
    float d = dh[row * lddh + c];
    if (relu && !(h[row * ldh + c] > 0.f)) d = 0.f;
    const float xh = (a[row * lda + c] - mu) * istd;
    da[row * ldda + c] = bn_input_gradient(d, xh, s1[(size_t)g * N + c], s2[(size_t)g * N + c],
                                           inv_count, istd);
  }
}

int bn_bwd_apply(hipStream_t stream, const float* dh, int lddh, const float* h, int ldh,
                 const float* a, int lda, const float* mean, const float* var, const float* s1,
                 const float* s2, int rows_per_group, int groups, int N, int relu, float inv_count,
                 float* da, int ldda) {
  SCVAE_ARG(dh && h && a && mean && var && s1 && s2 && da);
  const size_t total = (size_t)rows_per_group * groups * N;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, stream, dh, lddh, h, ldh, a,
                     lda, mean, var, s1, s2, rows_per_group, groups, N, relu, inv_count, da, ldda);
  SCVAE_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return 0;
}

// ---- one-launch batch norm for tensors of a few thousand rows (column-parallel) ----
// The statistics of batch norm are per column, so a workgroup that owns FOUR columns and ALL rows
// needs nobody else: it loads its [rows, 4] slab once (one float4 per row and thread, kept in
// registers), reduces it twice (mean, centred second moment), and writes the normalised
// activations -- statistics + finalize + apply in one launch instead of three (forward), and the
// backward sums + dbeta + moving-average update + gradient in one instead of three.  N / 4
// workgroups of 1024 threads, up to BNC_RPT rows per thread.  Fixed reduction order
// (wave shuffles, then the 16 wave sums in sequence): deterministic.  Used for small minibatches
// (bn_cols_pays) when there is one group, no data-parallel exchange between the statistics and
// their use, and the layout is 16-byte friendly; everything else takes the chunked kernels above.
constexpr int BNC_RPT = 8;
constexpr int BNC_THREADS = 1024;

__device__ __forceinline__ float4 bnc_block_sum(float4 v, float4* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_xor(v.x, off, WAVE); v.y += __shfl_xor(v.y, off, WAVE);
    v.z += __shfl_xor(v.z, off, WAVE); v.w += __shfl_xor(v.w, off, WAVE);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float4 s = red[0];
#pragma unroll
  for (int i = 1; i < BNC_THREADS / 64; ++i) {
    s.x += red[i].x; s.y += red[i].y; s.z += red[i].z; s.w += red[i].w;
  }
  return s;
}

__global__ __launch_bounds__(BNC_THREADS) void bn_fwd_cols_kernel(
    const float* __restrict__ a, int lda, int R, int N, const float* __restrict__ beta, int relu,
    float* __restrict__ h, int ldh, float* __restrict__ mean, float* __restrict__ var) {
  __shared__ float4 red[BNC_THREADS / 64];
  const int c = blockIdx.x * 4;
  float4 v[BNC_RPT];
  float4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < BNC_RPT; ++i) {
    const int r = threadIdx.x + BNC_THREADS * i;
    v[i] = r < R ? *reinterpret_cast<const float4*>(a + (size_t)r * lda + c)
                 : float4{0.f, 0.f, 0.f, 0.f};
    s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w;
  }
  s = bnc_block_sum(s, red);
  const float inv = 1.f / (float)R;
  const float4 mu = {s.x * inv, s.y * inv, s.z * inv, s.w * inv};
  float4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < BNC_RPT; ++i) {
    if (threadIdx.x + BNC_THREADS * i < R) {
      const float dx = v[i].x - mu.x, dy = v[i].y - mu.y, dz = v[i].z - mu.z, dw = v[i].w - mu.w;
      q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y);
      q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
    }
  }
  q = bnc_block_sum(q, red);
  const float4 vr = {q.x * inv, q.y * inv, q.z * inv, q.w * inv};
  if (threadIdx.x == 0) {
    *reinterpret_cast<float4*>(mean + c) = mu;
    *reinterpret_cast<float4*>(var + c) = vr;
  }
  const float4 is = {rsqrtf(vr.x + BN_EPSILON), rsqrtf(vr.y + BN_EPSILON),
                     rsqrtf(vr.z + BN_EPSILON), rsqrtf(vr.w + BN_EPSILON)};
  const float4 be = *reinterpret_cast<const float4*>(beta + c);
#pragma unroll
  for (int i = 0; i < BNC_RPT; ++i) {
    const int r = threadIdx.x + BNC_THREADS * i;
    if (r < R) {
      float4 o = {(v[i].x - mu.x) * is.x + be.x, (v[i].y - mu.y) * is.y + be.y,
                  (v[i].z - mu.z) * is.z + be.z, (v[i].w - mu.w) * is.w + be.w};
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      *reinterpret_cast<float4*>(h + (size_t)r * ldh + c) = o;
    }
  }
}

static bool bnc_aligned(const void* p, int ld) { return ((uintptr_t)p & 15) == 0 && (ld & 3) == 0; }

bool bn_cols_supported(int rows, int N) {
  return rows >= 1 && rows <= BNC_RPT * BNC_THREADS && N >= 4 && (N & 3) == 0;
}
// ... and where it pays: the 16-byte-per-row column slabs stream badly, so beyond ~1000 rows the
// chunked kernels win (4096 rows: 13 vs 14 us forward but 32 vs 16 us backward); below, the step
// is a chain of dependent latencies and one launch replaces three
bool bn_cols_pays(int rows) { return rows <= 1024; }

int bn_fwd_cols(hipStream_t stream, const float* a, int lda, int rows, int N, const float* beta,
                int relu, float* h, int ldh, float* mean, float* var) {
  SCVAE_ARG(a && beta && h && mean && var && bn_cols_supported(rows, N));
  SCVAE_ARG(bnc_aligned(a, lda) && bnc_aligned(h, ldh) && bnc_aligned(beta, 4) &&
            bnc_aligned(mean, 4) && bnc_aligned(var, 4));
  hipLaunchKernelGGL(bn_fwd_cols_kernel, dim3(N / 4), dim3(BNC_THREADS), 0, stream, a, lda, rows, N,
                     beta, relu, h, ldh, mean, var);
  SCVAE_LAUNCH_CHECK("bn_fwd_cols_kernel");
  return 0;
}

// backward: g = dh * (h > 0) [relu]; s1 = sum g; s2 = sum g xhat; dbeta = s1;
// da = istd * (g - s1 / R - xhat * s2 / R); moving statistics <- batch statistics (UPDATE_OPS)
__global__ __launch_bounds__(BNC_THREADS) void bn_bwd_cols_kernel(
    const float* __restrict__ dh, int lddh, const float* __restrict__ h, int ldh,
    const float* __restrict__ a, int lda, const float* __restrict__ mean,
    const float* __restrict__ var, int R, int N, int relu, float* __restrict__ da, int ldda,
    float* __restrict__ s1_out, float* __restrict__ s2_out, float* __restrict__ dbeta,
    float* __restrict__ moving_mean, float* __restrict__ moving_var, float bessel) {
  __shared__ float4 red[BNC_THREADS / 64];
  const int c = blockIdx.x * 4;
  const float4 mu = *reinterpret_cast<const float4*>(mean + c);
  const float4 vr = *reinterpret_cast<const float4*>(var + c);
  const float4 is = {rsqrtf(vr.x + BN_EPSILON), rsqrtf(vr.y + BN_EPSILON),
                     rsqrtf(vr.z + BN_EPSILON), rsqrtf(vr.w + BN_EPSILON)};
  float4 g[BNC_RPT], xh[BNC_RPT];
  float4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < BNC_RPT; ++i) {
    const int r = threadIdx.x + BNC_THREADS * i;
    g[i] = float4{0.f, 0.f, 0.f, 0.f};
    xh[i] = float4{0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      float4 d = *reinterpret_cast<const float4*>(dh + (size_t)r * lddh + c);
      if (relu) {
        const float4 hv = *reinterpret_cast<const float4*>(h + (size_t)r * ldh + c);
        if (!(hv.x > 0.f)) d.x = 0.f;
        if (!(hv.y > 0.f)) d.y = 0.f;
        if (!(hv.z > 0.f)) d.z = 0.f;
        if (!(hv.w > 0.f)) d.w = 0.f;
      }
      const float4 av = *reinterpret_cast<const float4*>(a + (size_t)r * lda + c);
      g[i] = d;
      xh[i] = float4{(av.x - mu.x) * is.x, (av.y - mu.y) * is.y, (av.z - mu.z) * is.z,
                     (av.w - mu.w) * is.w};
      t1.x += d.x; t1.y += d.y; t1.z += d.z; t1.w += d.w;
      t2.x = fmaf(d.x, xh[i].x, t2.x); t2.y = fmaf(d.y, xh[i].y, t2.y);
      t2.z = fmaf(d.z, xh[i].z, t2.z); t2.w = fmaf(d.w, xh[i].w, t2.w);
    }
  }
  t1 = bnc_block_sum(t1, red);
  t2 = bnc_block_sum(t2, red);
  const float inv = 1.f / (float)R;
  if (threadIdx.x == 0) {
    *reinterpret_cast<float4*>(s1_out + c) = t1;
    *reinterpret_cast<float4*>(s2_out + c) = t2;
    if (dbeta != nullptr) *reinterpret_cast<float4*>(dbeta + c) = t1;
    if (moving_mean != nullptr) {
      float4 mm = *reinterpret_cast<float4*>(moving_mean + c);
      float4 mv = *reinterpret_cast<float4*>(moving_var + c);
      mm.x -= (mm.x - mu.x) * BN_UPDATE_RATE; mm.y -= (mm.y - mu.y) * BN_UPDATE_RATE;
      mm.z -= (mm.z - mu.z) * BN_UPDATE_RATE; mm.w -= (mm.w - mu.w) * BN_UPDATE_RATE;
      mv.x -= (mv.x - vr.x * bessel) * BN_UPDATE_RATE; mv.y -= (mv.y - vr.y * bessel) * BN_UPDATE_RATE;
      mv.z -= (mv.z - vr.z * bessel) * BN_UPDATE_RATE; mv.w -= (mv.w - vr.w * bessel) * BN_UPDATE_RATE;
      *reinterpret_cast<float4*>(moving_mean + c) = mm;
      *reinterpret_cast<float4*>(moving_var + c) = mv;
    }
  }
  const float4 m1 = {t1.x * inv, t1.y * inv, t1.z * inv, t1.w * inv};
  const float4 m2 = {t2.x * inv, t2.y * inv, t2.z * inv, t2.w * inv};
#pragma unroll
  for (int i = 0; i < BNC_RPT; ++i) {
    const int r = threadIdx.x + BNC_THREADS * i;
    if (r < R) {
      const float4 o = {is.x * (g[i].x - m1.x - xh[i].x * m2.x), is.y * (g[i].y - m1.y - xh[i].y * m2.y),
                        is.z * (g[i].z - m1.z - xh[i].z * m2.z), is.w * (g[i].w - m1.w - xh[i].w * m2.w)};
      *reinterpret_cast<float4*>(da + (size_t)r * ldda + c) = o;
    }
  }
}

int bn_bwd_cols(hipStream_t stream, const float* dh, int lddh, const float* h, int ldh,
                const float* a, int lda, const float* mean, const float* var, int rows, int N,
                int relu, float* da, int ldda, float* s1, float* s2, float* dbeta,
                float* moving_mean, float* moving_var) {
  SCVAE_ARG(dh && h && a && mean && var && da && s1 && s2 && bn_cols_supported(rows, N));
  SCVAE_ARG((moving_mean == nullptr) == (moving_var == nullptr));
  SCVAE_ARG(bnc_aligned(dh, lddh) && bnc_aligned(h, ldh) && bnc_aligned(a, lda) &&
            bnc_aligned(da, ldda) && bnc_aligned(mean, 4) && bnc_aligned(var, 4) &&
            bnc_aligned(s1, 4) && bnc_aligned(s2, 4) && (!dbeta || bnc_aligned(dbeta, 4)) &&
            (!moving_mean || (bnc_aligned(moving_mean, 4) && bnc_aligned(moving_var, 4))));
  const float bessel = (float)rows / (float)(rows > 1 ? rows - 1 : 1);
  hipLaunchKernelGGL(bn_bwd_cols_kernel, dim3(N / 4), dim3(BNC_THREADS), 0, stream, dh, lddh, h, ldh,
                     a, lda, mean, var, rows, N, relu, da, ldda, s1, s2, dbeta, moving_mean,
                     moving_var, bessel);
  SCVAE_LAUNCH_CHECK("bn_bwd_cols_kernel");
  return 0;
}

bool bn_cols_layout_ok(const void* p, int ld) { return bnc_aligned(p, ld); }

// Chan et al. merge of per-rank (count, mean, biased var): gathered = [ranks][mean(n)|var(n)]
__global__ void bn_merge_kernel(const float* __restrict__ gathered,
                                const int64_t* __restrict__ counts, int ranks, int n,
                                float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double total = 0.0, mean = 0.0;
  for (int r = 0; r < ranks; ++r) {
    total += (double)counts[r];
    mean += (double)counts[r] * (double)gathered[(size_t)r * 2 * n + c];
  }
  mean /= total;
  double m2 = 0.0;
  for (int r = 0; r < ranks; ++r) {
    const double d = (double)gathered[(size_t)r * 2 * n + c] - mean;
    m2 += (double)counts[r] * ((double)gathered[(size_t)r * 2 * n + n + c] + d * d);
  }
  out[c] = (float)mean;
  out[n + c] = (float)(m2 / total);
}

int bn_merge(hipStream_t stream, const float* gathered, const int64_t* counts, int ranks, int n,
             float* out) {
  SCVAE_ARG(gathered && counts && out && ranks > 0 && n > 0);
  hipLaunchKernelGGL(bn_merge_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, gathered, counts,
                     ranks, n, out);
  SCVAE_LAUNCH_CHECK("bn_merge_kernel");
  return 0;
}

__global__ void relu_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                float* __restrict__ da, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    da[i] = h[i] > 0.f ? dh[i] : 0.f;
}
int relu_bwd(hipStream_t stream, const float* dh, const float* h, float* da, size_t n) {
  SCVAE_ARG(dh && h && da);
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dh, h, da, n);
  SCVAE_LAUNCH_CHECK("relu_bwd_kernel");
  return 0;
}

// out[c] (+)= scale * sum_r a[r,c].  Small row counts: one workgroup per 64 columns.
// Large: grid (ceil(N/64), chunks) partial sums into `partial`, then a fixed-order finalize.
__global__ __launch_bounds__(1024) void col_sum_kernel(const float* __restrict__ a, int lda,
                                                       int rows, int N, int chunk,
                                                       float* __restrict__ out, float scale,
                                                       int accumulate, int to_partial) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + cl;
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 16) s += a[(size_t)r * lda + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cl];
    if (to_partial) {
      out[(size_t)blockIdx.y * N + c] = t;
    } else {
      t *= scale;
      out[c] = accumulate ? out[c] + t : t;
    }
  }
}

__global__ void col_sum_finalize_kernel(const float* __restrict__ partial, int N, int chunks,
                                        float* __restrict__ out, float scale, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float t = 0.f;
  for (int z = 0; z < chunks; ++z) t += partial[(size_t)z * N + c];
  t *= scale;
  out[c] = accumulate ? out[c] + t : t;
}

size_t col_sum_partial_floats(int N) { return (size_t)COLSUM_MAX_CHUNKS * N; }

int col_sum(hipStream_t stream, const float* a, int lda, int rows, int N, float* out, float scale,
            int accumulate, float* partial) {
  SCVAE_ARG(a && out && N > 0 && rows >= 0);
  int chunks = 1;
  if (partial != nullptr && rows >= 512) {
    chunks = (rows + 255) / 256;
    if (chunks > COLSUM_MAX_CHUNKS) chunks = COLSUM_MAX_CHUNKS;
  }
  const int chunk = (rows + chunks - 1) / chunks;
  if (chunks > 1) chunks = (rows + chunk - 1) / chunk;
  if (chunks <= 1) {
    hipLaunchKernelGGL(col_sum_kernel, dim3((N + 63) / 64, 1), dim3(1024), 0, stream, a, lda, rows,
                       N, rows > 0 ? rows : 1, out, scale, accumulate, 0);
    SCVAE_LAUNCH_CHECK("col_sum_kernel");
    return 0;
  }
  hipLaunchKernelGGL(col_sum_kernel, dim3((N + 63) / 64, chunks), dim3(1024), 0, stream, a, lda,
                     rows, N, chunk, partial, 1.f, 0, 1);
  SCVAE_LAUNCH_CHECK("col_sum_kernel");
  hipLaunchKernelGGL(col_sum_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, partial,
                     N, chunks, out, scale, accumulate);
  SCVAE_LAUNCH_CHECK("col_sum_finalize_kernel");
  return 0;
}

// ---- decoder input [z | extra] (batch one-hot / count sum appended to z, va:2407-2441) ----
__global__ __launch_bounds__(256) void concat_extra_kernel(const float* __restrict__ z, int L,
                                                           const float* __restrict__ extra, int E,
                                                           size_t rows, size_t cells,
                                                           float* __restrict__ out) {
  const int W = L + E;
  const size_t total = rows * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / W;
    const int c = (int)(i - r * W);
    out[i] = c < L ? z[r * L + c] : extra[(r % cells) * E + (c - L)];
  }
}
__global__ __launch_bounds__(256) void slice_cols_kernel(const float* __restrict__ in, int ld,
                                                         int L, size_t rows,
                                                         float* __restrict__ out) {
  const size_t total = rows * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / L;
    out[i] = in[r * ld + (i - r * L)];
  }
}
int concat_extra(hipStream_t stream, const float* z, int L, const float* extra, int E, size_t rows,
                 size_t cells, float* out) {
  SCVAE_ARG(z && extra && out && L > 0 && E > 0 && cells > 0);
  if (rows == 0) return 0;
  size_t blocks = (rows * (L + E) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(concat_extra_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, z, L, extra,
                     E, rows, cells, out);
  SCVAE_LAUNCH_CHECK("concat_extra_kernel");
  return 0;
}
int slice_cols(hipStream_t stream, const float* in, int ld, int L, size_t rows, float* out) {
  SCVAE_ARG(in && out && L > 0 && ld >= L);
  if (rows == 0) return 0;
  size_t blocks = (rows * L + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slice_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld, L,
                     rows, out);
  SCVAE_LAUNCH_CHECK("slice_cols_kernel");
  return 0;
}

// ================================ optimiser ================================

// g <- clip(g * grad_scale, -1, 1); TF Adam: m,v update; theta -= lr_t * m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ theta,
                                                        float* __restrict__ grad,
                                                        float* __restrict__ m,
                                                        float* __restrict__ v, size_t n,
                                                        float grad_scale, float lr_t, float beta1,
                                                        float beta2, float epsilon) {
  const size_t n4 = n / 4;
  float4* th4 = reinterpret_cast<float4*>(theta);
  const float4* g4 = reinterpret_cast<const float4*>(grad);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float& th, float g, float& mm, float& vv) {
    g = fminf(fmaxf(g * grad_scale, -1.f), 1.f);
    mm = beta1 * mm + (1.f - beta1) * g;
    vv = beta2 * vv + (1.f - beta2) * g * g;
    th -= lr_t * mm / (sqrtf(vv) + epsilon);
  };
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 th = th4[i], g = g4[i], mm = m4[i], vv = v4[i];
    upd(th.x, g.x, mm.x, vv.x);
    upd(th.y, g.y, mm.y, vv.y);
    upd(th.z, g.z, mm.z, vv.z);
    upd(th.w, g.w, mm.w, vv.w);
    th4[i] = th; m4[i] = mm; v4[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    upd(theta[i], grad[i], m[i], v[i]);
  }
}

int adam_clip_step(hipStream_t stream, float* theta, float* grad, float* m, float* v, size_t n,
                   float grad_scale, float lr_t, float beta1, float beta2, float epsilon) {
  SCVAE_ARG(theta && grad && m && v);
  SCVAE_ARG(((uintptr_t)theta | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) % 16 == 0);
  if (n == 0) return 0;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, theta, grad,
                     m, v, n, grad_scale, lr_t, beta1, beta2, epsilon);
  SCVAE_LAUNCH_CHECK("adam_clip_kernel");
  return 0;
}

// ================================ Philox ===================================

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// thread per (row, group of 4 columns): counter = (row_lo, row_hi, col_group, stream_id),
// key = seed.  u = ((x >> 8) + 0.5) * 2^-24; Box-Muller on (u0,u1) and (u2,u3).
// Rows may come in blocks (the stacked passes of a sharded minibatch): row r of the buffer is
// row `(r / block_rows) * block_stride + row_offset + r % block_rows` of the noise field.
// (the arguments of one noise fill; also handed to the minibatch kernels, whose trailing
//  workgroups then draw the noise of the same step: one launch for fetch + noise)
struct NoiseJob {
  float* out = nullptr;     // nullptr: none
  int64_t rows = 0;
  int cols = 0;
  int64_t row_offset = 0, block_rows = 1, block_stride = 0;
  uint32_t seed_lo = 0, seed_hi = 0, stream_id = 0;
};
__device__ __forceinline__ void philox_fill(const NoiseJob& q, int64_t first, int64_t stride) {
  float* __restrict__ out = q.out;
  const int cols = q.cols;
  const int64_t rows = q.rows, row_offset = q.row_offset, block_rows = q.block_rows,
                block_stride = q.block_stride;
  const uint32_t seed_lo = q.seed_lo, seed_hi = q.seed_hi, stream_id = q.stream_id;
  const int groups = (cols + 3) / 4;
  const int64_t total = rows * groups;
  for (int64_t i = first; i < total; i += stride) {
    const int64_t row = i / groups;
    const int cg = (int)(i % groups);
    const uint64_t grow =
        (uint64_t)((row / block_rows) * block_stride + row_offset + row % block_rows);
    uint32_t c[4] = {(uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)cg, stream_id};
    uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    float u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = ((float)(c[j] >> 8) + 0.5f) * 5.9604644775390625e-8f;
    float n[4];
    const float r0 = sqrtf(-2.f * logf(u[0])), r1 = sqrtf(-2.f * logf(u[2]));
    const float th0 = 6.283185307179586f * u[1], th1 = 6.283185307179586f * u[3];
    n[0] = r0 * cosf(th0); n[1] = r0 * sinf(th0);
    n[2] = r1 * cosf(th1); n[3] = r1 * sinf(th1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = cg * 4 + j;
      if (col < cols) out[row * cols + col] = n[j];
    }
  }
}
__global__ __launch_bounds__(256) void philox_normal_kernel(NoiseJob q) {
  philox_fill(q, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}
// workgroups of 1024 threads a fetch kernel appends for the noise of `q` (0: none)
static int noise_blocks_1024(const NoiseJob& q) {
  if (!q.out || q.rows == 0) return 0;
  const int64_t total = q.rows * ((q.cols + 3) / 4);
  int64_t blocks = (total + 1023) / 1024;
  return (int)(blocks > 1024 ? 1024 : blocks);
}
static NoiseJob noise_job(float* out, int64_t rows, int cols, int64_t row_offset, uint64_t seed,
                          uint64_t stream_id, int64_t block_rows, int64_t block_stride) {
  NoiseJob q;
  if (block_rows <= 0) { block_rows = rows > 0 ? rows : 1; block_stride = 0; }   // one block
  q.out = out; q.rows = rows; q.cols = cols; q.row_offset = row_offset;
  q.block_rows = block_rows; q.block_stride = block_stride;
  q.seed_lo = (uint32_t)seed;
  // the high half of the 64-bit stream id goes into the key (callers use the high bits as
  // domain separators: evaluation passes, model.sample())
  q.seed_hi = (uint32_t)(seed >> 32) ^ (uint32_t)(stream_id >> 32);
  q.stream_id = (uint32_t)stream_id;
  return q;
}

int philox_normal(hipStream_t stream, float* out, int64_t rows, int cols, int64_t row_offset,
                  uint64_t seed, uint64_t stream_id, int64_t block_rows, int64_t block_stride) {
  SCVAE_ARG(out && rows >= 0 && cols > 0);
  if (rows == 0) return 0;
  SCVAE_ARG(block_rows <= 0 || block_stride >= 0);
  const NoiseJob q = noise_job(out, rows, cols, row_offset, seed, stream_id, block_rows,
                               block_stride);
  const int64_t total = rows * ((cols + 3) / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, q);
  SCVAE_LAUNCH_CHECK("philox_normal_kernel");
  return 0;
}

// ============================== CSR minibatch ==============================

// One workgroup per gathered row writes the whole dense row in one pass: float4 zero fill of
// the row (16-byte aligned part), a workgroup barrier, then the row's nonzeros.  Both phases of
// a row are issued by the same workgroup, whose stores to one address stay ordered across
// __syncthreads(), so no separate memset of the [B, F] buffer is needed.
__global__ __launch_bounds__(256) void csr_densify_kernel(const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices,
                                                          const float* __restrict__ values,
                                                          const int64_t* __restrict__ rows, int F,
                                                          float* __restrict__ out, int ldo,
                                                          const float* __restrict__ row_values,
                                                          float* __restrict__ row_values_out) {
  const int b = blockIdx.x;
  const int64_t r = rows[b];
  if (row_values_out != nullptr && threadIdx.x == 0) row_values_out[b] = row_values[r];
  const int64_t lo = indptr[r], hi = indptr[r + 1];
  float* orow = out + (size_t)b * ldo;
  // head elements up to the first 16-byte boundary, float4 body, scalar tail
  const uintptr_t addr = reinterpret_cast<uintptr_t>(orow);
  int head = (int)(((16 - (addr & 15)) & 15) / 4);
  if (head > F) head = F;
  const int n4 = (F - head) / 4;
  float4* o4 = reinterpret_cast<float4*>(orow + head);
  for (int i = threadIdx.x; i < n4; i += 256) o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((int)threadIdx.x < head) orow[threadIdx.x] = 0.f;
  const int tail0 = head + 4 * n4;
  if (tail0 + (int)threadIdx.x < F) orow[tail0 + threadIdx.x] = 0.f;
  __threadfence_block();
  __syncthreads();
  for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
    const int32_t c = indices[j];
    if (c >= 0 && c < F) orow[c] = values[j];
  }
}

// Rows that fit into LDS (F <= 38 000 genes): the dense row is assembled in LDS (zero fill,
// scatter of the nonzeros) and streamed out once with 16-byte stores -- HBM sees one coalesced
// write per element instead of a fill plus scattered 4-byte stores.
// NOISE (compile time): trailing workgroups draw the noise of the same step (small minibatches:
// one launch fewer).  The plain instantiation carries none of that code: with it the 4096-cell
// fetch ran 92 instead of 66 us inside the training step.
template <bool NOISE>
__global__ __launch_bounds__(1024) void csr_densify_lds_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ values, const int64_t* __restrict__ rows, int F,
    float* __restrict__ out, int ldo, const float* __restrict__ row_values,
    float* __restrict__ row_values_out, int B, NoiseJob noise) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  if (NOISE && (int)blockIdx.x >= B) {       // (the trailing workgroups: this step's noise)
    philox_fill(noise, (int64_t)(blockIdx.x - B) * 1024 + threadIdx.x,
                (int64_t)(gridDim.x - B) * 1024);
    return;
  }
  const int b = blockIdx.x;
  const int64_t r = rows[b];
  // (a per-row value of the matrix gathered on the way: the lgamma term of the likelihoods)
  if (row_values_out != nullptr && threadIdx.x == 0) row_values_out[b] = row_values[r];
  const int64_t lo = indptr[r], hi = indptr[r + 1];
  const int n4 = (F + 3) / 4;
  float4* row4 = reinterpret_cast<float4*>(row);
  for (int i = threadIdx.x; i < n4; i += 1024) row4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  for (int64_t j = lo + threadIdx.x; j < hi; j += 1024) {
    const int32_t c = indices[j];
    if (c >= 0 && c < F) row[c] = values[j];
  }
  __syncthreads();
  float* orow = out + (size_t)b * ldo;
  // 16-byte non-temporal stores whatever the row pitch (global accesses need only 4-byte
  // alignment): the dense batch is written once and streamed by its readers
  {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int full = F / 4;
    for (int i = threadIdx.x; i < full; i += 1024) {
      const float4 v = row4[i];
      const f32x4u q = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(q, reinterpret_cast<f32x4u*>(orow + 4 * i));
    }
    for (int i = 4 * full + threadIdx.x; i < F; i += 1024) orow[i] = row[i];
  }
}

int csr_densify(hipStream_t stream, const int64_t* indptr, const int32_t* indices,
                const float* values, const int64_t* rows, int B, int F, float* out, int ldo,
                const float* row_values, float* row_values_out, const NoiseRequest* nr) {
  SCVAE_ARG(indptr && indices && values && rows && out && F > 0 && ldo >= F);
  SCVAE_ARG(row_values_out == nullptr || row_values != nullptr);
  SCVAE_ARG(!nr || !nr->out || (nr->rows >= 0 && nr->cols > 0));
  if (B == 0) return nr && nr->out ? philox_normal(stream, nr->out, nr->rows, nr->cols, nr->row_offset, nr->seed, nr->stream_id, nr->block_rows, nr->block_stride) : 0;
  const NoiseJob noise = (nr && nr->out)
                             ? noise_job(nr->out, nr->rows, nr->cols, nr->row_offset, nr->seed,
                                         nr->stream_id, nr->block_rows, nr->block_stride)
                             : NoiseJob();
  const size_t lds = ((size_t)F + 3) / 4 * 16;
  if (lds <= 152 * 1024) {
    auto kfn = noise.out ? csr_densify_lds_kernel<true> : csr_densify_lds_kernel<false>;
    SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)lds));
    hipLaunchKernelGGL(kfn, dim3(B + noise_blocks_1024(noise)), dim3(1024), lds, stream, indptr,
                       indices, values, rows, F, out, ldo, row_values, row_values_out, B, noise);
    SCVAE_LAUNCH_CHECK("csr_densify_lds_kernel");
    return 0;
  }
  hipLaunchKernelGGL(csr_densify_kernel, dim3(B), dim3(256), 0, stream, indptr, indices, values,
                     rows, F, out, ldo, row_values, row_values_out);
  SCVAE_LAUNCH_CHECK("csr_densify_kernel");
  if (noise.out)
    return philox_normal(stream, nr->out, nr->rows, nr->cols, nr->row_offset, nr->seed,
                         nr->stream_id, nr->block_rows, nr->block_stride);
  return 0;
}

// The minibatch as uint16 counts for the kernels that stream it (count_gemm.hip, the fused
// likelihood heads): half the bytes of the fp32 batch.  Precondition: integer counts below
// 65 536 (DeviceCSR.integer_counts).  Row pitch ldo: a multiple of 8 (16-byte rows), the pad
// columns F .. ldo - 1 are zeroed.  The row is assembled in LDS and streamed out once.
template <bool NOISE>
__global__ __launch_bounds__(1024) void csr_densify_u16_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ values, const int64_t* __restrict__ rows, int F,
    uint16_t* __restrict__ out, int ldo, const float* __restrict__ row_values,
    float* __restrict__ row_values_out, int B, NoiseJob noise) {
  extern __shared__ __attribute__((aligned(16))) uint16_t row16[];
  if (NOISE && (int)blockIdx.x >= B) {       // (the trailing workgroups: this step's noise)
    philox_fill(noise, (int64_t)(blockIdx.x - B) * 1024 + threadIdx.x,
                (int64_t)(gridDim.x - B) * 1024);
    return;
  }
  const int b = blockIdx.x;
  const int64_t r = rows[b];
  if (row_values_out != nullptr && threadIdx.x == 0) row_values_out[b] = row_values[r];
  const int64_t lo = indptr[r], hi = indptr[r + 1];
  const int n8 = ldo / 8;
  uint4* rowv = reinterpret_cast<uint4*>(row16);
  for (int i = threadIdx.x; i < n8; i += 1024) rowv[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  for (int64_t j = lo + threadIdx.x; j < hi; j += 1024) {
    const int32_t c = indices[j];
    if (c >= 0 && c < F) row16[c] = (uint16_t)(int)values[j];
  }
  __syncthreads();
  typedef unsigned u32x4nt __attribute__((ext_vector_type(4)));
  u32x4nt* orow = reinterpret_cast<u32x4nt*>(out + (size_t)b * ldo);
  for (int i = threadIdx.x; i < n8; i += 1024) {
    const uint4 v = rowv[i];
    const u32x4nt q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, orow + i);
  }
}

bool csr_densify_u16_supported(int F, int ldo) {
  return F > 0 && ldo >= F && (ldo & 7) == 0 && (size_t)ldo * 2 <= 152 * 1024;
}

int csr_densify_u16(hipStream_t stream, const int64_t* indptr, const int32_t* indices,
                    const float* values, const int64_t* rows, int B, int F, uint16_t* out,
                    int ldo, const float* row_values, float* row_values_out,
                    const NoiseRequest* nr) {
  SCVAE_ARG(indptr && indices && values && rows && out);
  SCVAE_ARG(row_values_out == nullptr || row_values != nullptr);
  SCVAE_ARG(csr_densify_u16_supported(F, ldo) && ((uintptr_t)out & 15) == 0);
  SCVAE_ARG(!nr || !nr->out || (nr->rows >= 0 && nr->cols > 0));
  if (B == 0) return nr && nr->out ? philox_normal(stream, nr->out, nr->rows, nr->cols, nr->row_offset, nr->seed, nr->stream_id, nr->block_rows, nr->block_stride) : 0;
  const NoiseJob noise = (nr && nr->out)
                             ? noise_job(nr->out, nr->rows, nr->cols, nr->row_offset, nr->seed,
                                         nr->stream_id, nr->block_rows, nr->block_stride)
                             : NoiseJob();
  const size_t lds = (size_t)ldo * 2;
  auto kfn = noise.out ? csr_densify_u16_kernel<true> : csr_densify_u16_kernel<false>;
  SCVAE_HIP(max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)lds));
  hipLaunchKernelGGL(kfn, dim3(B + noise_blocks_1024(noise)), dim3(1024), lds, stream, indptr,
                     indices, values, rows, F, out, ldo, row_values, row_values_out, B, noise);
  SCVAE_LAUNCH_CHECK("csr_densify_u16_kernel");
  return 0;
}

// out[r] = sum_j lgamma(1 + values[j]) over the nonzeros of row r (one wave per row)
__global__ __launch_bounds__(256) void csr_row_lgamma1p_kernel(const int64_t* __restrict__ indptr,
                                                               const float* __restrict__ values,
                                                               int64_t n_rows,
                                                               float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int64_t j = indptr[r] + lane; j < indptr[r + 1]; j += 64) s += lgamma1p(values[j]);
  s = wave_sum(s);
  if (lane == 0) out[r] = s;
}

int csr_row_lgamma1p(hipStream_t stream, const int64_t* indptr, const float* values,
                     int64_t n_rows, float* out) {
  SCVAE_ARG(indptr && values && out);
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(csr_row_lgamma1p_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0,
                     stream, indptr, values, n_rows, out);
  SCVAE_LAUNCH_CHECK("csr_row_lgamma1p_kernel");
  return 0;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ rows,
                                   int B, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = src[rows[b]];
}
int gather_rows_f32(hipStream_t stream, const float* src, const int64_t* rows, int B, float* out) {
  SCVAE_ARG(src && rows && out);
  if (B == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, src, rows, B,
                     out);
  SCVAE_LAUNCH_CHECK("gather_rows_kernel");
  return 0;
}


// ================================ dropout ==================================

// tf.contrib.layers.dropout(inputs, keep_prob, is_training) -> tf.nn.dropout (mu:45-50):
// out = in * m / keep, m ~ Bernoulli(keep) per element.  The mask is a pure function of
// (seed, site, row, column): Philox4x32-10 with key = seed and counter = (row_lo, row_hi,
// column / 4, 0x80000000 | site), element kept iff its uniform draw is below keep.  The backward
// pass calls the same kernel on the gradient (optionally accumulating), so no mask is stored.
__global__ __launch_bounds__(256) void dropout_apply_kernel(
    const float* __restrict__ in, int ld_in, float* __restrict__ out, int ld_out, int64_t rows,
    int cols, float keep, float inv_keep, uint32_t seed_lo, uint32_t seed_hi, uint32_t site,
    int accumulate, RowMap map) {
  const int groups = (cols + 3) / 4;
  const int64_t total = rows * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / groups;
    const int cg = (int)(i % groups);
    // the mask belongs to the row of the *global* minibatch (data parallel: rows of a rank's
    // shard are block `row / cells` of the stacked passes, cell `offset + row % cells`)
    const uint64_t grow = map.cells > 0
        ? (uint64_t)((row / map.cells) * map.global_cells + map.offset + row % map.cells)
        : (uint64_t)row;
    uint32_t c[4] = {(uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)cg, 0x80000000u | site};
    uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = cg * 4 + j;
      if (col >= cols) break;
      const float u = ((float)(c[j] >> 8) + 0.5f) * 5.9604644775390625e-8f;
      const float v = u < keep ? in[row * ld_in + col] * inv_keep : 0.f;
      float* o = out + row * ld_out + col;
      *o = accumulate ? *o + v : v;
    }
  }
}

int dropout_apply(hipStream_t stream, const float* in, int ld_in, float* out, int ld_out,
                  int64_t rows, int cols, float keep, uint64_t seed, uint32_t site,
                  int accumulate, RowMap map) {
  SCVAE_ARG(in && out && rows >= 0 && cols > 0 && ld_in >= cols && ld_out >= cols);
  SCVAE_ARG(keep > 0.f && keep <= 1.f && site < 0x80000000u);
  if (rows == 0) return 0;
  const int64_t total = rows * ((cols + 3) / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld_in,
                     out, ld_out, rows, cols, keep, 1.f / keep, (uint32_t)seed,
                     (uint32_t)(seed >> 32), site, accumulate, map);
  SCVAE_LAUNCH_CHECK("dropout_apply_kernel");
  return 0;
}

// The same mask as bits (the fused decoder-head kernel applies a head's mask to its part of dd):
// thread = (row, word of 32 columns) = eight Philox blocks of four columns.
__global__ __launch_bounds__(256) void dropout_mask_words_kernel(
    uint32_t* __restrict__ words, int rows, int rows_pad, int cols, float keep, uint32_t seed_lo,
    uint32_t seed_hi, uint32_t site, RowMap map) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_pad * 4) return;
  const int row = i >> 2, wd = i & 3;
  uint32_t bits = 0u;
  if (row < rows) {
    const uint64_t grow = map.cells > 0
        ? (uint64_t)((row / map.cells) * map.global_cells + map.offset + row % map.cells)
        : (uint64_t)row;
    for (int g8 = 0; g8 < 8; ++g8) {
      const int cg = wd * 8 + g8;
      if (cg * 4 >= cols) break;
      uint32_t c[4] = {(uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)cg, 0x80000000u | site};
      uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
      for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = ((float)(c[j] >> 8) + 0.5f) * 5.9604644775390625e-8f;
        if (cg * 4 + j < cols && u < keep) bits |= 1u << (g8 * 4 + j);
      }
    }
  }
  words[i] = bits;
}
int dropout_mask_words(hipStream_t stream, uint32_t* words, int rows, int rows_pad, int cols,
                       float keep, uint64_t seed, uint32_t site, RowMap map) {
  SCVAE_ARG(words && rows >= 0 && rows_pad >= rows && cols > 0 && cols <= 128);
  SCVAE_ARG(keep > 0.f && keep <= 1.f && site < 0x80000000u);
  if (rows_pad == 0) return 0;
  hipLaunchKernelGGL(dropout_mask_words_kernel, dim3((rows_pad * 4 + 255) / 256), dim3(256), 0,
                     stream, words, rows, rows_pad, cols, keep, (uint32_t)seed,
                     (uint32_t)(seed >> 32), site, map);
  SCVAE_LAUNCH_CHECK("dropout_mask_words_kernel");
  return 0;
}

// rows of a [K, N] matrix scaled by the mask element (row k, column k): dropout of a one-hot
// input, whose only non-zero is on the diagonal (the GMVAE's p(z|y=k) layers, gm:3024-3040)
__global__ void dropout_scale_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                          int K, int N, float keep, float inv_keep,
                                          uint32_t seed_lo, uint32_t seed_hi, uint32_t site) {
  const int k = blockIdx.x;
  uint32_t c[4] = {(uint32_t)k, 0u, (uint32_t)(k >> 2), 0x80000000u | site};
  uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const uint32_t bits = (k & 3) == 0 ? c[0] : (k & 3) == 1 ? c[1] : (k & 3) == 2 ? c[2] : c[3];
  const float u = ((float)(bits >> 8) + 0.5f) * 5.9604644775390625e-8f;
  const float m = u < keep ? inv_keep : 0.f;
  for (int n = threadIdx.x; n < N; n += blockDim.x) out[(size_t)k * N + n] = in[(size_t)k * N + n] * m;
}

int dropout_scale_rows(hipStream_t stream, const float* in, float* out, int K, int N, float keep,
                       uint64_t seed, uint32_t site) {
  SCVAE_ARG(in && out && K > 0 && N > 0 && keep > 0.f && keep <= 1.f);
  hipLaunchKernelGGL(dropout_scale_rows_kernel, dim3(K), dim3(64), 0, stream, in, out, K, N, keep,
                     1.f / keep, (uint32_t)seed, (uint32_t)(seed >> 32), site);
  SCVAE_LAUNCH_CHECK("dropout_scale_rows_kernel");
  return 0;
}

// out[k*B + b, :] = [x[b, :F] | one_hot(k, K)]: the input of the GMVAE's q(z|x,y=k) encoder
// (gm:2862-2866), materialised for all K passes (needed only when that input is dropped out)
__global__ __launch_bounds__(256) void tile_onehot_kernel(const float* __restrict__ x,
                                                          float* __restrict__ out, int K, int B,
                                                          int F) {
  const int row = blockIdx.x;            // k*B + b
  const int k = row / B, b = row % B;
  float* o = out + (size_t)row * (F + K);
  const float* xi = x + (size_t)b * F;
  for (int c = threadIdx.x; c < F + K; c += 256) o[c] = c < F ? xi[c] : (c - F == k ? 1.f : 0.f);
}

int tile_onehot(hipStream_t stream, const float* x, float* out, int K, int B, int F) {
  SCVAE_ARG(x && out && K > 0 && F > 0);
  if (B == 0) return 0;
  hipLaunchKernelGGL(tile_onehot_kernel, dim3(K * B), dim3(256), 0, stream, x, out, K, B, F);
  SCVAE_LAUNCH_CHECK("tile_onehot_kernel");
  return 0;
}

}  // namespace scvae
