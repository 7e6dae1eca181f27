// Per-element count likelihoods of the scVAE decoder, from head pre-activations.
//
// Replaces the TF/TFP op chain activation -> clip_by_value -> distribution
// .log_prob/.mean/.variance built at
//   scvae/models/variational_autoencoder.py:2466-2505, 2583
//   scvae/distributions/utilities.py:206-216 (poisson), 247-264 (ZIP),
//   266-281 (negative binomial), 283-305 (ZINB)
//   scvae/distributions/zero_inflated.py:180-199
// in the algebraically equal stable form (log p = log_sigmoid(a),
// log(1-p) = log_sigmoid(-a)); see oracle/likelihoods.py for the formulas.
//
// The data-only term -lgamma(1+t) is NOT included here: callers add the
// per-cell constant sum_f lgamma(1+t[b,f]) once per row (it has no gradient).
#pragma once
#include "common.hpp"

namespace scvae {

#ifdef __HIPCC__
template <int KIND>
struct LikelihoodTraits {
  static constexpr int P = likelihood_heads(KIND);
  // negative-binomial kinds carry the terms lgamma(r+t)-lgamma(r) / digamma(r+t)-digamma(r),
  // which vanish at t == 0 (the "sparse correction")
  static constexpr bool HAS_R = (KIND == LK_NB || KIND == LK_ZINB);
};

// Everything except the sparse correction:
// lp  : log p(t | theta) + lgamma(1+t) - [lgamma(r+t) - lgamma(r)]
// g[] : d lp / d pre-activation of each head (only if GRAD), without the digamma term
// r, rgate: total_count and the clip gate of its head (negative-binomial kinds), so that the
//           caller can add  lp += A(r,t),  g[P-1] += rgate * r * D(r,t)  where t > 0.
template <int KIND, bool GRAD>
__device__ __forceinline__ void lik_dense(float t, const float* a, float& lp, float* g, float& r,
                                          float& rgate) {
  if constexpr (KIND == LK_CAT2 || KIND == LK_CAT3) {
    // tfp.distributions.Categorical(logits).log_prob(min(t, k)), k + 1 = P classes
    // (categorised.py:255-259): log softmax, gradient one-hot minus softmax
    constexpr int PC = likelihood_heads(KIND);
    float m = a[0];
#pragma unroll
    for (int j = 1; j < PC; ++j) m = fmaxf(m, a[j]);
    float e[PC], se = 0.f;
#pragma unroll
    for (int j = 0; j < PC; ++j) { e[j] = __expf(a[j] - m); se += e[j]; }
    const float lse = m + fast_log(se);
    const float inv = fast_rcp(se);
    lp = 0.f;
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const bool mine = j + 1 < PC ? t == (float)j : t >= (float)j;
      lp += mine ? a[j] - lse : 0.f;
      if (GRAD) g[j] = (mine ? 1.f : 0.f) - e[j] * inv;
    }
    r = 0.f; rgate = 0.f;
  } else if constexpr (KIND == LK_BERNOULLI) {
    // tfp.distributions.Bernoulli(logits): t log sigmoid(a) + (1 - t) log sigmoid(-a)
    float ls_pos, ls_neg, sig, sig_neg;
    log_sigmoid_pair(a[0], ls_pos, ls_neg, sig, sig_neg);
    lp = t * ls_pos + (1.f - t) * ls_neg;
    if (GRAD) g[0] = t * sig_neg - (1.f - t) * sig;    // = t - sigmoid(a)
    r = 0.f; rgate = 0.f;
  } else if constexpr (KIND == LK_POISSON) {
    const float ll = fminf(fmaxf(a[0], -10.f), 10.f);
    const float lam = __expf(ll);
    lp = t * ll - lam;
    if (GRAD) g[0] = (a[0] >= -10.f && a[0] <= 10.f) ? (t - lam) : 0.f;
    r = 0.f; rgate = 0.f;
  } else if constexpr (KIND == LK_NB) {
    const float ap = fmaxf(a[0], LOGIT_OF_TINY);
    float logp, log1mp, p, q;                 // q = 1 - p
    log_sigmoid_pair(ap, logp, log1mp, p, q);
    const float lr = fminf(fmaxf(a[1], -10.f), 10.f);
    r = __expf(lr);
    rgate = (a[1] >= -10.f && a[1] <= 10.f) ? 1.f : 0.f;
    lp = r * log1mp + t * logp;
    if (GRAD) {
      g[0] = (a[0] >= LOGIT_OF_TINY) ? (t * q - r * p) : 0.f;
      g[1] = rgate * r * log1mp;
    }
  } else {
    // zero-inflated: head 0 is pi, the rest belong to the base distribution
    float lpb;
    float gb[2];
    if constexpr (KIND == LK_ZIP) {
      lik_dense<LK_POISSON, GRAD>(t, a + 1, lpb, gb, r, rgate);
    } else {
      lik_dense<LK_NB, GRAD>(t, a + 1, lpb, gb, r, rgate);
    }
    const float api = fmaxf(a[0], LOGIT_OF_TINY);
    float logpi, log1mpi, pi, qi;             // qi = 1 - pi
    log_sigmoid_pair(api, logpi, log1mpi, pi, qi);
    const bool gate = a[0] >= LOGIT_OF_TINY;
    constexpr int NB_HEADS = (KIND == LK_ZIP) ? 1 : 2;
    if (t > 0.f) {
      lp = log1mpi + lpb;
      if (GRAD) {
        g[0] = gate ? -pi : 0.f;
#pragma unroll
        for (int j = 0; j < NB_HEADS; ++j) g[1 + j] = gb[j];
      }
    } else {
      // zero branch: at t == 0 the base log-probability has no sparse correction
      const float u1 = logpi, u2 = log1mpi + lpb;
      const float m = fmaxf(u1, u2);
      const float y0 = m + fast_log1p(__expf(-fabsf(u1 - u2)));
      lp = y0;
      if (GRAD) {
        const float u = __expf(u1 - y0);   // pi / (pi + (1-pi) e^lpb)
        const float w = __expf(u2 - y0);   // 1 - u
        g[0] = gate ? (u * qi - w * pi) : 0.f;
#pragma unroll
        for (int j = 0; j < NB_HEADS; ++j) g[1 + j] = w * gb[j];
      }
    }
  }
}

// lp  : log p(t | theta) + lgamma(1+t)
// g[] : d lp / d pre-activation of each head (only if GRAD)
template <int KIND, bool GRAD>
__device__ __forceinline__ void lik_elem(float t, const float* a, float& lp, float* g) {
  float r, rgate;
  lik_dense<KIND, GRAD>(t, a, lp, g, r, rgate);
  if constexpr (LikelihoodTraits<KIND>::HAS_R) {
    if (t > 0.f) {
      float A, D;
      lgamma_digamma_diff<GRAD>(r, t, A, D);
      lp += A;
      if (GRAD) g[LikelihoodTraits<KIND>::P - 1] += rgate * r * D;
    }
  }
}

// E[x|z] and Var[x|z] (TFP semantics; evaluate-time statistics, va:2665-2713)
template <int KIND>
__device__ __forceinline__ void lik_mean_var(const float* a, float& mean, float& var) {
  if constexpr (KIND == LK_CPOISSON) {
    // (the head has been normalised to the rate lambda * N by cpoisson_rate_rows)
    mean = a[0]; var = a[0];
  } else if constexpr (KIND == LK_BERNOULLI) {
    const float pr = sigmoidf(a[0]);
    mean = pr; var = pr * (1.f - pr);
  } else if constexpr (KIND == LK_POISSON) {
    const float lam = __expf(fminf(fmaxf(a[0], -10.f), 10.f));
    mean = lam; var = lam;
  } else if constexpr (KIND == LK_NB) {
    const float ap = fmaxf(a[0], LOGIT_OF_TINY);
    const float r = __expf(fminf(fmaxf(a[1], -10.f), 10.f));
    mean = r * __expf(ap);               // r * p/(1-p)
    var = mean * (1.f + __expf(ap));     // mean / (1-p)
  } else {
    float m, v;
    if constexpr (KIND == LK_ZIP) lik_mean_var<LK_POISSON>(a + 1, m, v);
    else lik_mean_var<LK_NB>(a + 1, m, v);
    const float api = fmaxf(a[0], LOGIT_OF_TINY);
    const float omp = sigmoidf(-api);
    mean = omp * m;
    var = omp * (v + m * m) - mean * mean;
  }
}
#endif

}  // namespace scvae
