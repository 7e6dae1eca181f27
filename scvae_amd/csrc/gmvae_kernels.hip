// Small kernels specific to the Gaussian-mixture VAE
// (scvae/models/gaussian_mixture_variational_autoencoder.py:2788-3434): categorical
// q(y|x), softplus-Gaussian q(z|x,y=k) / p(z|y=k) sample + log-ratio, the mixture ELBO and
// their backward passes.  All tensors here are [cells x <=128]-sized; the kernels are
// HBM/latency trivial next to the decoder, so they favour clarity and determinism.
#include "common.hpp"
#include "kernels.hpp"

namespace scvae {

// out[k, b, :] = [relu](a0[b, :] + rows[k, :])  -- one-hot input column of q(z|x,y=k)
__global__ void add_group_rows_kernel(const float* __restrict__ a0, const float* __restrict__ rows,
                                      float* __restrict__ out, int K, int B, int N, int relu) {
  const size_t total = (size_t)K * B * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % N);
    const size_t r = i / N;
    const int b = (int)(r % B), k = (int)(r / B);
    float v = a0[(size_t)b * N + c] + rows[(size_t)k * N + c];
    if (relu) v = fmaxf(v, 0.f);
    out[i] = v;
  }
}
int add_group_rows(hipStream_t s, const float* a0, const float* rows, float* out, int K, int B,
                   int N, int relu) {
  SCVAE_ARG(a0 && rows && out);
  const size_t total = (size_t)K * B * N;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(add_group_rows_kernel, dim3(blocks), dim3(256), 0, s, a0, rows, out, K, B, N,
                     relu);
  SCVAE_LAUNCH_CHECK("add_group_rows_kernel");
  return 0;
}

// out[g, c] = sum_r a[g*R + r, c]  (row-chunked partials + fixed-order finalize)
__global__ __launch_bounds__(1024) void group_col_sum_partial_kernel(const float* __restrict__ a,
                                                                     int lda, int R, int N,
                                                                     int chunk,
                                                                     float* __restrict__ partial) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + cl;
  const int rl = threadIdx.x >> 6;
  const int g = blockIdx.y, z = blockIdx.z, G = gridDim.y;
  const int r0 = z * chunk, r1 = min(R, r0 + chunk);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 16) s += a[((size_t)g * R + r) * lda + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cl];
    partial[((size_t)z * G + g) * N + c] = t;
  }
}
__global__ void group_col_sum_finalize_kernel(const float* __restrict__ partial, int N, int chunks,
                                              int G, float scale, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (c >= N) return;
  float t = 0.f;
  for (int z = 0; z < chunks; ++z) t += partial[((size_t)z * G + g) * N + c];
  out[(size_t)g * N + c] = t * scale;
}
int group_col_sum(hipStream_t s, const float* a, int lda, int R, int G, int N, float scale,
                  float* out, float* partial) {
  SCVAE_ARG(a && out && partial && R > 0 && G > 0 && N > 0);
  int chunks = (R + 255) / 256;
  if (chunks > BN_MAX_CHUNKS) chunks = BN_MAX_CHUNKS;
  const int chunk = (R + chunks - 1) / chunks;
  chunks = (R + chunk - 1) / chunk;
  hipLaunchKernelGGL(group_col_sum_partial_kernel, dim3((N + 63) / 64, G, chunks), dim3(1024), 0,
                     s, a, lda, R, N, chunk, partial);
  SCVAE_LAUNCH_CHECK("group_col_sum_partial_kernel");
  hipLaunchKernelGGL(group_col_sum_finalize_kernel, dim3((N + 63) / 64, G), dim3(64), 0, s,
                     partial, N, chunks, G, scale, out);
  SCVAE_LAUNCH_CHECK("group_col_sum_finalize_kernel");
  return 0;
}

// out[r, c] = sum_g w[r, g] * a[g*R + r, c]   (w == nullptr: plain sum over groups)
__global__ void sum_groups_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                  int ldw, int G, int R, int N, float* __restrict__ out) {
  const size_t total = (size_t)R * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % N);
    const size_t r = i / N;
    float s = 0.f;
    for (int g = 0; g < G; ++g) {
      const float v = a[((size_t)g * R + r) * N + c];
      s += w ? w[r * ldw + g] * v : v;
    }
    out[i] = s;
  }
}
int sum_groups(hipStream_t s, const float* a, const float* w, int ldw, int G, int R, int N,
               float* out) {
  SCVAE_ARG(a && out);
  const size_t total = (size_t)R * N;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sum_groups_kernel, dim3(blocks), dim3(256), 0, s, a, w, ldw, G, R, N, out);
  SCVAE_LAUNCH_CHECK("sum_groups_kernel");
  return 0;
}

// q(y|x) = Categorical(logits) (gm:3050-3092): y = softmax, KL(q(y|x) || uniform) = log K - H[q]
// one wave per cell
// log-normaliser of the prior logits (wave-wide; K is small)
__device__ __forceinline__ float prior_log_normaliser(const float* __restrict__ m, int K,
                                                      int lane) {
  float mx = -INFINITY;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, m[k]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int k = lane; k < K; k += 64) se += __expf(m[k] - mx);
  se = wave_sum(se);
  return mx + __logf(se);
}

__global__ __launch_bounds__(256) void categorical_fwd_kernel(const float* __restrict__ logits,
                                                              float* __restrict__ y,
                                                              float* __restrict__ kl_y_cell, int B,
                                                              int K,
                                                              const float* __restrict__ prior) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* row = logits + (size_t)b * K;
  float mx = -INFINITY;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, row[k]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int k = lane; k < K; k += 64) se += __expf(row[k] - mx);
  se = wave_sum(se);
  const float lse = mx + __logf(se);
  const float lse_p = prior ? prior_log_normaliser(prior, K, lane) : 0.f;
  float h = 0.f;   // uniform: -sum q log q; otherwise sum q (log q - log p)
  for (int k = lane; k < K; k += 64) {
    const float ly = row[k] - lse;
    const float p = __expf(ly);
    y[(size_t)b * K + k] = p;
    if (prior) h += p * (ly - (prior[k] - lse_p));
    else h -= p * ly;
  }
  h = wave_sum(h);
  if (lane == 0) kl_y_cell[b] = prior ? h : __logf((float)K) - h;
}
int categorical_fwd(hipStream_t s, const float* logits, float* y, float* kl_y_cell, int B, int K,
                    const float* prior_logits) {
  SCVAE_ARG(logits && y && kl_y_cell && K > 0);
  if (B == 0) return 0;
  hipLaunchKernelGGL(categorical_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, s, logits, y,
                     kl_y_cell, B, K, prior_logits);
  SCVAE_LAUNCH_CHECK("categorical_fwd_kernel");
  return 0;
}

// dlogits_j = y_j (dy_j - sum_k y_k dy_k) + c * y_j (log y_j + H)
__global__ __launch_bounds__(256) void categorical_bwd_kernel(const float* __restrict__ y,
                                                              const float* __restrict__ dy,
                                                              float c, float* __restrict__ dlogits,
                                                              int B, int K) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  float dot = 0.f, h = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float p = y[(size_t)b * K + k];
    dot += p * dy[(size_t)b * K + k];
    h -= p > 0.f ? p * __logf(p) : 0.f;
  }
  dot = wave_sum(dot);
  h = wave_sum(h);
  for (int k = lane; k < K; k += 64) {
    const float p = y[(size_t)b * K + k];
    const float lp = p > 0.f ? __logf(p) : 0.f;
    dlogits[(size_t)b * K + k] = p * (dy[(size_t)b * K + k] - dot) + c * p * (lp + h);
  }
}
int categorical_bwd(hipStream_t s, const float* y, const float* dy, float c, float* dlogits, int B,
                    int K) {
  SCVAE_ARG(y && dy && dlogits);
  if (B == 0) return 0;
  hipLaunchKernelGGL(categorical_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, s, y, dy, c, dlogits,
                     B, K);
  SCVAE_LAUNCH_CHECK("categorical_bwd_kernel");
  return 0;
}

__device__ __forceinline__ float clip_big(float v) {
  return fminf(fmaxf(v, -F32_MAX_HALF), F32_MAX_HALF);
}

// "softplus gaussian" posterior/prior (du:52-73): sigma = sqrt(softplus(s)).
// z[k,s,b,:] = mean + sigma*eps ; klz[k,s,b] = sum_l log q(z) - log p(z|y=k)
// one workgroup (64*ceil(L/64) threads) per (k, b); prior parameters p(z|y=k) are row k of
// the Z/P dense layers on the one-hot input (gm:3009-3048).
__global__ void softplus_gaussian_fwd_kernel(
    const float* __restrict__ qm, const float* __restrict__ qs, const float* __restrict__ Wpm,
    const float* __restrict__ bpm, const float* __restrict__ Wps, const float* __restrict__ bps,
    const float* __restrict__ eps, float* __restrict__ z, float* __restrict__ klz,
    float* __restrict__ qvar, int K, int S, int B, int L) {
  __shared__ float red[16];
  const int b = blockIdx.x, k = blockIdx.y, l = threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float m = 0.f, sg = 1.f, pm = 0.f, sp = 1.f;
  if (l < L) {
    const size_t i = ((size_t)k * B + b) * L + l;
    m = clip_big(qm[i]);
    sg = sqrtf(softplusf(clip_big(qs[i])));
    pm = clip_big(Wpm[(size_t)k * L + l] + bpm[l]);
    sp = sqrtf(softplusf(clip_big(Wps[(size_t)k * L + l] + bps[l])));
    if (qvar) qvar[i] = sg * sg;
  }
  for (int s = 0; s < S; ++s) {
    float kl = 0.f;
    if (l < L) {
      const size_t o = (((size_t)k * S + s) * B + b) * L + l;
      const float e = eps[o];
      const float zz = fmaf(sg, e, m);
      z[o] = zz;
      const float u = (zz - pm) / sp;
      kl = -0.5f * e * e - __logf(sg) + 0.5f * u * u + __logf(sp);
    }
    kl = wave_sum(kl);
    __syncthreads();
    if (lane == 0) red[w] = kl;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < nw; ++i) t += red[i];
      klz[((size_t)k * S + s) * B + b] = t;
    }
  }
}
int softplus_gaussian_fwd(hipStream_t st, const float* qm, const float* qs, const float* Wpm,
                          const float* bpm, const float* Wps, const float* bps, const float* eps,
                          float* z, float* klz, float* qvar, int K, int S, int B, int L) {
  SCVAE_ARG(qm && qs && Wpm && bpm && Wps && bps && eps && z && klz);
  SCVAE_ARG(L > 0 && L <= 1024);
  if (B == 0) return 0;
  hipLaunchKernelGGL(softplus_gaussian_fwd_kernel, dim3(B, K), dim3((L + 63) / 64 * 64), 0, st, qm,
                     qs, Wpm, bpm, Wps, bps, eps, z, klz, qvar, K, S, B, L);
  SCVAE_LAUNCH_CHECK("softplus_gaussian_fwd_kernel");
  return 0;
}

// backward of the above.  dz [K,S,B,L]: gradient from the decoder; gklz [K,S,B]: d loss / d klz.
// Outputs dqm, dqs [K*B, L] and per-element prior gradients dpr [K*B, 2L] = (d pm | d ps-pre)
__global__ void softplus_gaussian_bwd_kernel(
    const float* __restrict__ qm, const float* __restrict__ qs, const float* __restrict__ Wpm,
    const float* __restrict__ bpm, const float* __restrict__ Wps, const float* __restrict__ bps,
    const float* __restrict__ eps, const float* __restrict__ dz, const float* __restrict__ gklz,
    float* __restrict__ dqm, float* __restrict__ dqs, float* __restrict__ dpr, int K, int S, int B,
    int L) {
  const size_t total = (size_t)K * B * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i % L);
    const size_t kb = i / L;
    const int b = (int)(kb % B), k = (int)(kb / B);
    const float qm_pre = qm[i], qs_pre = qs[i];
    const float m = clip_big(qm_pre);
    const float sq = clip_big(qs_pre);
    const float sg = sqrtf(softplusf(sq));
    const float pm_pre = Wpm[(size_t)k * L + l] + bpm[l];
    const float ps_pre = Wps[(size_t)k * L + l] + bps[l];
    const float pm = clip_big(pm_pre);
    const float sp = sqrtf(softplusf(clip_big(ps_pre)));
    float gm = 0.f, gs = 0.f, gpm = 0.f, gsp = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t row = ((size_t)k * S + s) * B + b;
      const size_t o = row * L + l;
      const float e = eps[o];
      const float g = gklz[row];
      const float zz = fmaf(sg, e, m);
      const float u = (zz - pm) / sp;
      const float dze = dz[o] + g * u / sp;   // d/dz of decoder path + KL path
      gm += dze;
      gs += dze * e - g / sg;
      gpm -= g * u / sp;
      gsp += g * (1.f - u * u) / sp;
    }
    const float big = F32_MAX_HALF;
    dqm[i] = (qm_pre >= -big && qm_pre <= big) ? gm : 0.f;
    dqs[i] = (qs_pre >= -big && qs_pre <= big) ? gs * sigmoidf(sq) / (2.f * sg) : 0.f;
    dpr[kb * 2 * L + l] = gpm;
    dpr[kb * 2 * L + L + l] = gsp * sigmoidf(clip_big(ps_pre)) / (2.f * sp);
  }
}
int softplus_gaussian_bwd(hipStream_t st, const float* qm, const float* qs, const float* Wpm,
                          const float* bpm, const float* Wps, const float* bps, const float* eps,
                          const float* dz, const float* gklz, float* dqm, float* dqs, float* dpr,
                          int K, int S, int B, int L) {
  SCVAE_ARG(qm && qs && eps && dz && gklz && dqm && dqs && dpr);
  const size_t total = (size_t)K * B * L;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(softplus_gaussian_bwd_kernel, dim3(blocks), dim3(256), 0, st, qm, qs, Wpm,
                     bpm, Wps, bps, eps, dz, gklz, dqm, dqs, dpr, K, S, B, L);
  SCVAE_LAUNCH_CHECK("softplus_gaussian_bwd_kernel");
  return 0;
}

// Mixture ELBO (gm:3242-3410), phase A: this rank's share of the batch means.
// sums[0] reconstruction_error, [1] kl_divergence_z, [2] kl_divergence_y  (single workgroup)
__global__ __launch_bounds__(256) void gmvae_elbo_sums_kernel(const float* __restrict__ ll,
                                                              const float* __restrict__ klz,
                                                              const float* __restrict__ y,
                                                              const float* __restrict__ kl_y_cell,
                                                              int K, int S, int B, float inv_gb,
                                                              float* __restrict__ sums,
                                                              float* __restrict__ rec_cell) {
  __shared__ float red[4];
  float rec = 0.f, kz = 0.f, ky = 0.f;
  const float inv_s = 1.f / (float)S;
  for (int b = threadIdx.x; b < B; b += 256) {
    float rc = 0.f, kc = 0.f;
    if (S == 1) {
      // (the loads of eight passes in flight; same order of the sums as the general loop)
      for (int k0 = 0; k0 < K; k0 += 8) {
        float av[8], cv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = min(k0 + u, K - 1);
          av[u] = ll[(size_t)k * B + b];
          cv[u] = klz[(size_t)k * B + b];
          yv[u] = y[(size_t)b * K + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + u < K) {
            rc += (0.f + av[u]) * inv_s * yv[u];
            kc += (0.f + cv[u]) * inv_s * yv[u];
          }
        }
      }
    } else
    for (int k = 0; k < K; ++k) {
      float a = 0.f, c = 0.f;
      for (int s = 0; s < S; ++s) {
        const size_t row = ((size_t)k * S + s) * B + b;
        a += ll[row];
        c += klz[row];
      }
      const float yk = y[(size_t)b * K + k];
      rc += a * inv_s * yk;
      kc += c * inv_s * yk;
    }
    if (rec_cell) rec_cell[b] = rc;
    rec += rc; kz += kc; ky += kl_y_cell[b];
  }
  rec = block_sum<256>(rec, red);
  kz = block_sum<256>(kz, red);
  ky = block_sum<256>(ky, red);
  if (threadIdx.x == 0) {
    sums[0] = rec * inv_gb;
    sums[1] = kz * inv_gb;
    sums[2] = ky * inv_gb;
  }
}
// phase B: scalars from the (global) sums; free-nats gate (gm:3391-3398) -> gate[0]
__global__ void gmvae_elbo_finish_kernel(const float* __restrict__ sums, float w, float thr,
                                         int use_free_nats, float share,
                                         float* __restrict__ scalars, float* __restrict__ gate,
                                         const float* __restrict__ prior, int K, float free_nats) {
  const float rec = sums[0], kz = sums[1], ky = sums[2];
  if (prior != nullptr) {   // thr = free_nats * H[p(y)] (gm:3258-3261)
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, prior[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += __expf(prior[k] - mx);
    const float lse = mx + __logf(se);
    float h = 0.f;
    for (int k = 0; k < K; ++k) {
      const float lp = prior[k] - lse;
      h -= __expf(lp) * lp;
    }
    thr = free_nats * h;
  }
  const float ky_mod = use_free_nats ? (ky > thr ? ky : thr) : ky;
  // `share` scales the global values back to this rank's share (scalars are summed by the
  // caller over ranks); 1 on a single GPU
  scalars[0] = (rec - (kz + ky)) * share;
  scalars[1] = (rec - w * (kz + ky_mod)) * share;
  scalars[2] = rec * share;
  scalars[3] = kz * share;
  scalars[4] = ky * share;
  if (!isfinite(rec - (kz + ky))) scalars[7] += 1.f;   // sticky non-finite counter (gm:1124-1127)
  gate[0] = use_free_nats ? (ky > thr ? 1.f : 0.f) : 1.f;
}
int gmvae_elbo(hipStream_t s, const float* ll, const float* klz, const float* y,
               const float* kl_y_cell, int K, int S, int B, float inv_gb, float* sums,
               float* rec_cell) {
  SCVAE_ARG(ll && klz && y && kl_y_cell && sums);
  hipLaunchKernelGGL(gmvae_elbo_sums_kernel, dim3(1), dim3(256), 0, s, ll, klz, y, kl_y_cell, K, S,
                     B, inv_gb, sums, rec_cell);
  SCVAE_LAUNCH_CHECK("gmvae_elbo_sums_kernel");
  return 0;
}
int gmvae_elbo_finish(hipStream_t s, const float* sums, float w, float thr, int use_free_nats,
                      float share, float* scalars, float* gate, const float* prior_logits, int K,
                      float free_nats) {
  SCVAE_ARG(sums && scalars && gate);
  hipLaunchKernelGGL(gmvae_elbo_finish_kernel, dim3(1), dim3(1), 0, s, sums, w, thr, use_free_nats,
                     share, scalars, gate, prior_logits, K, free_nats);
  SCVAE_LAUNCH_CHECK("gmvae_elbo_finish_kernel");
  return 0;
}

// gradients of -lower_bound_weighted w.r.t. ll, klz rows and y; dlogits via the softmax
// gw[k,s,b] = -y_bk/(S GB); gklz = +w y_bk/(S GB); dy_bk = (-mean_s ll + w mean_s klz)/GB
__global__ __launch_bounds__(256) void gmvae_elbo_bwd_kernel(
    const float* __restrict__ ll, const float* __restrict__ klz, const float* __restrict__ y,
    const float* __restrict__ gate, int K, int S, int B, float w, float inv_gb,
    float* __restrict__ gw, float* __restrict__ gklz, float* __restrict__ dy) {
  const size_t total = (size_t)B * K;
  const float inv_s = 1.f / (float)S;
  (void)gate;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int b = (int)(i / K);
    const float yk = y[i];
    float a = 0.f, c = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t row = ((size_t)k * S + s) * B + b;
      a += ll[row];
      c += klz[row];
      gw[row] = -yk * inv_s * inv_gb;
      gklz[row] = w * yk * inv_s * inv_gb;
    }
    dy[i] = (-a * inv_s + w * c * inv_s) * inv_gb;
  }
}
int gmvae_elbo_bwd(hipStream_t s, const float* ll, const float* klz, const float* y,
                   const float* gate, int K, int S, int B, float w, float inv_gb, float* gw,
                   float* gklz, float* dy) {
  SCVAE_ARG(ll && klz && y && gw && gklz && dy);
  const size_t total = (size_t)B * K;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(gmvae_elbo_bwd_kernel, dim3(blocks), dim3(256), 0, s, ll, klz, y, gate, K, S,
                     B, w, inv_gb, gw, gklz, dy);
  SCVAE_LAUNCH_CHECK("gmvae_elbo_bwd_kernel");
  return 0;
}

// categorical backward with the free-nats gate read from device memory:
// dlogits = softmax-bwd(dy) + (w * gate / GB) * d kl_y_cell
__global__ __launch_bounds__(256) void categorical_bwd_gated_kernel(
    const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ gate,
    float c, float* __restrict__ dlogits, int B, int K, const float* __restrict__ prior) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  const float cc = c * gate[0];
  const float lse_p = prior ? prior_log_normaliser(prior, K, lane) : 0.f;
  // h = -KL_b + const: the KL gradient is q_j (log q_j - log p_j - KL_b); with the uniform prior
  // log p_j is constant and drops out against KL_b = log K - H
  float dot = 0.f, h = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float p = y[(size_t)b * K + k];
    dot += p * dy[(size_t)b * K + k];
    const float lq = p > 0.f ? __logf(p) : 0.f;
    h -= p * (lq - (prior ? prior[k] - lse_p : 0.f));
  }
  dot = wave_sum(dot);
  h = wave_sum(h);
  for (int k = lane; k < K; k += 64) {
    const float p = y[(size_t)b * K + k];
    const float lq = p > 0.f ? __logf(p) : 0.f;
    const float rel = lq - (prior ? prior[k] - lse_p : 0.f);
    dlogits[(size_t)b * K + k] = p * (dy[(size_t)b * K + k] - dot) + cc * p * (rel + h);
  }
}
int categorical_bwd_gated(hipStream_t s, const float* y, const float* dy, const float* gate,
                          float c, float* dlogits, int B, int K, const float* prior_logits) {
  SCVAE_ARG(y && dy && gate && dlogits);
  if (B == 0) return 0;
  hipLaunchKernelGGL(categorical_bwd_gated_kernel, dim3((B + 3) / 4), dim3(256), 0, s, y, dy, gate,
                     c, dlogits, B, K, prior_logits);
  SCVAE_LAUNCH_CHECK("categorical_bwd_gated_kernel");
  return 0;
}

// gradient of w * max(mean_b KL_y, free_nats * H[p]) w.r.t. the prior logits m (one workgroup):
//   KL_b = sum_k q_bk (log q_bk - log p_k)        =>  dKL_b/dm_j = p_j - q_bj
//   H[p] = -sum_k p_k log p_k                      =>  dH/dm_j   = -p_j (log p_j + H)
__global__ __launch_bounds__(256) void prior_logits_bwd_kernel(
    const float* __restrict__ y, const float* __restrict__ prior, const float* __restrict__ gate,
    float c, float off_scale, float free_nats, int B, int K, float* __restrict__ dprior) {
  __shared__ float red[4];
  float mx = -INFINITY;
  for (int k = 0; k < K; ++k) mx = fmaxf(mx, prior[k]);
  float se = 0.f;
  for (int k = 0; k < K; ++k) se += __expf(prior[k] - mx);
  const float lse = mx + __logf(se);
  float H = 0.f;
  for (int k = 0; k < K; ++k) {
    const float lp = prior[k] - lse;
    H -= __expf(lp) * lp;
  }
  const bool on = gate[0] != 0.f;
  for (int j = 0; j < K; ++j) {
    const float lp = prior[j] - lse, pj = __expf(lp);
    float acc = 0.f;
    if (on)
      for (int b = threadIdx.x; b < B; b += 256) acc += pj - y[(size_t)b * K + j];
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0)
      dprior[j] = on ? c * acc : off_scale * free_nats * (-pj * (lp + H));
    __syncthreads();
  }
}
int prior_logits_bwd(hipStream_t s, const float* y, const float* prior_logits, const float* gate,
                     float c, float off_scale, float free_nats, int B, int K, float* dprior) {
  SCVAE_ARG(y && prior_logits && gate && dprior && K > 0);
  hipLaunchKernelGGL(prior_logits_bwd_kernel, dim3(1), dim3(256), 0, s, y, prior_logits, gate, c,
                     off_scale, free_nats, B, K, dprior);
  SCVAE_LAUNCH_CHECK("prior_logits_bwd_kernel");
  return 0;
}

// p(z|y=k) statistics for logging (gm:2879-2882): means and variances [K, L]
__global__ void prior_stats_kernel(const float* __restrict__ Wpm, const float* __restrict__ bpm,
                                   const float* __restrict__ Wps, const float* __restrict__ bps,
                                   int K, int L, float* __restrict__ means,
                                   float* __restrict__ variances) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * L) return;
  const int l = i % L;
  means[i] = clip_big(Wpm[i] + bpm[l]);
  variances[i] = softplusf(clip_big(Wps[i] + bps[l]));
}
int prior_stats(hipStream_t s, const float* Wpm, const float* bpm, const float* Wps,
                const float* bps, int K, int L, float* means, float* variances) {
  SCVAE_ARG(Wpm && bpm && Wps && bps && means && variances);
  hipLaunchKernelGGL(prior_stats_kernel, dim3((K * L + 255) / 256), dim3(256), 0, s, Wpm, bpm, Wps,
                     bps, K, L, means, variances);
  SCVAE_LAUNCH_CHECK("prior_stats_kernel");
  return 0;
}

}  // namespace scvae
