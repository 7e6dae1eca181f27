// Host-side launchers of the gfx950 kernels (internal C++ interface; the
// exported C ABI is include/scvae_hip.h).  Every launcher is asynchronous on
// `stream`, never allocates, and returns 0 / -1 (bad argument) / -2 (HIP error).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace scvae {

enum Activation : int { ACT_NONE = 0, ACT_RELU = 1 };

// ---- gemm.hip ----
int gemm_choose_splits(int M, int N, int K);
size_t gemm_workspace_bytes(int M, int N, int K);
int gemm(hipStream_t stream, bool ta, bool tb, const float* A, const float* B, const float* bias,
         float* C, int M, int N, int K, int lda, int ldb, int ldc, int act, bool accumulate,
         float* workspace, size_t workspace_bytes);

// ---- count_gemm.hip: the encoder input layer's two large products on a count matrix ----
// (x integers in [0, 65536): exact bf16 hi/lo cut of x times an exact three-term bf16 split of
// the fp32 operand, fp32 accumulation)
// mode 0: C[rows, N] = x[rows, cols] other[cols, N] + bias;  mode 1: C[cols, N] = x^T other[rows, N]
bool count_gemm_supported(int N);
size_t count_gemm_workspace_bytes(int mode, int rows, int cols, int N);
int count_gemm(hipStream_t stream, int mode, const float* x, int ldx, int rows, int cols,
               const float* other, int ld_other, int N, const float* bias, int act, float* C,
               int ldc, void* workspace, size_t workspace_bytes);
// the same with x as uint16 counts (row pitch ldx even, base 4-byte aligned)
int count_gemm_u16(hipStream_t stream, int mode, const uint16_t* x, int ldx, int rows, int cols,
               const float* other, int ld_other, int N, const float* bias, int act, float* C,
               int ldc, void* workspace, size_t workspace_bytes);
// The same minibatch as a list of its non-zeros grouped by (16 rows, 32 genes) -- count_gemm.hip
// describes the format -- and the two products read from it (bit-identical to count_gemm_u16).
struct CountTiles {
  uint32_t* ent = nullptr;    // [groups][cap]
  uint32_t* tptr = nullptr;   // [groups][count_tiles_padded(F) + 1]
  uint32_t* gptr = nullptr;   // [groups][count_tiles_padded(F) / 16 + 1]
  int64_t cap = 0;            // entries per group of 16 rows
  int* status = nullptr;      // optional device word: bit 0 set when a group overflowed cap
};
int count_tiles_padded(int F);
bool count_tiles_supported(int F);
int csr_row_entries(hipStream_t stream, const int64_t* indptr, const float* values, int64_t n_rows,
                    int32_t* out);
int csr_count_tiles(hipStream_t stream, const int64_t* indptr, const int32_t* indices,
                    const float* values, const int64_t* rows, int B, int F, CountTiles tiles);
int count_gemm_tiles(hipStream_t stream, int mode, CountTiles tiles, const uint16_t* x, int ldx,
                     int rows, int cols, const float* other, int ld_other, int N,
                     const float* bias, int act, float* C, int ldc, void* workspace,
                     size_t workspace_bytes);
// *bad = 1 unless every value is an integer in [0, 65536) (the precondition of count_gemm)
int check_counts(hipStream_t stream, const float* values, size_t n, int* bad);

// ---- elementwise.hip ----
struct HeadPtrs {
  float* p[3];
};
// per-row log-likelihood sums; rows r = s*B + b use target row b
int loglik_fwd(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre, int ldp,
               const float* row_const, float* ll, int rows, int B, int F);
// in place: pre_j <- gw[r] * d loglik / d pre_j ; also (re)computes ll if ll != null
int loglik_bwd(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre, int ldp,
               const float* gw, const float* row_const, float* ll, int rows, int B, int F);
// constrained Poisson (du:218-228): softmax over the genes times the cell's count sum; `pre`
// [rows, F] logits, overwritten with the upstream-scaled gradient (bwd) or the rate (rate)
int cpoisson_fwd(hipStream_t stream, const float* t, int ldt, float* pre, int ldp,
                 const float* count_sum, const float* row_const, float* ll, int rows, int B, int F);
int cpoisson_bwd(hipStream_t stream, const float* t, int ldt, float* pre, int ldp, const float* gw,
                 const float* count_sum, const float* row_const, float* ll, int rows, int B, int F);
int cpoisson_rate(hipStream_t stream, float* pre, int ldp, const float* count_sum, int rows, int B,
                  int F);
// Piecewise categorical likelihood (`Categorised`, distributions/categorised.py:210-263; -k):
// `logits` [rows, F*(K+1)] of the P_K head, class c of feature f at f*(K+1)+c.  Counts below K are
// classes of the categorical, class K hands the excess t-K to the count distribution (kind:
// Poisson or negative binomial).  GRAD form: pre_j and logits are overwritten by
// gw[r] * d loglik / d(.).  The data term lgamma(1 + t - K) is evaluated inline.
int loglik_cat_fwd(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre, int ldp,
                   float* logits, int K, float* ll, int rows, int B, int F);
int loglik_cat_bwd(hipStream_t stream, int kind, const float* t, int ldt, HeadPtrs pre, int ldp,
                   float* logits, int K, const float* gw, float* ll, int rows, int B, int F);
// px_statistics with the categorised mean / variance (categorised.py:210-253)
int px_statistics_cat(hipStream_t stream, int kind, HeadPtrs pre, int ldp, const float* logits,
                      int K, int S, int B, int F, const float* weight, int ldw, int accumulate,
                      float* p_x_mean, float* mean_of_var, float* var_of_mean);
// evaluate-time statistics over the S samples of each cell (va:2665-2713):
// p_x_mean, p_x_stddev, stddev_of_p_x_given_z_mean, each [B,F]. `weight` (optional, [B], stride
// ldw) and `accumulate` implement the GMVAE mixture sums (gm:3311-3386).
int px_statistics(hipStream_t stream, int kind, HeadPtrs pre, int ldp, int S, int B, int F,
                  const float* weight, int ldw, int accumulate, float* p_x_mean,
                  float* mean_of_var, float* var_of_mean);
int loglik_elementwise(hipStream_t stream, int kind, const float* t, HeadPtrs pre, float* out,
                       float* mean, float* var, size_t n);
int sqrt_sum(hipStream_t stream, const float* a, const float* b, float* out, size_t n);

// Gaussian posterior: clip, reparameterise, KL (va:2266-2289, 2346-2369, 2624-2656).
// kl_sample / kl_gw non-null: Monte-Carlo KL per sample row (va:2633-2640), else analytic per
// cell; ls_pre == nullptr: unit-variance posterior (du:323-337).
int gauss_latent_fwd(hipStream_t stream, const float* mu_pre, const float* ls_pre,
                     const float* eps, float* z, float* kl_elem, float* kl_cell,
                     float* kl_sample, int S, int B, int L, int deterministic);
int gauss_latent_bwd(hipStream_t stream, const float* mu_pre, const float* ls_pre,
                     const float* eps, const float* dz, float kl_coeff, const float* kl_gw,
                     float* dmu_pre, float* dls_pre, int S, int B, int L);
// ELBO terms and d(-ELBO_weighted)/d loglik (va:2717-2734, mu:129-137)
// scalars: [0] lower_bound [1] lower_bound_weighted [2] reconstruction_error [3] kl_divergence
// kl_per_sample: kl_cell holds [n_iw * n_mc * B] Monte-Carlo values instead of [B] analytic ones
int vae_elbo(hipStream_t stream, const float* ll, const float* kl_cell, int kl_per_sample,
             int n_iw, int n_mc, int B, float kl_weight_total, float row_scale, float* scalars,
             float* gw);

// batch normalisation (tf.contrib.layers.batch_norm(center=True, scale=False), fused semantics)
constexpr int BN_MAX_CHUNKS = 64;
constexpr int COLSUM_MAX_CHUNKS = 32;
size_t bn_partial_floats(int groups, int N);
size_t col_sum_partial_floats(int N);
int bn_stats(hipStream_t stream, const float* a, int lda, int rows_per_group, int groups, int N,
             float* mean, float* var, float* partial);
int bn_stats_partial(hipStream_t stream, const float* a, int lda, int rows, int N, float* partial,
                     int* chunk_out, int* chunks_out);
int bn_apply(hipStream_t stream, const float* a, int lda, const float* mean, const float* var,
             int stat_stride, const float* beta, float* h, int ldh, int rows_per_group, int groups,
             int N, int relu);
int bn_bwd_stats(hipStream_t stream, const float* dh, int lddh, const float* h, int ldh,
                 const float* a, int lda, const float* mean, const float* var, int rows_per_group,
                 int groups, int N, int relu, float* s1, float* s2, float* partial,
                 float* dbeta = nullptr, float* moving_mean = nullptr,
                 float* moving_var = nullptr, int64_t global_rows_per_group = 1);
int bn_bwd_apply(hipStream_t stream, const float* dh, int lddh, const float* h, int ldh,
                 const float* a, int lda, const float* mean, const float* var, const float* s1,
                 const float* s2, int rows_per_group, int groups, int N, int relu, float inv_count,
                 float* da, int ldda);
int bn_merge(hipStream_t stream, const float* gathered, const int64_t* counts, int ranks, int n,
             float* out);
// one-launch (column-parallel) batch norm of a single group of <= 8192 rows, N % 4 == 0, 16-byte
// aligned rows: statistics + normalise (+ relu) / backward sums + dbeta + moving averages + da
bool bn_cols_supported(int rows, int N);
bool bn_cols_pays(int rows);
bool bn_cols_layout_ok(const void* p, int ld);
int bn_fwd_cols(hipStream_t stream, const float* a, int lda, int rows, int N, const float* beta,
                int relu, float* h, int ldh, float* mean, float* var);
int bn_bwd_cols(hipStream_t stream, const float* dh, int lddh, const float* h, int ldh,
                const float* a, int lda, const float* mean, const float* var, int rows, int N,
                int relu, float* da, int ldda, float* s1, float* s2, float* dbeta,
                float* moving_mean, float* moving_var);
// out[r, :] = [z[r, :L] | extra[r % cells, :E]];  slice: out[r, :L] = in[r, :L] of [rows, ld]
int concat_extra(hipStream_t stream, const float* z, int L, const float* extra, int E, size_t rows,
                 size_t cells, float* out);
int slice_cols(hipStream_t stream, const float* in, int ld, int L, size_t rows, float* out);
int relu_bwd(hipStream_t stream, const float* dh, const float* h, float* da, size_t n);
int col_sum(hipStream_t stream, const float* a, int lda, int rows, int N, float* out, float scale,
            int accumulate, float* partial);

// ---- midchain.hip: the hidden layers, posterior heads and latent stage of a small VAE step
//      in one workgroup (two launches instead of ~27) ----
struct MidLayer {
  const float* W;        // [n_in, n_out]
  const float* b;        // [n_out]
  const float* beta;     // [n_out]
  float* mov_mean;       // [n_out]
  float* mov_var;
  float* a;              // [rows, n_out] pre-normalisation output
  float* h;              // [rows, n_out] layer output
  float* stats;          // [mean | var | s1 | s2]
  float* dW;
  float* db;
  float* dbeta;
  int n_in, n_out;
};
constexpr int MID_MAX_LAYERS = 8;
struct MidChainArgs {
  int cells, samples, latent, n_enc, n_dec, training, deterministic;
  float kl_coeff;                       // d(-ELBO_w) / d KL_cell
  MidLayer enc[MID_MAX_LAYERS], dec[MID_MAX_LAYERS], mu, ls;
  const float* eps;
  float *mu_pre, *ls_pre, *z, *kl_elem, *kl_cell;
  float *dz, *dmu, *dls;
  float* da0;                           // out: gradient w.r.t. the input layer's pre-activation
  float* buf[3];                        // [rows, <=128] scratch; buf[0] holds dd on entry (backward)
  unsigned* bar;                        // grid-barrier counter (never reset)
  unsigned bar_base;                    // its value when this launch starts
};
unsigned vae_mid_barrier_advance(const MidChainArgs& args, bool backward);
bool vae_mid_chain_resident();   // all workgroups of the two kernels co-resident on this device
int vae_mid_forward(hipStream_t stream, const MidChainArgs& args);
int vae_mid_backward(hipStream_t stream, const MidChainArgs& args);

// ---- tilechain.hip: the hidden layers of a large training minibatch, one launch per layer and
//      direction (64-row tiles; batch-norm statistics merged by the consuming kernel) ----
struct TileBN {
  const float* a = nullptr;   // [rows, n] pre-normalisation output of the layer (nullptr: none)
  float* h = nullptr;         // [rows, n] normalised (+ relu) output
  const float* beta = nullptr;
  float* mean = nullptr;      // [n] batch statistics of the forward pass
  float* var = nullptr;
  const float* part = nullptr;   // forward: chunk (mean, M2); backward: chunk (sum dxh, sum dxh xh)
  int chunks = 0, chunk = 0;
  // > 0: the rows are `groups` consecutive groups of group_tiles 64-row tiles, each normalised by
  // its OWN batch statistics (the GMVAE's K passes through shared weights, gm:2859-2922): mean,
  // var, s1, s2 are [groups][n]; dbeta sums the groups, the moving averages take the groups'
  // statistics one after the other (as K executions of the layer do); chunk must be 64
  int group_tiles = 0, groups = 1;
  float* part_out = nullptr;  // backward epilogue: where the chunk sums of THIS layer go
  float* s1 = nullptr;        // backward: merged sums, dbeta, moving statistics (tile 0 writes)
  float* s2 = nullptr;
  float* dbeta = nullptr;
  float* mov_mean = nullptr;
  float* mov_var = nullptr;
};

struct TileFwdArgs {
  int rows = 0, K = 0;
  const float* x = nullptr;   // plain input [rows, K] (pitch ldx) when bn.a == nullptr
  int ldx = 0;
  TileBN bn;                  // the layer below, normalisation pending
  struct Out {
    const float* W = nullptr;   // [K, N]
    const float* b = nullptr;
    float* out = nullptr;       // [rows, N]
    float* part = nullptr;      // chunk statistics of `out` (one chunk per tile), or nullptr
    int N = 0;
  } o[2];
  int n_out = 0;
};

struct TileBwdArgs {
  int rows = 0;
  float inv_count = 0.f, bessel = 1.f;
  // the gradients arriving from above: for a batch-normalised layer ONE, w.r.t. its output h
  // (bn.a != nullptr: dA is formed here); for plain layers (the posterior heads) one or two,
  // w.r.t. their pre-activations
  int n_up = 0;
  struct Up {
    const float* g = nullptr;   // [rows, N]
    const float* W = nullptr;   // [K, N]
    float* dW_slab = nullptr;   // [G][K][N]
    float* db_slab = nullptr;   // [G][N] (plain layers; nullptr: no bias gradient)
    float* dA_out = nullptr;    // [rows, N]: dA written out (the layer that sees x: its dW is the
                                // count kernels' job), nullptr otherwise
    int N = 0;
  } up[2];
  TileBN bn;                  // this layer's batch norm (n_up == 1)
  const float* in = nullptr;  // [rows, K] the layer's input (nullptr: no dW / d_in here)
  int K = 0;
  float* d_in = nullptr;      // [rows, K] gradient w.r.t. the input (nullptr: not needed)
  TileBN below;               // the batch-normalised layer that produced `in` (a != nullptr):
                              // its chunk sums are formed from d_in (written to below.part_out)
};

// out[i] = sum_g slabs[g][i] (g < G), fixed order; up to eight (slabs, n, G, out) jobs in one
// launch: the weight-gradient slabs of a backward pass are independent of the chain of layers, so
// they are summed together at its end
constexpr int TC_MAX_JOBS = 10;
struct SlabJobs {
  int n_jobs = 0;
  struct Job { const float* slabs; float* out; int n; int G; } job[TC_MAX_JOBS];
};
size_t tile_chain_part_floats(int rows);     // chunk statistics / sums of one layer
size_t tile_chain_slab_floats(int rows);     // dW (+ db) slabs of one weight matrix
int tile_forward(hipStream_t s, const TileFwdArgs& q);
int tile_backward(hipStream_t s, const TileBwdArgs& q);
int tile_backward_stats(hipStream_t s, const float* dh, const TileBN& bn, int rows, int N);
int tile_slab_reduce(hipStream_t s, const SlabJobs& q);
// data-parallel steps: a layer's chunk statistics (forward) / chunk sums (backward) of this rank
// merged into the layer's buffers, for the caller's hook to merge over the ranks; the tile kernels
// take them as given where TileBN::part == nullptr
// (groups > 1: `chunks` chunks and `rows` rows PER GROUP, statistics [groups][N] -- the GMVAE's
//  K passes)
int tile_stats_merge(hipStream_t s, const float* part, int chunks, int chunk, int rows, int N,
                     float* mean, float* var, int groups = 1);
int tile_sums_merge(hipStream_t s, const float* part, int chunks, int N, const TileBN& bn,
                    float bessel, int groups = 1);

// ---- the resident chain (tilechain.hip): the stages of a whole pass in one launch ----
constexpr int TCR_MAX_STAGES = 12;   // stages of a launch
constexpr int TCR_MAX_TILES = 7;     // tile stages (TileFwdArgs / TileBwdArgs) among them
enum { TCS_TILE = 0, TCS_LATENT = 1, TCS_STATS = 2, TCS_REDUCE = 3 };
struct TileLatent {            // the latent stage between the posterior heads and the decoder
  const float* mu_pre = nullptr;
  const float* ls_pre = nullptr;
  const float* eps = nullptr;  // [S, B, L]
  float* z = nullptr;          // forward: [S, B, L]
  float* kl_elem = nullptr;    // [B, L]
  float* kl_cell = nullptr;    // [B]
  const float* dz = nullptr;   // backward: [S, B, L]
  float* dmu = nullptr;
  float* dls = nullptr;
  float kl_coeff = 0.f;
  int S = 1, B = 0, L = 0;
};
struct TileChainFwdArgs {
  unsigned* bar = nullptr;     // grid-barrier counter and its value when this launch starts
  unsigned bar_base = 0;
  int n = 0;                   // stages
  int kind[TCR_MAX_STAGES] = {};   // TCS_TILE (f[idx]) or TCS_LATENT
  int idx[TCR_MAX_STAGES] = {};
  int sync[TCR_MAX_STAGES] = {};   // after stage i: 1 workgroup barrier, 2 grid barrier (chunk
                                   // statistics cross), 3 grid barrier with release / acquire
  TileFwdArgs f[TCR_MAX_TILES];
  TileLatent lat;
};
struct TileChainBwdArgs {
  unsigned* bar = nullptr;
  unsigned bar_base = 0;
  int n = 0;
  int kind[TCR_MAX_STAGES] = {};   // TCS_TILE (b[idx]), TCS_LATENT, TCS_STATS, TCS_REDUCE
  int idx[TCR_MAX_STAGES] = {};
  int sync[TCR_MAX_STAGES] = {};
  TileBwdArgs b[TCR_MAX_TILES];
  const float* stats_dh = nullptr;   // TCS_STATS: tile_backward_stats(stats_dh, stats_bn, ...)
  TileBN stats_bn;
  int stats_rows = 0, stats_N = 0;
  TileLatent lat;
  SlabJobs jobs;                     // TCS_REDUCE
};
int tile_chain_resident_capacity();   // workgroups the device holds at once (0: unknown)
// one launch; *advance: what the barrier counter has gained once it has run
int tile_chain_forward(hipStream_t s, const TileChainFwdArgs& q, int tiles, unsigned* advance);
int tile_chain_backward(hipStream_t s, const TileChainBwdArgs& q, int tiles, unsigned* advance);

struct HeadDropout;   // (below, with dropout_apply)
// per-row inputs / second output of the constrained Poisson passes of decoder_head3_kernel
struct CpRows {
  const float* count_sum = nullptr;   // [cells]: N
  const float* lse = nullptr;         // [rows]: log-sum-exp of the row's logits (passes 2, 3)
  const float* S = nullptr;           // [rows]: sum_f gate_f (t_f - N lambda_f) (pass 3)
  float* out2 = nullptr;              // [strips][rows]: pass 1 sum of exponentials, pass 2 part of S
};
// ---- decoder_fused.hip ----
// where a fused likelihood kernel reads its targets t[row % B, gene] from: fp32 [B, F] (pitch F)
// or the uint16 minibatch of scvae_csr_densify_u16 (pitch ld; integer counts convert exactly)
struct Targets {
  const void* p;
  int ld;
  int u16;
  // (the count part of the piecewise categorical likelihood, decoder_head3_kernel only) the count
  // distribution sees t - shift where t >= shift and nothing -- no term, no gradient -- elsewhere
  float shift;
};
inline Targets targets_f32(const float* t, int F) { return Targets{t, F, 0, 0.f}; }
inline Targets targets_u16(const uint16_t* t, int ld) { return Targets{t, ld, 1, 0.f}; }
#ifdef __HIPCC__
// a target as loaded (kept raw while the load is in flight: no instruction touches it) and as
// the fp32 value the likelihood uses
__device__ __forceinline__ float target_raw(float v) { return v; }
__device__ __forceinline__ float target_raw(uint16_t v) { return __uint_as_float((unsigned)v); }
__device__ __forceinline__ float target_value(float raw, int u16) {
  return u16 ? (float)__float_as_uint(raw) : raw;
}
#endif
struct HeadParams {
  const float* W[3];
  const float* b[3];
  float* dW[3];
  float* db[3];
  // (decoder_head3_kernel only; 0: the plain [H, F] / [F] layout) element stride between the
  // genes of a head and pitch of its weight rows: the k + 1 class logits of the P_K head are
  // columns c, c + (k + 1), ... of one [H, F (k + 1)] matrix (va:2507-2518)
  int gene_stride = 0;
  int row_pitch = 0;
};
bool decoder_fused_supported(int H);                       // every fused kernel: even H <= 126
// a TRAINING launch of the fused heads: the above, or (bf16x9) the producer / consumer kernel's
// wider range -- H <= 256 (one / two heads), <= 159 (three), odd H included
bool decoder_fused_train_supported(int P, int H, int arith);
size_t decoder_fused_workspace_floats(int rows, int H, int F, bool train);
// the piecewise categorical likelihood -k (k_max = 1, 2; Poisson / negative-binomial counts) as
// two launches of the bf16x9 all-in-one-phase kernel: Wk / bk / dWk / dbk the P_K head
// [H, F (k_max + 1)] / [F (k_max + 1)]; t fp32 [B, F]; scratch: rows * (H + 1) + 64 floats
bool decoder_fused_cat_supported(int kind, int k_max, int H, int arith);
// floats of the `scratch` argument of decoder_fused_train_cat / decoder_fused_forward_cat (ll and dd
// of the second launch: the plans lend the unfused path's logits buffer and size it for both)
size_t decoder_fused_cat_scratch_floats(int rows, int H);
// (forward only: two launches of decoder_forward_kernel; scratch: rows floats)
bool decoder_fused_forward_cat_supported(int kind, int k_max, int H);
int decoder_fused_forward_cat(hipStream_t s, int kind, int k_max, const float* d, int rows, int H,
                              HeadParams hp, const float* Wk, const float* bk, int F,
                              const float* t, int B, float* ll, float* workspace, float* scratch);
int decoder_fused_train_cat(hipStream_t s, int kind, int k_max, const float* d, int rows, int H,
                            HeadParams hp, const float* Wk, const float* bk, float* dWk,
                            float* dbk, int F, const float* t, int B, const float* gw, float* ll,
                            float* dd, float* workspace, int arith, float* scratch);
size_t decoder_fused_lds_bytes(int P, int H, bool train);
int decoder_fused_variant(int P, int H);   // 1: decoder_head_kernel, 2: decoder_head2_kernel
bool decoder_fused2_supported(int P, int H);
size_t decoder_fused2_lds_bytes(int P, int H);
int decoder_fused2_launch(hipStream_t s, bool train, int kind, const float* d, int rows, int H,
                          HeadParams hp, int F, Targets t, int B, const float* gw,
                          int inline_lgamma, float* ll_part, float* dd_part);
// arith: 0 fp32 MFMA, 1 the exact nine-term bf16 split where decoder_fused3 applies, 2 = 1 with
// six-term products in the producer / consumer training kernel
int decoder_fused_forward(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                          int F, Targets t, int B, const float* row_const, float* ll,
                          float* workspace, int arith);
int decoder_fused_train(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                        int F, Targets t, int B, const float* gw, const float* row_const,
                        float* ll, float* dd, float* workspace, int arith,
                        bool kernel_only = false,
                        const HeadDropout* drop = nullptr,    // drop: bf16x9 kernel only
                        int dd_mode = 0);                     // 1: XCD-local atomics for dd

// training kernel on the bf16 matrix cores, exact nine-term split (decoder_fused3.hip)
bool decoder_fused3_supported(int P, int H);
size_t decoder_fused3_lds_bytes(int P, int H);
int d4_strip_genes(int P, int H);      // ... of the producer / consumer kernel (any row count)
int decoder_fused3_strip_genes(int P);   // genes per workgroup (= per slab of ll_part / dd_part)
int decoder_fused3_train_kernel_name(int kind, int H, int rows, bool u16, char* out, size_t n,
                                     int terms = 9);
size_t decoder_fused3_workspace_floats(int rows, int H);
// the producer / consumer training kernel (decoder_head4_kernel): any H up to 256 for one and two
// heads, up to 159 for three (LDS), odd H included; plain training launches only
bool decoder_fused4_supported(int P, int H);
// genes per workgroup (= per slab of ll_part / dd_part) of a TRAINING launch
int decoder_fused3_train_strip_genes(int P, int H, int rows, bool drop, int cp_pass);
// (train = false: the forward half alone, one- and two-head likelihoods; gw / dd_part unused)
// drop (training only): dropout of the heads' input connections inside the kernel
int decoder_fused3_launch(hipStream_t s, bool train, int kind, const float* d, int rows, int H,
                          HeadParams hp, int F, Targets t, int B, const float* gw,
                          int inline_lgamma, float* ll_part, float* dd_part, float* planes,
                          const HeadDropout* drop = nullptr, int cp_pass = 0,
                          const CpRows* cp = nullptr, int dd_mode = 0, float* rg_slab = nullptr);
// rg_slab (decoder_fused3_rg_slab_floats(H, F) floats, or nullptr: one row group): where the
// producer / consumer kernel's row groups behind the first leave their dW / db (decoder_fused3.hip,
// "row groups")
size_t decoder_fused3_rg_slab_floats(int H, int F);
// dd_mode 1: the per-strip partials of dd are not written as slabs but added (fp32 atomics, not
// bit-repeatable) into eight XCD-local [H][rows] accumulators at dd_part; only where
// decoder_fused3_dd_atomics says so (the producer / consumer training kernel)
bool decoder_fused3_dd_atomics(int kind, int H, int rows, bool drop, int cp_pass, int dd_mode);
// Constrained Poisson (du:218-228) through the bf16x9 head kernel in three passes over the strip
// grid (row maximum / sum of exponentials | log-likelihood and S | gradients): ll[rows] and, with
// train, dW / db (in hp) and dd[rows, H].  workspace: decoder_fused_workspace_floats(.., true).
bool decoder_fused_cpoisson_supported(int H, int arith);
int decoder_fused_cpoisson(hipStream_t s, bool train, const float* d, int rows, int H,
                           HeadParams hp, int F, Targets t, int B, const float* gw,
                           const float* count_sum, const float* row_const, float* ll, float* dd,
                           float* workspace);
// Measurement aid (scvae_plan_probe_stages): HIP event pairs around the HBM-bound stages of the
// step this host thread launches next.  stage_probe(stage, 0 / 1, stream) records the begin / end
// event of an armed table and is a no-op otherwise.
enum ProbeStage : int {
  PS_FETCH = 0,      // next minibatch: CSR rows -> dense (uint16) batch (+ its noise)
  PS_COUNT_FWD = 1,  // x W1 + b on the count kernels (split + kernel + reduce)
  PS_COUNT_DW = 2,   // x^T dA
  PS_DD_REDUCE = 3,  // reduce of the decoder gradient's per-strip (or per-XCD) partials
  PS_ADAM = 4,       // clip + Adam over the whole parameter buffer
  PS_COUNT = 5
};
void stage_probe_arm(hipEvent_t* events /* [PS_COUNT][2] or nullptr */, unsigned* recorded);
void stage_probe(int stage, int which, hipStream_t s);
void decoder_fused_set_probe(hipEvent_t before, hipEvent_t after);   // (nullptr, nullptr): off
hipEvent_t decoder_fused_probe(int which);
bool decoder_fused_probe_recorded();   // both events of the pair went into a stream
int default_dd_atomics();   // SCVAE_DD_ACCUMULATION, read once: 1 (default) atomics, 0 slabs
int default_head_arith();   // SCVAE_HEAD_ARITH, read once: 0 fp32 MFMA, 1 (default) bf16x9, 2 bf16x6
int decoder_train_kernel(int P, int H, int arith);   // 1 / 2: the fp32 schedules, 3: decoder_fused3.hip

// forward-only variant with the pre-activations in registers (decoder_forward.hip)
bool decoder_forward_supported(int P, int H);
int decoder_forward_launch(hipStream_t s, int kind, const float* d, int rows, int H, HeadParams hp,
                           int F, Targets t, int B, int inline_lgamma, float* ll_part);

// ---- gmvae_kernels.hip ----
int add_group_rows(hipStream_t s, const float* a0, const float* rows, float* out, int K, int B,
                   int N, int relu);
int group_col_sum(hipStream_t s, const float* a, int lda, int R, int G, int N, float scale,
                  float* out, float* partial);
int sum_groups(hipStream_t s, const float* a, const float* w, int ldw, int G, int R, int N,
               float* out);
// `prior_logits` (K floats or NULL = uniform p(y)): kl_y_cell[b] = KL(q(y|x_b) || p(y))
// (gm:3242-3258: log K - H[q] for the uniform prior, tfp kl_divergence otherwise)
int categorical_fwd(hipStream_t s, const float* logits, float* y, float* kl_y_cell, int B, int K,
                    const float* prior_logits = nullptr);
// d/d prior_logits of w * max(mean_b KL_y, free_nats * H[p]) (learned p(y), gm:2799-2803):
// gate on:  c * sum_b (p_j - q_bj);  gate off: off_scale * free_nats * dH[p]/dm_j
int prior_logits_bwd(hipStream_t s, const float* y, const float* prior_logits, const float* gate,
                     float c, float off_scale, float free_nats, int B, int K, float* dprior);
// (with `prior_logits`: the KL term's gradient is q_j (log q_j - log p_j - KL_b))
int categorical_bwd_gated(hipStream_t s, const float* y, const float* dy, const float* gate,
                          float c, float* dlogits, int B, int K,
                          const float* prior_logits = nullptr);
int softplus_gaussian_fwd(hipStream_t st, const float* qm, const float* qs, const float* Wpm,
                          const float* bpm, const float* Wps, const float* bps, const float* eps,
                          float* z, float* klz, float* qvar, int K, int S, int B, int L);
int softplus_gaussian_bwd(hipStream_t st, const float* qm, const float* qs, const float* Wpm,
                          const float* bpm, const float* Wps, const float* bps, const float* eps,
                          const float* dz, const float* gklz, float* dqm, float* dqs, float* dpr,
                          int K, int S, int B, int L);
int gmvae_elbo(hipStream_t s, const float* ll, const float* klz, const float* y,
               const float* kl_y_cell, int K, int S, int B, float inv_gb, float* sums,
               float* rec_cell);
// thr = free_nats * H[p(y)]: `thr` as given for the uniform prior, computed on the device from
// `prior_logits` (K floats) otherwise
int gmvae_elbo_finish(hipStream_t s, const float* sums, float w, float thr, int use_free_nats,
                      float share, float* scalars, float* gate,
                      const float* prior_logits = nullptr, int K = 0, float free_nats = 0.f);
int gmvae_elbo_bwd(hipStream_t s, const float* ll, const float* klz, const float* y,
                   const float* gate, int K, int S, int B, float w, float inv_gb, float* gw,
                   float* gklz, float* dy);
int prior_stats(hipStream_t s, const float* Wpm, const float* bpm, const float* Wps,
                const float* bps, int K, int L, float* means, float* variances);

// clip-by-value(+-1) and TF Adam on a flat parameter buffer (va:2742-2759)
int adam_clip_step(hipStream_t stream, float* theta, float* grad, float* m, float* v, size_t n,
                   float grad_scale, float lr_t, float beta1, float beta2, float epsilon);

// the noise of the same step, drawn by trailing workgroups of the minibatch launch (the arguments
// of philox_normal below; a step that carries both for the next one: scvae_side_work)
struct NoiseRequest {
  float* out = nullptr;
  int64_t rows = 0;
  int cols = 0;
  int64_t row_offset = 0;
  uint64_t seed = 0, stream_id = 0;
  int64_t block_rows = 0, block_stride = 0;
};
// CSR row gather + densify (va:985-998)
int csr_densify(hipStream_t stream, const int64_t* indptr, const int32_t* indices,
                const float* values, const int64_t* rows, int B, int F, float* out, int ldo,
                const float* row_values = nullptr, float* row_values_out = nullptr,
                const NoiseRequest* noise = nullptr);
bool csr_densify_u16_supported(int F, int ldo);
int csr_densify_u16(hipStream_t stream, const int64_t* indptr, const int32_t* indices,
                    const float* values, const int64_t* rows, int B, int F, uint16_t* out,
                    int ldo, const float* row_values = nullptr, float* row_values_out = nullptr,
                    const NoiseRequest* noise = nullptr);
int csr_row_lgamma1p(hipStream_t stream, const int64_t* indptr, const float* values, int64_t n_rows,
                     float* out);
int gather_rows_f32(hipStream_t stream, const float* src, const int64_t* rows, int B, float* out);

// counter-based standard-normal draws (Philox4x32-10 + Box-Muller), keyed by
// (seed, stream id, global row, column): identical for any sharding of the rows
int philox_normal(hipStream_t stream, float* out, int64_t rows, int cols, int64_t row_offset,
                  uint64_t seed, uint64_t stream_id, int64_t block_rows = 0,
                  int64_t block_stride = 0);
// Local row -> row of the global minibatch (data parallel).  A buffer of `rows` rows stacks
// Evaluation steps of the VAE: everything between the input layer's product and the likelihood
// heads for 16 cells per workgroup, one launch (tilechain.hip: eval_mlp_kernel).  With
// is_training = False a batch-normalised layer uses its moving statistics (mu:60-70), so a cell's
// path through the hidden layers, the posterior heads, the reparameterised sample (one sample per
// cell) and the decoder's layers needs no other cell.
constexpr int EM_MAX_OPS = 10;
enum { EM_HIDDEN = 0, EM_MU = 1, EM_LOG_SIGMA = 2 };
struct EvalMlpArgs {
  int rows = 0, n_ops = 0;
  const float* a0 = nullptr;      // [rows, K0] pre-normalisation output of the input layer
  float* h0 = nullptr;            // [rows, K0] its normalised output (written)
  int K0 = 0;
  const float* mean0 = nullptr;   // its moving statistics, its beta
  const float* var0 = nullptr;
  const float* beta0 = nullptr;
  struct Op {
    const float* W = nullptr;     // [K, N]
    const float* b = nullptr;
    const float* mean = nullptr;  // EM_HIDDEN: the layer's moving statistics, its beta
    const float* var = nullptr;
    const float* beta = nullptr;
    float* pre = nullptr;         // [rows, N] x W + b (EM_HIDDEN: may be nullptr)
    float* out = nullptr;         // EM_HIDDEN: [rows, N] normalised + relu
    int K = 0, N = 0, kind = EM_HIDDEN;
  } op[EM_MAX_OPS];
  // the latent stage, behind the EM_LOG_SIGMA op (gauss_latent_fwd for one sample per cell)
  const float* eps = nullptr;     // [rows, L]; nullptr: z = mu (deterministic_z)
  float* z = nullptr;             // [rows, L]
  float* kl_elem = nullptr;       // [rows, L]
  float* kl_cell = nullptr;       // [rows]
  int L = 0;
};
int eval_mlp(hipStream_t stream, const EvalMlpArgs& q);

// rows / cells passes (samples, GMVAE clusters) of this rank's `cells` cells, which are cells
// offset .. offset + cells - 1 of the global_cells cells of the step: local row p*cells + b is
// global row p*global_cells + offset + b.  cells == 0: identity.
struct RowMap {
  int64_t cells = 0, global_cells = 0, offset = 0;
};
// Dropout of the likelihood heads' input connections in a training step (mu:45-50 inside each
// X_TILDE dense_layer, va:2475-2488): every head has a mask of its own.
struct HeadDropout {
  const float* d[3] = {nullptr, nullptr, nullptr};   // the heads' dropped-out copies of d [rows, H]
  float keep = 1.f;
  uint64_t seed = 0;
  uint32_t site[3] = {0, 0, 0};                      // mask streams of the heads (dropout_apply)
  RowMap map;
};
// the mask of dropout_apply(seed, site) as bits: words[row][4], bit c % 32 of word c / 32 set iff
// element (row, c) is kept; rows [rows, rows_pad) zero; cols <= 128
int dropout_mask_words(hipStream_t stream, uint32_t* words, int rows, int rows_pad, int cols,
                       float keep, uint64_t seed, uint32_t site, RowMap map = RowMap());
// dropout (mu:45-50): out (+)= in * mask(seed, site, global row, col) / keep; forward and backward
int dropout_apply(hipStream_t stream, const float* in, int ld_in, float* out, int ld_out,
                  int64_t rows, int cols, float keep, uint64_t seed, uint32_t site,
                  int accumulate, RowMap map = RowMap());
// out[k, :] = in[k, :] * mask(seed, site, row k, column k) / keep (dropout of a one-hot input)
int dropout_scale_rows(hipStream_t stream, const float* in, float* out, int K, int N, float keep,
                       uint64_t seed, uint32_t site);
// out[k*B + b, :] = [x[b, :] | one_hot(k)]  ([K*B, F + K])
int tile_onehot(hipStream_t stream, const float* x, float* out, int K, int B, int F);

}  // namespace scvae
