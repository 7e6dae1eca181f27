/* C ABI of libscvae_hip.so -- the MI355X (gfx950) execution engine underneath
 * the Python model classes of scvae_amd (the drop-in for scvae.models).
 *
 * The reference (scvae/scvae v2.1.4) has no FFI: its hot path is a TensorFlow
 * graph executed by session.run().  Each entry point below names the piece of
 * the reference graph / loop it replaces (paths relative to /root/reference).
 *
 * Conventions:
 *  - plain pointers and sizes only; every buffer is a DEVICE pointer owned by
 *    the caller (the library never allocates device memory), row-major fp32
 *    unless stated otherwise;
 *  - asynchronous on the given hipStream_t (as void*), no internal sync (the set-up calls
 *    without a stream argument -- scvae_plan_create / _bind -- excepted);
 *  - return 0 on success, -1 bad argument, -2 HIP error; the text is returned
 *    by scvae_last_error() (per host thread);
 *  - one host thread per plan.
 */
#ifndef SCVAE_HIP_H
#define SCVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCVAE_MAX_HIDDEN 8
#define SCVAE_NAME_MAX 96

/* likelihood kinds; head order = parameter order of the DISTRIBUTIONS registry
 * (scvae/distributions/utilities.py:206-305) */
enum {
  SCVAE_POISSON = 0, /* log_lambda            */
  SCVAE_NB = 1,      /* p, log_r              */
  SCVAE_ZIP = 2,     /* pi, log_lambda        */
  SCVAE_ZINB = 3,    /* pi, p, log_r          */
  SCVAE_CONSTRAINED_POISSON = 4, /* lambda: softmax over the genes, rate = lambda * count sum of
                        the cell (du:218-228, va:2400-2405, 2490-2496); needs
                        scvae_step_args.count_sum; heads and likelihood run unfused */
  SCVAE_BERNOULLI = 5 /* logits (du:194-204), targets binarised by the caller; unfused */
};

enum { SCVAE_MODEL_VAE = 0, SCVAE_MODEL_GMVAE = 1 };

/* Constructor arguments of VariationalAutoencoder /
 * GaussianMixtureVariationalAutoencoder that shape the graph
 * (scvae/models/variational_autoencoder.py:114-292,
 *  scvae/models/gaussian_mixture_variational_autoencoder.py:136-326). */
typedef struct scvae_model_config {
  int32_t model_type;         /* SCVAE_MODEL_* */
  int32_t feature_size;       /* F */
  int32_t latent_size;        /* L */
  int32_t n_hidden;           /* len(hidden_sizes) */
  int32_t hidden[SCVAE_MAX_HIDDEN];
  int32_t likelihood;         /* SCVAE_POISSON ... */
  int32_t batch_norm;         /* minibatch_normalisation */
  int32_t n_clusters;         /* K (GMVAE), 1 for the VAE */
  float kl_weight;
  float free_nats_proportion; /* proportion_of_free_nats_for_y_kl_divergence */
  int32_t k_max;              /* piecewise categorical likelihood (-k, va:2507-2532,
                                 distributions/categorised.py): counts below k_max are classes
                                 of a categorical head P_K [F * (k_max + 1)]; 0 = off.  Poisson
                                 and negative binomial only */
  int32_t prior_mode;         /* GMVAE p(y), gm:2794-2808 (prior_probabilities_method): 0 uniform,
                                 1 custom (fixed logits in a hidden slot of the parameter buffer,
                                 see scvae_plan_prior_offset), 2 learn (trainable Y/P/LOGITS) */
  int32_t linear_factor;      /* VAE: bit 0 inference_architecture == "LFM" (posterior heads
                                 directly on x, va:2233-2234), bit 1 generative_architecture ==
                                 "LFM" (likelihood heads directly on z, va:2456-2457): the hidden
                                 layers of that side are not built */
  int32_t decoder_extra;      /* E: extra decoder input columns appended to z -- one-hot batch
                                 indices (batch_correction) and/or the normalised count sum
                                 (use_count_sum_as_feature), va:2407-2441, gm:3094-3130 */
  int32_t latent_mode;        /* VAE: bit 0 Monte-Carlo KL term, log q(z|x) - log p(z) at the
                                 samples (analytical_kl_term == False, va:2633-2640; the GMVAE's
                                 KL(z) is always of that form); bit 1 latent_distribution ==
                                 "unit-variance gaussian" (du:323-337): the posterior's log_sigma
                                 is the constant 0 and POSTERIOR/LOG_SIGMA is not built.
                                 GMVAE: bit 2 (value 4) "legacy gaussian mixture" (du:349-352):
                                 the z layers live in scope MODIFIED_GAUSSIAN instead of
                                 SOFTPLUS_GAUSSIAN, same graph */
  float dropout_keep[4];      /* dropout_keep_probabilities (va:245-269, gm:281-301), applied to
                                 the input connections of a dense layer while training
                                 (mu:45-50): [0] h: hidden layers and every parameter head,
                                 [1] x: first encoder layer, [2] z: first decoder layer,
                                 [3] y: GMVAE p(z|y) layers.  0 or 1: no dropout */
} scvae_model_config;

typedef struct scvae_plan scvae_plan; /* opaque */

/* Collective hook used by data-parallel training (one process per GPU).
 * kind 0: all-reduce(sum) `count` floats in place at `buf`;
 * kind 1: merge batch-norm statistics: buf = [mean(n) | var(n)] of `local_rows`
 *         rows -> global mean/var over all ranks, in place (count = 2n);
 * kind 2: notification: the `count` gradient floats at `buf` (a range of the
 *         bound gradient buffer) are final although the step is still running;
 *         the caller may start their all-reduce(sum) asynchronously so that it
 *         overlaps the rest of the backward pass, and must complete it (and
 *         reduce the rest of the buffer) before the optimiser step.  The VAE step
 *         announces the likelihood heads right after their kernel and the hidden
 *         layers when it reaches the first encoder layer.  May be ignored
 *         (return 0).
 * Kinds 0 and 1 must enqueue on the plan's stream.  NULL = single process. */
typedef int (*scvae_sync_fn)(void* user, float* buf, int64_t count, int32_t kind,
                             int64_t local_rows);

const char* scvae_last_error(void);
int scvae_version(void);

/* ---- graph build: _setup_model_graph (va:2219-2558, gm:2788-3221) ---- */
int scvae_plan_create(const scvae_model_config* cfg, scvae_plan** out);
void scvae_plan_destroy(scvae_plan* plan);
/* trainable variables in reference creation order */
int64_t scvae_plan_param_count(const scvae_plan* plan);
int64_t scvae_plan_param_floats(const scvae_plan* plan);   /* flat buffer length  */
int64_t scvae_plan_moving_floats(const scvae_plan* plan);  /* BN moving mean/var  */
int scvae_plan_param_info(const scvae_plan* plan, int64_t index, char* name /*SCVAE_NAME_MAX*/,
                          int64_t* offset, int64_t* rows, int64_t* cols);
int64_t scvae_plan_moving_count(const scvae_plan* plan);
/* float offset of the K prior logits of p(y) in the parameter buffer (prior_mode 1: the caller
 * writes log(prior_probabilities) there once, the step never changes them; prior_mode 2: the
 * trainable Y/P/LOGITS); -1 for the uniform prior */
int64_t scvae_plan_prior_offset(const scvae_plan* plan);
int scvae_plan_moving_info(const scvae_plan* plan, int64_t index, char* name, int64_t* offset,
                           int64_t* size);
int64_t scvae_plan_workspace_bytes(const scvae_plan* plan, int64_t max_cells, int64_t max_samples);
/* Set-up call (once per plan and workspace size), NOT on the step path: it has no stream argument
 * and clears a few words of the gradient buffer and the workspace with blocking hipMemset calls
 * (the gradient slots of the batch-normalised layers' biases, which no kernel ever writes, and the
 * barrier counter of the mid-chain kernels), i.e. it synchronises with the device. */
int scvae_plan_bind(scvae_plan* plan, float* params, float* grads, float* moving, void* workspace,
                    int64_t workspace_bytes, int64_t max_cells, int64_t max_samples);
int scvae_plan_set_sync(scvae_plan* plan, scvae_sync_fn fn, void* user);
/* 1 (default): the X_TILDE heads + likelihood + their backward run as one fused kernel;
 * 0: separate GEMM and likelihood kernels (same results; kept for A/B tests and for the
 * evaluate-time statistics, which need the materialised pre-activations) */
int scvae_plan_set_fused(scvae_plan* plan, int32_t enabled);
/* 1 if the steps of this (bound) plan run the piecewise categorical
 * likelihood `-k` (cfg.k_max = 1 or 2, Poisson / negative-binomial counts; categorised.py:255-263,
 * va:2507-2532) on the fused kernels: two launches of the bf16x9 head kernel -- the count heads on
 * shifted, masked targets and the k + 1 class logits of every gene as the heads of a categorical
 * kind -- instead of materialised [rows, (P + k + 1) F] pre-activations; evaluation passes the
 * same two terms on the forward kernel (also the first pass of an importance-weighted
 * training step).  Larger k, head dropout, evaluate-time statistics: the unfused kernels, as
 * before. */
int32_t scvae_plan_fused_categorised(const scvae_plan* plan);
/* Arithmetic of this plan's fused head kernels (see scvae_default_head_arith below): 0 fp32
 * matrix cores, 1 the exact nine-term bf16 split, 2 the six-term split (the nine terms without
 * a2 b3, a3 b2, a3 b3, together <= 2^-26 of a product on the rounded split: fp32-class, not exact; only the
 * producer / consumer training kernel has it -- H <= 126, more than 128 rows, no head dropout --
 * every other launch runs as under 1).  Affects which kernels the step launches and
 * what scvae_plan_accepts_counts_u16 answers; call it before the first step. */
int scvae_plan_set_head_arith(scvae_plan* plan, int32_t mode);
int32_t scvae_plan_head_arith(const scvae_plan* plan);
/* 1 (the default of a new plan, scvae_default_dd_atomics): this plan's training steps accumulate
 * the decoder gradient dd with XCD-local fp32 atomics where the head kernel has that store
 * (SCVAE_HEADS_DD_ATOMICS below; the producer / consumer kernel, more than 128 rows) -- the order
 * of the additions, hence the last bits of dd, differ from run to run, as the reference's
 * multi-threaded TensorFlow reductions do; 0: per-strip slabs and a fixed-order reduce,
 * bit-repeatable (+ 0.1 ms per 4096-row step).  SCVAE_DD_ACCUMULATION=slabs in the environment
 * (read once) makes 0 the default; `scvae train --deterministic` / `train(deterministic=True)`
 * set it per model. */
int scvae_plan_set_dd_atomics(scvae_plan* plan, int32_t enabled);
int32_t scvae_plan_dd_atomics(const scvae_plan* plan);
int32_t scvae_default_dd_atomics(void);
/* The exact bf16-split kernels for products with a count matrix: 1 (default) where they pay
 * (minibatches from a few hundred cells upwards, see plan_gemm), 2 always, 0 never -- those
 * products then take the fp32 MFMA kernels even when scvae_step_args.x_counts is set (A/B
 * measurements, the parity test between the two) */
int scvae_plan_set_count_gemm(scvae_plan* plan, int32_t enabled);
/* One-launch batch norm (statistics + normalise, backward sums + gradient in a single
 * column-parallel kernel each) for single-group layers: 1 (default) for minibatches of up to 1024
 * rows, where it pays; 2 whenever it applies (<= 8192 rows); 0 never -- always the chunked
 * statistics / finalize / apply kernels (the two implementations are compared in
 * tests/test_gpu_vae_step.py) */
int scvae_plan_set_bn_one_launch(scvae_plan* plan, int32_t enabled);
/* Measurement aid (bench.py's `roofline`): HIP events around the likelihood-head TRAINING kernel
 * proper -- the dominant kernel of a step; bf16x9 kernel only -- of the next n training steps, on
 * the stream of each step.  scvae_plan_probe_heads(plan, n) arms n pairs (n = 0: off, events
 * released); scvae_plan_probe_heads_ms waits (host) for the pairs recorded so far and writes their
 * elapsed times in milliseconds to out[0 .. n), returning how many (or -1 / -2). */
int scvae_plan_probe_heads(scvae_plan* plan, int32_t n);
int scvae_plan_probe_heads_ms(scvae_plan* plan, float* out, int32_t n);
/* The same for the HBM-bound stages of a training step: SCVAE_PROBE_STAGES event pairs per probed
 * step, in this order: 0 fetch of the next minibatch (scvae_side_work), 1 / 2 the input layer's
 * products x W1 + b and x^T dA on the count kernels (operand split + kernel + reduce), 3 the
 * reduce of the decoder gradient's partial sums, 4 clip + Adam over the parameter buffer.
 * scvae_plan_probe_stages_us writes [steps][SCVAE_PROBE_STAGES] microseconds (-1: that stage
 * did not run in that step) and returns the number of steps recorded. */
#define SCVAE_PROBE_STAGES 5
int scvae_plan_probe_stages(scvae_plan* plan, int32_t n);
int scvae_plan_probe_stages_us(scvae_plan* plan, float* out, int32_t n);
/* Large VAE training minibatches (more than 128 rows, batch norm, no dropout): 1 (default) =
 * every hidden layer and the posterior heads as ONE launch per layer and direction -- a
 * workgroup owns a 64-row tile, merges the batch-norm chunk statistics of the layer below,
 * normalises its rows of it, multiplies (mu:38-76), and leaves the chunk statistics of its own
 * output (tilechain.hip); 0 = the chain of GEMM / statistics / merge / normalise launches.
 * With a data-parallel hook (scvae_plan_set_sync) the rank's chunk statistics are merged by a
 * small kernel, handed to the hook (kinds 1 and 0, as from the launch chain) and taken as given by
 * the consuming tile kernel.  scvae_plan_uses_tile_chain: whether a training step of `cells`
 * cells x `samples` samples of this plan, as configured now, takes that path (VAE plans; GMVAE
 * plans: the K stacked passes as tile-chain groups, whole 64-row tiles per pass). */
int scvae_plan_set_tile_chain(scvae_plan* plan, int32_t enabled);
/* Single process (no scvae_plan_set_sync hook), VAE plans: 1 = the tile chain's
 * stages of a pass -- hidden layers, posterior heads, the latent stage, the dW slab sums -- in
 * ONE resident launch per direction, grid barriers where a stage needs every tile's batch-norm
 * statistics (same tile code, same bits as the per-layer launches; minibatches whose tiles fit
 * half the device's CUs); 0 (default) = one launch per layer and direction.  Measured on
 * MI355X at 4096 rows the resident launches are SLOWER (112 + 144 us against 79 + 128 us of
 * kernel time: the stages are bound by their memory-instruction issue, not by launch gaps, and
 * the resident kernel's first stage pays a longer cold start), so the option is off by default;
 * DESIGN.md section 8. */
int scvae_plan_set_tile_resident(scvae_plan* plan, int32_t enabled);
int32_t scvae_plan_uses_tile_resident(const scvae_plan* plan, int64_t cells, int32_t samples);
int32_t scvae_plan_uses_tile_chain(const scvae_plan* plan, int64_t cells, int32_t samples);
/* Small VAE minibatches (cells x samples <= 128, widths <= 128, batch norm, analytic KL, no
 * dropout / decoder extras, single process): the hidden layers, posterior heads and latent stage
 * of a step run as TWO cooperative launches (forwards, backwards: sixteen workgroups with a grid
 * barrier per layer, instead of ~27 launches; midchain.hip).
 * Default on; 0 keeps the chain of launches (the two are compared in tests/test_gpu_vae_step.py) */
int scvae_plan_set_mid_chain(scvae_plan* plan, int32_t enabled);

/* One graph execution = session.run(...) in the reference loops
 * (train step va:1026-1029 / gm:1094-1097; evaluation va:1124-1135, 1983-2014).
 * x, t: [cells, F]; row_const: [cells] sum_f lgamma(1+t) (or NULL);
 * eps: VAE [S, cells, L], GMVAE [K, S, cells, L] standard normal draws
 * (S = n_iw*n_mc; may be NULL when deterministic_z);
 * global_cells: minibatch size over all data-parallel ranks (= cells if single).
 * training!=0: is_training=True and gradients of -lower_bound_weighted are
 * written to the bound grads buffer (and BN moving statistics are updated).
 * scalars (device, 8 floats): [0] lower_bound [1] lower_bound_weighted
 * [2] reconstruction_error [3] kl_divergence (VAE) / kl_divergence_z (GMVAE)
 * [4] kl_divergence_y (GMVAE) ; each is this rank's share of the global mean.
 * [7] is NOT overwritten but incremented by 1 whenever the execution's lower_bound is not
 * finite: a caller that keeps passing the same buffer gets a sticky counter of non-finite
 * steps.  (The reference tests the loss at the steps it prints, va:1034-1044; with the counter
 * the same test at the same steps also catches a non-finite loss of any step in between.) */
/* ---- the minibatch as tile-indexed non-zeros (optional: scvae_step_args.count_tiles) ----
 * A count minibatch is ~5 % non-zeros (x_train[idx].toarray() of va:997-998 is 95 % zeros): next
 * to its uint16 form the fetch can leave the list of its non-zeros, grouped so that the two
 * products of the input layer (mu:53-59: x W + b and its weight gradient x^T dA) build their
 * operand tiles in LDS from (position, value) pairs instead of streaming the dense batch.
 * Group g holds minibatch rows 16 g .. 16 g + 15; the genes are cut into tiles of 32, 16 tiles
 * make a block of 512.  T = scvae_count_tiles_padded(F) tiles per group (a multiple of 16).
 *   entries[tile_ptr[g][t] & 0x7FFFFFFF .. tile_ptr[g][t + 1] & 0x7FFFFFFF)  bucket (g, t),
 *     tiles ascending, so block b of group g is the run block_ptr[g][b] .. block_ptr[g][b + 1];
 *   entry = bf16(value) << 16 | lo << 13 | (tile & 15) << 9 | (row & 15) << 5 | (gene & 31)
 *     (the value as its bfloat16 bit pattern: the kernels store the upper half as it is);
 *   a count of up to 8 significant bits is one entry (lo = 0); a larger one is two: its upper 8
 *   significant bits (lo = 0) and the remainder (lo = 1) -- the exact bf16 cut of the dense
 *   kernels; bit 31 of a pointer: that bucket / block holds lo entries.
 * Group g owns entries[g * capacity .. (g + 1) * capacity); capacity >= 16 x the largest
 * scvae_csr_row_entries value of the matrix never overflows (status, optional device word: bit 0
 * is set if a group did -- that group is then left empty).  groups * capacity < 2^31. */
typedef struct scvae_count_tiles {
  uint32_t* entries;   /* [groups][capacity] */
  uint32_t* tile_ptr;  /* [groups][T + 1] */
  uint32_t* block_ptr; /* [groups][T / 16 + 1] */
  int64_t capacity;    /* entries per group of 16 rows */
  int32_t* status;     /* optional */
} scvae_count_tiles;

/* ---- work a step carries along (optional: scvae_step_args.side) ----
 * The reference's loop is fetch -> session.run(optimiser) -> fetch -> ... (va:985-1013): four
 * calls per step through this ABI (minibatch, noise, step, optimiser).  A step may carry its own
 * optimiser update and the fetch + noise of the NEXT minibatch: one call, everything ordered on
 * the caller's stream when scvae_plan_step returns (no host synchronisation); the results are
 * those of calling scvae_adam_clip_step, scvae_csr_minibatch and scvae_philox_normal_blocks right
 * after the step.  The buffers written here must not be inputs of the step that carries them
 * (double-buffer the minibatch and the noise): the library may run this work on a second stream
 * of its own beside the step's backward pass (VAE steps whose next minibatch has 1024 cells or
 * more, since round 5: a steady 10 us of a 2.0 ms step, see plan.hip; SCVAE_SIDE_STREAM=1 / 0:
 * always / never). */
typedef struct scvae_side_work {
  /* clip + Adam (scvae_adam_clip_step on the plan's whole parameter buffer with this step's
   * gradients); adam_m == NULL: none.  Training steps only; refused while a data-parallel hook is
   * set (the all-reduce comes between the step and the update). */
  float* adam_m;
  float* adam_v;
  float adam_grad_scale;
  float adam_lr_t;
  float adam_beta1;
  float adam_beta2;
  float adam_epsilon;
  /* the next minibatch: the arguments of scvae_csr_minibatch; fetch_out == NULL: none */
  int32_t fetch_as_u16;
  const int64_t* fetch_indptr;
  const int32_t* fetch_indices;
  const float* fetch_values;
  const int64_t* fetch_rows;
  int64_t fetch_n;
  int64_t fetch_features;
  void* fetch_out;
  int64_t fetch_ld;
  const float* fetch_row_values;
  float* fetch_row_values_out;
  /* optional: the same rows also as tile-indexed non-zeros (scvae_csr_count_tiles); integer
   * count matrices only */
  const scvae_count_tiles* fetch_tiles;
  /* the next step's noise: the arguments of scvae_philox_normal_blocks; noise_out == NULL: none */
  float* noise_out;
  int64_t noise_blocks;
  int64_t noise_block_rows;
  int64_t noise_cols;
  int64_t noise_block_stride;
  int64_t noise_row_offset;
  uint64_t noise_seed;
  uint64_t noise_stream_id;
} scvae_side_work;

typedef struct scvae_step_args {
  const float* x;
  const float* t;
  const float* row_const;
  const float* eps;
  int64_t cells;
  int64_t global_cells;
  int32_t n_iw;
  int32_t n_mc;
  int32_t training;
  int32_t deterministic_z;
  float warm_up_weight;
  float* scalars;
  /* optional outputs (NULL to skip) */
  float* log_p_x_given_z;  /* VAE [S*cells]; GMVAE [K*S*cells]: per-cell sum_F log p(t|z) */
  float* q_z_mean;         /* [cells, L]  (GMVAE: z_mean = sum_k y_k mean_k) */
  float* kl_neurons;       /* [L] share of kl_divergence_neurons */
  float* q_y_logits;       /* GMVAE [cells, K] */
  float* p_x_mean;         /* [cells, F] */
  float* p_x_stddev;       /* [cells, F] */
  float* stddev_of_p_x_given_z_mean; /* [cells, F] */
  float* cluster_stats;    /* GMVAE: [4, K, L] p_z_means, p_z_variances, q_z_means(sum share),
                              q_z_variances(sum share) */
  /* [cells, E] extra decoder inputs (required when cfg.decoder_extra > 0), tiled over the
   * samples like t: the decoder's first layer sees [z | decoder_extra] */
  const float* decoder_extra;
  /* dropout (training steps of a plan with a dropout_keep in ]0, 1[): seed of this step's
   * masks, a new value every step; the mask of every layer input is a function of
   * (dropout_seed, site, row, column), see scvae_dropout_apply */
  uint64_t dropout_seed;
  /* [cells] count sum N of every cell (count_sum_parameter, va:1017-1019): the total of the
   * constrained Poisson's rates.  Required for SCVAE_CONSTRAINED_POISSON, ignored otherwise */
  const float* count_sum;
  /* data parallel: index of this rank's first cell within the global minibatch of
   * global_cells cells (0 on a single GPU).  The dropout masks are a function of the global
   * row, so a sharded step draws exactly the masks of the single-process step */
  int64_t row_offset;
  /* != 0: the caller vouches that x holds integers in [0, 65536) (a count matrix; see
   * scvae_check_counts).  The products x W and x^T dA of the layer that sees x then run on the
   * exact bf16-split kernels (count_gemm.hip: fp32-accurate, 16x the fp32 matrix rate) instead of
   * the fp32 MFMA kernels; 0 (preprocessed / dropped-out / unknown x): fp32 MFMA */
  int32_t x_counts;
  /* Optional: the minibatch as uint16 counts [cells, F] with row pitch counts_ld (a multiple of 8
   * and at least F rounded up to a multiple of 64 -- whole strips of the likelihood kernels --,
   * 16-byte aligned base; scvae_csr_densify_u16).  When given, it IS x and t of this step (x and t
   * may be NULL) and x_counts is implied: the three kernels that stream the minibatch (x W, x^T dA,
   * the fused likelihood heads) read half the bytes; same arithmetic on the same values, so the
   * step is bit-identical to the fp32 batch.  Only where scvae_plan_accepts_counts_u16 says so;
   * anything else is refused with an error. */
  const uint16_t* counts_u16;
  int64_t counts_ld;
  /* Optional, with counts_u16: the same minibatch as tile-indexed non-zeros
   * (scvae_csr_count_tiles of the same rows).  The input layer's two products then read it
   * instead of the dense batch -- same MFMAs on the same operand tiles in the same order:
   * bit-identical to the step without it, a tenth of the bytes through the fabric. */
  const scvae_count_tiles* count_tiles;
  /* optional: optimiser update of this step / fetch and noise of the next one (see above) */
  const scvae_side_work* side;
} scvae_step_args;
/* Debugging aid: with SCVAE_WS_GUARD=1 in the environment (read once per process) every buffer
 * carved out of a plan's workspace is followed by a guard region (scvae_plan_workspace_bytes
 * grows accordingly), filled when the plan is bound; scvae_plan_step then synchronises the stream
 * after the step, checks the regions and fails (-3, scvae_last_error names the buffer's offset)
 * if a kernel wrote past its buffer. */
int scvae_plan_step(scvae_plan* plan, const scvae_step_args* args, void* stream);
/* 1 if a step of `cells` cells of this plan can take its minibatch as uint16 counts
 * (training: 0 = evaluation step, 1 = training step, 2 = training step without importance
 * weighting, n_iw == 1): a plan on the fused likelihood kernels (the four count likelihoods, or
 * the constrained Poisson one under the bf16x9 head arithmetic; no -k, no
 * evaluation statistics requested; while training no dropout on the input layer, and dropout of
 * the likelihood heads only with training == 2 under the bf16x9 head arithmetic),
 * the layer that sees x at most 128 units wide, the count kernels enabled, and a minibatch large
 * enough for them to pay (the threshold of scvae_plan_set_count_gemm).  0 otherwise. */
int scvae_plan_accepts_counts_u16(const scvae_plan* plan, int64_t cells, int32_t training);
/* Decoder only, is_training = False: p_x_mean[rows, F] = mean of p(x|z) for given latent values
 * z[rows, L] -- `session.run(self.p_x_mean, feed_dict={self.z: z, self.is_training: False})`
 * in model.sample() (va:1680-1715; gm:2055-2079, where the one-hot y selects the fed z).
 * rows <= the bound max_cells. */
int scvae_plan_decode(scvae_plan* plan, const float* z, int64_t rows, float* p_x_mean,
                      void* stream);

/* _setup_optimiser (va:2736-2770): g <- clip(g*grad_scale, +-1); tf.train.AdamOptimizer
 * with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller. */
int scvae_adam_clip_step(float* theta, float* grad, float* m, float* v, int64_t n,
                         float grad_scale, float lr_t, float beta1, float beta2, float epsilon,
                         void* stream);

/* ---- individual ops (also used by the tests) ---- */
/* dense_layer (scvae/models/utilities.py:38-76): C (+)= act(op(A) op(B) + bias) */
int scvae_gemm(int32_t trans_a, int32_t trans_b, const float* A, const float* B,
               const float* bias, float* C, int64_t M, int64_t N, int64_t K, int64_t lda,
               int64_t ldb, int64_t ldc, int32_t relu, int32_t accumulate, void* workspace,
               int64_t workspace_bytes, void* stream);
int64_t scvae_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
/* The same dense layer when its input is the count matrix itself (mu:53-59 on
 * x_train[idx].toarray(), va:997-998), x integers in [0, 65536):
 *   mode 0: C[rows, N] = act(x[rows, cols] other[cols, N] + bias)   (x W + b)
 *   mode 1: C[cols, N] = x[rows, cols]^T other[rows, N]             (dW = x^T dA)
 * exact hi/lo bf16 cut of x times an exact three-term bf16 split of `other`, fp32 accumulation
 * on the bf16 matrix cores; N <= 128.  workspace: scvae_count_gemm_workspace_bytes (16-byte
 * aligned).  The precondition on x is the caller's (scvae_check_counts). */
int scvae_count_gemm(int32_t mode, const float* x, int64_t ldx, int64_t rows, int64_t cols,
                     const float* other, int64_t ld_other, int64_t N, const float* bias,
                     int32_t relu, float* C, int64_t ldc, void* workspace, int64_t workspace_bytes,
                     void* stream);
/* scvae_count_gemm with x as uint16 counts (row pitch ldx even, x 4-byte aligned): the same
 * arithmetic on the same values -- bit-identical results */
int scvae_count_gemm_u16(int32_t mode, const uint16_t* x, int64_t ldx, int64_t rows, int64_t cols,
                         const float* other, int64_t ld_other, int64_t N, const float* bias,
                         int32_t relu, float* C, int64_t ldc, void* workspace,
                         int64_t workspace_bytes, void* stream);
int64_t scvae_count_gemm_workspace_bytes(int32_t mode, int64_t rows, int64_t cols, int64_t N);
/* *bad (device int32) = 1 unless every one of the n values is an integer in [0, 65536) */
int scvae_check_counts(const float* values, int64_t n, int32_t* bad, void* stream);
/* p(x|z).log_prob(t) summed over F (va:2583-2590); pre = heads' pre-activations [rows,F] */
int scvae_loglik_fwd(int32_t kind, const float* t, const float* const* pre, const float* row_const,
                     float* ll, int64_t rows, int64_t cells, int64_t F, void* stream);
/* in place: pre_j <- gw[r] * d log p / d pre_j; ll (optional) as above */
int scvae_loglik_bwd(int32_t kind, const float* t, float* const* pre, const float* gw,
                     const float* row_const, float* ll, int64_t rows, int64_t cells, int64_t F,
                     void* stream);
/* Fused X_TILDE heads (va:2466-2505) + log p(t|z) summed over genes (va:2583-2590) and, with
 * train != 0, their backward: dW_j = d^T G_j, db_j = colsum G_j, dd = sum_j G_j W_j^T with
 * G_j = gw[row] * d log p / d pre_j.  d: [rows, H] (H even, <= 126); W_j: [H, F]; b_j: [F];
 * t: [cells, F] (row r uses t[r % cells]); ll: [rows]; dd: [rows, H].
 * train: 0 forward, 1 forward+backward, 3 forward+backward main kernel only (the per-strip
 * partial sums are left unreduced; used by bench.py to time that kernel alone).
 * workspace: scvae_decoder_fused_workspace_bytes for every value of train (the forward-only call
 * keeps the bf16 planes of d there under the bf16x9 arithmetic), 256-byte aligned. */
int64_t scvae_decoder_fused_workspace_bytes(int64_t rows, int64_t H, int64_t F);
/* which schedule scvae_decoder_fused / the step launch for this likelihood and hidden size:
 * 2 = decoder_head2_kernel (two pipelined halves), 1 = decoder_head_kernel, 0 = unsupported H
 * (the step then uses the unfused GEMM + likelihood kernels) */
int32_t scvae_decoder_fused_variant(int32_t kind, int64_t H);
/* Arithmetic of the three products (va:2466-2489 and their backward) inside the fused head
 * kernels: 0 = fp32 matrix cores (v_mfma_f32_32x32x2_f32), 1 = the bf16 matrix cores in the
 * exact nine-term form -- every fp32 operand cut exactly into three bf16 terms, all nine
 * products of a pair (each exact in fp32) accumulated in fp32 -- where that kernel applies (up
 * to three heads, hidden width within its LDS budget; otherwise the fp32 kernel runs).
 * Forward-only calls (train = 0, evaluation steps) of one- and two-head likelihoods follow the
 * same setting (the forward instantiation of the same kernel); three heads evaluate on the fp32
 * matrix cores.
 * There is no process-wide switch: a plan carries its arithmetic (scvae_plan_set_head_arith,
 * below), the stand-alone entry takes it per call in `train`.  scvae_default_head_arith: what a
 * new plan and a call without an arithmetic flag start from -- 1, or
 * SCVAE_HEAD_ARITH=fp32|bf16x9|bf16x6 of the environment, read once.
 * scvae_decoder_train_kernel: which kernel a training launch takes under `arith`:
 * 1 / 2 = scvae_decoder_fused_variant's fp32 schedules, 3 = decoder_fused3.hip (bf16x9),
 * 0 = unsupported H. */
int32_t scvae_default_head_arith(void);
int32_t scvae_decoder_train_kernel(int32_t kind, int64_t H, int32_t arith);
/* the instantiation of that training kernel as a profiler prints it for a launch of `rows` rows
 * (u16: uint16 targets; up to 128 rows the bf16x9 arithmetic runs its all-in-one-phase kernel,
 * beyond them the producer / consumer one) */
int scvae_decoder_train_kernel_name(int32_t kind, int64_t H, int64_t rows, int32_t arith,
                                    int32_t u16, char* out, int64_t n);
/* flags of scvae_decoder_fused's / scvae_decoder_fused_u16's `train` (or'ed to 0 / 1 / 3) */
#define SCVAE_HEADS_FP32 0x100
#define SCVAE_HEADS_BF16X9 0x200
#define SCVAE_HEADS_BF16X6 0x800
/* dd = sum over the gene strips of G W^T: with this flag the producer / consumer training kernel
 * adds each strip's part into eight XCD-local [H][rows] accumulators with fp32 atomics (L2
 * resident) instead of writing per-strip slabs to HBM (0.84 GB at 4096 x 32 738) -- the sums
 * are then NOT bit-repeatable from run to run; default off */
#define SCVAE_HEADS_DD_ATOMICS 0x400
int scvae_decoder_fused(int32_t kind, int32_t train, const float* d, int64_t rows, int64_t H,
                        const float* const* W, const float* const* b, float* const* dW,
                        float* const* db, int64_t F, const float* t, int64_t cells,
                        const float* gw, const float* row_const, float* ll, float* dd,
                        void* workspace, void* stream);
/* the same with the targets as the uint16 minibatch of scvae_csr_densify_u16 (row pitch ldt, a
 * multiple of 8 and at least F rounded up to 64; t 16-byte aligned): what a step given
 * scvae_step_args.counts_u16 launches */
int scvae_decoder_fused_u16(int32_t kind, int32_t train, const float* d, int64_t rows, int64_t H,
                            const float* const* W, const float* const* b, float* const* dW,
                            float* const* db, int64_t F, const uint16_t* t, int64_t ldt,
                            int64_t cells, const float* gw, const float* row_const, float* ll,
                            float* dd, void* workspace, void* stream);
/* element-wise .log_prob(t) / .mean() / .variance() of the DISTRIBUTIONS registry classes
 * (scvae/distributions/utilities.py:206-305, zero_inflated.py:180-199); pre = head
 * pre-activations (n elements each); log_prob and/or (mean, variance) may be NULL */
int scvae_likelihood_elementwise(int32_t kind, const float* t, const float* const* pre,
                                 float* log_prob, float* mean, float* variance, int64_t n,
                                 void* stream);
/* q(z|x) sample + analytic KL (va:2346-2369, 2624-2656) */
int scvae_gauss_latent_fwd(const float* mu_pre, const float* ls_pre, const float* eps, float* z,
                           float* kl_elem, float* kl_cell, int64_t S, int64_t cells, int64_t L,
                           int32_t deterministic, void* stream);
/* tf.nn.dropout as the plan applies it (mu:45-50): out (+)= in * m / keep, m the Bernoulli(keep)
 * mask of (seed, site, row, column) [Philox4x32-10]; in, out: [rows, cols] contiguous.  Sites of
 * a VAE plan: i (ENCODER/i+1), 16 (POSTERIOR/MU), 17 (POSTERIOR/LOG_SIGMA), 32+i (i-th decoder
 * layer in execution order), 48+j (X_TILDE head j), 51 (X_TILDE/P_K); of a GMVAE plan: 64+i
 * (Y/CATEGORICAL/ENCODER/LAYER_i+1), 80 (Y/CATEGORICAL/LOGITS), i (Z/Q/ENCODER/LAYER_i+1),
 * 16, 17 (Z/Q mean, scale), 24, 25 (Z/P mean, scale), 32+i (X/DECODER/LAYER_i+1), 48+j, 51
 * (X/DISTRIBUTION heads); rows of the K passes are stacked, pass-major. */
int scvae_dropout_apply(const float* in, float* out, int64_t rows, int64_t cols, float keep,
                        uint64_t seed, int32_t site, int32_t accumulate, void* stream);
/* minibatch fetch x_train[idx].toarray() (va:985-998) from a device-resident CSR matrix */
int scvae_csr_densify(const int64_t* indptr, const int32_t* indices, const float* values,
                      const int64_t* rows, int64_t n, int64_t F, float* out, void* stream);
/* One launch for the whole minibatch fetch: the dense rows as fp32 (as_u16 = 0: out float
 * [n, ld], ld >= F) or as uint16 counts (as_u16 = 1: see scvae_csr_densify_u16) and, when
 * row_values_out is given, row_values_out[i] = row_values[rows[i]] -- the per-cell lgamma term
 * of scvae_csr_row_lgamma1p that the step takes as row_const */
int scvae_csr_minibatch(const int64_t* indptr, const int32_t* indices, const float* values,
                        const int64_t* rows, int64_t n, int64_t F, void* out, int64_t ld,
                        int32_t as_u16, const float* row_values, float* row_values_out,
                        void* stream);
/* the same minibatch as uint16 counts (precondition: integer counts below 65 536, see
 * scvae_check_counts) with row pitch ld -- a multiple of 8, ld >= F, ld * 2 <= 152 KiB; out
 * 16-byte aligned; the pad columns are zeroed.  Half the bytes for the kernels that stream the
 * minibatch: scvae_step_args.counts_u16 */
int scvae_csr_densify_u16(const int64_t* indptr, const int32_t* indices, const float* values,
                          const int64_t* rows, int64_t n, int64_t F, uint16_t* out, int64_t ld,
                          void* stream);
int scvae_csr_row_lgamma1p(const int64_t* indptr, const float* values, int64_t n_rows, float* out,
                           void* stream);
/* tiles per group of a scvae_count_tiles over F genes: ceil(F / 512) * 16 (F <= 65 536) */
int64_t scvae_count_tiles_padded(int64_t F);
/* out[r] = entries row r contributes to a scvae_count_tiles: its non-zeros, plus one for every
 * count of more than 8 significant bits with a non-zero remainder (int32 per row) */
int scvae_csr_row_entries(const int64_t* indptr, const float* values, int64_t n_rows, int32_t* out,
                          void* stream);
/* the minibatch `rows` of an integer count matrix (values in [0, 65536): scvae_check_counts) as
 * tile-indexed non-zeros; one launch, a workgroup per group of 16 rows */
int scvae_csr_count_tiles(const int64_t* indptr, const int32_t* indices, const float* values,
                          const int64_t* rows, int64_t n, int64_t F,
                          const scvae_count_tiles* tiles, void* stream);
/* scvae_count_gemm_u16 with the contraction read from `tiles` (the same rows); x, the uint16
 * batch, is read only for the leftover terms of the contraction (cols % 32 genes in mode 0,
 * rows % 16 cells in mode 1) and may be NULL when there are none.  Bit-identical results. */
int scvae_count_gemm_tiles(int32_t mode, const scvae_count_tiles* tiles, const uint16_t* x,
                           int64_t ldx, int64_t rows, int64_t cols, const float* other,
                           int64_t ld_other, int64_t N, const float* bias, int32_t relu, float* C,
                           int64_t ldc, void* workspace, int64_t workspace_bytes, void* stream);
int scvae_gather_rows(const float* src, const int64_t* rows, int64_t n, float* out, void* stream);
/* ---- the small ops of the graph, stand-alone (the kernels scvae_plan_step launches; SURVEY.md
 *      section 8b).  Row-major contiguous fp32 unless a pitch is given. ---- */
/* tf.contrib.layers.batch_norm(center=True, scale=False, is_training=True) inside dense_layer
 * (scvae/models/utilities.py:60-70): batch mean and biased variance of the columns of
 * a[rows, N] (row pitch lda).  workspace: scvae_bn_workspace_floats(N) floats. */
int64_t scvae_bn_workspace_floats(int64_t N);
int scvae_bn_stats(const float* a, int64_t lda, int64_t rows, int64_t N, float* mean, float* var,
                   float* workspace, void* stream);
/* h = [relu]((a - mean) / sqrt(var + 1e-3) + beta)   (mu:60-76; the moving statistics in
 * evaluation mode, the batch statistics of scvae_bn_stats while training) */
int scvae_bn_apply_relu_fwd(const float* a, int64_t lda, const float* mean, const float* var,
                            const float* beta, float* h, int64_t ldh, int64_t rows, int64_t N,
                            int32_t relu, void* stream);
/* its backward through the batch statistics: da[rows, N] and dbeta[N] from dh (the gradient
 * w.r.t. h), the forward's h, a, mean, var.  workspace: scvae_bn_workspace_floats(N) + 2 N */
int scvae_bn_apply_relu_bwd(const float* dh, int64_t lddh, const float* h, int64_t ldh,
                            const float* a, int64_t lda, const float* mean, const float* var,
                            int64_t rows, int64_t N, int32_t relu, float* da, int64_t ldda,
                            float* dbeta, float* workspace, void* stream);
/* The GMVAE's "softplus gaussian" pair (du:52-73; gm:2936-3048, 3272-3292): posterior
 * q(z|x,y=k) = N(qm, sqrt(softplus(qs))) with qm, qs [K*B, L]; prior p(z|y=k) = row k of the
 * Z/P dense layers (Wpm, Wps [K, L]; bpm, bps [L]).  z[k,s,b,:] = mean + sigma eps[k,s,b,:];
 * klz[k,s,b] = sum_l log q(z) - log p(z|y=k); qvar (optional) [K*B, L] = sigma^2. */
int scvae_softplus_gaussian_logprob_pair_fwd(const float* qm, const float* qs, const float* Wpm,
                                             const float* bpm, const float* Wps,
                                             const float* bps, const float* eps, float* z,
                                             float* klz, float* qvar, int64_t K, int64_t S,
                                             int64_t B, int64_t L, void* stream);
/* backward: dz [K,S,B,L] (from the decoder) and gklz [K,S,B] (d loss / d klz) -> dqm, dqs
 * [K*B, L] and the per-element prior gradients dprior [K*B, 2 L] = (d mean | d scale
 * pre-activation), to be summed over b for the Z/P layers */
int scvae_softplus_gaussian_logprob_pair_bwd(const float* qm, const float* qs, const float* Wpm,
                                             const float* bpm, const float* Wps,
                                             const float* bps, const float* eps, const float* dz,
                                             const float* gklz, float* dqm, float* dqs,
                                             float* dprior, int64_t K, int64_t S, int64_t B,
                                             int64_t L, void* stream);
/* q(y|x) = Categorical(logits) (gm:3050-3092): y = softmax(logits) [B, K] and
 * kl_y_cell[b] = KL(q(y|x_b) || p(y)): log K - H[q] for the uniform prior (prior_logits NULL,
 * gm:3242-3254), tfp kl_divergence against softmax(prior_logits) otherwise (gm:3256-3258) */
int scvae_categorical_entropy_kl_fwd(const float* logits, float* y, float* kl_y_cell, int64_t B,
                                     int64_t K, const float* prior_logits, void* stream);
/* backward: dlogits = softmax-backward(dy) + c * gate[0] * d kl_y_cell / d logits, with gate a
 * device float (the free-nats switch of gm:3391-3398: 1 when the KL term is above its
 * threshold) */
int scvae_categorical_entropy_kl_bwd(const float* y, const float* dy, const float* gate, float c,
                                     float* dlogits, int64_t B, int64_t K,
                                     const float* prior_logits, void* stream);
/* Importance-weighted bound (va:2717-2734 with log_reduce_exp, mu:129-137):
 * lower_bound = mean_{MC,B} log mean_IW exp(ll - KL), its KL-weighted twin, ENRE and KL into
 * scalars[0..3], and the backward in the same launch: gw[s*B + b] = d(-lower_bound_weighted) /
 * d ll (the softmax weights over the importance samples, scaled by row_scale = 1 / (MC * B)).
 * ll [n_iw*n_mc*B]; kl_cell [B] (analytic) or [n_iw*n_mc*B] (kl_per_sample != 0). */
int scvae_iw_logmeanexp(const float* ll, const float* kl_cell, int32_t kl_per_sample,
                        int32_t n_iw, int32_t n_mc, int64_t B, float kl_weight, float row_scale,
                        float* scalars, float* gw, void* stream);
/* Evaluate-time statistics over the S samples of a cell (va:2665-2713): from the heads'
 * pre-activations pre_j [S*B, F] the mean of p(x|z) averaged over the samples (p_x_mean), the
 * sample mean of its variance (mean_of_var) and the variance of its mean over the samples
 * (var_of_mean), each [B, F].  weight (optional, [B] with stride ldw) and accumulate != 0 form
 * the GMVAE's mixture sums over the clusters (gm:3311-3386). */
int scvae_pxmean_stats(int32_t kind, const float* const* pre, int64_t S, int64_t B, int64_t F,
                       const float* weight, int64_t ldw, int32_t accumulate, float* p_x_mean,
                       float* mean_of_var, float* var_of_mean, void* stream);
/* q.sample() noise: Philox4x32-10 + Box-Muller keyed by (seed, stream_id, row_offset+row, col):
 * key = (seed_lo, seed_hi ^ stream_id_hi), counter = (row_lo, row_hi, col / 4, stream_id_lo) */
int scvae_philox_normal(float* out, int64_t rows, int64_t cols, int64_t row_offset, uint64_t seed,
                        uint64_t stream_id, void* stream);
/* the same for out[blocks][block_rows][cols] in one launch: row r of block g is row
 * `g * block_stride + row_offset + r` of the noise field -- the stacked passes (GMVAE clusters,
 * importance / Monte-Carlo samples) of a rank's shard of a global minibatch of block_stride cells */
int scvae_philox_normal_blocks(float* out, int64_t blocks, int64_t block_rows, int64_t cols,
                               int64_t block_stride, int64_t row_offset, uint64_t seed,
                               uint64_t stream_id, void* stream);
/* batch-norm statistic merge for the data-parallel hook: gathered = [ranks][mean(n)|var(n)],
 * counts = rows per rank (DEVICE int64 array) -> out [mean(n)|var(n)] */
int scvae_bn_merge(const float* gathered, const int64_t* counts, int64_t ranks, int64_t n,
                   float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCVAE_HIP_H */
