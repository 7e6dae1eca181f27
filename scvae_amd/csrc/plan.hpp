#pragma once
// (declarations shared by plan.hip and plan_gmvae.hip)
// Host-side execution plan: the MI355X replacement of the reference's TF graph
// (build: variational_autoencoder.py:2219-2770; run: session.run in the loops at
// variational_autoencoder.py:987-1044, 1092-1150, 1969-2014).  A plan owns no device
// memory: parameters, gradients, moving statistics and workspace are bound by the
// caller.  scvae_plan_step enqueues the whole forward (+backward) kernel sequence on
// one stream with no host synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/scvae_hip.h"
#include "common.hpp"
#include "kernels.hpp"

namespace scvae {

// ------------------------------ layout ------------------------------------
constexpr size_t NPOS = (size_t)-1;
constexpr size_t ALIGN_FLOATS = 64;  // every tensor starts on a 256-byte boundary

struct ParamInfo {
  std::string name;
  size_t offset;
  int rows, cols;  // cols == 0 for vectors
};
struct MovingInfo {
  std::string name;
  size_t offset;
  int size;
};
struct Dense {
  int n_in = 0, n_out = 0;
  size_t w = NPOS, b = NPOS, beta = NPOS;      // offsets in the flat parameter buffer
  size_t mov_mean = NPOS, mov_var = NPOS;      // offsets in the moving-statistics buffer
  bool bn = false;
  // workspace (assigned at bind)
  float* a = nullptr;      // pre-normalisation output [rows, n_out] (BN only)
  float* h = nullptr;      // layer output [rows, n_out]
  float* stats = nullptr;  // [mean | var | s1 | s2], each groups*n_out
  // dropout of the layer's input connections (mu:45-50); keep == 0: none
  float keep = 0.f;
  uint32_t site = 0;           // which mask stream of the step (see dropout_apply)
  float* in_drop = nullptr;    // [rows, n_in] the dropped-out input of the training pass
};

struct Layout {
  std::vector<ParamInfo> params;
  std::vector<MovingInfo> moving;
  size_t n_params = 0, n_moving = 0;
  size_t add(const std::string& name, int rows, int cols) {
    const size_t off = n_params;
    params.push_back({name, off, rows, cols});
    const size_t n = (size_t)rows * (cols ? cols : 1);
    n_params += (n + ALIGN_FLOATS - 1) / ALIGN_FLOATS * ALIGN_FLOATS;
    return off;
  }
  size_t add_moving(const std::string& name, int size) {
    const size_t off = n_moving;
    moving.push_back({name, off, size});
    n_moving += ((size_t)size + ALIGN_FLOATS - 1) / ALIGN_FLOATS * ALIGN_FLOATS;
    return off;
  }
  Dense dense(const std::string& scope, int n_in, int n_out, bool bn) {
    Dense d;
    d.n_in = n_in; d.n_out = n_out; d.bn = bn;
    d.w = add(scope + "/DENSE/weights", n_in, n_out);
    d.b = add(scope + "/DENSE/biases", n_out, 0);
    if (bn) {
      d.beta = add(scope + "/BATCH_NORM/beta", n_out, 0);
      d.mov_mean = add_moving(scope + "/BATCH_NORM/moving_mean", n_out);
      d.mov_var = add_moving(scope + "/BATCH_NORM/moving_variance", n_out);
    }
    return d;
  }
};

static const char* head_names(int kind, int j) {
  static const char* P[] = {"LOG_LAMBDA"};
  static const char* NB[] = {"P", "LOG_R"};
  static const char* ZIP[] = {"PI", "LOG_LAMBDA"};
  static const char* ZINB[] = {"PI", "P", "LOG_R"};
  static const char* CP[] = {"LAMBDA"};
  static const char* BE[] = {"LOGITS"};
  switch (kind) {
    case LK_CPOISSON: return CP[j];
    case LK_BERNOULLI: return BE[j];
    case LK_POISSON: return P[j];
    case LK_NB: return NB[j];
    case LK_ZIP: return ZIP[j];
    default: return ZINB[j];
  }
}

// SCVAE_WS_GUARD=1 (a debugging aid, read once per process): every buffer carved out of a plan's
// workspace is followed by a guard region -- the rest of its 256-byte line and one more line --
// filled with a byte pattern when the plan is bound and checked after every step
// (scvae_plan_step then synchronises the stream and fails with the buffer's offset).  Found the
// overrun of the fused -k scratch in round 5 the second time round; GPU suite: tests/conftest.py.
bool workspace_guard_on();
constexpr unsigned char WS_GUARD_BYTE = 0xA5;

struct Bump {
  char* base;
  size_t used = 0, cap;
  bool dry;  // dry run: only measure
  std::vector<std::pair<size_t, size_t>>* guards = nullptr;   // (offset, bytes), real runs
  Bump(void* b, size_t c, bool d) : base((char*)b), cap(c), dry(d) {}
  float* floats(size_t n) {
    size_t bytes = (n * sizeof(float) + 255) / 256 * 256;
    if (workspace_guard_on()) {
      bytes += 256;
      if (guards) guards->push_back({used + n * sizeof(float), bytes - n * sizeof(float)});
    }
    char* p = dry ? nullptr : base + used;
    used += bytes;
    return (float*)p;
  }
};

}  // namespace scvae

using namespace scvae;

struct scvae_plan {
  scvae_model_config cfg;
  Layout layout;
  int P = 1;  // likelihood heads
  // VAE graph
  std::vector<Dense> enc, dec;
  Dense mu, ls;
  Dense heads[3];
  Dense head_k;               // P_K logits of the piecewise categorical likelihood (cfg.k_max > 0)
  float* pre_k = nullptr;     // [rows, F * (k_max + 1)]
  // bound buffers
  float *params = nullptr, *grads = nullptr, *moving = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  int64_t max_cells = 0, max_samples = 0;
  // workspace views
  float *mu_pre = nullptr, *ls_pre = nullptr, *kl_elem = nullptr, *kl_cell = nullptr;
  float *z = nullptr, *ll = nullptr, *gw = nullptr;
  float* pre[3] = {nullptr, nullptr, nullptr};
  float *dbuf[3] = {nullptr, nullptr, nullptr}, *dz = nullptr, *dmu = nullptr, *dls = nullptr;
  float *mov = nullptr, *vom = nullptr;  // evaluate statistics scratch [cells, F]
  float* gemm_ws = nullptr;
  size_t gemm_ws_bytes = 0;
  float* partial = nullptr;  // row-chunk partial sums (batch norm, column sums)
  float* fused_ws = nullptr;  // fused decoder: per-strip ll / dd partial slabs
  float gw_value = 0.f;       // what the first gw_rows entries of `gw` currently hold (VAE, IW = 1:
  size_t gw_rows = 0;         //  the constant -1/(MC*B) is only rewritten when it changes)
  float *zcat = nullptr, *dzcat = nullptr;  // [rows, L + E]: decoder input [z | extra] and its gradient
  int use_fused = 1;          // fused decoder head kernel (0 = unfused GEMM + likelihood path)
  int dd_atomics = 1;         // 1 (default, default_dd_atomics()): the head kernel adds its part of dd
                              // into XCD-local accumulators (fp32 atomics: sums not bit-repeatable
                              // from run to run); 0: per-strip slabs + a fixed-order reduce
  int head_arith = 1;         // arithmetic of the fused head kernels: 0 fp32 MFMA, 1 bf16x9, 2 bf16x6
                              // (scvae_plan_set_head_arith; a new plan: default_head_arith())
  int use_count_gemm = 1;     // exact bf16-split kernels for products with a count matrix x:
                              // 0 never, 1 where they pay (plan_gemm), 2 always
  int use_bn_cols = 1;        // one-launch batch norm for single-group layers (bn_*_cols)
  unsigned* mid_bar = nullptr;   // midchain.hip's grid-barrier counter (workspace, zeroed at bind)
  unsigned mid_bar_count = 0;    // its value once every launch enqueued so far has run
  bool x_u16 = false;            // this step's minibatch is the uint16 count matrix below
  scvae::CountTiles step_tiles;  // ... and, when ent != nullptr, the same rows as tile-indexed non-zeros
  const uint16_t* step_u16 = nullptr;
  int step_u16_ld = 0;
  // scvae_step_args.side: the plan's second stream, forked where the likelihood heads' gradients
  // are final and joined before the step ends
  const scvae_side_work* side = nullptr;
  hipStream_t side_stream = nullptr;
  // SCVAE_WS_GUARD: the guard regions of the bound workspace, host and device copy, and the flag
  std::vector<std::pair<size_t, size_t>> ws_guards;
  size_t* ws_guards_dev = nullptr;
  int* ws_guard_flag = nullptr;
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  bool side_forked = false;
  bool side_jobs_done = false;  // fetch + noise issued
  size_t side_adam_from = 0;    // parameters [side_adam_from, n) were updated on the second stream
  // scvae_plan_probe_heads: event pairs around the likelihood-head training kernel of the next
  // steps (measurement aid of bench.py)
  std::vector<hipEvent_t> probe_events;
  int probe_next = 0;
  // scvae_plan_probe_stages: per probed step PS_COUNT event pairs (kernels.hpp: ProbeStage) and
  // the mask of those that were recorded
  std::vector<hipEvent_t> stage_events;
  std::vector<unsigned> stage_recorded;
  int stage_next = 0;
  ~scvae_plan();
  int use_mid_chain = 1;      // small VAE steps: hidden layers + heads + latent in two launches
  int use_tile_chain = 1;     // large VAE training steps: one launch per hidden layer and direction
  int use_tile_resident = 0;  // ... and, single process, the layers of a pass in ONE resident launch
                              // (tilechain.hip: grid barriers instead of kernel boundaries).  Off
                              // by default: measured slower than the per-layer launches
  float* tc_part[2] = {nullptr, nullptr};    // tilechain.hip: chunk statistics (forward), ping-pong
  float* tc_spart[2] = {nullptr, nullptr};   // ... chunk sums of the batch-norm backward
  float* tc_slab[TC_MAX_JOBS] = {};          // ... dW / db slabs of the layers of a backward pass
  const float* step_x = nullptr;   // this step's x and whether the caller vouches that it holds
  bool x_counts = false;           //  integers in [0, 65536) (scvae_step_args.x_counts)
  uint64_t drop_seed = 0;     // dropout: this step's mask seed (scvae_step_args.dropout_seed)
  RowMap drop_rows;           // ... and this rank's rows within the global minibatch
  scvae_sync_fn sync = nullptr;
  void* sync_user = nullptr;
  // data parallel: when the backward reaches this layer's weight gradient (the last large GEMM of
  // the step), every gradient at offset >= early_reduce_start is final and is announced to the
  // hook (kind 2) so that its all-reduce overlaps that GEMM
  const Dense* early_reduce_layer = nullptr;
  size_t early_reduce_start = 0;
  // ... in two pieces: [heads_start, end) right after the likelihood-head kernel (the largest
  // block, final first), [early_reduce_start, heads_start) at ENCODER/1's weight gradient
  size_t heads_start = 0;
  // GMVAE graph (gm:2788-3221)
  std::vector<Dense> yenc, zenc, xdec;
  Dense ylogits, qmean, qscale, pmean, pscale;
  size_t prior_off = NPOS;    // K logits of p(y) in the parameter buffer (cfg.prior_mode != 0)
  float *logits = nullptr, *yprob = nullptr, *kl_y_cell = nullptr, *a0 = nullptr;
  float *qm = nullptr, *qs = nullptr, *klz = nullptr, *gklz = nullptr, *dy = nullptr;
  float *dlogits = nullptr, *dqm = nullptr, *dqs = nullptr, *dprior = nullptr;
  float *sum_scratch = nullptr;
};

namespace scvae {
inline CountTiles count_tiles_of(const scvae_count_tiles* t) {
  CountTiles c;
  if (t) { c.ent = t->entries; c.tptr = t->tile_ptr; c.gptr = t->block_ptr; c.cap = t->capacity;
           c.status = t->status; }
  return c;
}
// SCVAE_COUNT_TILES=0: steps ignore scvae_step_args.count_tiles (A/B against the dense batch)
bool count_tiles_enabled();
const char* last_error();
int dense_forward(scvae_plan* p, hipStream_t s, Dense& d, const float* in, int ld_in, int rows,
                  int groups, bool relu, bool training);
int dense_affine(scvae_plan* p, hipStream_t s, Dense& d, const float* in, int ld_in, int rows,
                 int groups, bool relu, bool training);
int dense_backward_activation(scvae_plan* p, hipStream_t s, Dense& d, int rows, int groups,
                              bool relu, const float* dh, float* scratch,
                              int64_t global_rows_per_group, const float** da_out);
int dense_backward(scvae_plan* p, hipStream_t s, Dense& d, const float* in, int ld_in, int rows,
                   int groups, bool relu, const float* dh, float* scratch, float* d_in,
                   bool accumulate_d_in, int64_t global_rows_per_group);
// dropout of a layer's input: in training returns d.in_drop (= mask * in / keep), else `in`
int dense_input(scvae_plan* p, hipStream_t s, Dense& d, const float* in, int ld_in, int rows,
                bool training, const float** in_out, int* ld_out);
// the same mask on the gradient w.r.t. that input: out (+)= mask * g / keep
int dense_input_backward(scvae_plan* p, hipStream_t s, const Dense& d, const float* g, float* out,
                         int rows, bool accumulate);
float dropout_keep(const scvae_model_config& c, int which);
int fill(hipStream_t s, float* dst, float v, size_t n);
// gemm() of kernels.hpp on the plan's workspace; products whose A operand is this step's count
// matrix x (x W, x^T dA) take count_gemm.hip's exact bf16-split kernels
int plan_gemm(scvae_plan* p, hipStream_t s, bool ta, bool tb, const float* A, const float* B,
              const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc, int act,
              bool accumulate);
size_t plan_x_gemm_workspace_bytes(int cells, int features, int n_out);
HeadParams head_params(scvae_plan* p);
// head dropout inside the fused decoder-head kernel (bf16x9 kernel, one likelihood pass per
// step): may this training step take it / the heads' dropped-out inputs and mask parameters
bool heads_fused_dropout_ok(scvae_plan* p, int n_iw);
int heads_dropout_inputs(scvae_plan* p, hipStream_t s, const float* dch, int ld, int R,
                         HeadDropout* out);
int heads_forward(scvae_plan* p, hipStream_t s, const float* dch, int ld, int R, bool training,
                  const float* (&head_in)[4]);
int heads_backward(scvae_plan* p, hipStream_t s, const float* const (&head_in)[4], int R,
                   bool head_drop, float* dd, float* scratch);
int copy(hipStream_t s, const float* src, float* dst, size_t n);
int build_gmvae(scvae_plan* p);
size_t carve_gmvae(scvae_plan* p, void* base, size_t cap, int64_t cells, int64_t samples, bool dry);
int gmvae_step(scvae_plan* p, const scvae_step_args* a, hipStream_t s);
bool gm_tile_chain_ok(const scvae_plan* p, int B, int S, bool training);   // plan_gmvae.hip
int plan_side_fork(scvae_plan* p, hipStream_t s, int point);   // scvae_step_args.side (plan.hip)
int plan_side_finish(scvae_plan* p, hipStream_t s);
}  // namespace scvae

