// Host-side execution plan of the VAE + the exported C ABI (see plan.hpp).
#include <mutex>
#include <vector>
#include "plan.hpp"

namespace scvae {
// ------------------------------ errors ------------------------------------
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return -2;
}
const char* last_error() { return g_error; }

hipError_t max_dynamic_lds(const void* fn, int bytes) {
  // (the attribute is a maximum: it is only ever raised, so a launch that needs less than an
  //  earlier one of the same kernel finds it large enough)
  // (one table for the process: the attribute belongs to the function, not to a host thread)
  struct Seen { const void* fn; int device; int bytes; };
  static std::vector<Seen> seen;
  static std::mutex lock;
  std::lock_guard<std::mutex> guard(lock);
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  Seen* hit = nullptr;
  for (Seen& s : seen)
    if (s.fn == fn && s.device == device) { hit = &s; break; }
  if (hit && hit->bytes >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  if (hit) hit->bytes = bytes; else seen.push_back(Seen{fn, device, bytes});
  return hipSuccess;
}
}  // namespace scvae

namespace scvae {

static int build_vae(scvae_plan* p) {
  const scvae_model_config& c = p->cfg;
  const bool bn = c.batch_norm != 0;
  Layout& L = p->layout;
  int n_in = c.feature_size;
  char scope[64];
  const int n_enc = (c.linear_factor & 1) ? 0 : c.n_hidden;   // LFM: no hidden layers on that side
  const int n_dec = (c.linear_factor & 2) ? 0 : c.n_hidden;
  for (int i = 0; i < n_enc; ++i) {
    snprintf(scope, sizeof scope, "ENCODER/%d", i + 1);
    p->enc.push_back(L.dense(scope, n_in, c.hidden[i], bn));
    n_in = c.hidden[i];
    if (i == 0) p->early_reduce_start = L.n_params;   // everything after ENCODER/1
  }
  p->mu = L.dense("POSTERIOR/MU", n_in, c.latent_size, false);
  // "unit-variance gaussian" (du:323-337): log_sigma is the constant 0, no LOG_SIGMA layer
  if (!(c.latent_mode & 2)) p->ls = L.dense("POSTERIOR/LOG_SIGMA", n_in, c.latent_size, false);
  n_in = c.latent_size + c.decoder_extra;   // decoder input [z | batch one-hot | count sum]
  // dense_layers(reverse_order=True): sizes reversed, scopes numbered n..1 (mu:102-105)
  for (int i = 0; i < n_dec; ++i) {
    const int h = c.hidden[c.n_hidden - 1 - i];
    snprintf(scope, sizeof scope, "DECODER/%d", c.n_hidden - i);
    p->dec.push_back(L.dense(scope, n_in, h, bn));
    n_in = h;
  }
  p->heads_start = L.n_params;   // the likelihood heads are the tail of the parameter buffer
  for (int j = 0; j < p->P; ++j) {
    snprintf(scope, sizeof scope, "X_TILDE/%s", head_names(c.likelihood, j));
    p->heads[j] = L.dense(scope, n_in, c.feature_size, false);
  }
  if (c.k_max > 0)
    p->head_k = L.dense("X_TILDE/P_K", n_in, c.feature_size * (c.k_max + 1), false);
  // the first encoder layer's dW (x^T dA, the last large GEMM of the backward pass) is the only
  // gradient still missing when the hook is told that the rest may be all-reduced
  if (!p->enc.empty()) p->early_reduce_layer = &p->enc[0];
  // dropout of the input connections (va:2221-2232, 2286-2287, 2444-2455, 2487, 2516)
  const float kh = dropout_keep(c, 0), kx = dropout_keep(c, 1), kz = dropout_keep(c, 2);
  for (size_t i = 0; i < p->enc.size(); ++i) {
    p->enc[i].keep = i == 0 ? kx : kh;
    p->enc[i].site = (uint32_t)i;
  }
  p->mu.keep = kh; p->mu.site = 16;
  p->ls.keep = kh; p->ls.site = 17;
  for (size_t i = 0; i < p->dec.size(); ++i) {
    p->dec[i].keep = i == 0 ? kz : kh;
    p->dec[i].site = 32 + (uint32_t)i;
  }
  for (int j = 0; j < p->P; ++j) { p->heads[j].keep = kh; p->heads[j].site = 48 + j; }
  p->head_k.keep = kh; p->head_k.site = 51;
  return 0;
}

bool count_tiles_enabled() {
  static const bool on = [] { const char* e = getenv("SCVAE_COUNT_TILES"); return !(e && e[0] == '0'); }();
  return on;
}
bool workspace_guard_on() {
  static const bool on = [] { const char* e = getenv("SCVAE_WS_GUARD"); return e && e[0] == '1'; }();
  return on;
}
// one workgroup per guard region: the first region (lowest index) holding a byte that is not the
// pattern
__global__ __launch_bounds__(256) void ws_guard_check_kernel(const unsigned char* __restrict__ base,
                                                             const size_t* __restrict__ regions,
                                                             int* __restrict__ first_bad) {
  const size_t off = regions[2 * blockIdx.x], len = regions[2 * blockIdx.x + 1];
  bool bad = false;
  for (size_t i = threadIdx.x; i < len; i += blockDim.x) bad |= base[off + i] != WS_GUARD_BYTE;
  if (bad) atomicMin(first_bad, (int)blockIdx.x);
}
static int ws_guard_arm(scvae_plan* p) {
  if (p->ws_guards_dev) { (void)hipFree(p->ws_guards_dev); p->ws_guards_dev = nullptr; }
  if (p->ws_guards.empty()) return 0;
  std::vector<size_t> flat;
  for (auto& g : p->ws_guards) {
    flat.push_back(g.first); flat.push_back(g.second);
    SCVAE_HIP(hipMemset((char*)p->ws + g.first, WS_GUARD_BYTE, g.second));
  }
  SCVAE_HIP(hipMalloc((void**)&p->ws_guards_dev, flat.size() * sizeof(size_t)));
  SCVAE_HIP(hipMemcpy(p->ws_guards_dev, flat.data(), flat.size() * sizeof(size_t),
                      hipMemcpyHostToDevice));
  if (!p->ws_guard_flag) SCVAE_HIP(hipMalloc((void**)&p->ws_guard_flag, sizeof(int)));
  return 0;
}
static int ws_guard_check(scvae_plan* p, hipStream_t s) {
  if (!p->ws_guards_dev) return 0;
  const int none = 0x7fffffff;
  int first = none;
  SCVAE_HIP(hipMemcpyAsync(p->ws_guard_flag, &first, sizeof(int), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(ws_guard_check_kernel, dim3((unsigned)p->ws_guards.size()), dim3(256), 0, s,
                     (const unsigned char*)p->ws, p->ws_guards_dev, p->ws_guard_flag);
  SCVAE_LAUNCH_CHECK("ws_guard_check_kernel");
  SCVAE_HIP(hipMemcpyAsync(&first, p->ws_guard_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  SCVAE_HIP(hipStreamSynchronize(s));
  if (first != none) {
    set_error("SCVAE_WS_GUARD: a kernel wrote past workspace buffer %d of this plan (the buffer "
              "ending at byte offset %zu)", first, p->ws_guards[(size_t)first].first);
    return -3;
  }
  return 0;
}

// carve (or measure) the workspace
static size_t carve(scvae_plan* p, void* base, size_t cap, int64_t cells, int64_t samples,
                    bool dry) {
  const scvae_model_config& c = p->cfg;
  Bump b(base, cap, dry);
  if (!dry && workspace_guard_on()) { p->ws_guards.clear(); b.guards = &p->ws_guards; }
  const size_t B = (size_t)cells, R = (size_t)cells * samples;
  const size_t Lz = c.latent_size, F = c.feature_size;
  size_t hmax = Lz;
  size_t gws = 0;
  auto track = [&](size_t M, size_t N, size_t K) {
    const size_t w = gemm_workspace_bytes((int)M, (int)N, (int)K);
    if (w > gws) gws = w;
  };
  auto layer_ws = [&](Dense& d, size_t rows) {
    float* a = d.bn ? b.floats(rows * d.n_out) : nullptr;
    float* h = b.floats(rows * d.n_out);
    float* st = d.bn ? b.floats(4 * (size_t)d.n_out) : nullptr;
    if (!dry) { d.a = a; d.h = h; d.stats = st; }
    if ((size_t)d.n_out > hmax) hmax = d.n_out;
    if ((size_t)d.n_in > hmax && d.n_in != c.feature_size) hmax = d.n_in;
    track(rows, d.n_out, d.n_in);  // forward
    track(d.n_in, d.n_out, rows);  // dW
    track(rows, d.n_in, d.n_out);  // dX
  };
  for (auto& d : p->enc) layer_ws(d, B);
  for (auto& d : p->dec) layer_ws(d, R);
  float* mid_bar = b.floats(64);
  if (!dry) p->mid_bar = reinterpret_cast<unsigned*>(mid_bar);
  {   // tilechain.hip (sized by the decoder's rows, the larger count)
    float* q[4 + TC_MAX_JOBS];
    for (int i = 0; i < 4; ++i) q[i] = b.floats(tile_chain_part_floats((int)R));
    for (int i = 0; i < TC_MAX_JOBS; ++i) q[4 + i] = b.floats(tile_chain_slab_floats((int)R));
    if (!dry) {
      p->tc_part[0] = q[0]; p->tc_part[1] = q[1]; p->tc_spart[0] = q[2]; p->tc_spart[1] = q[3];
      for (int i = 0; i < TC_MAX_JOBS; ++i) p->tc_slab[i] = q[4 + i];
    }
  }
  float* mu_pre = b.floats(B * Lz);
  float* ls_pre = b.floats(B * Lz);
  float* kl_elem = b.floats(B * Lz);
  float* kl_cell = b.floats((c.latent_mode & 1) ? R : B);   // Monte-Carlo KL: one per sample row
  float* z = b.floats(R * Lz);
  float* ll = b.floats(R);
  float* gw = b.floats(R);
  float* pre[3] = {nullptr, nullptr, nullptr};
  for (int j = 0; j < p->P; ++j) pre[j] = b.floats(R * F);
  float* d0 = b.floats(R * hmax);
  float* d1 = b.floats(R * hmax);
  float* d2 = b.floats(R * hmax);
  float* dz = b.floats(R * Lz);
  float* dmu = b.floats(B * Lz);
  float* dls = b.floats(B * Lz);
  float* mov = b.floats(B * F);
  float* vom = b.floats(B * F);
  const int hn = p->enc.empty() ? c.feature_size : p->enc.back().n_out;
  const int h1 = p->heads[0].n_in;
  track(B, Lz, hn); track(hn, Lz, B); track(B, hn, Lz);
  track(R, F, h1); track(h1, F, R); track(R, h1, F);
  if (c.k_max > 0) {
    const size_t FC = F * (size_t)(c.k_max + 1);
    track(R, FC, h1); track(h1, FC, R); track(R, h1, FC);
  }
  {   // products with the count matrix x itself (count_gemm.hip): the layer that sees x
    const int n_x = p->enc.empty() ? (int)Lz : p->enc[0].n_out;
    const size_t w = plan_x_gemm_workspace_bytes((int)B, (int)F, n_x);
    if (w > gws) gws = w;
  }
  float* gemm_ws = gws ? b.floats(gws / sizeof(float)) : nullptr;
  size_t pmax = col_sum_partial_floats((int)(F > hmax ? F : hmax));
  if (c.k_max > 0) pmax = col_sum_partial_floats((int)(F * (size_t)(c.k_max + 1)));
  { const size_t q = bn_partial_floats(1, (int)hmax); if (q > pmax) pmax = q; }
  float* partial = b.floats(pmax);
  // (sized whatever the plan's head arithmetic is set to later: the wide range of the bf16x9
  //  training kernel included)
  float* fused_ws = decoder_fused_train_supported(p->P, h1, 1)
                        ? b.floats(decoder_fused_workspace_floats((int)R, h1, (int)F, true))
                        : nullptr;
  const size_t E = (size_t)c.decoder_extra;
  float* zcat = E ? b.floats(R * (Lz + E)) : nullptr;
  float* dzcat = E ? b.floats(R * (Lz + E)) : nullptr;
  // (the class logits of the unfused path; the fused -k launches keep the second launch's ll and
  //  dd there -- more than the logits when there are fewer genes than hidden units)
  size_t pre_k_floats = R * F * (size_t)(c.k_max + 1);
  if (pre_k_floats < decoder_fused_cat_scratch_floats((int)R, (int)h1))
    pre_k_floats = decoder_fused_cat_scratch_floats((int)R, (int)h1);
  float* pre_k = c.k_max > 0 ? b.floats(pre_k_floats) : nullptr;
  // dropped-out layer inputs of the training pass (kept for the weight gradients)
  auto drop_ws = [&](Dense& d, size_t rows) {
    float* q = (d.keep > 0.f && d.n_in > 0) ? b.floats(rows * (size_t)d.n_in) : nullptr;
    if (!dry) d.in_drop = q;
  };
  for (auto& d : p->enc) drop_ws(d, B);
  drop_ws(p->mu, B);
  if (!(c.latent_mode & 2)) drop_ws(p->ls, B);
  for (auto& d : p->dec) drop_ws(d, R);
  for (int j = 0; j < p->P; ++j) drop_ws(p->heads[j], R);
  if (c.k_max > 0) drop_ws(p->head_k, R);
  if (!dry) {
    p->fused_ws = fused_ws;
    p->zcat = zcat; p->dzcat = dzcat;
    p->pre_k = pre_k;
    p->mu_pre = mu_pre; p->ls_pre = ls_pre; p->kl_elem = kl_elem; p->kl_cell = kl_cell;
    p->z = z; p->ll = ll; p->gw = gw;
    for (int j = 0; j < 3; ++j) p->pre[j] = pre[j];
    p->dbuf[0] = d0; p->dbuf[1] = d1; p->dbuf[2] = d2; p->dz = dz; p->dmu = dmu; p->dls = dls;
    p->mov = mov; p->vom = vom;
    p->gemm_ws = gemm_ws; p->gemm_ws_bytes = gws; p->partial = partial;
  }
  return b.used;
}

// ---- dense layer forward: fully_connected (+ batch_norm) (+ relu), mu:38-76 ----
float dropout_keep(const scvae_model_config& c, int which) {
  const float k = c.dropout_keep[which];
  return (k > 0.f && k < 1.f) ? k : 0.f;   // p in {0, 1, False}: no dropout (mu:45)
}

size_t plan_x_gemm_workspace_bytes(int cells, int features, int n_out) {
  if (!count_gemm_supported(n_out)) return 0;
  // The plan runs any minibatch of 1 .. cells rows (the tail of an epoch).  The forward product's
  // need is not monotone in the rows -- fewer row tiles take more split-K slabs -- so reserve
  // the maximum over the row-tile counts (the largest row count of each is the worst of it).
  size_t need = count_gemm_workspace_bytes(1, cells, features, n_out);
  for (int r = cells; r > 0; r = (r - 1) / 256 * 256) {
    const size_t f = count_gemm_workspace_bytes(0, r, features, n_out);
    if (f > need) need = f;
  }
  return need;
}

int plan_gemm(scvae_plan* p, hipStream_t s, bool ta, bool tb, const float* A, const float* B,
              const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc, int act,
              bool accumulate) {
  if (p->x_u16 && A == p->step_x) {
    // the uint16 minibatch: only the count kernels read it
    const int mode = ta ? 1 : 0;
    const int rows = ta ? K : M, cols = ta ? M : K;
    if (tb || accumulate || !count_gemm_supported(N) ||
        count_gemm_workspace_bytes(mode, rows, cols, N) > p->gemm_ws_bytes) {
      set_error("the uint16 minibatch reached a product the count kernels do not cover");
      return -1;
    }
    stage_probe(mode ? PS_COUNT_DW : PS_COUNT_FWD, 0, s);
    // (the same minibatch as tile-indexed non-zeros, when the caller left it: same MFMAs on the
    //  same operand tiles, a tenth of the bytes)
    const int rc = p->step_tiles.ent
        ? count_gemm_tiles(s, mode, p->step_tiles, p->step_u16, p->step_u16_ld, rows, cols, B, ldb,
                           N, bias, act, C, ldc, p->gemm_ws, p->gemm_ws_bytes)
        : count_gemm_u16(s, mode, p->step_u16, p->step_u16_ld, rows, cols, B, ldb, N,
                         bias, act, C, ldc, p->gemm_ws, p->gemm_ws_bytes);
    stage_probe(mode ? PS_COUNT_DW : PS_COUNT_FWD, 1, s);
    return rc;
  }
  if (p->x_counts && p->use_count_gemm && A == p->step_x && !tb && !accumulate &&
      count_gemm_supported(N)) {
    // x [rows, cols]: forward (x W) contracts over the columns, x^T dA over the rows
    const int mode = ta ? 1 : 0;
    const int rows = ta ? K : M, cols = ta ? M : K;
    // the split kernels carry a fixed cost (the split / transpose of the fp32 operand, the
    // slab reduction): measured against the fp32 MFMA kernels at 32 738 genes they win from
    // ~600 cells (forward) / ~300 cells (weight gradient) upwards
    const bool pays = p->use_count_gemm >= 2 ||
                      ((double)rows * cols >= (mode == 0 ? 768.0 : 384.0) * 32768.0);
    if (pays && count_gemm_workspace_bytes(mode, rows, cols, N) <= p->gemm_ws_bytes) {
      stage_probe(mode ? PS_COUNT_DW : PS_COUNT_FWD, 0, s);
      const int rc = count_gemm(s, mode, A, lda, rows, cols, B, ldb, N, bias, act, C, ldc,
                                p->gemm_ws, p->gemm_ws_bytes);
      stage_probe(mode ? PS_COUNT_DW : PS_COUNT_FWD, 1, s);
      return rc;
    }
  }
  return gemm(s, ta, tb, A, B, bias, C, M, N, K, lda, ldb, ldc, act, accumulate, p->gemm_ws,
              p->gemm_ws_bytes);
}

int dense_input(scvae_plan* p, hipStream_t s, Dense& d, const float* in, int ld_in, int rows,
                bool training, const float** in_out, int* ld_out) {
  *in_out = in; *ld_out = ld_in;
  if (!training || d.keep <= 0.f) return 0;
  if (p->x_u16 && in == p->step_x) {
    set_error("dropout on the input layer needs the fp32 minibatch");
    return -1;
  }
  int rc = dropout_apply(s, in, ld_in, d.in_drop, d.n_in, rows, d.n_in, d.keep, p->drop_seed,
                         d.site, 0, p->drop_rows);
  if (rc) return rc;
  *in_out = d.in_drop; *ld_out = d.n_in;
  return 0;
}

int dense_input_backward(scvae_plan* p, hipStream_t s, const Dense& d, const float* g, float* out,
                         int rows, bool accumulate) {
  return dropout_apply(s, g, d.n_in, out, d.n_in, rows, d.n_in, d.keep, p->drop_seed, d.site,
                       accumulate ? 1 : 0, p->drop_rows);
}

int dense_forward(scvae_plan* p, hipStream_t s, Dense& d, const float* in, int ld_in,
                         int rows, int groups, bool relu, bool training) {
  int rc = dense_input(p, s, d, in, ld_in, rows, training, &in, &ld_in);
  if (rc) return rc;
  return dense_affine(p, s, d, in, ld_in, rows, groups, relu, training);
}

// fully_connected (+ batch_norm) (+ relu) on an input that already passed the dropout
int dense_affine(scvae_plan* p, hipStream_t s, Dense& d, const float* in, int ld_in, int rows,
                 int groups, bool relu, bool training) {
  const float* W = p->params + d.w;
  const float* bias = p->params + d.b;
  if (!d.bn) {
    return plan_gemm(p, s, false, false, in, W, bias, d.h, rows, d.n_out, d.n_in, ld_in, d.n_out,
                     d.n_out, relu ? ACT_RELU : ACT_NONE, false);
  }
  int rc = plan_gemm(p, s, false, false, in, W, bias, d.a, rows, d.n_out, d.n_in, ld_in, d.n_out,
                     d.n_out, ACT_NONE, false);
  if (rc) return rc;
  const int N = d.n_out;
  if (training) {
    const int rpg = rows / groups;
    float* mean = d.stats;
    float* var = d.stats + (size_t)groups * N;
    // one group, nothing to exchange between the statistics and their use: one launch
    if (groups == 1 && !p->sync && p->use_bn_cols && bn_cols_supported(rows, N) &&
        (p->use_bn_cols >= 2 || bn_cols_pays(rows)) && bn_cols_layout_ok(d.a, N) && bn_cols_layout_ok(d.h, N) &&
        bn_cols_layout_ok(p->params + d.beta, 4) && bn_cols_layout_ok(mean, N))
      return bn_fwd_cols(s, d.a, N, rows, N, p->params + d.beta, relu ? 1 : 0, d.h, N, mean, var);
    if ((rc = bn_stats(s, d.a, N, rpg, groups, N, mean, var, p->partial))) return rc;
    if (p->sync) {
      // statistics of the global minibatch (sync batch norm); groups == 1 on this path
      if (p->sync(p->sync_user, d.stats, 2 * (int64_t)groups * N, 1, rpg)) {
        set_error("batch-norm sync hook failed");
        return -2;
      }
    }
    if ((rc = bn_apply(s, d.a, N, mean, var, N, p->params + d.beta, d.h, N, rpg, groups, N,
                       relu ? 1 : 0)))
      return rc;
    return 0;
  }
  return bn_apply(s, d.a, N, p->moving + d.mov_mean, p->moving + d.mov_var, 0, p->params + d.beta,
                  d.h, N, rows, 1, N, relu ? 1 : 0);
}

// ---- dense layer backward, part 1: through relu / batch norm.  dh: gradient w.r.t. the
// layer output h [rows, n_out]; *da_out: gradient w.r.t. the affine output (dh itself, or
// `scratch`).  Writes dbeta. ----
int dense_backward_activation(scvae_plan* p, hipStream_t s, Dense& d, int rows, int groups,
                              bool relu, const float* dh, float* scratch,
                              int64_t global_rows_per_group, const float** da_out) {
  const int N = d.n_out;
  int rc;
  *da_out = dh;
  if (d.bn) {
    const int rpg = rows / groups;
    float* mean = d.stats;
    float* var = d.stats + (size_t)groups * N;
    float* s1 = d.stats + 2 * (size_t)groups * N;
    float* s2 = d.stats + 3 * (size_t)groups * N;
    if (groups == 1 && !p->sync && p->use_bn_cols && bn_cols_supported(rows, N) &&
        (p->use_bn_cols >= 2 || bn_cols_pays(rows)) && global_rows_per_group == rows && bn_cols_layout_ok(dh, N) && bn_cols_layout_ok(d.h, N) &&
        bn_cols_layout_ok(d.a, N) && bn_cols_layout_ok(scratch, N) && bn_cols_layout_ok(mean, N) &&
        bn_cols_layout_ok(p->grads + d.beta, 4) && bn_cols_layout_ok(p->moving + d.mov_mean, 4) &&
        bn_cols_layout_ok(p->moving + d.mov_var, 4)) {
      // sums + dbeta + moving averages + da in one launch
      if ((rc = bn_bwd_cols(s, dh, N, d.h, N, d.a, N, mean, var, rows, N, relu ? 1 : 0, scratch, N,
                            s1, s2, p->grads + d.beta, p->moving + d.mov_mean,
                            p->moving + d.mov_var)))
        return rc;
      *da_out = scratch;
      return 0;
    }
    // the statistics launch also writes dbeta (sum over this rank's rows of dA, taken before s1
    // becomes a global sum) and updates the layer's moving averages from the (possibly synced)
    // batch statistics of the forward pass
    if ((rc = bn_bwd_stats(s, dh, N, d.h, N, d.a, N, mean, var, rpg, groups, N, relu ? 1 : 0, s1,
                           s2, p->partial, p->grads + d.beta, p->moving + d.mov_mean,
                           p->moving + d.mov_var, global_rows_per_group)))
      return rc;
    if (p->sync) {
      if (p->sync(p->sync_user, s1, 2 * (int64_t)groups * N, 0, rpg)) {
        set_error("batch-norm backward sync hook failed");
        return -2;
      }
    }
    if ((rc = bn_bwd_apply(s, dh, N, d.h, N, d.a, N, mean, var, s1, s2, rpg, groups, N,
                           relu ? 1 : 0, 1.f / (float)global_rows_per_group, scratch, N)))
      return rc;
    *da_out = scratch;
  } else if (relu) {
    if ((rc = relu_bwd(s, dh, d.h, scratch, (size_t)rows * N))) return rc;
    *da_out = scratch;
  }
  return 0;
}

// ---- dense layer backward: activation, then dW = in^T dA, db = colsum(dA), d_in = dA W^T ----
int dense_backward(scvae_plan* p, hipStream_t s, Dense& d, const float* in, int ld_in, int rows,
                   int groups, bool relu, const float* dh, float* scratch, float* d_in,
                   bool accumulate_d_in, int64_t global_rows_per_group) {
  const int N = d.n_out;
  int rc;
  const float* da = nullptr;
  if (d.keep > 0.f) { in = d.in_drop; ld_in = d.n_in; }   // what the forward GEMM read
  if ((rc = dense_backward_activation(p, s, d, rows, groups, relu, dh, scratch,
                                      global_rows_per_group, &da)))
    return rc;
  if (p->sync && p->early_reduce_layer == &d) {
    // everything between ENCODER/1 and the likelihood heads (announced after the head kernel)
    if (p->sync(p->sync_user, p->grads + p->early_reduce_start,
                (int64_t)(p->heads_start - p->early_reduce_start), 2, 0)) {
      set_error("gradient all-reduce hook failed");
      return -2;
    }
  }
  if ((rc = plan_gemm(p, s, true, false, in, da, nullptr, p->grads + d.w, d.n_in, N, rows, ld_in, N,
                      N, ACT_NONE, false)))
    return rc;
  // bias of a batch-normalised layer: the batch mean is subtracted again, its gradient is
  // identically zero (the reference computes rounding noise there); the slot was zeroed at bind
  if (!d.bn)
    if ((rc = col_sum(s, da, N, rows, N, p->grads + d.b, 1.f, 0, p->partial))) return rc;
  if (d_in) {
    if (d.keep > 0.f && accumulate_d_in) {
      set_error("dense_backward: accumulation into a dropped-out input");
      return -1;
    }
    if ((rc = gemm(s, false, true, da, p->params + d.w, nullptr, d_in, rows, d.n_in, N, N, N,
                   d.n_in, ACT_NONE, accumulate_d_in, p->gemm_ws, p->gemm_ws_bytes)))
      return rc;
    if (d.keep > 0.f)
      if ((rc = dense_input_backward(p, s, d, d_in, d_in, rows, false))) return rc;
  }
  return 0;
}

}  // namespace scvae

namespace scvae {
__global__ void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
__global__ void fill_kernel(float* __restrict__ dst, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    dst[i] = v;
}
int fill(hipStream_t s, float* dst, float v, size_t n) {
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, dst, v, n);
  SCVAE_LAUNCH_CHECK("fill_kernel");
  return 0;
}
int copy(hipStream_t s, const float* src, float* dst, size_t n) {
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n);
  SCVAE_LAUNCH_CHECK("copy_kernel");
  return 0;
}
}  // namespace scvae

namespace scvae {

// ---- unfused X_TILDE heads (shared by the VAE and the GMVAE step) ----
// pre_j = head_in_j W_j + b_j for every likelihood head and the P_K head, each on its own
// dropped-out copy of the decoder output in a training step with dropout (va:2475-2518)
int heads_forward(scvae_plan* p, hipStream_t s, const float* dch, int ld, int R, bool training,
                  const float* (&head_in)[4]) {
  const int F = p->cfg.feature_size, KM = p->cfg.k_max, FC = F * (KM + 1);
  int rc, ldh = ld;
  for (int j = 0; j < p->P; ++j) {
    Dense& hd = p->heads[j];
    if ((rc = dense_input(p, s, hd, dch, ld, R, training, &head_in[j], &ldh))) return rc;
    if ((rc = gemm(s, false, false, head_in[j], p->params + hd.w, p->params + hd.b, p->pre[j], R, F,
                   hd.n_in, ldh, F, F, ACT_NONE, false, p->gemm_ws, p->gemm_ws_bytes)))
      return rc;
  }
  if (KM > 0) {
    Dense& hk = p->head_k;
    if ((rc = dense_input(p, s, hk, dch, ld, R, training, &head_in[3], &ldh))) return rc;
    if ((rc = gemm(s, false, false, head_in[3], p->params + hk.w, p->params + hk.b, p->pre_k, R, FC,
                   hk.n_in, ldh, FC, FC, ACT_NONE, false, p->gemm_ws, p->gemm_ws_bytes)))
      return rc;
  }
  return 0;
}

// dW_j = head_in_j^T G_j, db_j = colsum(G_j), dd = sum_j G_j W_j^T with G_j in place of pre_j;
// with dropout every head's dd passes its own mask before the sum (through `scratch`)
int heads_backward(scvae_plan* p, hipStream_t s, const float* const (&head_in)[4], int R,
                   bool head_drop, float* dd, float* scratch) {
  const int F = p->cfg.feature_size, KM = p->cfg.k_max, FC = F * (KM + 1);
  const int h1 = p->heads[0].n_in;
  int rc;
  auto one = [&](Dense& hd, const float* in, float* G, int N, bool first) -> int {
    if ((rc = gemm(s, true, false, in, G, nullptr, p->grads + hd.w, h1, N, R, h1, N, N, ACT_NONE,
                   false, p->gemm_ws, p->gemm_ws_bytes)))
      return rc;
    if ((rc = col_sum(s, G, N, R, N, p->grads + hd.b, 1.f, 0, p->partial))) return rc;
    if ((rc = gemm(s, false, true, G, p->params + hd.w, nullptr, head_drop ? scratch : dd, R, h1,
                   N, N, N, h1, ACT_NONE, !head_drop && !first, p->gemm_ws, p->gemm_ws_bytes)))
      return rc;
    if (head_drop) return dense_input_backward(p, s, hd, scratch, dd, R, !first);
    return 0;
  };
  for (int j = 0; j < p->P; ++j)
    if ((rc = one(p->heads[j], head_in[j], p->pre[j], F, j == 0))) return rc;
  if (KM > 0)   // the P_K head, same three products on [rows, F * (K + 1)]
    if ((rc = one(p->head_k, head_in[3], p->pre_k, FC, false))) return rc;
  return 0;
}

// Dropout of the heads' input connections inside the fused kernel: the bf16x9 kernel reads one
// dropped-out copy of the decoder output per head (DROP instantiation).  Not with importance
// weights (their separate forward pass has no such instantiation) nor on the fp32 head kernels.
bool heads_fused_dropout_ok(scvae_plan* p, int n_iw) {
  return n_iw == 1 && decoder_fused3_supported(p->P, p->heads[0].n_in) &&
         decoder_train_kernel(p->P, p->heads[0].n_in, p->head_arith) == 3;
}
int heads_dropout_inputs(scvae_plan* p, hipStream_t s, const float* dch, int ld, int R,
                         HeadDropout* out) {
  int rc, ldh = ld;
  for (int j = 0; j < p->P; ++j) {
    Dense& hd = p->heads[j];
    if ((rc = dense_input(p, s, hd, dch, ld, R, true, &out->d[j], &ldh))) return rc;
    out->site[j] = hd.site;
  }
  out->keep = p->heads[0].keep;
  out->seed = p->drop_seed;
  out->map = p->drop_rows;
  return 0;
}

HeadParams head_params(scvae_plan* p) {
  HeadParams hp;
  for (int j = 0; j < 3; ++j) {
    const bool on = j < p->P;
    hp.W[j] = on ? p->params + p->heads[j].w : nullptr;
    hp.b[j] = on ? p->params + p->heads[j].b : nullptr;
    hp.dW[j] = (on && p->grads) ? p->grads + p->heads[j].w : nullptr;
    hp.db[j] = (on && p->grads) ? p->grads + p->heads[j].b : nullptr;
  }
  return hp;
}

// ---- large training minibatches: one launch per hidden layer and direction (tilechain.hip) ----
static bool tile_chain_ok(const scvae_plan* p, int B, int S, bool training) {
  const scvae_model_config& c = p->cfg;
  static const bool env_on = [] { const char* e = getenv("SCVAE_TILE_CHAIN"); return !(e && e[0] == '0'); }();
  if (!env_on || !p->use_tile_chain || !training) return false;
  if (!c.batch_norm || p->enc.empty() || p->dec.empty()) return false;
  if (c.latent_mode != 0 || c.decoder_extra != 0 || c.latent_size > 128) return false;
  if ((int64_t)B * S <= 128) return false;      // (the mid-chain kernels' regime)
  for (const auto& d : p->enc) if (d.n_out > 128 || !d.bn) return false;
  for (const auto& d : p->dec) if (d.n_out > 128 || !d.bn) return false;
  for (int i = 0; i < 3; ++i) if (dropout_keep(c, i) > 0.f) return false;
  return p->tc_part[0] != nullptr;
}
// ... and its layers as ONE resident launch per direction (tilechain.hip): single process (a hook
// needs the host between the stages), every workgroup of the launch co-resident with room to
// spare for the step's second stream, stage and slab-job tables large enough
// SCVAE_TILE_SEGMENTS=1 (A/B runs; a resident plan records as well): stages that only need their
// own tile's rows share a launch.  Off by default: measured SLOWER than one launch per stage
// (2.02-2.03 against 1.99-2.00 ms per 4096-cell step) -- the chain kernels pay more for their
// size (registers, cold start) than two launch boundaries cost.
static bool tile_segments_on() {
  static const bool on = [] { const char* e = getenv("SCVAE_TILE_SEGMENTS"); return e && e[0] == '1'; }();
  return on;
}
static bool tile_resident_ok(const scvae_plan* p, int R) {
  static const bool env_on = [] { const char* e = getenv("SCVAE_TILE_RESIDENT"); return !(e && e[0] == '0'); }();
  if (!env_on || !p->use_tile_resident || p->sync || !p->mid_bar) return false;
  const int tiles = (R + 63) / 64;
  const int cap = tile_chain_resident_capacity();
  if (tiles > cap / 2) return false;
  const int n_enc = (int)p->enc.size(), n_dec = (int)p->dec.size();
  if ((n_enc - 1) + 1 + n_dec + 1 > TCR_MAX_TILES) return false;            // forward tile stages
  if (n_dec + 1 + n_enc > TCR_MAX_TILES) return false;                      // backward tile stages
  if (n_dec + 4 + (n_enc - 1) > TC_MAX_JOBS) return false;                  // dW / db slab jobs
  return true;
}
// the batch norm of layer d as the tile kernels see it
static TileBN tile_bn(scvae_plan* p, Dense& d, const float* part, int chunks, int chunk,
                      float* part_out) {
  TileBN t;
  const int N = d.n_out;
  t.a = d.a; t.h = d.h; t.beta = p->params + d.beta;
  t.mean = d.stats; t.var = d.stats + N;
  t.s1 = d.stats + 2 * (size_t)N; t.s2 = d.stats + 3 * (size_t)N;
  t.part = part; t.chunks = chunks; t.chunk = chunk; t.part_out = part_out;
  t.dbeta = p->grads ? p->grads + d.beta : nullptr;
  t.mov_mean = p->moving + d.mov_mean; t.mov_var = p->moving + d.mov_var;
  return t;
}

// ---- evaluation steps: the hidden layers of the pass in ONE launch ----
// With is_training = False a batch-normalised layer uses its moving statistics (mu:60-70): every
// layer between the input layer's product and the likelihood heads then needs nothing but the
// cell's own row of the layer below -- normalise + relu, product, bias, the posterior heads, the
// reparameterised sample and the decoder's layers are one launch of eval_mlp_kernel
// (tilechain.hip: 16 cells per workgroup, the activations stay in LDS) instead of five GEMM, four
// normalisation and one latent launch (round 6: 120 of an 845 us evaluation step; the same
// stages recorded into one tile_chain_fwd_kernel launch took as long as the launches).
// One sample per cell (drawn, or the deterministic z = mu), analytic KL.  SCVAE_EVAL_CHAIN=0:
// the launches.
static bool eval_chain_ok(const scvae_plan* p, const scvae_step_args* a, int B, int S,
                          bool training) {
  const scvae_model_config& c = p->cfg;
  static const bool env_on = [] { const char* e = getenv("SCVAE_EVAL_CHAIN"); return !(e && e[0] == '0'); }();
  if (!env_on || training) return false;
  if (!c.batch_norm || p->enc.empty() || p->dec.empty()) return false;
  if (c.latent_mode != 0 || c.decoder_extra != 0 || c.latent_size > 128) return false;
  if (S != 1 || (!a->deterministic_z && !a->eps)) return false;
  if (B <= 128) return false;                   // (the mid-chain kernels' regime)
  for (const auto& d : p->enc) if (d.n_out > 128 || !d.bn) return false;
  for (const auto& d : p->dec) if (d.n_out > 128 || !d.bn) return false;
  return ((int)p->enc.size() - 1) + 2 + (int)p->dec.size() <= EM_MAX_OPS;
}
static int eval_chain(scvae_plan* p, const scvae_step_args* a, hipStream_t s, int B) {
  EvalMlpArgs q;
  Dense& d0 = p->enc[0];
  q.rows = B;
  q.a0 = d0.a; q.h0 = d0.h; q.K0 = d0.n_out;
  q.mean0 = p->moving + d0.mov_mean; q.var0 = p->moving + d0.mov_var; q.beta0 = p->params + d0.beta;
  int n = 0;
  auto hidden = [&](Dense& d) {
    EvalMlpArgs::Op& o = q.op[n++];
    o.W = p->params + d.w; o.b = p->params + d.b;
    o.mean = p->moving + d.mov_mean; o.var = p->moving + d.mov_var; o.beta = p->params + d.beta;
    o.pre = d.a; o.out = d.h; o.K = d.n_in; o.N = d.n_out; o.kind = EM_HIDDEN;
  };
  for (size_t i = 1; i < p->enc.size(); ++i) hidden(p->enc[i]);
  const int L = p->cfg.latent_size;
  {
    EvalMlpArgs::Op& o = q.op[n++];
    o.W = p->params + p->mu.w; o.b = p->params + p->mu.b; o.pre = p->mu_pre;
    o.K = p->mu.n_in; o.N = L; o.kind = EM_MU;
  }
  {
    EvalMlpArgs::Op& o = q.op[n++];
    o.W = p->params + p->ls.w; o.b = p->params + p->ls.b; o.pre = p->ls_pre;
    o.K = p->ls.n_in; o.N = L; o.kind = EM_LOG_SIGMA;
  }
  for (auto& d : p->dec) hidden(d);
  q.n_ops = n;
  q.eps = a->deterministic_z ? nullptr : a->eps;
  q.z = p->z; q.kl_elem = p->kl_elem; q.kl_cell = p->kl_cell; q.L = L;
  return eval_mlp(s, q);
}

// Data-parallel steps (scvae_plan_set_sync): the statistics of layer d over the GLOBAL minibatch
// before the tile kernel that consumes them -- this rank's chunks merged into d.stats, the hook
// (kind 1: all-gather + Chan merge over the ranks), and a TileBN that hands them over as given
// (part == nullptr).  Without a hook: the chunks themselves, merged by the consuming kernel.
static int tile_bn_forward(scvae_plan* p, hipStream_t s, Dense& d, const float* part, int chunks,
                           int chunk, int rows, TileBN* out) {
  if (!p->sync) {
    *out = tile_bn(p, d, part, chunks, chunk, nullptr);
    return 0;
  }
  const int N = d.n_out;
  int rc = tile_stats_merge(s, part, chunks, chunk, rows, N, d.stats, d.stats + N);
  if (rc) return rc;
  if (p->sync(p->sync_user, d.stats, 2 * (int64_t)N, 1, rows)) {
    set_error("batch-norm sync hook failed");
    return -2;
  }
  *out = tile_bn(p, d, nullptr, 0, 0, nullptr);
  return 0;
}
// ... and the sums of the backward pass (kind 0: all-reduce); dbeta (this rank's rows) and the
// moving averages are written by the merge
static int tile_bn_backward(scvae_plan* p, hipStream_t s, Dense& d, const float* part, int chunks,
                            int rows, float bessel, TileBN* out) {
  if (!p->sync) {
    *out = tile_bn(p, d, part, chunks, 64, nullptr);
    return 0;
  }
  const int N = d.n_out;
  TileBN t = tile_bn(p, d, nullptr, 0, 0, nullptr);
  int rc = tile_sums_merge(s, part, chunks, N, t, bessel);
  if (rc) return rc;
  if (p->sync(p->sync_user, t.s1, 2 * (int64_t)N, 0, rows)) {   // (rows as the launch chain passes them)
    set_error("batch-norm backward sync hook failed");
    return -2;
  }
  *out = t;
  return 0;
}

// ---- small minibatches: the chain between the input layer and the likelihood heads in one
//      workgroup (midchain.hip) ----
static bool mid_chain_ok(const scvae_plan* p, int B, int S, bool training) {
  const scvae_model_config& c = p->cfg;
  if (!p->use_mid_chain || p->sync) return false;
  if (!vae_mid_chain_resident()) return false;   // (the grid barrier needs co-resident workgroups)
  if (!c.batch_norm || p->enc.empty() || p->dec.empty()) return false;
  if (c.latent_mode != 0 || c.decoder_extra != 0) return false;
  if ((int64_t)B * S > 128 || c.latent_size > 128) return false;
  if (p->enc.size() > (size_t)MID_MAX_LAYERS || p->dec.size() > (size_t)MID_MAX_LAYERS) return false;
  for (const auto& d : p->enc) if (d.n_out > 128) return false;
  for (const auto& d : p->dec) if (d.n_out > 128) return false;
  if (training)   // every parameter layer of the chain would need its own dropped-out input
    for (int i = 0; i < 3; ++i) if (dropout_keep(c, i) > 0.f) return false;
  return true;
}

static MidLayer mid_layer(const scvae_plan* p, const Dense& d) {
  MidLayer m;
  m.W = p->params + d.w;
  m.b = p->params + d.b;
  m.beta = d.bn ? p->params + d.beta : nullptr;
  m.mov_mean = d.bn ? p->moving + d.mov_mean : nullptr;
  m.mov_var = d.bn ? p->moving + d.mov_var : nullptr;
  m.a = d.a; m.h = d.h; m.stats = d.stats;
  m.dW = p->grads ? p->grads + d.w : nullptr;
  m.db = p->grads ? p->grads + d.b : nullptr;
  m.dbeta = (p->grads && d.bn) ? p->grads + d.beta : nullptr;
  m.n_in = d.n_in; m.n_out = d.n_out;
  return m;
}

static MidChainArgs mid_chain_args(const scvae_plan* p, const scvae_step_args* a, int B, int S,
                                   bool training, float kl_coeff) {
  MidChainArgs q;
  memset(&q, 0, sizeof q);
  q.cells = B; q.samples = S; q.latent = p->cfg.latent_size;
  q.n_enc = (int)p->enc.size(); q.n_dec = (int)p->dec.size();
  q.training = training ? 1 : 0;
  q.deterministic = a->deterministic_z ? 1 : 0;
  q.kl_coeff = kl_coeff;
  for (int i = 0; i < q.n_enc; ++i) q.enc[i] = mid_layer(p, p->enc[i]);
  for (int i = 0; i < q.n_dec; ++i) q.dec[i] = mid_layer(p, p->dec[i]);
  q.mu = mid_layer(p, p->mu);
  q.ls = mid_layer(p, p->ls);
  q.eps = a->eps;
  q.mu_pre = p->mu_pre; q.ls_pre = p->ls_pre; q.z = p->z;
  q.kl_elem = p->kl_elem; q.kl_cell = p->kl_cell;
  q.dz = p->dz; q.dmu = p->dmu; q.dls = p->dls;
  q.da0 = p->dbuf[2];
  for (int i = 0; i < 3; ++i) q.buf[i] = p->dbuf[i];
  q.bar = p->mid_bar;
  q.bar_base = p->mid_bar_count;
  return q;
}

// ---- scvae_step_args.side: the optimiser update of this step and the fetch / noise of the next
//      one on the plan's second stream ----
// By default everything runs IN LINE at the end of the step, on the caller's stream: one call
// per step instead of four, nothing else.  Measured on MI355X, every
// way of overlapping this work with the step cost as much as it saved or more:
//   * Adam of the likelihood heads + the fetch forked where the heads' gradients are final
//     (point 1, under the backward pass of the hidden layers): the two HBM-bound kernels triple
//     the duration of the small, latency-bound launches they share the chip with (tile_bwd 22 ->
//     40 us, its statistics 7 -> 31 us) -- step 2.447 vs 2.450 ms;
//   * Adam of the heads forked before the input layer's weight gradient (point 2): the same;
//   * the fetch as an LDS-free kernel co-resident with the likelihood-head kernel (forked before
//     it): that kernel goes from 1.65 to 2.3 ms (step 3.12 ms) -- it does not tolerate a
//     neighbour on its compute units.
// SCVAE_SIDE_STREAM=1 turns the fork on (points 1 / 2, SCVAE_SIDE_ADAM_AT = 1 | 2) for A/B runs.
static int side_jobs(const scvae_side_work* w, hipStream_t st) {
  int rc;
  // (fetch + noise of the same next step: one launch -- the noise is drawn by trailing
  //  workgroups of the minibatch kernel)
  NoiseRequest nr;
  // (small minibatches only: with the noise code in it the minibatch kernel of a 4096-cell fetch
  //  runs 92 instead of 66 us inside the step -- measured, not understood: alone, on a matrix that
  //  fits the infinity cache, it is as fast as before -- which is five launches' worth)
  const bool merge = w->fetch_out && w->fetch_n <= 512;
  if (w->noise_out && w->fetch_out && !merge) {
    if ((rc = philox_normal(st, w->noise_out, w->noise_blocks * w->noise_block_rows,
                            (int)w->noise_cols, w->noise_row_offset, w->noise_seed,
                            w->noise_stream_id, w->noise_block_rows, w->noise_block_stride)))
      return rc;
  } else if (w->noise_out) {
    nr.out = w->noise_out; nr.rows = w->noise_blocks * w->noise_block_rows;
    nr.cols = (int)w->noise_cols; nr.row_offset = w->noise_row_offset; nr.seed = w->noise_seed;
    nr.stream_id = w->noise_stream_id; nr.block_rows = w->noise_block_rows;
    nr.block_stride = w->noise_block_stride;
  }
  if (w->fetch_out) {
    stage_probe(PS_FETCH, 0, st);
    struct EndProbe { hipStream_t s; ~EndProbe() { stage_probe(PS_FETCH, 1, s); } } end_probe{st};
    if (w->fetch_as_u16)
      rc = csr_densify_u16(st, w->fetch_indptr, w->fetch_indices, w->fetch_values, w->fetch_rows,
                           (int)w->fetch_n, (int)w->fetch_features,
                           static_cast<uint16_t*>(w->fetch_out), (int)w->fetch_ld,
                           w->fetch_row_values, w->fetch_row_values_out, &nr);
    else
      rc = csr_densify(st, w->fetch_indptr, w->fetch_indices, w->fetch_values, w->fetch_rows,
                       (int)w->fetch_n, (int)w->fetch_features, static_cast<float*>(w->fetch_out),
                       (int)w->fetch_ld, w->fetch_row_values, w->fetch_row_values_out, &nr);
    if (!rc && w->fetch_tiles)
      rc = csr_count_tiles(st, w->fetch_indptr, w->fetch_indices, w->fetch_values, w->fetch_rows,
                           (int)w->fetch_n, (int)w->fetch_features, count_tiles_of(w->fetch_tiles));
    return rc;
  }
  if (w->noise_out)
    return philox_normal(st, nr.out, nr.rows, nr.cols, nr.row_offset, nr.seed, nr.stream_id,
                         nr.block_rows, nr.block_stride);
  return 0;
}
static int side_adam(scvae_plan* p, hipStream_t st, size_t begin, size_t end) {
  const scvae_side_work* w = p->side;
  if (!w->adam_m || end <= begin) return 0;
  const bool whole = begin == 0 && end == p->layout.n_params;
  if (whole) stage_probe(PS_ADAM, 0, st);
  const int rc = adam_clip_step(st, p->params + begin, p->grads + begin, w->adam_m + begin,
                                w->adam_v + begin, end - begin, w->adam_grad_scale, w->adam_lr_t,
                                w->adam_beta1, w->adam_beta2, w->adam_epsilon);
  if (whole) stage_probe(PS_ADAM, 1, st);
  return rc;
}
static int side_adam_point() {
  static const int at = [] {
    const char* e = getenv("SCVAE_SIDE_ADAM_AT");
    return e ? atoi(e) : 1;
  }();
  return at;
}
int plan_side_fork(scvae_plan* p, hipStream_t s, int point) {
  const scvae_side_work* w = p->side;
  if (!w) return 0;
  // SCVAE_SIDE_STREAM=1 / 0: always / never.  Default (round 5): for minibatches of 1024 cells
  // and more -- the next fetch, its noise and the likelihood heads' share of clip + Adam under the
  // backward pass of the hidden layers (64 workgroups on 256 CUs).  The kernels do overlap (kernel
  // trace: the step's span shrinks by the 100 us moved) but each side slows the other, and what is
  // left without the profiler is a steady 10 us of a 2.0 ms step (300-step A/B, three
  // alternations: 634 -> 624 us outside the head kernel); smaller minibatches: not measured, off.
  static const int env = [] { const char* e = getenv("SCVAE_SIDE_STREAM"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
  const bool on = env >= 0 ? env == 1
                           : (p->cfg.model_type == SCVAE_MODEL_VAE && !p->sync && w->fetch_out &&
                              w->fetch_n >= 1024);   // (single process: never run beside RCCL's kernels)
  if (!on) return 0;
  // (SCVAE_SIDE_JOBS_AT: where the next fetch and its noise leave the step's stream -- 0 at the
  //  start of the step, beside the input layer; 1 after the head kernel, beside the backward pass
  //  of the hidden layers; 2 beside the input layer's weight gradient; 3 never (in line at the end).
  //  Default: 2 from 4096 cells on, 1 below -- three alternations per minibatch size on one box
  //  (tools/ab_side_jobs2.sh), 2 against 1: 1024 cells + 16 us, 2048 + 15-20 us, 4096 -13 / -13 /
  //  + 15 us (a second run: -13 / -11 / -9), ZINB 4096 -31 / -11 / -6, Poisson 4096 -7 / -17 / -4,
  //  16384 cells -64 / -190 / -170 us of 7.3 ms)
  static const int jobs_env = [] { const char* e = getenv("SCVAE_SIDE_JOBS_AT"); return e ? atoi(e) : -1; }();
  const int jobs_at = jobs_env >= 0 ? jobs_env : (w->fetch_n >= 4096 ? 2 : 1);
  // (point 4: an EVALUATION step, right after the input layer's product -- the next fetch runs
  //  beside the hidden layers' small launches, ahead of the head kernel, which does not tolerate
  //  a neighbour)
  const bool jobs = !p->side_jobs_done && (w->fetch_out || w->noise_out) &&
                    (point == 4 || (point < 4 && point >= jobs_at));
  // (VAE plans: the likelihood heads are the tail of the parameter buffer)
  const bool adam = w->adam_m && p->side_adam_from == p->layout.n_params &&
                    p->cfg.model_type == SCVAE_MODEL_VAE &&
                    point == side_adam_point();
  if (!jobs && !adam) return 0;
  if (!p->side_stream) {
    SCVAE_HIP(hipStreamCreateWithFlags(&p->side_stream, hipStreamNonBlocking));
    SCVAE_HIP(hipEventCreateWithFlags(&p->side_fork, hipEventDisableTiming));
    SCVAE_HIP(hipEventCreateWithFlags(&p->side_join, hipEventDisableTiming));
  }
  SCVAE_HIP(hipEventRecord(p->side_fork, s));
  SCVAE_HIP(hipStreamWaitEvent(p->side_stream, p->side_fork, 0));
  p->side_forked = true;
  int rc;
  if (jobs) {
    if ((rc = side_jobs(w, p->side_stream))) return rc;
    p->side_jobs_done = true;
  }
  if (adam) {
    if ((rc = side_adam(p, p->side_stream, p->heads_start, p->layout.n_params))) return rc;
    p->side_adam_from = p->heads_start;
  }
  return 0;
}
// at the end of the step, on the caller's stream: join, then what is left
int plan_side_finish(scvae_plan* p, hipStream_t s) {
  const scvae_side_work* w = p->side;
  if (!w) return 0;
  int rc = 0;
  if (p->side_forked) {
    if (hipEventRecord(p->side_join, p->side_stream) != hipSuccess ||
        hipStreamWaitEvent(s, p->side_join, 0) != hipSuccess) {
      set_error("side stream join");
      rc = -2;
    }
  }
  if (!rc) rc = side_adam(p, s, 0, p->side_adam_from);
  if (!rc && !p->side_jobs_done) rc = side_jobs(w, s);
  p->side = nullptr;
  return rc;
}

static int vae_step(scvae_plan* p, const scvae_step_args* a, hipStream_t s) {
  const scvae_model_config& c = p->cfg;
  const int B = (int)a->cells;
  const int S = a->deterministic_z ? 1 : a->n_iw * a->n_mc;
  const int n_iw = a->deterministic_z ? 1 : a->n_iw;
  const int n_mc = a->deterministic_z ? 1 : a->n_mc;
  const int R = B * S;
  const int F = c.feature_size, L = c.latent_size;
  const bool training = a->training != 0;
  const int64_t GB = a->global_cells > 0 ? a->global_cells : a->cells;
  const float w = a->warm_up_weight * c.kl_weight;
  int rc;
  p->drop_seed = a->dropout_seed;

  // ---------------- forward ----------------
  if (training)
    if ((rc = plan_side_fork(p, s, 0))) return rc;
  const bool mid = mid_chain_ok(p, B, S, training);
  const bool tile = !mid && tile_chain_ok(p, B, S, training);
  // The tile stages of the pass: one launch each (default), or -- single process: a
  // data-parallel hook needs the host between them -- RECORDED and launched together, both
  // measured slower and off by default: in SEGMENTS (SCVAE_TILE_SEGMENTS=1: stages that only
  // need their own tile's rows of the stage before -- posterior heads -> latent stage -> first
  // decoder layer, one sample per cell -- in one launch behind a workgroup barrier,
  // tile_chain_fwd_kernel; a stage that needs every tile's batch-norm statistics starts a new
  // launch) or `resident` (scvae_plan_set_tile_resident: the whole pass in ONE launch, grid
  // barriers where the segments end).
  const bool resident = tile && tile_resident_ok(p, R);
  const bool evalc = !mid && !tile && eval_chain_ok(p, a, B, S, training);
  const bool record = resident || (tile && !p->sync && p->mid_bar && tile_segments_on());
  TileChainFwdArgs cf;
  int cf_tiles = 0;
  auto fwd_flush = [&]() -> int {
    if (cf.n == 0) return 0;
    int r;
    if (cf.n == 1 && cf.kind[0] == TCS_TILE) {
      r = tile_forward(s, cf.f[0]);
    } else if (cf.n == 1) {
      const TileLatent& t = cf.lat;
      r = gauss_latent_fwd(s, t.mu_pre, t.ls_pre, t.eps, t.z, t.kl_elem, t.kl_cell, nullptr, t.S,
                           t.B, t.L, 0);
    } else {
      cf.bar = p->mid_bar; cf.bar_base = p->mid_bar_count;
      unsigned advance = 0;
      r = tile_chain_forward(s, cf, (R + 63) / 64, &advance);
      p->mid_bar_count += advance;
    }
    cf.n = 0; cf_tiles = 0;
    return r;
  };
  auto fwd_stage = [&](const TileFwdArgs& q, int sync_after) -> int {
    if (!record) return tile_forward(s, q);
    SCVAE_ARG(cf.n < TCR_MAX_STAGES && cf_tiles < TCR_MAX_TILES);
    cf.f[cf_tiles] = q;
    cf.kind[cf.n] = TCS_TILE; cf.idx[cf.n] = cf_tiles++; cf.sync[cf.n] = sync_after;
    ++cf.n;
    return (!resident && sync_after >= 2) ? fwd_flush() : 0;
  };
  const float* h = p->step_x;   // (the fp32 batch, or the token of the uint16 one: plan_gemm)
  int ld = F;
  if (mid) {
    // the input-layer product, then everything up to the decoder's output in one workgroup
    Dense& d0 = p->enc[0];
    if ((rc = plan_gemm(p, s, false, false, p->step_x, p->params + d0.w, p->params + d0.b, d0.a, B,
                        d0.n_out, d0.n_in, F, d0.n_out, d0.n_out, ACT_NONE, false)))
      return rc;
    const MidChainArgs q = mid_chain_args(p, a, B, S, training, 0.f);
    if ((rc = vae_mid_forward(s, q))) return rc;
    p->mid_bar_count += vae_mid_barrier_advance(q, false);
    h = p->enc.back().h; ld = p->enc.back().n_out;
  } else if (tile) {
    // the input layer's product, its chunk statistics, then one launch per layer: the consumer
    // of a layer merges its statistics and normalises its own rows of it
    Dense& d0 = p->enc[0];
    if ((rc = plan_gemm(p, s, false, false, p->step_x, p->params + d0.w, p->params + d0.b, d0.a, B,
                        d0.n_out, d0.n_in, F, d0.n_out, d0.n_out, ACT_NONE, false)))
      return rc;
    int chunk = 0, chunks = 0;
    if ((rc = bn_stats_partial(s, d0.a, d0.n_out, B, d0.n_out, p->tc_part[0], &chunk, &chunks)))
      return rc;
    int cur = 0;
    for (size_t i = 1; i < p->enc.size(); ++i) {
      Dense& d = p->enc[i];
      TileFwdArgs q;
      q.rows = B; q.K = d.n_in;
      if ((rc = tile_bn_forward(p, s, p->enc[i - 1], p->tc_part[cur], chunks, chunk, B, &q.bn)))
        return rc;
      q.n_out = 1;
      q.o[0].W = p->params + d.w; q.o[0].b = p->params + d.b; q.o[0].out = d.a;
      q.o[0].part = p->tc_part[cur ^ 1]; q.o[0].N = d.n_out;
      if ((rc = fwd_stage(q, 2))) return rc;
      cur ^= 1; chunk = 64; chunks = (B + 63) / 64;
    }
    {   // the two posterior heads on the normalised output of the last encoder layer
      Dense& last = p->enc.back();
      TileFwdArgs q;
      q.rows = B; q.K = last.n_out;
      if ((rc = tile_bn_forward(p, s, last, p->tc_part[cur], chunks, chunk, B, &q.bn))) return rc;
      q.n_out = 2;
      q.o[0].W = p->params + p->mu.w; q.o[0].b = p->params + p->mu.b; q.o[0].out = p->mu_pre;
      q.o[0].N = L;
      q.o[1].W = p->params + p->ls.w; q.o[1].b = p->params + p->ls.b; q.o[1].out = p->ls_pre;
      q.o[1].N = L;
      if ((rc = fwd_stage(q, 1))) return rc;     // (the latent stage needs the tile's own rows)
    }
    h = p->enc.back().h; ld = p->enc.back().n_out;
  } else if (evalc) {
    // the input layer's product, then everything up to the decoder's output in one launch
    Dense& d0 = p->enc[0];
    if ((rc = plan_gemm(p, s, false, false, p->step_x, p->params + d0.w, p->params + d0.b, d0.a, B,
                        d0.n_out, d0.n_in, F, d0.n_out, d0.n_out, ACT_NONE, false)))
      return rc;
    // (the fetch / noise of the next step leave the stream here, as in the launch chain)
    if ((rc = plan_side_fork(p, s, 4))) return rc;
    if ((rc = eval_chain(p, a, s, B))) return rc;
    h = p->enc.back().h; ld = p->enc.back().n_out;
  } else {
  bool first = true;
  for (auto& d : p->enc) {
    if ((rc = dense_forward(p, s, d, h, ld, B, 1, true, training))) return rc;
    h = d.h; ld = d.n_out;
    // (evaluation steps: the fetch / noise of the next step leave the stream here)
    if (first && !training)
      if ((rc = plan_side_fork(p, s, 4))) return rc;
    first = false;
  }
  }
  Dense& mu = p->mu;
  Dense& ls = p->ls;
  // every parameter layer draws its own mask of the encoder output (va:2281-2289)
  const float* h_mu = h;
  const float* h_ls = h;
  int ld_mu = ld, ld_ls = ld;
  const bool mc_kl = (c.latent_mode & 1) != 0;      // va:2633-2640
  const bool unit_var = (c.latent_mode & 2) != 0;   // du:323-337
  const float* ls_pre = unit_var ? nullptr : p->ls_pre;
  if (record) {
    // (tile_chain_ok: analytic KL, a log_sigma head, training: the stage restates that case)
    SCVAE_ARG(cf.n < TCR_MAX_STAGES && !mc_kl && !unit_var && !a->deterministic_z && a->eps);
    TileLatent& t = cf.lat;
    t.mu_pre = p->mu_pre; t.ls_pre = p->ls_pre; t.eps = a->eps; t.z = p->z;
    t.kl_elem = p->kl_elem; t.kl_cell = p->kl_cell; t.S = S; t.B = B; t.L = L;
    cf.kind[cf.n] = TCS_LATENT; cf.sync[cf.n] = S == 1 ? 1 : 3;   // (decoder tile = its own cells)
    ++cf.n;
    if (!resident && S != 1)
      if ((rc = fwd_flush())) return rc;
  } else if (tile) {
    if ((rc = gauss_latent_fwd(s, p->mu_pre, ls_pre, a->eps, p->z, p->kl_elem, p->kl_cell,
                               mc_kl ? p->kl_cell : nullptr, S, B, L, a->deterministic_z)))
      return rc;
  } else if (!mid && !evalc) {
  if ((rc = dense_input(p, s, mu, h, ld, B, training, &h_mu, &ld_mu))) return rc;
  if ((rc = plan_gemm(p, s, false, false, h_mu, p->params + mu.w, p->params + mu.b, p->mu_pre, B, L,
                      mu.n_in, ld_mu, L, L, ACT_NONE, false)))
    return rc;
  if (!unit_var) {
    if ((rc = dense_input(p, s, ls, h, ld, B, training, &h_ls, &ld_ls))) return rc;
    if ((rc = plan_gemm(p, s, false, false, h_ls, p->params + ls.w, p->params + ls.b, p->ls_pre, B,
                        L, ls.n_in, ld_ls, L, L, ACT_NONE, false)))
      return rc;
  }
  if ((rc = gauss_latent_fwd(s, p->mu_pre, ls_pre, a->eps, p->z, p->kl_elem, p->kl_cell,
                             mc_kl ? p->kl_cell : nullptr, S, B, L, a->deterministic_z)))
    return rc;
  }
  // (recorded stages: the latent stage has not run yet -- these follow its launch)
  auto latent_outputs = [&]() -> int {
    if (a->kl_neurons)
      if (int r = col_sum(s, p->kl_elem, L, B, L, a->kl_neurons, 1.f / (float)GB, 0, p->partial)) return r;
    if (a->q_z_mean)
      if (int r = copy(s, p->mu_pre, a->q_z_mean, (size_t)B * L)) return r;
    return 0;
  };
  if (!record)
    if ((rc = latent_outputs())) return rc;

  const int E = c.decoder_extra;
  const float* dec_in = p->z;   // decoder input: z, or [z | extra] (va:2407-2441)
  if (E > 0) {
    if ((rc = concat_extra(s, p->z, L, a->decoder_extra, E, (size_t)R, (size_t)B, p->zcat)))
      return rc;
    dec_in = p->zcat;
  }
  const float* dch = dec_in;
  ld = L + E;
  if (mid) {
    dch = p->dec.back().h; ld = p->dec.back().n_out;
  } else if (tile) {
    int cur = 0;
    for (size_t i = 0; i <= p->dec.size(); ++i) {
      TileFwdArgs q;
      q.rows = R;
      if (i == 0) { q.x = dec_in; q.ldx = L; q.K = L; }
      else {
        q.K = p->dec[i - 1].n_out;
        if ((rc = tile_bn_forward(p, s, p->dec[i - 1], p->tc_part[cur], (R + 63) / 64, 64, R,
                                  &q.bn)))
          return rc;
      }
      if (i < p->dec.size()) {
        Dense& d = p->dec[i];
        q.n_out = 1;
        q.o[0].W = p->params + d.w; q.o[0].b = p->params + d.b; q.o[0].out = d.a;
        q.o[0].part = p->tc_part[i == 0 ? cur : cur ^ 1]; q.o[0].N = d.n_out;
      }   // (i == size: the last layer's normalisation alone -> its h feeds the likelihood heads)
      if ((rc = fwd_stage(q, 2))) return rc;
      if (i > 0) cur ^= 1;
    }
    if (record) {
      if ((rc = fwd_flush())) return rc;
      if ((rc = latent_outputs())) return rc;
    }
    dch = p->dec.back().h; ld = p->dec.back().n_out;
  } else if (evalc) {
    dch = p->dec.back().h; ld = p->dec.back().n_out;
  } else {
  for (auto& d : p->dec) {
    if ((rc = dense_forward(p, s, d, dch, ld, R, 1, true, training))) return rc;
    dch = d.h; ld = d.n_out;
  }
  }
  HeadPtrs pre;
  for (int j = 0; j < 3; ++j) pre.p[j] = p->pre[j];
  const int h1 = p->heads[0].n_in;
  // the fused kernel never materialises the [rows, P*F] pre-activations; the evaluate-time
  // statistics (p_x_mean, ...) need them, so that request takes the unfused path
  const int KM = c.k_max;   // piecewise categorical likelihood: unfused path
  // dropout gives every head its own mask of the decoder output (va:2475-2488, 2507-2518): the
  // bf16x9 kernel has an instantiation that reads one dropped-out copy per head
  // (heads_fused_dropout_ok); otherwise that training pass is unfused
  const bool head_drop = training && p->heads[0].keep > 0.f;
  // row softmax: three passes of the bf16x9 head kernel (decoder_fused_cpoisson), or unfused
  const bool cpoisson = c.likelihood == LK_CPOISSON;
  if (cpoisson && !a->count_sum) {
    set_error("the constrained Poisson likelihood needs scvae_step_args.count_sum");
    return -1;
  }
  // (every fused kernel takes even widths up to 126; the bf16x9 producer / consumer kernel a wider
  //  range -- odd widths, up to 256 -- for training steps and, its forward half, for evaluation
  //  and the first pass of an importance-weighted step)
  const bool fused_width =
      decoder_fused_supported(h1) ||
      (!head_drop && !cpoisson && decoder_fused_train_supported(p->P, h1, p->head_arith));
  const bool fused = p->use_fused && p->fused_ws && fused_width && ld == h1 &&
                     !a->p_x_mean && KM == 0 &&
                     (!head_drop || (heads_fused_dropout_ok(p, n_iw) && !cpoisson)) &&
                     (c.likelihood <= LK_ZINB || c.likelihood == LK_BERNOULLI ||
                      (cpoisson && decoder_fused_cpoisson_supported(h1, p->head_arith)));
  if (p->x_u16 && !fused) {
    set_error("the uint16 minibatch needs the fused likelihood kernels (no -k / constrained "
              "Poisson, evaluation statistics, or head dropout outside the bf16x9 kernel)");
    return -1;
  }
  const Targets tg = p->x_u16 ? targets_u16(p->step_u16, p->step_u16_ld) : targets_f32(a->t, F);
  const HeadParams hp = head_params(p);
  // -k (k = 1, 2) in a training step: two launches of the bf16x9 head kernel
  // (decoder_fused_train_cat) instead of materialised pre-activations and logits
  const bool fused_cat = training && KM > 0 && p->use_fused && p->fused_ws &&
                         p->pre_k && ld == h1 && !head_drop && !p->x_u16 && !a->p_x_mean &&
                         decoder_fused_cat_supported(c.likelihood, KM, h1, p->head_arith);
  // ... and its forward half (two launches of decoder_forward_kernel) in evaluation passes and
  // in the first pass of an importance-weighted training step
  const bool cat_forward = (!training || (fused_cat && n_iw > 1)) && KM > 0 && p->use_fused &&
                           p->fused_ws && p->pre_k && ld == h1 && !p->x_u16 && !a->p_x_mean &&
                           decoder_fused_forward_cat_supported(c.likelihood, KM, h1);
  const float* head_in[4] = {dch, dch, dch, dch};   // [3]: the P_K head
  if (!fused && !fused_cat && !cat_forward)
    if ((rc = heads_forward(p, s, dch, ld, R, training, head_in))) return rc;
  // per-row log-likelihood, forward only
  auto loglik_forward = [&]() -> int {
    if (fused && cpoisson)
      return decoder_fused_cpoisson(s, false, dch, R, h1, hp, F, tg, B, nullptr, a->count_sum,
                                    a->row_const, p->ll, nullptr, p->fused_ws);
    if (fused)
      return decoder_fused_forward(s, c.likelihood, dch, R, h1, hp, F, tg, B, a->row_const, p->ll,
                                   p->fused_ws, p->head_arith);
    if (cat_forward)
      return decoder_fused_forward_cat(s, c.likelihood, KM, dch, R, h1, hp,
                                       p->params + p->head_k.w, p->params + p->head_k.b, F, a->t,
                                       B, p->ll, p->fused_ws, p->pre_k);
    if (KM > 0)
      return loglik_cat_fwd(s, c.likelihood, a->t, F, pre, F, p->pre_k, KM, p->ll, R, B, F);
    if (cpoisson)
      return cpoisson_fwd(s, a->t, F, p->pre[0], F, a->count_sum, a->row_const, p->ll, R, B, F);
    return loglik_fwd(s, c.likelihood, a->t, F, pre, F, a->row_const, p->ll, R, B, F);
  };
  bool ll_done = false;
  if (a->p_x_mean) {
    if (!(a->p_x_stddev && a->stddev_of_p_x_given_z_mean)) {
      set_error("p_x_mean requires p_x_stddev and stddev_of_p_x_given_z_mean");
      return -1;
    }
    if (cpoisson) {
      // the statistics want the rates, the likelihood the logits: likelihood first, then the
      // logits are normalised in place
      if (training) {
        set_error("p_x_mean in a training step of the constrained Poisson likelihood");
        return -1;
      }
      if ((rc = loglik_forward())) return rc;
      ll_done = true;
      if ((rc = cpoisson_rate(s, p->pre[0], F, a->count_sum, R, B, F))) return rc;
    }
    if (KM > 0)
      rc = px_statistics_cat(s, c.likelihood, pre, F, p->pre_k, KM, S, B, F, nullptr, 0, 0,
                             a->p_x_mean, p->mov, p->vom);
    else
      rc = px_statistics(s, c.likelihood, pre, F, S, B, F, nullptr, 0, 0, a->p_x_mean, p->mov,
                         p->vom);
    if (rc) return rc;
    if ((rc = sqrt_sum(s, p->vom, p->mov, a->p_x_stddev, (size_t)B * F))) return rc;
    if ((rc = sqrt_sum(s, p->vom, nullptr, a->stddev_of_p_x_given_z_mean, (size_t)B * F)))
      return rc;
  }
  const float row_scale = 1.f / ((float)n_mc * (float)GB);
  if (!training) {
    if (!ll_done)
      if ((rc = loglik_forward())) return rc;
    if ((rc = vae_elbo(s, p->ll, p->kl_cell, mc_kl, n_iw, n_mc, B, w, row_scale, a->scalars, nullptr)))
      return rc;
    if (a->log_p_x_given_z)
      if ((rc = copy(s, p->ll, a->log_p_x_given_z, (size_t)R))) return rc;
    return 0;
  }

  // ---------------- backward ----------------
  float* dcur = p->dbuf[0];
  float* dalt = p->dbuf[1];
  if (n_iw == 1) {
    // d(-ELBO_w)/d log p = -1/(MC*B) for every row: known before the likelihood pass
    if (p->gw_rows < (size_t)R || p->gw_value != -row_scale) {
      if ((rc = fill(s, p->gw, -row_scale, (size_t)R))) return rc;
      p->gw_value = -row_scale;
      p->gw_rows = (size_t)R;
    }
  } else {
    p->gw_rows = 0;   // vae_elbo below overwrites gw with the importance weights
    // importance weights need all log-likelihoods first
    if ((rc = loglik_forward())) return rc;
    if ((rc = vae_elbo(s, p->ll, p->kl_cell, mc_kl, n_iw, n_mc, B, w, row_scale, a->scalars, p->gw)))
      return rc;
  }
  if (fused) {
    // heads forward + likelihood + dW_j, db_j, dd in one kernel
    HeadDropout hdrop;
    if (head_drop)
      if ((rc = heads_dropout_inputs(p, s, dch, ld, R, &hdrop))) return rc;
    if (cpoisson)
      rc = decoder_fused_cpoisson(s, true, dch, R, h1, hp, F, tg, B, p->gw, a->count_sum,
                                  a->row_const, p->ll, dcur, p->fused_ws);
    else
      rc = decoder_fused_train(s, c.likelihood, dch, R, h1, hp, F, tg, B, p->gw, a->row_const,
                               p->ll, dcur, p->fused_ws, p->head_arith, false,
                               head_drop ? &hdrop : nullptr, p->dd_atomics);
    if (rc) return rc;
  } else if (fused_cat) {
    // (pre_k -- the logits' buffer of the unfused path -- is free: ll / dd of the second launch)
    Dense& hk = p->head_k;
    if ((rc = decoder_fused_train_cat(s, c.likelihood, KM, dch, R, h1, hp, p->params + hk.w,
                                      p->params + hk.b, p->grads + hk.w, p->grads + hk.b, F, a->t,
                                      B, p->gw, p->ll, dcur, p->fused_ws, p->head_arith, p->pre_k)))
      return rc;
  } else {
    if (KM > 0)
      rc = loglik_cat_bwd(s, c.likelihood, a->t, F, pre, F, p->pre_k, KM, p->gw,
                          n_iw == 1 ? p->ll : nullptr, R, B, F);
    else if (cpoisson)
      rc = cpoisson_bwd(s, a->t, F, p->pre[0], F, p->gw, a->count_sum, a->row_const,
                        n_iw == 1 ? p->ll : nullptr, R, B, F);
    else
      rc = loglik_bwd(s, c.likelihood, a->t, F, pre, F, p->gw, a->row_const,
                      n_iw == 1 ? p->ll : nullptr, R, B, F);
    if (rc) return rc;
    if ((rc = heads_backward(p, s, head_in, R, head_drop, dcur, dalt))) return rc;
  }
  if (p->sync && p->early_reduce_layer != nullptr) {
    // data parallel: the gradients of the likelihood heads -- two thirds of the buffer -- are
    // final; their all-reduce may run under the whole backward pass of the hidden layers
    if (p->sync(p->sync_user, p->grads + p->heads_start,
                (int64_t)(p->layout.n_params - p->heads_start), 2, 0)) {
      set_error("gradient all-reduce hook failed");
      return -2;
    }
  }
  if ((rc = plan_side_fork(p, s, 1))) return rc;   // (scvae_step_args.side)
  if (n_iw == 1)
    if ((rc = vae_elbo(s, p->ll, p->kl_cell, mc_kl, n_iw, n_mc, B, w, row_scale, a->scalars, nullptr)))
      return rc;
  if (a->log_p_x_given_z)
    if ((rc = copy(s, p->ll, a->log_p_x_given_z, (size_t)R))) return rc;

  if (mid) {
    // hidden layers, latent stage and posterior heads backwards in one workgroup; what is left
    // is the input layer's weight gradient x^T dA
    const MidChainArgs q = mid_chain_args(p, a, B, S, true, w / (float)GB);
    if ((rc = vae_mid_backward(s, q))) return rc;
    p->mid_bar_count += vae_mid_barrier_advance(q, true);
    Dense& d0 = p->enc[0];
    if ((rc = plan_side_fork(p, s, 2))) return rc;
    return plan_gemm(p, s, true, false, p->step_x, p->dbuf[2], nullptr, p->grads + d0.w, d0.n_in,
                     d0.n_out, B, F, d0.n_out, d0.n_out, ACT_NONE, false);
  }
  const int64_t GR = GB * S;  // global decoder rows
  if (tile) {
    // one launch per layer (+ the fixed-order reduce of its dW slabs): the layer's batch-norm
    // sums are merged by its own kernel, which also leaves the chunk sums of the layer below
    auto bessel = [](int64_t n) { return (float)n / (float)(n > 1 ? n - 1 : 1); };
    int sp = 0;
    // the dW / db slabs of the layers wait for ONE fixed-order reduce at the end of the pass (they
    // are not on the chain's critical path: four launches fewer); slab buffer i <-> pending job i
    SlabJobs pending;
    TileChainBwdArgs cb;          // (the recorded stages: launched in segments, as going forward)
    int cb_tiles = 0;
    auto bwd_flush = [&]() -> int {
      if (cb.n == 0) return 0;
      int r;
      if (cb.n == 1 && cb.kind[0] == TCS_TILE) {
        r = tile_backward(s, cb.b[0]);
      } else if (cb.n == 1 && cb.kind[0] == TCS_STATS) {
        r = tile_backward_stats(s, cb.stats_dh, cb.stats_bn, cb.stats_rows, cb.stats_N);
      } else if (cb.n == 1) {
        const TileLatent& t = cb.lat;
        r = gauss_latent_bwd(s, t.mu_pre, t.ls_pre, t.eps, t.dz, t.kl_coeff, nullptr, t.dmu, t.dls,
                             t.S, t.B, t.L);
      } else {
        cb.bar = p->mid_bar; cb.bar_base = p->mid_bar_count;
        unsigned advance = 0;
        r = tile_chain_backward(s, cb, (R + 63) / 64, &advance);
        p->mid_bar_count += advance;
      }
      cb.n = 0; cb_tiles = 0;
      return r;
    };
    auto bwd_stage = [&](const TileBwdArgs& q, int sync_after) -> int {
      if (!record) return tile_backward(s, q);
      SCVAE_ARG(cb.n < TCR_MAX_STAGES && cb_tiles < TCR_MAX_TILES);
      cb.b[cb_tiles] = q;
      cb.kind[cb.n] = TCS_TILE; cb.idx[cb.n] = cb_tiles++; cb.sync[cb.n] = sync_after;
      ++cb.n;
      return (!resident && sync_after >= 2) ? bwd_flush() : 0;
    };
    auto flush = [&]() -> int {
      if (pending.n_jobs == 0) return 0;
      if (cb.n != 0) {            // (a segment in flight still writes slabs of this table)
        const int rf = bwd_flush();
        if (rf) return rf;
      }
      const int r = tile_slab_reduce(s, pending);
      pending.n_jobs = 0;
      return r;
    };
    {
      Dense& top = p->dec.back();
      const TileBN tb = tile_bn(p, top, nullptr, 0, 0, p->tc_spart[sp]);
      if (record) {
        cb.stats_dh = dcur; cb.stats_bn = tb; cb.stats_rows = R; cb.stats_N = top.n_out;
        cb.kind[cb.n] = TCS_STATS; cb.sync[cb.n] = 3; ++cb.n;
        if (!resident)
          if ((rc = bwd_flush())) return rc;
      } else if ((rc = tile_backward_stats(s, dcur, tb, R, top.n_out))) {
        return rc;
      }
    }
    auto layer_backward = [&](Dense& d, Dense* below, const float* in, int rows, int64_t grows,
                              const float* dh_in, float* d_in, float* dA_out) -> int {
      TileBwdArgs q;
      const int G = (rows + 63) / 64;
      q.rows = rows; q.inv_count = 1.f / (float)grows; q.bessel = bessel(grows);
      q.n_up = 1;
      if (in && pending.n_jobs == TC_MAX_JOBS) { const int r = flush(); if (r) return r; }
      float* slab = p->tc_slab[pending.n_jobs % TC_MAX_JOBS];
      q.up[0].g = dh_in; q.up[0].W = p->params + d.w; q.up[0].N = d.n_out;
      q.up[0].dW_slab = slab; q.up[0].dA_out = dA_out;
      {
        const int r = tile_bn_backward(p, s, d, p->tc_spart[sp], G, rows, q.bessel, &q.bn);
        if (r) return r;
      }
      q.in = in; q.K = in ? d.n_in : 0; q.d_in = d_in;
      if (below) q.below = tile_bn(p, *below, nullptr, 0, 0, p->tc_spart[sp ^ 1]);
      // what follows needs every tile's chunk sums (a layer below), every slab (the last stage)
      // or -- the first decoder layer, then the latent stage -- the rows of this tile's cells
      int r = bwd_stage(q, (below || !in) ? 3 : (S == 1 ? 1 : 3));
      if (r || !in) return r;
      pending.job[pending.n_jobs++] = {slab, p->grads + d.w, d.n_in * d.n_out, G};
      sp ^= 1;
      return 0;
    };
    for (int i = (int)p->dec.size() - 1; i >= 0; --i) {
      Dense& d = p->dec[i];
      const float* in = i > 0 ? p->dec[i - 1].h : dec_in;
      float* d_in = i > 0 ? dalt : p->dz;
      if ((rc = layer_backward(d, i > 0 ? &p->dec[i - 1] : nullptr, in, R, GR, dcur, d_in, nullptr)))
        return rc;
      if (i > 0) { float* t = dcur; dcur = dalt; dalt = t; }
    }
    if (record) {
      SCVAE_ARG(cb.n < TCR_MAX_STAGES);
      TileLatent& t = cb.lat;
      t.mu_pre = p->mu_pre; t.ls_pre = p->ls_pre; t.eps = a->eps; t.dz = p->dz;
      t.dmu = p->dmu; t.dls = p->dls; t.kl_coeff = w / (float)GB; t.S = S; t.B = B; t.L = L;
      cb.kind[cb.n] = TCS_LATENT; cb.sync[cb.n] = 1; ++cb.n;   // (the heads' tile: the same cells)
    } else if ((rc = gauss_latent_bwd(s, p->mu_pre, ls_pre, a->eps, p->dz, w / (float)GB, nullptr,
                                      p->dmu, p->dls, S, B, L))) {
      return rc;
    }
    float* dh = p->dbuf[0];
    float* dh_alt = p->dbuf[1];
    {   // the two posterior heads: dW, db of both, dh of the last encoder layer and its chunk sums
      Dense& last = p->enc.back();
      const int G = (B + 63) / 64, K = last.n_out;
      if (pending.n_jobs + 4 > TC_MAX_JOBS) { if ((rc = flush())) return rc; }
      float* slab2[2] = {p->tc_slab[pending.n_jobs], p->tc_slab[pending.n_jobs + 1]};
      TileBwdArgs q;
      q.rows = B; q.n_up = 2;
      for (int u = 0; u < 2; ++u) {
        Dense& hd = u == 0 ? p->mu : p->ls;
        q.up[u].g = u == 0 ? p->dmu : p->dls;
        q.up[u].W = p->params + hd.w; q.up[u].N = L;
        q.up[u].dW_slab = slab2[u];
        q.up[u].db_slab = slab2[u] + (size_t)G * 128 * 128;
      }
      q.in = last.h; q.K = K; q.d_in = dh;
      q.below = tile_bn(p, last, nullptr, 0, 0, p->tc_spart[sp]);
      if ((rc = bwd_stage(q, 3))) return rc;
      // (jobs i and i + 1 own slab buffers i and i + 1; the two bias jobs ride in the same
      //  buffers and only take job slots)
      const int j0 = pending.n_jobs;
      pending.job[j0] = {slab2[0], p->grads + p->mu.w, K * L, G};
      pending.job[j0 + 1] = {slab2[1], p->grads + p->ls.w, K * L, G};
      pending.job[j0 + 2] = {q.up[0].db_slab, p->grads + p->mu.b, L, G};
      pending.job[j0 + 3] = {q.up[1].db_slab, p->grads + p->ls.b, L, G};
      pending.n_jobs = j0 + 4;
    }
    for (int i = (int)p->enc.size() - 1; i >= 1; --i) {
      if ((rc = layer_backward(p->enc[i], &p->enc[i - 1], p->enc[i - 1].h, B, GB, dh, dh_alt,
                               nullptr)))
        return rc;
      float* t = dh; dh = dh_alt; dh_alt = t;
    }
    // the layer that sees x: its dA here, its weight gradient x^T dA on the count kernels
    Dense& d0 = p->enc[0];
    if ((rc = layer_backward(d0, nullptr, nullptr, B, GB, dh, nullptr, p->dbuf[2]))) return rc;
    // (the slab sums stay a launch of their own: every tile's slabs would have to cross the
    //  XCDs' L2s behind a full release / acquire barrier)
    if (record)
      if ((rc = bwd_flush())) return rc;
    if ((rc = flush())) return rc;
    if (p->sync && p->early_reduce_layer == &d0) {
      // (data parallel: everything between ENCODER/1 and the likelihood heads is final -- its
      //  all-reduce runs under x^T dA, as in dense_backward)
      if (p->sync(p->sync_user, p->grads + p->early_reduce_start,
                  (int64_t)(p->heads_start - p->early_reduce_start), 2, 0)) {
        set_error("gradient all-reduce hook failed");
        return -2;
      }
    }
    if ((rc = plan_side_fork(p, s, 2))) return rc;
    return plan_gemm(p, s, true, false, p->step_x, p->dbuf[2], nullptr, p->grads + d0.w, d0.n_in,
                     d0.n_out, B, F, d0.n_out, d0.n_out, ACT_NONE, false);
  }
  // decoder layers, last to first; the first decoder layer's input is z
  for (int i = (int)p->dec.size() - 1; i >= 0; --i) {
    Dense& d = p->dec[i];
    const float* in = i > 0 ? p->dec[i - 1].h : dec_in;
    const int ld_in = d.n_in;
    float* d_in = i > 0 ? dalt : (E > 0 ? p->dzcat : p->dz);
    float* scratch = p->dbuf[2];
    if ((rc = dense_backward(p, s, d, in, ld_in, R, 1, true, dcur, scratch, d_in, false, GR)))
      return rc;
    if (i > 0) { float* t = dcur; dcur = dalt; dalt = t; }
  }
  if (E > 0 && !p->dec.empty())
    if ((rc = slice_cols(s, p->dzcat, L + E, L, (size_t)R, p->dz))) return rc;
  // no hidden layers: the heads sit directly on z, their dd is dz
  if (p->dec.empty())
    if ((rc = copy(s, dcur, p->dz, (size_t)R * L))) return rc;
  // latent: dz -> dmu_pre, dls_pre  (d(-ELBO_w)/dKL_cell = w / B_global)
  //         Monte-Carlo KL: d(-ELBO_w)/dKL[s,b] = -w * gw[s,b]
  if ((rc = gauss_latent_bwd(s, p->mu_pre, ls_pre, a->eps, p->dz, mc_kl ? w : w / (float)GB,
                             mc_kl ? p->gw : nullptr, p->dmu, unit_var ? nullptr : p->dls, S, B,
                             L)))
    return rc;
  float* dh = p->dbuf[0];
  float* dh_alt = p->dbuf[1];
  const bool need_dh = !p->enc.empty();
  for (int q = 0; q < (unit_var ? 1 : 2); ++q) {
    Dense& hd = q == 0 ? mu : ls;
    const float* dpre = q == 0 ? p->dmu : p->dls;
    const float* hq = q == 0 ? h_mu : h_ls;        // the (dropped-out) input of that layer
    const int ldq = q == 0 ? ld_mu : ld_ls;
    const bool drop = hd.keep > 0.f;
    if ((rc = plan_gemm(p, s, true, false, hq, dpre, nullptr, p->grads + hd.w, hd.n_in, L, B, ldq,
                        L, L, ACT_NONE, false)))
      return rc;
    if ((rc = col_sum(s, dpre, L, B, L, p->grads + hd.b, 1.f, 0, p->partial))) return rc;
    if (need_dh) {
      if ((rc = gemm(s, false, true, dpre, p->params + hd.w, nullptr, drop ? dh_alt : dh, B,
                     hd.n_in, L, L, L, hd.n_in, ACT_NONE, !drop && q > 0, p->gemm_ws,
                     p->gemm_ws_bytes)))
        return rc;
      if (drop)
        if ((rc = dense_input_backward(p, s, hd, dh_alt, dh, B, q > 0))) return rc;
    }
  }
  for (int i = (int)p->enc.size() - 1; i >= 0; --i) {
    Dense& d = p->enc[i];
    const float* in = i > 0 ? p->enc[i - 1].h : p->step_x;
    const int ld_in = d.n_in;
    float* d_in = i > 0 ? dh_alt : nullptr;
    float* scratch = p->dbuf[2];
    if ((rc = dense_backward(p, s, d, in, ld_in, B, 1, true, dh, scratch, d_in, false, GB)))
      return rc;
    if (i > 0) { float* t = dh; dh = dh_alt; dh_alt = t; }
  }
  // (the batch-norm moving averages were updated by the layers' backward statistics launches)
  return 0;
}

}  // namespace scvae

// =============================== C ABI =====================================
scvae_plan::~scvae_plan() {
  for (hipEvent_t e : probe_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : stage_events) (void)hipEventDestroy(e);
  if (side_stream) {
    (void)hipStreamSynchronize(side_stream);
    (void)hipStreamDestroy(side_stream);
  }
  if (side_fork) (void)hipEventDestroy(side_fork);
  if (side_join) (void)hipEventDestroy(side_join);
  if (ws_guards_dev) (void)hipFree(ws_guards_dev);
  if (ws_guard_flag) (void)hipFree(ws_guard_flag);
}

extern "C" {

const char* scvae_last_error(void) { return scvae::last_error(); }
int scvae_version(void) { return 1; }

int scvae_plan_create(const scvae_model_config* cfg, scvae_plan** out) {
  SCVAE_ARG(cfg && out);
  SCVAE_ARG(cfg->feature_size > 0 && cfg->latent_size > 0 && cfg->latent_size <= 1024);
  SCVAE_ARG(cfg->n_hidden >= 0 && cfg->n_hidden <= SCVAE_MAX_HIDDEN);
  SCVAE_ARG(cfg->likelihood >= 0 && cfg->likelihood <= 5);
  for (int i = 0; i < cfg->n_hidden; ++i) SCVAE_ARG(cfg->hidden[i] > 0);
  SCVAE_ARG(cfg->model_type == SCVAE_MODEL_VAE || cfg->model_type == SCVAE_MODEL_GMVAE);
  SCVAE_ARG(cfg->model_type == SCVAE_MODEL_VAE || (cfg->n_clusters >= 1 && cfg->n_clusters <= 1024));
  SCVAE_ARG(cfg->decoder_extra >= 0 && cfg->decoder_extra <= 4096);
  SCVAE_ARG(cfg->k_max >= 0 && cfg->k_max <= 64);
  SCVAE_ARG(cfg->prior_mode >= 0 && cfg->prior_mode <= 2);
  SCVAE_ARG(cfg->prior_mode == 0 || cfg->model_type == SCVAE_MODEL_GMVAE);
  SCVAE_ARG(cfg->k_max == 0 || cfg->likelihood == SCVAE_POISSON || cfg->likelihood == SCVAE_NB);
  SCVAE_ARG(cfg->linear_factor >= 0 && cfg->linear_factor <= 3);
  SCVAE_ARG(cfg->linear_factor == 0 || cfg->model_type == SCVAE_MODEL_VAE);
  SCVAE_ARG(cfg->decoder_extra == 0 || (cfg->n_hidden > 0 && !(cfg->linear_factor & 2)));
  SCVAE_ARG(cfg->model_type == SCVAE_MODEL_VAE
                ? (cfg->latent_mode >= 0 && cfg->latent_mode <= 3)
                : (cfg->latent_mode == 0 || cfg->latent_mode == 4));
  for (int i = 0; i < 4; ++i) SCVAE_ARG(cfg->dropout_keep[i] >= 0.f && cfg->dropout_keep[i] <= 1.f);
  scvae_plan* p = new scvae_plan();
  p->head_arith = scvae::default_head_arith();
  p->dd_atomics = scvae::default_dd_atomics();
  p->cfg = *cfg;
  p->P = scvae::likelihood_heads(cfg->likelihood);
  if (cfg->model_type == SCVAE_MODEL_GMVAE) scvae::build_gmvae(p);
  else scvae::build_vae(p);
  *out = p;
  return 0;
}

void scvae_plan_destroy(scvae_plan* plan) { delete plan; }

int64_t scvae_plan_param_count(const scvae_plan* p) { return p ? (int64_t)p->layout.params.size() : -1; }
int64_t scvae_plan_param_floats(const scvae_plan* p) { return p ? (int64_t)p->layout.n_params : -1; }
int64_t scvae_plan_moving_floats(const scvae_plan* p) { return p ? (int64_t)p->layout.n_moving : -1; }
int64_t scvae_plan_moving_count(const scvae_plan* p) { return p ? (int64_t)p->layout.moving.size() : -1; }

int scvae_plan_param_info(const scvae_plan* p, int64_t i, char* name, int64_t* offset,
                          int64_t* rows, int64_t* cols) {
  SCVAE_ARG(p && i >= 0 && i < (int64_t)p->layout.params.size());
  const auto& q = p->layout.params[(size_t)i];
  if (name) { strncpy(name, q.name.c_str(), SCVAE_NAME_MAX - 1); name[SCVAE_NAME_MAX - 1] = 0; }
  if (offset) *offset = (int64_t)q.offset;
  if (rows) *rows = q.rows;
  if (cols) *cols = q.cols;
  return 0;
}

int64_t scvae_plan_prior_offset(const scvae_plan* p) {
  return (p && p->prior_off != scvae::NPOS) ? (int64_t)p->prior_off : -1;
}

int scvae_plan_moving_info(const scvae_plan* p, int64_t i, char* name, int64_t* offset,
                           int64_t* size) {
  SCVAE_ARG(p && i >= 0 && i < (int64_t)p->layout.moving.size());
  const auto& q = p->layout.moving[(size_t)i];
  if (name) { strncpy(name, q.name.c_str(), SCVAE_NAME_MAX - 1); name[SCVAE_NAME_MAX - 1] = 0; }
  if (offset) *offset = (int64_t)q.offset;
  if (size) *size = q.size;
  return 0;
}

int64_t scvae_plan_workspace_bytes(const scvae_plan* p, int64_t max_cells, int64_t max_samples) {
  if (!p || max_cells <= 0 || max_samples <= 0) return -1;
  scvae_plan* q = const_cast<scvae_plan*>(p);
  if (p->cfg.model_type == SCVAE_MODEL_GMVAE)
    return (int64_t)scvae::carve_gmvae(q, nullptr, 0, max_cells, max_samples, true);
  return (int64_t)scvae::carve(q, nullptr, 0, max_cells, max_samples, true);
}

int scvae_plan_bind(scvae_plan* p, float* params, float* grads, float* moving, void* workspace,
                    int64_t workspace_bytes, int64_t max_cells, int64_t max_samples) {
  SCVAE_ARG(p && params && workspace && max_cells > 0 && max_samples > 0);
  SCVAE_ARG(p->layout.n_moving == 0 || moving);
  const bool gm = p->cfg.model_type == SCVAE_MODEL_GMVAE;
  const size_t need = gm ? scvae::carve_gmvae(p, nullptr, 0, max_cells, max_samples, true)
                         : scvae::carve(p, nullptr, 0, max_cells, max_samples, true);
  if ((size_t)workspace_bytes < need) {
    scvae::set_error("workspace too small: %lld < %zu bytes", (long long)workspace_bytes, need);
    return -1;
  }
  SCVAE_ARG(((uintptr_t)workspace % 256) == 0 && ((uintptr_t)params % 256) == 0);
  p->params = params; p->grads = grads; p->moving = moving;
  p->gw_rows = 0;
  p->ws = workspace; p->ws_bytes = (size_t)workspace_bytes;
  p->max_cells = max_cells; p->max_samples = max_samples;
  if (gm) scvae::carve_gmvae(p, workspace, (size_t)workspace_bytes, max_cells, max_samples, false);
  else scvae::carve(p, workspace, (size_t)workspace_bytes, max_cells, max_samples, false);
  if (scvae::workspace_guard_on()) {
    const int rc = scvae::ws_guard_arm(p);
    if (rc) return rc;
  }
  if (p->mid_bar) {
    SCVAE_HIP(hipMemset(p->mid_bar, 0, 64 * sizeof(float)));
    p->mid_bar_count = 0;
  }
  if (grads) {
    // bias gradients of batch-normalised layers are identically zero and never written
    for (auto* layers : {&p->enc, &p->dec, &p->yenc, &p->zenc, &p->xdec})
      for (auto& d : *layers)
        if (d.bn) SCVAE_HIP(hipMemset(grads + d.b, 0, (size_t)d.n_out * sizeof(float)));
    if (p->cfg.prior_mode == 1)   // fixed prior logits: no gradient, ever
      SCVAE_HIP(hipMemset(grads + p->prior_off, 0, (size_t)p->cfg.n_clusters * sizeof(float)));
  }
  return 0;
}

int scvae_plan_set_fused(scvae_plan* p, int32_t enabled) {
  SCVAE_ARG(p);
  p->use_fused = enabled ? 1 : 0;
  return 0;
}
int scvae_plan_set_head_arith(scvae_plan* p, int32_t mode) {
  SCVAE_ARG(p && mode >= 0 && mode <= 2);
  p->head_arith = mode;
  return 0;
}
int32_t scvae_plan_head_arith(const scvae_plan* p) { return p ? p->head_arith : -1; }
int scvae_plan_set_dd_atomics(scvae_plan* p, int32_t enabled) {
  SCVAE_ARG(p);
  p->dd_atomics = enabled ? 1 : 0;
  return 0;
}

int scvae_plan_probe_heads(scvae_plan* p, int32_t n) {
  SCVAE_ARG(p && n >= 0 && n <= 4096);
  for (hipEvent_t e : p->probe_events) (void)hipEventDestroy(e);
  p->probe_events.clear();
  p->probe_next = 0;
  for (int i = 0; i < 2 * n; ++i) {
    hipEvent_t e = nullptr;
    SCVAE_HIP(hipEventCreate(&e));
    p->probe_events.push_back(e);
  }
  return 0;
}
int scvae_plan_probe_heads_ms(scvae_plan* p, float* out, int32_t n) {
  SCVAE_ARG(p && out && n >= 0);
  int got = 0;
  for (int i = 0; i < p->probe_next && i < n; ++i) {
    SCVAE_HIP(hipEventSynchronize(p->probe_events[2 * (size_t)i + 1]));
    float ms = 0.f;
    SCVAE_HIP(hipEventElapsedTime(&ms, p->probe_events[2 * (size_t)i],
                                  p->probe_events[2 * (size_t)i + 1]));
    out[got++] = ms;
  }
  return got;
}
int scvae_plan_probe_stages(scvae_plan* p, int32_t n) {
  SCVAE_ARG(p && n >= 0 && n <= 4096);
  for (hipEvent_t e : p->stage_events) (void)hipEventDestroy(e);
  p->stage_events.clear();
  p->stage_recorded.clear();   // (armed per step by its size: set only once every event exists)
  p->stage_next = 0;
  for (int i = 0; i < 2 * scvae::PS_COUNT * n; ++i) {
    hipEvent_t e = nullptr;
    const hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) {
      for (hipEvent_t made : p->stage_events) (void)hipEventDestroy(made);
      p->stage_events.clear();
      return ::scvae::check_hip(err, "hipEventCreate (scvae_plan_probe_stages)");
    }
    p->stage_events.push_back(e);
  }
  p->stage_recorded.assign((size_t)n, 0u);
  return 0;
}
int scvae_plan_probe_stages_us(scvae_plan* p, float* out, int32_t n) {
  SCVAE_ARG(p && out && n >= 0);
  int got = 0;
  for (int i = 0; i < p->stage_next && i < n; ++i, ++got) {
    for (int st = 0; st < scvae::PS_COUNT; ++st) {
      float ms = -1.f;
      if (((p->stage_recorded[(size_t)i] >> (2 * st)) & 3u) == 3u) {
        hipEvent_t* ev = &p->stage_events[(size_t)(i * scvae::PS_COUNT + st) * 2];
        SCVAE_HIP(hipEventSynchronize(ev[1]));
        SCVAE_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
      }
      out[(size_t)i * scvae::PS_COUNT + st] = ms < 0.f ? -1.f : ms * 1e3f;
    }
  }
  return got;
}
int scvae_plan_set_tile_chain(scvae_plan* p, int32_t enabled) {
  SCVAE_ARG(p);
  p->use_tile_chain = enabled ? 1 : 0;
  return 0;
}
int32_t scvae_plan_uses_tile_chain(const scvae_plan* p, int64_t cells, int32_t samples) {
  if (!p || cells <= 0 || samples <= 0) return 0;
  if (p->cfg.model_type == SCVAE_MODEL_GMVAE)   // (the K stacked passes as tile-chain groups)
    return scvae::gm_tile_chain_ok(p, (int)cells, samples, true) ? 1 : 0;
  return tile_chain_ok(p, (int)cells, samples, true) ? 1 : 0;
}
int scvae_plan_set_tile_resident(scvae_plan* p, int32_t enabled) {
  SCVAE_ARG(p);
  p->use_tile_resident = enabled ? 1 : 0;
  return 0;
}
int32_t scvae_plan_uses_tile_resident(const scvae_plan* p, int64_t cells, int32_t samples) {
  if (!p || cells <= 0 || samples <= 0 || p->cfg.model_type != SCVAE_MODEL_VAE) return 0;
  return tile_chain_ok(p, (int)cells, samples, true) &&
                 tile_resident_ok(p, (int)(cells * samples))
             ? 1 : 0;
}
int32_t scvae_plan_fused_categorised(const scvae_plan* p) {
  // (what vae_step / gmvae_step test per step, without the step's own arguments)
  if (!p || p->cfg.k_max <= 0 || !p->use_fused || !p->fused_ws || !p->pre_k) return 0;
  if (p->heads[0].keep > 0.f) return 0;
  return scvae::decoder_fused_cat_supported(p->cfg.likelihood, p->cfg.k_max, p->heads[0].n_in,
                                            p->head_arith) ? 1 : 0;
}
int scvae_plan_set_mid_chain(scvae_plan* p, int32_t enabled) {
  SCVAE_ARG(p);
  p->use_mid_chain = enabled ? 1 : 0;
  return 0;
}

int scvae_plan_set_bn_one_launch(scvae_plan* p, int32_t enabled) {
  SCVAE_ARG(p && enabled >= 0 && enabled <= 2);
  p->use_bn_cols = enabled;
  return 0;
}

int scvae_plan_set_count_gemm(scvae_plan* p, int32_t enabled) {
  SCVAE_ARG(p);
  SCVAE_ARG(enabled >= 0 && enabled <= 2);
  p->use_count_gemm = enabled;
  return 0;
}

int scvae_plan_set_sync(scvae_plan* p, scvae_sync_fn fn, void* user) {
  SCVAE_ARG(p);
  p->sync = fn; p->sync_user = user;
  return 0;
}

int scvae_plan_decode(scvae_plan* p, const float* z, int64_t rows, float* p_x_mean,
                      void* stream) {
  SCVAE_ARG(p && z && p_x_mean);
  SCVAE_ARG(p->params && p->ws);
  SCVAE_ARG(rows > 0 && rows <= p->max_cells);
  if (p->cfg.likelihood == LK_CPOISSON) {   // as the reference: NotImplementedError (va:1642-1645)
    set_error("decode: sampling with the count sum as a likelihood parameter is not defined");
    return -1;
  }
  if (p->cfg.decoder_extra > 0) {   // as the reference: NotImplementedError (va:1638-1650)
    scvae::set_error("sampling with batch correction / count-sum decoder inputs is not supported");
    return -1;
  }
  using namespace scvae;
  hipStream_t s = (hipStream_t)stream;
  const scvae_model_config& c = p->cfg;
  const int R = (int)rows, F = c.feature_size;
  std::vector<Dense>& dec = c.model_type == SCVAE_MODEL_GMVAE ? p->xdec : p->dec;
  int rc;
  const float* h = z;
  int ld = c.latent_size;
  for (auto& d : dec) {
    if ((rc = dense_forward(p, s, d, h, ld, R, 1, true, false))) return rc;
    h = d.h; ld = d.n_out;
  }
  HeadPtrs pre;
  for (int j = 0; j < 3; ++j) pre.p[j] = p->pre[j];
  for (int j = 0; j < p->P; ++j) {
    Dense& hd = p->heads[j];
    if ((rc = gemm(s, false, false, h, p->params + hd.w, p->params + hd.b, p->pre[j], R, F, hd.n_in,
                   ld, F, F, ACT_NONE, false, p->gemm_ws, p->gemm_ws_bytes)))
      return rc;
  }
  if (c.k_max > 0) {
    Dense& hk = p->head_k;
    const int FC = F * (c.k_max + 1);
    if ((rc = gemm(s, false, false, h, p->params + hk.w, p->params + hk.b, p->pre_k, R, FC, hk.n_in,
                   ld, FC, FC, ACT_NONE, false, p->gemm_ws, p->gemm_ws_bytes)))
      return rc;
    return px_statistics_cat(s, c.likelihood, pre, F, p->pre_k, c.k_max, 1, R, F, nullptr, 0, 0,
                             p_x_mean, p->mov, p->vom);
  }
  return px_statistics(s, c.likelihood, pre, F, 1, R, F, nullptr, 0, 0, p_x_mean, p->mov, p->vom);
}

int scvae_plan_accepts_counts_u16(const scvae_plan* p, int64_t cells, int32_t training) {
  if (!p || cells <= 0) return 0;
  const scvae_model_config& c = p->cfg;
  const bool gm = c.model_type == SCVAE_MODEL_GMVAE;
  if (!p->use_count_gemm || !p->use_fused || !p->fused_ws) return 0;
  if (c.k_max > 0) return 0;
  if (c.likelihood > scvae::LK_ZINB &&
      !(c.likelihood == scvae::LK_CPOISSON &&
        scvae::decoder_fused_cpoisson_supported(p->heads[0].n_in, p->head_arith) &&
        !(training && p->heads[0].keep > 0.f)))
    return 0;
  if (!scvae::decoder_fused_supported(p->heads[0].n_in) &&
      !(training == 2 && p->heads[0].keep <= 0.f && c.likelihood <= scvae::LK_ZINB &&
        c.model_type != SCVAE_MODEL_GMVAE &&
        scvae::decoder_fused_train_supported(p->P, p->heads[0].n_in, p->head_arith)))
    return 0;
  // the layers that see x: the VAE's first encoder layer (or the posterior heads of a model
  // without hidden layers); the GMVAE's first q(y|x) and q(z|x,y) layers
  int n_x[2] = {0, 0};
  if (gm) {
    if (p->zenc.empty()) return 0;
    n_x[0] = p->zenc[0].n_out;
    n_x[1] = p->yenc.empty() ? c.n_clusters : p->yenc[0].n_out;
  } else {
    n_x[0] = n_x[1] = p->enc.empty() ? c.latent_size : p->enc[0].n_out;
  }
  for (int i = 0; i < 2; ++i)
    if (!scvae::count_gemm_supported(n_x[i])) return 0;
  if (training) {
    // head dropout: only inside the bf16x9 head kernel, i.e. one likelihood pass per step
    // (training == 2: the caller vouches for n_iw == 1; the GMVAE has no other kind of step)
    if (p->heads[0].keep > 0.f &&
        !((training == 2 || gm) &&
          scvae::decoder_train_kernel(p->P, p->heads[0].n_in, p->head_arith) == 3))
      return 0;
    if (gm) {
      if (p->zenc[0].keep > 0.f) return 0;
      if (!p->yenc.empty() ? p->yenc[0].keep > 0.f : p->ylogits.keep > 0.f) return 0;
    } else if (!p->enc.empty() ? p->enc[0].keep > 0.f : (p->mu.keep > 0.f || p->ls.keep > 0.f)) {
      return 0;
    }
  }
  // both products on the count kernels by the plan's own rule (plan_gemm); the GMVAE, whose K
  // decoder passes re-read the targets, from the size where the weight-gradient kernel pays
  if (p->use_count_gemm < 2 &&
      (double)cells * c.feature_size < (gm ? 384.0 : 768.0) * 32768.0)
    return 0;
  // ... and inside the workspace the plan reserved for them (plan_gemm would refuse the step)
  for (int mode = 0; mode < 2; ++mode)
    for (int i = 0; i < 2; ++i)
      if (scvae::count_gemm_workspace_bytes(mode, (int)cells, c.feature_size, n_x[i]) >
          p->gemm_ws_bytes)
        return 0;
  return 1;
}

int scvae_plan_step(scvae_plan* p, const scvae_step_args* a, void* stream) {
  SCVAE_ARG(p && a);
  SCVAE_ARG(p->params && p->ws);
  const bool u16 = a->counts_u16 != nullptr;
  SCVAE_ARG((u16 || (a->x && a->t)) && a->scalars);
  SCVAE_ARG(a->cells > 0 && a->cells <= p->max_cells);
  if (u16) {
    const bool single_pass = a->deterministic_z || a->n_iw == 1;
    if (!scvae_plan_accepts_counts_u16(p, a->cells,
                                       a->training ? (single_pass ? 2 : 1) : 0)) {
      scvae::set_error("this plan / step does not take a uint16 minibatch "
                       "(scvae_plan_accepts_counts_u16)");
      return -1;
    }
    // (whole 64-gene strips: the likelihood kernels read four counts per lane without a bound)
    SCVAE_ARG(a->counts_ld >= (p->cfg.feature_size + 63) / 64 * 64 && (a->counts_ld & 7) == 0 &&
              ((uintptr_t)a->counts_u16 & 15) == 0);
  }
  SCVAE_ARG(a->n_iw > 0 && a->n_mc > 0);
  SCVAE_ARG(a->deterministic_z || (int64_t)a->n_iw * a->n_mc <= p->max_samples);
  SCVAE_ARG(a->deterministic_z || a->eps);
  SCVAE_ARG(p->cfg.decoder_extra == 0 || a->decoder_extra);
  SCVAE_ARG(!a->training || p->grads);
  SCVAE_ARG(!(a->training && a->deterministic_z));
  SCVAE_ARG(a->row_offset >= 0 &&
            (a->global_cells <= 0 || a->row_offset + a->cells <= a->global_cells));
  p->x_u16 = u16;
  p->step_u16 = a->counts_u16;
  p->step_u16_ld = (int)a->counts_ld;
  p->step_tiles = scvae::CountTiles();
  if (a->count_tiles) {
    SCVAE_ARG(u16 && a->count_tiles->entries && a->count_tiles->tile_ptr &&
              a->count_tiles->block_ptr && a->count_tiles->capacity > 0 &&
              scvae::count_tiles_supported(p->cfg.feature_size));
    if (scvae::count_tiles_enabled()) p->step_tiles = scvae::count_tiles_of(a->count_tiles);
  }
  // (the uint16 batch travels through the layers as an opaque token; only plan_gemm, which
  //  hands it to the count kernels, and the fused likelihood launch look behind it)
  p->step_x = u16 ? reinterpret_cast<const float*>(a->counts_u16) : a->x;
  p->x_counts = u16 || a->x_counts != 0;
  p->drop_seed = a->dropout_seed;
  p->drop_rows = RowMap();
  if (a->global_cells > a->cells) {   // a shard of a data-parallel minibatch
    p->drop_rows.cells = a->cells;
    p->drop_rows.global_cells = a->global_cells;
    p->drop_rows.offset = a->row_offset;
  }
  p->side = nullptr;
  p->side_forked = false;
  p->side_jobs_done = false;
  p->side_adam_from = p->layout.n_params;
  if (a->side) {
    const scvae_side_work* w = a->side;
    if (w->adam_m) {
      SCVAE_ARG(w->adam_v && a->training);
      if (p->sync) {
        scvae::set_error("scvae_side_work: the optimiser update cannot ride with a data-parallel "
                         "step (the gradient all-reduce comes first)");
        return -1;
      }
    }
    // (the outputs of the side work must not OVERLAP what this step reads -- byte ranges, not
    //  base pointers: a slice of one tensor at another offset would race the step's reads on
    //  the second stream)
    auto overlaps = [](const void* p1, size_t n1, const void* p2, size_t n2) {
      if (!p1 || !p2 || !n1 || !n2) return false;
      const char* a1 = (const char*)p1;
      const char* a2 = (const char*)p2;
      return a1 < a2 + n2 && a2 < a1 + n1;
    };
    const size_t Fsz = (size_t)p->cfg.feature_size;
    if (w->fetch_out) {
      SCVAE_ARG(w->fetch_indptr && w->fetch_indices && w->fetch_values && w->fetch_rows &&
                w->fetch_n > 0 && w->fetch_features > 0 && w->fetch_ld >= w->fetch_features &&
                (w->fetch_as_u16 == 0 || w->fetch_as_u16 == 1) &&
                (w->fetch_row_values_out == nullptr || w->fetch_row_values));
      SCVAE_ARG(!w->fetch_tiles ||
                (w->fetch_as_u16 && w->fetch_tiles->entries && w->fetch_tiles->tile_ptr &&
                 w->fetch_tiles->block_ptr && w->fetch_tiles->capacity > 0 &&
                 w->fetch_features <= 65536 &&
                 (!a->count_tiles || a->count_tiles->entries != w->fetch_tiles->entries)));
      const size_t out_bytes =
          (size_t)w->fetch_n * (size_t)w->fetch_ld * (w->fetch_as_u16 ? 2 : 4);
      const size_t x_bytes = (size_t)a->cells * Fsz * sizeof(float);
      SCVAE_ARG(!overlaps(w->fetch_out, out_bytes, a->x, x_bytes) &&
                !overlaps(w->fetch_out, out_bytes, a->t, x_bytes) &&
                !overlaps(w->fetch_out, out_bytes, a->counts_u16,
                          (size_t)a->cells * (size_t)a->counts_ld * 2) &&
                !overlaps(w->fetch_row_values_out, (size_t)w->fetch_n * sizeof(float),
                          a->row_const, (size_t)a->cells * sizeof(float)));
    }
    if (w->noise_out) {
      SCVAE_ARG(w->noise_blocks >= 0 && w->noise_block_rows >= 0 && w->noise_cols > 0);
      // the output is a contiguous [blocks, block_rows, cols] tensor (noise_block_stride is the
      // row stride of the Philox FIELD, not of the output); the step reads one eps row per
      // (sample, cell), times the K passes of a GMVAE
      const size_t blocks = (size_t)(w->noise_blocks > 0 ? w->noise_blocks : 1);
      const size_t out_bytes =
          blocks * (size_t)w->noise_block_rows * (size_t)w->noise_cols * sizeof(float);
      const size_t samples = (size_t)(a->n_iw > 0 ? a->n_iw : 1) * (size_t)(a->n_mc > 0 ? a->n_mc : 1);
      const size_t passes =
          p->cfg.model_type == SCVAE_MODEL_GMVAE ? (size_t)(p->cfg.n_clusters > 0 ? p->cfg.n_clusters : 1) : 1;
      SCVAE_ARG(!overlaps(w->noise_out, out_bytes, a->eps,
                          passes * samples * (size_t)a->cells * (size_t)p->cfg.latent_size *
                              sizeof(float)));
    }
    p->side = w;
  }
  int rc;
  const bool probing = a->training && 2 * (size_t)p->probe_next + 1 < p->probe_events.size();
  if (probing)
    scvae::decoder_fused_set_probe(p->probe_events[2 * (size_t)p->probe_next],
                                   p->probe_events[2 * (size_t)p->probe_next + 1]);
  const bool staging = a->training && (size_t)p->stage_next < p->stage_recorded.size();
  if (staging)
    scvae::stage_probe_arm(&p->stage_events[(size_t)p->stage_next * scvae::PS_COUNT * 2],
                           &p->stage_recorded[(size_t)p->stage_next]);
  if (p->cfg.model_type == SCVAE_MODEL_GMVAE) {
    SCVAE_ARG(!a->deterministic_z);
    rc = scvae::gmvae_step(p, a, (hipStream_t)stream);
  } else {
    rc = scvae::vae_step(p, a, (hipStream_t)stream);
  }
  if (probing) {
    if (scvae::decoder_fused_probe_recorded()) ++p->probe_next;
    scvae::decoder_fused_set_probe(nullptr, nullptr);
  }
  if (rc) {
    // a failed step: the caller's stream still waits for what the second stream was given
    if (p->side_forked && hipEventRecord(p->side_join, p->side_stream) == hipSuccess)
      (void)hipStreamWaitEvent((hipStream_t)stream, p->side_join, 0);
    p->side = nullptr;
    if (staging) scvae::stage_probe_arm(nullptr, nullptr);
    return rc;
  }
  rc = scvae::plan_side_finish(p, (hipStream_t)stream);
  if (staging) {
    scvae::stage_probe_arm(nullptr, nullptr);
    ++p->stage_next;
  }
  if (!rc && p->ws_guards_dev) rc = scvae::ws_guard_check(p, (hipStream_t)stream);
  return rc;
}

int scvae_adam_clip_step(float* theta, float* grad, float* m, float* v, int64_t n,
                         float grad_scale, float lr_t, float beta1, float beta2, float epsilon,
                         void* stream) {
  SCVAE_ARG(n >= 0);
  return scvae::adam_clip_step((hipStream_t)stream, theta, grad, m, v, (size_t)n, grad_scale, lr_t,
                               beta1, beta2, epsilon);
}

int scvae_gemm(int32_t ta, int32_t tb, const float* A, const float* B, const float* bias, float* C,
               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t relu,
               int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  return scvae::gemm((hipStream_t)stream, ta != 0, tb != 0, A, B, bias, C, (int)M, (int)N, (int)K,
                     (int)lda, (int)ldb, (int)ldc, relu ? scvae::ACT_RELU : scvae::ACT_NONE,
                     accumulate != 0, (float*)workspace, (size_t)workspace_bytes);
}
int scvae_count_gemm(int32_t mode, const float* x, int64_t ldx, int64_t rows, int64_t cols,
                     const float* other, int64_t ld_other, int64_t N, const float* bias,
                     int32_t relu, float* C, int64_t ldc, void* workspace, int64_t workspace_bytes,
                     void* stream) {
  SCVAE_ARG(workspace_bytes >= 0);
  return scvae::count_gemm((hipStream_t)stream, mode, x, (int)ldx, (int)rows, (int)cols, other,
                           (int)ld_other, (int)N, bias, relu ? scvae::ACT_RELU : scvae::ACT_NONE,
                           C, (int)ldc, workspace, (size_t)workspace_bytes);
}
int64_t scvae_count_gemm_workspace_bytes(int32_t mode, int64_t rows, int64_t cols, int64_t N) {
  if (!scvae::count_gemm_supported((int)N)) return -1;
  return (int64_t)scvae::count_gemm_workspace_bytes(mode, (int)rows, (int)cols, (int)N);
}
int scvae_check_counts(const float* values, int64_t n, int32_t* bad, void* stream) {
  SCVAE_ARG(n >= 0);
  return scvae::check_counts((hipStream_t)stream, values, (size_t)n, bad);
}
int64_t scvae_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  return (int64_t)scvae::gemm_workspace_bytes((int)M, (int)N, (int)K);
}

int scvae_loglik_fwd(int32_t kind, const float* t, const float* const* pre, const float* row_const,
                     float* ll, int64_t rows, int64_t cells, int64_t F, void* stream) {
  SCVAE_ARG(pre && kind >= 0 && kind <= 3);
  scvae::HeadPtrs hp = {{nullptr, nullptr, nullptr}};
  for (int j = 0; j < scvae::likelihood_heads(kind); ++j) hp.p[j] = const_cast<float*>(pre[j]);
  return scvae::loglik_fwd((hipStream_t)stream, kind, t, (int)F, hp, (int)F, row_const, ll,
                           (int)rows, (int)cells, (int)F);
}
int scvae_loglik_bwd(int32_t kind, const float* t, float* const* pre, const float* gw,
                     const float* row_const, float* ll, int64_t rows, int64_t cells, int64_t F,
                     void* stream) {
  SCVAE_ARG(pre && kind >= 0 && kind <= 3);
  scvae::HeadPtrs hp = {{nullptr, nullptr, nullptr}};
  for (int j = 0; j < scvae::likelihood_heads(kind); ++j) hp.p[j] = pre[j];
  return scvae::loglik_bwd((hipStream_t)stream, kind, t, (int)F, hp, (int)F, gw, row_const, ll,
                           (int)rows, (int)cells, (int)F);
}
int64_t scvae_decoder_fused_workspace_bytes(int64_t rows, int64_t H, int64_t F) {
  if (!scvae::decoder_fused_train_supported(1, (int)H, 1)) return -1;
  return (int64_t)(scvae::decoder_fused_workspace_floats((int)rows, (int)H, (int)F, true) *
                   sizeof(float));
}
int32_t scvae_decoder_fused_variant(int32_t kind, int64_t H) {
  if (kind < 0 || (kind > 3 && kind != scvae::LK_BERNOULLI) ||
      !scvae::decoder_fused_supported((int)H))
    return 0;
  return scvae::decoder_fused_variant(scvae::likelihood_heads(kind), (int)H);
}
int32_t scvae_default_head_arith(void) { return scvae::default_head_arith(); }
int32_t scvae_default_dd_atomics(void) { return scvae::default_dd_atomics(); }
int32_t scvae_plan_dd_atomics(const scvae_plan* p) { return p ? p->dd_atomics : -1; }
int scvae_decoder_train_kernel_name(int32_t kind, int64_t H, int64_t rows, int32_t arith,
                                    int32_t u16, char* out, int64_t n) {
  SCVAE_ARG(out && n > 0);
  out[0] = 0;
  const int which = scvae_decoder_train_kernel(kind, H, arith);
  SCVAE_ARG(which > 0);
  const int P = scvae::likelihood_heads(kind);
  if (which == 3) {
    scvae::decoder_fused3_train_kernel_name(kind, (int)H, (int)rows, u16 != 0, out, (size_t)n,
                                            arith == 2 ? 6 : 9);
  } else if (which == 2) {
    snprintf(out, (size_t)n, "decoder_head2_kernel<%d, true, %s>", kind,
             (P <= 2 && H > 96 && H <= 111) ? "true|false" : "false");
  } else {
    snprintf(out, (size_t)n, "decoder_head_kernel<%d, true, %d>", kind, P >= 3 ? 32 : 64);
  }
  return 0;
}
int32_t scvae_decoder_train_kernel(int32_t kind, int64_t H, int32_t arith) {
  if (kind < 0 || (kind > 3 && kind != scvae::LK_BERNOULLI) || (arith < 0 || arith > 2) ||
      !scvae::decoder_fused_train_supported(scvae::likelihood_heads(kind), (int)H, arith))
    return 0;
  return scvae::decoder_train_kernel(scvae::likelihood_heads(kind), (int)H, arith);
}
static int decoder_fused_entry(int32_t kind, int32_t train, const float* d, int64_t rows, int64_t H,
                               const float* const* W, const float* const* b, float* const* dW,
                               float* const* db, int64_t F, scvae::Targets t, int64_t cells,
                               const float* gw, const float* row_const, float* ll, float* dd,
                               void* workspace, void* stream) {
  SCVAE_ARG(((kind >= 0 && kind <= 3) || kind == scvae::LK_BERNOULLI) && W && b);
  // bits 8, 9, 11 of `train`: the arithmetic of this call (none: the process default)
  const int arith_bits = train & (SCVAE_HEADS_FP32 | SCVAE_HEADS_BF16X9 | SCVAE_HEADS_BF16X6);
  SCVAE_ARG((train & ~0xF03) == 0 && (arith_bits & (arith_bits - 1)) == 0);
  const int dd_mode = (train & SCVAE_HEADS_DD_ATOMICS) ? 1 : 0;
  const int arith = (train & SCVAE_HEADS_FP32) ? 0
                    : (train & SCVAE_HEADS_BF16X9) ? 1
                    : (train & SCVAE_HEADS_BF16X6) ? 2 : scvae::default_head_arith();
  train &= 3;
  // (even widths up to 126: every arithmetic; the bf16x9 kernel's wider range -- odd widths, up
  //  to 256 -- for training and, its forward half, forward-only calls)
  SCVAE_ARG(scvae::decoder_fused_train_supported(scvae::likelihood_heads(kind), (int)H, arith));
  scvae::HeadParams hp;
  for (int j = 0; j < 3; ++j) {
    const bool on = j < scvae::likelihood_heads(kind);
    hp.W[j] = on ? W[j] : nullptr;
    hp.b[j] = on ? b[j] : nullptr;
    hp.dW[j] = (on && dW) ? dW[j] : nullptr;
    hp.db[j] = (on && db) ? db[j] : nullptr;
  }
  if (train) {
    SCVAE_ARG(dW && db);
    return scvae::decoder_fused_train((hipStream_t)stream, kind, d, (int)rows, (int)H, hp, (int)F,
                                      t, (int)cells, gw, row_const, ll, dd, (float*)workspace,
                                      arith, (train & 2) != 0, nullptr, dd_mode);
  }
  return scvae::decoder_fused_forward((hipStream_t)stream, kind, d, (int)rows, (int)H, hp, (int)F,
                                      t, (int)cells, row_const, ll, (float*)workspace, arith);
}
int scvae_decoder_fused(int32_t kind, int32_t train, const float* d, int64_t rows, int64_t H,
                        const float* const* W, const float* const* b, float* const* dW,
                        float* const* db, int64_t F, const float* t, int64_t cells,
                        const float* gw, const float* row_const, float* ll, float* dd,
                        void* workspace, void* stream) {
  return decoder_fused_entry(kind, train, d, rows, H, W, b, dW, db, F,
                             scvae::targets_f32(t, (int)F), cells, gw, row_const, ll, dd,
                             workspace, stream);
}
int scvae_decoder_fused_u16(int32_t kind, int32_t train, const float* d, int64_t rows, int64_t H,
                            const float* const* W, const float* const* b, float* const* dW,
                            float* const* db, int64_t F, const uint16_t* t, int64_t ldt,
                            int64_t cells, const float* gw, const float* row_const, float* ll,
                            float* dd, void* workspace, void* stream) {
  SCVAE_ARG(t && ldt >= (F + 63) / 64 * 64 && (ldt & 7) == 0 && ((uintptr_t)t & 15) == 0);
  return decoder_fused_entry(kind, train, d, rows, H, W, b, dW, db, F,
                             scvae::targets_u16(t, (int)ldt), cells, gw, row_const, ll, dd,
                             workspace, stream);
}
int scvae_likelihood_elementwise(int32_t kind, const float* t, const float* const* pre,
                                 float* log_prob, float* mean, float* variance, int64_t n,
                                 void* stream) {
  SCVAE_ARG(pre && ((kind >= 0 && kind <= 3) || kind == LK_BERNOULLI) && n >= 0);
  scvae::HeadPtrs hp = {{nullptr, nullptr, nullptr}};
  for (int j = 0; j < scvae::likelihood_heads(kind); ++j) hp.p[j] = const_cast<float*>(pre[j]);
  return scvae::loglik_elementwise((hipStream_t)stream, kind, t, hp, log_prob, mean, variance,
                                   (size_t)n);
}
int scvae_gauss_latent_fwd(const float* mu_pre, const float* ls_pre, const float* eps, float* z,
                           float* kl_elem, float* kl_cell, int64_t S, int64_t cells, int64_t L,
                           int32_t deterministic, void* stream) {
  return scvae::gauss_latent_fwd((hipStream_t)stream, mu_pre, ls_pre, eps, z, kl_elem, kl_cell,
                                 nullptr, (int)S, (int)cells, (int)L, deterministic);
}
int scvae_dropout_apply(const float* in, float* out, int64_t rows, int64_t cols, float keep,
                        uint64_t seed, int32_t site, int32_t accumulate, void* stream) {
  SCVAE_ARG(site >= 0 && cols > 0 && cols <= INT32_MAX);
  return scvae::dropout_apply((hipStream_t)stream, in, (int)cols, out, (int)cols, rows, (int)cols,
                              keep, seed, (uint32_t)site, accumulate);
}
int scvae_csr_minibatch(const int64_t* indptr, const int32_t* indices, const float* values,
                        const int64_t* rows, int64_t n, int64_t F, void* out, int64_t ld,
                        int32_t as_u16, const float* row_values, float* row_values_out,
                        void* stream) {
  SCVAE_ARG(as_u16 == 0 || as_u16 == 1);
  if (as_u16)
    return scvae::csr_densify_u16((hipStream_t)stream, indptr, indices, values, rows, (int)n,
                                  (int)F, static_cast<uint16_t*>(out), (int)ld, row_values,
                                  row_values_out);
  return scvae::csr_densify((hipStream_t)stream, indptr, indices, values, rows, (int)n, (int)F,
                            static_cast<float*>(out), (int)ld, row_values, row_values_out);
}

int scvae_csr_densify_u16(const int64_t* indptr, const int32_t* indices, const float* values,
                          const int64_t* rows, int64_t n, int64_t F, uint16_t* out, int64_t ld,
                          void* stream) {
  return scvae::csr_densify_u16((hipStream_t)stream, indptr, indices, values, rows, (int)n, (int)F,
                                out, (int)ld);
}

int scvae_count_gemm_u16(int32_t mode, const uint16_t* x, int64_t ldx, int64_t rows, int64_t cols,
                         const float* other, int64_t ld_other, int64_t N, const float* bias,
                         int32_t relu, float* C, int64_t ldc, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  SCVAE_ARG(workspace_bytes >= 0);
  return scvae::count_gemm_u16((hipStream_t)stream, mode, x, (int)ldx, (int)rows, (int)cols, other,
                               (int)ld_other, (int)N, bias,
                               relu ? scvae::ACT_RELU : scvae::ACT_NONE, C, (int)ldc, workspace,
                               (size_t)workspace_bytes);
}

int64_t scvae_count_tiles_padded(int64_t F) {
  return scvae::count_tiles_supported((int)F) && F > 0 && F <= 65536 ? scvae::count_tiles_padded((int)F) : -1;
}
int scvae_csr_row_entries(const int64_t* indptr, const float* values, int64_t n_rows, int32_t* out,
                          void* stream) {
  return scvae::csr_row_entries((hipStream_t)stream, indptr, values, n_rows, out);
}
int scvae_csr_count_tiles(const int64_t* indptr, const int32_t* indices, const float* values,
                          const int64_t* rows, int64_t n, int64_t F,
                          const scvae_count_tiles* tiles, void* stream) {
  SCVAE_ARG(tiles && n >= 0 && n <= INT32_MAX && F > 0 && F <= 65536);
  return scvae::csr_count_tiles((hipStream_t)stream, indptr, indices, values, rows, (int)n, (int)F,
                                scvae::count_tiles_of(tiles));
}
int scvae_count_gemm_tiles(int32_t mode, const scvae_count_tiles* tiles, const uint16_t* x,
                           int64_t ldx, int64_t rows, int64_t cols, const float* other,
                           int64_t ld_other, int64_t N, const float* bias, int32_t relu, float* C,
                           int64_t ldc, void* workspace, int64_t workspace_bytes, void* stream) {
  SCVAE_ARG(tiles && workspace_bytes >= 0);
  return scvae::count_gemm_tiles((hipStream_t)stream, mode, scvae::count_tiles_of(tiles), x, (int)ldx,
                                 (int)rows, (int)cols, other, (int)ld_other, (int)N, bias,
                                 relu ? scvae::ACT_RELU : scvae::ACT_NONE, C, (int)ldc, workspace,
                                 (size_t)workspace_bytes);
}

int scvae_csr_densify(const int64_t* indptr, const int32_t* indices, const float* values,
                      const int64_t* rows, int64_t n, int64_t F, float* out, void* stream) {
  return scvae::csr_densify((hipStream_t)stream, indptr, indices, values, rows, (int)n, (int)F, out,
                            (int)F);
}
int scvae_csr_row_lgamma1p(const int64_t* indptr, const float* values, int64_t n_rows, float* out,
                           void* stream) {
  return scvae::csr_row_lgamma1p((hipStream_t)stream, indptr, values, n_rows, out);
}
int scvae_gather_rows(const float* src, const int64_t* rows, int64_t n, float* out, void* stream) {
  return scvae::gather_rows_f32((hipStream_t)stream, src, rows, (int)n, out);
}
int scvae_bn_merge(const float* gathered, const int64_t* counts, int64_t ranks, int64_t n,
                   float* out, void* stream) {
  return scvae::bn_merge((hipStream_t)stream, gathered, counts, (int)ranks, (int)n, out);
}
int64_t scvae_bn_workspace_floats(int64_t N) {
  return N > 0 ? (int64_t)scvae::bn_partial_floats(1, (int)N) : -1;
}
int scvae_bn_stats(const float* a, int64_t lda, int64_t rows, int64_t N, float* mean, float* var,
                   float* workspace, void* stream) {
  SCVAE_ARG(rows > 0 && rows <= INT32_MAX && N > 0 && lda >= N);
  return scvae::bn_stats((hipStream_t)stream, a, (int)lda, (int)rows, 1, (int)N, mean, var,
                         workspace);
}
int scvae_bn_apply_relu_fwd(const float* a, int64_t lda, const float* mean, const float* var,
                            const float* beta, float* h, int64_t ldh, int64_t rows, int64_t N,
                            int32_t relu, void* stream) {
  SCVAE_ARG(a && mean && var && beta && h && rows >= 0 && N > 0 && lda >= N && ldh >= N);
  if (rows == 0) return 0;
  return scvae::bn_apply((hipStream_t)stream, a, (int)lda, mean, var, (int)N, beta, h, (int)ldh,
                         (int)rows, 1, (int)N, relu ? 1 : 0);
}
int scvae_bn_apply_relu_bwd(const float* dh, int64_t lddh, const float* h, int64_t ldh,
                            const float* a, int64_t lda, const float* mean, const float* var,
                            int64_t rows, int64_t N, int32_t relu, float* da, int64_t ldda,
                            float* dbeta, float* workspace, void* stream) {
  SCVAE_ARG(dh && h && a && mean && var && da && dbeta && workspace && rows > 0 && N > 0);
  SCVAE_ARG(lddh >= N && ldh >= N && lda >= N && ldda >= N && rows <= INT32_MAX);
  float* s1 = workspace;
  float* s2 = workspace + N;
  float* partial = workspace + 2 * N;
  int rc = scvae::bn_bwd_stats((hipStream_t)stream, dh, (int)lddh, h, (int)ldh, a, (int)lda, mean,
                               var, (int)rows, 1, (int)N, relu ? 1 : 0, s1, s2, partial, dbeta,
                               nullptr, nullptr, rows);
  if (rc) return rc;
  return scvae::bn_bwd_apply((hipStream_t)stream, dh, (int)lddh, h, (int)ldh, a, (int)lda, mean,
                             var, s1, s2, (int)rows, 1, (int)N, relu ? 1 : 0, 1.f / (float)rows,
                             da, (int)ldda);
}
int scvae_softplus_gaussian_logprob_pair_fwd(const float* qm, const float* qs, const float* Wpm,
                                             const float* bpm, const float* Wps,
                                             const float* bps, const float* eps, float* z,
                                             float* klz, float* qvar, int64_t K, int64_t S,
                                             int64_t B, int64_t L, void* stream) {
  SCVAE_ARG(K > 0 && S > 0 && B >= 0 && K <= 65535 && B <= INT32_MAX);
  return scvae::softplus_gaussian_fwd((hipStream_t)stream, qm, qs, Wpm, bpm, Wps, bps, eps, z,
                                      klz, qvar, (int)K, (int)S, (int)B, (int)L);
}
int scvae_softplus_gaussian_logprob_pair_bwd(const float* qm, const float* qs, const float* Wpm,
                                             const float* bpm, const float* Wps,
                                             const float* bps, const float* eps, const float* dz,
                                             const float* gklz, float* dqm, float* dqs,
                                             float* dprior, int64_t K, int64_t S, int64_t B,
                                             int64_t L, void* stream) {
  SCVAE_ARG(Wpm && bpm && Wps && bps && K > 0 && S > 0 && B >= 0 && L > 0);
  return scvae::softplus_gaussian_bwd((hipStream_t)stream, qm, qs, Wpm, bpm, Wps, bps, eps, dz,
                                      gklz, dqm, dqs, dprior, (int)K, (int)S, (int)B, (int)L);
}
int scvae_categorical_entropy_kl_fwd(const float* logits, float* y, float* kl_y_cell, int64_t B,
                                     int64_t K, const float* prior_logits, void* stream) {
  SCVAE_ARG(logits && y && kl_y_cell && B >= 0 && K > 0 && B <= INT32_MAX);
  return scvae::categorical_fwd((hipStream_t)stream, logits, y, kl_y_cell, (int)B, (int)K,
                                prior_logits);
}
int scvae_categorical_entropy_kl_bwd(const float* y, const float* dy, const float* gate, float c,
                                     float* dlogits, int64_t B, int64_t K,
                                     const float* prior_logits, void* stream) {
  SCVAE_ARG(y && dy && gate && dlogits && B >= 0 && K > 0 && B <= INT32_MAX);
  return scvae::categorical_bwd_gated((hipStream_t)stream, y, dy, gate, c, dlogits, (int)B,
                                      (int)K, prior_logits);
}
int scvae_iw_logmeanexp(const float* ll, const float* kl_cell, int32_t kl_per_sample,
                        int32_t n_iw, int32_t n_mc, int64_t B, float kl_weight, float row_scale,
                        float* scalars, float* gw, void* stream) {
  SCVAE_ARG(ll && kl_cell && scalars && n_iw > 0 && n_mc > 0 && B > 0 && B <= INT32_MAX);
  return scvae::vae_elbo((hipStream_t)stream, ll, kl_cell, kl_per_sample ? 1 : 0, n_iw, n_mc,
                         (int)B, kl_weight, row_scale, scalars, gw);
}
int scvae_pxmean_stats(int32_t kind, const float* const* pre, int64_t S, int64_t B, int64_t F,
                       const float* weight, int64_t ldw, int32_t accumulate, float* p_x_mean,
                       float* mean_of_var, float* var_of_mean, void* stream) {
  SCVAE_ARG(pre && ((kind >= 0 && kind <= 3) || kind == LK_BERNOULLI) && S > 0 && B >= 0 && F > 0);
  scvae::HeadPtrs hp = {{nullptr, nullptr, nullptr}};
  for (int j = 0; j < scvae::likelihood_heads(kind); ++j) hp.p[j] = const_cast<float*>(pre[j]);
  return scvae::px_statistics((hipStream_t)stream, kind, hp, (int)F, (int)S, (int)B, (int)F,
                              weight, (int)ldw, accumulate ? 1 : 0, p_x_mean, mean_of_var,
                              var_of_mean);
}
int scvae_philox_normal(float* out, int64_t rows, int64_t cols, int64_t row_offset, uint64_t seed,
                        uint64_t stream_id, void* stream) {
  return scvae::philox_normal((hipStream_t)stream, out, rows, (int)cols, row_offset, seed,
                              stream_id);
}
int scvae_philox_normal_blocks(float* out, int64_t blocks, int64_t block_rows, int64_t cols,
                               int64_t block_stride, int64_t row_offset, uint64_t seed,
                               uint64_t stream_id, void* stream) {
  SCVAE_ARG(blocks >= 0 && block_rows >= 0);
  return scvae::philox_normal((hipStream_t)stream, out, blocks * block_rows, (int)cols, row_offset,
                              seed, stream_id, block_rows, block_stride);
}

}  // extern "C"
