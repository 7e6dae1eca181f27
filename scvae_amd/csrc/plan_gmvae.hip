// Execution plan of the Gaussian-mixture VAE: y is marginalised by K passes through the
// q(z|x,y) encoder and the decoder with shared weights
// (scvae/models/gaussian_mixture_variational_autoencoder.py:2788-3434).  The K passes run as
// ONE batch of K*S*B rows with per-pass (grouped) batch-norm statistics, and the first
// q(z|x,y=k) layer computes x*W_x once and adds row W_y[k] per pass (the reference recomputes
// x*W_x K times).
#include <math.h>

#include "plan.hpp"

namespace scvae {

int build_gmvae(scvae_plan* p) {
  const scvae_model_config& c = p->cfg;
  const bool bn = c.batch_norm != 0;
  const int K = c.n_clusters, Lz = c.latent_size, F = c.feature_size;
  Layout& L = p->layout;
  char scope[96];
  int n_in = F;
  // learned p(y): tf.Variable "LOGITS" in scope Y/P, the first variable of the graph (gm:2799-2803)
  if (c.prior_mode == 2) p->prior_off = L.add("Y/P/LOGITS", K, 0);
  for (int i = 0; i < c.n_hidden; ++i) {
    snprintf(scope, sizeof scope, "Y/CATEGORICAL/ENCODER/LAYER_%d", i + 1);
    p->yenc.push_back(L.dense(scope, n_in, c.hidden[i], bn));
    n_in = c.hidden[i];
  }
  p->ylogits = L.dense("Y/CATEGORICAL/LOGITS", n_in, K, false);
  n_in = F + K;
  for (int i = 0; i < c.n_hidden; ++i) {
    snprintf(scope, sizeof scope, "Z/Q/ENCODER/LAYER_%d", i + 1);
    p->zenc.push_back(L.dense(scope, n_in, c.hidden[i], bn));
    n_in = c.hidden[i];
  }
  // the scope is the upper-cased distribution name (gm:2962-2963, 3013-3014): "softplus gaussian",
  // or its alias "modified gaussian" of the "legacy gaussian mixture" (du:307, 349-352)
  const std::string dist = (c.latent_mode & 4) ? "MODIFIED_GAUSSIAN" : "SOFTPLUS_GAUSSIAN";
  p->qmean = L.dense("Z/Q/" + dist + "/MEAN", n_in, Lz, false);
  p->qscale = L.dense("Z/Q/" + dist + "/SOFTPLUS_SCALE", n_in, Lz, false);
  p->pmean = L.dense("Z/P/" + dist + "/MEAN", K, Lz, false);
  p->pscale = L.dense("Z/P/" + dist + "/SOFTPLUS_SCALE", K, Lz, false);
  n_in = Lz + c.decoder_extra;   // decoder input [z | batch one-hot | count sum]
  // gm:3135-3146: hidden_sizes[::-1] without reverse_order => LAYER_1.. in execution order
  for (int i = 0; i < c.n_hidden; ++i) {
    snprintf(scope, sizeof scope, "X/DECODER/LAYER_%d", i + 1);
    p->xdec.push_back(L.dense(scope, n_in, c.hidden[c.n_hidden - 1 - i], bn));
    n_in = c.hidden[c.n_hidden - 1 - i];
  }
  for (int j = 0; j < p->P; ++j) {
    snprintf(scope, sizeof scope, "X/DISTRIBUTION/%s", head_names(c.likelihood, j));
    p->heads[j] = L.dense(scope, n_in, F, false);
  }
  if (c.k_max > 0) p->head_k = L.dense("X/DISTRIBUTION/P_K", n_in, F * (c.k_max + 1), false);
  if (c.prior_mode == 1) {
    // custom p(y) (a tf.constant in the reference): K fixed logits in a slot of the parameter
    // buffer that is not listed among the variables; its gradient slot stays zero
    p->prior_off = L.n_params;
    L.n_params += ((size_t)K + ALIGN_FLOATS - 1) / ALIGN_FLOATS * ALIGN_FLOATS;
  }
  // dropout of the input connections (gm:2946-2953, 2978-2984, 3034-3040, 3058-3065, 3080-3086,
  // 3137-3143, 3167-3172, 3197-3202); sites: include/scvae_hip.h, scvae_dropout_apply
  const float kh = dropout_keep(c, 0), kx = dropout_keep(c, 1), kz = dropout_keep(c, 2);
  const float ky = dropout_keep(c, 3);
  for (size_t i = 0; i < p->yenc.size(); ++i) {
    p->yenc[i].keep = i == 0 ? kx : kh;
    p->yenc[i].site = 64 + (uint32_t)i;
  }
  p->ylogits.keep = kh; p->ylogits.site = 80;
  for (size_t i = 0; i < p->zenc.size(); ++i) {
    p->zenc[i].keep = i == 0 ? kx : kh;
    p->zenc[i].site = (uint32_t)i;
  }
  p->qmean.keep = kh; p->qmean.site = 16;
  p->qscale.keep = kh; p->qscale.site = 17;
  p->pmean.keep = ky; p->pmean.site = 24;
  p->pscale.keep = ky; p->pscale.site = 25;
  for (size_t i = 0; i < p->xdec.size(); ++i) {
    p->xdec[i].keep = i == 0 ? kz : kh;
    p->xdec[i].site = 32 + (uint32_t)i;
  }
  for (int j = 0; j < p->P; ++j) { p->heads[j].keep = kh; p->heads[j].site = 48 + j; }
  p->head_k.keep = kh; p->head_k.site = 51;
  return 0;
}

size_t carve_gmvae(scvae_plan* p, void* base, size_t cap, int64_t cells, int64_t samples,
                   bool dry) {
  const scvae_model_config& c = p->cfg;
  Bump b(base, cap, dry);
  if (!dry && workspace_guard_on()) { p->ws_guards.clear(); b.guards = &p->ws_guards; }
  const size_t K = c.n_clusters, B = (size_t)cells, KB = K * B, R = K * B * samples;
  const size_t Lz = c.latent_size, F = c.feature_size;
  size_t hmax = Lz > K ? Lz : K;
  size_t gws = 0;
  auto track = [&](size_t M, size_t N, size_t Kd) {
    const size_t w = gemm_workspace_bytes((int)M, (int)N, (int)Kd);
    if (w > gws) gws = w;
  };
  auto layer_ws = [&](Dense& d, size_t rows, size_t groups, size_t n_in_eff) {
    float* a = d.bn ? b.floats(rows * d.n_out) : nullptr;
    float* h = b.floats(rows * d.n_out);
    float* st = d.bn ? b.floats(4 * groups * (size_t)d.n_out) : nullptr;
    if (!dry) { d.a = a; d.h = h; d.stats = st; }
    if ((size_t)d.n_out > hmax) hmax = d.n_out;
    track(rows, d.n_out, n_in_eff);
    track(n_in_eff, d.n_out, rows);
    track(rows, n_in_eff, d.n_out);
  };
  for (auto& d : p->yenc) layer_ws(d, B, 1, d.n_in);
  for (size_t i = 0; i < p->zenc.size(); ++i)
    layer_ws(p->zenc[i], KB, K, i == 0 ? F : (size_t)p->zenc[i].n_in);
  for (auto& d : p->xdec) layer_ws(d, R, K, d.n_in);
  {   // tilechain.hip (one launch per hidden layer and direction for the K stacked passes)
    float* q[4 + TC_MAX_JOBS];
    for (int i = 0; i < 4; ++i) q[i] = b.floats(tile_chain_part_floats((int)R));
    for (int i = 0; i < TC_MAX_JOBS; ++i) q[4 + i] = b.floats(tile_chain_slab_floats((int)R));
    if (!dry) {
      p->tc_part[0] = q[0]; p->tc_part[1] = q[1]; p->tc_spart[0] = q[2]; p->tc_spart[1] = q[3];
      for (int i = 0; i < TC_MAX_JOBS; ++i) p->tc_slab[i] = q[4 + i];
    }
  }
  const size_t h1z = p->zenc.empty() ? F : (size_t)p->zenc[0].n_out;
  track(B, h1z, F); track(F, h1z, B);
  if (dropout_keep(c, 1) > 0.f) { track(KB, h1z, F + K); track(F + K, h1z, KB); }
  float* logits = b.floats(B * K);
  float* yprob = b.floats(B * K);
  float* kl_y_cell = b.floats(B);
  float* a0 = b.floats(B * h1z);
  float* qm = b.floats(KB * Lz);
  float* qs = b.floats(KB * Lz);
  float* z = b.floats(R * Lz);
  float* klz = b.floats(R);
  float* gklz = b.floats(R);
  float* ll = b.floats(R);
  float* gw = b.floats(R);
  float* dy = b.floats(B * K);
  float* dlogits = b.floats(B * K);
  float* dqm = b.floats(KB * Lz);
  float* dqs = b.floats(KB * Lz);
  float* dprior = b.floats(KB * 2 * Lz);
  float* kl_cell = b.floats(B + 64);  // rec_cell / scalar sums scratch
  float* pre[3] = {nullptr, nullptr, nullptr};
  for (int j = 0; j < p->P; ++j) pre[j] = b.floats(R * F);
  float* d0 = b.floats(R * hmax);
  float* d1 = b.floats(R * hmax);
  float* d2 = b.floats(R * hmax);
  float* dz = b.floats(R * Lz);
  float* mov = b.floats(B * F);
  float* vom = b.floats(B * F);
  float* sum_scratch = b.floats(B * hmax + KB * Lz);
  const int hn = c.n_hidden ? c.hidden[c.n_hidden - 1] : (int)F;
  const int h1 = c.n_hidden ? c.hidden[0] : (int)Lz;
  track(B, K, hn); track(hn, K, B); track(B, hn, K);
  track(KB, Lz, hn); track(hn, Lz, KB); track(KB, hn, Lz);
  track(R, F, h1); track(h1, F, R); track(R, h1, F);
  if (c.k_max > 0) {
    const size_t FC = F * (size_t)(c.k_max + 1);
    track(R, FC, h1); track(h1, FC, R); track(R, h1, FC);
  }
  {   // products with the count matrix x itself (count_gemm.hip): q(y|x) and q(z|x,y) layer 1
    size_t w = plan_x_gemm_workspace_bytes((int)B, (int)F, (int)h1z);
    if (!p->yenc.empty()) {
      const size_t wy = plan_x_gemm_workspace_bytes((int)B, (int)F, p->yenc[0].n_out);
      if (wy > w) w = wy;
    }
    if (w > gws) gws = w;
  }
  float* gemm_ws = gws ? b.floats(gws / sizeof(float)) : nullptr;
  size_t pmax = col_sum_partial_floats((int)(F > hmax ? F : hmax));
  if (c.k_max > 0) pmax = col_sum_partial_floats((int)(F * (size_t)(c.k_max + 1)));
  {
    const size_t q = bn_partial_floats((int)K, (int)(hmax > 2 * Lz ? hmax : 2 * Lz));
    if (q > pmax) pmax = q;
  }
  float* partial = b.floats(pmax);
  float* fused_ws = decoder_fused_train_supported(p->P, h1, 1)
                        ? b.floats(decoder_fused_workspace_floats((int)R, h1, (int)F, true))
                        : nullptr;
  const size_t E = (size_t)c.decoder_extra;
  float* zcat = E ? b.floats(R * (Lz + E)) : nullptr;
  float* dzcat = E ? b.floats(R * (Lz + E)) : nullptr;
  // (the class logits of the unfused path; the fused -k launches keep the second launch's ll and
  //  dd there -- more than the logits when there are fewer genes than hidden units)
  size_t pre_k_floats = R * F * (size_t)(c.k_max + 1);
  if (pre_k_floats < decoder_fused_cat_scratch_floats((int)R, p->heads[0].n_in))
    pre_k_floats = decoder_fused_cat_scratch_floats((int)R, p->heads[0].n_in);
  float* pre_k = c.k_max > 0 ? b.floats(pre_k_floats) : nullptr;
  // dropped-out layer inputs of the training pass (kept for the weight gradients); the p(z|y)
  // layers keep their effective (row-scaled) weights [K, L] there
  auto drop_ws = [&](Dense& d, size_t rows) {
    float* q = d.keep > 0.f ? b.floats(rows * (size_t)d.n_in) : nullptr;
    if (!dry) d.in_drop = q;
  };
  for (auto& d : p->yenc) drop_ws(d, B);
  drop_ws(p->ylogits, B);
  for (auto& d : p->zenc) drop_ws(d, KB);   // zenc[0]: [K*B, F + K]
  drop_ws(p->qmean, KB);
  drop_ws(p->qscale, KB);
  { Dense& d = p->pmean; float* q = d.keep > 0.f ? b.floats(K * Lz) : nullptr; if (!dry) d.in_drop = q; }
  { Dense& d = p->pscale; float* q = d.keep > 0.f ? b.floats(K * Lz) : nullptr; if (!dry) d.in_drop = q; }
  for (auto& d : p->xdec) drop_ws(d, R);
  for (int j = 0; j < p->P; ++j) drop_ws(p->heads[j], R);
  if (c.k_max > 0) drop_ws(p->head_k, R);
  if (!dry) {
    p->fused_ws = fused_ws;
    p->zcat = zcat; p->dzcat = dzcat;
    p->pre_k = pre_k;
    p->logits = logits; p->yprob = yprob; p->kl_y_cell = kl_y_cell; p->a0 = a0;
    p->qm = qm; p->qs = qs; p->z = z; p->klz = klz; p->gklz = gklz; p->ll = ll; p->gw = gw;
    p->dy = dy; p->dlogits = dlogits; p->dqm = dqm; p->dqs = dqs; p->dprior = dprior;
    p->kl_cell = kl_cell;
    for (int j = 0; j < 3; ++j) p->pre[j] = pre[j];
    p->dbuf[0] = d0; p->dbuf[1] = d1; p->dbuf[2] = d2; p->dz = dz;
    p->mov = mov; p->vom = vom; p->sum_scratch = sum_scratch;
    p->gemm_ws = gemm_ws; p->gemm_ws_bytes = gws; p->partial = partial;
  }
  return b.used;
}

// ---- the K stacked passes through q(z|x,y) and p(x|z,y) with one launch per hidden layer and
//      direction (tilechain.hip with groups: the rows of pass k are a range of 64-row tiles with
//      their own batch statistics).  Training steps with batch normalisation, no dropout, whole
//      tiles per pass, layers and latent at most 128 wide.  Under a data-parallel hook
//      (scvae_plan_set_sync) the statistics of a layer are those of the global minibatch: the
//      rank's chunks are merged per pass by a small kernel, the hook merges the ranks (the K
//      passes' statistics in ONE collective, as the launch chain issues it), and the consuming
//      tile kernel takes them as given -- gm_tile_bn_forward / gm_tile_bn_backward, the VAE's
//      tile_bn_forward / tile_bn_backward (plan.hip) with groups. ----
bool gm_tile_chain_ok(const scvae_plan* p, int B, int S, bool training) {
  const scvae_model_config& c = p->cfg;
  static const bool env_on = [] { const char* e = getenv("SCVAE_TILE_CHAIN"); return !(e && e[0] == '0'); }();
  if (!env_on || !p->use_tile_chain || !training || !c.batch_norm) return false;
  if (p->zenc.empty() || p->xdec.empty() || c.decoder_extra != 0 || c.latent_size > 128) return false;
  if (B % 64 != 0 || ((int64_t)B * S) % 64 != 0) return false;
  for (const auto& d : p->zenc) if (d.n_out > 128 || !d.bn) return false;
  for (const auto& d : p->xdec) if (d.n_out > 128 || !d.bn) return false;
  for (int i = 0; i < 4; ++i) if (dropout_keep(c, i) > 0.f) return false;
  return p->tc_part[0] != nullptr;
}
// the batch norm of layer d (K groups of `group_rows` rows) as the tile kernels see it
static TileBN gm_tile_bn(scvae_plan* p, Dense& d, int K, int group_rows, const float* part,
                         float* part_out) {
  TileBN t;
  const int N = d.n_out;
  t.a = d.a; t.h = d.h; t.beta = p->params + d.beta;
  t.mean = d.stats; t.var = d.stats + (size_t)K * N;
  t.s1 = d.stats + 2 * (size_t)K * N; t.s2 = d.stats + 3 * (size_t)K * N;
  t.part = part; t.chunks = K * (group_rows / 64); t.chunk = 64; t.part_out = part_out;
  t.group_tiles = group_rows / 64; t.groups = K;
  t.dbeta = p->grads ? p->grads + d.beta : nullptr;
  t.mov_mean = p->moving + d.mov_mean; t.mov_var = p->moving + d.mov_var;
  return t;
}

// the batch norm of layer d for the tile kernel that CONSUMES its forward statistics: merged by
// that kernel from the chunk statistics `part` (single process), or -- data parallel -- merged
// here per pass, over the ranks by the hook, and handed over as given
static int gm_tile_bn_forward(scvae_plan* p, hipStream_t s, Dense& d, int K, int group_rows,
                              const float* part, TileBN* out) {
  if (!p->sync) {
    *out = gm_tile_bn(p, d, K, group_rows, part, nullptr);
    return 0;
  }
  const int N = d.n_out;
  TileBN t = gm_tile_bn(p, d, K, group_rows, nullptr, nullptr);
  int rc = tile_stats_merge(s, part, group_rows / 64, 64, group_rows, N, t.mean, t.var, K);
  if (rc) return rc;
  if (p->sync(p->sync_user, d.stats, 2 * (int64_t)K * N, 1, group_rows)) {
    set_error("batch-norm sync hook failed");
    return -2;
  }
  *out = t;
  return 0;
}
// ... and for the tile kernel that consumes its backward sums (s1, s2: [K][N] behind mean / var)
static int gm_tile_bn_backward(scvae_plan* p, hipStream_t s, Dense& d, int K, int group_rows,
                               const float* part, float bessel, TileBN* out) {
  if (!p->sync) {
    *out = gm_tile_bn(p, d, K, group_rows, part, nullptr);
    return 0;
  }
  const int N = d.n_out;
  TileBN t = gm_tile_bn(p, d, K, group_rows, nullptr, nullptr);
  int rc = tile_sums_merge(s, part, group_rows / 64, N, t, bessel, K);
  if (rc) return rc;
  if (p->sync(p->sync_user, t.s1, 2 * (int64_t)K * N, 0, group_rows)) {
    set_error("batch-norm backward sync hook failed");
    return -2;
  }
  *out = t;
  return 0;
}

int gmvae_step(scvae_plan* p, const scvae_step_args* a, hipStream_t s) {
  const scvae_model_config& c = p->cfg;
  const int K = c.n_clusters, B = (int)a->cells, S = a->n_iw * a->n_mc;
  const int KB = K * B, SB = S * B, R = K * SB;
  const int F = c.feature_size, L = c.latent_size;
  const bool training = a->training != 0;
  const int64_t GB = a->global_cells > 0 ? a->global_cells : a->cells;
  const float w = a->warm_up_weight * c.kl_weight;
  const float inv_gb = 1.f / (float)GB;
  int rc;
  p->drop_seed = a->dropout_seed;
#define GEMM(...)                                                              \
  do {                                                                         \
    if ((rc = plan_gemm(p, s, __VA_ARGS__))) return rc;                        \
  } while (0)
#define TRY(call)                  \
  do {                             \
    if ((rc = (call))) return rc;  \
  } while (0)

  // ---------------- q(y|x) (gm:3050-3092) ----------------
  // (the fp32 batch, or the token of the uint16 one: plan_gemm hands that to the count kernels)
  const float* h = p->step_x;
  int ld = F;
  for (auto& d : p->yenc) {
    TRY(dense_forward(p, s, d, h, ld, B, 1, true, training));
    h = d.h; ld = d.n_out;
  }
  const float* hy = h;
  int ldy = ld;
  TRY(dense_input(p, s, p->ylogits, h, ld, B, training, &hy, &ldy));
  GEMM(false, false, hy, p->params + p->ylogits.w, p->params + p->ylogits.b, p->logits, B, K,
       p->ylogits.n_in, ldy, K, K, ACT_NONE, false);
  const float* prior = p->prior_off != NPOS ? p->params + p->prior_off : nullptr;
  TRY(categorical_fwd(s, p->logits, p->yprob, p->kl_y_cell, B, K, prior));
  if (a->q_y_logits) TRY(copy(s, p->logits, a->q_y_logits, (size_t)B * K));

  // ---------------- q(z|x,y=k), all k (gm:2936-3007) ----------------
  const bool tile = gm_tile_chain_ok(p, B, S, training);
  int tcur = 0;     // (ping-pong of the tile chain's chunk statistics)
  const float* hz = p->step_x;
  int ldz = F;
  for (size_t i = 0; i < p->zenc.size(); ++i) {
    Dense& d = p->zenc[i];
    if (i == 0 && training && d.keep > 0.f) {
      // every pass drops its own elements of [x | one-hot]: K separate inputs, one GEMM
      if (p->x_u16) {
        set_error("dropout on the input layer needs the fp32 minibatch");
        return -1;
      }
      TRY(tile_onehot(s, a->x, d.in_drop, K, B, F));
      TRY(dropout_apply(s, d.in_drop, F + K, d.in_drop, F + K, KB, F + K, d.keep, p->drop_seed,
                        d.site, 0, p->drop_rows));
      TRY(dense_affine(p, s, d, d.in_drop, F + K, KB, K, true, training));
    } else if (i == 0) {
      // x*W[:F] + b once, then + W[F+k] per pass
      const float* W = p->params + d.w;
      GEMM(false, false, p->step_x, W, p->params + d.b, p->a0, B, d.n_out, F, F, d.n_out, d.n_out,
           ACT_NONE, false);
      float* target = d.bn ? d.a : d.h;
      TRY(add_group_rows(s, p->a0, W + (size_t)F * d.n_out, target, K, B, d.n_out, d.bn ? 0 : 1));
      if (d.bn) {
        const int N = d.n_out;
        if (training) {
          float* mean = d.stats;
          float* var = d.stats + (size_t)K * N;
          TRY(bn_stats(s, d.a, N, B, K, N, mean, var, p->partial));
          if (p->sync && p->sync(p->sync_user, d.stats, 2 * (int64_t)K * N, 1, B)) {
            set_error("batch-norm sync hook failed");
            return -2;
          }
          TRY(bn_apply(s, d.a, N, mean, var, N, p->params + d.beta, d.h, N, B, K, N, 1));
        } else {
          TRY(bn_apply(s, d.a, N, p->moving + d.mov_mean, p->moving + d.mov_var, 0,
                       p->params + d.beta, d.h, N, KB, 1, N, 1));
        }
      }
    } else if (tile) {
      // one launch: normalise the layer below (i >= 2; layer 1 reads the finished h of layer 0),
      // the product with this layer's weights, the chunk statistics of its output
      TileFwdArgs q;
      q.rows = KB; q.K = d.n_in;
      if (i == 1) { q.x = p->zenc[0].h; q.ldx = d.n_in; }
      else TRY(gm_tile_bn_forward(p, s, p->zenc[i - 1], K, B, p->tc_part[tcur], &q.bn));
      q.n_out = 1;
      q.o[0].W = p->params + d.w; q.o[0].b = p->params + d.b; q.o[0].out = d.a;
      q.o[0].part = p->tc_part[i == 1 ? tcur : tcur ^ 1]; q.o[0].N = d.n_out;
      TRY(tile_forward(s, q));
      if (i > 1) tcur ^= 1;
    } else {
      TRY(dense_forward(p, s, d, hz, ldz, KB, K, true, training));
    }
    hz = d.h; ldz = d.n_out;
  }
  if (p->zenc.empty()) {
    set_error("GMVAE needs at least one hidden layer");
    return -1;
  }
  const float* hz_m = hz;
  const float* hz_s = hz;
  int ldz_m = ldz, ldz_s = ldz;
  if (tile) {
    // the two posterior heads on the (here normalised) output of the last q(z|x,y) layer
    TileFwdArgs q;
    q.rows = KB; q.K = p->zenc.back().n_out;
    if (p->zenc.size() == 1) { q.x = p->zenc[0].h; q.ldx = q.K; }
    else TRY(gm_tile_bn_forward(p, s, p->zenc.back(), K, B, p->tc_part[tcur], &q.bn));
    q.n_out = 2;
    q.o[0].W = p->params + p->qmean.w; q.o[0].b = p->params + p->qmean.b; q.o[0].out = p->qm;
    q.o[0].N = L;
    q.o[1].W = p->params + p->qscale.w; q.o[1].b = p->params + p->qscale.b; q.o[1].out = p->qs;
    q.o[1].N = L;
    TRY(tile_forward(s, q));
  } else {
  TRY(dense_input(p, s, p->qmean, hz, ldz, KB, training, &hz_m, &ldz_m));
  TRY(dense_input(p, s, p->qscale, hz, ldz, KB, training, &hz_s, &ldz_s));
  GEMM(false, false, hz_m, p->params + p->qmean.w, p->params + p->qmean.b, p->qm, KB, L,
       p->qmean.n_in, ldz_m, L, L, ACT_NONE, false);
  GEMM(false, false, hz_s, p->params + p->qscale.w, p->params + p->qscale.b, p->qs, KB, L,
       p->qscale.n_in, ldz_s, L, L, ACT_NONE, false);
  }
  const float* Wpm = p->params + p->pmean.w;
  const float* bpm = p->params + p->pmean.b;
  const float* Wps = p->params + p->pscale.w;
  const float* bps = p->params + p->pscale.b;
  // p(z|y=k): a dense layer on the one-hot (gm:3009-3048); its dropout keeps or drops the one
  // non-zero input of pass k, i.e. scales row k of the weights
  const bool prior_drop = training && p->pmean.keep > 0.f;
  if (prior_drop) {
    TRY(dropout_scale_rows(s, Wpm, p->pmean.in_drop, K, L, p->pmean.keep, p->drop_seed,
                           p->pmean.site));
    TRY(dropout_scale_rows(s, Wps, p->pscale.in_drop, K, L, p->pscale.keep, p->drop_seed,
                           p->pscale.site));
    Wpm = p->pmean.in_drop;
    Wps = p->pscale.in_drop;
  }
  float* qvar = a->cluster_stats ? p->dqs : nullptr;  // scratch, free in the forward pass
  TRY(softplus_gaussian_fwd(s, p->qm, p->qs, Wpm, bpm, Wps, bps, a->eps, p->z, p->klz, qvar, K, S,
                            B, L));
  if (a->q_z_mean)  // z_mean = sum_k y_k mean_k (gm:2895-2899)
    TRY(sum_groups(s, p->qm, p->yprob, K, K, B, L, a->q_z_mean));
  if (a->cluster_stats) {
    float* cs = a->cluster_stats;
    TRY(prior_stats(s, Wpm, bpm, Wps, bps, K, L, cs, cs + (size_t)K * L));
    // q_z_means / q_z_variances: this rank's share of the batch means (gm:2884-2887)
    TRY(group_col_sum(s, p->qm, L, B, K, L, inv_gb, cs + 2 * (size_t)K * L, p->partial));
    TRY(group_col_sum(s, qvar, L, B, K, L, inv_gb, cs + 3 * (size_t)K * L, p->partial));
  }

  // ---------------- decoder p(x|z_k), all k (gm:3094-3221) ----------------
  const int E = c.decoder_extra;
  const float* dec_in = p->z;   // decoder input: z_k, or [z_k | extra] (gm:3094-3130)
  if (E > 0) {
    TRY(concat_extra(s, p->z, L, a->decoder_extra, E, (size_t)R, (size_t)B, p->zcat));
    dec_in = p->zcat;
  }
  const float* dch = dec_in;
  ld = L + E;
  if (tile) {
    int cur = 0;
    for (size_t i = 0; i <= p->xdec.size(); ++i) {
      TileFwdArgs q;
      q.rows = R;
      if (i == 0) { q.x = dec_in; q.ldx = L; q.K = L; }
      else {
        q.K = p->xdec[i - 1].n_out;
        TRY(gm_tile_bn_forward(p, s, p->xdec[i - 1], K, SB, p->tc_part[cur], &q.bn));
      }
      if (i < p->xdec.size()) {
        Dense& d = p->xdec[i];
        q.n_out = 1;
        q.o[0].W = p->params + d.w; q.o[0].b = p->params + d.b; q.o[0].out = d.a;
        q.o[0].part = p->tc_part[i == 0 ? cur : cur ^ 1]; q.o[0].N = d.n_out;
      }   // (i == size: the last layer's normalisation alone -> its h feeds the likelihood heads)
      TRY(tile_forward(s, q));
      if (i > 0) cur ^= 1;
    }
    dch = p->xdec.back().h; ld = p->xdec.back().n_out;
  } else {
  for (auto& d : p->xdec) {
    TRY(dense_forward(p, s, d, dch, ld, R, K, true, training));
    dch = d.h; ld = d.n_out;
  }
  }
  HeadPtrs pre;
  for (int j = 0; j < 3; ++j) pre.p[j] = p->pre[j];
  const int h1 = p->heads[0].n_in;
  // fused heads + likelihood (+ backward) unless the evaluate-time statistics are requested
  const int KM = c.k_max, FC = F * (KM + 1);   // piecewise categorical likelihood: unfused path
  // (dropout: every head draws its own mask of the decoder output: the bf16x9 kernel's DROP
  //  instantiation, or the unfused path)
  const bool head_drop = training && p->heads[0].keep > 0.f;
  // row softmax: three passes of the bf16x9 head kernel (decoder_fused_cpoisson), or unfused
  const bool cpoisson = c.likelihood == LK_CPOISSON;
  if (cpoisson && !a->count_sum) {
    set_error("the constrained Poisson likelihood needs scvae_step_args.count_sum");
    return -1;
  }
  const bool fused_width =
      decoder_fused_supported(h1) ||
      (!head_drop && !cpoisson && decoder_fused_train_supported(p->P, h1, p->head_arith));
  const bool fused = p->use_fused && p->fused_ws && fused_width && ld == h1 &&
                     !a->p_x_mean && KM == 0 &&
                     (!head_drop || (heads_fused_dropout_ok(p, 1) && !cpoisson)) &&
                     (c.likelihood <= LK_ZINB || c.likelihood == LK_BERNOULLI ||
                      (cpoisson && decoder_fused_cpoisson_supported(h1, p->head_arith)));
  if (p->x_u16 && !fused) {
    set_error("the uint16 minibatch needs the fused likelihood kernels (no -k / constrained "
              "Poisson, evaluation statistics, or head dropout outside the bf16x9 kernel)");
    return -1;
  }
  // the K stacked passes read the same targets (row r uses t[r % B]): as uint16 they are half
  // the bytes of every pass
  const Targets tg = p->x_u16 ? targets_u16(p->step_u16, p->step_u16_ld) : targets_f32(a->t, F);
  const HeadParams hp = head_params(p);
  // -k (k = 1, 2) in a training step: two launches of the bf16x9 head kernel over the K stacked
  // passes (decoder_fused_train_cat, see plan.hip)
  const bool fused_cat = training && KM > 0 && p->use_fused && p->fused_ws && p->pre_k &&
                         ld == h1 && !head_drop && !p->x_u16 && !a->p_x_mean &&
                         decoder_fused_cat_supported(c.likelihood, KM, h1, p->head_arith);
  const bool cat_forward = !training && KM > 0 && p->use_fused && p->fused_ws && p->pre_k &&
                           ld == h1 && !p->x_u16 && !a->p_x_mean &&
                           decoder_fused_forward_cat_supported(c.likelihood, KM, h1);
  const float* head_in[4] = {dch, dch, dch, dch};   // [3]: the P_K head
  if (!fused && !fused_cat && !cat_forward)
    TRY(heads_forward(p, s, dch, ld, R, training, head_in));
  bool ll_done = false;
  if (a->p_x_mean) {
    if (!(a->p_x_stddev && a->stddev_of_p_x_given_z_mean)) {
      set_error("p_x_mean requires p_x_stddev and stddev_of_p_x_given_z_mean");
      return -1;
    }
    if (cpoisson) {   // likelihood of the logits first, then rates in place for the statistics
      if (training) {
        set_error("p_x_mean in a training step of the constrained Poisson likelihood");
        return -1;
      }
      TRY(cpoisson_fwd(s, a->t, F, p->pre[0], F, a->count_sum, a->row_const, p->ll, R, B, F));
      ll_done = true;
      TRY(cpoisson_rate(s, p->pre[0], F, a->count_sum, R, B, F));
    }
    for (int k = 0; k < K; ++k) {
      HeadPtrs pk;
      for (int j = 0; j < 3; ++j)
        pk.p[j] = p->pre[j] ? p->pre[j] + (size_t)k * SB * F : nullptr;
      if (KM > 0)
        TRY(px_statistics_cat(s, c.likelihood, pk, F, p->pre_k + (size_t)k * SB * FC, KM, S, B, F,
                              p->yprob + k, K, k > 0 ? 1 : 0, a->p_x_mean, p->mov, p->vom));
      else
        TRY(px_statistics(s, c.likelihood, pk, F, S, B, F, p->yprob + k, K, k > 0 ? 1 : 0,
                          a->p_x_mean, p->mov, p->vom));
    }
    TRY(sqrt_sum(s, p->vom, p->mov, a->p_x_stddev, (size_t)B * F));
    TRY(sqrt_sum(s, p->vom, nullptr, a->stddev_of_p_x_given_z_mean, (size_t)B * F));
  }

  // ---------------- loss (gm:3223-3410) ----------------
  float* sums = p->kl_cell + B;     // 3 floats (+ gate at [8])
  float* gate = sums + 8;
  const float p_y_entropy = logf((float)K);
  const float thr = c.free_nats_proportion * p_y_entropy;
  const int use_free_nats = c.free_nats_proportion != 0.f;
  if (!training) {
    if (fused && cpoisson)
      TRY(decoder_fused_cpoisson(s, false, dch, R, h1, hp, F, tg, B, nullptr, a->count_sum,
                                 a->row_const, p->ll, nullptr, p->fused_ws));
    else if (fused)
      TRY(decoder_fused_forward(s, c.likelihood, dch, R, h1, hp, F, tg, B, a->row_const, p->ll,
                                p->fused_ws, p->head_arith));
    else if (cat_forward)
      TRY(decoder_fused_forward_cat(s, c.likelihood, KM, dch, R, h1, hp, p->params + p->head_k.w,
                                    p->params + p->head_k.b, F, a->t, B, p->ll, p->fused_ws,
                                    p->pre_k));
    else if (KM > 0)
      TRY(loglik_cat_fwd(s, c.likelihood, a->t, F, pre, F, p->pre_k, KM, p->ll, R, B, F));
    else if (cpoisson) {
      if (!ll_done)
        TRY(cpoisson_fwd(s, a->t, F, p->pre[0], F, a->count_sum, a->row_const, p->ll, R, B, F));
    } else
      TRY(loglik_fwd(s, c.likelihood, a->t, F, pre, F, a->row_const, p->ll, R, B, F));
    TRY(gmvae_elbo(s, p->ll, p->klz, p->yprob, p->kl_y_cell, K, S, B, inv_gb, sums, nullptr));
    TRY(gmvae_elbo_finish(s, sums, w, thr, use_free_nats, 1.f, a->scalars, gate, prior, K,
                          c.free_nats_proportion));
    if (a->log_p_x_given_z) TRY(copy(s, p->ll, a->log_p_x_given_z, (size_t)R));
    return 0;
  }
  // d(-ELBO_w)/d log p(t|z_k)[k,s,b] = -y[b,k]/(S*GB): known before the likelihood pass
  TRY(gmvae_elbo_bwd(s, p->klz, p->klz, p->yprob, gate, K, S, B, w, inv_gb, p->gw, p->gklz,
                     p->dy));  // (fills gw, gklz; dy is recomputed below with ll)
  float* dcur = p->dbuf[0];
  float* dalt = p->dbuf[1];
  float* scratch = p->dbuf[2];
  if (fused) {
    HeadDropout hdrop;
    if (head_drop) TRY(heads_dropout_inputs(p, s, dch, ld, R, &hdrop));
    if (cpoisson)
      TRY(decoder_fused_cpoisson(s, true, dch, R, h1, hp, F, tg, B, p->gw, a->count_sum,
                                 a->row_const, p->ll, dcur, p->fused_ws));
    else
      TRY(decoder_fused_train(s, c.likelihood, dch, R, h1, hp, F, tg, B, p->gw, a->row_const,
                              p->ll, dcur, p->fused_ws, p->head_arith, false,
                              head_drop ? &hdrop : nullptr, p->dd_atomics));
  } else if (fused_cat) {
    Dense& hk = p->head_k;
    TRY(decoder_fused_train_cat(s, c.likelihood, KM, dch, R, h1, hp, p->params + hk.w,
                                p->params + hk.b, p->grads + hk.w, p->grads + hk.b, F, a->t, B,
                                p->gw, p->ll, dcur, p->fused_ws, p->head_arith, p->pre_k));
  } else if (KM > 0) {
    TRY(loglik_cat_bwd(s, c.likelihood, a->t, F, pre, F, p->pre_k, KM, p->gw, p->ll, R, B, F));
  } else if (cpoisson) {
    TRY(cpoisson_bwd(s, a->t, F, p->pre[0], F, p->gw, a->count_sum, a->row_const, p->ll, R, B, F));
  } else {
    TRY(loglik_bwd(s, c.likelihood, a->t, F, pre, F, p->gw, a->row_const, p->ll, R, B, F));
  }
  TRY(gmvae_elbo(s, p->ll, p->klz, p->yprob, p->kl_y_cell, K, S, B, inv_gb, sums, nullptr));
  float share = 1.f;
  if (p->sync) {
    // the free-nats gate depends on the global kl_divergence_y
    if (p->sync(p->sync_user, sums, 3, 0, B)) {
      set_error("ELBO sync hook failed");
      return -2;
    }
    share = (float)a->cells / (float)GB;
  }
  TRY(gmvae_elbo_finish(s, sums, w, thr, use_free_nats, share, a->scalars, gate, prior, K,
                        c.free_nats_proportion));
  TRY(gmvae_elbo_bwd(s, p->ll, p->klz, p->yprob, gate, K, S, B, w, inv_gb, p->gw, p->gklz, p->dy));
  if (a->log_p_x_given_z) TRY(copy(s, p->ll, a->log_p_x_given_z, (size_t)R));

  // ---------------- backward: heads + decoder ----------------
  if (!fused && !fused_cat) TRY(heads_backward(p, s, head_in, R, head_drop, dcur, dalt));
  // the next minibatch and its noise (scvae_step_args.side) under the rest of the backward pass
  TRY(plan_side_fork(p, s, 1));
  const int64_t GSB = GB * S;  // global rows per group (pass) in the decoder
  // (tile chain) the dW / db slabs of the layers wait for one fixed-order reduce at the end
  SlabJobs pending;
  int sp = 0;
  auto bessel = [](int64_t n) { return (float)n / (float)(n > 1 ? n - 1 : 1); };
  auto flush = [&]() -> int {
    if (pending.n_jobs == 0) return 0;
    const int r = tile_slab_reduce(s, pending);
    pending.n_jobs = 0;
    return r;
  };
  // one batch-normalised layer backwards: its own sums merged per group, dA, d_in, its dW slab and
  // the chunk sums of the layer below
  auto tile_layer_backward = [&](Dense& d, Dense* below, const float* in, int rows, int group_rows,
                                 int64_t grows, const float* dh_in, float* d_in) -> int {
    TileBwdArgs q;
    const int G = rows / 64;
    q.rows = rows; q.inv_count = 1.f / (float)grows; q.bessel = bessel(grows);
    q.n_up = 1;
    if (pending.n_jobs == TC_MAX_JOBS) { const int r = flush(); if (r) return r; }
    float* slab = p->tc_slab[pending.n_jobs % TC_MAX_JOBS];
    q.up[0].g = dh_in; q.up[0].W = p->params + d.w; q.up[0].N = d.n_out;
    q.up[0].dW_slab = slab;
    {
      const int r = gm_tile_bn_backward(p, s, d, K, group_rows, p->tc_spart[sp], q.bessel, &q.bn);
      if (r) return r;
    }
    q.in = in; q.K = d.n_in; q.d_in = d_in;
    if (below) q.below = gm_tile_bn(p, *below, K, group_rows, nullptr, p->tc_spart[sp ^ 1]);
    const int r = tile_backward(s, q);
    if (r) return r;
    pending.job[pending.n_jobs++] = {slab, p->grads + d.w, d.n_in * d.n_out, G};
    sp ^= 1;
    return 0;
  };
  if (tile) {
    Dense& top = p->xdec.back();
    TRY(tile_backward_stats(s, dcur, gm_tile_bn(p, top, K, SB, nullptr, p->tc_spart[sp]), R,
                            top.n_out));
    for (int i = (int)p->xdec.size() - 1; i >= 0; --i) {
      const float* in = i > 0 ? p->xdec[i - 1].h : dec_in;
      float* d_in = i > 0 ? dalt : p->dz;
      TRY(tile_layer_backward(p->xdec[i], i > 0 ? &p->xdec[i - 1] : nullptr, in, R, SB, GSB, dcur,
                              d_in));
      if (i > 0) { float* t = dcur; dcur = dalt; dalt = t; }
    }
  } else {
  for (int i = (int)p->xdec.size() - 1; i >= 0; --i) {
    Dense& d = p->xdec[i];
    const float* in = i > 0 ? p->xdec[i - 1].h : dec_in;
    float* d_in = i > 0 ? dalt : (E > 0 ? p->dzcat : p->dz);
    TRY(dense_backward(p, s, d, in, d.n_in, R, K, true, dcur, scratch, d_in, false, GSB));
    if (i > 0) { float* t = dcur; dcur = dalt; dalt = t; }
  }
  }
  if (E > 0 && !p->xdec.empty()) TRY(slice_cols(s, p->dzcat, L + E, L, (size_t)R, p->dz));

  // ---------------- backward: latent, prior, q(z|x,y) ----------------
  TRY(softplus_gaussian_bwd(s, p->qm, p->qs, Wpm, bpm, Wps, bps, a->eps, p->dz, p->gklz, p->dqm,
                            p->dqs, p->dprior, K, S, B, L));
  // prior dense layers on the one-hot: dW[k,:] = sum_b, db = sum_k dW[k,:]
  TRY(group_col_sum(s, p->dprior, 2 * L, B, K, 2 * L, 1.f, p->sum_scratch, p->partial));
  {
    // sum_scratch: [K, 2L] = (d pm | d ps) rows; scatter into the two weight matrices
    float* dWpm = p->grads + p->pmean.w;
    float* dWps = p->grads + p->pscale.w;
    TRY(hipMemcpy2DAsync(dWpm, L * sizeof(float), p->sum_scratch, 2 * L * sizeof(float),
                         L * sizeof(float), K, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -2);
    TRY(hipMemcpy2DAsync(dWps, L * sizeof(float), p->sum_scratch + L, 2 * L * sizeof(float),
                         L * sizeof(float), K, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -2);
    TRY(col_sum(s, dWpm, L, K, L, p->grads + p->pmean.b, 1.f, 0, nullptr));
    TRY(col_sum(s, dWps, L, K, L, p->grads + p->pscale.b, 1.f, 0, nullptr));
    if (prior_drop) {   // the weights saw the masked one-hot; the biases did not
      TRY(dropout_scale_rows(s, dWpm, dWpm, K, L, p->pmean.keep, p->drop_seed, p->pmean.site));
      TRY(dropout_scale_rows(s, dWps, dWps, K, L, p->pscale.keep, p->drop_seed, p->pscale.site));
    }
  }
  float* dh = p->dbuf[0];
  float* dh_alt = p->dbuf[1];
  if (tile) {
    // the two posterior heads: dW, db of both, dh of the last q(z|x,y) layer and (where that
    // layer belongs to the chain) its chunk sums
    Dense& last = p->zenc.back();
    const int G = KB / 64, Kl = last.n_out;
    if (pending.n_jobs + 4 > TC_MAX_JOBS) TRY(flush());
    float* slab2[2] = {p->tc_slab[pending.n_jobs], p->tc_slab[pending.n_jobs + 1]};
    TileBwdArgs q;
    q.rows = KB; q.n_up = 2;
    for (int u = 0; u < 2; ++u) {
      Dense& hd = u == 0 ? p->qmean : p->qscale;
      q.up[u].g = u == 0 ? p->dqm : p->dqs;
      q.up[u].W = p->params + hd.w; q.up[u].N = L;
      q.up[u].dW_slab = slab2[u];
      q.up[u].db_slab = slab2[u] + (size_t)G * 128 * 128;
    }
    q.in = last.h; q.K = Kl; q.d_in = dh;
    if (p->zenc.size() > 1) q.below = gm_tile_bn(p, last, K, B, nullptr, p->tc_spart[sp]);
    TRY(tile_backward(s, q));
    const int j0 = pending.n_jobs;
    pending.job[j0] = {slab2[0], p->grads + p->qmean.w, Kl * L, G};
    pending.job[j0 + 1] = {slab2[1], p->grads + p->qscale.w, Kl * L, G};
    pending.job[j0 + 2] = {q.up[0].db_slab, p->grads + p->qmean.b, L, G};
    pending.job[j0 + 3] = {q.up[1].db_slab, p->grads + p->qscale.b, L, G};
    pending.n_jobs = j0 + 4;
  } else
  for (int q = 0; q < 2; ++q) {
    Dense& hd = q == 0 ? p->qmean : p->qscale;
    const float* dpre = q == 0 ? p->dqm : p->dqs;
    const float* hq = q == 0 ? hz_m : hz_s;   // the (dropped-out) input of that layer
    const bool drop = hd.keep > 0.f;
    GEMM(true, false, hq, dpre, nullptr, p->grads + hd.w, hd.n_in, L, KB, hd.n_in, L, L, ACT_NONE,
         false);
    TRY(col_sum(s, dpre, L, KB, L, p->grads + hd.b, 1.f, 0, p->partial));
    GEMM(false, true, dpre, p->params + hd.w, nullptr, drop ? dh_alt : dh, KB, hd.n_in, L, L, L,
         hd.n_in, ACT_NONE, !drop && q > 0);
    if (drop) TRY(dense_input_backward(p, s, hd, dh_alt, dh, KB, q > 0));
  }
  for (int i = (int)p->zenc.size() - 1; i >= 0; --i) {
    Dense& d = p->zenc[i];
    if (i > 0 && tile) {
      TRY(tile_layer_backward(d, i > 1 ? &p->zenc[i - 1] : nullptr, p->zenc[i - 1].h, KB, B, GB,
                              dh, dh_alt));
      float* t = dh; dh = dh_alt; dh_alt = t;
    } else if (i > 0) {
      TRY(dense_backward(p, s, d, p->zenc[i - 1].h, d.n_in, KB, K, true, dh, scratch, dh_alt,
                         false, GB));
      float* t = dh; dh = dh_alt; dh_alt = t;
    } else {
      const float* da = nullptr;
      TRY(dense_backward_activation(p, s, d, KB, K, true, dh, scratch, GB, &da));
      const int N = d.n_out;
      float* dW = p->grads + d.w;
      if (!d.bn) TRY(col_sum(s, da, N, KB, N, p->grads + d.b, 1.f, 0, p->partial));
      if (d.keep > 0.f) {
        // the K passes read K different dropped-out inputs: dW = [x | one-hot]_dropped^T dA
        GEMM(true, false, d.in_drop, da, nullptr, dW, F + K, N, KB, F + K, N, N, ACT_NONE, false);
      } else {
        // one-hot rows: dW[F+k,:] = sum_b dA[k,b,:]
        TRY(group_col_sum(s, da, N, B, K, N, 1.f, dW + (size_t)F * N, p->partial));
        // data rows: dW[:F] = x^T (sum_k dA[k])
        TRY(sum_groups(s, da, nullptr, 0, K, B, N, p->sum_scratch));
        GEMM(true, false, p->step_x, p->sum_scratch, nullptr, dW, F, N, B, F, N, N, ACT_NONE, false);
      }
    }
  }

  if (tile) TRY(flush());     // (the weight-gradient slabs of the chain: one fixed-order reduce)

  // ---------------- backward: q(y|x) ----------------
  TRY(categorical_bwd_gated(s, p->yprob, p->dy, gate, w * inv_gb, p->dlogits, B, K, prior));
  if (c.prior_mode == 2)   // learned p(y): this rank's share of the gradient of w * KL_y_modified
    TRY(prior_logits_bwd(s, p->yprob, prior, gate, w * inv_gb, w * share, c.free_nats_proportion,
                         B, K, p->grads + p->prior_off));
  {
    Dense& hd = p->ylogits;
    GEMM(true, false, hy, p->dlogits, nullptr, p->grads + hd.w, hd.n_in, K, B, ldy, K, K, ACT_NONE,
         false);
    TRY(col_sum(s, p->dlogits, K, B, K, p->grads + hd.b, 1.f, 0, p->partial));
    dh = p->dbuf[0];
    dh_alt = p->dbuf[1];
    if (!p->yenc.empty()) {
      GEMM(false, true, p->dlogits, p->params + hd.w, nullptr, dh, B, hd.n_in, K, K, K, hd.n_in,
           ACT_NONE, false);
      if (hd.keep > 0.f) TRY(dense_input_backward(p, s, hd, dh, dh, B, false));
    }
  }
  for (int i = (int)p->yenc.size() - 1; i >= 0; --i) {
    Dense& d = p->yenc[i];
    const float* in = i > 0 ? p->yenc[i - 1].h : p->step_x;
    float* d_in = i > 0 ? dh_alt : nullptr;
    TRY(dense_backward(p, s, d, in, d.n_in, B, 1, true, dh, scratch, d_in, false, GB));
    if (i > 0) { float* t = dh; dh = dh_alt; dh_alt = t; }
  }

  // ---------------- batch-norm moving averages (K passes update in pass order) --------
  // (the batch-norm moving averages were updated by the layers' backward statistics launches)
#undef GEMM
#undef TRY
  return 0;
}

}  // namespace scvae
