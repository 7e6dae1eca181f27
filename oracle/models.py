"""Oracle (TEST INFRASTRUCTURE): VAE / GMVAE forward, loss, gradients, Adam.

PARITY UNPINNED (see ``oracle/__init__.py``): the reference graph cannot be
executed in the build container, so this restatement follows the reference
source line by line but is not checked against reference outputs.

Restates, dtype-generically in torch on the CPU (float64 for parity/fixtures,
float32 for ``bench.py``'s ``cpu_baseline``):

* ``dense_layer`` / ``dense_layers``      scvae/models/utilities.py:38-126
  (``tf.contrib.layers.fully_connected`` + ``batch_norm(center=True,
  scale=False)``: epsilon 1e-3, decay 0.999, fused kernel => normalise with
  the biased batch variance, update moving_variance with the unbiased one)
* VAE graph                               scvae/models/variational_autoencoder.py:2219-2558
* VAE loss                                scvae/models/variational_autoencoder.py:2560-2734
* GMVAE graph                             scvae/models/gaussian_mixture_variational_autoencoder.py:2788-3221
* GMVAE loss                              scvae/models/gaussian_mixture_variational_autoencoder.py:3223-3434
* optimiser (clip-by-value +-1, TF Adam)  scvae/models/variational_autoencoder.py:2736-2770
* ``log_reduce_exp``                      scvae/models/utilities.py:129-137

Gradients come from torch autograd over these same formulas.
"""

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Tuple

import torch
import torch.nn.functional as F

from oracle import likelihoods as lk

BN_EPSILON = 1e-3
BN_DECAY = 0.999
ADAM_BETA1 = 0.9
ADAM_BETA2 = 0.999
ADAM_EPSILON = 1e-8
HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)
FLOAT32_MAX_HALF = 3.4028234663852886e38 / 2


@dataclass
class ModelConfig:
    feature_size: int
    latent_size: int = 2
    hidden_sizes: Tuple[int, ...] = (100,)
    likelihood: str = "poisson"
    minibatch_normalisation: bool = True
    n_iw: int = 1
    n_mc: int = 1
    kl_weight: float = 1.0
    # GMVAE only
    n_clusters: int = 1
    free_nats_proportion: float = 0.0
    # extra decoder input columns appended to z: one-hot batch indices
    # (batch_correction) and/or the normalised count sum
    # (use_count_sum_as_feature), va:2407-2441, gm:3094-3130
    decoder_extra_size: int = 0
    # linear factor model on either side (va:2233-2234, 2456-2457): that side's
    # hidden layers are not built
    inference_architecture: str = "MLP"
    generative_architecture: str = "MLP"
    # GMVAE p(y), gm:2794-2808: "uniform", "custom" (``prior_probabilities``, a
    # constant) or "learn" (trainable variable Y/P/LOGITS, initialised to zeros)
    prior_probabilities_method: str = "uniform"
    prior_probabilities: Tuple[float, ...] = ()
    # piecewise categorical likelihood (-k): counts below k_max are classes of
    # a categorical head P_K, va:2507-2532 (0 = off)
    k_max: int = 0
    # VAE latent part, du:309-338: "gaussian" or "unit-variance gaussian" (the
    # posterior's log_sigma is the constant 0.0, no POSTERIOR/LOG_SIGMA
    # layer); analytical_kl_term (va:186-192, 2624-2640): closed-form
    # KL(q||p), else log q(z|x) - log p(z) evaluated at the samples
    latent_distribution: str = "gaussian"
    analytical_kl_term: bool = True

    @property
    def heads(self):
        return lk.LIKELIHOOD_PARAMETERS[self.likelihood]


# --------------------------------------------------------------------------
# parameter tables (names follow the reference's variable scopes,
# SURVEY.md appendix B)
# --------------------------------------------------------------------------

def _dense_entries(scope, n_in, n_out, bn):
    e = [(scope + "/DENSE/weights", (n_in, n_out)),
         (scope + "/DENSE/biases", (n_out,))]
    if bn:
        e.append((scope + "/BATCH_NORM/beta", (n_out,)))
    return e


def vae_parameter_shapes(cfg):
    bn = cfg.minibatch_normalisation
    H = list(cfg.hidden_sizes)
    n = len(H)
    shapes = []
    n_in = cfg.feature_size
    for i, h in enumerate(H if cfg.inference_architecture == "MLP" else []):
        shapes += _dense_entries("ENCODER/{}".format(i + 1), n_in, h, bn)
        n_in = h
    shapes += _dense_entries("POSTERIOR/MU", n_in, cfg.latent_size, False)
    if cfg.latent_distribution != "unit-variance gaussian":
        shapes += _dense_entries("POSTERIOR/LOG_SIGMA", n_in, cfg.latent_size,
                                 False)
    n_in = cfg.latent_size + cfg.decoder_extra_size
    # reverse_order=True: sizes reversed, scopes numbered n..1
    for i, h in enumerate(H[::-1] if cfg.generative_architecture == "MLP"
                          else []):
        shapes += _dense_entries("DECODER/{}".format(n - i), n_in, h, bn)
        n_in = h
    for p in cfg.heads:
        shapes += _dense_entries("X_TILDE/" + p.upper(), n_in,
                                 cfg.feature_size, False)
    if cfg.k_max:
        shapes += _dense_entries(
            "X_TILDE/P_K", n_in, cfg.feature_size * (cfg.k_max + 1), False)
    return OrderedDict(shapes)


def gmvae_parameter_shapes(cfg):
    bn = cfg.minibatch_normalisation
    H = list(cfg.hidden_sizes)
    K, L, Fs = cfg.n_clusters, cfg.latent_size, cfg.feature_size
    shapes = []
    if cfg.prior_probabilities_method == "learn":
        shapes.append(("Y/P/LOGITS", (K,)))
    n_in = Fs
    for i, h in enumerate(H):
        shapes += _dense_entries(
            "Y/CATEGORICAL/ENCODER/LAYER_{}".format(i + 1), n_in, h, bn)
        n_in = h
    shapes += _dense_entries("Y/CATEGORICAL/LOGITS", n_in, K, False)
    n_in = Fs + K
    for i, h in enumerate(H):
        shapes += _dense_entries(
            "Z/Q/ENCODER/LAYER_{}".format(i + 1), n_in, h, bn)
        n_in = h
    shapes += _dense_entries("Z/Q/SOFTPLUS_GAUSSIAN/MEAN", n_in, L, False)
    shapes += _dense_entries("Z/Q/SOFTPLUS_GAUSSIAN/SOFTPLUS_SCALE", n_in, L,
                             False)
    shapes += _dense_entries("Z/P/SOFTPLUS_GAUSSIAN/MEAN", K, L, False)
    shapes += _dense_entries("Z/P/SOFTPLUS_GAUSSIAN/SOFTPLUS_SCALE", K, L,
                             False)
    n_in = L + cfg.decoder_extra_size
    for i, h in enumerate(H[::-1]):
        shapes += _dense_entries(
            "X/DECODER/LAYER_{}".format(i + 1), n_in, h, bn)
        n_in = h
    for p in cfg.heads:
        shapes += _dense_entries("X/DISTRIBUTION/" + p.upper(), n_in, Fs,
                                 False)
    if cfg.k_max:
        shapes += _dense_entries("X/DISTRIBUTION/P_K", n_in,
                                 Fs * (cfg.k_max + 1), False)
    return OrderedDict(shapes)


def init_parameters(shapes, seed=0, dtype=torch.float64):
    """Glorot-uniform weights, zero biases/beta (tf.contrib defaults)."""
    g = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape in shapes.items():
        if name.endswith("weights"):
            limit = math.sqrt(6.0 / (shape[0] + shape[1]))
            w = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1)
            params[name] = (w * limit).to(dtype)
        else:
            params[name] = torch.zeros(shape, dtype=dtype)
    return params


def init_moving_statistics(shapes, dtype=torch.float64):
    stats = OrderedDict()
    for name, shape in shapes.items():
        if name.endswith("BATCH_NORM/beta"):
            base = name[:-len("beta")]
            stats[base + "moving_mean"] = torch.zeros(shape, dtype=dtype)
            stats[base + "moving_variance"] = torch.ones(shape, dtype=dtype)
    return stats


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------

def dense_layer(x, params, scope, bn, training, moving, new_moving,
                activation=True, extra_row=None, dropout=None):
    """utilities.py:38-76.  ``extra_row`` adds one weight row (GMVAE one-hot
    input column) to the affine map.  ``dropout``: dict scope -> the tensor
    ``mask / keep_prob`` of ``tf.nn.dropout`` on this layer's inputs
    (utilities.py:45-50, training only; the mask is an explicit input like
    eps so that a run can be reproduced)."""
    W = params[scope + "/DENSE/weights"]
    b = params[scope + "/DENSE/biases"]
    if training and dropout is not None and scope in dropout:
        if extra_row is not None:   # the one-hot column is part of the input
            n_x = x.shape[1]
            xy = torch.zeros(x.shape[0], W.shape[0], dtype=x.dtype)
            xy[:, :n_x] = x
            xy[:, n_x + extra_row] = 1.0
            x, extra_row = xy, None
        x = x * dropout[scope]
    if extra_row is None:
        a = x @ W + b
    else:
        n_x = x.shape[1]
        a = x @ W[:n_x] + W[n_x + extra_row] + b
    if bn:
        beta = params[scope + "/BATCH_NORM/beta"]
        mkey = scope + "/BATCH_NORM/moving_mean"
        vkey = scope + "/BATCH_NORM/moving_variance"
        if training:
            n = a.shape[0]
            mean = a.mean(dim=0)
            centred = a - mean
            var = (centred * centred).mean(dim=0)
            a = centred * torch.rsqrt(var + BN_EPSILON) + beta
            if new_moving is not None:
                mm = new_moving.get(mkey, moving[mkey])
                mv = new_moving.get(vkey, moving[vkey])
                unbiased = var.detach() * (n / max(n - 1, 1))
                new_moving[mkey] = mm - (mm - mean.detach()) * (1 - BN_DECAY)
                new_moving[vkey] = mv - (mv - unbiased) * (1 - BN_DECAY)
        else:
            a = ((a - moving[mkey])
                 * torch.rsqrt(moving[vkey] + BN_EPSILON) + beta)
    if activation:
        a = torch.relu(a)
    return a


def log_reduce_exp_mean(x, dim):
    """utilities.py:129-137 with reduction_function = mean."""
    m = x.max(dim=dim, keepdim=True).values
    return (torch.log(torch.exp(x - m).mean(dim=dim, keepdim=True))
            + m).squeeze(dim)


def _normal_log_prob(z, mean, sigma):
    return (-0.5 * ((z - mean) / sigma) ** 2 - torch.log(sigma)
            - HALF_LOG_2PI)


def _decoder_distribution(cfg, d, params, scope, training, moving,
                          dropout=None, count_sum=None):
    """Head pre-activations (and the P_K logits) of p(x|z) from the last
    decoder layer ``d``; returns (log_prob(t), mean_variance()) closures.
    ``count_sum``: [rows, 1], the N of the constrained Poisson (already
    tiled over the samples, va:2400-2405)."""
    pre = tuple(
        dense_layer(d, params, scope + p.upper(), False, training, moving,
                    None, activation=False, dropout=dropout)
        for p in cfg.heads)
    if not cfg.k_max:
        return (lambda t: lk.log_prob(cfg.likelihood, t, pre, count_sum),
                lambda: lk.mean_variance(cfg.likelihood, pre, count_sum))
    logits = dense_layer(d, params, scope + "P_K", False, training, moving,
                         None, activation=False, dropout=dropout)
    logits = logits.reshape(d.shape[0], cfg.feature_size, cfg.k_max + 1)
    return (lambda t: lk.categorised_log_prob(cfg.likelihood, t, pre, logits,
                                              cfg.k_max),
            lambda: lk.categorised_mean_variance(cfg.likelihood, pre, logits,
                                                 cfg.k_max))


# --------------------------------------------------------------------------
# decoder only (model.sample(): va:1601-1779, gm:1949-2160)
# --------------------------------------------------------------------------

def decode_mean(cfg, params, moving, z, model_type="VAE"):
    """``session.run(self.p_x_mean, {self.z: z, self.is_training: False})``:
    mean of p(x|z) for given latent values z [rows, L] (moving statistics)."""
    bn = cfg.minibatch_normalisation
    H = list(cfg.hidden_sizes)
    n = len(H)
    if model_type == "VAE":
        d = z
        for i in range(n if cfg.generative_architecture == "MLP" else 0):
            d = dense_layer(d, params, "DECODER/{}".format(n - i), bn, False,
                            moving, None)
        scope = "X_TILDE/"
    else:
        d = _layers(z, params, "X/DECODER", H[::-1], bn, False, moving, None)
        scope = "X/DISTRIBUTION/"
    _, mean_variance = _decoder_distribution(cfg, d, params, scope, False,
                                             moving)
    return mean_variance()[0]


# --------------------------------------------------------------------------
# VAE
# --------------------------------------------------------------------------

def vae_forward(cfg, params, moving, x, t, eps, training, warm_up_weight=1.0,
                new_moving=None, deterministic_z=False, analytical_kl=None,
                evaluation_statistics=False, decoder_extra=None,
                dropout=None, count_sum=None):
    """One graph execution.  ``eps``: [S, B, L] standard-normal draws
    (S = n_iw * n_mc, IW-major) or None with ``deterministic_z``.
    ``dropout``: {layer scope: mask / keep_prob} (see dense_layer)."""
    bn = cfg.minibatch_normalisation
    H = list(cfg.hidden_sizes)
    n = len(H)
    B = x.shape[0]
    L = cfg.latent_size

    h = x
    for i in range(n if cfg.inference_architecture == "MLP" else 0):
        h = dense_layer(h, params, "ENCODER/{}".format(i + 1), bn, training,
                        moving, new_moving, dropout=dropout)
    mu = dense_layer(h, params, "POSTERIOR/MU", False, training, moving,
                     None, activation=False, dropout=dropout)
    mu = torch.clamp(mu, -FLOAT32_MAX_HALF, FLOAT32_MAX_HALF)
    if cfg.latent_distribution == "unit-variance gaussian":
        # a constant parameter skips the layer and its clip (va:2253-2265)
        log_sigma = torch.zeros_like(mu)
    else:
        log_sigma = dense_layer(h, params, "POSTERIOR/LOG_SIGMA", False,
                                training, moving, None, activation=False,
                                dropout=dropout)
        log_sigma = torch.clamp(log_sigma, -3.0, 3.0)
    sigma = torch.exp(log_sigma)
    if analytical_kl is None:
        analytical_kl = cfg.analytical_kl_term

    if deterministic_z:
        n_iw, n_mc = 1, 1
        z = mu.unsqueeze(0)
    else:
        n_iw, n_mc = cfg.n_iw, cfg.n_mc
        z = mu.unsqueeze(0) + sigma.unsqueeze(0) * eps  # [S, B, L]
    S = z.shape[0]

    d = z.reshape(S * B, L)
    if decoder_extra is not None:   # tf.tile(extra, [S, 1]); tf.concat
        d = torch.cat([d, decoder_extra.repeat(S, 1)], dim=1)
    for i in range(n if cfg.generative_architecture == "MLP" else 0):
        d = dense_layer(d, params, "DECODER/{}".format(n - i), bn, training,
                        moving, new_moving, dropout=dropout)
    log_prob, mean_variance = _decoder_distribution(
        cfg, d, params, "X_TILDE/", training, moving, dropout,
        None if count_sum is None else count_sum.reshape(-1, 1).repeat(S, 1))

    t_tiled = t.repeat(S, 1)
    log_p = log_prob(t_tiled).sum(dim=-1)
    log_p = log_p.reshape(n_iw, n_mc, B)

    if analytical_kl:
        kl = 0.5 * (mu * mu + sigma * sigma - 1.0) - log_sigma  # [B, L]
        kl_neurons = kl.mean(dim=0)
        kl_cell = kl.sum(dim=-1).reshape(1, 1, B)
    else:
        zr = z.reshape(n_iw, n_mc, B, L)
        log_q = _normal_log_prob(zr, mu, sigma)
        log_pz = _normal_log_prob(zr, torch.zeros_like(mu),
                                  torch.ones_like(sigma))
        kl = log_q - log_pz
        kl_neurons = kl.reshape(-1, L).mean(dim=0)
        kl_cell = kl.sum(dim=-1)
    out = {
        "reconstruction_error": log_p.mean(),
        "kl_divergence": kl_neurons.sum(),
        "kl_divergence_neurons": kl_neurons,
        "log_p_x_given_z": log_p,            # [IW, MC, B] per-cell log-lik
        "q_z_mean": mu,
        "q_z_log_sigma": log_sigma,
        "z": z,
    }
    out["lower_bound"] = log_reduce_exp_mean(log_p - kl_cell, 0).mean()
    w = warm_up_weight * cfg.kl_weight
    out["lower_bound_weighted"] = log_reduce_exp_mean(
        log_p - w * kl_cell, 0).mean()

    if evaluation_statistics:
        m, v = mean_variance()
        m = m.reshape(n_iw, n_mc, B, -1)
        v = v.reshape(n_iw, n_mc, B, -1)
        p_x_mean = m.mean(dim=1).mean(dim=0)
        var_of_mean = ((m - p_x_mean) ** 2).mean(dim=1).mean(dim=0)
        out["p_x_mean"] = p_x_mean
        out["stddev_of_p_x_given_z_mean"] = torch.sqrt(var_of_mean)
        out["p_x_stddev"] = torch.sqrt(
            var_of_mean + v.mean(dim=1).mean(dim=0))
    return out


# --------------------------------------------------------------------------
# GMVAE
# --------------------------------------------------------------------------

def _layers(x, params, prefix, sizes, bn, training, moving, new_moving,
            extra_row=None, dropout=None):
    h = x
    for i in range(len(sizes)):
        h = dense_layer(h, params, "{}/LAYER_{}".format(prefix, i + 1), bn,
                        training, moving, new_moving,
                        extra_row=extra_row if i == 0 else None,
                        dropout=dropout)
    return h


def _clip_big(a):
    return torch.clamp(a, -FLOAT32_MAX_HALF, FLOAT32_MAX_HALF)


def gmvae_forward(cfg, params, moving, x, t, eps, training,
                  warm_up_weight=1.0, new_moving=None,
                  evaluation_statistics=False, decoder_extra=None,
                  dropout=None, count_sum=None):
    """``eps``: [K, S, B, L].  ``dropout``: {layer scope: mask / keep_prob}
    (see dense_layer); the layers under Z/ and X/ are built once per cluster
    (gm:2859-2922), each copy with its own dropout op, so their entries carry
    a leading axis of K passes."""
    bn = cfg.minibatch_normalisation
    H = list(cfg.hidden_sizes)
    K, L = cfg.n_clusters, cfg.latent_size
    B = x.shape[0]
    S = cfg.n_iw * cfg.n_mc

    # q(y|x)
    if not training:
        dropout = None
    hy = _layers(x, params, "Y/CATEGORICAL/ENCODER", H, bn, training, moving,
                 new_moving, dropout=dropout)
    logits = dense_layer(hy, params, "Y/CATEGORICAL/LOGITS", False, training,
                         moving, None, activation=False, dropout=dropout)
    log_y = torch.log_softmax(logits, dim=-1)
    y = torch.exp(log_y)
    entropy = -(y * log_y).sum(dim=-1)
    if cfg.prior_probabilities_method == "uniform":   # gm:3242-3254
        p_y_entropy = math.log(K)
        kl_y_cell = p_y_entropy - entropy
    else:   # tfp kl_divergence(Categorical q, Categorical p), gm:3256-3258
        if cfg.prior_probabilities_method == "learn":
            log_p_y = torch.log_softmax(params["Y/P/LOGITS"], dim=-1)
        else:
            log_p_y = torch.log_softmax(torch.log(torch.as_tensor(
                cfg.prior_probabilities, dtype=logits.dtype)), dim=-1)
        kl_y_cell = (y * (log_y - log_p_y)).sum(dim=-1)
        p_y_entropy = -(torch.exp(log_p_y) * log_p_y).sum()

    Wpm = params["Z/P/SOFTPLUS_GAUSSIAN/MEAN/DENSE/weights"]
    bpm = params["Z/P/SOFTPLUS_GAUSSIAN/MEAN/DENSE/biases"]
    Wps = params["Z/P/SOFTPLUS_GAUSSIAN/SOFTPLUS_SCALE/DENSE/weights"]
    bps = params["Z/P/SOFTPLUS_GAUSSIAN/SOFTPLUS_SCALE/DENSE/biases"]

    t_tiled = t.repeat(S, 1)
    kl_z_cell = 0.0
    rec_cell = 0.0
    kl_z_neurons = 0.0
    z_mean = 0.0
    log_p_all = []
    p_z_means, p_z_variances, q_z_means, q_z_variances = [], [], [], []
    p_x_means, mean_of_var, var_of_mean = [], [], []
    for k in range(K):
        dk = None
        if dropout is not None:   # this pass's masks
            dk = {scope: m[k] for scope, m in dropout.items()
                  if scope.startswith(("Z/", "X/"))}
        h = _layers(x, params, "Z/Q/ENCODER", H, bn, training, moving,
                    new_moving, extra_row=k, dropout=dk)
        q_mean = _clip_big(dense_layer(
            h, params, "Z/Q/SOFTPLUS_GAUSSIAN/MEAN", False, training, moving,
            None, activation=False, dropout=dk))
        q_s = _clip_big(dense_layer(
            h, params, "Z/Q/SOFTPLUS_GAUSSIAN/SOFTPLUS_SCALE", False,
            training, moving, None, activation=False, dropout=dk))
        q_sigma = torch.sqrt(F.softplus(q_s))
        z = q_mean.unsqueeze(0) + q_sigma.unsqueeze(0) * eps[k]   # [S,B,L]
        # p(z|y=k): dense layers on the one-hot row (gm:3024-3040); their
        # dropout mask [K] keeps or drops the one non-zero input
        one_hot_m = torch.zeros(K, dtype=Wpm.dtype)
        one_hot_m[k] = 1.0
        one_hot_s = one_hot_m
        if dk and "Z/P/SOFTPLUS_GAUSSIAN/MEAN" in dk:
            one_hot_m = one_hot_m * dk["Z/P/SOFTPLUS_GAUSSIAN/MEAN"]
            one_hot_s = one_hot_s * dk["Z/P/SOFTPLUS_GAUSSIAN/SOFTPLUS_SCALE"]
        p_mean = _clip_big(one_hot_m @ Wpm + bpm)
        p_sigma = torch.sqrt(F.softplus(_clip_big(one_hot_s @ Wps + bps)))

        d = z.reshape(S * B, L)
        if decoder_extra is not None:
            d = torch.cat([d, decoder_extra.repeat(S, 1)], dim=1)
        d = _layers(d, params, "X/DECODER", H[::-1], bn, training, moving,
                    new_moving, dropout=dk)
        log_prob, mean_variance = _decoder_distribution(
            cfg, d, params, "X/DISTRIBUTION/", training, moving, dk,
            None if count_sum is None
            else count_sum.reshape(-1, 1).repeat(S, 1))
        log_p = log_prob(t_tiled).sum(dim=-1)
        log_p = log_p.reshape(S, B)
        log_p_all.append(log_p)

        kl_dims = (_normal_log_prob(z, q_mean, q_sigma)
                   - _normal_log_prob(z, p_mean, p_sigma))     # [S,B,L]
        yk = y[:, k]
        kl_z_cell = kl_z_cell + kl_dims.sum(dim=-1).mean(dim=0) * yk
        rec_cell = rec_cell + log_p.mean(dim=0) * yk
        kl_z_neurons = kl_z_neurons + kl_dims.mean(dim=0) * yk.unsqueeze(-1)
        z_mean = z_mean + q_mean * yk.unsqueeze(-1)

        p_z_means.append(p_mean)
        p_z_variances.append(p_sigma ** 2)
        q_z_means.append(q_mean.mean(dim=0))
        q_z_variances.append((q_sigma ** 2).mean(dim=0))

        if evaluation_statistics:
            m, v = mean_variance()
            m = m.reshape(S, B, -1)
            v = v.reshape(S, B, -1)
            pxm = m.mean(dim=0) * yk.unsqueeze(-1)
            p_x_means.append(pxm)
            mean_of_var.append(v.mean(dim=0) * yk.unsqueeze(-1))
            # gm:3338-3351 subtracts the already y-weighted mean (quirk)
            var_of_mean.append(((m - pxm) ** 2).mean(dim=0)
                               * yk.unsqueeze(-1))

    kl_z = kl_z_cell.mean()
    kl_y = kl_y_cell.mean()
    rec = rec_cell.mean()
    if cfg.free_nats_proportion:
        thr = cfg.free_nats_proportion * p_y_entropy
        kl_y_mod = torch.where(kl_y > thr, kl_y,
                               torch.as_tensor(thr, dtype=kl_y.dtype))
    else:
        kl_y_mod = kl_y
    w = warm_up_weight * cfg.kl_weight
    out = {
        "reconstruction_error": rec,
        "kl_divergence_z": kl_z,
        "kl_divergence_y": kl_y,
        "kl_divergence": kl_z + kl_y,
        "lower_bound": rec - (kl_z + kl_y),
        "lower_bound_weighted": rec - w * (kl_z + kl_y_mod),
        "kl_divergence_z_neurons": kl_z_neurons.mean(dim=0),
        "log_p_x_given_z": torch.stack(log_p_all),      # [K, S, B]
        "reconstruction_cell": rec_cell,                # [B]
        "q_y_logits": logits,
        "q_y_probabilities": y.mean(dim=0),
        "y": y,
        "z_mean": z_mean,
        "p_z_means": torch.stack(p_z_means),
        "p_z_variances": torch.stack(p_z_variances),
        "q_z_means": torch.stack(q_z_means),
        "q_z_variances": torch.stack(q_z_variances),
    }
    if evaluation_statistics:
        vm = sum(var_of_mean)
        out["p_x_mean"] = sum(p_x_means)
        out["stddev_of_p_x_given_z_mean"] = torch.sqrt(vm)
        out["p_x_stddev"] = torch.sqrt(sum(mean_of_var) + vm)
    return out


# --------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------

def adam_state(params):
    return {
        "m": OrderedDict((k, torch.zeros_like(v)) for k, v in params.items()),
        "v": OrderedDict((k, torch.zeros_like(v)) for k, v in params.items()),
        "t": 0,
    }


def clip_and_adam(params, grads, state, learning_rate):
    """va:2742-2759: clip every gradient element to [-1, 1], then
    ``tf.train.AdamOptimizer`` (beta1 .9, beta2 .999, epsilon 1e-8):
    ``lr_t = lr*sqrt(1-b2^t)/(1-b1^t)``; ``theta -= lr_t*m/(sqrt(v)+eps)``."""
    state["t"] += 1
    t = state["t"]
    lr_t = (learning_rate * math.sqrt(1.0 - ADAM_BETA2 ** t)
            / (1.0 - ADAM_BETA1 ** t))
    for k in params:
        g = torch.clamp(grads[k], -1.0, 1.0)
        m = state["m"][k]
        v = state["v"][k]
        m.mul_(ADAM_BETA1).add_(g, alpha=1.0 - ADAM_BETA1)
        v.mul_(ADAM_BETA2).addcmul_(g, g, value=1.0 - ADAM_BETA2)
        params[k] = params[k] - lr_t * m / (torch.sqrt(v) + ADAM_EPSILON)
    return params


def gradients(forward, params, *args, **kwargs):
    """Gradients of ``-lower_bound_weighted`` w.r.t. every parameter."""
    leaves = OrderedDict(
        (k, v.detach().clone().requires_grad_(True))
        for k, v in params.items())
    out = forward(leaves, *args, **kwargs)
    loss = -out["lower_bound_weighted"]
    grads = torch.autograd.grad(loss, list(leaves.values()),
                                allow_unused=True)
    grads = OrderedDict(
        (k, g if g is not None else torch.zeros_like(v))
        for (k, v), g in zip(leaves.items(), grads))
    out = {k: (v.detach() if torch.is_tensor(v) else v)
           for k, v in out.items()}
    return out, grads


def vae_train_step(cfg, params, moving, state, x, t, eps, learning_rate,
                   warm_up_weight=1.0):
    new_moving = {}
    out, grads = gradients(
        lambda p: vae_forward(cfg, p, moving, x, t, eps, True,
                              warm_up_weight, new_moving), params)
    params = clip_and_adam(params, grads, state, learning_rate)
    moving = OrderedDict((k, new_moving.get(k, v)) for k, v in moving.items())
    return params, moving, out, grads


def gmvae_train_step(cfg, params, moving, state, x, t, eps, learning_rate,
                     warm_up_weight=1.0):
    new_moving = {}
    out, grads = gradients(
        lambda p: gmvae_forward(cfg, p, moving, x, t, eps, True,
                                warm_up_weight, new_moving), params)
    params = clip_and_adam(params, grads, state, learning_rate)
    moving = OrderedDict((k, new_moving.get(k, v)) for k, v in moving.items())
    return params, moving, out, grads
