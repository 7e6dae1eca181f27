"""Pins the oracle (which cannot be checked against the reference itself:
"parity unpinned") to independent closed-form implementations."""
import math

import numpy as np
import pytest
import scipy.special as sc
import scipy.stats as st
import torch

from oracle import likelihoods as lk
from oracle import models as om

T = torch.tensor
rng = np.random.default_rng(0)
t_np = np.concatenate([np.zeros(5), rng.poisson(3.0, 40), [250., 4000.]])
a1 = rng.normal(0, 2.5, t_np.size)
a2 = rng.normal(0, 2.5, t_np.size)
a3 = rng.normal(0, 2.5, t_np.size)


def sigmoid(a):
    return 1 / (1 + np.exp(-a))


def test_poisson_matches_scipy():
    got = lk.poisson_log_prob(T(t_np), T(a1)).numpy()
    assert np.allclose(got, st.poisson.logpmf(t_np, np.exp(a1)), rtol=1e-12)


def test_negative_binomial_matches_scipy_and_torch():
    got = lk.negative_binomial_log_prob(T(t_np), T(a1), T(a2)).numpy()
    want = st.nbinom.logpmf(t_np, np.exp(a2), 1 - sigmoid(a1))
    assert np.allclose(got, want, rtol=1e-10, atol=1e-10)
    dist = torch.distributions.NegativeBinomial(
        total_count=torch.exp(T(a2)), probs=torch.sigmoid(T(a1)))
    assert np.allclose(got, dist.log_prob(T(t_np)).numpy(), rtol=1e-10,
                       atol=1e-10)
    m, v = lk.mean_variance("negative binomial", (T(a1), T(a2)))
    assert np.allclose(m.numpy(), st.nbinom.mean(np.exp(a2), 1 - sigmoid(a1)))
    assert np.allclose(v.numpy(), st.nbinom.var(np.exp(a2), 1 - sigmoid(a1)))


def test_tfp_logits_form_of_negative_binomial():
    """TFP 0.7: logits = log p - log1p(-p);
    r*log_sigmoid(-logits) + x*log_sigmoid(logits) - log-normalisation."""
    p, r = sigmoid(a1), np.exp(a2)
    logits = np.log(p) - np.log1p(-p)

    def log_sigmoid(u):
        return -np.logaddexp(0, -u)
    want = (r * log_sigmoid(-logits) + t_np * log_sigmoid(logits)
            + sc.gammaln(r + t_np) - sc.gammaln(1 + t_np) - sc.gammaln(r))
    got = lk.negative_binomial_log_prob(T(t_np), T(a1), T(a2)).numpy()
    assert np.allclose(got, want, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("name", ["zero-inflated poisson",
                                  "zero-inflated negative binomial"])
def test_zero_inflated_composition(name):
    """zero_inflated.py:194-199: where(x > 0, log(1-pi) + logp,
    log(pi + (1-pi) p))."""
    pi = sigmoid(a3)
    if name == "zero-inflated poisson":
        base = st.poisson.pmf(t_np, np.exp(a1))
        got = lk.zero_inflated_poisson_log_prob(T(t_np), T(a3), T(a1)).numpy()
        mean, var = st.poisson.mean(np.exp(a1)), st.poisson.var(np.exp(a1))
        pre = (T(a3), T(a1))
    else:
        base = st.nbinom.pmf(t_np, np.exp(a2), 1 - sigmoid(a1))
        got = lk.zero_inflated_negative_binomial_log_prob(
            T(t_np), T(a3), T(a1), T(a2)).numpy()
        mean = st.nbinom.mean(np.exp(a2), 1 - sigmoid(a1))
        var = st.nbinom.var(np.exp(a2), 1 - sigmoid(a1))
        pre = (T(a3), T(a1), T(a2))
    with np.errstate(divide="ignore"):
        want = np.where(t_np > 0, np.log(1 - pi) + np.log(base),
                        np.log(pi + (1 - pi) * base))
    ok = np.isfinite(want)   # scipy's pmf underflows for the huge counts
    assert np.allclose(got[ok], want[ok], rtol=1e-9, atol=1e-9)
    m, v = lk.mean_variance(name, pre)
    assert np.allclose(m.numpy(), (1 - pi) * mean)
    assert np.allclose(v.numpy(),
                       (1 - pi) * (var + mean ** 2) - ((1 - pi) * mean) ** 2)


def test_support_clips():
    # log_lambda / log_r clipped to [-10, 10], zero gradient outside
    a = T([-12.0, 12.0, 3.0], requires_grad=True, dtype=torch.float64)
    lp = lk.poisson_log_prob(T([2.0, 2.0, 2.0], dtype=torch.float64), a)
    lp.sum().backward()
    assert np.allclose(lp.detach().numpy()[:2],
                       st.poisson.logpmf(2, np.exp([-10.0, 10.0])))
    assert a.grad[0] == 0 and a.grad[1] == 0 and a.grad[2] != 0
    # sigmoid output clipped from below at float32.tiny
    f64 = torch.float64
    lp = lk.negative_binomial_log_prob(
        T([3.0], dtype=f64), T([-200.0], dtype=f64), T([0.0], dtype=f64))
    assert np.isfinite(lp.item())
    assert abs(lp.item() - 3 * lk.LOGIT_OF_TINY) < 1e-9


def test_analytic_gaussian_kl_and_elbo_terms():
    cfg = om.ModelConfig(feature_size=9, latent_size=4, hidden_sizes=(6,),
                         likelihood="poisson", n_iw=3, n_mc=2)
    shapes = om.vae_parameter_shapes(cfg)
    params = om.init_parameters(shapes, seed=1)
    moving = om.init_moving_statistics(shapes)
    g = torch.Generator().manual_seed(0)
    x = torch.poisson(torch.rand(7, 9, generator=g, dtype=torch.float64) * 3)
    eps = torch.randn(6, 7, 4, generator=g, dtype=torch.float64)
    out = om.vae_forward(cfg, params, moving, x, x, eps, True, 0.5)
    mu, ls = out["q_z_mean"], out["q_z_log_sigma"]
    q = torch.distributions.Normal(mu, torch.exp(ls))
    p = torch.distributions.Normal(torch.zeros_like(mu), torch.ones_like(mu))
    kl = torch.distributions.kl_divergence(q, p)
    assert torch.allclose(out["kl_divergence_neurons"], kl.mean(dim=0))
    assert torch.allclose(out["kl_divergence"], kl.mean(dim=0).sum())
    log_p = out["log_p_x_given_z"]                      # [IW, MC, B]
    lw = log_p - kl.sum(dim=-1)
    want = (torch.logsumexp(lw, dim=0) - math.log(3)).mean()
    assert torch.allclose(out["lower_bound"], want)
    lw = log_p - 0.5 * kl.sum(dim=-1)
    want = (torch.logsumexp(lw, dim=0) - math.log(3)).mean()
    assert torch.allclose(out["lower_bound_weighted"], want)
    assert torch.allclose(out["reconstruction_error"], log_p.mean())


def test_batch_norm_semantics_match_torch():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(13, 5, generator=g, dtype=torch.float64)
    params = {"S/DENSE/weights": torch.eye(5, dtype=torch.float64),
              "S/DENSE/biases": torch.zeros(5, dtype=torch.float64),
              "S/BATCH_NORM/beta": torch.randn(5, generator=g,
                                               dtype=torch.float64)}
    moving = {"S/BATCH_NORM/moving_mean": torch.zeros(5, dtype=torch.float64),
              "S/BATCH_NORM/moving_variance": torch.ones(5,
                                                         dtype=torch.float64)}
    new = {}
    y = om.dense_layer(x, params, "S", True, True, moving, new,
                       activation=False)
    rm = torch.zeros(5, dtype=torch.float64)
    rv = torch.ones(5, dtype=torch.float64)
    want = torch.nn.functional.batch_norm(
        x, rm, rv, weight=None, bias=params["S/BATCH_NORM/beta"],
        training=True, momentum=1 - om.BN_DECAY, eps=om.BN_EPSILON)
    assert torch.allclose(y, want)
    # torch also updates running_var with the unbiased variance
    assert torch.allclose(new["S/BATCH_NORM/moving_mean"], rm)
    assert torch.allclose(new["S/BATCH_NORM/moving_variance"], rv)


def test_tf_adam_first_steps():
    p = {"w": T([1.0, -2.0, 0.5])}
    state = om.adam_state(p)
    g = {"w": T([0.3, -5.0, 0.0])}      # -5 is clipped to -1
    p = om.clip_and_adam(p, g, state, 0.1)
    gc = np.array([0.3, -1.0, 0.0])
    m, v = 0.1 * gc, 0.001 * gc ** 2
    lr_t = 0.1 * math.sqrt(1 - 0.999) / (1 - 0.9)
    want = np.array([1.0, -2.0, 0.5]) - lr_t * m / (np.sqrt(v) + 1e-8)
    assert np.allclose(p["w"].numpy(), want)
    p = om.clip_and_adam(p, g, state, 0.1)
    m2, v2 = 0.9 * m + 0.1 * gc, 0.999 * v + 0.001 * gc ** 2
    lr_t = 0.1 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    want = want - lr_t * m2 / (np.sqrt(v2) + 1e-8)
    assert np.allclose(p["w"].numpy(), want)


def test_gmvae_loss_terms():
    cfg = om.ModelConfig(feature_size=8, latent_size=3, hidden_sizes=(5,),
                         likelihood="negative binomial", n_clusters=4,
                         n_iw=2, free_nats_proportion=0.9)
    shapes = om.gmvae_parameter_shapes(cfg)
    params = om.init_parameters(shapes, seed=2)
    moving = om.init_moving_statistics(shapes)
    g = torch.Generator().manual_seed(0)
    x = torch.poisson(torch.rand(6, 8, generator=g, dtype=torch.float64) * 3)
    eps = torch.randn(4, 2, 6, 3, generator=g, dtype=torch.float64)
    out = om.gmvae_forward(cfg, params, moving, x, x, eps, True, 0.7)
    y = out["y"]
    q = torch.distributions.Categorical(probs=y)
    uniform = torch.distributions.Categorical(probs=torch.full((4,), 0.25))
    kl_y = torch.distributions.kl_divergence(q, uniform).mean()
    assert torch.allclose(out["kl_divergence_y"], kl_y)
    rec = (out["log_p_x_given_z"].mean(dim=1) * y.T).sum(dim=0).mean()
    assert torch.allclose(out["reconstruction_error"], rec)
    assert torch.allclose(
        out["lower_bound"],
        rec - out["kl_divergence_z"] - out["kl_divergence_y"])
    thr = 0.9 * math.log(4)
    assert kl_y < thr   # free nats active in this draw
    assert torch.allclose(
        out["lower_bound_weighted"],
        rec - 0.7 * (out["kl_divergence_z"] + thr))


@pytest.mark.parametrize("name", ["poisson", "negative binomial"])
def test_categorised_likelihood_against_explicit_pmf(name):
    """``Categorised`` (distributions/categorised.py:210-263) against its
    definition spelled out with scipy: P(x = k) = pi_k for k < K and
    pi_K * P_dist(x - K) for x >= K; mean and variance by summing that pmf."""
    K = 3
    n = 12
    local = np.random.default_rng(4)
    logits = local.normal(0, 1.5, (n, K + 1))
    b1 = np.clip(local.normal(0, 1.0, n), -3, 3)
    b2 = np.clip(local.normal(0, 1.0, n), -3, 3)
    pre = (T(b1),) if name == "poisson" else (T(b1), T(b2))
    pi = sc.softmax(logits, axis=1)
    if name == "poisson":
        dist = st.poisson(np.exp(b1))
    else:
        dist = st.nbinom(np.exp(b2), 1 - sigmoid(b1))
    counts = np.arange(0, 4000)

    def pmf(x):   # [len(x), n]
        x = np.asarray(x)[:, None]
        head = np.take_along_axis(
            pi.T, np.clip(x, 0, K).astype(int) * np.ones((1, n), int), axis=0)
        return np.where(x < K, head, pi[:, K] * dist.pmf(x - K))
    table = pmf(counts)
    assert np.allclose(table.sum(axis=0), 1.0, atol=1e-9)
    t = np.array([0, 1, 2, 3, 4, 7, 30, 0, 2, 3, 5, 11], dtype=np.float64)
    got = lk.categorised_log_prob(name, T(t), pre, T(logits), K).numpy()
    want = np.log(pmf(t)[np.arange(n), np.arange(n)])
    assert np.allclose(got, want, rtol=1e-10, atol=1e-12)
    mean, var = lk.categorised_mean_variance(name, pre, T(logits), K)
    m1 = (counts[:, None] * table).sum(axis=0)
    m2 = (counts[:, None] ** 2 * table).sum(axis=0)
    assert np.allclose(mean.numpy(), m1, rtol=1e-8)
    assert np.allclose(var.numpy(), m2 - m1 ** 2, rtol=1e-7)


def test_philox_known_answer_vectors():
    """Philox4x32-10 against the known-answer vectors published with
    Random123 (kat_vectors): the generator behind the HIP path's noise and
    dropout masks is pinned to a published golden."""
    from oracle import philox
    vectors = [
        ((0, 0, 0, 0), (0, 0),
         (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2,
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
         (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for counter, key, want in vectors:
        got = philox.philox4x32_10(np.array(counter, dtype=np.uint32),
                                   np.array(key, dtype=np.uint32))
        assert got.tolist() == list(want)
    # batched evaluation equals one-by-one evaluation
    counters = np.array([v[0] for v in vectors], dtype=np.uint32)
    keys = np.array([v[1] for v in vectors], dtype=np.uint32)
    assert philox.philox4x32_10(counters, keys).tolist() == [
        list(v[2]) for v in vectors]
    # the derived draws: uniform in (0, 1), normal moments, mask frequency
    z = philox.standard_normal(2000, 7, 0, 99, 3)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03
    m = philox.dropout_mask(500, 33, 0.8, 5, 2)
    assert set(np.unique(m)) == {np.float32(0), np.float32(1) / np.float32(0.8)}
    assert abs((m > 0).mean() - 0.8) < 0.01


def test_constrained_poisson_matches_scipy():
    """du:218-228: Poisson(rate = softmax(pre) * N) against scipy; the rates
    of a cell sum to its count sum; gradient wrt the logits is t - T*lambda."""
    import scipy.stats as st
    rng = np.random.default_rng(0)
    pre = torch.from_numpy(rng.normal(0, 1.5, (6, 13))).requires_grad_(True)
    t = torch.from_numpy(rng.poisson(2.0, (6, 13)).astype(np.float64))
    N = t.sum(dim=1, keepdim=True) + 3.0   # N need not equal sum(t)
    got = lk.log_prob("constrained poisson", t, (pre,), N)
    lam = np.exp(pre.detach().numpy())
    lam /= lam.sum(axis=1, keepdims=True)
    want = st.poisson.logpmf(t.numpy(), lam * N.numpy())
    assert np.allclose(got.detach().numpy(), want, rtol=1e-12, atol=1e-12)
    mean, var = lk.mean_variance("constrained poisson", (pre,), N)
    assert np.allclose(mean.detach().sum(dim=1), N.reshape(-1))
    assert torch.equal(mean, var)
    got.sum().backward()
    T = t.sum(dim=1, keepdim=True)
    assert np.allclose(pre.grad.numpy(), (t - T * torch.from_numpy(lam)).numpy(),
                       atol=1e-10)
