"""GPU parity: one GMVAE graph execution through the C ABI vs the fp64 oracle."""
import numpy as np
import pytest
import torch

from oracle import models as om

from _parity import LL_ATOL, LL_RTOL, close_elementwise
pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-4, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "{}: max err {:.3e} of scale {:.3e}".format(
        what, err, scale)


def _setup(device, likelihood, F, L, H, B, K, bn, S=1, free_nats=0.0, seed=0):
    from scvae_amd.engine import Engine
    eng = Engine(F, L, H, likelihood, batch_norm=bn, model_type="GMVAE",
                 n_clusters=K, free_nats_proportion=free_nats, device=device,
                 seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=tuple(H),
                         likelihood=likelihood, minibatch_normalisation=bn,
                         n_clusters=K, n_iw=S, n_mc=1,
                         free_nats_proportion=free_nats)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    assert list(params) == list(om.gmvae_parameter_shapes(cfg))
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(seed)
    lam = rng.gamma(0.5, 3.0, size=(1, F))
    x = rng.poisson(lam, size=(B, F)).astype(np.float64)
    x *= rng.random((B, F)) > 0.6
    x = torch.from_numpy(x)
    eps = torch.from_numpy(rng.standard_normal((K, S, B, L)))
    return eng, cfg, params, moving, x, eps


@pytest.mark.parametrize("likelihood,bn,S,free_nats", [
    ("negative binomial", True, 1, 0.0),
    ("zero-inflated negative binomial", True, 2, 0.0),
    ("poisson", False, 1, 0.0),
    ("negative binomial", True, 1, 0.8),   # free-nats threshold active
    ("zero-inflated poisson", True, 1, 0.0),
])
@pytest.mark.parametrize("B,H", [
    (29, (24, 16)),            # the launch chain
    (64, (24, 16)),            # whole 64-row tiles per pass: the tile chain (tilechain.hip with
    (128, (24, 16, 20)),       # groups) where the model has batch norm; one / two tiles per pass,
    (64, (24,)),               # three / two / one hidden layers
])
def test_gmvae_train_step_matches_oracle(cuda_device, likelihood, bn, S,
                                         free_nats, B, H):
    F, L, K = 157, 6, 4
    if not bn and B != 29:
        pytest.skip("the tile chain needs batch norm: nothing new to cover")
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, likelihood, F, L, H, B, K, bn, S, free_nats)
    xd = x.float().to(cuda_device)
    epsd = eps.float().to(cuda_device)
    ll = torch.zeros(K * S * B, device=cuda_device)
    logits = torch.zeros(B, K, device=cuda_device)
    zmean = torch.zeros(B, L, device=cuda_device)
    cstats = torch.zeros(4, K, L, device=cuda_device)
    sc = eng.step(xd, xd, eps=epsd, training=True, n_iw=S, n_mc=1,
                  warm_up_weight=0.6,
                  outputs={"log_p_x_given_z": ll, "q_y_logits": logits,
                           "q_z_mean": zmean, "cluster_stats": cstats}
                  ).cpu().numpy()
    eng.adam_step(1e-3)
    torch.cuda.synchronize()

    state = om.adam_state(params)
    new_params, new_moving, out, grads = om.gmvae_train_step(
        cfg, dict(params), moving, state, x, x, eps, 1e-3, warm_up_weight=0.6)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    _close(sc[1], out["lower_bound_weighted"], what="lower_bound_weighted")
    _close(sc[2], out["reconstruction_error"], what="reconstruction_error")
    _close(sc[3], out["kl_divergence_z"], what="kl_divergence_z")
    _close(sc[4], out["kl_divergence_y"], rtol=2e-4, what="kl_divergence_y")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    _close(logits.cpu(), out["q_y_logits"], what="q_y_logits")
    _close(zmean.cpu(), out["z_mean"], what="z_mean")
    _close(cstats[0].cpu(), out["p_z_means"], what="p_z_means")
    _close(cstats[1].cpu(), out["p_z_variances"], what="p_z_variances")
    _close(cstats[2].cpu(), out["q_z_means"], what="q_z_means")
    _close(cstats[3].cpu(), out["q_z_variances"], what="q_z_variances")
    for name, g in eng.named_gradients().items():
        if bn and name.endswith("DENSE/biases") and "LAYER_" in name:
            assert g.abs().max().item() < 1e-5, name
            continue
        _close(g.cpu(), grads[name], rtol=3e-4, what="grad " + name)
    for name, p in eng.named_parameters().items():
        if bn and name.endswith("DENSE/biases") and "LAYER_" in name:
            continue
        got, want = p.cpu(), new_params[name]
        if bn and name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            # the one-hot rows W[F+k] are cancelled by the per-pass batch norm:
            # zero gradient, Adam only amplifies rounding noise there
            assert eng.gradient(name)[F:].abs().max().item() < 1e-5
            got, want = got[:F], want[:F]
        _close(got, want, rtol=3e-4, what="param " + name)
    for name, m in eng.named_moving_statistics().items():
        _close(m.cpu(), new_moving[name], rtol=2e-5, what="moving " + name)


def test_gmvae_evaluation_statistics(cuda_device):
    F, L, H, B, K, S = 120, 5, (16, 16), 17, 3, 2
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, "negative binomial", F, L, H, B, K, True, S)
    xd = x.float().to(cuda_device)
    epsd = eps.float().to(cuda_device)
    outs = {k: torch.zeros(B, F, device=cuda_device) for k in (
        "p_x_mean", "p_x_stddev", "stddev_of_p_x_given_z_mean")}
    sc = eng.step(xd, xd, eps=epsd, training=False, n_iw=S, n_mc=1,
                  outputs=outs).cpu().numpy()
    out = om.gmvae_forward(cfg, params, moving, x, x, eps, False,
                           evaluation_statistics=True)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    for k in outs:
        _close(outs[k].cpu(), out[k], rtol=2e-4, what=k)


@pytest.mark.parametrize("method,free_nats", [("custom", 0.0), ("learn", 0.0),
                                              ("learn", 0.95),
                                              ("custom", 0.95)])
def test_non_uniform_prior_over_clusters(cuda_device, method, free_nats):
    """p(y) given (`custom`) or trainable (`learn`, variable Y/P/LOGITS):
    KL(q(y|x) || p(y)), the free-nats threshold on H[p(y)] and the gradient of
    the prior logits (gm:2794-2808, 3242-3261)."""
    from scvae_amd.engine import Engine
    F, L, H, B, K = 60, 4, (12,), 26, 4
    prior = (0.1, 0.2, 0.3, 0.4)
    eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                 model_type="GMVAE", n_clusters=K, device=cuda_device, seed=1,
                 free_nats_proportion=free_nats,
                 prior_probabilities_method=method,
                 prior_probabilities=prior if method == "custom" else None)
    g = torch.Generator().manual_seed(3)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="negative binomial", n_clusters=K,
                         free_nats_proportion=free_nats,
                         prior_probabilities_method=method,
                         prior_probabilities=prior if method == "custom"
                         else ())
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    assert list(params) == list(om.gmvae_parameter_shapes(cfg))
    assert (list(params)[0] == "Y/P/LOGITS") == (method == "learn")
    if method == "custom":
        assert torch.allclose(eng.prior_logits.cpu().double(),
                              torch.log(torch.tensor(prior,
                                                     dtype=torch.float64)),
                              atol=1e-6)
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(2)
    x = torch.from_numpy((rng.poisson(2.0, (B, F))
                          * (rng.random((B, F)) > 0.5)).astype(np.float64))
    eps = torch.from_numpy(rng.standard_normal((K, 1, B, L)))
    sc = eng.step(x.float().to(cuda_device), x.float().to(cuda_device),
                  eps=eps.float().to(cuda_device), training=True,
                  warm_up_weight=0.8).cpu().numpy()
    out, grads = om.gradients(
        lambda p: om.gmvae_forward(cfg, p, moving, x, x, eps, True, 0.8),
        params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    _close(sc[1], out["lower_bound_weighted"], what="lower_bound_weighted")
    _close(sc[4], out["kl_divergence_y"], rtol=2e-4, what="kl_divergence_y")
    for name, got in eng.named_gradients().items():
        if name.endswith("DENSE/biases") and "LAYER_" in name:
            continue
        want = grads[name]
        got = got.cpu()
        if name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            got, want = got[:F], want[:F]
        _close(got, want, rtol=5e-4, what="grad " + name)
    if method == "custom":   # the fixed logits never move
        before = eng.prior_logits.clone()
        eng.adam_step(1e-2)
        assert torch.equal(before, eng.prior_logits)


def test_legacy_gaussian_mixture_is_the_same_graph(cuda_device):
    """latent_mode 4: scope MODIFIED_GAUSSIAN instead of SOFTPLUS_GAUSSIAN
    (du:349-352), identical arithmetic."""
    from scvae_amd.engine import Engine
    F, L, H, B, K = 60, 4, (12,), 15, 3
    kwargs = dict(batch_norm=True, model_type="GMVAE", n_clusters=K,
                  device=cuda_device, seed=3)
    modern = Engine(F, L, H, "negative binomial", **kwargs)
    legacy = Engine(F, L, H, "negative binomial",
                    latent_distribution="legacy gaussian mixture", **kwargs)
    names = list(legacy.named_parameters())
    assert "Z/P/MODIFIED_GAUSSIAN/SOFTPLUS_SCALE/DENSE/biases" in names
    assert [n.replace("MODIFIED", "SOFTPLUS") for n in names] == list(
        modern.named_parameters())
    legacy.params.copy_(modern.params)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.poisson(1.5, size=(B, F)).astype(np.float32)).to(
        cuda_device)
    eps = torch.randn(K, 1, B, L, device=cuda_device)
    a = modern.step(x, x, eps=eps, training=True).clone()
    b = legacy.step(x, x, eps=eps, training=True).clone()
    assert torch.equal(a, b)
    assert torch.equal(modern.grads, legacy.grads)
