"""GPU parity: the VAE's other latent parts -- the Monte-Carlo KL term
(``analytical_kl_term=False``, va:2633-2640) and the unit-variance posterior
(``latent_distribution="unit-variance gaussian"``, du:323-337) -- through the
C ABI vs the fp64 oracle.  Tolerances as in test_gpu_vae_step.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import models as om

from _parity import LL_ATOL, LL_RTOL, close_elementwise
pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _counts(rng, cells, features):
    lam = rng.gamma(0.5, 3.0, size=(1, features))
    x = rng.poisson(lam, size=(cells, features)).astype(np.float64)
    x *= rng.random((cells, features)) > 0.7
    return x


def _close(a, b, rtol=RTOL, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "{}: max err {:.3e} of scale {:.3e}".format(
        what, err, scale)


def _setup(cuda_device, latent, analytical, F=120, L=6, H=(18, 14), B=29,
           n_iw=1, n_mc=1, likelihood="negative binomial", seed=0):
    from scvae_amd.engine import Engine
    eng = Engine(F, L, H, likelihood, batch_norm=True, device=cuda_device,
                 seed=seed, latent_distribution=latent,
                 analytical_kl_term=analytical)
    g = torch.Generator().manual_seed(seed + 1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        elif "POSTERIOR" in name:
            # a posterior that is not close to the prior: the KL terms matter
            p.mul_(3.0)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=tuple(H),
                         likelihood=likelihood, n_iw=n_iw, n_mc=n_mc,
                         latent_distribution=latent,
                         analytical_kl_term=analytical)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    assert list(params) == list(om.vae_parameter_shapes(cfg))
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(_counts(rng, B, F))
    eps = torch.from_numpy(rng.standard_normal((n_iw * n_mc, B, L)))
    return eng, cfg, params, moving, x, eps


def _skip_bias(name):
    return name.endswith("DENSE/biases") and (
        "ENCODER" in name or "DECODER" in name)


@pytest.mark.parametrize("latent,analytical", [
    ("gaussian", False),
    ("unit-variance gaussian", False),
    ("unit-variance gaussian", True),
])
@pytest.mark.parametrize("n_iw,n_mc", [(1, 1), (1, 3), (3, 2)])
def test_train_step_matches_oracle(cuda_device, latent, analytical, n_iw,
                                   n_mc):
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, latent, analytical, n_iw=n_iw, n_mc=n_mc)
    if latent.startswith("unit"):
        assert not any("LOG_SIGMA" in n for n in params)
    B, L = x.shape[0], cfg.latent_size
    xd = x.float().to(cuda_device)
    epsd = eps.float().to(cuda_device)
    klz = torch.zeros(L, device=cuda_device)
    ll = torch.zeros(n_iw * n_mc * B, device=cuda_device)
    sc = eng.step(xd, xd, eps=epsd, training=True, warm_up_weight=0.6,
                  n_iw=n_iw, n_mc=n_mc,
                  outputs={"kl_neurons": klz,
                           "log_p_x_given_z": ll}).cpu().numpy()
    eng.adam_step(1e-3)
    torch.cuda.synchronize()

    state = om.adam_state(params)
    new_params, new_moving, out, grads = om.vae_train_step(
        cfg, dict(params), moving, state, x, x, eps, 1e-3, warm_up_weight=0.6)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    _close(sc[1], out["lower_bound_weighted"], what="lower_bound_weighted")
    _close(sc[2], out["reconstruction_error"], what="reconstruction_error")
    _close(sc[3], out["kl_divergence"], what="kl_divergence")
    _close(klz.cpu(), out["kl_divergence_neurons"], what="kl neurons")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    for name, g in eng.named_gradients().items():
        if _skip_bias(name):
            continue
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)
    for name, p in eng.named_parameters().items():
        if _skip_bias(name):
            continue
        _close(p.cpu(), new_params[name], rtol=2e-4, what="param " + name)


@pytest.mark.parametrize("latent", ["gaussian", "unit-variance gaussian"])
def test_evaluation_step(cuda_device, latent):
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, latent, False, n_iw=2, n_mc=2, L=70)   # L > one wave
    xd = x.float().to(cuda_device)
    epsd = eps.float().to(cuda_device)
    klz = torch.zeros(cfg.latent_size, device=cuda_device)
    sc = eng.step(xd, xd, eps=epsd, training=False, n_iw=2, n_mc=2,
                  outputs={"kl_neurons": klz}).cpu().numpy()
    out = om.vae_forward(cfg, params, moving, x, x, eps, False)
    for i, k in enumerate(("lower_bound", "lower_bound_weighted",
                           "reconstruction_error", "kl_divergence")):
        _close(sc[i], out[k], what=k)
    _close(klz.cpu(), out["kl_divergence_neurons"], what="kl neurons")
    sc = eng.step(xd, xd, training=False, deterministic_z=True).cpu().numpy()
    out = om.vae_forward(cfg, params, moving, x, x, None, False,
                         deterministic_z=True)
    _close(sc[0], out["lower_bound"], what="lower_bound (deterministic z)")
    _close(sc[3], out["kl_divergence"], what="kl (deterministic z)")


def test_model_class_trains(cuda_device, tmp_path):
    """The model class builds the graph the latent distribution names
    (va:186-192: the unit-variance posterior defaults to the sampled KL)."""
    from scvae_amd.data import DataSet
    from scvae_amd.models import VariationalAutoencoder
    from scvae_amd.models.utilities import load_learning_curves
    rng = np.random.default_rng(3)
    x = rng.poisson(2.0, size=(160, 50)).astype(np.float32)
    data = DataSet("toy", values=x, example_names=np.arange(160).astype(str),
                   feature_names=np.arange(50).astype(str), kind="training")
    model = VariationalAutoencoder(
        feature_size=50, latent_size=4, hidden_sizes=[12],
        reconstruction_distribution="poisson",
        latent_distribution="unit-variance gaussian",
        log_directory=str(tmp_path))
    assert model.analytical_kl_term is False
    assert "kl" not in os.path.basename(model.name).split("-")
    assert model.train(data, number_of_epochs=2, minibatch_size=40) == 0
    names = list(model.engine.named_parameters())
    assert "POSTERIOR/MU/DENSE/weights" in names
    assert not any("LOG_SIGMA" in n for n in names)
    curves = load_learning_curves(model)["training"]
    assert np.isfinite(curves["lower_bound"]).all()
    # explicit Monte-Carlo KL for the default posterior
    model = VariationalAutoencoder(
        feature_size=50, latent_size=4, hidden_sizes=[12],
        reconstruction_distribution="poisson", analytical_kl_term=False,
        log_directory=str(tmp_path))
    assert model.train(data, number_of_epochs=1, minibatch_size=40) == 0
