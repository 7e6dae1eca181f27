"""GPU parity: dropout of the dense layers' input connections
(``dropout_keep_probabilities``, mu:45-50; va:2221-2232, 2281-2289, 2444-2455,
2475-2518) through the C ABI vs the fp64 oracle.

The masks are explicit on both sides: the HIP path derives them from
``dropout_seed`` (Philox, include/scvae_hip.h: scvae_dropout_apply), the test
reads the very same masks back through that entry and hands them to the oracle.
"""
import numpy as np
import pytest
import torch

from oracle import models as om

from _parity import LL_ATOL, LL_RTOL, close_elementwise
pytestmark = pytest.mark.gpu

RTOL = 1e-4
SEED = 0x1234ABCD5678


def _close(a, b, rtol=RTOL, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "{}: max err {:.3e} of scale {:.3e}".format(
        what, err, scale)


def _counts(rng, cells, features):
    lam = rng.gamma(0.5, 3.0, size=(1, features))
    x = rng.poisson(lam, size=(cells, features)).astype(np.float64)
    x *= rng.random((cells, features)) > 0.6
    return x


def _skip_bias(name):
    return name.endswith("DENSE/biases") and (
        "ENCODER" in name or "DECODER" in name)


def test_mask_statistics(cuda_device):
    from scvae_amd.engine import Engine
    eng = Engine(8, 2, (4,), "poisson", device=cuda_device)
    m = eng.dropout_mask(3, 2000, 301, 0.7, SEED).cpu().numpy()
    assert set(np.unique(m)) == {0.0, np.float32(1 / 0.7)}
    kept = (m > 0)
    assert abs(kept.mean() - 0.7) < 0.005
    assert abs(kept.mean(axis=0) - 0.7).max() < 0.06
    assert abs(kept.mean(axis=1) - 0.7).max() < 0.12
    # neighbouring elements are uncorrelated
    c = np.corrcoef(kept[:, :-1].ravel(), kept[:, 1:].ravel())[0, 1]
    assert abs(c) < 0.01
    # another site, another seed: different masks; same arguments: the same
    again = eng.dropout_mask(3, 2000, 301, 0.7, SEED).cpu().numpy()
    assert (again == m).all()
    for other in (eng.dropout_mask(4, 2000, 301, 0.7, SEED),
                  eng.dropout_mask(3, 2000, 301, 0.7, SEED + 1)):
        agree = ((other.cpu().numpy() > 0) == kept).mean()
        assert abs(agree - (0.49 + 0.09)) < 0.01


def _vae_masks(eng, cfg, B, R, keeps, k_max=0):
    """{oracle layer scope: mask / keep} of a VAE step with seed SEED."""
    kh, kx, kz = keeps
    H = list(cfg.hidden_sizes)
    n = len(H)
    masks = {}
    n_enc = n if cfg.inference_architecture == "MLP" else 0
    n_dec = n if cfg.generative_architecture == "MLP" else 0
    width = cfg.feature_size
    for i in range(n_enc):
        keep = kx if i == 0 else kh
        if keep:
            masks["ENCODER/{}".format(i + 1)] = eng.dropout_mask(
                i, B, width, keep, SEED)
        width = H[i]
    if kh:
        masks["POSTERIOR/MU"] = eng.dropout_mask(16, B, width, kh, SEED)
        masks["POSTERIOR/LOG_SIGMA"] = eng.dropout_mask(17, B, width, kh, SEED)
    width = cfg.latent_size + cfg.decoder_extra_size
    for i in range(n_dec):
        keep = kz if i == 0 else kh
        if keep:
            masks["DECODER/{}".format(n - i)] = eng.dropout_mask(
                32 + i, R, width, keep, SEED)
        width = H[n - 1 - i]
    if kh:
        for j, head in enumerate(cfg.heads):
            masks["X_TILDE/" + head.upper()] = eng.dropout_mask(
                48 + j, R, width, kh, SEED)
        if k_max:
            masks["X_TILDE/P_K"] = eng.dropout_mask(51, R, width, kh, SEED)
    return {k: v.cpu().double() for k, v in masks.items()}


@pytest.mark.parametrize("keeps", [
    (0.8, 0.0, 0.0), (0.0, 0.9, 0.0), (0.0, 0.0, 0.7), (0.8, 0.9, 0.7)])
@pytest.mark.parametrize("likelihood,k_max,n_iw,n_mc", [
    ("negative binomial", 0, 1, 1),
    ("zero-inflated negative binomial", 0, 2, 2),
    ("poisson", 3, 1, 2),
])
def test_vae_train_step_matches_oracle(cuda_device, keeps, likelihood, k_max,
                                       n_iw, n_mc):
    _vae_step_case(cuda_device, keeps, likelihood, k_max, n_iw, n_mc,
                   96, 5, (20, 16), 31)


@pytest.mark.parametrize("likelihood", [
    "poisson", "negative binomial", "zero-inflated poisson",
    "zero-inflated negative binomial"])
@pytest.mark.parametrize("H,B", [((100, 30), 300), ((40, 24), 200)])
def test_head_dropout_in_the_fused_kernel(cuda_device, likelihood, H, B):
    """Hidden-layer dropout reaches the likelihood heads, each with its own mask
    (mu:45-50 inside every X_TILDE dense_layer, va:2475-2488): with one
    likelihood pass per step the bf16x9 head kernel takes it (one dropped-out
    copy of the decoder output per head, the heads' parts of dd through their
    masks).  Several row tiles, decoder widths 100 and 40, all four count
    likelihoods, against the fp64 oracle with the same masks."""
    from scvae_amd import _lib
    lib = _lib.load()
    kind, heads = _lib.LIKELIHOOD_KINDS[likelihood]
    assert lib.scvae_decoder_train_kernel(kind, H[0], 1) == 3
    _vae_step_case(cuda_device, (0.8, 0.0, 0.0), likelihood, 0, 1, 1,
                   150, 7, H, B)
    _vae_step_case(cuda_device, (0.6, 0.0, 0.9), likelihood, 0, 1, 2,
                   70, 3, H, B // 2)


def _vae_step_case(cuda_device, keeps, likelihood, k_max, n_iw, n_mc, F, L, H,
                   B):
    from scvae_amd.engine import Engine
    S = n_iw * n_mc
    eng = Engine(F, L, H, likelihood, device=cuda_device, k_max=k_max,
                 dropout_keep_probabilities=keeps)
    g = torch.Generator().manual_seed(1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood=likelihood, n_iw=n_iw, n_mc=n_mc,
                         k_max=k_max)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(0)
    x = torch.from_numpy(_counts(rng, B, F))
    eps = torch.from_numpy(rng.standard_normal((S, B, L)))
    masks = _vae_masks(eng, cfg, B, S * B, keeps, k_max)
    assert masks

    xd = x.float().to(cuda_device)
    ll = torch.zeros(S * B, device=cuda_device)
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=True,
                  n_iw=n_iw, n_mc=n_mc, dropout_seed=SEED,
                  outputs={"log_p_x_given_z": ll}).cpu().numpy()
    torch.cuda.synchronize()
    new_moving = {}
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, x, x, eps, True, 1.0,
                                 new_moving, dropout=masks), params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    _close(sc[2], out["reconstruction_error"], what="reconstruction_error")
    _close(sc[3], out["kl_divergence"], what="kl_divergence")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    for name, g in eng.named_gradients().items():
        if _skip_bias(name):
            continue
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)
    for name, m in eng.named_moving_statistics().items():
        _close(m.cpu(), new_moving[name], rtol=1e-5, what="moving " + name)

    # evaluation: is_training = False, no dropout, the fused path again
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=False,
                  n_iw=n_iw, n_mc=n_mc).cpu().numpy()
    moving = {k: v.detach().cpu().double()   # the training step moved them
              for k, v in eng.named_moving_statistics().items()}
    out = om.vae_forward(cfg, params, moving, x, x, eps, False, dropout=masks)
    _close(sc[0], out["lower_bound"], what="lower_bound (evaluation)")


@pytest.mark.parametrize("inference,generative", [
    ("LFM", "MLP"), ("MLP", "LFM")])
def test_linear_factor_model_with_dropout(cuda_device, inference, generative):
    """LFM sides: the parameter layers sit directly on x / z and drop those."""
    from scvae_amd.engine import Engine
    F, L, H, B = 64, 6, (14,), 23
    keeps = (0.75, 0.85, 0.65)
    eng = Engine(F, L, H, "negative binomial", device=cuda_device,
                 inference_architecture=inference,
                 generative_architecture=generative,
                 dropout_keep_probabilities=keeps)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="negative binomial",
                         inference_architecture=inference,
                         generative_architecture=generative)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(2)
    x = torch.from_numpy(_counts(rng, B, F))
    eps = torch.from_numpy(rng.standard_normal((1, B, L)))
    masks = _vae_masks(eng, cfg, B, B, keeps)
    xd = x.float().to(cuda_device)
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=True,
                  dropout_seed=SEED).cpu().numpy()
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, x, x, eps, True,
                                 dropout=masks), params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    for name, g in eng.named_gradients().items():
        if _skip_bias(name):
            continue
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)


def test_default_seed_changes_every_step(cuda_device):
    from scvae_amd.engine import Engine
    F, L, H, B = 40, 3, (10,), 16
    eng = Engine(F, L, H, "poisson", device=cuda_device,
                 dropout_keep_probabilities=(0.5,))
    assert eng.dropout_keep_probabilities == (0.5, 0.0, 0.0, 0.0)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(_counts(rng, B, F)).float().to(cuda_device)
    eps = torch.randn(1, B, L, device=cuda_device)
    a = eng.step(x, x, eps=eps, training=True).clone()
    b = eng.step(x, x, eps=eps, training=True).clone()
    c = eng.step(x, x, eps=eps, training=True, dropout_seed=7).clone()
    d = eng.step(x, x, eps=eps, training=True, dropout_seed=7).clone()
    assert a[0].item() != b[0].item()
    assert c[0].item() == d[0].item()
    # p in {0, 1, False}: no dropout at all
    eng = Engine(F, L, H, "poisson", device=cuda_device,
                 dropout_keep_probabilities=(1, False, 0))
    assert not eng.uses_dropout


# ------------------------------- GMVAE -------------------------------------

def _gmvae_masks(eng, cfg, B, S, keeps, k_max=0):
    """{oracle layer scope: mask / keep}; the layers under Z/ and X/ carry a
    leading axis of K passes (the HIP path stacks the passes' rows)."""
    kh, kx, kz, ky = keeps
    H = list(cfg.hidden_sizes)
    K, F, L = cfg.n_clusters, cfg.feature_size, cfg.latent_size
    masks = {}

    def add(scope, site, rows, width, keep, passes=0):
        if not keep:
            return
        if passes:
            m = eng.dropout_mask(site, passes * rows, width, keep, SEED)
            masks[scope] = m.view(passes, rows, width).cpu().double()
        else:
            masks[scope] = eng.dropout_mask(
                site, rows, width, keep, SEED).cpu().double()

    width = F
    for i, h in enumerate(H):
        add("Y/CATEGORICAL/ENCODER/LAYER_{}".format(i + 1), 64 + i, B, width,
            kx if i == 0 else kh)
        width = h
    add("Y/CATEGORICAL/LOGITS", 80, B, width, kh)
    width = F + K
    for i, h in enumerate(H):
        add("Z/Q/ENCODER/LAYER_{}".format(i + 1), i, B, width,
            kx if i == 0 else kh, passes=K)
        width = h
    add("Z/Q/SOFTPLUS_GAUSSIAN/MEAN", 16, B, width, kh, passes=K)
    add("Z/Q/SOFTPLUS_GAUSSIAN/SOFTPLUS_SCALE", 17, B, width, kh, passes=K)
    if ky:   # [K passes, K one-hot columns]
        for scope, site in (("MEAN", 24), ("SOFTPLUS_SCALE", 25)):
            masks["Z/P/SOFTPLUS_GAUSSIAN/" + scope] = eng.dropout_mask(
                site, K, K, ky, SEED).cpu().double()
    width = L + cfg.decoder_extra_size
    for i, h in enumerate(H[::-1]):
        add("X/DECODER/LAYER_{}".format(i + 1), 32 + i, S * B, width,
            kz if i == 0 else kh, passes=K)
        width = h
    for j, head in enumerate(cfg.heads):
        add("X/DISTRIBUTION/" + head.upper(), 48 + j, S * B, width, kh,
            passes=K)
    if k_max:
        add("X/DISTRIBUTION/P_K", 51, S * B, width, kh, passes=K)
    return masks


@pytest.mark.parametrize("keeps", [
    (0.8, 0.0, 0.0, 0.0), (0.0, 0.9, 0.0, 0.0), (0.0, 0.0, 0.7, 0.0),
    (0.0, 0.0, 0.0, 0.5), (0.8, 0.9, 0.7, 0.6)])
@pytest.mark.parametrize("likelihood,k_max,S,bn", [
    ("negative binomial", 0, 1, True),
    ("zero-inflated poisson", 0, 2, True),
    ("poisson", 2, 1, False),
])
def test_gmvae_train_step_matches_oracle(cuda_device, keeps, likelihood,
                                         k_max, S, bn):
    from scvae_amd.engine import Engine
    F, L, H, B, K = 90, 5, (18, 14), 27, 4
    eng = Engine(F, L, H, likelihood, batch_norm=bn, model_type="GMVAE",
                 n_clusters=K, device=cuda_device, k_max=k_max,
                 dropout_keep_probabilities=keeps)
    g = torch.Generator().manual_seed(1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood=likelihood, minibatch_normalisation=bn,
                         n_clusters=K, n_iw=S, n_mc=1, k_max=k_max)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(0)
    x = torch.from_numpy(_counts(rng, B, F))
    eps = torch.from_numpy(rng.standard_normal((K, S, B, L)))
    masks = _gmvae_masks(eng, cfg, B, S, keeps, k_max)
    assert masks

    xd = x.float().to(cuda_device)
    ll = torch.zeros(K * S * B, device=cuda_device)
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=True,
                  n_iw=S, n_mc=1, warm_up_weight=0.7, dropout_seed=SEED,
                  outputs={"log_p_x_given_z": ll}).cpu().numpy()
    torch.cuda.synchronize()
    new_moving = {}
    out, grads = om.gradients(
        lambda p: om.gmvae_forward(cfg, p, moving, x, x, eps, True, 0.7,
                                   new_moving, dropout=masks), params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    _close(sc[1], out["lower_bound_weighted"], what="lower_bound_weighted")
    _close(sc[3], out["kl_divergence_z"], what="kl_divergence_z")
    _close(sc[4], out["kl_divergence_y"], rtol=2e-4, what="kl_divergence_y")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    for name, g in eng.named_gradients().items():
        if bn and name.endswith("DENSE/biases") and "LAYER_" in name:
            continue
        got, want = g.cpu(), grads[name]
        if (bn and name == "Z/Q/ENCODER/LAYER_1/DENSE/weights"
                and not keeps[1]):
            got, want = got[:F], want[:F]   # one-hot rows: cancelled by BN
        _close(got, want, rtol=3e-4, what="grad " + name)
    for name, m in eng.named_moving_statistics().items():
        _close(m.cpu(), new_moving[name], rtol=2e-5, what="moving " + name)

    # evaluation: no dropout
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=False,
                  n_iw=S, n_mc=1).cpu().numpy()
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    out = om.gmvae_forward(cfg, params, moving, x, x, eps, False,
                           dropout=masks)
    _close(sc[0], out["lower_bound"], what="lower_bound (evaluation)")


# ----------------------------- model classes --------------------------------

def _toy_data(n=120, F=40):
    from scvae_amd.data import DataSet
    rng = np.random.default_rng(5)
    x = rng.poisson(2.0, size=(n, F)).astype(np.float32)
    return DataSet("toy", values=x, example_names=np.arange(n).astype(str),
                   feature_names=np.arange(F).astype(str), kind="training")


def test_model_classes_train_with_dropout(cuda_device, tmp_path):
    from scvae_amd.models import (
        GaussianMixtureVariationalAutoencoder, VariationalAutoencoder)
    from scvae_amd.models.utilities import load_learning_curves
    data = _toy_data()
    for cls, keeps, extra in (
            (VariationalAutoencoder, [0.9, 0.8, 0.7], {}),
            (GaussianMixtureVariationalAutoencoder, [0.9, 0.8, 0.7, 0.6],
             {"number_of_latent_clusters": 3})):
        model = cls(feature_size=40, latent_size=3, hidden_sizes=[12, 10],
                    reconstruction_distribution="negative binomial",
                    dropout_keep_probabilities=keeps,
                    log_directory=str(tmp_path), **extra)
        assert "dropout_" + "_".join(map(str, keeps)) in model.name
        assert model.train(data, data, number_of_epochs=3,
                           minibatch_size=30, learning_rate=1e-2) == 0
        assert model.engine.uses_dropout
        curves = load_learning_curves(model)
        # the epoch evaluations run without dropout and improve
        lb = curves["validation"]["lower_bound"]
        assert np.isfinite(lb).all() and lb[-1] > lb[0]
        out = model.evaluate(data, minibatch_size=30,
                             output_versions="reconstructed")
        assert np.isfinite(out.values).all()
