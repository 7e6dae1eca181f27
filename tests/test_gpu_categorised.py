"""Piecewise categorical likelihood (-k; `Categorised`, distributions/categorised.py:210-263,
va:2507-2532): step parity with the oracle for both models, evaluation statistics, decoder-only
entry, and the model classes end to end."""
import numpy as np
import pytest
import torch

from oracle import models as om

from _parity import LL_ATOL, LL_RTOL, close_elementwise
pytestmark = pytest.mark.gpu


def _close(a, b, rtol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "{}: max err {:.3e} of scale {:.3e}".format(
        what, err, scale)


@pytest.mark.parametrize("model_type,likelihood,S", [
    ("VAE", "negative binomial", 1), ("VAE", "poisson", 3),
    ("GMVAE", "negative binomial", 1), ("GMVAE", "poisson", 2)])
def test_step_with_categorised_likelihood(cuda_device, model_type, likelihood,
                                          S):
    from scvae_amd.engine import Engine
    F, L, H, B, K, KM = 70, 4, (14, 12), 21, 3, 3
    gm = model_type == "GMVAE"
    eng = Engine(F, L, H, likelihood, batch_norm=True, model_type=model_type,
                 n_clusters=K, device=cuda_device, seed=2, k_max=KM)
    g = torch.Generator().manual_seed(7)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood=likelihood, n_clusters=K, n_iw=S, n_mc=1,
                         k_max=KM)
    shapes = (om.gmvae_parameter_shapes if gm else om.vae_parameter_shapes)(cfg)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    assert [(k, tuple(v.shape)) for k, v in params.items()] == [
        (k, tuple(v)) for k, v in shapes.items()]
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(3)
    # counts on both sides of k_max, including exactly k_max
    x = rng.poisson(2.5, (B, F)) * (rng.random((B, F)) > 0.4)
    x[0, :6] = [0, 1, 2, 3, 4, 40]
    x = torch.from_numpy(x.astype(np.float64))
    eps = torch.from_numpy(rng.standard_normal(
        (K, S, B, L) if gm else (S, B, L)))
    xd, epsd = x.float().to(cuda_device), eps.float().to(cuda_device)
    rows = (K if gm else 1) * S * B
    ll = torch.zeros(rows, device=cuda_device)
    sc = eng.step(xd, xd, eps=epsd, training=True, n_iw=S, n_mc=1,
                  outputs={"log_p_x_given_z": ll}).cpu().numpy()
    torch.cuda.synchronize()
    forward = om.gmvae_forward if gm else om.vae_forward
    out, grads = om.gradients(
        lambda p: forward(cfg, p, moving, x, x, eps, True), params)
    _close(sc[0], out["lower_bound"], 1e-4, "lower_bound")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    for name, g in eng.named_gradients().items():
        if name.endswith("DENSE/biases") and (
                "LAYER_" in name or "ENCODER/" in name or "DECODER/" in name):
            continue   # bias under batch norm: zero gradient
        if gm and name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            g, want = g[:F], grads[name][:F]
        else:
            want = grads[name]
        _close(g.cpu(), want, 3e-4, "grad " + name)

    # evaluation statistics (mean / variance of the categorised distribution)
    outs = {k: torch.zeros(B, F, device=cuda_device) for k in (
        "p_x_mean", "p_x_stddev", "stddev_of_p_x_given_z_mean")}
    eng.step(xd, xd, eps=epsd, training=False, n_iw=S, n_mc=1, outputs=outs)
    moving = {k: v.detach().cpu().double()    # updated by the training step
              for k, v in eng.named_moving_statistics().items()}
    want = forward(cfg, params, moving, x, x, eps, False,
                   evaluation_statistics=True)
    for k in outs:
        if S == 1 and not gm and k == "stddev_of_p_x_given_z_mean":
            # exactly zero for one sample; the kernel's two evaluations of the
            # mean may differ in the last bit
            assert outs[k].abs().max().item() <= 1e-5 * float(
                want["p_x_mean"].abs().max())
            continue
        _close(outs[k].cpu(), want[k], 3e-4, k)
    # decoder-only entry
    z = torch.from_numpy(rng.standard_normal((9, L)))
    got = eng.decode(z.float().to(cuda_device)).cpu()
    _close(got, om.decode_mean(cfg, params, moving, z, model_type), 2e-4,
           "decode")


@pytest.mark.parametrize("model_type,S", [("VAE", 1), ("GMVAE", 1), ("VAE", 3),
                                          ("GMVAE", 2)])
@pytest.mark.parametrize("likelihood", ["negative binomial", "poisson"])
@pytest.mark.parametrize("KM", [1, 2])
def test_fused_categorised_training_step(cuda_device, model_type, likelihood,
                                         KM, S):
    """-k with k = 1, 2 in a training step: two launches of the bf16x9 head
    kernel (``decoder_fused_train_cat``: the count distribution on shifted,
    masked targets + the k + 1 class logits of every gene as heads of a
    categorical kind, read with a stride from the P_K matrix) -- against the
    fp64 oracle and against the unfused kernels of the same build.  Several
    row tiles, a ragged last gene strip, counts on both sides of k; S > 1:
    importance-weighted (VAE: the weights come from a first pass on the fused
    forward half) / several samples per cell (GMVAE)."""
    from scvae_amd.engine import Engine
    F, L, H, B, K = 150, 5, (30, 20), 100, 3
    gm = model_type == "GMVAE"
    rng = np.random.default_rng(5 + KM)
    x = rng.poisson(1.5, (B, F)) * (rng.random((B, F)) > 0.4)
    x[0, :7] = [0, 1, 2, 3, 4, 40, 300]
    x = torch.from_numpy(x.astype(np.float64))
    eps = torch.from_numpy(rng.standard_normal((K, S, B, L) if gm else (S, B, L)))
    xd, epsd = x.float().to(cuda_device), eps.float().to(cuda_device)
    rows = (K if gm else 1) * S * B
    results = {}
    for fused in (True, False):
        eng = Engine(F, L, H, likelihood, batch_norm=True,
                     model_type=model_type, n_clusters=K, device=cuda_device,
                     seed=2, k_max=KM)
        g = torch.Generator().manual_seed(7)
        for name, p in eng.named_parameters().items():
            if not name.endswith("weights"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        eng.set_fused(fused)
        eng.reserve(B, S)
        assert eng.fused_categorised == fused
        ll = torch.zeros(rows, device=cuda_device)
        sc = eng.step(xd, xd, eps=epsd, training=True, n_iw=S, n_mc=1,
                      outputs={"log_p_x_given_z": ll}).cpu().numpy()
        torch.cuda.synchronize()
        # ... and an evaluation step (the forward half: two launches of the fp32
        # forward kernel) with the moving statistics the training step left
        ll_e = torch.zeros(rows, device=cuda_device)
        sc_e = eng.step(xd, xd, eps=epsd, training=False, n_iw=S, n_mc=1,
                        outputs={"log_p_x_given_z": ll_e}).cpu().numpy()
        torch.cuda.synchronize()
        results[fused] = (sc, ll.cpu(), {k: v.clone().cpu() for k, v in
                                         eng.named_gradients().items()},
                          sc_e, ll_e.cpu())
        if fused:
            params = {k: v.detach().cpu().double()
                      for k, v in eng.named_parameters().items()}
            moving = {k: v.detach().cpu().double()
                      for k, v in eng.named_moving_statistics().items()}
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood=likelihood, n_clusters=K, k_max=KM, n_iw=S,
                         n_mc=1)
    # (the moving statistics were read before the step updated them? no: after;
    #  a training step normalises with batch statistics, they do not enter)
    forward = om.gmvae_forward if gm else om.vae_forward
    out, grads = om.gradients(
        lambda p: forward(cfg, p, moving, x, x, eps, True), params)
    sc, ll, dev, sc_e, ll_e = results[True]
    out_e = forward(cfg, params, moving, x, x, eps, False)
    _close(sc_e[0], out_e["lower_bound"], 1e-4, "lower_bound (evaluation)")
    close_elementwise(ll_e, out_e["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll (evaluation)")
    close_elementwise(ll_e, results[False][4], rtol=LL_RTOL, atol=LL_ATOL,
                      what="fused vs unfused per-cell ll (evaluation)")
    _close(sc[0], out["lower_bound"], 1e-4, "lower_bound")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    for name, g in dev.items():
        if name.endswith("DENSE/biases") and (
                "LAYER_" in name or "ENCODER/" in name or "DECODER/" in name):
            continue   # bias under batch norm: zero gradient
        want = grads[name]
        if gm and name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            g, want = g[:F], want[:F]
        _close(g, want, 3e-4, "grad " + name)
        # ... and the unfused kernels agree with the fused ones
        u = results[False][2][name]
        if gm and name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            u = u[:F]
        _close(g, u, 3e-4, "fused vs unfused grad " + name)
    close_elementwise(ll, results[False][1], rtol=LL_RTOL, atol=LL_ATOL,
                      what="fused vs unfused per-cell ll")


@pytest.mark.parametrize("model_type", ["VAE", "GMVAE"])
def test_fused_categorised_step_with_fewer_genes_than_hidden_units(
        cuda_device, model_type):
    """The fused -k launches keep the second launch's ll and dd in the plan's
    buffer of the unfused path's class logits, [rows, F (k + 1)]: with fewer
    genes than hidden units that is LESS than rows + rows x H, and what follows
    it in the workspace are the dropped-out layer inputs the weight gradients
    read (found by tools/fuzz_options.py, seed 5041: wrong dW of the layers
    with dropout, everything else right).  Same step, same dropout seed, on the
    fused and on the unfused kernels."""
    from scvae_amd.engine import Engine
    F, L, H, B, K, KM = 10, 2, (20, 28), 13, 3, 1
    gm = model_type == "GMVAE"
    keeps = (0.0, 0.0, 0.9, 0.9) if gm else (0.0, 0.0, 0.9)
    rng = np.random.default_rng(11)
    x = rng.poisson(2.0, (B, F)) * (rng.random((B, F)) < 0.5)
    xd = torch.from_numpy(x.astype(np.float32)).to(cuda_device)
    eps = torch.from_numpy(rng.standard_normal(
        (K, 1, B, L) if gm else (1, B, L)).astype(np.float32)).to(cuda_device)
    results = {}
    for fused in (True, False):
        eng = Engine(F, L, H, "poisson", batch_norm=True, model_type=model_type,
                     n_clusters=K, device=cuda_device, seed=4, k_max=KM,
                     dropout_keep_probabilities=keeps)
        eng.set_fused(fused)
        eng.reserve(B, 1)
        assert eng.fused_categorised == fused
        sc = eng.step(xd, xd, eps=eps, training=True, dropout_seed=99).cpu().numpy()
        torch.cuda.synchronize()
        results[fused] = (sc, {k: v.clone().cpu()
                               for k, v in eng.named_gradients().items()})
    _close(results[True][0][0], results[False][0][0], 1e-5, "lower_bound")
    for name, g in results[True][1].items():
        u = results[False][1][name]
        if float(u.abs().max()) == 0.0:
            assert float(g.abs().max()) == 0.0, name
            continue
        _close(g, u, 3e-4, "fused vs unfused grad " + name)


def test_categorised_model_trains_and_evaluates(tmp_path, cuda_device, capsys):
    from scvae_amd.data import DataSet
    from scvae_amd.models import VariationalAutoencoder
    rng = np.random.default_rng(1)
    n, F = 64, 24
    values = (rng.poisson(2.0, (n, F)) * (rng.random((n, F)) > 0.5)).astype(
        np.float32)
    data = DataSet("counts", values=values,
                   example_names=np.array(["c%d" % i for i in range(n)]),
                   feature_names=np.array(["g%d" % i for i in range(F)]))
    model = VariationalAutoencoder(
        feature_size=F, latent_size=3, hidden_sizes=[8],
        reconstruction_distribution="negative binomial",
        number_of_reconstruction_classes=2,
        log_directory=str(tmp_path), device=cuda_device)
    assert "-k_2-" in model.name
    assert "X_TILDE/P_K/DENSE/weights:0" in model.parameters
    model.train(data, None, number_of_epochs=2, minibatch_size=32)
    _, reconstructed, _ = model.evaluate(data)
    assert np.isfinite(reconstructed.values).all()
    with pytest.raises(ValueError):
        VariationalAutoencoder(
            feature_size=F, latent_size=3, hidden_sizes=[8],
            reconstruction_distribution="zero-inflated poisson",
            number_of_reconstruction_classes=2,
            log_directory=str(tmp_path), device=cuda_device)
