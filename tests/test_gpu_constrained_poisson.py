"""GPU parity: the constrained Poisson likelihood (du:218-228 -- softmax over
the genes times the count sum of the cell, va:2400-2405, 2490-2496) through
the C ABI vs the fp64 oracle, VAE and GMVAE, training and evaluation."""
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


def _counts(rng, cells, features):
    lam = rng.gamma(0.5, 3.0, size=(1, features))
    x = rng.poisson(lam, size=(cells, features)).astype(np.float64)
    x *= rng.random((cells, features)) > 0.5
    x[:, 0] += 1   # no empty cell
    return x


def _skip_bias(name):
    return name.endswith("DENSE/biases") and "LAYER_" in name or (
        name.endswith("DENSE/biases") and (
            "ENCODER" in name or "DECODER" in name))


@pytest.mark.parametrize("n_iw,n_mc,F,H,B", [
    (1, 1, 130, (18, 14), 21), (2, 2, 77, (18, 14), 21),
    (1, 1, 1100, (18, 14), 21),
    # several row tiles and strips of the three-pass head kernel, decoder
    # widths 100 and 40
    (1, 1, 700, (100, 40), 300), (2, 1, 333, (40, 24), 150)])
def test_vae_step_matches_oracle(cuda_device, n_iw, n_mc, F, H, B):
    from scvae_amd.engine import Engine
    L = 5
    S = n_iw * n_mc
    eng = Engine(F, L, H, "constrained poisson", device=cuda_device)
    assert "X_TILDE/LAMBDA/DENSE/weights" in eng.named_parameters()
    g = torch.Generator().manual_seed(1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="constrained poisson", n_iw=n_iw,
                         n_mc=n_mc)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    assert list(params) == list(om.vae_parameter_shapes(cfg))
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(0)
    x = torch.from_numpy(_counts(rng, B, F))
    count_sum = x.sum(dim=1)
    eps = torch.from_numpy(rng.standard_normal((S, B, L)))
    xd = x.float().to(cuda_device)
    csd = count_sum.float().to(cuda_device)
    ll = torch.zeros(S * B, device=cuda_device)
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=True,
                  n_iw=n_iw, n_mc=n_mc, count_sum=csd,
                  outputs={"log_p_x_given_z": ll}).cpu().numpy()
    torch.cuda.synchronize()
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, x, x, eps, True, 1.0, {},
                                 count_sum=count_sum), params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    _close(sc[2], out["reconstruction_error"], what="reconstruction_error")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    for name, g in eng.named_gradients().items():
        if _skip_bias(name):
            continue
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)

    # evaluation, likelihood only (the forward passes of the fused kernel)
    moving_now = {k: v.detach().cpu().double()
                  for k, v in eng.named_moving_statistics().items()}
    ll_e = torch.zeros(S * B, device=cuda_device)
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=False,
                  n_iw=n_iw, n_mc=n_mc, count_sum=csd,
                  outputs={"log_p_x_given_z": ll_e}).cpu().numpy()
    out_e = om.vae_forward(cfg, params, moving_now, x, x, eps, False,
                           count_sum=count_sum)
    _close(sc[0], out_e["lower_bound"], what="lower_bound (evaluation)")
    close_elementwise(ll_e, out_e["log_p_x_given_z"].reshape(-1),
                      rtol=LL_RTOL, atol=LL_ATOL,
                      what="per-cell ll (evaluation)")

    # evaluation with the reconstruction statistics (mean = variance = rate)
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    outs = {k: torch.zeros(B, F, device=cuda_device) for k in (
        "p_x_mean", "p_x_stddev", "stddev_of_p_x_given_z_mean")}
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=False,
                  n_iw=n_iw, n_mc=n_mc, count_sum=csd,
                  outputs=outs).cpu().numpy()
    out = om.vae_forward(cfg, params, moving, x, x, eps, False,
                         evaluation_statistics=True, count_sum=count_sum)
    _close(sc[0], out["lower_bound"], what="lower_bound (evaluation)")
    for k in outs:
        _close(outs[k].cpu(), out[k], rtol=2e-4, what=k)
    # every reconstructed cell carries its count sum
    _close(outs["p_x_mean"].sum(dim=1).cpu(), count_sum, rtol=1e-4,
           what="sum of rates")
    with pytest.raises(ValueError):
        eng.step(xd, xd, eps=eps.float().to(cuda_device), training=False,
                 n_iw=n_iw, n_mc=n_mc)


def test_uint16_minibatch_is_the_fp32_step_bit_for_bit(cuda_device):
    """The constrained Poisson step on the uint16 minibatch (the three passes of
    the head kernel read their targets as uint16, the input layer runs on the
    count kernels): identical bits to the fp32 batch."""
    import scipy.sparse as sp
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import DeviceCSR
    F, L, H, B = 1300, 6, (100, 48), 200
    rng = np.random.default_rng(29)
    dense = _counts(rng, 400, F)
    csr = DeviceCSR.from_scipy(sp.csr_matrix(dense), cuda_device)
    rows = torch.from_numpy(rng.permutation(400)[:B]).to(cuda_device)
    x32 = csr.gather_dense(rows)
    rc16 = torch.zeros(B, device=cuda_device)
    x16 = csr.gather_counts_u16(rows, row_const_out=rc16)
    cs = x32.sum(dim=1)
    eps = torch.from_numpy(rng.standard_normal((1, B, L)).astype(np.float32)
                           ).to(cuda_device)
    results = []
    for u16 in (False, True):
        eng = Engine(F, L, H, "constrained poisson", batch_norm=True,
                     device=cuda_device, seed=1)
        eng.set_count_gemm(True, always=True)
        assert eng.accepts_counts_u16(B, True) and eng.accepts_counts_u16(B, False)
        x = x16 if u16 else x32
        out = []
        for _ in range(2):
            ll = torch.zeros(B, device=cuda_device)
            s = eng.step(x, x, eps=eps, training=True, row_const=rc16,
                         x_counts=True, count_sum=cs,
                         outputs={"log_p_x_given_z": ll}).clone()
            eng.adam_step(1e-3)
            out += [s, ll.clone()]
        ev = eng.step(x, x, eps=eps, training=False, row_const=rc16,
                      x_counts=True, count_sum=cs).clone()
        torch.cuda.synchronize()
        results.append([t.cpu() for t in out + [ev, eng.grads, eng.moving,
                                                eng.params]])
    for a, b in zip(*results):
        assert torch.equal(a, b)


def test_gmvae_step_matches_oracle(cuda_device):
    from scvae_amd.engine import Engine
    F, L, H, B, K = 90, 4, (16,), 19, 3
    eng = Engine(F, L, H, "constrained poisson", model_type="GMVAE",
                 n_clusters=K, device=cuda_device)
    g = torch.Generator().manual_seed(1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="constrained poisson", n_clusters=K)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(0)
    x = torch.from_numpy(_counts(rng, B, F))
    count_sum = x.sum(dim=1)
    eps = torch.from_numpy(rng.standard_normal((K, 1, B, L)))
    xd = x.float().to(cuda_device)
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=True,
                  count_sum=count_sum.float().to(cuda_device)).cpu().numpy()
    torch.cuda.synchronize()
    out, grads = om.gradients(
        lambda p: om.gmvae_forward(cfg, p, moving, x, x, eps, True, 1.0, {},
                                   count_sum=count_sum), params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    for name, g in eng.named_gradients().items():
        if _skip_bias(name):
            continue
        got, want = g.cpu(), grads[name]
        if name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            got, want = got[:F], want[:F]
        _close(got, want, rtol=3e-4, what="grad " + name)


def test_model_class_trains_and_evaluates(cuda_device, tmp_path):
    from scvae_amd.data import DataSet
    from scvae_amd.models import VariationalAutoencoder
    from scvae_amd.models.utilities import load_learning_curves
    rng = np.random.default_rng(5)
    x = rng.poisson(2.0, size=(120, 40)).astype(np.float32)
    x[:, 0] += 1
    data = DataSet("toy", values=x, example_names=np.arange(120).astype(str),
                   feature_names=np.arange(40).astype(str), kind="training")
    model = VariationalAutoencoder(
        feature_size=40, latent_size=3, hidden_sizes=[12],
        reconstruction_distribution="constrained poisson",
        log_directory=str(tmp_path))
    assert model.use_count_sum_as_parameter
    assert model.train(data, data, number_of_epochs=3, minibatch_size=30,
                       learning_rate=1e-2) == 0
    lb = load_learning_curves(model)["validation"]["lower_bound"]
    assert np.isfinite(lb).all() and lb[-1] > lb[0]
    reconstructed = model.evaluate(data, minibatch_size=30,
                                   output_versions="reconstructed")
    values = np.asarray(reconstructed.values)
    assert np.allclose(values.sum(axis=1), x.sum(axis=1), rtol=1e-4)
    with pytest.raises(NotImplementedError):
        model.sample(sample_size=5)


# ------------------------------ Bernoulli -----------------------------------

@pytest.mark.parametrize("model_type", ["VAE", "GMVAE"])
def test_bernoulli_step_matches_oracle(cuda_device, model_type):
    """du:194-204: Bernoulli(logits) on binarised targets, the raw counts as
    encoder input (va:845-857)."""
    from scvae_amd.engine import Engine
    F, L, H, B, K = 110, 4, (16, 12), 23, 3
    gm = model_type == "GMVAE"
    eng = Engine(F, L, H, "bernoulli", model_type=model_type,
                 n_clusters=K if gm else 1, device=cuda_device)
    assert any(n.endswith("LOGITS/DENSE/weights") and "X" in n
               for n in eng.named_parameters())
    g = torch.Generator().manual_seed(1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="bernoulli", n_clusters=K if gm else 1)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(0)
    x = torch.from_numpy(_counts(rng, B, F))
    t = (x > 0.5).double()
    eps = torch.from_numpy(rng.standard_normal(
        (K, 1, B, L) if gm else (1, B, L)))
    xd, td = x.float().to(cuda_device), t.float().to(cuda_device)
    outs = {k: torch.zeros(B, F, device=cuda_device) for k in (
        "p_x_mean", "p_x_stddev", "stddev_of_p_x_given_z_mean")}
    sc = eng.step(xd, td, eps=eps.float().to(cuda_device),
                  training=True).cpu().numpy()
    torch.cuda.synchronize()
    forward = om.gmvae_forward if gm else om.vae_forward
    out, grads = om.gradients(
        lambda p: forward(cfg, p, moving, x, t, eps, True, 1.0, {}), params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    for name, gr in eng.named_gradients().items():
        if _skip_bias(name):
            continue
        got, want = gr.cpu(), grads[name]
        if name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            got, want = got[:F], want[:F]
        _close(got, want, rtol=3e-4, what="grad " + name)
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    sc = eng.step(xd, td, eps=eps.float().to(cuda_device), training=False,
                  outputs=outs).cpu().numpy()
    out = forward(cfg, params, moving, x, t, eps, False,
                  evaluation_statistics=True)
    _close(sc[0], out["lower_bound"], what="lower_bound (evaluation)")
    _close(outs["p_x_mean"].cpu(), out["p_x_mean"], rtol=2e-4, what="p_x_mean")
    _close(outs["p_x_stddev"].cpu(), out["p_x_stddev"], rtol=2e-4,
           what="p_x_stddev")
    assert float(outs["p_x_mean"].min()) >= 0 and float(
        outs["p_x_mean"].max()) <= 1


def test_bernoulli_model_class(cuda_device, tmp_path):
    from scvae_amd.data import DataSet
    from scvae_amd.distributions import DISTRIBUTIONS
    from scvae_amd.models import VariationalAutoencoder
    from scvae_amd.models.utilities import load_learning_curves
    rng = np.random.default_rng(5)
    x = rng.poisson(0.8, size=(120, 40)).astype(np.float32)
    data = DataSet("toy", values=x, example_names=np.arange(120).astype(str),
                   feature_names=np.arange(40).astype(str), kind="training")
    model = VariationalAutoencoder(
        feature_size=40, latent_size=3, hidden_sizes=[12],
        reconstruction_distribution="bernoulli", log_directory=str(tmp_path))
    assert model.train(data, data, number_of_epochs=3, minibatch_size=30,
                       learning_rate=1e-2) == 0
    lb = load_learning_curves(model)["validation"]["lower_bound"]
    assert np.isfinite(lb).all() and lb[-1] > lb[0]
    # a Bernoulli log-likelihood is never positive
    assert max(load_learning_curves(model)["validation"][
        "reconstruction_error"]) <= 0
    logits = torch.tensor([[-2.0, 0.0, 3.0]], device=cuda_device)
    d = DISTRIBUTIONS["bernoulli"]["class"]({"logits": logits})
    want = torch.distributions.Bernoulli(logits=logits.cpu())
    t = torch.tensor([[0.0, 1.0, 1.0]])
    assert torch.allclose(d.log_prob(t.to(cuda_device)).cpu(),
                          want.log_prob(t), atol=1e-6)
    assert torch.allclose(d.mean().cpu(), want.mean, atol=1e-6)
    assert torch.allclose(d.variance().cpu(), want.variance, atol=1e-6)
