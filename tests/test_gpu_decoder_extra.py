"""Extra decoder inputs -- one-hot batch indices (batch correction) and the
normalised count sum appended to z (va:2407-2441, gm:3094-3130): step parity
with the oracle for both models, and the model classes end to end."""
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


@pytest.mark.parametrize("model_type,S", [("VAE", 1), ("VAE", 3),
                                          ("GMVAE", 1), ("GMVAE", 2)])
def test_step_with_decoder_extra_matches_oracle(cuda_device, model_type, S):
    from scvae_amd.engine import Engine
    F, L, H, B, K, E = 90, 4, (14, 12), 23, 3, 4
    gm = model_type == "GMVAE"
    eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                 model_type=model_type, n_clusters=K, device=cuda_device,
                 seed=2, decoder_extra=E)
    g = torch.Generator().manual_seed(7)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="negative binomial", n_clusters=K,
                         n_iw=S, n_mc=1, decoder_extra_size=E)
    shapes = (om.gmvae_parameter_shapes if gm else om.vae_parameter_shapes)(cfg)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    assert [(k, tuple(v.shape)) for k, v in params.items()] == [
        (k, tuple(v)) for k, v in shapes.items()]
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(3)
    x = torch.from_numpy(
        (rng.poisson(2.0, (B, F)) * (rng.random((B, F)) > 0.6))
        .astype(np.float64))
    batches = rng.integers(0, E - 1, B)
    extra = np.zeros((B, E))
    extra[np.arange(B), batches] = 1.0          # one-hot batch indices
    extra[:, E - 1] = rng.random(B)             # normalised count sum
    extra = torch.from_numpy(extra)
    eps = torch.from_numpy(rng.standard_normal(
        (K, S, B, L) if gm else (S, B, L)))

    xd = x.float().to(cuda_device)
    rows = (K if gm else 1) * S * B
    ll = torch.zeros(rows, device=cuda_device)
    sc = eng.step(xd, xd, eps=eps.float().to(cuda_device), training=True,
                  n_iw=S, n_mc=1, decoder_extra=extra.float().to(cuda_device),
                  outputs={"log_p_x_given_z": ll}).cpu().numpy()
    torch.cuda.synchronize()
    forward = om.gmvae_forward if gm else om.vae_forward
    out, grads = om.gradients(
        lambda p: forward(cfg, p, moving, x, x, eps, True,
                          decoder_extra=extra), params)
    _close(sc[0], out["lower_bound"], 1e-4, "lower_bound")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    first = "X/DECODER/LAYER_1" if gm else "DECODER/{}".format(len(H))
    assert eng.gradient(first + "/DENSE/weights").shape[0] == L + E
    for name, g in eng.named_gradients().items():
        if name.endswith("DENSE/biases") and (
                "LAYER_" in name or "ENCODER/" in name or "DECODER/" in name):
            continue   # bias under batch norm: mathematically zero gradient
        if gm and name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            g, want = g[:F], grads[name][:F]
        else:
            want = grads[name]
        _close(g.cpu(), want, 3e-4, "grad " + name)
    # evaluation mode runs too and needs the input
    eng.step(xd, xd, eps=eps.float().to(cuda_device), training=False,
             n_iw=S, n_mc=1, decoder_extra=extra.float().to(cuda_device))
    with pytest.raises(ValueError):
        eng.step(xd, xd, eps=eps.float().to(cuda_device), training=False,
                 n_iw=S, n_mc=1)


def test_batch_corrected_model_trains_and_evaluates(tmp_path, cuda_device,
                                                    capsys):
    from scvae_amd.data import DataSet
    from scvae_amd.models import VariationalAutoencoder
    rng = np.random.default_rng(1)
    n, F = 80, 30
    values = (rng.poisson(2.0, (n, F)) * (rng.random((n, F)) > 0.5)).astype(
        np.float32)
    data = DataSet("batched", values=values,
                   example_names=np.array(["c%d" % i for i in range(n)]),
                   feature_names=np.array(["g%d" % i for i in range(F)]),
                   batch_indices=rng.integers(0, 3, n))
    assert data.number_of_batches == 3
    model = VariationalAutoencoder(
        feature_size=F, latent_size=3, hidden_sizes=[8],
        reconstruction_distribution="negative binomial",
        batch_correction=True, number_of_batches=3, count_sum=True,
        log_directory=str(tmp_path), device=cuda_device)
    assert model.decoder_extra_size == 4
    assert "-sum-" in model.name and "-bc" in model.name
    model.train(data, None, number_of_epochs=2, minibatch_size=32)
    transformed, reconstructed, latent = model.evaluate(data)
    assert reconstructed.values.shape == (n, F)
    assert np.isfinite(reconstructed.values).all()
    with pytest.raises(NotImplementedError):
        model.sample(sample_size=5)
