"""Randomised (fixed seeds) cross-check of the two kernel paths through the step: the fused
decoder-head kernels against the unfused GEMM + likelihood kernels for random model shapes,
likelihoods, sample counts and ragged sizes; both models."""
import numpy as np
import pytest
import torch

from _parity import LL_ATOL, LL_RTOL, close_elementwise
pytestmark = pytest.mark.gpu

LIKELIHOODS = ["poisson", "negative binomial", "zero-inflated poisson",
               "zero-inflated negative binomial"]


def _case(seed):
    rng = np.random.default_rng(seed)
    gm = bool(rng.integers(0, 2))
    return dict(
        gm=gm,
        F=int(rng.integers(3, 700)),
        L=int(rng.integers(1, 12)),
        H=tuple(int(2 * rng.integers(1, 60)) for _ in range(rng.integers(1, 3))),
        B=int(rng.integers(1, 150)),
        K=int(rng.integers(1, 5)),
        n_iw=int(rng.integers(1, 3)),
        n_mc=int(rng.integers(1, 3)),
        likelihood=LIKELIHOODS[int(rng.integers(0, 4))],
        bn=bool(rng.integers(0, 2)),
        density=float(rng.choice([0.02, 0.3, 1.0])),
        extra=int(rng.choice([0, 0, 3])),
    )


@pytest.mark.parametrize("seed", range(24))
def test_fused_equals_unfused_on_random_models(cuda_device, seed):
    from scvae_amd.engine import Engine
    c = _case(seed)
    if c["B"] == 1 and c["bn"]:
        c["B"] = 2   # batch norm over one row is degenerate in the reference as well
    rng = np.random.default_rng(1000 + seed)
    eng = Engine(c["F"], c["L"], c["H"], c["likelihood"], batch_norm=c["bn"],
                 model_type="GMVAE" if c["gm"] else "VAE", n_clusters=c["K"],
                 device=cuda_device, seed=seed, decoder_extra=c["extra"])
    g = torch.Generator().manual_seed(seed)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    B, F, L, S = c["B"], c["F"], c["L"], c["n_iw"] * c["n_mc"]
    x = (rng.poisson(3.0, (B, F)) * (rng.random((B, F)) < c["density"])
         ).astype(np.float32)
    x = torch.from_numpy(x).to(cuda_device)
    shape = (c["K"], S, B, L) if c["gm"] else (S, B, L)
    eps = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)
                           ).to(cuda_device)
    extra = None
    if c["extra"]:
        extra = torch.from_numpy(rng.random((B, c["extra"])).astype(
            np.float32)).to(cuda_device)
    rows = (c["K"] if c["gm"] else 1) * S * B
    results = []
    for fused in (True, False):
        eng.set_fused(fused)
        ll = torch.zeros(rows, device=cuda_device)
        scalars = eng.step(x, x, eps=eps, training=True, n_iw=c["n_iw"],
                           n_mc=c["n_mc"], decoder_extra=extra,
                           outputs={"log_p_x_given_z": ll}).clone()
        torch.cuda.synchronize()
        results.append((scalars.cpu().numpy(), ll.cpu().numpy(),
                        eng.grads.clone().cpu().numpy()))
    (s_f, ll_f, g_f), (s_u, ll_u, g_u) = results
    assert np.isfinite(s_f[:5]).all() and np.isfinite(g_f).all(), c
    assert abs(s_f[0] - s_u[0]) <= 2e-5 * abs(s_u[0]) + 1e-6, c
    assert np.abs(ll_f - ll_u).max() <= 2e-5 * np.abs(ll_u).max() + 1e-5, c
    for name, (offset, shape) in eng.param_table.items():
        n = int(np.prod(shape))
        a, b = g_f[offset:offset + n], g_u[offset:offset + n]
        # importance weights / cluster responsibilities amplify rounding of ll
        tolerance = 2e-3 if (c["gm"] or c["n_iw"] > 1) else 2e-4
        assert np.abs(a - b).max() <= tolerance * np.abs(b).max() + 1e-7, (
            name, c)


@pytest.mark.parametrize("H,B,likelihood", [
    (100, 1000, "negative binomial"),      # ragged last tile on the dealt schedule
    (110, 777, "zero-inflated poisson"),   # the widest remainder (rows 96 .. 110)
    (98, 513, "poisson"),                  # one head; remainder rows 96 .. 98
    (112, 600, "negative binomial"),       # just outside: the padded 32-wide tile
    (96, 600, "negative binomial"),        # no remainder at all
    (100, 511, "negative binomial"),       # below the row threshold: default schedule
])
def test_head_kernel_schedules_against_the_unfused_path(cuda_device, H, B,
                                                        likelihood):
    """The likelihood-head kernel runs the h remainder of its two backward
    products (rows 96 .. H of dW incl. the bias row, columns 96 .. H - 1 of dd)
    on 16-wide MFMA tiles with the jobs re-dealt over the waves when
    96 < H <= 111 and there are >= 512 rows; all schedules against the unfused
    GEMM + likelihood kernels, ragged row counts included."""
    from scvae_amd.engine import Engine
    F, L = 333, 6
    rng = np.random.default_rng(H * 1000 + B)
    eng = Engine(F, L, (H,), likelihood, batch_norm=True, device=cuda_device,
                 seed=2)
    g = torch.Generator().manual_seed(3)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    x = torch.from_numpy((rng.poisson(3.0, (B, F))
                          * (rng.random((B, F)) < 0.2)).astype(np.float32)
                         ).to(cuda_device)
    eps = torch.from_numpy(rng.standard_normal((1, B, L)).astype(np.float32)
                           ).to(cuda_device)
    results = []
    for fused in (True, False):
        eng.set_fused(fused)
        ll = torch.zeros(B, device=cuda_device)
        scalars = eng.step(x, x, eps=eps, training=True,
                           outputs={"log_p_x_given_z": ll}).clone()
        torch.cuda.synchronize()
        results.append((scalars.cpu().numpy(), ll.cpu().numpy(),
                        eng.grads.clone().cpu().numpy()))
    eng.set_fused(True)
    (s_f, ll_f, g_f), (s_u, ll_u, g_u) = results
    assert abs(s_f[0] - s_u[0]) <= 2e-5 * abs(s_u[0]) + 1e-6
    assert np.abs(ll_f - ll_u).max() <= 2e-5 * np.abs(ll_u).max() + 1e-5
    for name, (offset, shape) in eng.param_table.items():
        n = int(np.prod(shape))
        a, b = g_f[offset:offset + n], g_u[offset:offset + n]
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max() + 1e-7, name


def _oracle_case(seed):
    rng = np.random.default_rng(5000 + seed)
    likelihood = LIKELIHOODS[int(rng.integers(0, 4))]
    k_max = int(rng.choice([0, 0, 2, 4])) if "zero" not in likelihood else 0
    return dict(
        gm=bool(rng.integers(0, 2)), F=int(rng.integers(2, 150)),
        L=int(rng.integers(1, 8)),
        H=tuple(int(2 * rng.integers(1, 20)) for _ in range(rng.integers(1, 3))),
        B=int(rng.integers(2, 40)), K=int(rng.integers(1, 4)),
        S=int(rng.integers(1, 3)), likelihood=likelihood,
        bn=bool(rng.integers(0, 2)), extra=int(rng.choice([0, 0, 2])),
        k_max=k_max, free_nats=float(rng.choice([0.0, 0.5])),
        warm_up=float(rng.choice([1.0, 0.3])))


@pytest.mark.parametrize("seed", range(16))
def test_random_models_against_the_oracle(cuda_device, seed):
    from oracle import models as om
    from scvae_amd.engine import Engine
    c = _oracle_case(seed)
    rng = np.random.default_rng(7000 + seed)
    gm = c["gm"]
    eng = Engine(c["F"], c["L"], c["H"], c["likelihood"], batch_norm=c["bn"],
                 model_type="GMVAE" if gm else "VAE", n_clusters=c["K"],
                 device=cuda_device, seed=seed, decoder_extra=c["extra"],
                 k_max=c["k_max"], free_nats_proportion=c["free_nats"])
    g = torch.Generator().manual_seed(seed)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(
        feature_size=c["F"], latent_size=c["L"], hidden_sizes=c["H"],
        likelihood=c["likelihood"], minibatch_normalisation=c["bn"],
        n_iw=c["S"], n_mc=1, n_clusters=c["K"],
        free_nats_proportion=c["free_nats"], decoder_extra_size=c["extra"],
        k_max=c["k_max"])
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    B, F, L, S = c["B"], c["F"], c["L"], c["S"]
    x = torch.from_numpy((rng.poisson(2.5, (B, F))
                          * (rng.random((B, F)) < 0.5)).astype(np.float64))
    eps = torch.from_numpy(rng.standard_normal(
        (c["K"], S, B, L) if gm else (S, B, L)))
    extra = (torch.from_numpy(rng.random((B, c["extra"])))
             if c["extra"] else None)
    rows = (c["K"] if gm else 1) * S * B
    ll = torch.zeros(rows, device=cuda_device)
    sc = eng.step(
        x.float().to(cuda_device), x.float().to(cuda_device),
        eps=eps.float().to(cuda_device), training=True, n_iw=S, n_mc=1,
        warm_up_weight=c["warm_up"],
        decoder_extra=(extra.float().to(cuda_device)
                       if extra is not None else None),
        outputs={"log_p_x_given_z": ll}).cpu().numpy()
    torch.cuda.synchronize()
    forward = om.gmvae_forward if gm else om.vae_forward
    out, grads = om.gradients(
        lambda p: forward(cfg, p, moving, x, x, eps, True, c["warm_up"],
                          decoder_extra=extra), params)
    assert abs(sc[0] - float(out["lower_bound"])) <= 1e-4 * abs(
        float(out["lower_bound"])), c
    assert abs(sc[1] - float(out["lower_bound_weighted"])) <= 1e-4 * abs(
        float(out["lower_bound_weighted"])), c
    want = out["log_p_x_given_z"].reshape(-1).numpy()
    close_elementwise(ll, want, rtol=LL_RTOL, atol=LL_ATOL,
                      what="per-cell log-likelihood {}".format(c))
    for name, got in eng.named_gradients().items():
        if c["bn"] and name.endswith("DENSE/biases") and (
                "LAYER_" in name or "ENCODER/" in name or "DECODER/" in name):
            continue
        w = grads[name]
        got = got.cpu().double()
        if gm and c["bn"] and name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            got, w = got[:F], w[:F]
        scale = w.abs().max().item()
        assert (got - w).abs().max().item() <= 1e-3 * scale + 1e-7, (name, c)
