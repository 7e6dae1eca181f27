"""GPU parity: one VAE graph execution through the C ABI vs the fp64 oracle.

Tolerance: BASELINE.json asks for ELBO and per-cell reconstruction
log-likelihood within 1e-4 relative in fp32; gradients / post-Adam weights are
held to 2e-4 of the tensor's largest magnitude.
"""
import numpy as np
import pytest
import torch

from oracle import models as om

from _parity import LL_ATOL, LL_RTOL, close_elementwise
pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _counts(rng, cells, features, scale=3.0, sparsity=0.7):
    lam = rng.gamma(0.5, scale, size=(1, features))
    x = rng.poisson(lam, size=(cells, features)).astype(np.float64)
    x *= rng.random((cells, features)) > sparsity
    x[0, 0] = 1000.0  # one large count
    return x


def _setup(cuda_device, likelihood, F, L, H, B, bn, seed=0, n_iw=1, n_mc=1):
    from scvae_amd.engine import Engine
    eng = Engine(F, L, H, likelihood, batch_norm=bn, device=cuda_device,
                 seed=seed)
    # perturb biases / beta so they are not all zero
    g = torch.Generator().manual_seed(seed + 1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=tuple(H),
                         likelihood=likelihood, minibatch_normalisation=bn,
                         n_iw=n_iw, n_mc=n_mc)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    assert list(params) == list(om.vae_parameter_shapes(cfg))
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(_counts(rng, B, F))
    eps = torch.from_numpy(rng.standard_normal((n_iw * n_mc, B, L)))
    return eng, cfg, params, moving, x, eps


def _close(a, b, rtol=RTOL, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "{}: max err {:.3e} of scale {:.3e}".format(
        what, err, scale)


@pytest.mark.parametrize("likelihood", [
    "poisson", "negative binomial", "zero-inflated poisson",
    "zero-inflated negative binomial"])
@pytest.mark.parametrize("bn", [True, False])
def test_train_step_matches_oracle(cuda_device, likelihood, bn):
    F, L, H, B = 203, 7, (24, 20), 37
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, likelihood, F, L, H, B, bn)
    xd = x.float().to(cuda_device)
    epsd = eps.float().to(cuda_device)
    ll = torch.zeros(B, device=cuda_device)
    klz = torch.zeros(L, device=cuda_device)
    qz = torch.zeros(B, L, device=cuda_device)
    sc = eng.step(xd, xd, eps=epsd, training=True, warm_up_weight=0.7,
                  outputs={"log_p_x_given_z": ll, "kl_neurons": klz,
                           "q_z_mean": qz}).cpu().numpy()
    eng.adam_step(1e-3)
    torch.cuda.synchronize()

    state = om.adam_state(params)
    new_params, new_moving, out, grads = om.vae_train_step(
        cfg, dict(params), moving, state, x, x, eps, 1e-3, warm_up_weight=0.7)

    _close(sc[0], out["lower_bound"], what="lower_bound")
    _close(sc[1], out["lower_bound_weighted"], what="lower_bound_weighted")
    _close(sc[2], out["reconstruction_error"], what="reconstruction_error")
    _close(sc[3], out["kl_divergence"], what="kl_divergence")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    _close(klz.cpu(), out["kl_divergence_neurons"], what="kl neurons")
    _close(qz.cpu(), out["q_z_mean"], what="q_z_mean")
    for name, g in eng.named_gradients().items():
        if bn and name.endswith("DENSE/biases") and "X_TILDE" not in name \
                and "POSTERIOR" not in name:
            # bias under batch norm: mathematically zero gradient
            assert g.abs().max().item() < 1e-5
            continue
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)
    for name, p in eng.named_parameters().items():
        if bn and name.endswith("DENSE/biases") and "X_TILDE" not in name \
                and "POSTERIOR" not in name:
            continue
        _close(p.cpu(), new_params[name], rtol=2e-4, what="param " + name)
    for name, m in eng.named_moving_statistics().items():
        _close(m.cpu(), new_moving[name], rtol=1e-5, what="moving " + name)


@pytest.mark.parametrize("likelihood,H,B", [
    ("negative binomial", (128, 128), 300),
    ("negative binomial", (40, 200), 150),
    ("poisson", (64, 256), 100),
    ("zero-inflated negative binomial", (32, 129), 130),
])
def test_wide_decoder_trains_on_the_fused_kernel(cuda_device, likelihood, H, B):
    """``-H`` beyond 126 (mu:81-126 takes any size): a training step of one
    likelihood pass runs the heads on the bf16x9 producer / consumer kernel
    (``scvae_decoder_train_kernel`` == 3), evaluation steps of the same model on
    its forward half (``FWD``) -- both against the oracle."""
    from scvae_amd import _lib
    lib = _lib.load()
    kind, _ = _lib.LIKELIHOOD_KINDS[likelihood]
    assert lib.scvae_decoder_train_kernel(kind, H[-1], 1) == 3
    F, L = 333, 9
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, likelihood, F, L, H, B, True)
    xd = x.float().to(cuda_device)
    epsd = eps.float().to(cuda_device)
    ll = torch.zeros(B, device=cuda_device)
    sc = eng.step(xd, xd, eps=epsd, training=True,
                  outputs={"log_p_x_given_z": ll}).cpu().numpy()
    torch.cuda.synchronize()
    new_moving = {}
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, x, x, eps, True, 1.0, new_moving),
        params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    for name, g in eng.named_gradients().items():
        if name.endswith("DENSE/biases") and "X_TILDE" not in name \
                and "POSTERIOR" not in name:
            continue
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)
    # evaluation (forward only): the forward half of the same kernel at this width
    moving_now = {k: v.detach().cpu().double()
                  for k, v in eng.named_moving_statistics().items()}
    sc = eng.step(xd, xd, eps=epsd, training=False,
                  outputs={"log_p_x_given_z": ll}).cpu().numpy()
    out_e = om.vae_forward(cfg, params, moving_now, x, x, eps, False)
    _close(sc[0], out_e["lower_bound"], what="lower_bound (evaluation)")
    close_elementwise(ll, out_e["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll (evaluation)")


def test_importance_weighted_step(cuda_device):
    F, L, H, B = 150, 5, (16,), 19
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, "negative binomial", F, L, H, B, True, n_iw=3, n_mc=2)
    xd = x.float().to(cuda_device)
    epsd = eps.float().to(cuda_device)
    sc = eng.step(xd, xd, eps=epsd, training=True, n_iw=3,
                  n_mc=2).cpu().numpy()
    torch.cuda.synchronize()
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, x, x, eps, True, 1.0, {}),
        params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    _close(sc[2], out["reconstruction_error"], what="reconstruction_error")
    for name, g in eng.named_gradients().items():
        if name.endswith("DENSE/biases") and ("ENCODER" in name
                                              or "DECODER" in name):
            continue
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)


def test_evaluation_mode_statistics(cuda_device):
    F, L, H, B = 180, 6, (20, 20), 23
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, "zero-inflated negative binomial", F, L, H, B, True,
        n_iw=2, n_mc=2)
    # non-trivial moving statistics
    g = torch.Generator().manual_seed(5)
    for name, m in eng.named_moving_statistics().items():
        if name.endswith("moving_mean"):
            m.copy_(torch.randn(m.shape, generator=g) * 0.2)
        else:
            m.copy_(torch.rand(m.shape, generator=g) + 0.5)
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    xd = x.float().to(cuda_device)
    epsd = eps.float().to(cuda_device)
    outs = {k: torch.zeros(B, F, device=cuda_device) for k in (
        "p_x_mean", "p_x_stddev", "stddev_of_p_x_given_z_mean")}
    ll = torch.zeros(4 * B, device=cuda_device)
    outs["log_p_x_given_z"] = ll
    sc = eng.step(xd, xd, eps=epsd, training=False, n_iw=2, n_mc=2,
                  outputs=outs).cpu().numpy()
    out = om.vae_forward(cfg, params, moving, x, x, eps, False,
                         evaluation_statistics=True)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell ll")
    for k in ("p_x_mean", "p_x_stddev", "stddev_of_p_x_given_z_mean"):
        _close(outs[k].cpu(), out[k], rtol=2e-4, what=k)
    # deterministic z
    sc = eng.step(xd, xd, training=False, deterministic_z=True).cpu().numpy()
    out = om.vae_forward(cfg, params, moving, x, x, None, False,
                         deterministic_z=True)
    _close(sc[0], out["lower_bound"], what="lower_bound (deterministic z)")


@pytest.mark.parametrize("likelihood", ["negative binomial", "poisson"])
def test_model_without_hidden_layers(cuda_device, likelihood):
    """``hidden_sizes=[]``: posterior heads on x, likelihood heads directly on z
    (the reference's dense_layers() with an empty list)."""
    F, L, B = 40, 6, 17
    eng, cfg, params, moving, x, eps = _setup(
        cuda_device, likelihood, F, L, (), B, True)
    sc = eng.step(x.float().to(cuda_device), x.float().to(cuda_device),
                  eps=eps.float().to(cuda_device), training=True).cpu().numpy()
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, x, x, eps, True), params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    for name, g in eng.named_gradients().items():
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)


@pytest.mark.parametrize("inference,generative", [("LFM", "MLP"), ("MLP", "LFM"),
                                                  ("LFM", "LFM")])
def test_linear_factor_architectures(cuda_device, inference, generative):
    """``inference_architecture`` / ``generative_architecture`` = "LFM": the
    hidden layers of that side are not built (va:2233-2234, 2456-2457)."""
    from scvae_amd.engine import Engine
    F, L, H, B = 50, 6, (14, 12), 19
    eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                 device=cuda_device, seed=0, inference_architecture=inference,
                 generative_architecture=generative)
    g = torch.Generator().manual_seed(1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="negative binomial",
                         inference_architecture=inference,
                         generative_architecture=generative)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    assert list(params) == list(om.vae_parameter_shapes(cfg))
    assert any("ENCODER" in k for k in params) == (inference == "MLP")
    assert any("DECODER" in k for k in params) == (generative == "MLP")
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(0)
    x = torch.from_numpy(_counts(rng, B, F))
    eps = torch.from_numpy(rng.standard_normal((1, B, L)))
    sc = eng.step(x.float().to(cuda_device), x.float().to(cuda_device),
                  eps=eps.float().to(cuda_device), training=True).cpu().numpy()
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, x, x, eps, True), params)
    _close(sc[0], out["lower_bound"], what="lower_bound")
    for name, g in eng.named_gradients().items():
        if name.endswith("DENSE/biases") and ("ENCODER/" in name
                                              or "DECODER/" in name):
            continue
        _close(g.cpu(), grads[name], rtol=2e-4, what="grad " + name)
    z = torch.from_numpy(rng.standard_normal((5, L)))
    moving = {k: v.detach().cpu().double()    # updated by the training step
              for k, v in eng.named_moving_statistics().items()}
    _close(eng.decode(z.float().to(cuda_device)).cpu(),
           om.decode_mean(cfg, params, moving, z), rtol=2e-4, what="decode")


@pytest.mark.parametrize("B,H", [(37, (24, 20)), (4096, (100, 100)),
                                 (5000, (64, 32)), (8192, (100,))])
def test_one_launch_batch_norm_matches_the_chunked_kernels(cuda_device, B, H):
    """Single-group layers normalise in one column-parallel launch
    (``bn_fwd_cols`` / ``bn_bwd_cols``); the chunked statistics / finalize /
    apply kernels (other shapes, the GMVAE's grouped layers, data parallel)
    give the same step: scalars, gradients, moving statistics."""
    from scvae_amd.engine import Engine
    F, L = 300, 8
    rng = np.random.default_rng(B)
    x = torch.from_numpy(_counts(rng, B, F)).float().to(cuda_device)
    eps = torch.from_numpy(rng.standard_normal((1, B, L))).float().to(
        cuda_device)
    results = []
    for one_launch in (True, False):
        eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                     device=cuda_device, seed=4)
        eng.set_bn_one_launch(one_launch, always=True)
        scalars = eng.step(x, x, eps=eps, training=True).clone()
        torch.cuda.synchronize()
        results.append((scalars.cpu(), eng.grads.clone().cpu(),
                        eng.moving.clone().cpu()))
    for a, b in zip(*results):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-9


@pytest.mark.parametrize("B,H,L,n_iw", [(100, (100, 100), 25, 1),
                                         (37, (24, 20), 7, 1),
                                         (128, (128,), 128, 1),
                                         (19, (16, 12, 8), 5, 3),
                                         (64, (50, 30), 10, 2)])
def test_mid_chain_matches_the_launch_chain(cuda_device, B, H, L, n_iw):
    """Small minibatches run the hidden layers, posterior heads and latent stage
    in two cooperative launches (``midchain.hip``: sixteen workgroups, a grid
    barrier per layer; instead of ~27 launches).  Same step as the chain of
    launches: scalars, per-cell outputs, every gradient, the moving statistics;
    training and evaluation.  (Widths off the float4 path, ragged strips, three
    layers, importance samples.)"""
    from scvae_amd.engine import Engine
    F = 400
    rng = np.random.default_rng(B + L)
    x = torch.from_numpy(_counts(rng, B, F)).float().to(cuda_device)
    eps = torch.from_numpy(rng.standard_normal((n_iw, B, L))).float().to(
        cuda_device)
    results = []
    for mid in (True, False):
        eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                     device=cuda_device, seed=4)
        g = torch.Generator().manual_seed(9)
        for name, p in eng.named_parameters().items():
            if not name.endswith("weights"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        eng.set_mid_chain(mid)
        ll = torch.zeros(n_iw * B, device=cuda_device)
        qz = torch.zeros(B, L, device=cuda_device)
        klz = torch.zeros(L, device=cuda_device)
        outs = {"log_p_x_given_z": ll, "q_z_mean": qz, "kl_neurons": klz}
        scalars = eng.step(x, x, eps=eps, training=True, n_iw=n_iw,
                           warm_up_weight=0.7, outputs=outs).clone()
        torch.cuda.synchronize()
        train = [scalars.cpu(), ll.cpu().clone(), qz.cpu().clone(),
                 klz.cpu().clone(), eng.grads.clone().cpu(),
                 eng.moving.clone().cpu()]
        ev = eng.step(x, x, eps=eps, training=False, n_iw=n_iw,
                      outputs=outs).clone()
        det = eng.step(x, x, training=False, deterministic_z=True).clone()
        torch.cuda.synchronize()
        results.append(train + [ev.cpu(), ll.cpu().clone(), det.cpu()])
        if mid:
            # many launches on the same barrier counter, and bitwise repeatable
            moving = eng.moving.clone()
            again = []
            for _ in range(2):
                eng.moving.copy_(moving)
                for _ in range(25):
                    s25 = eng.step(x, x, eps=eps, training=True, n_iw=n_iw,
                                   warm_up_weight=0.7).clone()
                torch.cuda.synchronize()
                again.append((s25.cpu(), eng.grads.clone().cpu(),
                              eng.moving.clone().cpu()))
            for u, v in zip(*again):
                assert torch.equal(u, v)
    for a, b in zip(*results):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-9


@pytest.mark.parametrize("B,H,L,n_iw,n_mc", [(4096, (100, 100), 25, 1, 1),
                                              (300, (100, 100), 25, 1, 1),
                                              (129, (24, 20), 7, 1, 1),
                                              (1000, (128,), 128, 1, 1),
                                              (333, (16, 12, 8), 5, 1, 1),
                                              (200, (50, 30), 10, 2, 1),
                                              (150, (64, 64), 9, 1, 3)])
def test_resident_tile_chain_is_the_tile_chain_bit_for_bit(cuda_device, B, H, L,
                                                           n_iw, n_mc):
    """The stages of a pass in ONE resident launch per direction (grid barriers
    between the layers, ``scvae_plan_set_tile_resident``) against one launch per
    layer: the same tile code on the same tiles, so every output, gradient and
    moving statistic carries identical bits -- over two steps with the
    optimiser in between (the barrier counter carries over)."""
    from scvae_amd.engine import Engine
    F = 400
    S = n_iw * n_mc
    rng = np.random.default_rng(B + L)
    x = torch.from_numpy(_counts(rng, B, F)).float().to(cuda_device)
    eps = torch.from_numpy(rng.standard_normal((S, B, L))).float().to(
        cuda_device)
    results = []
    for resident in (True, False):
        eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                     device=cuda_device, seed=4)
        eng.set_dd_atomics(False)
        g = torch.Generator().manual_seed(9)
        for name, p in eng.named_parameters().items():
            if not name.endswith("weights"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, m in eng.named_moving_statistics().items():
            m.copy_(torch.rand(m.shape, generator=g) + 0.5)
        eng.set_tile_resident(resident)
        eng.reserve(B, S)
        assert eng.uses_tile_chain(B, S)
        assert eng.uses_tile_resident(B, S) == resident
        out = []
        for step in range(2):
            ll = torch.zeros(S * B, device=cuda_device)
            qz = torch.zeros(B, L, device=cuda_device)
            klz = torch.zeros(L, device=cuda_device)
            scalars = eng.step(
                x, x, eps=eps, training=True, n_iw=n_iw, n_mc=n_mc,
                warm_up_weight=0.7, outputs={
                    "log_p_x_given_z": ll, "q_z_mean": qz,
                    "kl_neurons": klz}).clone()
            torch.cuda.synchronize()
            out += [scalars.cpu(), ll.cpu(), qz.cpu(), klz.cpu(),
                    eng.grads.clone().cpu(), eng.moving.clone().cpu()]
            eng.adam_step(1e-3)
        out.append(eng.params.clone().cpu())
        results.append(out)
    for i, (a, b) in enumerate(zip(*results)):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("B,H,L,n_iw,n_mc", [(4096, (100, 100), 25, 1, 1),
                                              (300, (100, 100), 25, 1, 1),
                                              (129, (24, 20), 7, 1, 1),
                                              (1000, (128,), 128, 1, 1),
                                              (333, (16, 12, 8), 5, 1, 1),
                                              (200, (50, 30), 10, 2, 1),
                                              (150, (64, 64), 9, 1, 3)])
def test_tile_chain_matches_the_launch_chain(cuda_device, B, H, L, n_iw, n_mc):
    """Large training minibatches run every hidden layer and the posterior heads
    as one launch per layer and direction (``tilechain.hip``: 64-row tiles, the
    consumer of a layer merges its batch-norm chunk statistics; instead of ~10
    launches per layer).  Same step as the chain of launches: scalars, per-cell
    outputs, every gradient, the moving statistics -- and bitwise repeatable.
    (Ragged last tiles, odd widths, one / three layers, importance and
    Monte-Carlo samples.)"""
    from scvae_amd.engine import Engine
    F = 400
    S = n_iw * n_mc
    rng = np.random.default_rng(B + L)
    x = torch.from_numpy(_counts(rng, B, F)).float().to(cuda_device)
    eps = torch.from_numpy(rng.standard_normal((S, B, L))).float().to(
        cuda_device)
    results = []
    for tile in (True, False):
        eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                     device=cuda_device, seed=4)
        g = torch.Generator().manual_seed(9)
        for name, p in eng.named_parameters().items():
            if not name.endswith("weights"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, m in eng.named_moving_statistics().items():
            m.copy_(torch.rand(m.shape, generator=g) + 0.5)
        eng.set_tile_chain(tile)
        ll = torch.zeros(S * B, device=cuda_device)
        qz = torch.zeros(B, L, device=cuda_device)
        klz = torch.zeros(L, device=cuda_device)
        outs = {"log_p_x_given_z": ll, "q_z_mean": qz, "kl_neurons": klz}
        moving0 = eng.moving.clone()
        scalars = eng.step(x, x, eps=eps, training=True, n_iw=n_iw, n_mc=n_mc,
                           warm_up_weight=0.7, outputs=outs).clone()
        torch.cuda.synchronize()
        train = [scalars.cpu(), ll.cpu().clone(), qz.cpu().clone(),
                 klz.cpu().clone(), eng.grads.clone().cpu(),
                 eng.moving.clone().cpu()]
        results.append(train)
        if tile:
            eng.moving.copy_(moving0)
            again = eng.step(x, x, eps=eps, training=True, n_iw=n_iw,
                             n_mc=n_mc, warm_up_weight=0.7, outputs=outs).clone()
            torch.cuda.synchronize()
            assert torch.equal(again.cpu(), train[0])
            assert torch.equal(eng.grads.cpu(), train[4])
            assert torch.equal(eng.moving.cpu(), train[5])
    names = ["scalars", "ll", "q_z_mean", "kl_neurons", "grads", "moving"]
    for name, a, b in zip(names, *results):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-9, (
            name, (a - b).abs().max().item(), scale)
