"""Randomised (fixed seeds) cross-check of the two kernel paths through the step: the fused
decoder-head kernels against the unfused GEMM + likelihood kernels for random model shapes,
likelihoods, sample counts and ragged sizes; both models."""
import numpy as np
import pytest
import torch

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
