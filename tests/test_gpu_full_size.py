"""Parity at the benchmark's full size (4096 cells x 32 738 genes, NB VAE 100-100-25), where the
fp64 oracle cannot run the whole step in seconds:

* the fused decoder-head kernel against the unfused path (GEMM + element-wise likelihood kernels),
  two independent implementations of the same maths, on ELBO, per-cell log-likelihood and every
  gradient;
* the oracle on a subset of the cells in evaluation mode (moving statistics make the rows
  independent), per-cell log-likelihood and KL;
* size-independent properties: lower_bound == reconstruction_error - kl_divergence, bitwise
  reproducibility of a repeated step, and row-permutation equivariance of the per-cell outputs.
"""
import numpy as np
import pytest
import torch

from oracle import models as om

from _parity import LL_ATOL, LL_RTOL, close_elementwise
pytestmark = pytest.mark.gpu

CELLS, F, H, L = 4096, 32738, (100, 100), 25


@pytest.fixture(scope="module")
def full_size(cuda_device):
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import philox_normal, synthetic_count_matrix
    eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                 device=cuda_device, seed=0)
    g = torch.Generator().manual_seed(1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    for name, m in eng.named_moving_statistics().items():
        if name.endswith("moving_mean"):
            m.copy_(torch.randn(m.shape, generator=g) * 0.2)
        else:
            m.copy_(torch.rand(m.shape, generator=g) + 0.5)
    matrix, _ = synthetic_count_matrix(CELLS, F, density=0.05, seed=60,
                                       device=cuda_device)
    x = torch.empty(CELLS, F, device=cuda_device)
    row_const = torch.empty(CELLS, device=cuda_device)
    matrix.gather_dense(torch.arange(CELLS, device=cuda_device), out=x,
                        row_const_out=row_const)
    eps = torch.empty(1, CELLS, L, device=cuda_device)
    philox_normal(eps[0], row_offset=0, seed=1, stream_id=0)
    return eng, x, row_const, eps


def _training_step(eng, x, row_const, eps):
    ll = torch.zeros(CELLS, device=x.device)
    scalars = eng.step(x, x, eps=eps, row_const=row_const, training=True,
                       outputs={"log_p_x_given_z": ll}).clone()
    torch.cuda.synchronize()
    return scalars.cpu().numpy(), ll.cpu().numpy(), eng.grads.clone()


def test_fused_and_unfused_paths_agree(full_size):
    eng, x, row_const, eps = full_size
    from scvae_amd import _lib
    assert eng.lib.scvae_decoder_fused_variant(_lib.LIKELIHOOD_KINDS[
        "negative binomial"][0], H[0]) in (1, 2)
    eng.set_fused(True)
    eng.set_dd_atomics(False)    # (`scvae train --deterministic`: the fixed-order slabs)
    s_f, ll_f, g_f = _training_step(eng, x, row_const, eps)
    again = _training_step(eng, x, row_const, eps)
    assert np.array_equal(s_f, again[0]) and torch.equal(g_f, again[2]), \
        "a repeated step must be bitwise reproducible"
    eng.set_dd_atomics(True)     # (the plan's default for everything below)
    eng.set_fused(False)
    s_u, ll_u, g_u = _training_step(eng, x, row_const, eps)
    eng.set_fused(True)
    assert abs(s_f[0] - s_u[0]) <= 1e-6 * abs(s_u[0])
    assert np.abs(ll_f - ll_u).max() <= 1e-5 * np.abs(ll_u).max()
    # lower_bound == reconstruction_error - kl_divergence (va:2596-2632)
    assert abs(s_f[0] - (s_f[2] - s_f[3])) <= 1e-6 * abs(s_f[0])
    for name, (offset, shape) in eng.param_table.items():
        n = int(np.prod(shape))
        a, b = g_f[offset:offset + n], g_u[offset:offset + n]
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-4 * scale + 1e-12, name


def test_subset_of_cells_matches_oracle_in_evaluation_mode(full_size):
    eng, x, row_const, eps = full_size
    ll = torch.zeros(CELLS, device=x.device)
    eng.step(x, x, eps=eps, row_const=row_const, training=False,
             outputs={"log_p_x_given_z": ll})
    torch.cuda.synchronize()
    rows = np.arange(0, CELLS, 97)[:40]
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="negative binomial")
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    xs = x[rows].cpu().double()
    out = om.vae_forward(cfg, params, moving, xs, xs,
                         eps[:, rows].cpu().double(), False)
    want = out["log_p_x_given_z"].reshape(-1).numpy()
    got = ll.cpu().numpy()[rows]
    close_elementwise(got, want, rtol=LL_RTOL, atol=LL_ATOL,
                      what="per-cell log-likelihood")


def test_per_cell_outputs_follow_a_row_permutation(full_size):
    eng, x, row_const, eps = full_size
    dev = x.device
    ll = torch.zeros(CELLS, device=dev)
    eng.step(x, x, eps=eps, row_const=row_const, training=False,
             outputs={"log_p_x_given_z": ll})
    perm = torch.randperm(CELLS, generator=torch.Generator().manual_seed(3)
                          ).to(dev)
    ll_p = torch.zeros(CELLS, device=dev)
    eng.step(x[perm].contiguous(), x[perm].contiguous(),
             eps=eps[:, perm].contiguous(),
             row_const=row_const[perm].contiguous(), training=False,
             outputs={"log_p_x_given_z": ll_p})
    torch.cuda.synchronize()
    assert (ll_p - ll[perm]).abs().max().item() <= 2e-6 * ll.abs().max().item()


@pytest.mark.parametrize("likelihood", ["negative binomial",
                                        "zero-inflated negative binomial"])
def test_gmvae_fused_and_unfused_paths_agree_at_full_width(cuda_device,
                                                          likelihood):
    """GMVAE (K = 5 passes x 512 cells) at the full gene count: the fused decoder
    kernel against the unfused kernels, incl. the three-head ZINB schedule."""
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import philox_normal, synthetic_count_matrix
    K, B, Lz = 5, 512, 20
    eng = Engine(F, Lz, H, likelihood, batch_norm=True, model_type="GMVAE",
                 n_clusters=K, device=cuda_device, seed=0)
    matrix, _ = synthetic_count_matrix(B, F, density=0.05, seed=61,
                                       device=cuda_device)
    x = torch.empty(B, F, device=cuda_device)
    row_const = torch.empty(B, device=cuda_device)
    matrix.gather_dense(torch.arange(B, device=cuda_device), out=x,
                        row_const_out=row_const)
    eps = torch.empty(K, 1, B, Lz, device=cuda_device)
    for k in range(K):
        philox_normal(eps[k, 0], row_offset=k * B, seed=1, stream_id=0)
    results = []
    for fused in (True, False):
        eng.set_fused(fused)
        ll = torch.zeros(K * B, device=cuda_device)
        scalars = eng.step(x, x, eps=eps, row_const=row_const, training=True,
                           outputs={"log_p_x_given_z": ll}).clone()
        torch.cuda.synchronize()
        results.append((scalars.cpu().numpy(), ll.cpu().numpy(),
                        eng.grads.clone()))
    eng.set_fused(True)
    (s_f, ll_f, g_f), (s_u, ll_u, g_u) = results
    assert abs(s_f[0] - s_u[0]) <= 2e-6 * abs(s_u[0])
    assert np.abs(ll_f - ll_u).max() <= 1e-5 * np.abs(ll_u).max()
    for name, (offset, shape) in eng.param_table.items():
        n = int(np.prod(shape))
        a, b = g_f[offset:offset + n], g_u[offset:offset + n]
        scale = b.abs().max().item()
        # (the q(y|x) gradients take differences of per-cluster log-likelihoods of
        # magnitude 1e4: their fp32 rounding shows at the 1e-4 level)
        assert (a - b).abs().max().item() <= 5e-4 * scale + 1e-12, name


def test_dropout_at_full_size(full_size, cuda_device):
    """Dropout on every site at the benchmark's size: the masks are a function
    of the seed (bitwise repeatable step, different seed -> different step), a
    mask keeps its fraction of a [4096, 32738] input, and the gradient of the
    first encoder layer only sees the genes / cells its mask kept."""
    from scvae_amd.engine import Engine
    _, x, row_const, eps = full_size
    keeps = (0.9, 0.8, 0.7)
    eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                 device=cuda_device, seed=0, dropout_keep_probabilities=keeps)

    def step(seed):
        scalars = eng.step(x, x, eps=eps, row_const=row_const, training=True,
                           dropout_seed=seed).clone()
        torch.cuda.synchronize()
        return scalars.cpu().numpy(), eng.grads.clone()
    s1, g1 = step(11)
    s2, g2 = step(11)
    s3, g3 = step(12)
    assert np.isfinite(s1[:4]).all() and torch.isfinite(g1).all()
    assert np.array_equal(s1, s2) and torch.equal(g1, g2)
    assert s1[0] != s3[0] and not torch.equal(g1, g3)
    mask = eng.dropout_mask(0, CELLS, F, keeps[1], 11)
    kept = (mask > 0)
    assert abs(kept.float().mean().item() - keeps[1]) < 1e-3
    # dW(ENCODER/1) = (x * mask / keep)^T dA: a gene whose kept entries are all
    # zero counts gets a zero row
    dead = ((x * mask) != 0).sum(dim=0) == 0
    if dead.any():
        dW = eng.gradient("ENCODER/1/DENSE/weights")
        assert dW[dead].abs().max().item() == 0.0
    # evaluation ignores the dropout and takes the fused path
    a = eng.step(x, x, eps=eps, row_const=row_const, training=False).clone()
    b = eng.step(x, x, eps=eps, row_const=row_const, training=False).clone()
    assert torch.equal(a, b)
