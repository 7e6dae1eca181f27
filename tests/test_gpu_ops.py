"""GPU unit tests of individual C-ABI ops against fp64 CPU references."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import likelihoods as lk

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [
    (1, 1, 1), (37, 100, 203), (64, 64, 16), (100, 25, 100),
    (130, 70, 5000),   # split-K
    (300, 100, 33),
])
def test_gemm_matches_fp64(cuda_device, ta, tb, M, N, K):
    from scvae_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    # asymmetric operands (catches transposed fragment layouts)
    A = torch.randn((K, M) if ta else (M, K), generator=g, dtype=torch.float64)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g,
                     dtype=torch.float64)
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    ref = (A.T if ta else A) @ (Bm.T if tb else Bm) + bias
    Ad, Bd, bd = (v.float().to(cuda_device) for v in (A, Bm, bias))
    C = torch.full((M, N), 7.0, device=cuda_device)
    ws_bytes = lib.scvae_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=cuda_device)
    for relu, acc in ((0, 0), (1, 0), (0, 1)):
        C.fill_(7.0)
        _lib.check(lib.scvae_gemm(
            ta, tb, _p(Ad), _p(Bd), _p(bd), _p(C), M, N, K, Ad.shape[1],
            Bd.shape[1], N, relu, acc, _p(ws), ws_bytes, _stream()), "gemm")
        torch.cuda.synchronize()
        want = ref.clamp(min=0) if relu else ref
        if acc:
            want = want + 7.0
        scale = (A.abs().max() * Bm.abs().max() * np.sqrt(K)).item() + 1.0
        err = (C.cpu().double() - want).abs().max().item()
        assert err < 2e-6 * scale * max(1.0, np.sqrt(K) / 8), (err, scale)


@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("M,N,K", [(300, 100, 140000), (4500, 100, 9000),
                                   (2100, 130, 16000),
                                   # narrow-N kernel: 1-3 full column tiles + a
                                   # remainder of 1..8 columns (4x4x1 MFMA groups)
                                   (700, 97, 60000), (520, 104, 80000),
                                   (1000, 68, 70000), (900, 37, 120000)])
def test_big_gemm_matches_fp64(cuda_device, ta, M, N, K):
    """Shapes routed to the 128x128 and the narrow-N kernels (encoder input layer
    and its dW), with and without split-K, edge tiles in M, N and K."""
    from scvae_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    A = A * (torch.rand(A.shape, generator=g) < 0.1)      # sparse like counts
    Bm = torch.randn((K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = (A.double().T if ta else A.double()) @ Bm.double() + bias.double()
    Ad, Bd, bd = (v.to(cuda_device) for v in (A, Bm, bias))
    C = torch.zeros((M, N), device=cuda_device)
    ws_bytes = lib.scvae_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=cuda_device)
    _lib.check(lib.scvae_gemm(
        ta, 0, _p(Ad), _p(Bd), _p(bd), _p(C), M, N, K, Ad.shape[1], N, N, 0, 0,
        _p(ws), ws_bytes, _stream()), "gemm")
    torch.cuda.synchronize()
    err = (C.cpu().double() - ref).abs().max().item()
    assert err < 3e-5 * ref.abs().max().item(), err


@pytest.mark.parametrize("name", list(lk.ELEMENTWISE_LIKELIHOODS))
def test_loglik_forward_backward(cuda_device, name):
    from scvae_amd import _lib
    lib = _lib.load()
    kind, heads = _lib.LIKELIHOOD_KINDS[name]
    P = len(heads)
    cells, S, F = 9, 2, 517
    rows = cells * S
    rng = np.random.default_rng(3)
    t = rng.poisson(2.0, size=(cells, F)).astype(np.float64)
    t *= rng.random((cells, F)) > 0.6
    t[0, :6] = [0, 1, 17, 250, 4000, 30000]
    pre = [rng.normal(0, 3.0, size=(rows, F)) for _ in range(P)]
    # exercise the clips
    for j in range(P):
        pre[j][1, :4] = [-95.0, 40.0, -11.0, 11.0]
    gw = rng.normal(size=rows)
    tt = torch.from_numpy(t)
    pre_t = [torch.from_numpy(a).requires_grad_(True) for a in pre]
    lp = lk.log_prob(name, tt.repeat(S, 1), tuple(pre_t)).sum(dim=-1)
    (lp * torch.from_numpy(gw)).sum().backward()

    td = tt.float().to(cuda_device)
    pre_d = [torch.from_numpy(a).float().to(cuda_device) for a in pre]
    arr = (ctypes.c_void_p * P)(*[a.data_ptr() for a in pre_d])
    ll = torch.zeros(rows, device=cuda_device)
    _lib.check(lib.scvae_loglik_fwd(kind, _p(td), arr, None, _p(ll), rows,
                                    cells, F, _stream()), "loglik_fwd")
    torch.cuda.synchronize()
    want = lp.detach().numpy()
    assert np.abs(ll.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()

    # row constant variant
    rc = torch.lgamma(tt + 1).sum(dim=1).float().to(cuda_device)
    ll2 = torch.zeros(rows, device=cuda_device)
    gwd = torch.from_numpy(gw).float().to(cuda_device)
    _lib.check(lib.scvae_loglik_bwd(kind, _p(td), arr, _p(gwd), _p(rc),
                                    _p(ll2), rows, cells, F, _stream()),
               "loglik_bwd")
    torch.cuda.synchronize()
    assert np.abs(ll2.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    for j in range(P):
        g_ref = pre_t[j].grad.numpy()
        g = pre_d[j].cpu().numpy()
        # element-wise: relative to the element's scale + a small absolute floor
        tol = 2e-5 * np.abs(g_ref) + 2e-5 * (1 + np.abs(t).repeat(S, 0).reshape(
            S, cells, F).reshape(rows, F) * 0 + np.abs(gw)[:, None])
        bad = np.abs(g - g_ref) > tol * 5
        assert not bad.any(), (heads[j], np.argwhere(bad)[:5],
                               g[bad][:5], g_ref[bad][:5])


def test_philox_normal_blocks_equals_block_by_block(cuda_device):
    """The stacked passes of a shard in one launch (``philox_normal_blocks``)
    are the per-block launches, bit for bit."""
    from scvae_amd.minibatch import philox_normal, philox_normal_blocks
    blocks, rows, cols, stride, offset = 5, 37, 11, 100, 23
    got = torch.full((blocks, rows, cols), float("nan"), device=cuda_device)
    philox_normal_blocks(got, block_stride=stride, row_offset=offset, seed=99,
                         stream_id=(3 << 40) + 5)
    want = torch.empty_like(got)
    for g in range(blocks):
        philox_normal(want[g], row_offset=g * stride + offset, seed=99,
                      stream_id=(3 << 40) + 5)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_philox_normal_is_sharding_invariant(cuda_device):
    from scvae_amd import _lib
    lib = _lib.load()
    rows, cols = 1000, 25
    full = torch.zeros(rows, cols, device=cuda_device)
    _lib.check(lib.scvae_philox_normal(_p(full), rows, cols, 0, 1234, 7,
                                       _stream()), "philox")
    part = torch.zeros(400, cols, device=cuda_device)
    _lib.check(lib.scvae_philox_normal(_p(part), 400, cols, 600, 1234, 7,
                                       _stream()), "philox")
    torch.cuda.synchronize()
    assert torch.equal(full[600:], part)
    z = full.cpu().double()
    assert abs(z.mean().item()) < 0.03 and abs(z.std().item() - 1) < 0.03
    other = torch.zeros(rows, cols, device=cuda_device)
    _lib.check(lib.scvae_philox_normal(_p(other), rows, cols, 0, 1234, 8,
                                       _stream()), "philox")
    torch.cuda.synchronize()
    assert not torch.equal(full, other)


def test_random_streams_match_the_numpy_restatement(cuda_device):
    """Bit-exact known answers for the counter-based generator: the dropout
    masks and the N(0,1) draws of the kernels against oracle/philox.py, which
    itself reproduces Random123's published vectors (test_oracle_kat.py)."""
    from oracle import philox
    from scvae_amd import _lib
    lib = _lib.load()
    for rows, cols, keep, seed, site in [
            (37, 101, 0.8, 0x1234ABCD5678, 3), (5, 4, 0.5, 7, 0),
            (130, 33, 0.9, 2 ** 63 + 11, 51), (1, 1, 0.25, 1, 80)]:
        ones = torch.ones(rows, cols, device=cuda_device)
        out = torch.empty_like(ones)
        _lib.check(lib.scvae_dropout_apply(
            _p(ones), _p(out), rows, cols, keep, seed, site, 0, _stream()),
            "dropout")
        torch.cuda.synchronize()
        want = philox.dropout_mask(rows, cols, keep, seed, site)
        assert np.array_equal(out.cpu().numpy(), want), (rows, cols, site)
    for rows, cols, offset, seed, stream_id in [
            (50, 25, 0, 1, 0), (33, 7, 1000, 0xDEADBEEF12, 41),
            (8, 100, 2 ** 33, 5, 2),
            # 64-bit stream ids: the host uses the high bits as domain separators
            # (evaluation passes (1 << 40) + ..., model.sample() (1 << 41) + j)
            (16, 25, 0, 1, (1 << 40) + 3), (16, 25, 0, 1, (1 << 41) + 3)]:
        out = torch.empty(rows, cols, device=cuda_device)
        _lib.check(lib.scvae_philox_normal(
            _p(out), rows, cols, offset, seed, stream_id, _stream()), "philox")
        torch.cuda.synchronize()
        want = philox.standard_normal(rows, cols, offset, seed, stream_id)
        assert np.abs(out.cpu().numpy() - want).max() < 2e-5
    # streams that differ only above bit 31 are different streams
    draws = []
    for stream_id in (3, (1 << 40) + 3, (1 << 41) + 3):
        out = torch.empty(16, 25, device=cuda_device)
        _lib.check(lib.scvae_philox_normal(
            _p(out), 16, 25, 0, 1, stream_id, _stream()), "philox")
        draws.append(out.cpu())
    assert not torch.equal(draws[0], draws[1])
    assert not torch.equal(draws[0], draws[2])
    assert not torch.equal(draws[1], draws[2])


def test_csr_densify(cuda_device):
    import scipy.sparse as sp
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    N, F = 50, 333
    dense = rng.poisson(0.3, size=(N, F)).astype(np.float32)
    dense[7] = 0  # empty row
    m = sp.csr_matrix(dense)
    indptr = torch.from_numpy(m.indptr.astype(np.int64)).to(cuda_device)
    indices = torch.from_numpy(m.indices.astype(np.int32)).to(cuda_device)
    values = torch.from_numpy(m.data.astype(np.float32)).to(cuda_device)
    rows = torch.tensor([49, 7, 0, 7, 13], dtype=torch.int64,
                        device=cuda_device)
    out = torch.full((5, F), -1.0, device=cuda_device)
    _lib.check(lib.scvae_csr_densify(_p(indptr), _p(indices), _p(values),
                                     _p(rows), 5, F, _p(out), _stream()),
               "densify")
    lg = torch.zeros(N, device=cuda_device)
    _lib.check(lib.scvae_csr_row_lgamma1p(_p(indptr), _p(values), N, _p(lg),
                                          _stream()), "lgamma1p")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), dense[[49, 7, 0, 7, 13]])
    want = torch.lgamma(torch.from_numpy(dense).double() + 1).sum(dim=1)
    assert np.allclose(lg.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("F", [1, 3, 5, 4097, 8191, 8192, 8193, 16390, 20001])
def test_csr_densify_segments_and_alignment(cuda_device, F):
    """Widths around the 8192-column segments, odd row pitches (every
    misalignment of a row start), unsorted column indices; a canary row after
    the output must stay untouched."""
    import scipy.sparse as sp
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(F)
    N = 9
    dense = (rng.poisson(0.2, size=(N, F))
             * rng.integers(1, 50, size=(N, F))).astype(np.float32)
    dense[3] = 0
    dense[:, 0] = 7
    dense[:, F - 1] = 9
    m = sp.csr_matrix(dense)
    # shuffle the entries inside every row: the kernel assumes no order
    data, idx = m.data.copy(), m.indices.copy()
    for r in range(N):
        lo, hi = m.indptr[r], m.indptr[r + 1]
        perm = rng.permutation(hi - lo)
        data[lo:hi] = data[lo:hi][perm]
        idx[lo:hi] = idx[lo:hi][perm]
    indptr = torch.from_numpy(m.indptr.astype(np.int64)).to(cuda_device)
    indices = torch.from_numpy(idx.astype(np.int32)).to(cuda_device)
    values = torch.from_numpy(data.astype(np.float32)).to(cuda_device)
    order = [8, 3, 0, 5, 5, 1, 2]
    rows = torch.tensor(order, dtype=torch.int64, device=cuda_device)
    for offset in range(4):   # start of the output buffer: any 4-byte phase
        buf = torch.full((offset + (len(order) + 1) * F,), -1.0,
                         device=cuda_device)
        out = buf[offset:]
        _lib.check(lib.scvae_csr_densify(
            _p(indptr), _p(indices), _p(values), _p(rows), len(order), F,
            ctypes.c_void_p(out.data_ptr()), _stream()), "densify")
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[:len(order) * F].reshape(-1, F),
                              dense[order]), (F, offset)
        assert (got[len(order) * F:] == -1).all() and (
            buf[:offset].cpu().numpy() == -1).all()


def test_gemm_small_shape_sweep(cuda_device):
    """Every transpose combination over a grid of small / ragged shapes (the
    [rows, <=128] layers of the model), with and without accumulation."""
    import ctypes
    import itertools
    from scvae_amd import _lib
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(0)
    for ta, tb in itertools.product((0, 1), (0, 1)):
        for M, N, K in itertools.product((1, 11, 28, 64, 84, 130),
                                         (1, 4, 16, 25, 100),
                                         (1, 2, 7, 16, 84, 100)):
            for acc in (0, 1):
                A = rng.standard_normal((K, M) if ta else (M, K))
                B = rng.standard_normal((N, K) if tb else (K, N))
                C0 = rng.standard_normal((M, N))
                Ad = torch.tensor(A, dtype=torch.float32, device=cuda_device)
                Bd = torch.tensor(B, dtype=torch.float32, device=cuda_device)
                Cd = torch.tensor(C0, dtype=torch.float32, device=cuda_device)
                ws_bytes = lib.scvae_gemm_workspace_bytes(M, N, K)
                ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8,
                                 device=cuda_device)
                _lib.check(lib.scvae_gemm(
                    ta, tb, Ad.data_ptr(), Bd.data_ptr(), None, Cd.data_ptr(),
                    M, N, K, Ad.shape[1], Bd.shape[1], N, 0, acc,
                    ws.data_ptr(), ws_bytes, stream), "gemm")
                want = ((A.T if ta else A) @ (B.T if tb else B)
                        + (C0 if acc else 0))
                err = np.abs(Cd.cpu().double().numpy() - want).max()
                assert err <= 1e-5 * max(1.0, np.abs(want).max()) * max(
                    1.0, np.sqrt(K)), (ta, tb, M, N, K, acc, err)


def test_workspace_guard_mode_catches_a_write_past_a_buffer(cuda_device):
    """``SCVAE_WS_GUARD=1`` (a debugging aid of the library, read once per
    process): every buffer carved out of a plan's workspace is followed by a
    guard region that ``scvae_plan_step`` checks after the step.  A clean step
    passes; a byte changed in the last buffer's guard makes the next step fail
    and name the mode."""
    import os
    import subprocess
    import sys
    code = r"""
import torch
from scvae_amd import _lib
from scvae_amd.engine import Engine
dev = torch.device("cuda:0")
eng = Engine(40, 3, (16, 12), "negative binomial", batch_norm=True, device=dev, seed=0, k_max=1,
             dropout_keep_probabilities=(0.9, 0.9, 0.9))
eng.reserve(24, 2)
x = torch.poisson(torch.full((24, 40), 2.0, device=dev))
eps = torch.randn(2, 24, 3, device=dev)
for training in (True, False):
    eng.step(x, x, eps=eps, training=training, n_iw=2, n_mc=1, dropout_seed=5)
torch.cuda.synchronize()
nbytes = eng.lib.scvae_plan_workspace_bytes(eng.handle, 24, 2)
eng.workspace[nbytes - 1] = 0        # the last buffer's guard
try:
    eng.step(x, x, eps=eps, training=True, n_iw=2, n_mc=1, dropout_seed=5)
except _lib.HipLibraryError as error:
    assert "SCVAE_WS_GUARD" in str(error), str(error)
    print("CAUGHT")
"""
    env = dict(os.environ, SCVAE_WS_GUARD="1",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CAUGHT" in out.stdout, out.stdout + out.stderr[-2000:]
