"""The minibatch as tile-indexed non-zeros (``scvae_count_tiles``,
include/scvae_hip.h) and the input layer's two products read from it
(``scvae_count_gemm_tiles``: mu:53-59 on ``x_train[idx].toarray()``,
va:997-998, and its weight gradient).

* the tiles decode to exactly the rows scipy returns (``csr[idx].toarray()``),
  bucket by bucket: pointers ascending, every entry in its bucket, the hi / lo
  cut of counts above 8 significant bits, the lo flags;
* ``scvae_count_gemm_tiles`` is BIT-IDENTICAL to ``scvae_count_gemm_u16`` on the
  same rows, both modes, at ragged shapes, with dense buckets (more entries than
  the kernels' static loads), counts up to 65 535, empty rows, a last group of
  fewer than 16 rows, and at the benchmark's size;
* a training / evaluation step given the tiles is bit-identical to the step on
  the uint16 batch alone (scalars, per-cell log-likelihood, every gradient,
  updated weights, moving statistics), carried fetch included.
"""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu

H = (100, 100)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _matrix(rng, n, F, density, big=0.002, empty_rows=True):
    x = rng.poisson(3.0, size=(n, F)) * (rng.random((n, F)) < density)
    x = x.astype(np.int64)
    k = int(n * F * density * big) + 2
    x.flat[rng.integers(0, x.size, k)] = rng.integers(256, 65536, k)
    x.flat[0] = 65535
    x.flat[-1] = 257          # 9 significant bits: hi 256, lo 1
    x[n // 2, F // 3] = 256   # 9 bits, remainder zero: ONE entry
    if empty_rows and n > 3:
        x[1] = 0
        x[n - 2] = 0
    return sp.csr_matrix(x.astype(np.float32))


def _decode(tiles, n, F, lib):
    """Dense [n, F] int64 hi + lo planes, and per-bucket checks, on the host."""
    T = int(lib.scvae_count_tiles_padded(F))
    groups = (n + 15) // 16
    ent = tiles.entries.cpu().numpy().view(np.uint32)
    tp = tiles.tile_ptr.cpu().numpy().view(np.uint32).reshape(-1, T + 1)[:groups]
    bp = tiles.block_ptr.cpu().numpy().view(np.uint32).reshape(-1, T // 16 + 1)[:groups]
    hi = np.zeros((groups * 16, T * 32), np.int64)
    lo = np.zeros_like(hi)
    for g in range(groups):
        ptr = tp[g] & 0x7FFFFFFF
        flag = tp[g] >> 31
        assert ptr[0] == g * tiles.capacity
        assert (np.diff(ptr.astype(np.int64)) >= 0).all()
        assert ptr[-1] - ptr[0] <= tiles.capacity
        assert np.array_equal(bp[g] & 0x7FFFFFFF, ptr[::16])
        has_lo = np.zeros(T, bool)
        e = ent[ptr[0]:ptr[-1]]
        tile_of = np.searchsorted(ptr, np.arange(ptr[0], ptr[-1]), side="right") - 1
        assert np.array_equal((e >> 9) & 15, tile_of & 15)
        row = (e >> 5) & 15
        gene = tile_of * 32 + (e & 31)
        # the value travels as its bf16 bit pattern (exact: at most 8 significant bits)
        as_float = ((e >> 16).astype(np.uint32) << 16).view(np.float32)
        value = as_float.astype(np.int64)
        assert np.array_equal(value.astype(np.float32), as_float)
        is_lo = ((e >> 13) & 1).astype(bool)
        assert (value > 0).all()
        # at most 8 significant bits per entry: exact in bf16
        sig = np.floor(np.log2(value)).astype(np.int64) + 1
        tz = np.array([(int(v) & -int(v)).bit_length() - 1 for v in value])
        assert ((sig - tz) <= 8).all()
        for plane, sel in ((hi, ~is_lo), (lo, is_lo)):
            assert (plane[g * 16 + row[sel], gene[sel]] == 0).all()   # one entry per position
            plane[g * 16 + row[sel], gene[sel]] = value[sel]
        has_lo[np.unique(tile_of[is_lo])] = True
        assert np.array_equal(flag[:-1].astype(bool), has_lo)
        assert np.array_equal((bp[g][:-1] >> 31).astype(bool),
                              has_lo.reshape(-1, 16).any(axis=1))
    return hi[:, :], lo


@pytest.mark.parametrize("n,F,density", [
    (64, 1000, 0.05),
    (37, 203, 0.3),          # a last group of five rows, 203 genes: one block
    (16, 32, 0.9),           # one dense bucket
    (100, 32738, 0.05),      # the reference's default minibatch, cfg2-4 width
    (5, 65536, 0.01),        # the widest matrix the format takes
])
def test_count_tiles_hold_the_minibatch(cuda_device, n, F, density):
    from scvae_amd import _lib
    from scvae_amd.minibatch import DeviceCSR
    lib = _lib.load()
    rng = np.random.default_rng(n + F)
    N = 3 * n + 7
    csr = _matrix(rng, N, F, density)
    m = DeviceCSR.from_scipy(csr, cuda_device)
    assert m.count_tiles_supported
    idx = rng.permutation(N)[:n]
    rows = torch.from_numpy(idx).to(cuda_device)
    tiles = m.count_tiles(n)
    m.gather_count_tiles(rows, tiles)
    torch.cuda.synchronize()
    assert int(tiles.status.item()) == 0
    hi, lo = _decode(tiles, n, F, lib)
    want = np.asarray(csr[idx].toarray()).astype(np.int64)
    got = hi + lo
    assert (got[n:] == 0).all() and (got[:, F:] == 0).all()
    assert np.array_equal(got[:n, :F], want)
    # the cut is the dense kernels': hi = the upper 8 significant bits
    f = want.astype(np.float32).view(np.uint32) & 0xFFFF0000
    assert np.array_equal(hi[:n, :F], f.view(np.float32).astype(np.int64))
    # the capacity bound is exact: no row contributes more than max_row_entries
    per_row = (want > 0).sum(1) + (lo[:n, :F] > 0).sum(1)
    assert per_row.max() <= m.max_row_entries


def _both(lib, mode, m, rows_idx, other, bias, relu=False, x16=None, tiles=None):
    from scvae_amd import _lib
    n, F = int(rows_idx.numel()), m.shape[1]
    if x16 is None:
        x16 = m.gather_counts_u16(rows_idx)
    if tiles is None:
        tiles = m.count_tiles(n)
        m.gather_count_tiles(rows_idx, tiles)
    N = other.shape[1]
    M = n if mode == 0 else F
    nbytes = lib.scvae_count_gemm_workspace_bytes(mode, n, F, N)
    assert nbytes >= 0
    ws = torch.empty(nbytes + 16, dtype=torch.uint8, device=x16.device)
    dense = torch.full((M, N), float("nan"), device=x16.device)
    _lib.check(lib.scvae_count_gemm_u16(
        mode, _p(x16), x16.stride(0), n, F, _p(other), N, N, _p(bias),
        1 if relu else 0, _p(dense), N, _p(ws), nbytes, _stream()),
        "scvae_count_gemm_u16")
    sparse = torch.full((M, N), float("nan"), device=x16.device)
    _lib.check(lib.scvae_count_gemm_tiles(
        mode, ctypes.byref(tiles.struct), _p(x16), x16.stride(0), n, F,
        _p(other), N, N, _p(bias), 1 if relu else 0, _p(sparse), N, _p(ws),
        nbytes, _stream()), "scvae_count_gemm_tiles")
    torch.cuda.synchronize()
    return dense, sparse


@pytest.mark.parametrize("n,F,N,density", [
    (100, 32738, 100, 0.05),
    (1024, 32738, 100, 0.05),
    (512, 27998, 100, 0.05),     # cfg5's gene count
    (37, 203, 24, 0.3),          # ragged everything
    (64, 64, 128, 0.5),          # the widest layer; dense buckets
    (272, 2048, 100, 0.6),       # buckets beyond the static loads, both kernels
    (5, 1000, 1, 0.2),
    (700, 33, 7, 0.4),
    (16, 100, 100, 1.0),         # every position a count
])
@pytest.mark.parametrize("mode", [0, 1])
def test_count_gemm_tiles_is_count_gemm_u16(cuda_device, n, F, N, density, mode):
    from scvae_amd import _lib
    from scvae_amd.minibatch import DeviceCSR
    lib = _lib.load()
    rng = np.random.default_rng(n * 7 + F + N + mode)
    total = n + 9
    csr = _matrix(rng, total, F, density, empty_rows=n > 16)
    m = DeviceCSR.from_scipy(csr, cuda_device)
    idx = rng.permutation(total)[:n]
    rows_idx = torch.from_numpy(idx).to(cuda_device)
    K = F if mode == 0 else n
    oh = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-12, 1, (K, N)))
          ).astype(np.float32)
    other = torch.from_numpy(oh).to(cuda_device)
    bias = (torch.from_numpy(rng.standard_normal(N).astype(np.float32))
            .to(cuda_device) if mode == 0 else None)
    dense, sparse = _both(lib, mode, m, rows_idx, other, bias, relu=mode == 0)
    assert torch.isfinite(sparse).all()
    assert torch.equal(dense, sparse)
    # and both are the product (fp64, scipy on the host)
    x = csr[idx].astype(np.float64)
    want = np.asarray(x @ oh.astype(np.float64) if mode == 0
                      else x.T @ oh.astype(np.float64))
    if bias is not None:
        want = np.maximum(want + bias.cpu().numpy().astype(np.float64), 0.0)
    scale = max(np.abs(want).max(), 1e-30)
    assert np.abs(sparse.cpu().numpy() - want).max() <= 2e-6 * scale


@pytest.mark.parametrize("n,F", [(4096, 32738), (512, 27998)])
@pytest.mark.parametrize("mode", [0, 1])
def test_count_gemm_tiles_at_benchmark_size(cuda_device, n, F, mode):
    from scvae_amd import _lib
    from scvae_amd.minibatch import synthetic_count_matrix
    lib = _lib.load()
    m, _ = synthetic_count_matrix(n + 1000, F, density=0.05, seed=70 + mode,
                                  device=cuda_device)
    g = torch.Generator(device=cuda_device).manual_seed(5)
    rows_idx = torch.randperm(n + 1000, generator=g, device=cuda_device)[:n]
    N = 100
    K = F if mode == 0 else n
    other = torch.randn(K, N, generator=g, device=cuda_device) * 0.03
    bias = torch.randn(N, generator=g, device=cuda_device) if mode == 0 else None
    dense, sparse = _both(lib, mode, m, rows_idx, other, bias)
    assert torch.equal(dense, sparse)


def _engine(device, F, L):
    from scvae_amd.engine import Engine
    eng = Engine(F, L, H, "negative binomial", batch_norm=True, device=device,
                 seed=0)
    g = torch.Generator().manual_seed(1)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return eng


@pytest.mark.parametrize("B,F", [(4096, 32738), (1024, 32738)])
def test_step_with_count_tiles_is_the_step_without(cuda_device, bit_repeatable,
                                                   B, F):
    """Two training steps (the second minibatch fetched by the first step as
    carried work, tiles included) with the optimiser in between and an
    evaluation step: the same bits with and without the tiles."""
    from scvae_amd.minibatch import philox_normal, synthetic_count_matrix
    L = 25
    m, _ = synthetic_count_matrix(2 * B + 50, F, density=0.05, seed=64,
                                  device=cuda_device)
    g = torch.Generator(device=cuda_device).manual_seed(2)
    perm = torch.randperm(2 * B + 50, generator=g, device=cuda_device)
    rows = [perm[:B].contiguous(), perm[B:2 * B].contiguous()]
    eps = torch.empty(1, B, L, device=cuda_device)
    philox_normal(eps[0], row_offset=0, seed=3, stream_id=0)
    results = []
    for with_tiles in (False, True):
        eng = _engine(cuda_device, F, L)
        assert eng.accepts_counts_u16(B, True)
        x = [torch.empty(B, m.u16_pitch, dtype=torch.uint16, device=cuda_device)
             for _ in range(2)]
        rc = [torch.empty(B, device=cuda_device) for _ in range(2)]
        tl = [m.count_tiles(B) if with_tiles else None for _ in range(2)]
        m.request(rows[0], x[0], rc[0], tiles=tl[0]).issue()
        out = []
        for i in range(2):
            ll = torch.zeros(B, device=cuda_device)
            nxt = (m.request(rows[1], x[1], rc[1], tiles=tl[1])
                   if i == 0 else None)
            s = eng.step(x[i], x[i], eps=eps, row_const=rc[i], training=True,
                         x_counts=True, outputs={"log_p_x_given_z": ll},
                         next_minibatch=nxt, count_tiles=tl[i]).clone()
            out += [s, ll.clone(), eng.grads.clone()]
            eng.adam_step(1e-4)
        ev = eng.step(x[1], x[1], eps=eps, row_const=rc[1], training=False,
                      x_counts=True, count_tiles=tl[1]).clone()
        torch.cuda.synchronize()
        if with_tiles:
            assert all(int(t.status.item()) == 0 for t in tl)
        results.append([t.cpu() for t in out + [ev, eng.moving, eng.params]])
    for i, (a, b) in enumerate(zip(*results)):
        assert torch.isfinite(a).all(), i
        assert torch.equal(a, b), i


_TWO_ROLE_SNIPPET = r"""
import ctypes, hashlib, sys
import numpy as np, torch
from scvae_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
digest = hashlib.sha256()
for rows, F, N, density, epilogue in ((1024, 32738, 100, 0.05, False), (300, 1000, 100, 0.3, False),
                                      (37, 203, 24, 0.3, True), (256, 64, 128, 0.5, True),
                                      (100, 32738, 64, 0.05, False), (5, 96, 1, 0.5, True)):
    rng = np.random.default_rng(rows + F)
    x = rng.poisson(3.0, size=(rows, F)) * (rng.random((rows, F)) < density)
    k = max(1, rows * F // 500)
    x.flat[rng.integers(0, x.size, k)] = rng.integers(256, 65536, k)
    x.flat[0] = 65535
    ld = (F + 7) // 8 * 8
    x16 = torch.zeros(rows, ld, dtype=torch.uint16, device=dev)
    x16[:, :F] = torch.from_numpy(x.astype(np.uint16)).to(dev)
    W = torch.from_numpy(rng.standard_normal((F, N)).astype(np.float32) * 0.05).to(dev)
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev) if epilogue else None
    nb = lib.scvae_count_gemm_workspace_bytes(0, rows, F, N)
    ws = torch.empty(nb + 16, dtype=torch.uint8, device=dev)
    out = torch.full((rows, N), float("nan"), device=dev)
    _lib.check(lib.scvae_count_gemm_u16(0, P(x16), ld, rows, F, P(W), N, N, P(b), 1 if epilogue else 0,
                                        P(out), N, P(ws), nb, st), "scvae_count_gemm_u16")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    digest.update(out.cpu().numpy().tobytes())
print("digest", digest.hexdigest())
"""


def test_two_role_forward_kernel_on_the_uint16_batch_is_bit_identical(cuda_device):
    """``SCVAE_CG_FWD2=1`` -- count_fwd2_kernel with x staged from the dense
    uint16 batch (four waves multiply, four stage) -- against the default
    count_gemm_fwd_kernel: the same bits on every shape, counts that need the lo
    plane, bias + ReLU written directly, ragged tails (mu:53-59)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for flag in ("0", "1"):
        env = dict(os.environ, SCVAE_CG_FWD2=flag, PYTHONPATH=root)
        done = subprocess.run([sys.executable, "-c", _TWO_ROLE_SNIPPET], env=env, cwd=root,
                              capture_output=True, text=True, timeout=300)
        assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
        digests.append(done.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1] and digests[0].startswith("digest ")
