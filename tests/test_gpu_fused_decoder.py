"""Edge shapes of the fused decoder-head kernel through the C ABI against an
fp64 torch reference: row / gene counts around the 64-wide tiles, hidden sizes
around the 32-wide MFMA tiles (incl. multiples of 32, where the ones-column of
db lands in a tile of its own), queue overflow on dense counts, all kinds."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import likelihoods as lk

pytestmark = pytest.mark.gpu


ARITH = {"flag": 0}


@pytest.fixture(params=["fp32", "bf16x9", "bf16x6"], autouse=True)
def head_arith(request):
    """Every case on the three arithmetics of the training kernels: the fp32
    matrix cores, the exact nine-term bf16 split (decoder_fused3.hip, where it
    applies) and its six-term form (the producer / consumer kernel beyond 128
    rows; everything else then runs as bf16x9) -- the arithmetic travels with
    each call (``train | SCVAE_HEADS_*``), the tolerances are the same."""
    from scvae_amd import _lib
    ARITH["flag"] = _lib.HEAD_ARITH_FLAGS[request.param]
    yield request.param
    ARITH["flag"] = 0


def _run(device, name, rows, cells, F, H, density, seed=0, row_const=True,
         extra_flags=0, modes=(0, 1)):
    from scvae_amd import _lib
    lib = _lib.load()
    kind, heads = _lib.LIKELIHOOD_KINDS[name]
    P = len(heads)
    rng = np.random.default_rng(seed)
    d = np.maximum(rng.normal(0, 1, (rows, H)), 0)
    W = [rng.normal(0, 0.3, (H, F)) for _ in range(P)]
    b = [rng.normal(0, 0.3, F) for _ in range(P)]
    t = rng.poisson(3.0, (cells, F)) * (rng.random((cells, F)) < density)
    t = t.astype(np.float64)
    if t.size:
        t.flat[0] = 5000.0
    gw = rng.normal(0, 1, rows)

    T = torch.from_numpy
    dt = T(d).requires_grad_(True)
    Wt = [T(w).requires_grad_(True) for w in W]
    bt = [T(v).requires_grad_(True) for v in b]
    pre = tuple(dt @ w + v for w, v in zip(Wt, bt))
    reps = rows // cells
    ll_ref = lk.log_prob(name, T(t).repeat(reps, 1), pre).sum(dim=1)
    (ll_ref * T(gw)).sum().backward()

    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(device)
    dd_, td, gwd = f32(d), f32(t), f32(gw)
    Wd, bd = [f32(w) for w in W], [f32(v) for v in b]
    dWd = [torch.full_like(w, 7.0) for w in Wd]
    dbd = [torch.full_like(v, 7.0) for v in bd]
    ll = torch.full((rows,), 7.0, device=device)
    dd = torch.full((rows, H), 7.0, device=device)
    rc = torch.lgamma(td.double() + 1).sum(dim=1).float() if row_const else None
    ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F),
                     dtype=torch.uint8, device=device)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for train in modes:
        _lib.check(lib.scvae_decoder_fused(
            kind, train | ARITH["flag"] | (extra_flags if train else 0),
            dd_.data_ptr(), rows, H, arr(Wd), arr(bd), arr(dWd),
            arr(dbd), F, td.data_ptr(), cells, gwd.data_ptr(),
            rc.data_ptr() if rc is not None else None, ll.data_ptr(),
            dd.data_ptr(), ws.data_ptr(), stream), "scvae_decoder_fused")
        torch.cuda.synchronize()
        want = ll_ref.detach().numpy()
        got = ll.cpu().double().numpy()
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-4, (
            train, np.abs(got - want).max(), np.abs(want).max())

    def close(got, want, what):
        want = want.numpy()
        got = got.cpu().double().numpy()
        scale = np.abs(want).max() + 1e-12
        assert np.abs(got - want).max() <= 3e-5 * scale, (
            what, np.abs(got - want).max(), scale)
    close(dd, dt.grad, "dd")
    for j in range(P):
        close(dWd[j], Wt[j].grad, "dW%d" % j)
        close(dbd[j], bt[j].grad, "db%d" % j)


@pytest.mark.parametrize("rows,F", [(1, 1), (33, 63), (64, 64), (65, 65),
                                    (130, 200), (97, 129)])
@pytest.mark.parametrize("name", list(lk.ELEMENTWISE_LIKELIHOODS))
def test_tile_edges(cuda_device, name, rows, F):
    _run(cuda_device, name, rows, rows, F, 20, 0.3)


@pytest.mark.parametrize("H", [2, 30, 32, 34, 64, 96, 100, 126])
def test_hidden_sizes(cuda_device, H):
    _run(cuda_device, "negative binomial", 70, 70, 150, H, 0.2)
    _run(cuda_device, "zero-inflated negative binomial", 40, 40, 90, H, 0.2)


@pytest.mark.parametrize("H", [111, 126, 127, 128, 129, 160, 192, 200, 224, 255, 256])
def test_wide_and_odd_hidden_sizes(cuda_device, H, head_arith):
    """``-H`` beyond the all-in-one-phase kernels' LDS budget (mu:81-126 takes any
    size): training launches of the bf16x9 producer / consumer kernel up to
    H = 255 -- 32-gene strips for two heads from H = 111, five to eight
    contraction steps and two h tiles per consumer wave from H = 127, odd widths
    included (three heads: up to H = 159) -- and the forward half of the same
    kernel (``FWD``: evaluation passes, the first pass of an importance-weighted
    step; mode 0 of ``_run``)."""
    from scvae_amd import _lib
    lib = _lib.load()
    if head_arith == "fp32":
        assert H > 126 or H % 2 == 0 or lib.scvae_decoder_train_kernel(1, H, 0) == 0
        pytest.skip("the fp32 kernels stop at even H <= 126")
    assert lib.scvae_decoder_train_kernel(1, H, 1) == 3
    _run(cuda_device, "negative binomial", 200, 200, 150, H, 0.2)
    _run(cuda_device, "poisson", 70, 70, 130, H, 0.2)
    _run(cuda_device, "negative binomial", 96, 48, 100, H, 0.3,
         extra_flags=_lib.HEADS_DD_ATOMICS)
    if H <= 159:
        _run(cuda_device, "zero-inflated negative binomial", 100, 100, 90, H, 0.2)
    else:
        assert lib.scvae_decoder_train_kernel(3, H, 1) == 0


@pytest.mark.parametrize("name", ["poisson", "negative binomial",
                                  "zero-inflated poisson",
                                  "zero-inflated negative binomial"])
@pytest.mark.parametrize("H", [40, 100])
def test_many_row_tiles(cuda_device, name, H):
    """Seven row tiles at hidden widths whose last contraction step reads past
    the weight planes (H + 1 = 41 -> 48 rows of 64, 101 -> 112 of 128): what
    lies behind them in LDS must stay harmless from the third tile on (the
    forward instantiation of the bf16x9 kernel keeps a row-sum buffer there)."""
    _run(cuda_device, name, 400, 400, 130, H, 0.1)


def test_dense_counts_overflow_the_queue(cuda_device):
    # every element is nonzero: the t > 0 queue overflows and falls back inline
    _run(cuda_device, "negative binomial", 128, 128, 256, 32, 1.0)
    _run(cuda_device, "zero-inflated negative binomial", 64, 64, 128, 32, 1.0)


def test_bernoulli_heads(cuda_device):
    """du:194-204 in the fused kernels: Bernoulli(logits) on binarised targets
    (no data-only term, no t > 0 correction); training and forward-only."""
    import oracle.likelihoods  # noqa: F401  (the fp64 reference of _run)
    for rows, F, H in ((70, 150, 20), (130, 200, 100), (64, 64, 32)):
        _run_binarised(cuda_device, rows, F, H)


def _run_binarised(device, rows, F, H):
    from scvae_amd import _lib
    lib = _lib.load()
    kind, heads = _lib.LIKELIHOOD_KINDS["bernoulli"]
    rng = np.random.default_rng(rows + F)
    d = np.maximum(rng.normal(0, 1, (rows, H)), 0)
    W = rng.normal(0, 0.3, (H, F))
    b = rng.normal(0, 0.3, F)
    t = (rng.random((rows, F)) < 0.2).astype(np.float64)
    gw = rng.normal(0, 1, rows)
    T = torch.from_numpy
    dt, Wt, bt = (T(a).requires_grad_(True) for a in (d, W, b))
    pre = dt @ Wt + bt
    ll_ref = torch.distributions.Bernoulli(logits=pre).log_prob(T(t)).sum(dim=1)
    (ll_ref * T(gw)).sum().backward()
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(device)
    dd_, td, gwd, Wd, bd = f32(d), f32(t), f32(gw), f32(W), f32(b)
    dWd, dbd = torch.full_like(Wd, 7.0), torch.full_like(bd, 7.0)
    ll = torch.full((rows,), 7.0, device=device)
    dd = torch.full((rows, H), 7.0, device=device)
    ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F),
                     dtype=torch.uint8, device=device)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for train in (0, 1):
        _lib.check(lib.scvae_decoder_fused(
            kind, train | ARITH["flag"], dd_.data_ptr(), rows, H, arr([Wd]), arr([bd]),
            arr([dWd]), arr([dbd]), F, td.data_ptr(), rows, gwd.data_ptr(), None,
            ll.data_ptr(), dd.data_ptr(), ws.data_ptr(), stream),
            "scvae_decoder_fused")
        torch.cuda.synchronize()
        want = ll_ref.detach().numpy()
        got = ll.cpu().double().numpy()
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-4
    for got, want, what in ((dd, dt.grad, "dd"), (dWd, Wt.grad, "dW"),
                            (dbd, bt.grad, "db")):
        want = want.numpy()
        err = np.abs(got.cpu().double().numpy() - want).max()
        assert err <= 3e-5 * (np.abs(want).max() + 1e-12), (what, err)


def test_repeated_targets_and_inline_lgamma(cuda_device):
    # rows = samples x cells (importance samples / GMVAE passes): row r uses t[r % cells]
    _run(cuda_device, "negative binomial", 96, 32, 100, 24, 0.3)
    _run(cuda_device, "poisson", 90, 30, 77, 24, 0.3, row_const=False)
    _run(cuda_device, "zero-inflated poisson", 60, 20, 77, 24, 0.3,
         row_const=False)


@pytest.mark.parametrize("name", list(lk.ELEMENTWISE_LIKELIHOODS))
def test_dd_through_xcd_local_atomics(cuda_device, name, head_arith):
    """``SCVAE_HEADS_DD_ATOMICS``: the strips' parts of ``dd`` added into eight
    XCD-local accumulators (fp32 atomics) instead of written as slabs -- against
    the fp64 reference like every other case (several row tiles and strips, a
    ragged last tile), and against the slab path: the same ``ll`` / ``dW`` /
    ``db`` bit for bit, ``dd`` to fp32 rounding of a differently ordered sum."""
    from scvae_amd import _lib
    _run(cuda_device, name, 300, 300, 500, 100, 0.1,
         extra_flags=_lib.HEADS_DD_ATOMICS)
    _run(cuda_device, name, 70, 35, 130, 20, 0.3,
         extra_flags=_lib.HEADS_DD_ATOMICS)
    if head_arith == "fp32":
        return          # (the fp32 kernels have no such store: the flag is ignored)
    lib = _lib.load()
    kind, heads = _lib.LIKELIHOOD_KINDS[name]
    P = len(heads)
    rows, F, H = 1000, 700, 100
    g = torch.Generator(device=cuda_device).manual_seed(11)
    d = torch.relu(torch.randn(rows, H, device=cuda_device, generator=g))
    W = [torch.randn(H, F, device=cuda_device, generator=g) * 0.1 for _ in range(P)]
    b = [torch.randn(F, device=cuda_device, generator=g) * 0.1 for _ in range(P)]
    t = torch.poisson(torch.full((rows, F), 2.0, device=cuda_device), generator=g)
    t = t * (torch.rand(rows, F, device=cuda_device, generator=g) < 0.1)
    gw = torch.randn(rows, device=cuda_device, generator=g)
    rc = torch.lgamma(t + 1).sum(dim=1)
    ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F),
                     dtype=torch.uint8, device=cuda_device)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for flags in (0, _lib.HEADS_DD_ATOMICS):
        dW = [torch.zeros_like(w) for w in W]
        db = [torch.zeros_like(v) for v in b]
        ll = torch.zeros(rows, device=cuda_device)
        dd = torch.zeros(rows, H, device=cuda_device)
        _lib.check(lib.scvae_decoder_fused(
            kind, 1 | _lib.HEADS_BF16X9 | flags, d.data_ptr(), rows, H, arr(W),
            arr(b), arr(dW), arr(db), F, t.data_ptr(), rows, gw.data_ptr(),
            rc.data_ptr(), ll.data_ptr(), dd.data_ptr(), ws.data_ptr(), stream),
            "scvae_decoder_fused")
        torch.cuda.synchronize()
        out[flags] = (ll, dd, dW, db)
    a, c = out[0], out[_lib.HEADS_DD_ATOMICS]
    assert torch.equal(a[0], c[0])
    for j in range(P):
        assert torch.equal(a[2][j], c[2][j]) and torch.equal(a[3][j], c[3][j])
    err = (a[1] - c[1]).abs().max().item()
    assert err <= 2e-6 * a[1].abs().max().item(), err


@pytest.mark.skipif(os.environ.get("SCVAE_FUSED_TEST_CHILD") == "1", reason="child process")
@pytest.mark.parametrize("schedule", ["3", "4"])
def test_edge_shapes_on_the_other_schedule(cuda_device, schedule):
    """Up to 128 rows a training launch takes the all-in-one-phase kernel,
    beyond them the producer / consumer one (bf16x9): the tile-edge and
    hidden-size cases again with each schedule forced for every row count
    (``SCVAE_D3_SCHEDULE``, read once per process)."""
    env = dict(os.environ, SCVAE_D3_SCHEDULE=schedule, SCVAE_FUSED_TEST_CHILD="1")
    out = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu",
         "-k", "bf16x9 and (tile_edges or test_hidden_sizes or many_row_tiles or repeated)"],
        env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
        capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]


@pytest.mark.parametrize("density", [0.3, 0.7, 1.0])
@pytest.mark.parametrize("name", ["negative binomial", "zero-inflated negative binomial"])
def test_dense_counts_through_the_queue(cuda_device, name, density):
    """The producer / consumer kernel queues the non-zeros of a wave densely and
    corrects them in passes of 64 (``lgamma(r + t) - lgamma(r)`` and the digamma
    term of dlog r, du:230-262): 5 % non-zeros fill a fraction of one pass --
    here 77, 180 and all 256 of a wave's elements per tile are non-zero (two to
    four passes, ragged last pass), row-replicated targets included; without
    the per-row constant the queue also carries ``lgamma(1 + t)``."""
    _run(cuda_device, name, 300, 300, 200, 100, density, seed=5)
    _run(cuda_device, name, 320, 160, 130, 36, density, seed=6, row_const=False)
