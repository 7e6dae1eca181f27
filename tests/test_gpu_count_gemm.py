"""The exact bf16-split kernels for the encoder input layer's two large products on
a count matrix (``scvae_count_gemm``: x W + b and x^T dA; mu:53-59 applied to
``x_train[idx].toarray()``, va:997-998) against the fp32 MFMA kernel
(``scvae_gemm``) and fp64 NumPy.

Bar: <= 1e-6 relative against the fp32-MFMA kernel,
where "relative" is to the output's largest magnitude; both kernels are also
held to the fp64 product.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _counts(rng, rows, cols, density, big=True):
    x = rng.poisson(3.0, size=(rows, cols)) * (rng.random((rows, cols)) < density)
    x = x.astype(np.float32)
    if big:   # counts that need the lo term: > 8 significant bits, up to 16
        k = max(1, rows * cols // 500)
        x.flat[rng.integers(0, x.size, k)] = rng.integers(
            256, 65536, k).astype(np.float32)
        x.flat[0] = 65535.0
        x.flat[-1] = 257.0
    return x


def _count_gemm(lib, mode, x, other, bias=None, relu=False):
    from scvae_amd import _lib
    rows, cols = x.shape
    N = other.shape[1]
    M = rows if mode == 0 else cols
    nbytes = lib.scvae_count_gemm_workspace_bytes(mode, rows, cols, N)
    assert nbytes >= 0
    ws = torch.empty(nbytes + 16, dtype=torch.uint8, device=x.device)
    out = torch.full((M, N), float("nan"), device=x.device)
    _lib.check(lib.scvae_count_gemm(
        mode, _p(x), x.stride(0), rows, cols, _p(other), other.stride(0), N,
        _p(bias), 1 if relu else 0, _p(out), N, _p(ws), nbytes, _stream()),
        "scvae_count_gemm")
    return out


def _fp32_gemm(lib, mode, x, other, bias=None, relu=False):
    from scvae_amd import _lib
    rows, cols = x.shape
    N = other.shape[1]
    M, K = (rows, cols) if mode == 0 else (cols, rows)
    nbytes = lib.scvae_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    out = torch.empty(M, N, device=x.device)
    _lib.check(lib.scvae_gemm(
        1 if mode == 1 else 0, 0, _p(x), _p(other), _p(bias), _p(out), M, N, K,
        x.stride(0), other.stride(0), N, 1 if relu else 0, 0, _p(ws), nbytes,
        _stream()), "scvae_gemm")
    return out


@pytest.mark.parametrize("rows,cols,N,density", [
    (100, 32738, 100, 0.05),     # the reference's default minibatch, cfg2-4 width
    (1024, 32738, 100, 0.05),
    (512, 27998, 100, 0.05),     # cfg5's gene count
    (37, 203, 24, 0.3),          # ragged everything
    (64, 64, 128, 0.5),          # the widest supported layer
    (5, 1000, 1, 0.2),
    (700, 33, 7, 0.4),
])
@pytest.mark.parametrize("mode", [0, 1])
def test_count_gemm_matches_fp32_mfma_and_fp64(cuda_device, rows, cols, N,
                                               density, mode):
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(rows * 7 + cols + N + mode)
    xh = _counts(rng, rows, cols, density)
    K = cols if mode == 0 else rows
    # weights / gradients spanning many binades, some tiny
    oh = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-12, 1, (K, N)))
          ).astype(np.float32)
    bh = rng.standard_normal(N).astype(np.float32) if mode == 0 else None
    x = torch.from_numpy(xh).to(cuda_device)
    other = torch.from_numpy(oh).to(cuda_device)
    bias = torch.from_numpy(bh).to(cuda_device) if bh is not None else None
    got = _count_gemm(lib, mode, x, other, bias)
    ref32 = _fp32_gemm(lib, mode, x, other, bias)
    torch.cuda.synchronize()
    want = (xh.astype(np.float64) @ oh.astype(np.float64) if mode == 0
            else xh.astype(np.float64).T @ oh.astype(np.float64))
    if bh is not None:
        want = want + bh.astype(np.float64)
    scale = np.abs(want).max()
    got, ref32 = got.cpu().numpy(), ref32.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref32).max() <= 1e-6 * scale, (
        np.abs(got - ref32).max() / scale)
    # both against fp64: the split path (exact products, the matrix core's fp32
    # accumulation of 16 products per instruction) stays in the fp32 path's class
    err_split = np.abs(got - want).max() / scale
    err_fp32 = np.abs(ref32 - want).max() / scale
    assert err_split <= max(4.0 * err_fp32, 5e-7), (err_split, err_fp32)


def test_count_gemm_is_exact_on_exactly_representable_products(cuda_device):
    """Small integer weights: every partial sum is an integer below 2^24, so
    fp32 accumulation is exact in any order -- the result must be the integer
    matrix product, bit for bit (hi/lo cut and three-term split included)."""
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    rows, cols, N = 96, 4100, 100
    xh = (rng.integers(0, 700, (rows, cols)) * (rng.random((rows, cols)) < 0.1)
          ).astype(np.float32)
    for mode in (0, 1):
        K = cols if mode == 0 else rows
        oh = rng.integers(-3, 4, (K, N)).astype(np.float32)
        got = _count_gemm(lib, mode, torch.from_numpy(xh).to(cuda_device),
                          torch.from_numpy(oh).to(cuda_device))
        torch.cuda.synchronize()
        want = (xh.astype(np.int64) @ oh.astype(np.int64) if mode == 0
                else xh.astype(np.int64).T @ oh.astype(np.int64))
        assert np.abs(want).max() < 2 ** 24
        assert np.array_equal(got.cpu().numpy().astype(np.int64), want)


def test_bias_and_relu_epilogue(cuda_device):
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for rows, cols in ((40, 300), (40, 20000)):     # direct store / split-K reduce
        xh = _counts(rng, rows, cols, 0.2, big=False)
        oh = rng.standard_normal((cols, 50)).astype(np.float32) * 0.05
        bh = rng.standard_normal(50).astype(np.float32)
        got = _count_gemm(lib, 0, torch.from_numpy(xh).to(cuda_device),
                          torch.from_numpy(oh).to(cuda_device),
                          torch.from_numpy(bh).to(cuda_device), relu=True)
        torch.cuda.synchronize()
        want = np.maximum(xh.astype(np.float64) @ oh.astype(np.float64) + bh,
                          0.0)
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-5 * np.abs(
            want).max()
        assert (got >= 0).all()


def test_count_precondition_check(cuda_device):
    from scvae_amd import _lib
    lib = _lib.load()

    def bad(values):
        v = torch.tensor(values, dtype=torch.float32, device=cuda_device)
        flag = torch.full((1,), 7, dtype=torch.int32, device=cuda_device)
        _lib.check(lib.scvae_check_counts(_p(v), v.numel(), _p(flag),
                                          _stream()), "scvae_check_counts")
        return int(flag.item())
    assert bad([0.0, 1.0, 255.0, 256.0, 65535.0]) == 0
    assert bad([0.0, 1.5]) == 1
    assert bad([3.0, -1.0]) == 1
    assert bad([65536.0]) == 1
    assert bad([float("nan")]) == 1
    assert bad([]) == 0
    # DeviceCSR runs it once at upload
    import scipy.sparse as sp
    from scvae_amd.minibatch import DeviceCSR
    counts = sp.random(50, 40, density=0.2, format="csr", random_state=1,
                       data_rvs=lambda n: np.random.default_rng(0).integers(
                           1, 900, n).astype(np.float64))
    assert DeviceCSR.from_scipy(counts, cuda_device).integer_counts
    logged = counts.copy()
    logged.data = np.log1p(logged.data)
    assert not DeviceCSR.from_scipy(logged, cuda_device).integer_counts


@pytest.mark.usefixtures("bit_repeatable")
@pytest.mark.parametrize("model_type", ["VAE", "GMVAE"])
def test_training_step_with_and_without_the_count_kernels(cuda_device,
                                                          model_type):
    """The same training step with ``x_counts`` on the exact bf16-split kernels
    and on the fp32 MFMA kernels: ELBO, per-cell log-likelihood and every
    gradient agree to fp32 rounding."""
    from scvae_amd.engine import Engine
    F, L, H, B, K = 5000, 10, (100, 100), 200, 4
    eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                 model_type=model_type, n_clusters=K, device=cuda_device,
                 seed=1)
    rng = np.random.default_rng(11)
    x = torch.from_numpy(_counts(rng, B, F, 0.05)).to(cuda_device)
    shape = (1, B, L) if model_type == "VAE" else (K, 1, B, L)
    eps = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(
        cuda_device)
    rows = B if model_type == "VAE" else K * B
    results = []
    for enabled in (True, False):
        eng.set_count_gemm(enabled, always=True)   # (B = 200 is below the auto threshold)
        ll = torch.zeros(rows, device=cuda_device)
        scalars = eng.step(x, x, eps=eps, training=True, x_counts=True,
                           outputs={"log_p_x_given_z": ll}).clone()
        torch.cuda.synchronize()
        results.append((scalars.cpu().numpy(), ll.cpu().numpy(),
                        eng.grads.clone()))
    eng.set_count_gemm(True)
    (s_c, ll_c, g_c), (s_f, ll_f, g_f) = results
    assert abs(s_c[0] - s_f[0]) <= 2e-6 * abs(s_f[0])
    assert np.abs(ll_c - ll_f).max() <= 1e-5 * np.abs(ll_f).max()
    for name, (offset, shape) in eng.param_table.items():
        n = int(np.prod(shape))
        a, b = g_c[offset:offset + n], g_f[offset:offset + n]
        scale = b.abs().max().item()
        # (fp32 rounding differences of the first layer's output, carried
        # through batch norm and the rest of the step; the same bound as the
        # fused-vs-unfused comparison at full size)
        assert (a - b).abs().max().item() <= 1e-4 * scale + 1e-12, name
    # without the caller's word the plan never takes the count kernels
    plain = eng.step(x, x, eps=eps, training=True).clone()
    torch.cuda.synchronize()
    assert torch.equal(eng.grads, g_f)
    assert np.array_equal(plain.cpu().numpy()[:5], s_f[:5])


@pytest.mark.usefixtures("bit_repeatable")
@pytest.mark.parametrize("likelihood", ["negative binomial",
                                        "zero-inflated negative binomial",
                                        "poisson"])
def test_uint16_minibatch_is_the_fp32_step_bit_for_bit(cuda_device, likelihood):
    """The minibatch densified as uint16 counts (``scvae_csr_densify_u16``,
    ``scvae_step_args.counts_u16``): the input layer's two products and the
    fused likelihood heads read half the bytes and do the same arithmetic on
    the same values -- scalars, per-cell log-likelihood, every gradient and the
    moving statistics carry identical bits; training and evaluation."""
    import scipy.sparse as sp
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import DeviceCSR
    F, L, H, B = 2101, 10, (100, 100), 300     # F: odd pitch for the fp32 batch
    rng = np.random.default_rng(17)
    counts = sp.csr_matrix(_counts(rng, 700, F, 0.05))
    csr = DeviceCSR.from_scipy(counts, cuda_device)
    assert csr.integer_counts
    rows = torch.from_numpy(rng.permutation(700)[:B]).to(cuda_device)
    x32 = csr.gather_dense(rows)
    rc16 = torch.zeros(B, device=cuda_device)
    x16 = csr.gather_counts_u16(rows, row_const_out=rc16)
    assert x16.dtype == torch.uint16 and x16.shape == (B, csr.u16_pitch)
    assert torch.equal(x16[:, :F].to(torch.float32), x32)
    assert int(x16[:, F:].to(torch.int32).abs().sum()) == 0
    eps = torch.from_numpy(rng.standard_normal((2, B, L)).astype(np.float32)
                           ).to(cuda_device)
    results = []
    for u16 in (False, True):
        eng = Engine(F, L, H, likelihood, batch_norm=True, device=cuda_device,
                     seed=1)
        eng.set_count_gemm(True, always=True)
        assert eng.accepts_counts_u16(B, True) and eng.accepts_counts_u16(B, False)
        x = x16 if u16 else x32
        out = []
        for _ in range(2):
            ll = torch.zeros(2 * B, device=cuda_device)
            s = eng.step(x, x, eps=eps, training=True, n_iw=2, row_const=rc16,
                         x_counts=True, outputs={"log_p_x_given_z": ll}).clone()
            eng.adam_step(1e-3)
            out += [s, ll.clone()]
        ev = eng.step(x, x, eps=eps, training=False, n_iw=2, row_const=rc16,
                      x_counts=True).clone()
        det = eng.step(x, x, training=False, deterministic_z=True,
                       row_const=rc16, x_counts=True).clone()
        torch.cuda.synchronize()
        results.append([t.cpu() for t in out + [ev, det, eng.grads, eng.moving,
                                                eng.params]])
    for a, b in zip(*results):
        assert torch.equal(a, b)


def test_uint16_minibatch_with_head_dropout(cuda_device):
    """Hidden-layer dropout reaches the likelihood heads; in a step without
    importance weighting the bf16x9 head kernel takes it, and with it the
    uint16 minibatch: identical bits to the fp32 batch; with importance
    weighting (a separate forward pass of the heads) the plan says no."""
    import scipy.sparse as sp
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import DeviceCSR
    F, L, H, B = 1500, 6, (100, 48), 200
    rng = np.random.default_rng(23)
    csr = DeviceCSR.from_scipy(sp.csr_matrix(_counts(rng, 400, F, 0.05)),
                               cuda_device)
    rows = torch.from_numpy(rng.permutation(400)[:B]).to(cuda_device)
    x32 = csr.gather_dense(rows)
    rc16 = torch.zeros(B, device=cuda_device)
    x16 = csr.gather_counts_u16(rows, row_const_out=rc16)
    eps = torch.from_numpy(rng.standard_normal((1, B, L)).astype(np.float32)
                           ).to(cuda_device)
    results = []
    for u16 in (False, True):
        eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                     device=cuda_device, seed=1,
                     dropout_keep_probabilities=(0.8, 0.0, 0.0))
        eng.set_count_gemm(True, always=True)
        assert eng.accepts_counts_u16(B, True, n_iw=1)
        assert not eng.accepts_counts_u16(B, True)
        assert not eng.accepts_counts_u16(B, True, n_iw=2)
        x = x16 if u16 else x32
        out = []
        for i in range(2):
            ll = torch.zeros(B, device=cuda_device)
            s = eng.step(x, x, eps=eps, training=True, row_const=rc16,
                         x_counts=True, dropout_seed=77 + i,
                         outputs={"log_p_x_given_z": ll}).clone()
            eng.adam_step(1e-3)
            out += [s, ll.clone()]
        torch.cuda.synchronize()
        results.append([t.cpu() for t in out + [eng.grads, eng.moving,
                                                eng.params]])
    for a, b in zip(*results):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        eng.step(x16, x16, eps=eps.repeat(2, 1, 1), training=True, n_iw=2,
                 row_const=rc16, x_counts=True, dropout_seed=1)


def test_uint16_minibatch_is_refused_where_it_does_not_apply(cuda_device):
    from scvae_amd.engine import Engine
    F, L, B = 600, 6, 64
    x16 = torch.zeros(B, 640, dtype=torch.uint16, device=cuda_device)
    eps = torch.zeros(1, B, L, device=cuda_device)
    # a GMVAE plan, and a VAE whose minibatch is too small for the count kernels
    gm = Engine(F, L, (32,), "negative binomial", batch_norm=True,
                model_type="GMVAE", n_clusters=3, device=cuda_device, seed=1)
    assert not gm.accepts_counts_u16(B, True)
    vae = Engine(F, L, (32,), "negative binomial", batch_norm=True,
                 device=cuda_device, seed=1)
    assert not vae.accepts_counts_u16(B, True)
    with pytest.raises(RuntimeError):
        vae.step(x16, x16, eps=eps, training=True)
    vae.set_count_gemm(True, always=True)
    assert vae.accepts_counts_u16(B, True)
    # dropout on the input layer reads the fp32 batch
    drop = Engine(F, L, (32,), "negative binomial", batch_norm=True,
                  dropout_keep_probabilities=(0, 0.9, 0, 0), device=cuda_device,
                  seed=1)
    drop.set_count_gemm(True, always=True)
    assert not drop.accepts_counts_u16(B, True)
    assert drop.accepts_counts_u16(B, False)


@pytest.mark.usefixtures("bit_repeatable")
def test_tail_minibatch_fits_the_reserved_count_workspace(cuda_device):
    """A plan bound for 1300 cells runs every smaller minibatch on the uint16
    path: the forward count kernel takes MORE split-K slabs for fewer row
    tiles (1259..1280 rows need more workspace than 1300), which the plan's
    reservation must cover (round-2 advisor finding)."""
    from scvae_amd import _lib
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import synthetic_count_matrix
    lib = _lib.load()
    F, L, H, BMAX = 20000, 8, (100,), 1300
    need = [lib.scvae_count_gemm_workspace_bytes(0, r, F, 100)
            for r in (BMAX, 1280, 1270)]
    assert max(need[1:]) > need[0]          # the non-monotone case is real
    matrix, _ = synthetic_count_matrix(BMAX, F, density=0.05, seed=9,
                                       device=cuda_device)
    eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                 device=cuda_device, seed=1)
    eng.set_count_gemm(True, always=True)   # (also below the size where they pay)
    eng.reserve(BMAX, 1)
    for cells in (BMAX, 1280, 1270, 1024, 800):
        assert eng.accepts_counts_u16(cells, True)
        rows = torch.arange(cells, device=cuda_device)
        rc = torch.zeros(cells, device=cuda_device)
        x16 = matrix.gather_counts_u16(rows, row_const_out=rc)
        x32 = matrix.gather_dense(rows)
        eps = torch.randn(1, cells, L, device=cuda_device)
        s16 = eng.step(x16, x16, eps=eps, row_const=rc, training=True,
                       x_counts=True).clone()
        g16 = eng.grads.clone()
        s32 = eng.step(x32, x32, eps=eps, row_const=rc, training=True,
                       x_counts=True).clone()
        torch.cuda.synchronize()
        assert torch.equal(s16, s32) and torch.equal(g16, eng.grads)


@pytest.mark.usefixtures("bit_repeatable")
@pytest.mark.parametrize("likelihood", ["negative binomial",
                                        "zero-inflated negative binomial"])
def test_uint16_minibatch_gmvae_is_the_fp32_step_bit_for_bit(cuda_device,
                                                            likelihood):
    """GMVAE: the K stacked decoder passes read their targets from the uint16
    minibatch and both layers that see x (q(y|x), q(z|x,y): x W[:F] once) take
    the uint16 count kernels -- same bits as the fp32 batch; training (with the
    optimiser in between) and evaluation."""
    import scipy.sparse as sp
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import DeviceCSR
    F, L, H, B, K = 2101, 10, (100, 100), 160, 4
    rng = np.random.default_rng(23)
    counts = sp.csr_matrix(_counts(rng, 400, F, 0.05))
    csr = DeviceCSR.from_scipy(counts, cuda_device)
    rows = torch.from_numpy(rng.permutation(400)[:B]).to(cuda_device)
    x32 = csr.gather_dense(rows)
    rc = torch.zeros(B, device=cuda_device)
    x16 = csr.gather_counts_u16(rows, row_const_out=rc)
    eps = torch.from_numpy(rng.standard_normal((K, 1, B, L)).astype(np.float32)
                           ).to(cuda_device)
    results = []
    for u16 in (False, True):
        eng = Engine(F, L, H, likelihood, batch_norm=True, model_type="GMVAE",
                     n_clusters=K, device=cuda_device, seed=1)
        eng.set_count_gemm(True, always=True)
        assert eng.accepts_counts_u16(B, True) and eng.accepts_counts_u16(B, False)
        x = x16 if u16 else x32
        out = []
        for _ in range(2):
            ll = torch.zeros(K * B, device=cuda_device)
            s = eng.step(x, x, eps=eps, training=True, row_const=rc,
                         x_counts=True, outputs={"log_p_x_given_z": ll}).clone()
            eng.adam_step(1e-3)
            out += [s, ll.clone()]
        ev = eng.step(x, x, eps=eps, training=False, row_const=rc,
                      x_counts=True).clone()
        torch.cuda.synchronize()
        results.append([t.cpu() for t in out + [ev, eng.grads, eng.moving,
                                                eng.params]])
    for i, (a, b) in enumerate(zip(*results)):
        assert torch.equal(a, b), i
