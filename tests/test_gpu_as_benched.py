"""The code path `bench.py` times, under test at its own size (round-2 review,
"what's weak" 2): the uint16 minibatch of 4096 cells x 32 738 genes through

* ``scvae_count_gemm_u16`` mode 0 (x W1 + b) and mode 1 (dW1 = x^T dA) against
  the fp32 MFMA kernel and the fp64 product, at (4096, 32 738) and
  (512, 27 998) -- cfg2-4's and cfg5's gene counts;
* ``scvae_decoder_fused_u16`` (one launch: heads + NB likelihood + backward)
  against the unfused kernels (``scvae_gemm`` + ``scvae_loglik_bwd``) at
  4096 x 32 738;
* a whole training step on the uint16 minibatch against the same step on the
  fp32 batch, bit for bit (scalars, per-cell log-likelihood, every gradient,
  the moving statistics), and against the fp64 oracle at B = 4096.

Reference ops: mu:53-59 on ``x_train[idx].toarray()`` (va:997-998),
va:2466-2505 / 2583-2590 (heads + log_prob + sum over genes), va:2717-2770.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import models as om
from _parity import (LL_ATOL, LL_RTOL, close_elementwise, close_maxnorm,
                     close_scalar)

pytestmark = pytest.mark.gpu

H = (100, 100)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _minibatches(device, cells, features, seed):
    """The benchmark's generator through the production fetch: the same rows
    as fp32 [cells, F] and as uint16 [cells, pitch], and the lgamma row term."""
    from scvae_amd.minibatch import synthetic_count_matrix
    matrix, _ = synthetic_count_matrix(cells, features, density=0.05,
                                       seed=seed, device=device)
    assert matrix.integer_counts
    rows = torch.arange(cells, device=device)
    rc = torch.empty(cells, device=device)
    x32 = matrix.gather_dense(rows, row_const_out=rc)
    rc16 = torch.empty(cells, device=device)
    x16 = matrix.gather_counts_u16(rows, row_const_out=rc16)
    assert torch.equal(rc, rc16)
    return matrix, x32, x16, rc


@pytest.mark.parametrize("rows,cols", [(4096, 32738), (512, 27998)])
@pytest.mark.parametrize("mode", [0, 1])
def test_count_gemm_u16_at_benchmark_size(cuda_device, rows, cols, mode):
    from scvae_amd import _lib
    lib = _lib.load()
    N = 100
    matrix, x32, x16, _ = _minibatches(cuda_device, rows, cols, seed=60 + mode)
    assert torch.equal(x16[:, :cols].to(torch.float32), x32)
    rng = np.random.default_rng(rows + mode)
    K = cols if mode == 0 else rows
    oh = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-9, 0, (K, N)))
          ).astype(np.float32)
    bh = rng.standard_normal(N).astype(np.float32) if mode == 0 else None
    other = torch.from_numpy(oh).to(cuda_device)
    bias = torch.from_numpy(bh).to(cuda_device) if bh is not None else None
    M = rows if mode == 0 else cols

    nbytes = lib.scvae_count_gemm_workspace_bytes(mode, rows, cols, N)
    assert nbytes >= 0
    ws = torch.empty(nbytes + 16, dtype=torch.uint8, device=cuda_device)
    got16 = torch.full((M, N), float("nan"), device=cuda_device)
    _lib.check(lib.scvae_count_gemm_u16(
        mode, _p(x16), x16.stride(0), rows, cols, _p(other), N, N, _p(bias), 0,
        _p(got16), N, _p(ws), nbytes, _stream()), "scvae_count_gemm_u16")
    got32 = torch.full((M, N), float("nan"), device=cuda_device)
    _lib.check(lib.scvae_count_gemm(
        mode, _p(x32), x32.stride(0), rows, cols, _p(other), N, N, _p(bias), 0,
        _p(got32), N, _p(ws), nbytes, _stream()), "scvae_count_gemm")
    nb2 = lib.scvae_gemm_workspace_bytes(M, N, K)
    ws2 = torch.empty(max(nb2, 16), dtype=torch.uint8, device=cuda_device)
    ref32 = torch.empty(M, N, device=cuda_device)
    _lib.check(lib.scvae_gemm(
        1 if mode == 1 else 0, 0, _p(x32), _p(other), _p(bias), _p(ref32), M, N,
        K, x32.stride(0), N, N, 0, 0, _p(ws2), nb2, _stream()), "scvae_gemm")
    torch.cuda.synchronize()
    # the uint16 kernel does the fp32-batch kernel's arithmetic on the same values
    assert torch.equal(got16, got32)
    # fp64 product from the CSR matrix itself (scipy on the host)
    import scipy.sparse as sp
    csr = sp.csr_matrix((matrix.values.cpu().numpy().astype(np.float64),
                         matrix.indices.cpu().numpy(),
                         matrix.indptr.cpu().numpy()), shape=(rows, cols))
    want = (csr @ oh.astype(np.float64) if mode == 0
            else csr.T @ oh.astype(np.float64))
    want = np.asarray(want)
    if bh is not None:
        want = want + bh.astype(np.float64)
    scale = np.abs(want).max()
    got, ref = got16.cpu().numpy(), ref32.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-6 * scale, np.abs(got - ref).max() / scale
    err_split = np.abs(got - want).max() / scale
    err_fp32 = np.abs(ref - want).max() / scale
    assert err_split <= max(4.0 * err_fp32, 5e-7), (err_split, err_fp32)


def test_decoder_fused_u16_against_the_unfused_kernels(cuda_device):
    """4096 x 32 738, NB, H = 100: the launch the benchmark's roofline figure is
    quoted on (uint16 targets, the dealt schedule with the 16-wide h remainder)
    against GEMM + element-wise likelihood kernels on the fp32 batch."""
    from scvae_amd import _lib
    lib = _lib.load()
    rows, F, Hd = 4096, 32738, 100
    kind, heads = _lib.LIKELIHOOD_KINDS["negative binomial"]
    P = len(heads)
    _, x32, x16, rc = _minibatches(cuda_device, rows, F, seed=62)
    g = torch.Generator(device=cuda_device).manual_seed(5)
    d = torch.relu(torch.randn(rows, Hd, generator=g, device=cuda_device))
    W = [torch.randn(Hd, F, generator=g, device=cuda_device) * 0.1
         for _ in range(P)]
    b = [torch.randn(F, generator=g, device=cuda_device) * 0.1
         for _ in range(P)]
    gw = -torch.rand(rows, generator=g, device=cuda_device) / rows
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])

    # ---- fused, uint16 targets (and fp32 targets: identical bits) ----
    ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, Hd, F),
                     dtype=torch.uint8, device=cuda_device)
    out = {}
    for name in ("u16", "f32"):
        dW = [torch.full_like(w, 7.0) for w in W]
        db = [torch.full_like(v, 7.0) for v in b]
        ll = torch.full((rows,), 7.0, device=cuda_device)
        dd = torch.full((rows, Hd), 7.0, device=cuda_device)
        if name == "u16":
            _lib.check(lib.scvae_decoder_fused_u16(
                kind, 1, _p(d), rows, Hd, arr(W), arr(b), arr(dW), arr(db), F,
                _p(x16), x16.stride(0), rows, _p(gw), _p(rc), _p(ll), _p(dd),
                _p(ws), _stream()), "scvae_decoder_fused_u16")
        else:
            _lib.check(lib.scvae_decoder_fused(
                kind, 1, _p(d), rows, Hd, arr(W), arr(b), arr(dW), arr(db), F,
                _p(x32), rows, _p(gw), _p(rc), _p(ll), _p(dd), _p(ws),
                _stream()), "scvae_decoder_fused")
        torch.cuda.synchronize()
        out[name] = (ll, dd, dW, db)
    for a, c in zip(out["u16"][:2], out["f32"][:2]):
        assert torch.equal(a, c)
    for j in range(P):
        assert torch.equal(out["u16"][2][j], out["f32"][2][j])
        assert torch.equal(out["u16"][3][j], out["f32"][3][j])

    # ---- unfused: pre_j = d W_j + b_j, G_j in place, dW_j = d^T G_j, dd = sum G_j W_j^T ----
    def gemm(ta, tb, A, B_, bias, C, M, N, K, lda, ldb, acc=0):
        nb = lib.scvae_gemm_workspace_bytes(M, N, K)
        w2 = torch.empty(max(nb, 16), dtype=torch.uint8, device=cuda_device)
        _lib.check(lib.scvae_gemm(ta, tb, _p(A), _p(B_), _p(bias), _p(C), M, N,
                                  K, lda, ldb, N, 0, acc, _p(w2), nb,
                                  _stream()), "scvae_gemm")
        torch.cuda.synchronize()
    pre = [torch.empty(rows, F, device=cuda_device) for _ in range(P)]
    for j in range(P):
        gemm(0, 0, d, W[j], b[j], pre[j], rows, F, Hd, Hd, F)
    ll_u = torch.empty(rows, device=cuda_device)
    _lib.check(lib.scvae_loglik_bwd(kind, _p(x32), arr(pre), _p(gw), _p(rc),
                                    _p(ll_u), rows, rows, F, _stream()),
               "scvae_loglik_bwd")
    dd_u = torch.zeros(rows, Hd, device=cuda_device)
    ll_f, dd_f, dW_f, db_f = out["u16"]
    for j in range(P):
        dW_u = torch.empty(Hd, F, device=cuda_device)
        gemm(1, 0, d, pre[j], None, dW_u, Hd, F, rows, Hd, F)
        gemm(0, 1, pre[j], W[j], None, dd_u, rows, Hd, F, F, F, acc=j > 0)
        close_maxnorm(dW_f[j], dW_u, rtol=2e-5, what="dW%d" % j)
        close_maxnorm(db_f[j], pre[j].double().sum(dim=0), rtol=2e-5,
                      what="db%d" % j)
    close_elementwise(ll_f, ll_u, rtol=2e-6, atol=1e-3,
                      what="per-cell log-likelihood")
    close_maxnorm(dd_f, dd_u, rtol=2e-5, what="dd")


@pytest.fixture(scope="module")
def benchmark_step(cuda_device):
    """Engine at the benchmark's shape and one minibatch in both encodings."""
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import philox_normal
    B, F, L = 4096, 32738, 25
    _, x32, x16, rc = _minibatches(cuda_device, B, F, seed=63)
    eps = torch.empty(1, B, L, device=cuda_device)
    philox_normal(eps[0], row_offset=0, seed=3, stream_id=0)

    def engine():
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
        return eng
    return engine, x32, x16, rc, eps


def test_u16_training_step_is_the_fp32_step_at_benchmark_size(benchmark_step):
    """What `bench.py` runs (uint16 minibatch, 4096 x 32 738) against the fp32
    batch: two training steps with the optimiser in between and an evaluation
    step -- scalars, per-cell log-likelihood, gradients, moving statistics and
    updated weights carry identical bits."""
    engine, x32, x16, rc, eps = benchmark_step
    B = x32.shape[0]
    results = []
    for x in (x32, x16):
        eng = engine()
        # (bit-for-bit: the decoder gradient through the fixed-order slabs; the plan's default
        #  -- fp32 atomics -- is what the oracle tests below run)
        eng.set_dd_atomics(False)
        if x.dtype == torch.uint16:
            assert eng.accepts_counts_u16(B, True)
            assert eng.accepts_counts_u16(B, False)
        out = []
        for _ in range(2):
            ll = torch.zeros(B, device=x.device)
            s = eng.step(x, x, eps=eps, row_const=rc, training=True,
                         x_counts=True,
                         outputs={"log_p_x_given_z": ll}).clone()
            out += [s, ll.clone(), eng.grads.clone()]
            eng.adam_step(1e-4)
        ev = eng.step(x, x, eps=eps, row_const=rc, training=False,
                      x_counts=True).clone()
        torch.cuda.synchronize()
        results.append([t.cpu() for t in out + [ev, eng.moving, eng.params]])
    for i, (a, b) in enumerate(zip(*results)):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("arith", ["bf16x9", "bf16x6"])
def test_u16_training_step_against_the_oracle_at_benchmark_size(
        benchmark_step, arith):
    """One training step of the benchmark (B = 4096, uint16 minibatch, the
    producer / consumer head kernel with the decoder gradient through atomics,
    the count kernels reading two genes per lane) against the fp64 oracle with
    autograd on the host -- under the head arithmetic `bench.py` times (the
    exact nine-term split) and under the six-term option, same tolerances."""
    engine, x32, x16, rc, eps = benchmark_step
    B, F = x32.shape
    L = eps.shape[-1]
    eng = engine()
    eng.set_head_arith(arith)
    assert eng.dd_atomics          # the plan's default, as bench.py and `scvae train` run it
    assert eng.head_arith == arith
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood="negative binomial")
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    ll = torch.zeros(B, device=x16.device)
    klz = torch.zeros(L, device=x16.device)
    sc = eng.step(x16, x16, eps=eps, row_const=rc, training=True,
                  x_counts=True, outputs={"log_p_x_given_z": ll,
                                          "kl_neurons": klz}).cpu().numpy()
    dev_grads = {k: v.detach().cpu().double()
                 for k, v in eng.named_gradients().items()}
    torch.cuda.synchronize()
    xh = x32.cpu().double()
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, xh, xh, eps.cpu().double(),
                                 True, 1.0, {}), params)
    close_scalar(sc[0], out["lower_bound"], what="lower_bound")
    close_scalar(sc[2], out["reconstruction_error"],
                 what="reconstruction_error")
    close_scalar(sc[3], out["kl_divergence"], what="kl_divergence")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell log-likelihood")
    close_elementwise(klz, out["kl_divergence_neurons"], rtol=1e-4, atol=1e-6,
                      what="kl per latent unit")
    for name, g in dev_grads.items():
        if name.endswith("DENSE/biases") and ("ENCODER/" in name
                                              or "DECODER/" in name):
            assert g.abs().max().item() == 0.0, name
            continue
        close_maxnorm(g, grads[name], rtol=2e-4, what="grad " + name)


@pytest.mark.parametrize("config,likelihood,features", [
    ("cfg4", "negative binomial", 32738),
    ("cfg5", "zero-inflated negative binomial", 27998),
])
def test_gmvae_training_step_against_the_oracle_as_benched(
        cuda_device, config, likelihood, features):
    """The GMVAE steps `bench.py` times (`other_workloads.cfg4_* / cfg5_*`):
    K = 20 passes x 512 cells = 10 240 stacked rows through the producer /
    consumer head kernel (decoder gradient through atomics, the plan's
    default), the hidden layers on the tile-chain groups, the minibatch in the
    encoding the bench's ``Workload`` picks -- against the fp64 oracle with
    autograd on the host (gm:3223-3434): ELBO terms, per-cell and per-cluster
    log-likelihood per element, q(y|x) logits, every gradient."""
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import philox_normal_blocks
    B, F, L, K = 512, features, 100, 20
    matrix, x32, x16, rc = _minibatches(cuda_device, B, F, seed=64)
    eng = Engine(F, L, H, likelihood, batch_norm=True, model_type="GMVAE",
                 n_clusters=K, device=cuda_device, seed=0)
    g = torch.Generator().manual_seed(3)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    for name, m in eng.named_moving_statistics().items():
        if name.endswith("moving_mean"):
            m.copy_(torch.randn(m.shape, generator=g) * 0.2)
        else:
            m.copy_(torch.rand(m.shape, generator=g) + 0.5)
    eng.reserve(B, 1)
    assert eng.dd_atomics
    assert eng.head_arith == "bf16x9"
    assert eng.uses_tile_chain(B, 1)
    # exactly bench.py's choice (Workload.__init__)
    u16 = bool(matrix.integer_counts and eng.accepts_counts_u16(B, True))
    x = x16 if u16 else x32
    eps = torch.empty(K, B, L, device=cuda_device)
    philox_normal_blocks(eps, block_stride=B, row_offset=0, seed=1, stream_id=7)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood=likelihood, n_clusters=K)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    ll = torch.zeros(K * B, device=cuda_device)
    logits = torch.zeros(B, K, device=cuda_device)
    zmean = torch.zeros(B, L, device=cuda_device)
    sc = eng.step(x, x, eps=eps, row_const=rc, training=True, x_counts=True,
                  outputs={"log_p_x_given_z": ll, "q_y_logits": logits,
                           "q_z_mean": zmean}).cpu().numpy()
    dev_grads = {k: v.detach().cpu().double()
                 for k, v in eng.named_gradients().items()}
    torch.cuda.synchronize()
    assert np.isfinite(sc[:5]).all()
    xh = x32.cpu().double()
    out, grads = om.gradients(
        lambda p: om.gmvae_forward(
            cfg, p, moving, xh, xh, eps.cpu().double().reshape(K, 1, B, L),
            True, 1.0, {}), params)
    close_scalar(sc[0], out["lower_bound"], what="lower_bound")
    close_scalar(sc[2], out["reconstruction_error"],
                 what="reconstruction_error")
    close_scalar(sc[3], out["kl_divergence_z"], what="kl_divergence_z")
    close_scalar(sc[4], out["kl_divergence_y"], rtol=2e-4, atol=1e-6,
                 what="kl_divergence_y")
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what="per-cell log-likelihood")
    close_elementwise(logits, out["q_y_logits"], rtol=1e-4, atol=1e-5,
                      what="q_y_logits")
    close_elementwise(zmean, out["z_mean"], rtol=1e-4, atol=1e-5,
                      what="z_mean")
    for name, gr in dev_grads.items():
        if name.endswith("DENSE/biases") and "/LAYER_" in name:
            assert gr.abs().max().item() == 0.0, name   # cancelled by batch norm
            continue
        got, want = gr, grads[name]
        if name == "Z/Q/ENCODER/LAYER_1/DENSE/weights":
            # the one-hot rows W[F+k] are cancelled by the per-pass batch norm
            assert got[F:].abs().max().item() < 1e-5
            got, want = got[:F], want[:F]
        # (Y/: differences of per-cluster log-likelihoods of order 2e4, see
        #  tests/test_gpu_baseline_configs.py)
        rtol = 2e-3 if name.startswith("Y/") else 5e-4
        close_maxnorm(got, want, rtol=rtol, what="grad " + name)


def test_six_term_heads_are_fp32_class(cuda_device):
    """``SCVAE_HEADS_BF16X6`` (nine terms without a2 b3, a3 b2, a3 b3) against
    the fp64 products at 4096 rows: its error next to the exact split's and
    next to the fp32 matrix cores' on the same operands -- the option may not
    be worse than twice the plain fp32 kernel (both round every accumulation
    to fp32; the six-term product adds <= 2^-23 of |a||b| per term)."""
    from scvae_amd import _lib
    lib = _lib.load()
    rows, F, Hd = 4096, 1500, 100
    kind, heads = _lib.LIKELIHOOD_KINDS["negative binomial"]
    P = len(heads)
    g = torch.Generator(device=cuda_device).manual_seed(11)
    d = torch.relu(torch.randn(rows, Hd, generator=g, device=cuda_device))
    W = [torch.randn(Hd, F, generator=g, device=cuda_device) * 0.1 for _ in range(P)]
    b = [torch.randn(F, generator=g, device=cuda_device) * 0.1 for _ in range(P)]
    t = torch.poisson(torch.full((rows, F), 3.0, device=cuda_device), generator=g)
    t = t * (torch.rand(rows, F, device=cuda_device, generator=g) < 0.1)
    gw = -torch.rand(rows, generator=g, device=cuda_device) / rows
    rc = torch.lgamma(t + 1).sum(dim=1)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
    ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, Hd, F),
                     dtype=torch.uint8, device=cuda_device)
    # fp64: pre-activations, likelihood and gradients through torch autograd
    d64 = d.double().requires_grad_(True)
    W64 = [w.double().requires_grad_(True) for w in W]
    b64 = [v.double().requires_grad_(True) for v in b]
    a0 = d64 @ W64[0] + b64[0]
    a1 = d64 @ W64[1] + b64[1]
    eps_ = torch.finfo(torch.float32).tiny
    p = torch.sigmoid(a0).clamp(eps_, 1 - eps_)
    r = torch.exp(a1.clamp(-10, 10))
    t64 = t.double()
    lp = (torch.lgamma(r + t64) - torch.lgamma(r) - torch.lgamma(t64 + 1)
          + t64 * torch.log(p) + r * torch.log1p(-p))
    ll64 = lp.sum(dim=1)
    (ll64 * gw.double()).sum().backward()
    ref = {"ll": ll64.detach(), "dd": d64.grad, "dW0": W64[0].grad,
           "dW1": W64[1].grad, "db1": b64[1].grad}
    errs = {}
    for arith in ("fp32", "bf16x9", "bf16x6"):
        dW = [torch.zeros_like(w) for w in W]
        db = [torch.zeros_like(v) for v in b]
        ll = torch.zeros(rows, device=cuda_device)
        dd = torch.zeros(rows, Hd, device=cuda_device)
        _lib.check(lib.scvae_decoder_fused(
            kind, 1 | _lib.HEAD_ARITH_FLAGS[arith], _p(d), rows, Hd, arr(W), arr(b),
            arr(dW), arr(db), F, _p(t), rows, _p(gw), _p(rc), _p(ll), _p(dd), _p(ws),
            _stream()), "scvae_decoder_fused")
        torch.cuda.synchronize()
        got = {"ll": ll, "dd": dd, "dW0": dW[0], "dW1": dW[1], "db1": db[1]}
        errs[arith] = {k: ((got[k].double() - ref[k]).abs().max()
                           / ref[k].abs().max()).item() for k in ref}
    for k in ref:
        assert errs["bf16x6"][k] <= max(2.0 * errs["fp32"][k], 2e-7), (k, errs)
        assert errs["bf16x9"][k] <= max(2.0 * errs["fp32"][k], 2e-7), (k, errs)
    print("max-norm relative errors against fp64:", errs)
