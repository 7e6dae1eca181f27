"""Clip / saturation regime of the FUSED decoder-head kernels against the fp64
oracle (SURVEY.md §7 step 0 edge cases; du:206-305, va:2475-2485,
``zero_inflated.py:194-199``).

Every other fused-kernel test draws ``W ~ N(0, 0.3)``, ``d = relu(N(0, 1))``:
pre-activations of a few units, where no clip is active.  Here the bias rows
put every head's pre-activation at

    -95, -88 (sigmoid below float32.tiny: the lower clip and its zero gradient),
    -87.4 / -87.2 (either side of logit(tiny) = -87.3365),
    -11, -10, -9.99, 9.99, 10, 11 (the [-10, 10] support of the log heads),
    16.5, 17, 40 (p -> 1: fp32 sigmoid rounds to 1.0 from ~16.6 on)

against targets 0, 1, 255, 256 (first count with a non-zero low bf16 half),
4000, 30000 and 65535 (uint16 and fp32 storage), with a row of zeros, a row of
65535s and a row of ones, over several row tiles (the third and later tiles are
where the over-read bug of round 3 lived), for

* ``decoder_head4_kernel`` (producer / consumer schedule) and
  ``decoder_head3_kernel`` (training and forward instantiations) -- bf16x9,
* ``decoder_head2_kernel`` / ``decoder_head_kernel`` / ``decoder_forward_kernel``
  -- fp32 MFMA,

through the C ABI (``scvae_decoder_fused`` / ``scvae_decoder_fused_u16``), and
for the head-dropout (DROP) and constrained-Poisson (CP 1-3) instantiations
through a training step of the engine.  The kernel variants a process does not
take by default (schedule 3, four / eight producer waves, ``decoder_head_kernel`` for two heads, the forward
instantiation of the fp32 training kernels) run in subprocesses with the
library's A/B environment switches.

Asserted: per-cell log-likelihood within 1e-4 relative (+ 1e-3, one fp32 ulp of
1e4), EXACTLY zero ``dW`` / ``db`` columns where the head is outside its clip
support for every row, gradients per gene column within 1e-4 of the column's
scale (+ the fp32 accumulation bound of the element's own terms where they
cancel), everything finite where the oracle is finite.

The ``p -> 1`` region (pre-activation > ~16.6): the reference's own fp32 graph
has ``sigmoid(a) == 1.0f`` there, its upper clip ``1 - float32.tiny == 1.0f``
is a no-op and TFP's ``log1p(-p)`` is ``-inf``: the
reference's log-likelihood of such an element is ``-inf`` / NaN.  The oracle
and the kernels evaluate the algebraically equal stable form
``log(1 - p) = log_sigmoid(-a)``, finite everywhere -- the one place where this
build deliberately differs from the reference's arithmetic, and it differs only
where the reference is itself non-finite.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import likelihoods as lk
from oracle import models as om

from _parity import LL_ATOL, LL_RTOL, close_elementwise, close_maxnorm

pytestmark = pytest.mark.gpu

PRE = (-95.0, -88.0, -87.4, -87.2, -11.0, -10.0, -9.99, 0.0, 9.99, 10.0, 11.0,
       16.5, 17.0, 40.0)
TGT = (0, 1, 255, 256, 4000, 30000, 65535)
LOGIT_OF_TINY = lk.LOGIT_OF_TINY
#: heads whose activation is the sigmoid clipped at float32.tiny (gate
#: ``a >= logit(tiny)``); the others are log heads clipped to [-10, 10]
SIGMOID_HEADS = ("p", "pi")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IN_CHILD = "SCVAE_EDGE_TEST_CHILD"


def _case(name, rows, F, H, seed=0):
    """(d, W[], b[], t, gw): head j of gene f sits at PRE[(f + 5 j) % 14] plus a
    perturbation of a few hundredths; target of (row, f) is TGT[(f + row) % 7]."""
    heads = lk.LIKELIHOOD_PARAMETERS[name]
    rng = np.random.default_rng(seed)
    d = np.maximum(rng.normal(0, 1, (rows, H)), 0)
    scale = 0.02 / np.sqrt(H / 20.0)
    W = [rng.normal(0, scale, (H, F)) for _ in heads]
    f = np.arange(F)
    b = [np.asarray(PRE)[(f + 5 * j) % len(PRE)].astype(np.float64)
         for j in range(len(heads))]
    t = np.asarray(TGT)[(f[None, :] + np.arange(rows)[:, None]) % len(TGT)]
    t = t.astype(np.float64)
    t[0] = 0.0
    if rows > 1:
        t[1] = 65535.0
    if rows > 2:
        t[2] = 1.0
    gw = rng.normal(0, 1, rows)
    return d, W, b, t, gw


def _reference(name, d, W, b, t, gw):
    T = torch.from_numpy
    dt = T(d).requires_grad_(True)
    Wt = [T(w).requires_grad_(True) for w in W]
    bt = [T(v).requires_grad_(True) for v in b]
    pre = tuple(dt @ w + v for w, v in zip(Wt, bt))
    for p in pre:
        p.retain_grad()
    ll = lk.log_prob(name, T(t), pre).sum(dim=1)
    (ll * T(gw)).sum().backward()
    assert torch.isfinite(ll).all()
    return (ll.detach().numpy(), dt.grad.numpy(), [w.grad.numpy() for w in Wt],
            [v.grad.numpy() for v in bt], [p.detach().numpy() for p in pre],
            [p.grad.numpy() for p in pre])


def _outside_support(name, pre):
    """per head: genes whose pre-activation is outside the head's clip support
    for EVERY row (the gradient of the whole column is exactly zero)."""
    out = []
    for head, a in zip(lk.LIKELIHOOD_PARAMETERS[name], pre):
        if head in SIGMOID_HEADS:
            out.append((a < LOGIT_OF_TINY - 0.01).all(axis=0))
        else:
            out.append(((a < -10.0) | (a > 10.0)).all(axis=0))
    return out


def _columns_close(got, want, what, terms=None):
    """per gene column: |got - want| <= 1e-4 of the column's largest |want| (+ an
    absolute floor of 1e-4 of the tensor's median column scale) + 4e-6 of the
    sum of the ABSOLUTE terms of the element's contraction (``terms``): an fp32
    accumulation of 200 terms of one sign carries ~sqrt(200) 2^-24 of their sum,
    and where the terms cancel (rate e^10 against targets below it) that, not
    the element's own magnitude, is the scale of the rounding error."""
    got = got.cpu().double().numpy()
    assert np.isfinite(got).all(), what + ": non-finite"
    want2, got2 = np.atleast_2d(want), np.atleast_2d(got)
    col = np.abs(want2).max(axis=0)
    floor = 1e-4 * max(np.median(col), 1e-30)
    allowed = 1e-4 * col[None, :] + floor
    if terms is not None:
        allowed = allowed + 4e-6 * np.atleast_2d(terms)
    excess = np.abs(got2 - want2) - allowed
    i = np.unravel_index(np.argmax(excess), excess.shape)
    assert excess[i] <= 0, "{}: element {} got {!r} want {!r}".format(
        what, i, got2[i], want2[i])


def _check(device, name, rows, F, H, arith, u16, extra_flags=0):
    from scvae_amd import _lib
    lib = _lib.load()
    kind, heads = _lib.LIKELIHOOD_KINDS[name]
    P = len(heads)
    # (the clips are discontinuities of the gradient: a pre-activation within fp32 rounding of
    #  a boundary may fall on the other side of it in the kernel -- draw again until no element
    #  of the fp64 reference is closer than 1e-5 to a boundary of its own head)
    for seed in range(200):
        d, W, b, t, gw = _case(name, rows, F, H, seed)
        ll_ref, dd_ref, dW_ref, db_ref, pre, G = _reference(name, d, W, b, t, gw)
        margin = min(
            np.abs(a - edge).min()
            for head, a in zip(lk.LIKELIHOOD_PARAMETERS[name], pre)
            for edge in ((LOGIT_OF_TINY,) if head in SIGMOID_HEADS else (-10.0, 10.0)))
        if margin > 1e-5:
            break
    else:
        raise AssertionError("no draw keeps clear of the clip boundaries")
    clipped = _outside_support(name, pre)

    flag = _lib.HEAD_ARITH_FLAGS[arith] | extra_flags   # (they travel with the call)
    if True:
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(device)
        dd_, gwd = f32(d), f32(gw)
        Wd, bd = [f32(w) for w in W], [f32(v) for v in b]
        dWd = [torch.full_like(w, 7.0) for w in Wd]
        dbd = [torch.full_like(v, 7.0) for v in bd]
        ll = torch.full((rows,), 7.0, device=device)
        dd = torch.full((rows, H), 7.0, device=device)
        rc = torch.lgamma(torch.from_numpy(t) + 1).sum(dim=1).float().to(device)
        ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F),
                         dtype=torch.uint8, device=device)
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if u16:
            ld = (F + 63) // 64 * 64
            t16 = torch.zeros(rows, ld, dtype=torch.int32)
            t16[:, :F] = torch.from_numpy(t).to(torch.int32)
            td = t16.to(torch.uint16).to(device)
        else:
            td = f32(t)
        for train in (0, 1):
            if u16:
                rcode = lib.scvae_decoder_fused_u16(
                    kind, train | flag, dd_.data_ptr(), rows, H, arr(Wd), arr(bd),
                    arr(dWd), arr(dbd), F, td.data_ptr(), ld, rows,
                    gwd.data_ptr(), rc.data_ptr(), ll.data_ptr(), dd.data_ptr(),
                    ws.data_ptr(), stream)
            else:
                rcode = lib.scvae_decoder_fused(
                    kind, train | flag, dd_.data_ptr(), rows, H, arr(Wd), arr(bd),
                    arr(dWd), arr(dbd), F, td.data_ptr(), rows, gwd.data_ptr(),
                    rc.data_ptr(), ll.data_ptr(), dd.data_ptr(), ws.data_ptr(),
                    stream)
            _lib.check(rcode, "scvae_decoder_fused")
            torch.cuda.synchronize()
            assert torch.isfinite(ll).all(), "ll not finite (train=%d)" % train
            close_elementwise(ll, ll_ref, rtol=LL_RTOL, atol=LL_ATOL,
                              what="{} per-cell ll (train={})".format(name, train))

    for j in range(P):
        zero = clipped[j]
        assert zero.any() and not zero.all()
        gdW, gdb = dWd[j].cpu().numpy(), dbd[j].cpu().numpy()
        assert (gdW[:, zero] == 0.0).all(), (
            "dW of head %d: non-zero outside the clip support" % j)
        assert (gdb[zero] == 0.0).all(), (
            "db of head %d: non-zero outside the clip support" % j)
        assert np.abs(dW_ref[j][:, zero]).max() == 0.0      # (the oracle agrees)
        _columns_close(dWd[j], dW_ref[j], "dW%d" % j,
                       terms=np.abs(d).T @ np.abs(G[j]))
        _columns_close(dbd[j], db_ref[j], "db%d" % j,
                       terms=np.abs(G[j]).sum(axis=0))
    # dd [rows, H]: per ROW scale (a row's gradient is gw[row] times sums over the genes)
    terms = sum(np.abs(G[j]) @ np.abs(W[j]).T for j in range(P))
    _columns_close(dd.T, dd_ref.T, "dd", terms=terms.T)


NAMES = list(lk.ELEMENTWISE_LIKELIHOODS)


@pytest.mark.parametrize("u16", [False, True], ids=["f32", "u16"])
@pytest.mark.parametrize("arith", ["fp32", "bf16x9", "bf16x6"])
@pytest.mark.parametrize("name", NAMES)
def test_clip_and_saturation_regime(cuda_device, name, arith, u16):
    # 200 rows: four 64-row / seven 32-row tiles; 196 genes: every (pre, target)
    # pair of a head occurs, the last strip is ragged
    _check(cuda_device, name, 200, 196, 20, arith, u16)


@pytest.mark.parametrize("name", ["negative binomial",
                                  "zero-inflated negative binomial"])
def test_clip_regime_at_the_benchmarked_width(cuda_device, name):
    # H = 100: four contraction steps, weight planes of 112 rows
    _check(cuda_device, name, 130, 140, 100, "bf16x9", True)
    _check(cuda_device, name, 130, 140, 100, "fp32", False)


@pytest.mark.parametrize("name", ["negative binomial",
                                  "zero-inflated negative binomial"])
def test_clip_regime_with_dd_through_atomics(cuda_device, name):
    from scvae_amd import _lib
    _check(cuda_device, name, 200, 196, 20, "bf16x9", True,
           extra_flags=_lib.HEADS_DD_ATOMICS)


VARIANTS = [
    {"SCVAE_D3_SCHEDULE": "3"},              # decoder_head3_kernel (training instantiation)
    {"SCVAE_D4_PRODUCERS": "4"},             # four producer waves (two heads)
    {"SCVAE_D4_PRODUCERS": "8"},             # eight producer waves (one head)
    {"SCVAE_DECODER_VARIANT": "1"},          # decoder_head_kernel for one / two heads
    {"SCVAE_DECODER_FORWARD": "0"},          # forward instantiation of the fp32 kernels
    {"SCVAE_DECODER_FORWARD": "1"},          # decoder_forward_kernel under bf16x9
]


@pytest.mark.skipif(os.environ.get(IN_CHILD) == "1", reason="child process")
@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: "-".join(
    "{}={}".format(k[6:], v) for k, v in e.items()))
def test_other_kernel_variants(cuda_device, env):
    """The same cases on the kernels this process does not take by default
    (the library reads its A/B switches once per process)."""
    child = dict(os.environ, **env)
    child[IN_CHILD] = "1"
    out = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q",
         "-m", "gpu", "-k", "test_clip_and_saturation_regime"],
        env=child, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]


# ---- DROP / CP instantiations: through a training step of the engine ----
# (the decoder output of an engine step is not under the test's control -- batch-normalised
#  outlier cells reach ~10 -- so the head weights are tiny and the bias cycle keeps half a unit
#  clear of the clip boundaries; the boundaries themselves are the C-ABI cases' above)
PRE_ENGINE = (-95.0, -88.0, -88.5, -86.0, -11.0, -10.7, -9.3, 0.0, 9.3, 10.7, 11.0,
              16.5, 17.0, 40.0)
def _edge_counts(cells, features):
    f = np.arange(features)
    x = np.asarray(TGT)[(f[None, :] + np.arange(cells)[:, None]) % len(TGT)]
    x = x.astype(np.float64)
    x[0] = 0.0
    x[1] = 65535.0
    x[2] = 1.0
    return x


def _engine_step_case(device, likelihood, keeps, B, F, H, L=5):
    from scvae_amd.engine import Engine
    eng = Engine(F, L, H, likelihood, device=device, batch_norm=True,
                 dropout_keep_probabilities=keeps)
    g = torch.Generator().manual_seed(3)
    f = np.arange(F)
    head = 0
    for name, p in eng.named_parameters().items():
        if name.startswith("X_TILDE/") and name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.002)
        elif name.startswith("X_TILDE/") and name.endswith("biases"):
            p.copy_(torch.tensor(
                np.asarray(PRE_ENGINE)[(f + 5 * head) % len(PRE_ENGINE)],
                dtype=torch.float32))
            head += 1
        elif not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood=likelihood)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    rng = np.random.default_rng(0)
    counts = _edge_counts(B, F)
    if likelihood == "constrained poisson":
        # (a cell without counts has rate lambda N = 0 and the reference's own
        #  t log(rate) = 0 * -inf = NaN: not a case the oracle can referee)
        counts[0, ::3] = 2.0
    x = torch.from_numpy(counts)
    eps = torch.from_numpy(rng.standard_normal((1, B, L)))
    masks = None
    kwargs, extra = {}, {}
    if likelihood == "constrained poisson":
        kwargs["count_sum"] = x.sum(dim=1).float().to(device)
        extra["count_sum"] = x.sum(dim=1)
    if keeps:
        from test_gpu_dropout import SEED, _vae_masks
        masks = _vae_masks(eng, cfg, B, B, keeps, 0)
        kwargs["dropout_seed"] = SEED
    xd = x.float().to(device)
    ll = torch.zeros(B, device=device)
    sc = eng.step(xd, xd, eps=eps.float().to(device), training=True,
                  outputs={"log_p_x_given_z": ll}, **kwargs).cpu().numpy()
    torch.cuda.synchronize()
    if masks:
        extra["dropout"] = masks
    out, grads = om.gradients(
        lambda p: om.vae_forward(cfg, p, moving, x, x, eps, True, 1.0, {},
                                 **extra), params)
    assert np.isfinite(sc).all()
    close_elementwise(ll, out["log_p_x_given_z"].reshape(-1), rtol=LL_RTOL,
                      atol=LL_ATOL, what=likelihood + " per-cell ll")
    close_elementwise(np.asarray(sc[0]), np.asarray(float(out["lower_bound"])),
                      rtol=1e-4, what="lower_bound")
    for name, gr in eng.named_gradients().items():
        if not name.startswith("X_TILDE/"):
            continue
        want = grads[name].numpy()
        assert torch.isfinite(gr).all(), name
        close_maxnorm(gr, want, 2e-4, what="grad " + name)
        if name.endswith("biases") and likelihood != "constrained poisson":
            head = name.split("/")[1].lower()
            a = params[name].numpy()
            if head in SIGMOID_HEADS:
                zero = a < LOGIT_OF_TINY - 0.4
            else:
                zero = (a < -10.4) | (a > 10.4)
            assert zero.any()
            assert (gr.cpu().numpy()[zero] == 0.0).all(), name


@pytest.mark.parametrize("likelihood", NAMES)
def test_head_dropout_instantiation_in_the_clip_regime(cuda_device, likelihood):
    """``decoder_head3_kernel<..., DROP = true>``: dropout of the heads' input
    connections with every head in its clip / saturation regime."""
    from scvae_amd import _lib
    lib = _lib.load()
    kind, _ = _lib.LIKELIHOOD_KINDS[likelihood]
    assert lib.scvae_decoder_train_kernel(kind, 24, 1) == 3
    _engine_step_case(cuda_device, likelihood, (0.8, 0.0, 0.0), 200, 196,
                      (24, 24))


def test_constrained_poisson_passes_in_the_clip_regime(cuda_device):
    """``decoder_head3_kernel<LK_CPOISSON, ..., CP = 1 / 2 / 3>``: logits from
    -95 to 40 (the softmax is carried by the 40s; every other lambda sits at
    the float32.tiny clip, du:218-228), counts up to 65535."""
    _engine_step_case(cuda_device, "constrained poisson", None, 200, 196,
                      (24, 24))
