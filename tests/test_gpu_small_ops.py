"""The small ops of the graph through their stand-alone C-ABI entries
(include/scvae_hip.h, "the small ops of the graph") against
the oracle's fp64 restatement (``oracle/models.py``: ``dense_layer``'s batch
norm, ``_normal_log_prob``, the KL_y block, ``log_reduce_exp_mean``;
``oracle/likelihoods.py``: ``mean_variance``) with autograd for the backward
entries.  Reference: mu:60-76, 129-137; du:52-73; gm:2936-3048, 3242-3261,
3272-3292; va:2665-2734."""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import likelihoods as lk
from oracle import models as om

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).float().to(device)


def _close(got, want, rtol, what):
    got = got.detach().cpu().double().numpy()
    want = np.asarray(want.detach().numpy() if hasattr(want, "detach") else want,
                      dtype=np.float64)
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert np.isfinite(got).all() and err <= rtol, (what, err)


@pytest.mark.parametrize("rows,N", [(100, 100), (4096, 100), (37, 7), (1000, 128)])
@pytest.mark.parametrize("relu", [0, 1])
def test_batch_norm_stats_apply_and_backward(cuda_device, rows, N, relu):
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(rows + N + relu)
    a = rng.normal(0.3, 2.0, (rows, N))
    beta = rng.normal(0, 0.5, N)
    dh = rng.normal(0, 1, (rows, N))
    ad, bd, dhd = _dev(a, cuda_device), _dev(beta, cuda_device), _dev(dh, cuda_device)
    ws = torch.empty(int(lib.scvae_bn_workspace_floats(N)) + 2 * N, device=cuda_device)
    mean = torch.empty(N, device=cuda_device)
    var = torch.empty(N, device=cuda_device)
    h = torch.empty(rows, N, device=cuda_device)
    da = torch.empty(rows, N, device=cuda_device)
    dbeta = torch.empty(N, device=cuda_device)
    _lib.check(lib.scvae_bn_stats(_p(ad), N, rows, N, _p(mean), _p(var), _p(ws), _stream()),
               "bn_stats")
    _lib.check(lib.scvae_bn_apply_relu_fwd(_p(ad), N, _p(mean), _p(var), _p(bd), _p(h), N,
                                           rows, N, relu, _stream()), "bn_apply_fwd")
    _lib.check(lib.scvae_bn_apply_relu_bwd(_p(dhd), N, _p(h), N, _p(ad), N, _p(mean), _p(var),
                                           rows, N, relu, _p(da), N, _p(dbeta), _p(ws),
                                           _stream()), "bn_apply_bwd")
    torch.cuda.synchronize()
    # oracle: dense_layer with an identity affine map (mu:53-76)
    at = torch.from_numpy(a).requires_grad_(True)
    bt = torch.from_numpy(beta).requires_grad_(True)
    params = {"S/DENSE/weights": torch.eye(N, dtype=torch.float64),
              "S/DENSE/biases": torch.zeros(N, dtype=torch.float64),
              "S/BATCH_NORM/beta": bt}
    moving = {"S/BATCH_NORM/moving_mean": torch.zeros(N, dtype=torch.float64),
              "S/BATCH_NORM/moving_variance": torch.ones(N, dtype=torch.float64)}
    out = om.dense_layer(at, params, "S", True, True, moving, None, activation=bool(relu))
    (out * torch.from_numpy(dh)).sum().backward()
    _close(mean, at.detach().mean(dim=0), 1e-5, "mean")
    _close(var, at.detach().var(dim=0, unbiased=False), 1e-5, "var")
    _close(h, out, 2e-6, "h")
    # (a unit within rounding of the ReLU kink may fall on the other side in fp32)
    before = om.dense_layer(at.detach(), {k: v.detach() for k, v in params.items()}, "S",
                            True, True, moving, None, activation=False)
    live = before.abs() > 1e-5
    assert live.double().mean() > 0.99
    _close(da * live.to(cuda_device), at.grad * live, 5e-4 if relu else 2e-5, "da")
    _close(dbeta, bt.grad, 5e-4 if relu else 2e-5, "dbeta")


@pytest.mark.parametrize("K,S,B,L", [(3, 1, 48, 5), (20, 2, 64, 100), (2, 3, 7, 130)])
def test_softplus_gaussian_logprob_pair(cuda_device, K, S, B, L):
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(K * 100 + L)
    arrays = dict(qm=rng.normal(0, 1, (K * B, L)), qs=rng.normal(0, 1.5, (K * B, L)),
                  Wpm=rng.normal(0, 1, (K, L)), bpm=rng.normal(0, 0.3, L),
                  Wps=rng.normal(0, 1, (K, L)), bps=rng.normal(0, 0.3, L),
                  eps=rng.normal(0, 1, (K, S, B, L)),
                  dz=rng.normal(0, 1, (K, S, B, L)), gklz=rng.normal(0, 1, (K, S, B)))
    d = {k: _dev(v, cuda_device) for k, v in arrays.items()}
    z = torch.empty(K, S, B, L, device=cuda_device)
    klz = torch.empty(K, S, B, device=cuda_device)
    qvar = torch.empty(K * B, L, device=cuda_device)
    _lib.check(lib.scvae_softplus_gaussian_logprob_pair_fwd(
        _p(d["qm"]), _p(d["qs"]), _p(d["Wpm"]), _p(d["bpm"]), _p(d["Wps"]), _p(d["bps"]),
        _p(d["eps"]), _p(z), _p(klz), _p(qvar), K, S, B, L, _stream()), "pair_fwd")
    dqm = torch.empty(K * B, L, device=cuda_device)
    dqs = torch.empty(K * B, L, device=cuda_device)
    dpr = torch.empty(K * B, 2 * L, device=cuda_device)
    _lib.check(lib.scvae_softplus_gaussian_logprob_pair_bwd(
        _p(d["qm"]), _p(d["qs"]), _p(d["Wpm"]), _p(d["bpm"]), _p(d["Wps"]), _p(d["bps"]),
        _p(d["eps"]), _p(d["dz"]), _p(d["gklz"]), _p(dqm), _p(dqs), _p(dpr), K, S, B, L,
        _stream()), "pair_bwd")
    torch.cuda.synchronize()
    t = {k: torch.from_numpy(v).requires_grad_(k in ("qm", "qs", "Wpm", "Wps", "bpm", "bps"))
         for k, v in arrays.items()}
    # du:52-73: sigma = sqrt(softplus(s)); gm:3272-3292: sum_L log q(z) - log p(z)
    sigma = torch.sqrt(torch.nn.functional.softplus(t["qs"])).reshape(K, 1, B, L)
    mean = t["qm"].reshape(K, 1, B, L)
    zz = mean + sigma * t["eps"]
    pm = (t["Wpm"] + t["bpm"]).reshape(K, 1, 1, L)
    ps = torch.sqrt(torch.nn.functional.softplus(t["Wps"] + t["bps"])).reshape(K, 1, 1, L)
    want_kl = (om._normal_log_prob(zz, mean, sigma)
               - om._normal_log_prob(zz, pm, ps)).sum(dim=-1)
    ((zz * t["dz"]).sum() + (want_kl * t["gklz"]).sum()).backward()
    _close(z, zz, 1e-6, "z")
    _close(klz, want_kl, 2e-5, "klz")
    _close(qvar, (sigma ** 2).reshape(K * B, L), 1e-6, "qvar")
    _close(dqm, t["qm"].grad, 5e-5, "dqm")
    _close(dqs, t["qs"].grad, 5e-5, "dqs")
    per = dpr.reshape(K, B, 2 * L).sum(dim=1)        # summed over the cells: the Z/P rows
    _close(per[:, :L], t["Wpm"].grad, 1e-4, "d prior mean rows")
    _close(per[:, L:], t["Wps"].grad, 1e-4, "d prior scale rows")


@pytest.mark.parametrize("B,K", [(100, 20), (7, 3), (513, 64)])
@pytest.mark.parametrize("prior", ["uniform", "learned"])
def test_categorical_entropy_kl(cuda_device, B, K, prior):
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(B + K)
    logits = rng.normal(0, 2, (B, K))
    dy = rng.normal(0, 1, (B, K))
    pl = rng.normal(0, 1, K) if prior == "learned" else None
    ld, dyd = _dev(logits, cuda_device), _dev(dy, cuda_device)
    pld = _dev(pl, cuda_device) if pl is not None else None
    y = torch.empty(B, K, device=cuda_device)
    kl = torch.empty(B, device=cuda_device)
    _lib.check(lib.scvae_categorical_entropy_kl_fwd(_p(ld), _p(y), _p(kl), B, K, _p(pld),
                                                    _stream()), "cat_fwd")
    c = 0.37
    gate = torch.ones(1, device=cuda_device)
    dl = torch.empty(B, K, device=cuda_device)
    _lib.check(lib.scvae_categorical_entropy_kl_bwd(_p(y), _p(dyd), _p(gate), c, _p(dl), B, K,
                                                    _p(pld), _stream()), "cat_bwd")
    gate0 = torch.zeros(1, device=cuda_device)
    dl0 = torch.empty(B, K, device=cuda_device)
    _lib.check(lib.scvae_categorical_entropy_kl_bwd(_p(y), _p(dyd), _p(gate0), c, _p(dl0), B, K,
                                                    _p(pld), _stream()), "cat_bwd")
    torch.cuda.synchronize()
    lt = torch.from_numpy(logits).requires_grad_(True)
    log_y = torch.log_softmax(lt, dim=-1)
    yt = torch.exp(log_y)
    if pl is None:       # gm:3242-3254: log K - H[q(y|x)]
        want_kl = math.log(K) + (yt * log_y).sum(dim=-1)
    else:                # gm:3256-3258: kl(q_y || p_y)
        log_p = torch.log_softmax(torch.from_numpy(pl), dim=-1)
        want_kl = (yt * (log_y - log_p)).sum(dim=-1)
    _close(y, yt, 1e-6, "y")
    _close(kl, want_kl, 2e-5, "kl_y")
    ((yt * torch.from_numpy(dy)).sum() + c * want_kl.sum()).backward()
    _close(dl, lt.grad, 5e-5, "dlogits (gate on)")
    lt2 = torch.from_numpy(logits).requires_grad_(True)
    (torch.softmax(lt2, dim=-1) * torch.from_numpy(dy)).sum().backward()
    _close(dl0, lt2.grad, 5e-5, "dlogits (gate off)")


@pytest.mark.parametrize("n_iw,n_mc,B,per_sample", [
    (1, 1, 100, 0), (5, 1, 64, 0), (3, 2, 50, 1), (4, 3, 1000, 0), (2, 1, 4096, 1)])
def test_iw_logmeanexp(cuda_device, n_iw, n_mc, B, per_sample):
    from scvae_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n_iw * 10 + n_mc + B)
    S = n_iw * n_mc
    ll = rng.normal(-300, 30, (n_iw, n_mc, B))
    kl = rng.gamma(2.0, 2.0, (n_iw, n_mc, B) if per_sample else (B,))
    w = 0.6
    lld, kld = _dev(ll, cuda_device), _dev(kl, cuda_device)
    scalars = torch.zeros(8, device=cuda_device)
    gw = torch.empty(S * B, device=cuda_device)
    _lib.check(lib.scvae_iw_logmeanexp(_p(lld), _p(kld), per_sample, n_iw, n_mc, B, w,
                                       1.0 / (n_mc * B), _p(scalars), _p(gw), _stream()),
               "iw_logmeanexp")
    torch.cuda.synchronize()
    lt = torch.from_numpy(ll).requires_grad_(True)
    kt = torch.from_numpy(kl)
    kk = kt if per_sample else kt.reshape(1, 1, B)
    # va:2717-2734 with mu:129-137 (log_reduce_exp_mean over the importance samples)
    lb = om.log_reduce_exp_mean(lt - kk, 0).mean()
    lbw = om.log_reduce_exp_mean(lt - w * kk, 0).mean()
    (-lbw).backward()
    s = scalars.cpu().double().numpy()
    assert abs(s[0] - lb.item()) <= 2e-6 * abs(lb.item())
    assert abs(s[1] - lbw.item()) <= 2e-6 * abs(lbw.item())
    assert abs(s[2] - ll.mean()) <= 2e-6 * abs(ll.mean())
    assert abs(s[3] - kl.mean()) <= 1e-5 * abs(kl.mean())
    _close(gw.reshape(n_iw, n_mc, B), lt.grad, 2e-5, "gw")


@pytest.mark.parametrize("likelihood", ["poisson", "negative binomial",
                                        "zero-inflated negative binomial"])
@pytest.mark.parametrize("S,B,F", [(1, 50, 300), (4, 20, 1000)])
def test_pxmean_stats(cuda_device, likelihood, S, B, F):
    from scvae_amd import _lib
    lib = _lib.load()
    kind, heads = _lib.LIKELIHOOD_KINDS[likelihood]
    rng = np.random.default_rng(S + B + F + kind)
    pre = [rng.normal(0, 1.5, (S * B, F)) for _ in heads]
    pred = [_dev(v, cuda_device) for v in pre]
    arr = (ctypes.c_void_p * len(pred))(*[t.data_ptr() for t in pred])
    weight = rng.random(B)
    wd = _dev(weight, cuda_device)
    outs = [torch.zeros(B, F, device=cuda_device) for _ in range(3)]
    _lib.check(lib.scvae_pxmean_stats(kind, arr, S, B, F, None, 0, 0, _p(outs[0]), _p(outs[1]),
                                      _p(outs[2]), _stream()), "pxmean_stats")
    acc = [torch.ones(B, F, device=cuda_device) for _ in range(3)]
    _lib.check(lib.scvae_pxmean_stats(kind, arr, S, B, F, _p(wd), 1, 1, _p(acc[0]), _p(acc[1]),
                                      _p(acc[2]), _stream()), "pxmean_stats (mixture)")
    torch.cuda.synchronize()
    mean, var = lk.mean_variance(likelihood, tuple(torch.from_numpy(v) for v in pre))
    mean, var = mean.reshape(S, B, F), var.reshape(S, B, F)
    # va:2665-2713: mean over the samples, mean of the variances, variance of the means
    _close(outs[0], mean.mean(dim=0), 2e-5, "p_x_mean")
    _close(outs[1], var.mean(dim=0), 2e-5, "mean of var")
    if S > 1:
        _close(outs[2], ((mean - mean.mean(dim=0)) ** 2).mean(dim=0), 2e-4, "var of mean")
    else:   # one sample: zero up to the rounding of (m - 1.0 * m)
        assert (outs[2].cpu().double() <= 1e-12 * (mean[0] ** 2 + 1e-30)).all()
    # gm:3311-3386: the cluster's share y_k times the statistics, the variance taken about the
    # ALREADY WEIGHTED mean (gm:3357-3368, reproduced), accumulated over the clusters
    wt = torch.from_numpy(weight).reshape(B, 1)
    pm = mean.mean(dim=0) * wt
    _close(acc[0], 1.0 + pm, 2e-5, "weighted p_x_mean")
    _close(acc[1], 1.0 + var.mean(dim=0) * wt, 2e-5, "weighted mean of var")
    _close(acc[2], 1.0 + ((mean - pm) ** 2).mean(dim=0) * wt, 2e-4, "weighted var of mean")
