#!/usr/bin/env python3
"""Random shapes through the fused decoder-head call (train and forward; fp32 and uint16 targets;
dd through slabs and through XCD-local atomics; the library's default arithmetic and the fp32
matrix cores) against an fp64 torch reference of heads + likelihood + gradients: rows on both
sides of the 128-row switch between the all-in-one-phase and the producer / consumer kernel
(row groups, ragged last tiles), hidden widths up to 256 (odd ones, multiples of 32), gene counts
around the strip widths, repeated targets.  The shapes the unit tests pin are a few dozen; this
draws them.   python tools/fuzz_heads.py [cases] [seed]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import likelihoods as lk
from scvae_amd import _lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
names = list(lk.ELEMENTWISE_LIKELIHOODS)
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
bad = 0
worst = {"ll": 0.0, "dd": 0.0, "dW": 0.0, "db": 0.0}
for case in range(cases):
    name = names[rng.integers(len(names))]
    kind, heads = _lib.LIKELIHOOD_KINDS[name]
    P = len(heads)
    hmax = 159 if P == 3 else 256
    H = int(rng.choice([2, 17, 31, 32, 33, 64, 96, 100, 101, 126, 127, 128, 129, 150, 159, 160,
                        200, 255, 256]))
    H = min(H, hmax)
    cells = int(rng.choice([rng.integers(1, 129), rng.integers(129, 700), rng.integers(129, 2200)]))
    reps = int(rng.choice([1, 1, 1, 2, 3]))
    rows = cells * reps
    F = int(rng.choice([rng.integers(1, 130), rng.integers(130, 1500), rng.integers(1500, 9000)]))
    density = float(rng.choice([0.02, 0.1, 0.6]))
    arith = str(rng.choice(["default", "default", "fp32"]))
    if arith == "fp32" and (H % 2 or H > 126):   # (the fp32 kernels: even widths up to 126)
        H = int(rng.choice([2, 16, 30, 64, 96, 100, 126]))
    atomics = bool(rng.integers(2))
    u16 = bool(rng.integers(2))
    use_rc = bool(rng.integers(2))
    d = np.maximum(rng.normal(0, 1, (rows, H)), 0)
    W = [rng.normal(0, 0.3 / np.sqrt(max(H, 16) / 16.0), (H, F)) for _ in range(P)]
    b = [rng.normal(0, 0.3, F) for _ in range(P)]
    t = (rng.poisson(3.0, (cells, F)) * (rng.random((cells, F)) < density)).astype(np.float64)
    if name != "bernoulli":
        t.flat[int(rng.integers(t.size))] = float(rng.integers(256, 60000))
    else:
        t = (t > 0).astype(np.float64)
        u16 = False
    gw = rng.normal(0, 1, rows)
    T = torch.from_numpy
    dt = T(d).requires_grad_(True)
    Wt = [T(w).requires_grad_(True) for w in W]
    bt = [T(v).requires_grad_(True) for v in b]
    pre = tuple(dt @ w + v for w, v in zip(Wt, bt))
    ll_ref = lk.log_prob(name, T(t).repeat(reps, 1), pre).sum(dim=1)
    (ll_ref * T(gw)).sum().backward()
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
    dd_, td, gwd = f32(d), f32(t), f32(gw)
    Wd, bd = [f32(w) for w in W], [f32(v) for v in b]
    ld = (F + 63) // 64 * 64
    t16 = torch.zeros(cells, ld, dtype=torch.int32, device=dev)
    t16[:, :F] = td.to(torch.int32)
    t16 = t16.to(torch.uint16)
    rc = torch.lgamma(td.double() + 1).sum(dim=1).float() if use_rc else None
    ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F), dtype=torch.uint8,
                     device=dev)
    flags = (_lib.HEADS_FP32 if arith == "fp32" else 0) | (_lib.HEADS_DD_ATOMICS if atomics else 0)
    what = "{} rows {} (cells {}) F {} H {} {} atomics {} u16 {} rc {}".format(
        name, rows, cells, F, H, arith, atomics, u16, use_rc)
    problems = []
    for train in (1, 0):
        dWd = [torch.full_like(w, 7.0) for w in Wd]
        dbd = [torch.full_like(v, 7.0) for v in bd]
        ll = torch.full((rows,), 7.0, device=dev)
        dd = torch.full((rows, H), 7.0, device=dev)
        rcp = rc.data_ptr() if rc is not None else None
        try:
            if u16:
                _lib.check(lib.scvae_decoder_fused_u16(
                    kind, train | flags, dd_.data_ptr(), rows, H, arr(Wd), arr(bd), arr(dWd),
                    arr(dbd), F, t16.data_ptr(), ld, cells, gwd.data_ptr(), rcp, ll.data_ptr(),
                    dd.data_ptr(), ws.data_ptr(), stream), "fused_u16")
            else:
                _lib.check(lib.scvae_decoder_fused(
                    kind, train | flags, dd_.data_ptr(), rows, H, arr(Wd), arr(bd), arr(dWd),
                    arr(dbd), F, td.data_ptr(), cells, gwd.data_ptr(), rcp, ll.data_ptr(),
                    dd.data_ptr(), ws.data_ptr(), stream), "fused")
            torch.cuda.synchronize()
        except Exception as error:
            problems.append("train {}: {!r}".format(train, error)[:200])
            continue

        def rel(got, want):
            want = want.detach().numpy()
            got = got.cpu().double().numpy()
            if not np.isfinite(got).all():
                return float("inf")
            return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))
        want_ll = ll_ref.detach()
        if use_rc is False and name in ("poisson", "negative binomial", "zero-inflated poisson",
                                        "zero-inflated negative binomial"):
            pass   # (without the row constant the kernel adds lgamma(t + 1) itself)
        e = rel(ll, want_ll)
        worst["ll"] = max(worst["ll"], e)
        if not e <= 3e-5:
            problems.append("train {} ll {:.2e}".format(train, e))
        if train:
            for key, got, want in [("dd", dd, dt.grad)] + [
                    ("dW", dWd[j], Wt[j].grad) for j in range(P)] + [
                    ("db", dbd[j], bt[j].grad) for j in range(P)]:
                e = rel(got, want)
                worst[key] = max(worst[key], e)
                if not e <= 5e-5:
                    problems.append("{} {:.2e}".format(key, e))
    if problems:
        bad += 1
        print("MISMATCH", what, problems[:5], flush=True)
print("{} cases, {} mismatches, worst relative differences {}".format(
    cases, bad, {k: "{:.1e}".format(v) for k, v in worst.items()}))
