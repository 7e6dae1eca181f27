"""Random shapes: the forward-only decoder-head call (train = 0; under the bf16x9 arithmetic the
forward instantiation of decoder_head3_kernel for one / two heads) against the training call's
log-likelihood on the same operands, fp32 and uint16 targets, with and without the row constant,
repeated targets (rows = reps x cells).
    python tools/fuzz_forward.py [cases] [seed]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scvae_amd import _lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
names = ["poisson", "negative binomial", "zero-inflated poisson",
         "zero-inflated negative binomial"]
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
worst, bad = 0.0, 0
for case in range(cases):
    name = names[rng.integers(len(names))]
    kind, heads = _lib.LIKELIHOOD_KINDS[name]
    P = len(heads)
    H = int(rng.choice([2, 16, 30, 40, 64, 96, 100, 110, 126]))
    cells = int(rng.integers(1, 700))
    reps = int(rng.choice([1, 1, 1, 2, 5]))
    rows = cells * reps
    F = int(rng.integers(1, 3000))
    density = float(rng.choice([0.02, 0.1, 0.5]))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    d = torch.relu(torch.randn(rows, H, device=dev, generator=g))
    W = [torch.randn(H, F, device=dev, generator=g) * 0.2 for _ in range(P)]
    b = [torch.randn(F, device=dev, generator=g) * 0.2 for _ in range(P)]
    t = torch.poisson(torch.full((cells, F), 3.0, device=dev), generator=g)
    t = t * (torch.rand(cells, F, device=dev, generator=g) < density)
    gw = torch.full((rows,), -1.0 / rows, device=dev)
    use_rc = bool(rng.integers(2))
    rc = torch.lgamma(t + 1).sum(dim=1) if use_rc else None
    ld = (F + 63) // 64 * 64
    t16 = torch.zeros(cells, ld, dtype=torch.int32, device=dev)
    t16[:, :F] = t.to(torch.int32)
    t16 = t16.to(torch.uint16)
    ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F), dtype=torch.uint8,
                     device=dev)
    dW = [torch.zeros_like(w) for w in W]
    db = [torch.zeros_like(v) for v in b]
    dd = torch.zeros(rows, H, device=dev)

    def call(train, u16):
        ll = torch.full((rows,), 7.0, device=dev)
        rcp = rc.data_ptr() if rc is not None else None
        if u16:
            _lib.check(lib.scvae_decoder_fused_u16(
                kind, train, d.data_ptr(), rows, H, arr(W), arr(b), arr(dW), arr(db), F,
                t16.data_ptr(), ld, cells, gw.data_ptr(), rcp, ll.data_ptr(), dd.data_ptr(),
                ws.data_ptr(), stream), "fused")
        else:
            _lib.check(lib.scvae_decoder_fused(
                kind, train, d.data_ptr(), rows, H, arr(W), arr(b), arr(dW), arr(db), F,
                t.data_ptr(), cells, gw.data_ptr(), rcp, ll.data_ptr(), dd.data_ptr(),
                ws.data_ptr(), stream), "fused")
        torch.cuda.synchronize()
        return ll.double()

    ref = call(1, False)
    for u16 in (False, True):
        got = call(0, u16)
        err = ((got - ref).abs() / (ref.abs() + 1.0)).max().item()
        worst = max(worst, err)
        if not (err <= 5e-6) or not torch.isfinite(got).all():
            bad += 1
            print("MISMATCH", name, "rows", rows, "cells", cells, "F", F, "H", H, "u16", u16,
                  "row_const", use_rc, "max rel", err, flush=True)
print("{} cases, {} mismatches, worst relative difference {:.2e}".format(cases, bad, worst))
