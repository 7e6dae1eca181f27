"""Forward-only decoder-head call (train = 0) against the training call's log-likelihood on the same
operands -- the same arithmetic under the bf16x9 head kernels, so bit for bit -- with fp32 and
uint16 targets, and its time.
    python tools/check_forward.py [rows] [F] [H] [likelihood] [launches]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scvae_amd import _lib

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32738
H = int(sys.argv[3]) if len(sys.argv) > 3 else 100
name = sys.argv[4] if len(sys.argv) > 4 else "negative binomial"
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 20
lib = _lib.load()
dev = torch.device("cuda:0")
kind, heads = _lib.LIKELIHOOD_KINDS[name]
P = len(heads)
g = torch.Generator(device=dev).manual_seed(5)
d = torch.relu(torch.randn(rows, H, device=dev, generator=g))
W = [torch.randn(H, F, device=dev, generator=g) * 0.1 for _ in range(P)]
b = [torch.randn(F, device=dev, generator=g) * 0.1 for _ in range(P)]
t = torch.poisson(torch.full((rows, F), 2.0, device=dev), generator=g)
t = t * (torch.rand(rows, F, device=dev, generator=g) < 0.05)
gw = torch.full((rows,), -1.0 / rows, device=dev)
rc = torch.lgamma(t + 1).sum(dim=1)
ld = (F + 63) // 64 * 64
t16 = torch.zeros(rows, ld, dtype=torch.int32, device=dev)
t16[:, :F] = t.to(torch.int32)
t16 = t16.to(torch.uint16)
ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F), dtype=torch.uint8, device=dev)
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
dW = [torch.zeros_like(w) for w in W]
db = [torch.zeros_like(v) for v in b]
dd = torch.zeros(rows, H, device=dev)


def call(train, u16):
    ll = torch.zeros(rows, device=dev)
    if u16:
        _lib.check(lib.scvae_decoder_fused_u16(
            kind, train, d.data_ptr(), rows, H, arr(W), arr(b), arr(dW), arr(db), F,
            t16.data_ptr(), ld, rows, gw.data_ptr(), rc.data_ptr(), ll.data_ptr(), dd.data_ptr(),
            ws.data_ptr(), stream), "fused")
    else:
        _lib.check(lib.scvae_decoder_fused(
            kind, train, d.data_ptr(), rows, H, arr(W), arr(b), arr(dW), arr(db), F,
            t.data_ptr(), rows, gw.data_ptr(), rc.data_ptr(), ll.data_ptr(), dd.data_ptr(),
            ws.data_ptr(), stream), "fused")
    torch.cuda.synchronize()
    return ll


ref = call(1, False)
for u16 in (False, True):
    got = call(0, u16)
    bad = (got != ref).nonzero().flatten()
    rel = ((got - ref).abs() / ref.abs()).max().item()
    print("forward, {} targets: {} of {} rows differ from the training call (max rel {:.2e}){}".format(
        "uint16" if u16 else "fp32", bad.numel(), rows, rel,
        "; first: " + str(bad[:12].tolist()) if bad.numel() else ""))
    for _ in range(3):
        call(0, u16)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    ll = torch.zeros(rows, device=dev)
    e0.record()
    for _ in range(launches):
        if u16:
            lib.scvae_decoder_fused_u16(kind, 0, d.data_ptr(), rows, H, arr(W), arr(b), arr(dW),
                                        arr(db), F, t16.data_ptr(), ld, rows, gw.data_ptr(),
                                        rc.data_ptr(), ll.data_ptr(), dd.data_ptr(), ws.data_ptr(),
                                        stream)
        else:
            lib.scvae_decoder_fused(kind, 0, d.data_ptr(), rows, H, arr(W), arr(b), arr(dW), arr(db),
                                    F, t.data_ptr(), rows, gw.data_ptr(), rc.data_ptr(),
                                    ll.data_ptr(), dd.data_ptr(), ws.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    print("   {:.3f} ms per forward call".format(e0.elapsed_time(e1) / launches))
