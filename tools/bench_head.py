"""Time the fused decoder-head TRAINING kernel alone (train = 3: main kernel only) for both
arithmetics at the benchmark's shape and compare their results.
    python tools/bench_head.py [rows] [F] [H] [likelihood] [launches]"""
import ctypes
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scvae_amd import _lib

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32738
H = int(sys.argv[3]) if len(sys.argv) > 3 else 100
name = sys.argv[4] if len(sys.argv) > 4 else "negative binomial"
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 10
lib = _lib.load()
dev = torch.device("cuda:0")
kind, heads = _lib.LIKELIHOOD_KINDS[name]
P = len(heads)
g = torch.Generator(device=dev).manual_seed(5)
d = torch.relu(torch.randn(rows, H, device=dev, generator=g))
WS = float(os.environ.get("WSCALE", "0.1"))
W = [torch.randn(H, F, device=dev, generator=g) * WS for _ in range(P)]
b = [torch.randn(F, device=dev, generator=g) * 0.1 for _ in range(P)]
t = torch.poisson(torch.full((rows, F), 2.0, device=dev), generator=g)
t = t * (torch.rand(rows, F, device=dev, generator=g) < 0.05)
gw = torch.full((rows,), -1.0 / rows, device=dev)
rc = torch.lgamma(t + 1).sum(dim=1)
ld = (F + 63) // 64 * 64
t16 = torch.zeros(rows, ld, dtype=torch.int32, device=dev)
t16[:, :F] = t.to(torch.int32)
t16 = t16.to(torch.uint16)
ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F),
                 dtype=torch.uint8, device=dev)
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
results = {}
for arith in (0, 1):
    flag = _lib.HEADS_BF16X9 if arith else _lib.HEADS_FP32
    dW = [torch.zeros_like(w) for w in W]
    db = [torch.zeros_like(v) for v in b]
    ll = torch.zeros(rows, device=dev)
    dd = torch.zeros(rows, H, device=dev)

    def launch(train):
        _lib.check(lib.scvae_decoder_fused_u16(
            kind, train | flag, d.data_ptr(), rows, H, arr(W), arr(b), arr(dW),
            arr(db), F, t16.data_ptr(), ld, rows, gw.data_ptr(), rc.data_ptr(),
            ll.data_ptr(), dd.data_ptr(), ws.data_ptr(), stream), "fused")
    launch(1)
    torch.cuda.synchronize()
    results[arith] = (ll.clone(), dd.clone(), [x.clone() for x in dW],
                      [x.clone() for x in db])
    for _ in range(3):
        launch(3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(launches):
        launch(3)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    flops = 2.0 * rows * F * P * 3 * H
    print("arith {} kernel {}: {:.3f} ms  {:.1f} TFLOP/s algorithmic".format(
        arith, lib.scvae_decoder_train_kernel(kind, H, arith), ms, flops / ms / 1e9))
a, c = results[0], results[1]


def rel(x, y):
    return ((x - y).abs().max() / y.abs().max()).item()
print("ll max rel diff", rel(a[0], c[0]), " dd", rel(a[1], c[1]),
      " dW", [rel(x, y) for x, y in zip(a[2], c[2])],
      " db", [rel(x, y) for x, y in zip(a[3], c[3])])
print("identical bits:", torch.equal(a[0], c[0]), torch.equal(a[1], c[1]))
