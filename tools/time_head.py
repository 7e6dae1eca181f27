"""Time the fused decoder-head training call of the library SCVAE_HIP_LIBRARY points at (default:
the regular build): main kernel alone (train = 3) and kernel + reduces (train = 1).
    python tools/time_head.py [rows] [F] [H] [likelihood] [launches]
    python tools/time_head.py --all <lib.so> ...   (one subprocess per library, NB and ZINB)"""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--all":
    for lib in sys.argv[2:]:
        for name, rows in (("negative binomial", 4096),
                           ("zero-inflated negative binomial", 4096)):
            env = dict(os.environ, SCVAE_HIP_LIBRARY=os.path.abspath(lib))
            out = subprocess.run([sys.executable, __file__, str(rows), "32738", "100", name, "20"],
                                 env=env, capture_output=True, text=True)
            print(os.path.basename(lib), out.stdout.strip() or out.stderr[-400:], flush=True)
    sys.exit(0)

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
ll = torch.zeros(rows, device=dev)
dd = torch.zeros(rows, H, device=dev)


FLAGS = int(os.environ.get("TIME_HEAD_FLAGS", "0"), 0)   # e.g. 0x400: dd with XCD-local atomics


def launch(train):
    _lib.check(lib.scvae_decoder_fused_u16(
        kind, train | FLAGS, d.data_ptr(), rows, H, arr(W), arr(b), arr(dW), arr(db), F, t16.data_ptr(),
        ld, rows, gw.data_ptr(), rc.data_ptr(), ll.data_ptr(), dd.data_ptr(), ws.data_ptr(),
        stream), "fused")


res = []
for train in (3, 1):
    for _ in range(3):
        launch(train)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(launches):
        launch(train)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / launches)
print("{} rows {}: kernel {:.3f} ms, kernel + reduces {:.3f} ms; checksum ll {:.6e} dd {:.6e} dW {:.6e}".format(
    name, rows, res[0], res[1], ll.double().sum().item(), dd.double().abs().sum().item(),
    dW[0].double().abs().sum().item()))
if os.environ.get("TIME_HEAD_SAVE"):
    launch(1)
    torch.cuda.synchronize()
    torch.save({"ll": ll.cpu(), "dd": dd.cpu(), "dW": [x.cpu() for x in dW], "db": [x.cpu() for x in db]},
               os.environ["TIME_HEAD_SAVE"])
if os.environ.get("TIME_HEAD_STRESS"):
    # run-to-run repeatability of the per-row log-likelihood (deterministic by construction)
    n = int(os.environ["TIME_HEAD_STRESS"])
    launch(1)
    torch.cuda.synchronize()
    ref_ll, ref_db = ll.clone(), [x.clone() for x in db]
    bad = 0
    for _ in range(n):
        launch(1)
        torch.cuda.synchronize()
        if not torch.equal(ll, ref_ll) or any(not torch.equal(x, y) for x, y in zip(db, ref_db)):
            bad += 1
    print("stress: {} of {} launches differ from the first (ll / db bitwise)".format(bad, n))
