"""Per-section cycle sums of decoder_head4_kernel (probe build scvae_amd/csrc/build_prof.sh):
    python tools/d4_prof.py [likelihood] [rows]
prints, for the eight waves of workgroup 0, the s_memtime cycles per section and tile."""
import ctypes
import os
import sys

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, here)
os.environ["SCVAE_HIP_LIBRARY"] = os.path.join(here, "scvae_amd", "csrc", "libscvae_hip_prof.so")
import torch
from scvae_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "negative binomial"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
F, H = 32738, 100
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
for _ in range(3):
    _lib.check(lib.scvae_decoder_fused_u16(
        kind, 3 | int(os.environ.get("TIME_HEAD_FLAGS", "0"), 0), d.data_ptr(), rows, H, arr(W), arr(b), arr(dW), arr(db), F, t16.data_ptr(),
        ld, rows, gw.data_ptr(), rc.data_ptr(), ll.data_ptr(), dd.data_ptr(), ws.data_ptr(),
        stream), "fused")
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ["SCVAE_HIP_LIBRARY"])
out = (ctypes.c_ulonglong * 96)()
NPW = 4 if (os.environ.get("SCVAE_D4_PRODUCERS") == "4" or P >= 3) else 8
raw.scvae_d4_prof_dump(out)
tiles = (rows + 31) // 32
names = {True: ["gemm1", "dense", "walk", "sum+G", "barrier"],
         False: ["prep", "gemm3", "ddstore", "gemm2", "barrier"]}
print(name, "rows", rows, "- cycles per 32-row tile, workgroup 0")
for w in range(12):
    if w >= NPW + 4:
        break
    v = [out[w * 8 + k] / tiles for k in range(5)]
    print("wave {} ({}): ".format(w, "producer" if w < NPW else "consumer") +
          "  ".join("{} {:7.0f}".format(n, x) for n, x in zip(names[w < NPW], v)) +
          "   total {:7.0f}".format(sum(v)))
