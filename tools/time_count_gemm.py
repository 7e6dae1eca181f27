"""Time scvae_count_gemm_u16 (forward x W and weight gradient x^T dA) at the benchmark's shape:
python tools/time_count_gemm.py [rows] [F] [N];  --all <lib.so> ...: one subprocess per library."""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--all":
    for lib in sys.argv[2:]:
        env = dict(os.environ, SCVAE_HIP_LIBRARY=os.path.abspath(lib))
        out = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), out.stdout.strip() or out.stderr[-400:], flush=True)
    sys.exit(0)
import torch
from scvae_amd import _lib
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32738
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
x = torch.poisson(torch.full((rows, F), 2.0, device=dev), generator=g)
x = x * (torch.rand(rows, F, device=dev, generator=g) < 0.05)
ld = (F + 63) // 64 * 64
x16 = torch.zeros(rows, ld, dtype=torch.int32, device=dev)
x16[:, :F] = x.to(torch.int32)
x16 = x16.to(torch.uint16)
W = torch.randn(F, N, device=dev, generator=g) * 0.05
dA = torch.randn(rows, N, device=dev, generator=g)
bias = torch.zeros(N, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
res, sums = [], []
for mode in (0, 1):
    other = W if mode == 0 else dA
    M = rows if mode == 0 else F
    out = torch.empty(M, N, device=dev)
    nb = lib.scvae_count_gemm_workspace_bytes(mode, rows, F, N)
    ws = torch.empty(nb + 16, dtype=torch.uint8, device=dev)

    def run():
        _lib.check(lib.scvae_count_gemm_u16(mode, P(x16), ld, rows, F, P(other), N, N,
                                            P(bias) if mode == 0 else None, 0, P(out), N, P(ws),
                                            nb, st), "count_gemm_u16")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 30 * 1e3)
    sums.append(out.double().abs().sum().item())
print("forward {:.1f} us, weight gradient {:.1f} us (split + kernel + reduce); checksums {:.9e} {:.9e}"
      .format(res[0], res[1], sums[0], sums[1]))
