"""Phase probe of the resident tile-chain kernels (a -DSCVAE_TC_PROBE build of tilechain.hip
loaded through SCVAE_HIP_LIBRARY): s_memtime ticks of workgroup 1 per (stage, phase), mean per launch.
    SCVAE_HIP_LIBRARY=$PWD/scvae_amd/csrc/libscvae_hip_tcprobe.so python tools/tc_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scvae_amd.minibatch import synthetic_count_matrix
from scvae_amd import _lib
dev = torch.device("cuda:0")
matrix, _ = synthetic_count_matrix(16384, 32738, density=0.05, seed=60, device=dev)
w = bench.Workload(matrix, dev, 4096, bench.LIKELIHOOD, bench.LATENT)
for _ in range(60):
    w.one_step()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 128)()
assert lib.scvae_debug_tc_probe(buf) == 0
names = ["run", "stores done", "issue", "wait", "partials"]
for d, label in ((0, "forward"), (1, "backward")):
    n = buf[d * 64 + 63]
    if not n:
        continue
    print(label, "launches", n)
    tot = 0
    for st in range(12):
        vals = [buf[d * 64 + 5 * st + k] / n for k in range(5)]
        if any(vals):
            tot += sum(vals)
            print("  stage {:2d}: ".format(st) + "  ".join(
                "{} {:7.0f}".format(nm, v) for nm, v in zip(names, vals)))
    print("  total ticks", round(tot))
    if d == 0:
        inner = ["merge", "normalise + tile -> LDS", "weights -> LDS", "product", "acc -> LDS",
                 "bias + store", "tile statistics"]
        print("  inside run, all tile stages: " + "  ".join(
            "{} {:.0f}".format(nm, buf[40 + k] / n) for k, nm in enumerate(inner)))
