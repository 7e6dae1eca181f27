"""Phase probe of tile_fwd_kernel (a -DSCVAE_TC_PROBE build of tilechain.hip loaded through
SCVAE_HIP_LIBRARY): s_memtime ticks per phase of workgroup 1, summed over the launches of a few
benchmark steps, split into launches with / without a batch-norm merge."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scvae_amd import _lib
from scvae_amd.minibatch import synthetic_count_matrix

dev = torch.device("cuda:0")
matrix, _ = synthetic_count_matrix(16384, bench.N_FEATURES, density=0.05, seed=60, device=dev)
w = bench.Workload(matrix, dev, 4096, bench.LIKELIHOOD, bench.LATENT)
w.run(20, 3, lambda: torch.cuda.synchronize(dev), min_warm_seconds=0.1)
torch.cuda.synchronize(dev)
lib = ctypes.CDLL(os.environ["SCVAE_HIP_LIBRARY"])
buf = (ctypes.c_ulonglong * 32)()
assert lib.scvae_debug_tc_probe(buf) == 0
names = ["stats0 in LDS", "stats1 in LDS", "merge done", "tile normalised", "weights in LDS",
         "product", "bias+store", "tile stats"]
for kind in (0, 1):
    row = buf[16 * kind:16 * kind + 16]
    n = max(1, row[15])
    print("bn-merge launches" if kind else "plain launches", "n =", row[15])
    for i, name in enumerate(names):
        print("   {:18s} {:9.0f} ticks per launch".format(name, row[i] / n))
    print("   total              {:9.0f}".format(sum(row[:8]) / n))
