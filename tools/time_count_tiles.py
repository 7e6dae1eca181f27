"""Time the tile-indexed minibatch at the benchmark's shape: the fetch of the tiles
(scvae_csr_count_tiles) and the input layer's two products from them
(scvae_count_gemm_tiles) beside the dense uint16 kernels (scvae_count_gemm_u16):
python tools/time_count_tiles.py [rows] [F] [N]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scvae_amd import _lib
from scvae_amd.minibatch import synthetic_count_matrix

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32738
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lib = _lib.load()
dev = torch.device("cuda:0")
m, _ = synthetic_count_matrix(4 * rows, F, density=0.05, seed=60, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
perm = torch.randperm(4 * rows, generator=g, device=dev)
idx = perm[:rows].contiguous()
x16 = m.gather_counts_u16(idx)
tiles = m.count_tiles(rows)
m.gather_count_tiles(idx, tiles)
torch.cuda.synchronize()
assert int(tiles.status.item()) == 0
W = torch.randn(F, N, device=dev, generator=g) * 0.05
dA = torch.randn(rows, N, device=dev, generator=g)
bias = torch.zeros(N, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("fetch of the tiles: {:.1f} us; of the uint16 batch: {:.1f} us".format(
    timed(lambda: m.gather_count_tiles(idx, tiles)),
    timed(lambda: m.gather_counts_u16(idx, out=x16))))
for mode in (0, 1):
    other = W if mode == 0 else dA
    M = rows if mode == 0 else F
    nb = lib.scvae_count_gemm_workspace_bytes(mode, rows, F, N)
    ws = torch.empty(nb + 16, dtype=torch.uint8, device=dev)
    outs = [torch.empty(M, N, device=dev) for _ in range(2)]

    def dense():
        _lib.check(lib.scvae_count_gemm_u16(mode, P(x16), x16.stride(0), rows, F, P(other), N, N,
                                            P(bias) if mode == 0 else None, 0, P(outs[0]), N,
                                            P(ws), nb, st), "count_gemm_u16")

    def sparse():
        _lib.check(lib.scvae_count_gemm_tiles(mode, ctypes.byref(tiles.struct), P(x16),
                                              x16.stride(0), rows, F, P(other), N, N,
                                              P(bias) if mode == 0 else None, 0, P(outs[1]), N,
                                              P(ws), nb, st), "count_gemm_tiles")
    td, ts = timed(dense), timed(sparse)
    print("mode {} ({}): dense uint16 {:.1f} us, tiles {:.1f} us (split + kernel + reduce); "
          "identical: {}".format(mode, "x W + b" if mode == 0 else "x^T dA", td, ts,
                                 torch.equal(outs[0], outs[1])))
