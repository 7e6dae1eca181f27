"""Per-section cycle sums of count_tiles_fwd_kernel (probe build scvae_amd/csrc/build_ctprof.sh):
    python tools/ct_prof.py [rows]
prints, for the eight waves of workgroup (0, 0), the s_memtime cycles per section and chunk."""
import ctypes
import os
import sys

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, here)
os.environ["SCVAE_HIP_LIBRARY"] = os.path.join(here, "scvae_amd", "csrc", os.environ.get("CT_LIB", "libscvae_hip_ctprof.so"))
import torch
from scvae_amd import _lib
from scvae_amd.minibatch import synthetic_count_matrix

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dense = len(sys.argv) > 2 and sys.argv[2] == "dense"      # the uint16 batch instead of the tiles
F, N = 32738, 100
lib = _lib.load()
dev = torch.device("cuda:0")
m, _ = synthetic_count_matrix(2 * rows, F, density=0.05, seed=60, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
idx = torch.randperm(2 * rows, generator=g, device=dev)[:rows].contiguous()
x16 = m.gather_counts_u16(idx)
tiles = m.count_tiles(rows)
m.gather_count_tiles(idx, tiles)
W = torch.randn(F, N, device=dev, generator=g) * 0.05
out = torch.empty(rows, N, device=dev)
nb = lib.scvae_count_gemm_workspace_bytes(0, rows, F, N)
ws = torch.empty(nb + 16, dtype=torch.uint8, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(3):
    if dense:
        _lib.check(lib.scvae_count_gemm_u16(0, P(x16), x16.stride(0), rows, F, P(W), N, N, None, 0,
                                            P(out), N, P(ws), nb, st), "dense")
    else:
        _lib.check(lib.scvae_count_gemm_tiles(0, ctypes.byref(tiles.struct), P(x16), x16.stride(0), rows, F,
                                              P(W), N, N, None, 0, P(out), N, P(ws), nb, st), "tiles")
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ["SCVAE_HIP_LIBRARY"])
buf = (ctypes.c_ulonglong * 64)()
raw.scvae_ct_prof_dump(buf)
chunks = (F // 32 + 15) // 16          # chunks of one k-split (16 splits at 4096 rows)
names = ["work/W planes", "barrier", "wait", "unscatter | hi", "scatter", "lo", "requests"]     # waves 0-3 multiply (work, barrier), waves 4-7 stage
print("count_fwd2_kernel ({}), {} rows: cycles per chunk ({} chunks per workgroup), workgroup (0, 0)".format("uint16 batch" if dense else "tiles", rows, chunks))
for w in range(8):
    v = [buf[w * 8 + k] / chunks for k in range(7)]
    print("wave {}: ".format(w) + "  ".join("{} {:6.0f}".format(n, x) for n, x in zip(names, v)) +
          "   total {:6.0f}".format(sum(v)))
