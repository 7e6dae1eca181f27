"""A 4096-cell NB training step at several hidden widths (every fused head kernel takes even
decoder widths up to 126; a training step under the bf16x9 arithmetic any width up to 256 on the
producer / consumer kernel; the input layer's count kernels and the tile chain stop at 128 units).
    python tools/bench_hidden.py [widths ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scvae_amd.engine import Engine
from scvae_amd.minibatch import synthetic_count_matrix

dev = torch.device("cuda:0")
F, B, L = 32738, 4096, 25
widths = [int(a) for a in sys.argv[1:]] or [100, 126, 128, 256]
matrix, _ = synthetic_count_matrix(8192, F, density=0.05, seed=60, device=dev)
for H in widths:
    eng = Engine(F, L, (H, H), "negative binomial", batch_norm=True, device=dev, seed=0)
    eng.reserve(B, 1)
    u16 = eng.accepts_counts_u16(B, True, n_iw=1)
    x = (torch.empty(B, matrix.u16_pitch, dtype=torch.uint16, device=dev) if u16
         else torch.empty(B, F, device=dev))
    rc = torch.empty(B, device=dev)
    matrix.request(torch.arange(B, device=dev), x, rc).issue()
    eps = torch.randn(1, B, L, device=dev)

    def step():
        eng.step(x, x, eps=eps, row_const=rc, training=True, x_counts=True)
        eng.adam_step(1e-4)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        step()
    e1.record()
    torch.cuda.synchronize()
    print("hidden {0}-{0}: {1:.3f} ms per step ({2} minibatch)".format(
        H, e0.elapsed_time(e1) / 20, "uint16" if u16 else "fp32"))
