"""A training step of the headline model with and without hidden-layer dropout (keep 0.9).
    python tools/bench_dropout.py [cells] [u16]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scvae_amd.engine import Engine
from scvae_amd.minibatch import synthetic_count_matrix
dev = torch.device("cuda:0")
F, B, L = 32738, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 25
matrix, _ = synthetic_count_matrix(8192, F, density=0.05, seed=60, device=dev)
for keeps in (None, [0.9, 1.0, 1.0]):
    eng = Engine(F, L, (100, 100), "negative binomial", batch_norm=True, device=dev, seed=0,
                 dropout_keep_probabilities=keeps)
    eng.reserve(B, 1)
    u16 = len(sys.argv) > 2 and sys.argv[2] == "u16" and eng.accepts_counts_u16(B, True, n_iw=1)
    x = (torch.empty(B, matrix.u16_pitch, dtype=torch.uint16, device=dev) if u16
         else torch.empty(B, F, device=dev))
    rc = torch.empty(B, device=dev)
    rows = torch.arange(B, device=dev)
    matrix.request(rows, x, rc).issue()
    eps = torch.randn(1, B, L, device=dev)
    def step(i):
        kw = {"dropout_seed": 100 + i} if keeps else {}
        eng.step(x, x, eps=eps, row_const=rc, training=True, x_counts=u16, **kw)
        eng.adam_step(1e-4)
    for i in range(5): step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for i in range(20): step(i)
    e1.record(); torch.cuda.synchronize()
    print("keeps", keeps, "B", B, ": %.3f ms per step (%s minibatch resident, no fetch)" % (e0.elapsed_time(e1) / 20, "uint16" if u16 else "fp32"))
