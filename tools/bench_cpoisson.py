"""A 4096-cell training step of the headline architecture with the Poisson likelihood (fused head
kernel) and the constrained Poisson likelihood (softmax over the genes x count sum: three passes of the
head kernel; before that unfused GEMMs + element-wise kernels), fp32 minibatch resident.  MI355X,
closing build of round 3: 1.83 / 2.35 ms (unfused: 3.13).
    python tools/bench_cpoisson.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scvae_amd.engine import Engine
from scvae_amd.minibatch import synthetic_count_matrix
dev = torch.device("cuda:0")
F, B, L = 32738, 4096, 25
matrix, _ = synthetic_count_matrix(8192, F, density=0.05, seed=60, device=dev)
for lk in ("poisson", "constrained poisson"):
    eng = Engine(F, L, (100, 100), lk, batch_norm=True, device=dev, seed=0)
    eng.reserve(B, 1)
    x = torch.empty(B, F, device=dev)
    rc = torch.empty(B, device=dev)
    rows = torch.arange(B, device=dev)
    matrix.request(rows, x, rc).issue()
    cs = x.sum(dim=1)
    eps = torch.randn(1, B, L, device=dev)
    def step(i):
        kw = {"count_sum": cs} if lk.startswith("constrained") else {}
        eng.step(x, x, eps=eps, row_const=rc, training=True, x_counts=False, **kw)
        eng.adam_step(1e-4)
    for i in range(5): step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for i in range(20): step(i)
    e1.record(); torch.cuda.synchronize()
    print(lk, ": %.3f ms per step" % (e0.elapsed_time(e1) / 20))
