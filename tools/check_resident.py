#!/usr/bin/env python3
"""Resident tile chain vs one launch per layer: per-parameter gradient differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scvae_amd.engine import Engine
B, H, L, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, (100, 100), 25, 1
F = 400
dev = torch.device("cuda:0")
rng = np.random.default_rng(B + L)
x = torch.from_numpy((rng.poisson(1.5, (B, F)) * (rng.random((B, F)) > 0.6)).astype(np.float32)).to(dev)
eps = torch.from_numpy(rng.standard_normal((S, B, L))).float().to(dev)
res = []
for resident in (True, False, True):
    eng = Engine(F, L, H, "negative binomial", batch_norm=True, device=dev, seed=4)
    eng.set_dd_atomics(False)
    eng.set_tile_resident(resident)
    eng.step(x, x, eps=eps, training=True).clone()
    torch.cuda.synchronize()
    res.append({k: v.clone().cpu() for k, v in eng.named_gradients().items()})
for k in res[0]:
    d = (res[0][k] - res[1][k]).abs().max().item()
    d2 = (res[0][k] - res[2][k]).abs().max().item()
    print("{:50s} resident-vs-launch {:.3e}  resident-vs-resident {:.3e}  scale {:.3e}".format(
        k, d, d2, res[1][k].abs().max().item()))
