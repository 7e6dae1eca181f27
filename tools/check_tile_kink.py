"""A mismatch reported by tools/fuzz_tilechain.py, replayed against the fp64 oracle: which
parameters differ between the tile chain and the chain of launches, how many hidden units are
involved, and which of the two paths the oracle sides with.  A difference confined to one unit
of a layer (its beta, its column of the weights, the layers below) at ~1 / rows of the tensor's
magnitude is the ReLU kink described in the fuzzer's header.
Usage (GPU box): PYTHONPATH=. python tools/check_tile_kink.py <seed> <B> <F>"""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
from oracle import models as om
from scvae_amd.engine import Engine

seed, wantB, wantF = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
LK = ["negative binomial", "poisson", "zero-inflated negative binomial", "zero-inflated poisson"]
for it in range(10000):
    n_iw = int(rng.integers(1, 3)); n_mc = int(rng.integers(1, 3)); S = n_iw * n_mc
    B = int(rng.integers(129, 3000 // S + 1))
    H = tuple(int(rng.integers(1, 129)) for _ in range(int(rng.integers(1, 4))))
    L = int(rng.integers(1, 129)); F = int(rng.integers(5, 500)); lk = LK[int(rng.integers(0, 4))]
    xh = (rng.poisson(1.5, (B, F)) * (rng.random((B, F)) > 0.6)).astype(np.float32)
    eh = rng.standard_normal((S, B, L)).astype(np.float32)
    if (B, F) == (wantB, wantF):
        break
print("configuration", it, "B", B, "H", H, "L", L, "F", F, lk, "n_iw", n_iw, "n_mc", n_mc)
x = torch.from_numpy(xh).to(dev); eps = torch.from_numpy(eh).to(dev)
res = {}
for tile in (True, False):
    eng = Engine(F, L, H, lk, batch_norm=True, device=dev, seed=it)
    g = torch.Generator().manual_seed(it)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    eng.set_tile_chain(tile)
    params = {k: v.detach().cpu().double() for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double() for k, v in eng.named_moving_statistics().items()}
    eng.step(x, x, eps=eps, training=True, n_iw=n_iw, n_mc=n_mc, warm_up_weight=0.6)
    torch.cuda.synchronize()
    res[tile] = {k: v.detach().cpu().double().clone() for k, v in eng.named_gradients().items()}
cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H, likelihood=lk,
                     minibatch_normalisation=True, n_iw=n_iw, n_mc=n_mc)
xd, ed = torch.from_numpy(xh).double(), torch.from_numpy(eh).double()
out, grads = om.gradients(lambda p: om.vae_forward(cfg, p, moving, xd, xd, ed, True, 0.6, {}), params)
for k in grads:
    s = grads[k].abs().max().item()
    if s < 1e-12:
        continue
    d = (res[True][k] - res[False][k]).abs()
    if d.max().item() <= 5e-6 * s:
        continue
    et = (res[True][k] - grads[k]).abs().max().item() / s
    ec = (res[False][k] - grads[k]).abs().max().item() / s
    units = (d > 5e-6 * s)
    n_units = int(units.any(dim=0).sum()) if d.dim() == 2 else int(units.sum())
    print("{:36s} tile-oracle {:.1e}  chain-oracle {:.1e}  tile-chain {:.1e}  units involved {}".format(
        k, et, ec, d.max().item() / s, n_units))
