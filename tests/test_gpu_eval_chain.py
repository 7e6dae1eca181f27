"""Evaluation steps with the hidden stack as one launch (``eval_mlp_kernel``,
tilechain.hip; plan.hip: eval_chain_ok) against the same step as a chain of
launches (``SCVAE_EVAL_CHAIN=0``): with ``is_training = False`` a
batch-normalised layer uses its moving statistics (mu:60-70), so the layers
between the input layer's product and the likelihood heads are row-independent
and run for 16 cells per workgroup without leaving LDS (va:2219-2457 forward
only).  Both paths are fp32 products with fp32 accumulation in different
orders: agreement to 2e-5 of each tensor's magnitude, the step's scalars to
1e-5 relative."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SNIPPET = r"""
import json, sys
import numpy as np, torch
from scvae_amd.engine import Engine
dev = torch.device("cuda:0")
cases = [  # cells, genes, hidden, latent, likelihood (a sixth element: deterministic z)
    (300, 500, (100, 100), 25, "negative binomial"),
    (500, 400, (100, 100), 25, "negative binomial", True),
    (4096, 2000, (100, 100), 25, "negative binomial"),
    (129, 300, (64,), 7, "poisson"),
    (1000, 700, (128, 96, 32), 100, "zero-inflated negative binomial"),
    (777, 640, (33, 17), 128, "negative binomial"),
]
report = []
for cells, F, hidden, L, likelihood, *more in cases:
    deterministic = bool(more and more[0])
    rng = np.random.default_rng(cells + F)          # (host draws: the same in both processes)
    eng = Engine(F, L, hidden, likelihood, batch_norm=True, device=dev, seed=3)
    eng.reserve(cells, 1)
    # moving statistics that are not the initial (0, 1)
    eng.moving.copy_(torch.from_numpy(
        (rng.random(eng.moving.numel()) * 1.5 + 0.25).astype(np.float32)))
    x = torch.from_numpy((rng.poisson(3.0, (cells, F)) * (rng.random((cells, F)) < 0.2))
                         .astype(np.float32)).to(dev)
    eps = torch.from_numpy(rng.standard_normal((1, cells, L)).astype(np.float32)).to(dev)
    out = {"q_z_mean": torch.empty(cells, L, device=dev),
           "kl_neurons": torch.empty(L, device=dev)}
    scalars = eng.step(x, x, eps=eps, training=False, outputs=out, x_counts=True,
                       deterministic_z=deterministic).clone()
    torch.cuda.synchronize()
    report.append({"scalars": scalars.cpu().double().tolist(),
                   "q_z_mean": out["q_z_mean"].cpu().double().numpy().ravel().tolist()[:4000],
                   "kl_neurons": out["kl_neurons"].cpu().double().tolist()})
print("REPORT" + json.dumps(report))
"""


def _run(flag):
    env = dict(os.environ, SCVAE_EVAL_CHAIN=flag, PYTHONPATH=ROOT)
    done = subprocess.run([sys.executable, "-c", _SNIPPET], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    line = [l for l in done.stdout.splitlines() if l.startswith("REPORT")][-1]
    return json.loads(line[len("REPORT"):])


def test_evaluation_step_in_one_launch_equals_the_launch_chain(cuda_device):
    launches, one = _run("0"), _run("1")
    assert len(launches) == len(one) == 6
    for a, b in zip(launches, one):
        sa, sb = np.array(a["scalars"]), np.array(b["scalars"])
        assert np.isfinite(sb).all()
        assert np.allclose(sa[:4], sb[:4], rtol=1e-5, atol=1e-6), (sa, sb)
        for key in ("q_z_mean", "kl_neurons"):
            va, vb = np.array(a[key]), np.array(b[key])
            assert np.abs(va - vb).max() <= 2e-5 * max(np.abs(va).max(), 1e-6), key
