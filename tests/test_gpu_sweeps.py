"""The random sweeps of ``tools/fuzz_*.py`` as part of the GPU suite, on fixed
seed lists and under ``SCVAE_WS_GUARD=1`` (guard regions behind every workspace
buffer, checked after every step): the sweep over the graph options found the
one real defect of round 5 (seed 5041: the fused ``-k`` step's scratch overran
with fewer genes than hidden units) after the hand-written tests were green.

Each sweep is a subprocess (its own process-wide ``SCVAE_WS_GUARD``) comparing
the HIP path with the fp64 oracle (``oracle/models.py``: va:2219-2770,
gm:2788-3470) or with an fp64 torch restatement of the heads + likelihoods
(``oracle/likelihoods.py``: du:206-305); the seed lists below ran clean on
MI355X when they were pinned -- a failure is a regression, not noise.
"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sweep(script, *args, env=None, timeout=240):
    environment = dict(os.environ, SCVAE_WS_GUARD="1", PYTHONPATH=ROOT)
    environment.update(env or {})
    done = subprocess.run(
        [sys.executable, os.path.join(ROOT, "tools", script)] + [str(a) for a in args],
        env=environment, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    return done.stdout


def test_graph_options_against_the_oracle_small_minibatches(cuda_device):
    """40 random graphs (VAE / GMVAE, every likelihood, -k, dropout sites,
    importance / Monte-Carlo samples, LFM architectures, decoder extras,
    priors, free nats) at 3-30 cells, seeds 5030-5069 -- 5041 among them --:
    scalars, every gradient and the evaluation statistics against fp64."""
    out = _sweep("fuzz_options.py", 40, 5030)
    assert re.search(r"^0 of 40 cases differ$", out, re.M), out[-3000:]


def test_graph_options_against_the_oracle_large_minibatches(cuda_device):
    """The same sweep at 130-700 cells: the producer / consumer head kernel,
    the tile chain, row groups."""
    out = _sweep("fuzz_options.py", 16, 0, env={"FUZZ_B_RANGE": "130,700"})
    assert re.search(r"^0 of 16 cases differ$", out, re.M), out[-3000:]


def test_fused_head_call_over_random_shapes(cuda_device):
    """40 random shapes through ``scvae_decoder_fused[_u16]`` (train and
    forward, both kernels, slabs and atomics, fp32 and default arithmetic)."""
    out = _sweep("fuzz_heads.py", 40, 0)
    m = re.search(r"^40 cases, (\d+) mismatches", out, re.M)
    assert m and int(m.group(1)) == 0, out[-3000:]


def test_tile_chain_over_random_shapes(cuda_device):
    """20 random large-minibatch configurations: tilechain.hip against the
    chain of launches, and bitwise repeatable."""
    out = _sweep("fuzz_tilechain.py", 0, 20)
    assert re.search(r"20 configurations, 0 failures", out), out[-3000:]
