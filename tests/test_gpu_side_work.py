"""GPU: the work a step carries along (``scvae_side_work``: clip + Adam of the
step, fetch and noise of the next minibatch) against the same calls issued one
after the other -- bit for bit, over several steps of a pipelined loop, in line
(default) and on the plan's second stream (``SCVAE_SIDE_STREAM=1``, run in a
subprocess: the switch is read once per process)."""
import numpy as np
import pytest
import scipy.sparse
import torch

pytestmark = pytest.mark.gpu


def _matrix(cells, features, seed, device):
    from scvae_amd.minibatch import DeviceCSR
    rng = np.random.default_rng(seed)
    lam = rng.gamma(0.5, 3.0, size=(1, features))
    x = rng.poisson(lam, size=(cells, features)).astype(np.float32)
    x *= rng.random((cells, features)) > 0.8
    return DeviceCSR.from_scipy(scipy.sparse.csr_matrix(x), device)


def _engine(device, F, L, H, likelihood, model):
    from scvae_amd.engine import Engine
    eng = Engine(F, L, H, likelihood, batch_norm=True, device=device, seed=3,
                 model_type=model, n_clusters=4)
    g = torch.Generator().manual_seed(5)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights"):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return eng


def _loop(device, matrix, B, L, H, likelihood, model, u16, carried, steps=4):
    """``steps`` training steps on consecutive minibatches; returns the state
    after them and every step's scalars."""
    from scvae_amd.minibatch import philox_normal_blocks
    N, F = matrix.shape
    eng = _engine(device, F, L, H, likelihood, model)
    K = 4 if model == "GMVAE" else 1
    if u16:
        eng.set_count_gemm(True, always=True)
        assert matrix.integer_counts and eng.accepts_counts_u16(B, True)
        x = [torch.zeros(B, matrix.u16_pitch, dtype=torch.uint16, device=device)
             for _ in range(2)]
    else:
        x = [torch.zeros(B, F, device=device) for _ in range(2)]
    rc = [torch.zeros(B, device=device) for _ in range(2)]
    eps = [torch.zeros(K, B, L, device=device) for _ in range(2)]
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(device)
    rows = [perm[i * B:(i + 1) * B] for i in range(steps + 1)]
    scalars = []

    def fetch(i, slot):
        matrix.request(rows[i], x[slot], rc[slot]).issue()
        philox_normal_blocks(eps[slot], block_stride=B, row_offset=0, seed=7,
                             stream_id=i)
    fetch(0, 0)
    for i in range(steps):
        cur, nxt = i & 1, (i & 1) ^ 1
        kw = dict(eps=eps[cur], row_const=rc[cur], training=True,
                  x_counts=matrix.integer_counts)
        if carried:
            s = eng.step(x[cur], x[cur], learning_rate=1e-3,
                         next_minibatch=matrix.request(rows[i + 1], x[nxt], rc[nxt]),
                         next_noise=dict(out=eps[nxt], block_stride=B, row_offset=0,
                                         seed=7, stream_id=i + 1), **kw)
        else:
            s = eng.step(x[cur], x[cur], **kw)
            eng.adam_step(1e-3)
            fetch(i + 1, nxt)
        scalars.append(s.clone())
    torch.cuda.synchronize()
    last = steps & 1
    return dict(params=eng.params.cpu(), m=eng.adam_m.cpu(), v=eng.adam_v.cpu(),
                moving=eng.moving.cpu(), x=x[last].cpu(), rc=rc[last].cpu(),
                eps=eps[last].cpu(), scalars=torch.stack(scalars).cpu(),
                adam_t=eng.adam_t)


@pytest.mark.usefixtures("bit_repeatable")
@pytest.mark.parametrize("B,H,L,likelihood,model,u16", [
    (100, (64, 48), 10, "negative binomial", "VAE", False),       # mid-chain kernels
    (1024, (100, 100), 25, "negative binomial", "VAE", True),      # tile chain, count kernels
    (4096, (100, 100), 25, "negative binomial", "VAE", True),      # ... the fetch forks beside x^T dA
    (300, (32,), 8, "zero-inflated negative binomial", "VAE", False),
    (256, (48, 32), 6, "poisson", "VAE", True),
    (192, (40, 40), 8, "negative binomial", "GMVAE", True),
])
def test_carried_work_equals_the_serial_calls(cuda_device, B, H, L, likelihood,
                                              model, u16):
    matrix = _matrix(5 * B + 7, 1500, B, cuda_device)
    serial = _loop(cuda_device, matrix, B, L, H, likelihood, model, u16, False)
    carried = _loop(cuda_device, matrix, B, L, H, likelihood, model, u16, True)
    assert carried["adam_t"] == serial["adam_t"] == 4
    for key in ("scalars", "x", "rc", "eps", "params", "m", "v", "moving"):
        assert torch.equal(carried[key], serial[key]), key
    assert carried["x"].float().abs().sum() > 0 and carried["eps"].abs().sum() > 0


def test_an_evaluation_step_may_carry_the_next_fetch(cuda_device):
    B, F, L = 128, 900, 6
    matrix = _matrix(3 * B, F, 11, cuda_device)
    eng = _engine(cuda_device, F, L, (32, 32), "negative binomial", "VAE")
    x = [torch.zeros(B, F, device=cuda_device) for _ in range(2)]
    rc = [torch.zeros(B, device=cuda_device) for _ in range(2)]
    rows = torch.arange(3 * B, device=cuda_device)
    matrix.request(rows[:B], x[0], rc[0]).issue()
    want = matrix.gather_dense(rows[B:2 * B])
    params = eng.params.clone()
    eng.step(x[0], x[0], row_const=rc[0], training=False, deterministic_z=True,
             next_minibatch=matrix.request(rows[B:2 * B], x[1], rc[1]))
    torch.cuda.synchronize()
    assert torch.equal(x[1], want) and torch.equal(eng.params, params)
    with pytest.raises(ValueError):
        eng.step(x[0], x[0], row_const=rc[0], training=False,
                 deterministic_z=True, learning_rate=1e-3)


def test_the_carried_fetch_must_not_overwrite_the_steps_input(cuda_device):
    from scvae_amd import _lib
    B, F, L = 64, 400, 4
    matrix = _matrix(2 * B, F, 13, cuda_device)
    eng = _engine(cuda_device, F, L, (16,), "poisson", "VAE")
    x = torch.zeros(B, F, device=cuda_device)
    rows = torch.arange(B, device=cuda_device)
    eps = torch.zeros(1, B, L, device=cuda_device)
    with pytest.raises(_lib.HipLibraryError):
        eng.step(x, x, eps=eps, training=True, learning_rate=1e-3,
                 next_minibatch=matrix.request(rows, x))


def test_second_stream_variant_in_a_subprocess(cuda_device):
    """The same comparison with the fork switched on (both Adam fork points)."""
    import os
    import subprocess
    import sys
    for at in ("1", "2"):
        env = dict(os.environ, SCVAE_SIDE_STREAM="1", SCVAE_SIDE_ADAM_AT=at)
        out = subprocess.run(
            [sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k",
             "carried_work or evaluation_step"], env=env, capture_output=True,
            text=True)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_head_kernel_probe_times_the_training_steps(cuda_device):
    """``scvae_plan_probe_heads``: one HIP-event pair per training step around
    the likelihood-head kernel (what ``bench.py`` reports as ``roofline``):
    as many positive durations as armed steps ran, none for evaluation steps,
    nothing once disarmed."""
    B, F, L = 256, 700, 6
    matrix = _matrix(B, F, 17, cuda_device)
    eng = _engine(cuda_device, F, L, (32, 32), "negative binomial", "VAE")
    x = matrix.gather_dense(torch.arange(B, device=cuda_device))
    eps = torch.randn(1, B, L, device=cuda_device)
    eng.probe_heads(5)
    for i in range(3):
        eng.step(x, x, eps=eps, training=True)
    eng.step(x, x, eps=eps, training=False)          # not probed
    times = eng.probe_heads_ms()
    assert len(times) == 3 and all(0.0 < t < 50.0 for t in times)
    for i in range(4):                               # only two pairs are left
        eng.step(x, x, eps=eps, training=True)
    assert len(eng.probe_heads_ms()) == 5
    eng.probe_heads(0)
    eng.step(x, x, eps=eps, training=True)
    assert eng.probe_heads_ms() == []
