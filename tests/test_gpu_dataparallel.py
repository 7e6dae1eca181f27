"""GPU test of the data-parallel hook plumbing with a single-rank RCCL group:
the step must give the same result with and without the collectives installed
(merge over one rank is the identity), for the VAE and the GMVAE."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def single_rank_group(cuda_device):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1,
                                device_id=cuda_device)
        created = True
    yield
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("model_type", ["VAE", "GMVAE"])
def test_step_with_sync_hook_equals_plain_step(cuda_device, single_rank_group,
                                               model_type):
    from scvae_amd.dataparallel import GradientSynchroniser
    from scvae_amd.engine import Engine
    F, L, H, B, K = 130, 5, (20, 16), 48, 3
    rng = np.random.default_rng(0)
    x = torch.from_numpy(
        (rng.poisson(2.0, (B, F)) * (rng.random((B, F)) > 0.6))
        .astype(np.float32)).to(cuda_device)
    shape = (1, B, L) if model_type == "VAE" else (K, 1, B, L)
    eps = torch.from_numpy(
        rng.standard_normal(shape).astype(np.float32)).to(cuda_device)
    results = []
    for with_sync in (False, True):
        eng = Engine(F, L, H, "negative binomial", batch_norm=True,
                     model_type=model_type, n_clusters=K, device=cuda_device,
                     seed=3, free_nats_proportion=0.5)
        eng.reserve(B, 1)
        if with_sync:
            sync = GradientSynchroniser(eng)
            sync.broadcast_state(0)
        scalars = eng.step(x, x, eps=eps, training=True).clone()
        if with_sync:
            sync.all_reduce_gradients()
            sync.all_reduce_scalars(scalars)
        torch.cuda.synchronize()
        results.append((scalars.cpu(), eng.grads.clone().cpu(),
                        eng.moving.clone().cpu()))
    # (weights after Adam are not compared: parameters whose gradient is
    # mathematically zero under batch norm only carry rounding noise, which
    # Adam normalises to O(lr) steps)
    for a, b in zip(results[0], results[1]):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-5 * scale + 1e-9


def _two_rank_worker(rank, port, model_type, result_path):
    """Rank body: both ranks share cuda:0 (gloo carries the collectives), each
    steps its half of the minibatch; rank 0 also steps the whole minibatch in a
    second engine and compares."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        from scvae_amd.dataparallel import GradientSynchroniser, shard_bounds
        from scvae_amd.engine import Engine
        device = torch.device("cuda:0")
        F, L, H, B, K = 130, 5, (20, 16), 48, 3
        rng = np.random.default_rng(0)
        x = torch.from_numpy(
            (rng.poisson(2.0, (B, F)) * (rng.random((B, F)) > 0.6))
            .astype(np.float32)).to(device)
        shape = (1, B, L) if model_type == "VAE" else (K, 1, B, L)
        eps = torch.from_numpy(
            rng.standard_normal(shape).astype(np.float32)).to(device)

        def engine():
            return Engine(F, L, H, "negative binomial", batch_norm=True,
                          model_type=model_type, n_clusters=K, device=device,
                          seed=3, free_nats_proportion=0.5)
        eng = engine()
        sync = GradientSynchroniser(eng)
        sync.broadcast_state(0)
        lo, hi = shard_bounds(B, 2, rank)
        eps_local = eps[..., lo:hi, :].contiguous()
        scalars = eng.step(x[lo:hi].contiguous(), x[lo:hi].contiguous(),
                           eps=eps_local, training=True,
                           global_cells=B).clone()
        # the VAE step announces everything but ENCODER/1 for an early all-reduce
        assert len(sync._pending) == (1 if model_type == "VAE" else 0)
        sync.all_reduce_gradients()
        assert not sync._pending
        sync.all_reduce_scalars(scalars)
        torch.cuda.synchronize()
        if rank == 0:
            ref = engine()
            ref_scalars = ref.step(x, x, eps=eps, training=True).clone()
            torch.cuda.synchronize()
            worst = 0.0
            for a, b in ((scalars, ref_scalars), (eng.grads, ref.grads),
                         (eng.moving, ref.moving)):
                scale = b.abs().max().item()
                worst = max(worst, (a - b).abs().max().item()
                            / (scale + 1e-12))
            with open(result_path, "w") as handle:
                handle.write(repr(worst))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model_type", ["VAE", "GMVAE"])
def test_two_ranks_equal_single_process(cuda_device, tmp_path, model_type):
    """Data parallel over 2 ranks == single process on the whole minibatch:
    gradients after the all-reduce, scalar sums, synchronised batch-norm moving
    statistics (real HIP kernels on both ranks; gloo only moves the bytes)."""
    import torch.multiprocessing as mp
    result = tmp_path / "worst.txt"
    port = 29600 + (os.getpid() % 200) + (0 if model_type == "VAE" else 1)
    mp.spawn(_two_rank_worker, args=(port, model_type, str(result)),
             nprocs=2, join=True)
    worst = float(result.read_text())
    assert worst <= 2e-5, worst
