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


@pytest.mark.parametrize("model_type,B", [("VAE", 48), ("GMVAE", 48),
                                          ("VAE", 300), ("VAE", 1000),
                                          ("GMVAE", 128), ("GMVAE", 512)])
def test_step_with_sync_hook_equals_plain_step(cuda_device, single_rank_group,
                                               model_type, B):
    """(B > 128: the hidden layers on the tile chain -- under the hook with a
    layer's statistics merged per rank, handed to the hook and taken as given by
    the consuming kernel; 300 and 1000 rows: ragged last tiles, 5 / 16 chunks.
    GMVAE at 128 / 512 cells: the K passes as tile-chain groups of 2 / 8 tiles,
    every pass's statistics in one collective per layer.)"""
    from scvae_amd.dataparallel import GradientSynchroniser
    from scvae_amd.engine import Engine
    F, L, H, K = 130, 5, (20, 16), 3
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
        assert eng.uses_tile_chain(B) == (
            B > 128 if model_type == "VAE" else B % 64 == 0)
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
    second engine and compares.  ``model_type`` "<MODEL>-options" switches on
    graph options whose terms are sharded or reduced in their own way."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        from scvae_amd.dataparallel import GradientSynchroniser, shard_bounds
        from scvae_amd.engine import Engine
        device = torch.device("cuda:0")
        options = model_type.endswith("-options")
        dropout = model_type.endswith("-dropout")
        large = model_type.endswith("-large")     # (the tile chain under the hook: 203 / 203 rows)
        model_type = model_type.split("-")[0]
        # (GMVAE: whole 64-row tiles per pass and rank)
        F, L, H, K = 130, 5, (20, 16), 3
        B = (406 if model_type == "VAE" else 256) if large else 48
        n_iw = 2 if options else 1
        rng = np.random.default_rng(0)
        x = torch.from_numpy(
            (rng.poisson(2.0, (B, F)) * (rng.random((B, F)) > 0.6))
            .astype(np.float32)).to(device)
        x[:, 0] += 1
        shape = (n_iw, B, L) if model_type == "VAE" else (K, n_iw, B, L)
        eps = torch.from_numpy(
            rng.standard_normal(shape).astype(np.float32)).to(device)
        extra = count_sum = None
        kwargs = dict(batch_norm=True, model_type=model_type, n_clusters=K,
                      device=device, seed=3, free_nats_proportion=0.5)
        likelihood = "negative binomial"
        if options and model_type == "VAE":
            # count sums sharded with the cells, per-sample KL, importance weights
            likelihood = "constrained poisson"
            kwargs.update(analytical_kl_term=False)
            count_sum = x.sum(dim=1)
        elif options:
            # learned p(y) (its gradient is a share per rank), -k head, extras
            kwargs.update(prior_probabilities_method="learn", k_max=2,
                          decoder_extra=2)
            extra = torch.from_numpy(
                rng.random((B, 2)).astype(np.float32)).to(device)

        if dropout:
            # masks are keyed by the global row (scvae_step_args.row_offset):
            # a sharded step draws the masks of the single-process step
            kwargs.update(dropout_keep_probabilities=(
                (0.8, 0.9, 0.7) if model_type == "VAE"
                else (0.8, 0.9, 0.7, 0.85)))

        def engine():
            eng = Engine(F, L, H, likelihood, **kwargs)
            if "Y/P/LOGITS" in eng.named_parameters():
                eng.parameter("Y/P/LOGITS").copy_(
                    torch.tensor([0.3, -0.2, 0.1]))
            return eng

        def step(eng, lo, hi, **more):
            return eng.step(
                x[lo:hi].contiguous(), x[lo:hi].contiguous(),
                eps=eps[..., lo:hi, :].contiguous(), training=True, n_iw=n_iw,
                decoder_extra=(extra[lo:hi].contiguous()
                               if extra is not None else None),
                count_sum=(count_sum[lo:hi].contiguous()
                           if count_sum is not None else None),
                dropout_seed=77 if dropout else None, row_offset=lo,
                **more).clone()
        eng = engine()
        sync = GradientSynchroniser(eng)
        sync.broadcast_state(0)
        lo, hi = shard_bounds(B, 2, rank)
        scalars = step(eng, lo, hi, global_cells=B)
        assert eng.uses_tile_chain(hi - lo, n_iw) == large   # (the workspace is bound by now)
        # the VAE step announces the likelihood heads, then the hidden layers, for an
        # early all-reduce (everything but ENCODER/1)
        assert len(sync._pending) == (2 if model_type == "VAE" else 0)
        if model_type == "VAE":
            (o1, c1, _), (o2, c2, _) = sorted(sync._pending, key=lambda p: p[0])
            assert o1 + c1 == o2 and o2 + c2 == eng.grads.numel()
        sync.all_reduce_gradients()
        assert not sync._pending
        sync.all_reduce_scalars(scalars)
        torch.cuda.synchronize()
        if rank == 0:
            ref = engine()
            ref_scalars = step(ref, 0, B)
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


CASES = ["VAE", "GMVAE", "VAE-options", "GMVAE-options", "VAE-dropout",
         "GMVAE-dropout", "VAE-large", "GMVAE-large"]


@pytest.mark.parametrize("model_type", CASES)
def test_two_ranks_equal_single_process(cuda_device, tmp_path, model_type):
    """Data parallel over 2 ranks == single process on the whole minibatch:
    gradients after the all-reduce, scalar sums, synchronised batch-norm moving
    statistics (real HIP kernels on both ranks; gloo only moves the bytes)."""
    import torch.multiprocessing as mp
    result = tmp_path / "worst.txt"
    port = 29600 + (os.getpid() % 200) + CASES.index(model_type)
    mp.spawn(_two_rank_worker, args=(port, model_type, str(result)),
             nprocs=2, join=True)
    worst = float(result.read_text())
    assert worst <= (2e-5 if model_type in ("VAE", "GMVAE", "VAE-large",
                                            "GMVAE-large") else 1e-4), worst


def _model_train_worker(rank, port, directory, result_path):
    """Both ranks share cuda:0 (gloo): ``model.train`` + ``model.evaluate`` under
    data parallelism."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        from scvae_amd.data import DataSet
        from scvae_amd.models import VariationalAutoencoder
        rng = np.random.default_rng(5)
        n, F = 96, 40
        values = (rng.poisson(2.0, (n, F)) * (rng.random((n, F)) > 0.5)
                  ).astype(np.float32)
        data = DataSet("dp", values=values,
                       example_names=np.array(["c%d" % i for i in range(n)]),
                       feature_names=np.array(["g%d" % i for i in range(F)]))
        model = VariationalAutoencoder(
            feature_size=F, latent_size=3, hidden_sizes=[10],
            reconstruction_distribution="negative binomial",
            log_directory=os.path.join(directory, "dp"), device="cuda:0")
        np.random.seed(11)
        model.train(data, None, number_of_epochs=2, minibatch_size=32,
                    learning_rate=1e-3)
        # (blocks of the caching allocator that evaluate() is likely to get
        # back for its [n, F] outputs hold NaN: every row must be written)
        poison = [torch.full((n, F), float("nan"), device="cuda:0")
                  for _ in range(3)]
        del poison
        def dense(m):
            return m.toarray() if hasattr(m, "toarray") else np.asarray(m)
        everything = set(range(n))
        _, reconstructed, latent = model.evaluate(
            data, log_results=False, evaluation_subset_indices=everything)
        z = np.asarray(latent["z"].values)
        p_x_mean = dense(reconstructed.values)
        p_x_stddev = dense(reconstructed.total_standard_deviations)
        params = model.engine.params.clone()
        if rank == 0:
            single = VariationalAutoencoder(
                feature_size=F, latent_size=3, hidden_sizes=[10],
                reconstruction_distribution="negative binomial",
                log_directory=os.path.join(directory, "single"),
                device="cuda:0")
            # (a model outside the process group: train it as one process)
            import scvae_amd.models.base as base
            original = base._distributed
            base._distributed = lambda: (1, 0)
            try:
                np.random.seed(11)
                single.train(data, None, number_of_epochs=2,
                             minibatch_size=32, learning_rate=1e-3)
                _, reconstructed_single, latent_single = single.evaluate(
                    data, log_results=False,
                    evaluation_subset_indices=everything)
            finally:
                base._distributed = original
            reference = single.engine.params
            worst = ((params - reference).abs().max()
                     / reference.abs().max()).item()
            worst_z = float(np.abs(z - np.asarray(
                latent_single["z"].values)).max())
            want_mean = dense(reconstructed_single.values)
            want_stddev = dense(
                reconstructed_single.total_standard_deviations)
            worst_mean = float(np.abs(p_x_mean - want_mean).max()
                               / np.abs(want_mean).max())
            worst_stddev = float(np.abs(p_x_stddev - want_stddev).max()
                                 / np.abs(want_stddev).max())
            with open(result_path, "w") as handle:
                handle.write(repr((worst, worst_z, worst_mean, worst_stddev)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_model_train_two_ranks_equals_single_process(cuda_device, tmp_path):
    """The training / evaluation loops of the model class under two data-parallel
    ranks (rank-0 logging and checkpoints, broadcast permutation, sharded
    evaluation) reproduce the single-process run."""
    import torch.multiprocessing as mp
    result = tmp_path / "worst.txt"
    port = 29850 + (os.getpid() % 100)
    mp.spawn(_model_train_worker, args=(port, str(tmp_path), str(result)),
             nprocs=2, join=True)
    worst, worst_z, worst_mean, worst_stddev = eval(
        result.read_text().replace("nan", "float('nan')"))
    assert worst <= 5e-4, worst
    assert worst_z <= 5e-3, worst_z
    # the reconstructed data set: rows filled by the other rank included
    assert worst_mean <= 5e-3, worst_mean
    assert worst_stddev <= 5e-3, worst_stddev


def test_bench_with_eight_ranks_on_one_gpu():
    """``bench.py --gpus 8`` as the driver's scaling run starts it -- rendezvous
    on 127.0.0.1, the two communicators, shard arithmetic at N = 8 (4096 / 8 rows
    per rank, strong scaling), sync batch norm, the three-piece gradient
    all-reduce, the per-rank report -- with gloo moving the bytes and all eight
    ranks on the one GPU of the test box.  Plumbing only: nothing is timed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCVAE_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run(
        [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8",
         "--scaling", "strong", "--batch", "4096", "--cells", "8192",
         "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
         "--no-other-workloads"],
        env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 8 and r["ranks_in_communicator"] == 8
    assert r["scaling"] == "strong"
    assert r["config"]["global_batch"] == 4096
    assert r["config"]["cells_per_gpu_per_step"] == 512
    assert len(r["rank_step_ms_median"]) == 8
    assert len(r["rank_exposed_allreduce_ms_median"]) == 8
    assert r["value"] > 0 and r["last_lower_bound"] < 0


def _cli_run(tmp_path, name, ranks):
    """`scvae train` + `scvae evaluate` as a user starts them: one process, or
    `python -m torch.distributed.run --nproc-per-node 2` (gloo moves the bytes:
    both ranks share the test box's one GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    models = str(tmp_path / name)
    arguments = ["synthetic_1k", "-M", models, "-r", "negative_binomial",
                 "-l", "3", "-H", "20", "-B", "100", "--split-data-set"]
    entry = os.path.join(root, "tests", "_cli_entry.py")
    env = dict(os.environ, SCVAE_DIST_BACKEND="gloo", OMP_NUM_THREADS="2",
               PYTHONPATH=root)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(key, None)
    if ranks == 1:
        launcher = [sys.executable, entry]
    else:
        port = 29400 + (os.getpid() % 300)
        launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                    "--nproc-per-node", str(ranks), "--master-addr",
                    "127.0.0.1", "--master-port", str(port), entry]
    outputs = []
    for command in (["train"] + arguments + ["-e", "3"],
                    ["evaluate"] + arguments):
        done = subprocess.run(launcher + command, env=env, cwd=root,
                              capture_output=True, text=True, timeout=600)
        assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
        outputs.append(done.stdout)
    directory = os.path.join(
        models, "synthetic_1k", "split-random_0.9", "no_preprocessing", "VAE",
        "gaussian", "negative_binomial-l_3-h_20-mc_1-iw_1-kl-bn")
    return directory, outputs


def test_cli_starts_the_data_parallel_job(cuda_device, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m scvae_amd train ...`
    (missing 3): `cli.main` joins the process group from the
    launcher's environment, the two ranks train ONE model -- one log
    directory, rank 0's voice only, the weights of the single-process run --
    and leave the group; `evaluate` runs sharded the same way."""
    from scvae_amd.models.utilities import (checkpoint_epoch,
                                            get_checkpoint_state,
                                            load_checkpoint)
    single, out1 = _cli_run(tmp_path, "single", 1)
    double, out2 = _cli_run(tmp_path, "double", 2)
    # one voice: every progress line once
    assert out2[0].count("Training model for 3 epochs") == 1
    assert out1[0].count("Epoch 3") == out2[0].count("Epoch 3") >= 1
    assert out1[1].count("Evaluating trained") == out2[1].count(
        "Evaluating trained") >= 1
    paths = [get_checkpoint_state(d) for d in (single, double)]
    assert all(paths) and [checkpoint_epoch(p) for p in paths] == [3, 3]
    states = [load_checkpoint(p) for p in paths]
    assert states[0]["adam_t"] == states[1]["adam_t"] > 0
    for key in ("params", "moving"):
        a, b = states[0][key], states[1][key]
        worst = ((a - b).abs().max() / b.abs().max()).item()
        assert worst <= 5e-4, (key, worst)
