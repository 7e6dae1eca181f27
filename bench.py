#!/usr/bin/env python3
"""Throughput benchmark of the scVAE training step on MI355X.

Metric (BASELINE.json): cells/s of NB-VAE training on a 68k-PBMC-shaped count
matrix (68 579 cells x 32 738 genes, synthetic, ~5 % nonzeros), hidden 100-100,
latent 25, batch norm on.  One "step" = one optimiser step over one minibatch:
CSR row gather + densify, encoder/decoder forward, likelihood + KL + ELBO,
backward, (gradient all-reduce), clip + Adam.  Inputs (the CSR matrix) are
resident in HBM before the timed region.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with two extra
objects: ``roofline`` (dominant kernel vs the fp32-MFMA / HBM peak, timed live
with HIP events) and ``cpu_baseline`` (the torch-CPU fp32 port of the same step,
``oracle/``, timed on a bounded sample on the host cores; rank 0, N=1 only).
"""

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_CELLS, N_FEATURES = 68579, 32738
HIDDEN, LATENT = (100, 100), 25
LIKELIHOOD = "negative binomial"
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def train_flops_per_cell(F, hidden, latent, heads):
    """Algorithmic training flops per cell:
    3 x forward MACs minus the input-layer dX, 2 flop per MAC."""
    sizes = [F] + list(hidden)
    enc = sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))
    post = 2 * hidden[-1] * latent
    dsz = [latent] + list(hidden[::-1])
    dec = sum(a * b for a, b in zip(dsz[:-1], dsz[1:]))
    head = heads * hidden[0] * F
    fwd = enc + post + dec + head
    train = 3 * fwd - F * hidden[0]
    return 2.0 * train


def cpu_baseline(matrix, batch, seconds_budget=15.0):
    """The reference-equivalent CPU path: scipy CSR gather + densify (x and t
    separately, as va:997-998 does), then the torch-CPU fp32 port of the step
    (oracle/models.py) on all host cores."""
    import numpy
    import scipy.sparse as sp
    from oracle import models as om
    torch.set_num_threads(os.cpu_count())
    n_rows = min(matrix.number_of_rows, 4 * batch)
    indptr = matrix.indptr[:n_rows + 1].cpu().numpy()
    nnz = int(indptr[-1])
    host = sp.csr_matrix(
        (matrix.values[:nnz].cpu().numpy(),
         matrix.indices[:nnz].cpu().numpy(), indptr),
        shape=(n_rows, matrix.shape[1]))
    cfg = om.ModelConfig(feature_size=matrix.shape[1], latent_size=LATENT,
                         hidden_sizes=HIDDEN, likelihood=LIKELIHOOD)
    shapes = om.vae_parameter_shapes(cfg)
    params = om.init_parameters(shapes, dtype=torch.float32)
    moving = om.init_moving_statistics(shapes, dtype=torch.float32)
    state = om.adam_state(params)
    rng = numpy.random.RandomState(2)
    steps, elapsed = 0, 0.0
    b = min(batch, n_rows)
    while True:
        idx = rng.permutation(n_rows)[:b]
        t0 = time.perf_counter()
        x = torch.from_numpy(host[idx].toarray())
        t = torch.from_numpy(host[idx].toarray())
        eps = torch.randn(1, b, LATENT)
        params, moving, _, _ = om.vae_train_step(
            cfg, params, moving, state, x, t, eps, 1e-4)
        dt = time.perf_counter() - t0
        if steps > 0:          # first step is warm-up
            elapsed += dt
        steps += 1
        if steps >= 2 and (elapsed >= seconds_budget or steps >= 12):
            break
    timed = steps - 1
    return {
        "value": timed * b / elapsed,
        "unit": "cells/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": "{} timed steps of {} cells (same model, same F), torch-CPU "
                  "fp32 port incl. scipy CSR gather+densify x2".format(
                      timed, b),
    }


def _measured_traffic(rows, F, H, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3
    PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section);
    only reported when the profiled kernel and shapes are the benchmarked ones."""
    for name in ("r01_pmc_decoder_head2.json", "r01_pmc_decoder_head.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            continue
        if (pmc.get("kernel") == "scvae::" + kernel and pmc.get("rows") == rows
                and pmc.get("features") == F and pmc.get("hidden") == H):
            return pmc["traffic_bytes_per_launch"]
    return None


def time_dominant_kernel(engine, rows, launches=10):
    """Average duration (HIP events on the launch stream) of the dominant
    kernel of the step -- the fused decoder-head kernel -- run standalone on
    the step's own shapes (main kernel only, without its two small reductions,
    so that the figure matches rocprofv3's per-kernel average)."""
    from scvae_amd import _lib
    lib = engine.lib
    F, H = engine.feature_size, engine.hidden_sizes[0]
    dev = engine.device
    kind, heads = _lib.LIKELIHOOD_KINDS[engine.likelihood]
    P = len(heads)
    g = torch.Generator(device=dev).manual_seed(5)
    d = torch.relu(torch.randn(rows, H, device=dev, generator=g))
    scope = "X/DISTRIBUTION" if engine.model_type == "GMVAE" else "X_TILDE"
    names = ["{}/{}/DENSE/".format(scope, h.upper()) for h in heads]
    W = [engine.parameter(n + "weights") for n in names]
    b = [engine.parameter(n + "biases") for n in names]
    dW = [torch.empty_like(w) for w in W]
    db = [torch.empty_like(v) for v in b]
    t = torch.poisson(torch.full((rows, F), 2.0, device=dev), generator=g)
    t = t * (torch.rand(rows, F, device=dev, generator=g) < 0.05)
    gw = torch.full((rows,), -1.0 / rows, device=dev)
    rc = torch.lgamma(t + 1).sum(dim=1)
    ll = torch.empty(rows, device=dev)
    dd = torch.empty(rows, H, device=dev)
    ws = torch.empty(lib.scvae_decoder_fused_workspace_bytes(rows, H, F),
                     dtype=torch.uint8, device=dev)

    def arr(ts):
        return (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
    aW, ab, adW, adb = arr(W), arr(b), arr(dW), arr(db)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _lib.check(lib.scvae_decoder_fused(
            kind, 3, d.data_ptr(), rows, H, aW, ab, adW, adb, F, t.data_ptr(),
            rows, gw.data_ptr(), rc.data_ptr(), ll.data_ptr(), dd.data_ptr(),
            ws.data_ptr(), stream), "scvae_decoder_fused")
    launch()
    torch.cuda.synchronize(dev)
    start, stop = torch.cuda.Event(True), torch.cuda.Event(True)
    start.record()
    for _ in range(launches):
        launch()
    stop.record()
    torch.cuda.synchronize(dev)
    seconds = start.elapsed_time(stop) / 1e3 / launches
    # algorithmic flops of the decoder heads: forward + dW + dX, 2 flop / MAC
    flops = 2.0 * rows * F * P * 3 * H
    if lib.scvae_decoder_fused_variant(kind, H) == 2:
        kernel = "decoder_head2_kernel<{}, true>".format(kind)
    else:
        kernel = "decoder_head_kernel<{}, true, {}>".format(
            kind, 32 if P >= 3 else 64)
    return {
        "kernel": "{} (X_TILDE heads + likelihood + dW/db/dd, "
                  "[rows,{}]x[{},{}]x{} heads)".format(kernel, H, H, F, P),
        "bound": "mfma",
        "achieved": flops / seconds / 1e12,
        "peak": PEAK_FP32_MFMA_TFLOPS,
        "unit": "TFLOP/s",
        "frac": flops / seconds / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "traffic": _measured_traffic(rows, F, H, kernel),
        "launch_us": seconds * 1e6,
        "algorithmic_flop_per_launch": flops,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096,
                    help="cells per GPU per step (weak scaling)")
    ap.add_argument("--cells", type=int, default=N_CELLS)
    ap.add_argument("--features", type=int, default=N_FEATURES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="vae", choices=["vae", "gmvae"],
                    help="extra (non-headline) workloads for DESIGN.md")
    ap.add_argument("--clusters", type=int, default=20)
    ap.add_argument("--latent", type=int, default=LATENT)
    ap.add_argument("--likelihood", default=LIKELIHOOD,
                    choices=["poisson", "negative binomial", "zero-inflated poisson",
                             "zero-inflated negative binomial"],
                    help="count likelihoods of the fused decoder-head kernel (the roofline "
                         "probe launches that kernel on its own)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(
                "--gpus {} needs torch.distributed.run with one rank per GPU"
                .format(args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback).")
    # (several ranks may share a GPU in the 1-GPU debugging set-up below)
    local_device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_device)
    device = torch.device("cuda", local_device)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL; SCVAE_BENCH_BACKEND=gloo only to exercise the N > 1 code path on a 1-GPU box
        backend = os.environ.get("SCVAE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import philox_normal, synthetic_count_matrix

    matrix, _ = synthetic_count_matrix(
        args.cells, args.features, density=0.05, seed=60, device=device)
    F = args.features
    B = args.batch
    GB = B * world
    gm = args.model == "gmvae"
    K = args.clusters if gm else 1
    L = args.latent
    engine = Engine(F, L, HIDDEN, args.likelihood, batch_norm=True,
                    model_type="GMVAE" if gm else "VAE", n_clusters=K,
                    device=device, seed=0)
    engine.reserve(B, 1)
    sync = None
    if world > 1:
        from scvae_amd.dataparallel import GradientSynchroniser
        sync = GradientSynchroniser(engine)
        sync.broadcast_state(0)

    x = torch.empty(B, F, device=device)
    row_const = torch.empty(B, device=device)
    eps = torch.empty(K, B, L, device=device)
    g = torch.Generator(device=device).manual_seed(2)
    n = matrix.number_of_rows

    def new_permutation():
        return torch.randperm(n, generator=g, device=device)
    perm = new_permutation()
    cursor = 0
    step_counter = 0

    def one_step():
        nonlocal perm, cursor, step_counter
        if cursor + GB > n:
            perm = new_permutation()
            cursor = 0
        rows = perm[cursor + rank * B: cursor + (rank + 1) * B]
        cursor += GB
        matrix.gather_dense(rows, out=x, row_const_out=row_const)
        for k in range(K):
            philox_normal(eps[k], row_offset=k * GB + rank * B, seed=1,
                          stream_id=step_counter)
        step_counter += 1
        engine.step(x, x, eps=eps, row_const=row_const, training=True,
                    global_cells=GB)
        if sync is not None:
            sync.all_reduce_gradients()
        engine.adam_step(1e-4)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    scalars = engine.scalars.clone()
    if sync is not None:
        sync.all_reduce_scalars(scalars)
    lower_bound = float(scalars[0].item())

    if rank == 0:
        value = args.steps * GB / elapsed
        from scvae_amd import _lib as _l
        heads = len(_l.LIKELIHOOD_KINDS[args.likelihood][1])
        flops_cell = train_flops_per_cell(F, HIDDEN, L, heads)
        result = {
            "metric": "cells/sec training (68k-PBMC NB-VAE)",
            "value": value,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "68k-PBMC-shaped synthetic counts {}x{} (~5% "
                            "nonzero, device CSR), {} {} hidden 100-100 "
                            "latent {} batch-norm, Adam lr 1e-4".format(
                                args.cells, F, args.likelihood,
                                "GMVAE K={}".format(K) if gm else "VAE", L),
                "cells_per_gpu_per_step": B,
                "global_batch": GB,
                "parallelism": "dp{}".format(world),
            },
            "train_flop_per_cell": flops_cell,
            "step_mfma_frac": value / world * flops_cell / 1e12
            / PEAK_FP32_MFMA_TFLOPS,
            "last_lower_bound": lower_bound,
        }
        if gm:   # informational run: per-cell flops of the VAE formula do not apply
            result["train_flop_per_cell"] = None
            result["step_mfma_frac"] = None
        result["roofline"] = time_dominant_kernel(engine, B * K)
        if world == 1 and not args.no_cpu_baseline and not gm:
            result["cpu_baseline"] = cpu_baseline(matrix, B)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
