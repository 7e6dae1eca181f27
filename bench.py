#!/usr/bin/env python3
"""Throughput benchmark of the scVAE training step on MI355X.

Metric (BASELINE.json): cells/s of NB-VAE training on a 68k-PBMC-shaped count
matrix (68 579 cells x 32 738 genes, synthetic, ~5 % nonzeros), hidden 100-100,
latent 25, batch norm on.  One "step" = one optimiser step over one minibatch:
CSR row gather + densify, encoder/decoder forward, likelihood + KL + ELBO,
backward, (gradient all-reduce), clip + Adam.  Inputs (the CSR matrix) are
resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the script starts one rank per GPU itself (it re-executes under
``python -m torch.distributed.run --nproc-per-node N``, rendezvous on
127.0.0.1) unless it already runs under a launcher (WORLD_SIZE set), so both
``python bench.py --gpus 8`` and ``python -m torch.distributed.run ... bench.py
--gpus 8`` work.  Ranks talk over RCCL (torch backend "nccl").

Prints ONE JSON line on rank 0 (contract in the task description) with extra
objects: ``roofline`` (dominant kernel vs the fp32-MFMA peak, timed live with
HIP events), ``cpu_baseline`` (the torch-CPU fp32 port of the same step,
``oracle/``, on the host cores; rank 0, N = 1 only) and ``other_workloads``
(the other BASELINE.json configurations' models measured in the same run).
"""

import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_CELLS, N_FEATURES = 68579, 32738
HIDDEN, LATENT = (100, 100), 25
LIKELIHOOD = "negative binomial"
PEAK_HBM_TBPS = 8.0          # MI355X_MICROARCH.md: HBM3E spec peak
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense (MI355X_MICROARCH.md); the exact nine-term split of an
                                # fp32 product costs nine bf16 MFMAs: 2500 / 9 = 277.8 TFLOP/s
PEAK_HBM_GBS = 8000.0
MIN_WARM_SECONDS = 1.0          # real steps before the timed window, whatever --warmup says


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096,
                    help="cells per GPU per step (weak scaling) / cells per step over all "
                         "GPUs (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch cells per GPU; strong: --batch cells in total, "
                         "split evenly over the ranks")
    ap.add_argument("--cells", type=int, default=N_CELLS)
    ap.add_argument("--features", type=int, default=N_FEATURES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dd-slabs", action="store_true",
                    help="accumulate the decoder gradient dd through per-strip slabs and a "
                         "fixed-order reduce (bit-repeatable; `scvae train --deterministic`) "
                         "instead of XCD-local fp32 atomics, the library's default")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--head-arith", default=None, choices=["fp32", "bf16x9", "bf16x6"],
                    help="arithmetic of the fused head kernels (default: the library's, "
                         "the exact nine-term bf16 split)")
    ap.add_argument("--model", default="vae", choices=["vae", "gmvae"],
                    help="extra (non-headline) workloads for DESIGN.md")
    ap.add_argument("--clusters", type=int, default=20)
    ap.add_argument("--latent", type=int, default=LATENT)
    ap.add_argument("--likelihood", default=LIKELIHOOD,
                    choices=["poisson", "negative binomial", "zero-inflated poisson",
                             "zero-inflated negative binomial"],
                    help="count likelihoods of the fused decoder-head kernel (the roofline "
                         "probe launches that kernel on its own)")
    return ap.parse_args()


def respawn_one_rank_per_gpu(args):
    """``python bench.py --gpus N`` without a launcher: become the launcher."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


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


# ----------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the torch-CPU fp32 port of the step
# ----------------------------------------------------------------------------
class GpuClocks:
    """Engine / memory clocks as rocm-smi reports them WHILE the warm-up steps run (the
    pool's boxes come in two speeds: without the clocks a reader cannot tell a regression
    from a slow box).  Started before the warm-up (seconds of real steps), read after the
    timed region; never fails the run."""

    def __init__(self):
        import shutil
        import subprocess
        self.out = {"source": "rocm-smi --showclocks --json -d 0, sampled ~0.5 s into the "
                              "warm-up steps (GPU busy with the benchmark's own steps)"}
        self.process = None
        tool = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        try:
            self.process = subprocess.Popen(
                ["/bin/sh", "-c", "sleep 0.3; exec {} --showclocks --json -d 0".format(tool)],
                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        except Exception as error:   # noqa: BLE001 (a report, not a dependency)
            self.out["error"] = repr(error)[:200]

    def read(self):
        if self.process is not None:
            try:
                text, _ = self.process.communicate(timeout=30)
                card = next(iter(json.loads(text).values()))
                # (rocm-smi reports a level and a speed per clock: "sclk clock level:" "1",
                #  "sclk clock speed:" "(2400Mhz)" -- keep what it says under its own key)
                for key, value in card.items():
                    low = key.lower().rstrip(":")
                    if low.split(" ")[0] in ("sclk", "mclk", "fclk", "socclk"):
                        self.out[low.replace(" ", "_")] = str(value).strip("()")
            except Exception as error:   # noqa: BLE001
                self.out["error"] = repr(error)[:200]
                try:
                    self.process.kill()
                except Exception:   # noqa: BLE001
                    pass
            self.process = None
        return self.out


def _cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(matrix, batch):
    """The reference-equivalent CPU path: scipy CSR
    gather + densify (x and t separately, as va:997-998 does), then the
    torch-CPU fp32 port of the step (oracle/models.py).  Two regimes: the
    reference's default minibatch B = 100 (all usable cores and one thread)
    and the benchmark's B.  The intra-op thread count is calibrated first (an
    oversubscribed pool is several times slower than a smaller one)."""
    import numpy
    import scipy.sparse as sp
    import torch
    from oracle import models as om
    cores = len(os.sched_getaffinity(0))
    n_rows = min(matrix.number_of_rows, 2 * batch)
    indptr = matrix.indptr[:n_rows + 1].cpu().numpy()
    nnz = int(indptr[-1])
    host = sp.csr_matrix(
        (matrix.values[:nnz].cpu().numpy(),
         matrix.indices[:nnz].cpu().numpy(), indptr),
        shape=(n_rows, matrix.shape[1]))
    cfg = om.ModelConfig(feature_size=matrix.shape[1], latent_size=LATENT,
                         hidden_sizes=HIDDEN, likelihood=LIKELIHOOD)
    shapes = om.vae_parameter_shapes(cfg)
    rng = numpy.random.RandomState(2)

    def run(b, threads, max_steps, budget, warm):
        """-> (median seconds per step, timed steps)"""
        torch.set_num_threads(threads)
        params = om.init_parameters(shapes, dtype=torch.float32)
        moving = om.init_moving_statistics(shapes, dtype=torch.float32)
        state = om.adam_state(params)
        times, spent = [], 0.0
        for i in range(warm + max_steps):
            idx = rng.permutation(n_rows)[:b]
            t0 = time.perf_counter()
            x = torch.from_numpy(host[idx].toarray())
            t = torch.from_numpy(host[idx].toarray())
            eps = torch.randn(1, b, LATENT)
            params, moving, _, _ = om.vae_train_step(
                cfg, params, moving, state, x, t, eps, 1e-4)
            dt = time.perf_counter() - t0
            if i >= warm:
                times.append(dt)
                spent += dt
                if spent >= budget and len(times) >= 5:
                    break
                if budget > 0 and spent >= 3 * budget and len(times) >= 2:
                    break   # hard cap: the default run must finish within minutes
        return statistics.median(times), len(times)

    # calibrate the pool size on the small regime
    candidates = sorted({c for c in (cores, 128, 64, 32, 16, 8) if c <= cores},
                        reverse=True)
    best_threads, best = candidates[0], None
    for c in candidates:
        sec, _ = run(100, c, 3, 0.0, 1)
        if best is None or sec < best:
            best_threads, best = c, sec
    small_sec, small_n = run(100, best_threads, 50, 8.0, 3)
    single_sec, single_n = run(100, 1, 20, 5.0, 1)
    b = min(batch, n_rows)
    big_sec, big_n = run(b, best_threads, 8, 20.0, 1)
    torch.set_num_threads(cores)
    return {
        "value": b / big_sec,
        "unit": "cells/s",
        "cores": best_threads,
        "kind": "port",
        "sample": "median of {} timed steps of {} cells (same model, same F) after 1 "
                  "warm-up step, torch-CPU fp32 port (oracle/models.py) incl. scipy CSR "
                  "gather+densify x2, {} intra-op threads (best of {} on {} usable cores)"
                  .format(big_n, b, best_threads, candidates, cores),
        "cpu_model": _cpu_model_string(),
        "usable_cores": cores,
        "minibatch_100": {
            "value": 100 / small_sec, "unit": "cells/s", "cores": best_threads,
            "steps": small_n,
            "note": "the reference's default minibatch (defaults.json:51), median"},
        "single_thread_minibatch_100": {
            "value": 100 / single_sec, "unit": "cells/s", "cores": 1,
            "steps": single_n},
    }


# ----------------------------------------------------------------------------
# roofline of the dominant kernel
# ----------------------------------------------------------------------------
DD_ATOMICS = True      # the library's default (scvae_default_dd_atomics: what `scvae train`
                       # runs); main() clears it for --dd-slabs and reads the default back


def _measured_traffic(rows, F, H, kernel, targets="f32"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3
    PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section);
    only reported when the profiled kernel and shapes are the benchmarked ones."""
    names = sorted((n for n in os.listdir(os.path.join(ROOT, "profiles"))
                    if n.endswith(".json") and "pmc_decoder_head" in n),
                   reverse=True)   # newest round first
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            continue
        if (pmc.get("kernel") == "scvae::" + kernel and pmc.get("rows") == rows
                and pmc.get("features") == F and pmc.get("hidden") == H
                and pmc.get("targets", "f32") == targets):
            return pmc["traffic_bytes_per_launch"]
    return None


def time_dominant_kernel(engine, rows, launches=10, u16=False, arith=None):
    """Average duration (HIP events on the launch stream) of the dominant
    kernel of the step -- the fused decoder-head kernel -- run standalone on
    the step's own shapes (main kernel only, without its small reductions,
    so that the figure matches rocprofv3's per-kernel average).  ``u16``: the
    targets as the uint16 minibatch, as the step launches it.  ``arith``: 0 =
    the fp32-MFMA kernel, 1 = the exact nine-term bf16 kernel, None = what the
    engine's plan runs (the arithmetic travels with each call)."""
    import torch
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
    if arith is None:
        arith = {"fp32": 0, "bf16x9": 1, "bf16x6": 2}[engine.head_arith]
    flag = (_lib.HEADS_FP32, _lib.HEADS_BF16X9, _lib.HEADS_BF16X6)[arith]
    if DD_ATOMICS:
        flag |= _lib.HEADS_DD_ATOMICS

    if u16:
        ld = (F + 63) // 64 * 64
        t16 = torch.zeros(rows, ld, dtype=torch.int32, device=dev)
        t16[:, :F] = t.to(torch.int32)
        t16 = t16.to(torch.uint16)

    def launch():
        if u16:
            _lib.check(lib.scvae_decoder_fused_u16(
                kind, 3 | flag, d.data_ptr(), rows, H, aW, ab, adW, adb, F,
                t16.data_ptr(), ld, rows, gw.data_ptr(), rc.data_ptr(),
                ll.data_ptr(), dd.data_ptr(), ws.data_ptr(), stream),
                "scvae_decoder_fused_u16")
            return
        _lib.check(lib.scvae_decoder_fused(
            kind, 3 | flag, d.data_ptr(), rows, H, aW, ab, adW, adb, F, t.data_ptr(),
            rows, gw.data_ptr(), rc.data_ptr(), ll.data_ptr(), dd.data_ptr(),
            ws.data_ptr(), stream), "scvae_decoder_fused")
    which = lib.scvae_decoder_train_kernel(kind, H, arith)
    buf = ctypes.create_string_buffer(128)
    _lib.check(lib.scvae_decoder_train_kernel_name(kind, H, rows, arith, 1 if u16 else 0,
                                                   buf, 128),
               "scvae_decoder_train_kernel_name")
    kernel = buf.value.decode()
    for _ in range(3):
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
    # the instantiation as rocprofv3 prints it
    if which == 3 and arith == 2 and ", 6>" in kernel:
        # (six of the nine terms: the roof of THAT arithmetic; the nine-term roof beside it)
        peak, arith_name = PEAK_BF16_MFMA_TFLOPS / 6.0, "bf16x6"
    elif which == 3:
        peak, arith_name = PEAK_BF16_MFMA_TFLOPS / 9.0, "bf16x9-exact"
    else:
        if which == 2:
            rem = P <= 2 and 96 < H <= 111 and rows >= 512
            kernel = "decoder_head2_kernel<{}, true, {}>".format(
                kind, "true" if rem else "false")
        peak, arith_name = PEAK_FP32_MFMA_TFLOPS, "f32"
    return {
        "kernel": "{} (X_TILDE heads + likelihood + dW/db/dd, "
                  "[rows,{}]x[{},{}]x{} heads)".format(kernel, H, H, F, P),
        "arith": arith_name,
        "bound": "mfma",
        "achieved": flops / seconds / 1e12,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": flops / seconds / 1e12 / peak,
        # the same launch against the fp32 matrix cores' peak (what an fp32-in / fp32-out
        # product can reach without the split): comparable across the two kernels
        "frac_of_fp32_mfma_peak": flops / seconds / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        # ... and against the nine-term roof the earlier rounds' figures were quoted on
        "frac_of_bf16x9_roof": flops / seconds / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 9.0),
        "traffic": _measured_traffic(rows, F, H, kernel, "u16" if u16 else "f32"),
        "targets": "u16" if u16 else "f32",
        "launch_us": seconds * 1e6,
        "algorithmic_flop_per_launch": flops,
    }


def hbm_rooflines(work, matrix, rows):
    """The HBM-bound stages of the step: microseconds from HIP events recorded
    around them inside training steps of the same workload (ten steps run right
    behind the timed ones; scvae_plan_probe_stages, medians), algorithmic bytes (what the stage must move once,
    stated per entry), achieved rate and its fraction of the 8 TB/s HBM peak."""
    engine = work.engine
    F, H = engine.feature_size, engine.hidden_sizes[0]
    ld = (F + 63) // 64 * 64 if work.u16 else F
    xb = 2 if work.u16 else 4
    n = engine.params.numel()
    density = matrix.values.numel() / float(matrix.shape[0] * matrix.shape[1])
    strips = (F + 63) // 64
    entries = [
        ("fetch", "csr_densify{}_kernel (minibatch fetch)".format("_u16" if work.u16 else "_lds"),
         xb * rows * ld + 8.0 * density * rows * F,
         "{} B x rows x ld written + 8 B per stored non-zero read".format(xb)),
        ("count_gemm_fwd", "count_gemm_fwd_kernel + operand split + reduce (x W1 + b)",
         xb * rows * ld + 4.0 * F * H + 4.0 * rows * H,
         "x read once + W1 [F, H] read + [rows, H] written, fp32"),
        ("count_gemm_dw", "count_gemm_dw_kernel + operand split + reduce (x^T dA)",
         xb * rows * ld + 4.0 * F * H + 4.0 * rows * H,
         "x read once + dA [rows, H] read + dW1 [F, H] written, fp32"),
        ("dd_reduce", "dd_reduce_xcd_kernel" if DD_ATOMICS else "dd_reduce_q_kernel",
         4.0 * rows * H * ((8 if DD_ATOMICS else strips) + 1),
         "the partial sums read once ({}) + dd [rows, H] written".format(
             "8 XCD-local accumulators" if DD_ATOMICS else "{} strip slabs".format(strips))),
        ("adam", "adam_clip_kernel", 28.0 * n,
         "28 B per parameter (theta, m, v read + written, gradient read)"),
    ]
    out = []
    for key, kernel, nbytes, note in entries:
        us = work.stage_us.get(key) or []
        if not us:
            continue
        med = statistics.median(us)
        out.append({"kernel": kernel, "algorithmic_bytes": int(nbytes), "launch_us": med,
                    "launches_timed": len(us), "achieved": nbytes / med * 1e-6,
                    "unit": "TB/s", "peak": PEAK_HBM_TBPS,
                    "frac": nbytes / med * 1e-6 / PEAK_HBM_TBPS, "bytes_counted": note})
        # single-process VAE steps of 4096 cells and more run the carried fetch on the plan's
        # second stream BESIDE the input layer's weight gradient (plan.hip, plan_side_fork): the
        # step is shorter for it, each of the two lasts longer than it does alone
        if (key in ("fetch", "count_gemm_dw") and rows >= 4096 and not work.gm
                and work.world == 1 and os.environ.get("SCVAE_SIDE_STREAM", "") != "0"
                and os.environ.get("SCVAE_SIDE_JOBS_AT", "2") == "2"):
            out[-1]["co_runs_with"] = (
                "count_gemm_dw_kernel" if key == "fetch" else "csr_densify_u16_kernel")
            out[-1]["note"] = ("the two share the chip's bandwidth while they overlap: launch_us "
                               "includes the other's traffic (alone: fetch 75-80 us, weight "
                               "gradient 120 us with split + reduce)")
    return out


# ----------------------------------------------------------------------------
# a training workload on one rank
# ----------------------------------------------------------------------------
class Workload:
    """Model + minibatch buffers + the per-step sequence
    (gather/densify -> noise -> step -> [all-reduce] -> clip + Adam)."""

    def __init__(self, matrix, device, batch, likelihood, latent, model="vae",
                 clusters=20, world=1, rank=0, data_parallel=False, head_arith=None):
        import torch
        from scvae_amd.engine import Engine
        self.torch = torch
        self.matrix, self.device = matrix, device
        self.B, self.world, self.rank = batch, world, rank
        self.GB = batch * world
        self.gm = model == "gmvae"
        self.K = clusters if self.gm else 1
        self.L = latent
        F = matrix.shape[1]
        self.engine = Engine(F, latent, HIDDEN, likelihood, batch_norm=True,
                             model_type="GMVAE" if self.gm else "VAE",
                             n_clusters=self.K, device=device, seed=0)
        self.engine.reserve(batch, 1)
        # dd = sum over the gene strips: the plan's default (XCD-local fp32 atomics, what
        # `scvae train` runs; run-to-run the sums differ in their last bits) unless --dd-slabs
        # asks for the bit-repeatable path (`scvae train --deterministic`)
        if not DD_ATOMICS:
            self.engine.set_dd_atomics(False)
        assert self.engine.dd_atomics == DD_ATOMICS
        if head_arith is not None:
            self.engine.set_head_arith(head_arith)
        if os.environ.get("SCVAE_BENCH_COUNT_ALWAYS"):
            self.engine.set_count_gemm(True, always=True)
        self.sync = None
        if data_parallel:
            from scvae_amd.dataparallel import GradientSynchroniser
            self.sync = GradientSynchroniser(self.engine)
            self.sync.broadcast_state(0)
        # integer count matrices: the minibatch is densified as uint16 where the
        # plan takes it (half the bytes for the kernels that stream it;
        # bit-identical step), fp32 otherwise
        self.u16 = bool(matrix.integer_counts
                        and self.engine.accepts_counts_u16(batch, True))
        # two sets of minibatch buffers: a step carries the fetch and the noise of
        # the NEXT one (scvae_side_work: they run on the plan's second stream under
        # the step's backward pass), so that one reads set i while set i ^ 1 fills
        if self.u16:
            self.x = [torch.empty(batch, matrix.u16_pitch, dtype=torch.uint16,
                                  device=device) for _ in range(2)]
        else:
            self.x = [torch.empty(batch, F, device=device) for _ in range(2)]
        self.row_const = [torch.empty(batch, device=device) for _ in range(2)]
        # ... and, opt-in (SCVAE_BENCH_COUNT_TILES=1), next to the uint16 batch its
        # non-zeros in tile-indexed form (scvae_count_tiles): the input layer's two
        # products read those instead of the dense batch -- same arithmetic, a
        # tenth of the bytes; measured in round 6: no faster
        self.use_tiles = bool(self.u16 and matrix.count_tiles_supported
                              and os.environ.get("SCVAE_BENCH_COUNT_TILES") == "1")
        self.tiles = [matrix.count_tiles(batch) if self.use_tiles else None
                      for _ in range(2)]
        self.eps = [torch.empty(self.K, batch, latent, device=device)
                    for _ in range(2)]
        self.generator = torch.Generator(device=device).manual_seed(2)
        self.perm = self._permutation()
        self.cursor = 0
        self.step_counter = 0
        self.slot = 0
        self.primed = False

    def _permutation(self):
        return self.torch.randperm(self.matrix.number_of_rows,
                                   generator=self.generator, device=self.device)

    def _next_rows(self):
        n, B, GB, rank = self.matrix.number_of_rows, self.B, self.GB, self.rank
        if self.cursor + GB > n:
            self.perm = self._permutation()
            self.cursor = 0
        rows = self.perm[self.cursor + rank * B: self.cursor + (rank + 1) * B]
        self.cursor += GB
        return rows

    def _noise(self, slot, step):
        return dict(out=self.eps[slot], block_stride=self.GB,
                    row_offset=self.rank * self.B, seed=1, stream_id=step)

    def one_step(self, comm_events=None):
        """One training step: this step's minibatch and noise are in buffer set
        ``slot`` (fetched by the step before; by hand for the very first one);
        the step carries the fetch + noise of the next minibatch and -- single
        process -- its own clip + Adam update; under data parallel the
        all-reduce comes first and Adam is a launch of its own."""
        from scvae_amd.minibatch import philox_normal_blocks
        B, GB, rank = self.B, self.GB, self.rank
        cur = self.slot
        if not self.primed:
            self.matrix.request(self._next_rows(), self.x[cur],
                                self.row_const[cur], tiles=self.tiles[cur]).issue()
            philox_normal_blocks(self.eps[cur], block_stride=GB,
                                 row_offset=rank * B, seed=1,
                                 stream_id=self.step_counter)
            self.primed = True
        nxt = cur ^ 1
        request = self.matrix.request(self._next_rows(), self.x[nxt],
                                      self.row_const[nxt], tiles=self.tiles[nxt])
        self.step_counter += 1
        self.engine.step(self.x[cur], self.x[cur], eps=self.eps[cur],
                         row_const=self.row_const[cur], training=True,
                         global_cells=GB, row_offset=rank * B,
                         x_counts=self.matrix.integer_counts,
                         learning_rate=1e-4 if self.sync is None else None,
                         next_minibatch=request,
                         next_noise=self._noise(nxt, self.step_counter),
                         count_tiles=self.tiles[cur])
        if self.sync is not None:
            self.sync.all_reduce_gradients(events=comm_events)
            self.engine.adam_step(1e-4)
        self.slot = nxt

    def run(self, steps, warmup, barrier, min_warm_seconds=MIN_WARM_SECONDS):
        """Warm up (>= warmup steps and >= min_warm_seconds of real steps; the
        number of steps is agreed over the ranks by running in chunks between
        barriers), then time exactly ``steps`` steps between barrier + device
        synchronisation.  Per-step durations come from HIP events on the
        launch stream (no synchronisation inside the window)."""
        torch = self.torch
        for _ in range(warmup):
            self.one_step()
        barrier()
        t0 = time.perf_counter()
        warm_steps = warmup
        while True:
            more = time.perf_counter() - t0 < min_warm_seconds
            if self.world > 1:   # rank 0 decides: all ranks leave the loop together
                import torch.distributed as dist
                flag = torch.tensor([1.0 if more else 0.0], device=self.device)
                dist.broadcast(flag, src=0)
                more = flag.item() != 0.0
            if not more:
                break
            for _ in range(8):
                self.one_step()
            warm_steps += 8
            barrier()
        events = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        # (data parallel) events around the waits for the gradient all-reduce: what the
        # step did not hide of the communication
        comm = ([(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                 for _ in range(steps)] if self.sync is not None else None)
        # HIP events around the dominant kernel of every timed step, on the step's stream
        # (scvae_plan_probe_heads): the `roofline` of the line is measured inside the region
        self.engine.probe_heads(steps)
        barrier()
        t0 = time.perf_counter()
        events[0].record()
        for i in range(steps):
            self.one_step(comm[i] if comm else None)
            events[i + 1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        self.head_kernel_ms = self.engine.probe_heads_ms()
        self.engine.probe_heads(0)
        # ... and, in ten further steps OUTSIDE the timed region (ten more event records per step
        # would cost the headline about half a percent), around the step's HBM-bound stages
        self.engine.probe_stages(10)
        for _ in range(10):
            self.one_step()
        barrier()
        self.stage_us = self.engine.probe_stages_us()
        self.engine.probe_stages(0)
        per_step = [events[i].elapsed_time(events[i + 1]) for i in range(steps)]
        self.exposed_comm_ms = ([a.elapsed_time(b) for a, b in comm] if comm else [])
        return elapsed, per_step, warm_steps


def describe(cells, F, likelihood, gm, K, L):
    return ("68k-PBMC-shaped synthetic counts {}x{} (~5% nonzero, device CSR), {} {} "
            "hidden 100-100 latent {} batch-norm, Adam lr 1e-4".format(
                cells, F, likelihood, "GMVAE K={}".format(K) if gm else "VAE", L))


def evaluation_step(matrix, device, batch, steps=60):
    """cells/s of the evaluation step (is_training = False: the epoch-end passes of
    model.train, va:1092-1150 / 1251-1304, and model.evaluate, va:1969-2055) of the headline
    model: minibatch fetch + one graph execution per step, HIP events around the loop."""
    import torch
    from scvae_amd.engine import Engine
    F = matrix.shape[1]
    eng = Engine(F, LATENT, HIDDEN, LIKELIHOOD, batch_norm=True, device=device, seed=0)
    eng.reserve(batch, 1)
    u16 = bool(matrix.integer_counts and eng.accepts_counts_u16(batch, False))
    # (as the model classes run their evaluation passes: two sets of buffers, a step
    #  carries the fetch of the next one -- scvae_side_work, forked after the input layer)
    x = [(torch.empty(batch, matrix.u16_pitch, dtype=torch.uint16, device=device) if u16
          else torch.empty(batch, F, device=device)) for _ in range(2)]
    rc = [torch.empty(batch, device=device) for _ in range(2)]
    eps = torch.randn(1, batch, LATENT, device=device)
    n = matrix.number_of_rows
    rows = torch.arange(n, device=device)

    def request(i):
        r = rows[(i * batch) % (n - batch + 1):][:batch]
        return matrix.request(r, x[i & 1], rc[i & 1])
    request(0).issue()

    def step(i):
        eng.step(x[i & 1], x[i & 1], eps=eps, row_const=rc[i & 1], training=False,
                 x_counts=matrix.integer_counts, next_minibatch=request(i + 1))
    for i in range(10):
        step(i)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10, 10 + steps):
        step(i)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / steps
    # ... and as the epoch-end passes of model.train run it from their second epoch on: the
    # set's uint16 rows are resident (models/base.py: _evaluation_resident), a step reads a view
    resident_ms = None
    if u16:
        blocks = min(n // batch, 4)
        dense = torch.empty(blocks * batch, matrix.u16_pitch, dtype=torch.uint16, device=device)
        constants = torch.empty(blocks * batch, device=device)
        matrix.gather_counts_u16(rows[:blocks * batch], out=dense, row_const_out=constants)

        def resident_step(i):
            lo = (i % blocks) * batch
            view = dense[lo:lo + batch]
            eng.step(view, view, eps=eps, row_const=constants[lo:lo + batch], training=False,
                     x_counts=True)
        for i in range(10):
            resident_step(i)
        torch.cuda.synchronize(device)
        e0.record()
        for i in range(steps):
            resident_step(i)
        e1.record()
        torch.cuda.synchronize(device)
        resident_ms = e0.elapsed_time(e1) / steps
        del dense, constants
    del eng
    torch.cuda.empty_cache()
    return {"workload": "evaluation step (is_training = False) of the headline model: fetch + "
                        "forward + likelihood + ELBO, " + describe(n, F, LIKELIHOOD, False, 1,
                                                                 LATENT),
            "cells_per_step": batch, "steps": steps, "ms_per_step": ms,
            "minibatch_storage": "u16" if u16 else "f32",
            "value": batch / ms * 1e3, "unit": "cells/s",
            "ms_per_step_resident_rows": resident_ms,
            "resident_rows_note": "the same step on rows already dense on the device (what the "
                                  "epoch-end passes of model.train read after the first epoch)"}


def categorised_step(matrix, device, batch, k, steps=12):
    """The headline model with the piecewise categorical likelihood `-k` (distributions/
    categorised.py:255-263, va:2507-2532): a training step (fp32 minibatch, forward + backward +
    clip + Adam; the fetch is not part of it) on the fused kernels -- two launches of the bf16x9
    head kernel -- and, beside it, on the unfused ones (GEMM + element-wise kernels)."""
    import torch
    from scvae_amd.engine import Engine
    F = matrix.shape[1]
    rows = torch.arange(batch, device=device)
    rc = torch.empty(batch, device=device)
    x = matrix.gather_dense(rows, row_const_out=rc)
    eps = torch.randn(1, batch, LATENT, device=device)
    out = {}
    for fused in (True, False):
        eng = Engine(F, LATENT, HIDDEN, LIKELIHOOD, batch_norm=True, device=device, seed=0,
                     k_max=k)
        eng.set_fused(fused)
        eng.reserve(batch, 1)
        assert eng.fused_categorised == fused
        for _ in range(3):
            eng.step(x, x, eps=eps, row_const=rc, training=True, x_counts=True)
            eng.adam_step(1e-4)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            eng.step(x, x, eps=eps, row_const=rc, training=True, x_counts=True)
            eng.adam_step(1e-4)
        e1.record()
        torch.cuda.synchronize(device)
        out["fused" if fused else "unfused"] = e0.elapsed_time(e1) / steps
        del eng
        torch.cuda.empty_cache()
    del x
    torch.cuda.empty_cache()
    return {"workload": "headline model with -k {} (piecewise categorical likelihood), training "
                        "step without the fetch, {} cells: ".format(k, batch)
                        + describe(matrix.shape[0], F, LIKELIHOOD, False, 1, LATENT),
            "cells_per_step": batch, "steps": steps, "ms_per_step": out["fused"],
            "ms_per_step_unfused_kernels": out["unfused"],
            "value": batch / out["fused"] * 1e3, "unit": "cells/s"}


def model_train_epoch(matrix, device, batch, epochs):
    """The drop-in's own entry point: ``VariationalAutoencoder.train`` (the Python epoch loop of
    va:958-1599 -- shuffled minibatches, one step call each, the printed lines, BOTH epoch-end
    evaluation passes over the training and the validation set, early-stopping bookkeeping, the
    checkpoint written in the background) on the benchmark's matrix as a DataSet; wall-clock
    between the ends of consecutive epochs (the first epoch -- allocations, warm-up -- is not
    counted), cells of the training set per second."""
    import contextlib
    import tempfile
    import numpy
    import scipy.sparse
    import torch
    from scvae_amd.data import DataSet
    from scvae_amd.models import VariationalAutoencoder
    n, F = matrix.shape
    host = scipy.sparse.csr_matrix(
        (matrix.values.cpu().numpy(), matrix.indices.cpu().numpy(),
         matrix.indptr.cpu().numpy()), shape=(n, F))
    n_valid = n // 10
    feature_names = numpy.array(["g%d" % j for j in range(F)])

    def data_set(values, kind, first):
        return DataSet("bench_shaped", values=values, kind=kind, feature_names=feature_names,
                       example_names=numpy.array(
                           ["c%d" % i for i in range(first, first + values.shape[0])]))
    training = data_set(host[:n - n_valid], "training", 0)
    validation = data_set(host[n - n_valid:], "validation", n - n_valid)
    ends = []

    def epoch_done(**kwargs):
        torch.cuda.synchronize(device)
        ends.append(time.perf_counter())
    with tempfile.TemporaryDirectory() as directory:
        model = VariationalAutoencoder(
            feature_size=F, latent_size=LATENT, hidden_sizes=list(HIDDEN),
            reconstruction_distribution=LIKELIHOOD, log_directory=directory,
            device=str(device))
        with open(os.devnull, "w") as sink, contextlib.redirect_stdout(sink):
            model.train(training, validation, number_of_epochs=epochs, minibatch_size=batch,
                        learning_rate=1e-4, intermediate_analyser=epoch_done)
        dd_atomics = model.engine.dd_atomics
        del model
    torch.cuda.empty_cache()
    spans = [b - a for a, b in zip(ends[:-1], ends[1:])]
    sec = statistics.median(spans)
    n_train = n - n_valid
    steps = -(-n_train // batch)
    return {"workload": "VariationalAutoencoder.train, one epoch: {} training steps of {} cells "
                        "+ evaluation of the training set ({} cells) and of the validation set "
                        "({} cells) + checkpoint; ".format(steps, batch, n_train, n_valid)
                        + describe(n, F, LIKELIHOOD, False, 1, LATENT),
            "cells_per_step": batch, "epochs_timed": len(spans), "seconds_per_epoch": sec,
            "value": n_train / sec, "unit": "training cells/s (epoch wall-clock, evaluations "
                                            "and checkpoint included)",
            "dd_accumulation": "atomics" if dd_atomics else "slabs"}


def other_workloads(matrix, device, barrier):
    """The other BASELINE.json configurations' models, measured in the same
    run (rank 0, N = 1): cells/s of the same step sequence."""
    import torch
    from scvae_amd.minibatch import synthetic_count_matrix
    out = {}

    def measure(key, note, mat, batch, likelihood, latent, model, steps, head_arith=None):
        w = Workload(mat, device, batch, likelihood, latent, model=model, clusters=20,
                     head_arith=head_arith)
        elapsed, per_step, _ = w.run(steps, 2, barrier, min_warm_seconds=0.3)
        out[key] = {
            "workload": note,
            "cells_per_step": batch,
            "steps": steps,
            "ms_per_step": elapsed / steps * 1e3,
            "step_ms_median": statistics.median(per_step),
            "value": steps * batch / elapsed,
            "unit": "cells/s",
            "last_lower_bound": float(w.engine.scalars[0].item()),
        }
        del w
        torch.cuda.empty_cache()

    n, F = matrix.shape
    measure("headline_model_minibatch_100",
            "cfg2/headline model at the reference's default minibatch: " +
            describe(n, F, LIKELIHOOD, False, 1, LATENT),
            matrix, 100, LIKELIHOOD, LATENT, "vae", 200)
    # the minibatch regimes SURVEY.md section 8d names besides 4096, and the per-rank shard of a
    # strong-scaled 4096-cell step on eight GPUs (512)
    for b, steps in ((512, 100), (1024, 60), (16384, 8)):
        measure("headline_model_minibatch_{}".format(b),
                "headline model, {} cells per step: ".format(b) +
                describe(n, F, LIKELIHOOD, False, 1, LATENT),
                matrix, b, LIKELIHOOD, LATENT, "vae", steps)
    measure("cfg3_zinb_vae_latent_100",
            describe(n, F, "zero-inflated negative binomial", False, 1, 100),
            matrix, 4096, "zero-inflated negative binomial", 100, "vae", 20)
    measure("cfg4_nb_gmvae_k20_latent_100",
            describe(n, F, "negative binomial", True, 20, 100),
            matrix, 512, "negative binomial", 100, "gmvae", 10)
    # cfg5's gene count (10x 1.3M mouse brain: 27 998 genes); rows are a sample,
    # the step only ever sees one minibatch
    m5, _ = synthetic_count_matrix(16384, 27998, density=0.05, seed=61, device=device)
    note5 = ("1.3M-mouse-brain-shaped synthetic counts (16384-row sample)x27998, "
             "zero-inflated negative binomial GMVAE K=20 hidden 100-100 latent 100")
    measure("cfg5_zinb_gmvae_k20_latent_100_f27998", note5,
            m5, 512, "zero-inflated negative binomial", 100, "gmvae", 10)
    out["evaluation_step"] = evaluation_step(matrix, device, 4096)
    out["headline_model_k1_categorised"] = categorised_step(matrix, device, 4096, 1)
    for b, epochs in ((4096, 4), (100, 3)):
        out["model_train_epoch_b{}".format(b)] = model_train_epoch(matrix, device, b, epochs)
    # Opt-in arithmetic, NOT the headline: the same steps with the heads' products as six of the
    # nine bf16 terms (Engine.set_head_arith('bf16x6'): the three smallest products, together
    # <= 2^-26 of a product with the rounded split, left out -- fp32-class, not exact;
    # tests/test_gpu_as_benched.py holds both forms to the same tolerances against fp64).
    # Every figure above this comment and the headline `value` run the exact nine-term form.
    six = " [head arithmetic bf16x6, opt-in]"
    measure("optin_bf16x6_headline_model",
            "headline model, 4096 cells per step" + six + ": " +
            describe(n, F, LIKELIHOOD, False, 1, LATENT),
            matrix, 4096, LIKELIHOOD, LATENT, "vae", 20, head_arith="bf16x6")
    measure("optin_bf16x6_cfg3_zinb_vae_latent_100",
            describe(n, F, "zero-inflated negative binomial", False, 1, 100) + six,
            matrix, 4096, "zero-inflated negative binomial", 100, "vae", 20,
            head_arith="bf16x6")
    measure("optin_bf16x6_cfg4_nb_gmvae_k20_latent_100",
            describe(n, F, "negative binomial", True, 20, 100) + six,
            matrix, 512, "negative binomial", 100, "gmvae", 10, head_arith="bf16x6")
    measure("optin_bf16x6_cfg5_zinb_gmvae_k20_latent_100_f27998", note5 + six,
            m5, 512, "zero-inflated negative binomial", 100, "gmvae", 10,
            head_arith="bf16x6")
    return out


def main():
    global DD_ATOMICS
    args = parse_args()
    DD_ATOMICS = not args.dd_slabs
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_one_rank_per_gpu(args))

    # (the host driver only supports dmabuf IPC: RCCL across processes needs this; set before
    #  the runtime loads, also when a launcher other than respawn_one_rank_per_gpu started us)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but the launcher started {} ranks".format(
            args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback).")
    # (several ranks may share a GPU in the 1-GPU debugging set-up below)
    local_device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_device)
    device = torch.device("cuda", local_device)

    import torch.distributed as dist
    backend, ranks_seen = None, 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL; SCVAE_BENCH_BACKEND=gloo only to exercise the N > 1 code path on a 1-GPU box
        backend = os.environ.get("SCVAE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        # one all-reduce over the communicator: how many ranks RCCL really connects
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    from scvae_amd.minibatch import synthetic_count_matrix
    # (the library is loaded after torch: both must share torch's HIP runtime)
    from scvae_amd import _lib as _l0
    DD_ATOMICS = bool(_l0.load().scvae_default_dd_atomics()) and not args.dd_slabs

    matrix, _ = synthetic_count_matrix(
        args.cells, args.features, density=0.05, seed=60, device=device)
    F, B, L = args.features, args.batch, args.latent
    if args.scaling == "strong":
        if B % world:
            raise SystemExit("--scaling strong: --batch {} does not split over {} ranks"
                             .format(B, world))
        B //= world
    gm = args.model == "gmvae"
    K = args.clusters if gm else 1
    work = Workload(matrix, device, B, args.likelihood, L, model=args.model,
                    clusters=args.clusters, world=world, rank=rank,
                    data_parallel=world > 1, head_arith=args.head_arith)
    GB = work.GB

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    clock_probe = GpuClocks() if rank == 0 else None
    elapsed, per_step, warm_steps = work.run(args.steps, args.warmup, barrier)
    clocks = clock_probe.read() if clock_probe else None
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # per-rank figures: median step time and the communication the step did not hide
    rank_ms = [statistics.median(per_step)]
    rank_comm = [statistics.median(work.exposed_comm_ms) if work.exposed_comm_ms else 0.0]
    if world > 1:
        # (gloo -- the 1-GPU debugging set-up -- gathers host tensors only)
        mine = torch.tensor([rank_ms[0], rank_comm[0]], dtype=torch.float64,
                            device=device if backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t[0]) for t in every]
        rank_comm = [float(t[1]) for t in every]

    engine = work.engine
    scalars = engine.scalars.clone()
    if work.sync is not None:
        work.sync.all_reduce_scalars(scalars)
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            # (decoder_head_arith, below: the heads' products as exact nine-term bf16 splits
            #  with fp32 accumulation where that kernel applies)
            # x W1 and x^T dA of the input layer: exact hi/lo bf16 cut of the integer
            # counts times an exact three-term bf16 split of the fp32 operand, fp32
            # accumulation (count_gemm.hip); everything else fp32 MFMA / VALU
            "encoder_input_arith": ("bf16x3-exact" if matrix.integer_counts
                                    else "f32"),
            # storage of the dense minibatch between the CSR gather and its three
            # readers (integer counts: uint16, exact; the arithmetic stays fp32)
            "minibatch_storage": "u16" if work.u16 else "f32",
            # the decoder gradient dd = sum over the gene strips of G W^T
            "dd_accumulation": ("xcd-local fp32 atomics (the library default, what `scvae "
                                "train` runs; not bit-repeatable)"
                                if DD_ATOMICS else "per-strip slabs + fixed-order reduce "
                                "(`scvae train --deterministic`)"),
            "data": "synthetic",
            "config": {
                "workload": describe(args.cells, F, args.likelihood, gm, K, L),
                "cells_per_gpu_per_step": B,
                "global_batch": GB,
                "parallelism": "dp{}".format(world),
            },
            "collective_backend": ("rccl (torch 'nccl')" if backend == "nccl"
                                   else backend),
            "ranks_in_communicator": ranks_seen,
            "rank_step_ms_median": rank_ms,
            # per rank, median over the timed steps: time between the first and the last wait
            # for the gradient all-reduce (three pieces, two of them issued under the backward
            # pass) -- the part of the collective the step did not hide
            "rank_exposed_allreduce_ms_median": rank_comm,
            "warm_steps_before_timing": warm_steps,
            "step_ms_min": min(per_step),
            "step_ms_median": statistics.median(per_step),
            "step_ms_max": max(per_step),
            "train_flop_per_cell": flops_cell,
            "step_mfma_frac": value / world * flops_cell / 1e12
            / PEAK_FP32_MFMA_TFLOPS,
            "last_lower_bound": lower_bound,
            "gpu_clocks": clocks,
        }
        if gm:   # informational run: per-cell flops of the VAE formula do not apply
            result["train_flop_per_cell"] = None
            result["step_mfma_frac"] = None
        result["roofline"] = time_dominant_kernel(engine, B * K, u16=work.u16)
        in_step = getattr(work, "head_kernel_ms", [])
        if in_step:
            # the same kernel timed INSIDE the timed steps (HIP events on the launch stream
            # around every launch): this is the figure rocprofv3's per-kernel average of the
            # same command agrees with; the standalone loop above stays for comparison
            roof = result["roofline"]
            seconds = sum(in_step) / len(in_step) / 1e3
            flops = roof["algorithmic_flop_per_launch"]
            roof["launch_us_standalone"] = roof["launch_us"]
            roof["launch_us"] = seconds * 1e6
            roof["launches_timed"] = len(in_step)
            roof["achieved"] = flops / seconds / 1e12
            roof["frac"] = roof["achieved"] / roof["peak"]
            roof["frac_of_fp32_mfma_peak"] = roof["achieved"] / PEAK_FP32_MFMA_TFLOPS
            roof["frac_of_bf16x9_roof"] = roof["achieved"] / (PEAK_BF16_MFMA_TFLOPS / 9.0)
        # arithmetic of the decoder heads' three products in the training kernel
        result["decoder_head_arith"] = result["roofline"]["arith"]
        if result["roofline"]["arith"] != "f32":
            # (the arithmetic type the path computes in: fp32 values, fp32 accumulation; the
            #  heads' products are issued as exact bf16 terms on the bf16 matrix cores)
            result["dtype"] = "f32 ({} heads)".format(result["roofline"]["arith"])
            # the fp32-MFMA kernel on the same shapes, for comparison (not what the step ran)
            result["roofline_fp32_kernel"] = time_dominant_kernel(
                engine, B * K, u16=work.u16, arith=0)
        headline = (not gm and args.likelihood == LIKELIHOOD and L == LATENT)
        if rank == 0 and not gm:
            # the HBM-bound kernels of the step (achieved bytes / peak)
            result["roofline_hbm"] = hbm_rooflines(work, matrix, B)
        if world == 1 and headline and not args.no_other_workloads:
            result["other_workloads"] = other_workloads(matrix, device, barrier)
        if world == 1 and not args.no_cpu_baseline and headline:
            result["cpu_baseline"] = cpu_baseline(matrix, B)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        from scvae_amd.dataparallel import release_gradient_groups
        release_gradient_groups()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
