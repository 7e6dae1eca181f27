"""``GradientSynchroniser`` itself on CPU (gloo, world size 2): the hook the C
side calls during a step (kinds 0 and 2 -- the batch-norm backward sums and the
gradient ranges announced early), ``all_reduce_gradients`` over the ranges that
are left, ``broadcast_state``, and the error path.  The engine is a stand-in
that owns the buffers the synchroniser touches; the real kernels (and hook kind
1, whose merge is a device kernel) run in ``tests/test_gpu_dataparallel.py``.

The property that matters: every element of the gradient buffer is summed over
the ranks exactly once, whatever ranges the step announced early
(``scvae_amd/dataparallel.py``; the reference is single-process: va:2751-2755
clips what here is the all-reduced sum)."""
import datetime
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

RENDEZVOUS_FAILED = "rendezvous failed"
N_GRADS = 300


class _Engine:
    """What GradientSynchroniser needs of an Engine."""

    def __init__(self, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.workspace = torch.zeros(4096, dtype=torch.uint8)
        self.grads = torch.arange(N_GRADS, dtype=torch.float32) + 1000.0 * rank
        self.params = torch.randn(50, generator=g)
        self.adam_m = torch.randn(50, generator=g)
        self.adam_v = torch.rand(50, generator=g)
        self.moving = torch.randn(20, generator=g)
        self.adam_t = 7 + 3 * rank
        self.hook = None

    def set_sync(self, hook):
        self.hook = hook


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=60))
    except Exception as error:   # the port was taken between pick and bind
        out.put((RENDEZVOUS_FAILED, rank, repr(error)))
        return
    try:
        from scvae_amd.dataparallel import GradientSynchroniser, release_gradient_groups
        engine = _Engine(rank)
        sync = GradientSynchroniser(engine)
        assert engine.hook is not None and sync.world_size == world
        assert sync.gradient_group is not None and sync.gradient_group is not sync.group
        report = {}
        # kind 0: sums of the batch-norm backward pass, 24 floats at byte 256 of the workspace
        view = engine.workspace[256:256 + 4 * 24].view(torch.float32)
        view.copy_(torch.arange(24, dtype=torch.float32) * (rank + 1))
        address = engine.workspace.data_ptr() + 256
        assert engine.hook(None, address, 24, 0, 12) == 0
        report["sums"] = view.tolist()
        # ... outside the workspace: refused (rc 1 -> the C side fails the step), nothing hangs
        assert engine.hook(None, engine.workspace.data_ptr() + 4096 - 8, 24, 0, 12) == 1
        assert engine.hook(None, engine.grads.data_ptr() + 4 * (N_GRADS - 3), 8, 2, 12) == 1
        # kind 2: two ranges announced early, in the order the backward pass reaches them
        base = engine.grads.data_ptr()
        assert engine.hook(None, base + 4 * 200, 100, 2, 12) == 0     # the tail (the heads)
        assert engine.hook(None, base + 4 * 40, 60, 2, 12) == 0       # a middle piece
        assert len(sync._pending) == 2
        sync.all_reduce_gradients()                                    # [0, 40), [100, 200) + waits
        assert sync._pending == []
        report["grads"] = engine.grads.tolist()
        # a step without early ranges: the whole buffer at once
        engine.grads.fill_(float(rank + 1))
        sync.all_reduce_gradients()
        report["plain"] = engine.grads.tolist()
        scalars = sync.all_reduce_scalars(torch.tensor([1.0 + rank, 10.0]))
        report["scalars"] = scalars.tolist()
        sync.broadcast_state(src=0)
        report["state"] = (engine.params.tolist(), engine.adam_m.tolist(), engine.adam_v.tolist(),
                           engine.moving.tolist(), engine.adam_t)
        release_gradient_groups()
        out.put((rank, report))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_gradient_synchroniser_sums_every_element_once():
    ctx = mp.get_context("spawn")
    reports = None
    for attempt in range(3):
        out = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        got = [out.get(timeout=180) for _ in range(2)]
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
                p.join()
                raise AssertionError("a rank did not exit")
        if any(isinstance(g[0], str) and g[0] == RENDEZVOUS_FAILED for g in got):
            continue
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        reports = dict(got)
        break
    assert reports is not None, "the gloo rendezvous failed three times"
    want_grads = (2.0 * torch.arange(N_GRADS, dtype=torch.float32) + 1000.0).tolist()
    reference = _Engine(0)
    for rank in (0, 1):
        r = reports[rank]
        assert r["sums"] == (3.0 * torch.arange(24, dtype=torch.float32)).tolist()
        assert r["grads"] == want_grads            # each element: rank 0's + rank 1's, once
        assert r["plain"] == [3.0] * N_GRADS
        assert r["scalars"] == [3.0, 20.0]
        params, adam_m, adam_v, moving, adam_t = r["state"]
        assert params == reference.params.tolist() and adam_m == reference.adam_m.tolist()
        assert adam_v == reference.adam_v.tolist() and moving == reference.moving.tolist()
        assert adam_t == reference.adam_t
