"""Data-parallel training over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on ROCm).

The reference is single-process; the build adds exactly
one form of parallelism: the global minibatch is split into equal contiguous
row shards, every rank holds a full replica of the weights and Adam slots, and
per optimiser step there is

* one all-reduce(sum) of the flat fp32 gradient buffer (each rank's gradient is
  already scaled by 1/global_batch, so the sum is the single-process gradient;
  clipping to [-1, 1] happens after it, as in va:2751-2755), issued in three
  pieces: the likelihood heads (two thirds of the buffer) right after their
  kernel, the hidden layers when the backward pass reaches the first encoder
  layer -- both asynchronously, under the rest of the backward pass -- and the
  first encoder layer after the step, and
* per batch-norm layer one all-gather of ``[mean | var]`` in the forward pass
  (merged with the parallel-variance formula) and one all-reduce of
  ``[sum dA | sum dA*xhat]`` in the backward pass, so that the result equals
  single-process batch norm over the whole minibatch (sync batch norm).

Tests: ``tests/test_dataparallel_cpu.py`` (gloo, world size 2, no GPU) covers
``shard_bounds``, ``merge_batch_norm_statistics`` and the same sequence of
collectives with plain fp64 torch as the compute stand-in;
``tests/test_dataparallel_sync_cpu.py`` (gloo, world size 2) drives
``GradientSynchroniser`` on a stand-in engine: hook kinds 0 and 2, every
gradient element summed exactly once around the ranges announced early, state
broadcast, refused ranges; ``tests/test_gpu_dataparallel.py`` drives it with the
real kernels (two ranks on one GPU, gloo moving the bytes, and a single-rank
RCCL group).  RCCL between several physical GPUs is only run by the
driver's scaling benchmark.
"""

import ctypes

import torch
import torch.distributed as dist


def shard_bounds(global_rows, world_size, rank):
    """Contiguous equal shards; the global minibatch must divide evenly."""
    if global_rows % world_size:
        raise ValueError(
            "Global minibatch of {} rows does not split evenly over {} ranks."
            .format(global_rows, world_size))
    per_rank = global_rows // world_size
    return rank * per_rank, (rank + 1) * per_rank


def merge_batch_norm_statistics(gathered, counts):
    """Chan et al. merge. ``gathered``: [ranks, 2, n] (mean, biased var) per
    rank; ``counts``: [ranks].  Returns (mean[n], var[n]) of the union."""
    counts = counts.to(gathered.dtype).view(-1, 1)
    total = counts.sum()
    mean = (gathered[:, 0] * counts).sum(dim=0) / total
    delta = gathered[:, 0] - mean
    m2 = (counts * (gathered[:, 1] + delta * delta)).sum(dim=0)
    return mean, m2 / total


# One gradient communicator per (backend, ranks) and process: ``dist.new_group`` is a collective
# that allocates an RCCL communicator which is never freed, and ``model.train`` builds a new
# synchroniser on every call.
_GRADIENT_GROUPS = {}


def gradient_group_for(group=None):
    """The communicator the asynchronous gradient all-reduces run on -- distinct
    from ``group``, created once per process and set of ranks.  COLLECTIVE over
    the default group on first use (``dist.new_group`` must be entered by every
    process of the default group, members or not): construct synchronisers at
    the same point on every rank, or create the group up front and pass it as
    ``gradient_group``."""
    ranks = tuple(dist.get_process_group_ranks(group) if group is not None
                  else range(dist.get_world_size()))
    # (the identity of the default process group is part of the key: after
    #  destroy_process_group() + init_process_group() in one process -- tests, notebooks,
    #  elastic restarts -- a communicator cached under the old world is stale)
    world = dist.group.WORLD
    key = (dist.get_backend(group), ranks, id(world))
    for stale in [k for k in _GRADIENT_GROUPS if k[2] != id(world)]:
        del _GRADIENT_GROUPS[stale]
    if key not in _GRADIENT_GROUPS:
        _GRADIENT_GROUPS[key] = (world, dist.new_group(ranks=list(ranks),
                                                       backend=key[0]))
    return _GRADIENT_GROUPS[key][1]


def release_gradient_groups():
    """Destroy the cached gradient communicators (before
    ``dist.destroy_process_group()`` at the end of a job)."""
    for _, g in _GRADIENT_GROUPS.values():
        try:
            dist.destroy_process_group(g)
        except Exception:   # the default group is already gone
            pass
    _GRADIENT_GROUPS.clear()


class GradientSynchroniser:
    """Collectives of one data-parallel rank, bound to an ``Engine``."""

    def __init__(self, engine, group=None, gradient_group=None):
        from scvae_amd import _lib
        self.engine = engine
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # The asynchronous gradient all-reduces get a communicator of their own: collectives
        # of one communicator execute in issue order on its stream, so on a shared one the
        # small, latency-critical batch-norm exchanges of the backward pass would queue
        # behind the 26 MB gradient bucket issued just before them.  (Collective call: every
        # rank constructs its synchroniser at the same point.)
        if gradient_group is None and self.world_size > 1:
            gradient_group = gradient_group_for(group)
        self.gradient_group = gradient_group if gradient_group is not None else group
        self.lib = _lib.load()
        self._check = _lib.check
        self._gathered = None
        self._counts = None
        self._pending = []   # (offset, count, work) of gradient ranges reduced early
        engine.set_sync(self._hook)

    def _view(self, address, count):
        ws = self.engine.workspace
        offset = address - ws.data_ptr()
        if offset < 0 or offset + 4 * count > ws.numel():
            raise RuntimeError("sync buffer is outside the bound workspace")
        return ws[offset:offset + 4 * count].view(torch.float32)

    def _hook(self, user, address, count, kind, local_rows):
        try:
            if kind == 2:
                # the tail of the gradient buffer is final: start its all-reduce
                # now, under the last large GEMM of the backward pass
                grads = self.engine.grads
                offset = (address - grads.data_ptr()) // 4
                if offset < 0 or offset + count > grads.numel():
                    raise RuntimeError("gradient range is outside the buffer")
                work = dist.all_reduce(grads[offset:offset + count],
                                       group=self.gradient_group, async_op=True)
                self._pending.append((offset, count, work))
                return 0
            view = self._view(address, count)
            if kind == 0:
                dist.all_reduce(view, group=self.group)
                return 0
            n = count // 2
            need = self.world_size * count
            if self._gathered is None or self._gathered.numel() < need:
                self._gathered = torch.empty(
                    need, dtype=torch.float32, device=view.device)
            if self._counts is None:
                self._counts = torch.empty(
                    self.world_size, dtype=torch.int64, device=view.device)
            # equal shards (shard_bounds): every rank contributes local_rows
            self._counts.fill_(int(local_rows))
            gathered = self._gathered[:need]
            if view.is_cuda and dist.get_backend(self.group) == "gloo":
                # gloo has no all-gather for device tensors: all-reduce a buffer
                # that is zero except for this rank's slot (debugging set-up:
                # several ranks on one GPU)
                gathered.zero_()
                gathered[self.rank * count:(self.rank + 1) * count] = view
                dist.all_reduce(gathered, group=self.group)
            else:
                dist.all_gather_into_tensor(gathered, view, group=self.group)
            from scvae_amd.engine import current_stream_handle
            self._check(self.lib.scvae_bn_merge(
                ctypes.c_void_p(gathered.data_ptr()),
                ctypes.c_void_p(self._counts.data_ptr()), self.world_size, n,
                ctypes.c_void_p(view.data_ptr()),
                current_stream_handle(view.device)), "scvae_bn_merge")
            return 0
        except Exception as error:  # surfaced by the C side as rc=-2
            print("[scvae_amd] sync hook failed:", repr(error), flush=True)
            return 1

    def all_reduce_gradients(self, events=None):
        """Sum the gradient buffer over the ranks: the ranges announced early by
        the step (hook kind 2) are already in flight, the rest is reduced here.
        ``events``: an optional pair of CUDA events recorded on the compute stream
        before the first and after the last wait -- the time between them is the
        communication the step did not hide."""
        grads = self.engine.grads
        if events is not None:
            events[0].record()
        position = 0
        for offset, count, _ in sorted(self._pending, key=lambda p: p[0]):
            if offset > position:
                dist.all_reduce(grads[position:offset],
                                group=self.gradient_group)
            position = max(position, offset + count)
        if position < grads.numel():
            dist.all_reduce(grads[position:], group=self.gradient_group)
        for _, _, work in self._pending:
            work.wait()
        self._pending = []
        if events is not None:
            events[1].record()

    def all_reduce_scalars(self, scalars):
        dist.all_reduce(scalars, group=self.group)
        return scalars

    def broadcast_state(self, src=0):
        """Weights, Adam slots, moving statistics and the Adam step count of
        rank ``src`` on every rank (the step count sets lr_t: replicas with
        different counts would silently diverge)."""
        for tensor in (self.engine.params, self.engine.adam_m,
                       self.engine.adam_v, self.engine.moving):
            dist.broadcast(tensor, src=src, group=self.group)
        step_count = torch.tensor([float(self.engine.adam_t)],
                                  dtype=torch.float64,
                                  device=self.engine.params.device)
        dist.broadcast(step_count, src=src, group=self.group)
        self.engine.adam_t = int(step_count.item())
