"""Device-side engine of a model: one ``scvae_plan`` plus the torch buffers it
is bound to.

This is the replacement of the reference's ``tf.Graph`` + ``tf.Session``
(``scvae/models/variational_autoencoder.py:310-410, 887``): parameters, Adam
slots, batch-norm moving statistics and the activation workspace live in flat
fp32 device buffers allocated through torch (plumbing only); all arithmetic is
done by the HIP kernels behind ``libscvae_hip.so``.
"""

import ctypes
import math
from collections import OrderedDict

import torch

from scvae_amd import _lib

ADAM_BETA1 = 0.9
ADAM_BETA2 = 0.999
ADAM_EPSILON = 1e-8

SCALAR_NAMES = ("lower_bound", "lower_bound_weighted", "reconstruction_error",
                "kl_divergence", "kl_divergence_y")


def _ptr(tensor):
    return ctypes.c_void_p(tensor.data_ptr()) if tensor is not None else None


def current_stream_handle(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class PendingState(dict):
    """A state dictionary whose tensors are still on their way to the host
    (``Engine.state_dict(non_blocking=True)``): complete after ``wait()``."""
    ready = None

    def wait(self):
        if self.ready is not None:
            self.ready.synchronize()
            self.ready = None
        return self


class Engine:
    """Flat-buffer model state + kernel plan on one GPU."""

    def __init__(self, feature_size, latent_size, hidden_sizes, likelihood,
                 batch_norm=True, model_type="VAE", n_clusters=1,
                 kl_weight=1.0, free_nats_proportion=0.0, device=None,
                 seed=0, decoder_extra=0, k_max=0,
                 prior_probabilities_method="uniform",
                 prior_probabilities=None, inference_architecture="MLP",
                 generative_architecture="MLP",
                 latent_distribution="gaussian", analytical_kl_term=True,
                 dropout_keep_probabilities=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError(
                "No GPU visible: the scVAE engine has no CPU path.")
        self.device = torch.device(device if device is not None else "cuda:0")
        kind, head_names = _lib.LIKELIHOOD_KINDS[likelihood]
        self.likelihood = likelihood
        self.head_names = head_names
        self.model_type = model_type
        self.feature_size = int(feature_size)
        self.latent_size = int(latent_size)
        self.hidden_sizes = [int(h) for h in hidden_sizes]
        self.n_clusters = int(n_clusters)
        self.decoder_extra = int(decoder_extra)
        self.k_max = int(k_max or 0)
        self.prior_probabilities_method = prior_probabilities_method
        self.prior_probabilities = (
            None if prior_probabilities is None
            else [float(p) for p in prior_probabilities])
        if prior_probabilities_method == "custom" and (
                self.prior_probabilities is None
                or len(self.prior_probabilities) != self.n_clusters):
            raise ValueError("custom prior probabilities: one per cluster")
        if len(self.hidden_sizes) > _lib.MAX_HIDDEN:
            raise ValueError("At most {} hidden layers are supported.".format(
                _lib.MAX_HIDDEN))

        cfg = _lib.ModelConfig()
        cfg.model_type = (_lib.MODEL_GMVAE if model_type == "GMVAE"
                          else _lib.MODEL_VAE)
        cfg.feature_size = self.feature_size
        cfg.latent_size = self.latent_size
        cfg.n_hidden = len(self.hidden_sizes)
        for i, h in enumerate(self.hidden_sizes):
            cfg.hidden[i] = h
        cfg.likelihood = kind
        cfg.batch_norm = 1 if batch_norm else 0
        cfg.n_clusters = self.n_clusters
        cfg.kl_weight = float(kl_weight)
        cfg.free_nats_proportion = float(free_nats_proportion)
        cfg.decoder_extra = self.decoder_extra
        cfg.k_max = self.k_max
        cfg.prior_mode = {"uniform": 0, "custom": 1, "learn": 2}[
            prior_probabilities_method]
        cfg.linear_factor = (
            (1 if inference_architecture.upper() == "LFM" else 0)
            | (2 if generative_architecture.upper() == "LFM" else 0))
        if latent_distribution not in (
                "gaussian", "unit-variance gaussian", "gaussian mixture",
                "legacy gaussian mixture"):
            raise ValueError("Latent distribution `{}`.".format(
                latent_distribution))
        self.latent_distribution = latent_distribution
        self.analytical_kl_term = bool(analytical_kl_term)
        cfg.latent_mode = 0
        if model_type == "GMVAE":
            if latent_distribution == "legacy gaussian mixture":
                cfg.latent_mode = 4
        else:
            cfg.latent_mode = (
                (0 if self.analytical_kl_term else 1)
                | (2 if latent_distribution == "unit-variance gaussian"
                   else 0))
        # (h, x, z, y) keep probabilities; False / 0 / 1: none (mu:45)
        keeps = [float(k) if k else 0.0
                 for k in (dropout_keep_probabilities or ())]
        keeps = (keeps + [0.0] * 4)[:4]
        self.dropout_keep_probabilities = tuple(
            k if 0.0 < k < 1.0 else 0.0 for k in keeps)
        for i, k in enumerate(self.dropout_keep_probabilities):
            cfg.dropout_keep[i] = k
        self.uses_dropout = any(self.dropout_keep_probabilities)
        self._dropout_steps = 0
        self.config = cfg

        handle = ctypes.c_void_p()
        _lib.check(self.lib.scvae_plan_create(ctypes.byref(cfg),
                                              ctypes.byref(handle)),
                   "scvae_plan_create")
        self.handle = handle

        n = self.lib.scvae_plan_param_floats(handle)
        nm = self.lib.scvae_plan_moving_floats(handle)
        dev = self.device
        self.params = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.moving = torch.zeros(max(nm, 1), dtype=torch.float32, device=dev)
        self.adam_t = 0
        self._copy_stream = None   # (state_dict(non_blocking=True))
        self.workspace = None
        self.max_cells = 0
        self.max_samples = 0
        self._sync_cb = None

        # named views, reference creation order
        self.param_table = OrderedDict()
        name = ctypes.create_string_buffer(_lib.NAME_MAX)
        off, rows, cols = (ctypes.c_int64(), ctypes.c_int64(),
                           ctypes.c_int64())
        for i in range(self.lib.scvae_plan_param_count(handle)):
            _lib.check(self.lib.scvae_plan_param_info(
                handle, i, name, ctypes.byref(off), ctypes.byref(rows),
                ctypes.byref(cols)), "scvae_plan_param_info")
            shape = ((rows.value, cols.value) if cols.value
                     else (rows.value,))
            self.param_table[name.value.decode()] = (off.value, shape)
        self.moving_table = OrderedDict()
        size = ctypes.c_int64()
        for i in range(self.lib.scvae_plan_moving_count(handle)):
            _lib.check(self.lib.scvae_plan_moving_info(
                handle, i, name, ctypes.byref(off), ctypes.byref(size)),
                "scvae_plan_moving_info")
            self.moving_table[name.value.decode()] = (off.value,
                                                      (size.value,))
        self.initialise(seed)

    def __del__(self):
        handle = getattr(self, "handle", None)
        if handle:
            self.lib.scvae_plan_destroy(handle)
            self.handle = None

    # ---- named views ----------------------------------------------------
    @staticmethod
    def _view(flat, offset, shape):
        n = 1
        for s in shape:
            n *= s
        return flat[offset:offset + n].view(*shape)

    def parameter(self, name):
        off, shape = self.param_table[name]
        return self._view(self.params, off, shape)

    def gradient(self, name):
        off, shape = self.param_table[name]
        return self._view(self.grads, off, shape)

    def moving_statistic(self, name):
        off, shape = self.moving_table[name]
        return self._view(self.moving, off, shape)

    def named_parameters(self):
        return OrderedDict((k, self.parameter(k)) for k in self.param_table)

    def named_gradients(self):
        return OrderedDict((k, self.gradient(k)) for k in self.param_table)

    def named_moving_statistics(self):
        return OrderedDict(
            (k, self.moving_statistic(k)) for k in self.moving_table)

    def number_of_parameters(self):
        total = 0
        for _, shape in self.param_table.values():
            n = 1
            for s in shape:
                n *= s
            total += n
        return total

    # ---- initialisation / state ------------------------------------------
    def initialise(self, seed=0):
        """``tf.global_variables_initializer`` with the tf.contrib defaults:
        Glorot-uniform weights, zero biases and beta, moving mean 0 /
        variance 1, Adam slots 0."""
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        self.params.zero_()
        for name, (off, shape) in self.param_table.items():
            if name.endswith("weights"):
                limit = math.sqrt(6.0 / (shape[0] + shape[1]))
                w = (torch.rand(shape, generator=g, dtype=torch.float64)
                     * 2 - 1) * limit
                self._view(self.params, off, shape).copy_(
                    w.to(torch.float32))
        self.moving.zero_()
        for name, (off, shape) in self.moving_table.items():
            if name.endswith("moving_variance"):
                self._view(self.moving, off, shape).fill_(1.0)
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.grads.zero_()
        self.adam_t = 0
        if self.prior_probabilities_method == "custom":
            # tf.log(tf.constant(prior_probabilities)), gm:2796-2798
            self.prior_logits.copy_(torch.log(torch.tensor(
                self.prior_probabilities, dtype=torch.float64)).float())

    @property
    def prior_logits(self):
        """The K logits of p(y) (view into the parameter buffer), or None for
        the uniform prior."""
        offset = self.lib.scvae_plan_prior_offset(self.handle)
        if offset < 0:
            return None
        return self.params[offset:offset + self.n_clusters]

    def load_parameters(self, named, moving=None):
        for k, v in named.items():
            self.parameter(k).copy_(torch.as_tensor(v).to(torch.float32))
        if moving:
            for k, v in moving.items():
                self.moving_statistic(k).copy_(
                    torch.as_tensor(v).to(torch.float32))

    def state_dict(self, non_blocking=False):
        """Parameters, Adam moments, moving statistics and the step count on the
        host.  ``non_blocking``: the state is snapshotted on the device now (a
        copy in HBM on the caller's stream, microseconds) and travels to pinned
        host memory on a second stream while the caller carries on; the
        returned ``PendingState`` is the finished dictionary once its ``wait()``
        has returned (the checkpoint queue calls it, models/utilities.py)."""
        tensors = {"params": self.params, "adam_m": self.adam_m,
                   "adam_v": self.adam_v, "moving": self.moving}
        if not non_blocking or self.device.type != "cuda":
            state = {k: v.detach().cpu() for k, v in tensors.items()}
            state["adam_t"] = self.adam_t
            return state
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        side = self._copy_stream
        snapshot = {k: v.detach().clone() for k, v in tensors.items()}
        side.wait_stream(torch.cuda.current_stream(self.device))
        state = PendingState()
        with torch.cuda.stream(side):
            for k, v in snapshot.items():
                v.record_stream(side)
                state[k] = torch.empty(v.shape, dtype=v.dtype,
                                       pin_memory=True).copy_(
                                           v, non_blocking=True)
            state.ready = torch.cuda.Event()
            state.ready.record(side)
        state["adam_t"] = self.adam_t
        return state

    def load_state_dict(self, state):
        self.params.copy_(state["params"])
        self.adam_m.copy_(state["adam_m"])
        self.adam_v.copy_(state["adam_v"])
        self.moving.copy_(state["moving"])
        self.adam_t = int(state["adam_t"])

    # ---- binding -----------------------------------------------------------
    def reserve(self, max_cells, max_samples=1):
        """(Re)allocate the activation workspace for minibatches of up to
        ``max_cells`` cells and ``max_samples`` latent samples per cell."""
        max_cells, max_samples = int(max_cells), int(max_samples)
        if (self.workspace is not None and max_cells <= self.max_cells
                and max_samples <= self.max_samples):
            return
        max_cells = max(max_cells, self.max_cells)
        max_samples = max(max_samples, self.max_samples)
        nbytes = self.lib.scvae_plan_workspace_bytes(
            self.handle, max_cells, max_samples)
        if nbytes < 0:
            raise _lib.HipLibraryError("scvae_plan_workspace_bytes failed")
        self.workspace = None
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8,
                                     device=self.device)
        _lib.check(self.lib.scvae_plan_bind(
            self.handle, _ptr(self.params), _ptr(self.grads),
            _ptr(self.moving), _ptr(self.workspace), nbytes, max_cells,
            max_samples), "scvae_plan_bind")
        self.max_cells, self.max_samples = max_cells, max_samples
        self.scalars = torch.zeros(8, dtype=torch.float32, device=self.device)

    def set_fused(self, enabled):
        """Select the fused decoder-head kernel (default) or the unfused path."""
        _lib.check(self.lib.scvae_plan_set_fused(
            self.handle, 1 if enabled else 0), "scvae_plan_set_fused")

    @property
    def fused_categorised(self):
        """Whether training and evaluation steps run ``-k`` on the fused head
        kernels (k = 1, 2; bound plan)."""
        return bool(self.lib.scvae_plan_fused_categorised(self.handle))

    def set_head_arith(self, arith):
        """Arithmetic of this engine's fused head kernels: ``"bf16x9"`` (the
        exact nine-term bf16 split, default), ``"bf16x6"`` (the nine terms
        without the three smallest, <= 2^-26 of a product: fp32-class, a
        third fewer matrix instructions in the producer / consumer training
        kernel) or ``"fp32"`` (fp32 matrix cores).  A plan attribute: engines
        of one process can differ."""
        mode = {"fp32": 0, "bf16x9": 1, "bf16x6": 2, 0: 0, 1: 1, 2: 2}[arith]
        _lib.check(self.lib.scvae_plan_set_head_arith(self.handle, mode),
                   "scvae_plan_set_head_arith")

    def set_dd_atomics(self, enabled):
        """Accumulate the decoder gradient ``dd`` of a training step with
        XCD-local fp32 atomics (the default of a plan: faster where the head
        kernel has that store; the last bits of the sums differ from run to
        run) or through per-strip slabs and a fixed-order reduce
        (``False``: bit-repeatable)."""
        _lib.check(self.lib.scvae_plan_set_dd_atomics(
            self.handle, 1 if enabled else 0), "scvae_plan_set_dd_atomics")

    @property
    def dd_atomics(self):
        return bool(self.lib.scvae_plan_dd_atomics(self.handle))

    @property
    def head_arith(self):
        return ("fp32", "bf16x9", "bf16x6")[
            self.lib.scvae_plan_head_arith(self.handle)]

    def set_count_gemm(self, enabled, always=False):
        """The exact bf16-split kernels for the products with a count matrix
        (``step(..., x_counts=True)``): where they pay (default: minibatches
        from a few hundred cells upwards), ``always``, or never."""
        mode = (2 if always else 1) if enabled else 0
        _lib.check(self.lib.scvae_plan_set_count_gemm(self.handle, mode),
                   "scvae_plan_set_count_gemm")

    def set_mid_chain(self, enabled):
        """Small VAE minibatches: hidden layers + posterior heads + latent stage
        in two cooperative launches (default) or as the chain of launches."""
        _lib.check(self.lib.scvae_plan_set_mid_chain(
            self.handle, 1 if enabled else 0), "scvae_plan_set_mid_chain")

    def uses_tile_chain(self, cells, samples=1):
        """Whether a training step of ``cells`` x ``samples`` rows runs its
        hidden layers on the tile chain (one launch per layer and direction)."""
        return bool(self.lib.scvae_plan_uses_tile_chain(
            self.handle, int(cells), int(samples)))

    def set_tile_resident(self, enabled):
        """The tile chain's stages of a pass in ONE resident launch per
        direction (single process; measured slower on MI355X, hence off by
        default) or one launch per layer."""
        _lib.check(self.lib.scvae_plan_set_tile_resident(
            self.handle, 1 if enabled else 0), "scvae_plan_set_tile_resident")

    def uses_tile_resident(self, cells, samples=1):
        return bool(self.lib.scvae_plan_uses_tile_resident(
            self.handle, int(cells), int(samples)))

    def set_tile_chain(self, enabled):
        """Large VAE training minibatches: one launch per hidden layer and
        direction (default) or the chain of GEMM / batch-norm launches."""
        _lib.check(self.lib.scvae_plan_set_tile_chain(
            self.handle, 1 if enabled else 0), "scvae_plan_set_tile_chain")

    def probe_heads(self, n):
        """Arm HIP event pairs around the likelihood-head training kernel of
        the next ``n`` training steps (0: off) -- ``bench.py``'s roofline."""
        _lib.check(self.lib.scvae_plan_probe_heads(self.handle, int(n)),
                   "scvae_plan_probe_heads")
        self._probe_n = int(n)

    PROBE_STAGES = ("fetch", "count_gemm_fwd", "count_gemm_dw", "dd_reduce", "adam")

    def probe_stages(self, n):
        """Arm HIP-event pairs around the HBM-bound stages of the next ``n``
        training steps (0: off); ``probe_stages_us`` reads them back."""
        _lib.check(self.lib.scvae_plan_probe_stages(self.handle, int(n)),
                   "scvae_plan_probe_stages")
        self._stage_n = int(n)

    def probe_stages_us(self):
        """{stage: [microseconds per probed step]} (waits for the events)."""
        n = getattr(self, "_stage_n", 0)
        k = len(self.PROBE_STAGES)
        out = (ctypes.c_float * (n * k))()
        got = self.lib.scvae_plan_probe_stages_us(self.handle, out, n)
        if got < 0:
            _lib.check(got, "scvae_plan_probe_stages_us")
        return {name: [out[i * k + j] for i in range(got) if out[i * k + j] >= 0]
                for j, name in enumerate(self.PROBE_STAGES)}

    def probe_heads_ms(self):
        """Durations (ms) of the probed kernel launches recorded so far (waits
        for them on the host)."""
        n = getattr(self, "_probe_n", 0)
        out = (ctypes.c_float * max(n, 1))()
        got = self.lib.scvae_plan_probe_heads_ms(self.handle, out, n)
        if got < 0:
            _lib.check(got, "scvae_plan_probe_heads_ms")
        return [float(out[i]) for i in range(got)]

    def set_bn_one_launch(self, enabled, always=False):
        """One-launch batch norm for single-group layers: for minibatches of
        up to 1024 rows (default), whenever it applies (``always``), or never
        (the chunked statistics / finalize / apply kernels)."""
        mode = (2 if always else 1) if enabled else 0
        _lib.check(self.lib.scvae_plan_set_bn_one_launch(self.handle, mode),
                   "scvae_plan_set_bn_one_launch")

    def accepts_counts_u16(self, cells, training, n_iw=None):
        """Whether a step of ``cells`` cells may take its minibatch as uint16
        counts (half the bytes for the three kernels that stream it).
        ``n_iw``: the importance samples of the training steps in question, if
        known (1: head dropout does not stand in the way)."""
        self.reserve(int(cells), 1)
        mode = 0 if not training else (2 if n_iw == 1 else 1)
        return bool(self.lib.scvae_plan_accepts_counts_u16(
            self.handle, int(cells), mode))

    def set_sync(self, callback):
        """Install the data-parallel collective hook (see scvae_sync_fn)."""
        if callback is None:
            self._sync_cb = ctypes.cast(None, _lib.SYNC_FN)
        else:
            self._sync_cb = _lib.SYNC_FN(callback)
        _lib.check(self.lib.scvae_plan_set_sync(self.handle, self._sync_cb,
                                                None), "scvae_plan_set_sync")

    # ---- execution -----------------------------------------------------------
    def step(self, x, t, eps=None, row_const=None, training=False,
             n_iw=1, n_mc=1, warm_up_weight=1.0, deterministic_z=False,
             global_cells=None, outputs=None, scalars=None,
             decoder_extra=None, dropout_seed=None, count_sum=None,
             row_offset=0, x_counts=False, learning_rate=None,
             grad_scale=1.0, next_minibatch=None, next_noise=None,
             count_tiles=None):
        """One graph execution (no host synchronisation).  ``outputs`` maps
        optional output names of ``scvae_step_args`` to preallocated tensors.
        ``dropout_seed``: seed of this training step's dropout masks (default:
        a counter of the training steps taken).  Returns the device tensor of
        scalars.

        Work the step may carry (``scvae_side_work``; results as if issued
        right after the step, but run on the plan's second stream under the
        backward pass of the hidden layers): ``learning_rate`` -- the clip +
        Adam update of this training step (instead of ``adam_step``; single
        process only); ``next_minibatch`` -- a ``DeviceCSR.request(...)`` for
        the following step's minibatch (into buffers this step does not read);
        ``next_noise`` -- ``dict(out=, block_stride=, row_offset=, seed=,
        stream_id=)`` for its noise (``philox_normal_blocks`` arguments).
        ``count_tiles``: this step's uint16 minibatch also as a
        ``minibatch.CountTiles`` (same rows)."""
        cells = x.shape[0]
        samples = 1 if deterministic_z else n_iw * n_mc
        self.reserve(cells, samples)
        a = _lib.StepArgs()
        if x.dtype == torch.uint16:
            # the minibatch as uint16 counts (DeviceCSR.gather_counts_u16): it
            # is x and t of the step; see accepts_counts_u16()
            if t is not x or x.dim() != 2 or x.stride(1) != 1:
                raise ValueError("a uint16 minibatch is both x and t")
            a.counts_u16 = x.data_ptr()
            a.counts_ld = x.stride(0)
            # the same rows as tile-indexed non-zeros (minibatch.CountTiles):
            # the input layer's two products read them instead of the batch
            if count_tiles is not None:
                a.count_tiles = count_tiles.address
        else:
            a.x = x.data_ptr()
            a.t = t.data_ptr()
        a.row_const = row_const.data_ptr() if row_const is not None else None
        if self.decoder_extra:
            if (decoder_extra is None or tuple(decoder_extra.shape)
                    != (cells, self.decoder_extra)):
                raise ValueError(
                    "decoder_extra must be [cells, {}]".format(
                        self.decoder_extra))
            a.decoder_extra = decoder_extra.data_ptr()
        a.eps = eps.data_ptr() if eps is not None else None
        if self.likelihood == "constrained poisson":
            # N of every cell: the total of the constrained rates (va:1017-1019)
            if count_sum is None or count_sum.numel() != cells:
                raise ValueError("count_sum must hold one value per cell")
            count_sum = count_sum.reshape(-1).contiguous()
            a.count_sum = count_sum.data_ptr()
        if training and self.uses_dropout:
            if dropout_seed is None:
                self._dropout_steps += 1
                dropout_seed = (0x5C7AE << 40) + self._dropout_steps
            a.dropout_seed = int(dropout_seed) & 0xFFFFFFFFFFFFFFFF
        a.cells = cells
        a.global_cells = global_cells if global_cells else cells
        # data parallel: this rank's first cell within the global minibatch
        a.row_offset = int(row_offset)
        # the caller vouches that x holds integer counts below 65 536
        # (DeviceCSR.integer_counts): the input layer's two large products take
        # the exact bf16-split kernels instead of the fp32 MFMA kernels
        a.x_counts = 1 if x_counts else 0
        a.n_iw, a.n_mc = n_iw, n_mc
        a.training = 1 if training else 0
        a.deterministic_z = 1 if deterministic_z else 0
        a.warm_up_weight = float(warm_up_weight)
        out_scalars = scalars if scalars is not None else self.scalars
        a.scalars = out_scalars.data_ptr()
        if outputs:
            for key, tensor in outputs.items():
                setattr(a, key, tensor.data_ptr())
        side = None
        if (learning_rate is not None or next_minibatch is not None
                or next_noise is not None):
            side = _lib.SideWork()
            if learning_rate is not None:
                if not training:
                    raise ValueError("learning_rate: training steps only")
                side.adam_m = self.adam_m.data_ptr()
                side.adam_v = self.adam_v.data_ptr()
                side.adam_grad_scale = float(grad_scale)
                # (adam_t advances only once the step has been accepted, below)
                side.adam_lr_t = self._adam_lr_t(learning_rate, self.adam_t + 1)
                side.adam_beta1 = ADAM_BETA1
                side.adam_beta2 = ADAM_BETA2
                side.adam_epsilon = ADAM_EPSILON
            if next_minibatch is not None:
                next_minibatch.fill(side)
            if next_noise is not None:
                out = next_noise["out"]
                if out.dim() != 3 or not out.is_contiguous():
                    raise ValueError("contiguous [blocks, rows, cols] expected")
                side.noise_out = out.data_ptr()
                (side.noise_blocks, side.noise_block_rows,
                 side.noise_cols) = out.shape
                side.noise_block_stride = int(next_noise["block_stride"])
                side.noise_row_offset = int(next_noise.get("row_offset", 0))
                side.noise_seed = int(next_noise["seed"])
                side.noise_stream_id = int(next_noise["stream_id"])
            a.side = ctypes.addressof(side)
        _lib.check(self.lib.scvae_plan_step(
            self.handle, ctypes.byref(a), current_stream_handle(self.device)),
            "scvae_plan_step")
        if side is not None and learning_rate is not None:
            self.adam_t += 1      # a refused step raised above: no update, no count
        return out_scalars

    def dropout_mask(self, site, rows, cols, keep, dropout_seed):
        """The [rows, cols] tensor ``mask / keep`` that a training step with
        ``dropout_seed`` multiplies into the input of layer ``site`` (site
        numbers: include/scvae_hip.h, scvae_dropout_apply)."""
        ones = torch.ones(rows, cols, device=self.device)
        out = torch.empty_like(ones)
        _lib.check(self.lib.scvae_dropout_apply(
            _ptr(ones), _ptr(out), rows, cols, float(keep),
            int(dropout_seed) & 0xFFFFFFFFFFFFFFFF, int(site), 0,
            current_stream_handle(self.device)), "scvae_dropout_apply")
        return out

    def decode(self, z, out=None):
        """Mean of p(x|z) for latent values ``z`` [rows, L] with the moving
        batch-norm statistics (the decoder half of the graph; ``model.sample``).
        """
        rows = z.shape[0]
        self.reserve(rows, 1)
        if out is None:
            out = torch.empty(rows, self.feature_size, device=self.device)
        z = z.contiguous()
        _lib.check(self.lib.scvae_plan_decode(
            self.handle, _ptr(z), rows, _ptr(out),
            current_stream_handle(self.device)), "scvae_plan_decode")
        return out

    @staticmethod
    def _adam_lr_t(learning_rate, t):
        """``lr_t`` of optimiser step ``t`` (tf.train.AdamOptimizer)."""
        return (learning_rate * math.sqrt(1.0 - ADAM_BETA2 ** t)
                / (1.0 - ADAM_BETA1 ** t))

    def adam_step(self, learning_rate, grad_scale=1.0):
        """clip-by-value(+-1) + ``tf.train.AdamOptimizer`` update (va:2742-2759)."""
        lr_t = self._adam_lr_t(learning_rate, self.adam_t + 1)
        _lib.check(self.lib.scvae_adam_clip_step(
            _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m),
            _ptr(self.adam_v), self.params.numel(), float(grad_scale),
            float(lr_t), ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON,
            current_stream_handle(self.device)), "scvae_adam_clip_step")
        self.adam_t += 1
