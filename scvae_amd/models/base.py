"""Shared machinery of the two model classes: log directories, the training,
evaluation loops and their bookkeeping.

The loops restate ``VariationalAutoencoder.train`` / ``.evaluate``
(``scvae/models/variational_autoencoder.py:640-1599, 1781-2217``) and their
GMVAE twins (``gaussian_mixture_variational_autoencoder.py:684-1947,
2162-2786``) with the TensorFlow session replaced by an ``Engine`` whose steps
are enqueued on the GPU without host synchronisation:

* the count matrix lives in HBM as CSR and minibatches are gathered on device;
* per-batch evaluation scalars are written to rows of a device tensor and read
  back once per pass;
* printed lines, scalar tags, checkpoint cadence, early stopping and the
  ``sum(batch means) / (N / B)`` epoch averages are
  those of the reference.

When ``torch.distributed`` is initialised with more than one rank the same
loops run data-parallel (``scvae_amd/dataparallel.py``); rank 0 does the I/O.
"""

import os
import shutil
from time import time

import numpy
import torch

from scvae_amd.defaults import defaults
from scvae_amd.models import utilities as mu
from scvae_amd.utilities import (
    capitalise_string, format_duration, format_time)


def _rank_zero_value(value, device):
    """Rank 0's ``value`` (a number) on every rank."""
    holder = torch.tensor([float(value)], dtype=torch.float64, device=device)
    torch.distributed.broadcast(holder, src=0)
    return float(holder.item())


def _distributed():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


class ModelBase:
    """Common state and loops of ``VariationalAutoencoder`` and
    ``GaussianMixtureVariationalAutoencoder``."""

    type = None
    early_stopping_rounds = 10

    # set by subclasses' constructors ---------------------------------------
    feature_size = None
    latent_size = None
    hidden_sizes = None
    reconstruction_distribution_name = None
    minibatch_normalisation = None
    kl_weight_value = 1
    number_of_warm_up_epochs = 0
    number_of_monte_carlo_samples = None
    number_of_importance_samples = None
    base_log_directory = None
    stopped_early = None
    _engine = None
    _device = None
    noise_seed = 1
    initial_seed = 0

    # -- engine --------------------------------------------------------------
    def _engine_arguments(self):
        raise NotImplementedError

    @property
    def engine(self):
        """The GPU engine (created on first use: building the object itself
        needs no GPU, exactly like building the TF graph needs no session)."""
        if self._engine is None:
            from scvae_amd.engine import Engine
            device = self._device
            if device is None:
                # one process per GPU; more ranks than GPUs (tests) share them
                import torch
                local_rank = int(os.environ.get("LOCAL_RANK", "0"))
                device = "cuda:{}".format(
                    local_rank % max(torch.cuda.device_count(), 1))
            self._engine = Engine(device=device, seed=self.initial_seed,
                                  **self._engine_arguments())
        return self._engine

    @property
    def parameters(self):
        """Trainable parameters in the model (names and shapes as the
        reference prints them)."""
        from scvae_amd import _lib
        table = self._parameter_table()
        parts = ["Trainable parameters"]
        width = max(len(name) for name, _ in table)
        for name, shape in table:
            parts.append("{:{}}  ({})".format(
                name + ":0", width + 2, ", ".join(map(str, shape))
                + ("," if len(shape) == 1 else "")))
        del _lib
        return "\n    ".join(parts)

    def _parameter_table(self):
        if self._engine is not None:
            return [(k, shape)
                    for k, (_, shape) in self._engine.param_table.items()]
        return self._parameter_shapes()

    # -- directories -----------------------------------------------------------
    def log_directory(self, base=None, run_id=None, early_stopping=False,
                      best_model=False):
        if not base:
            base = self.base_log_directory
        log_directory = os.path.join(base, self.name)
        if run_id is None:
            run_id = defaults["models"]["run_id"]
        if run_id:
            run_id = mu.check_run_id(run_id)
            log_directory = os.path.join(
                log_directory, "run_{}".format(run_id))
        if early_stopping and best_model:
            raise ValueError(
                "Early-stopping model and best model are mutually exclusive.")
        elif early_stopping:
            log_directory = os.path.join(log_directory, "early_stopping")
        elif best_model:
            log_directory = os.path.join(log_directory, "best")
        return log_directory

    def has_been_trained(self, run_id=None):
        return bool(mu.get_checkpoint_state(self.log_directory(run_id=run_id)))

    def early_stopping_status(self, run_id=None):
        stopped_early = False
        epochs_with_no_improvement = 0
        early_stopping_log_directory = self.log_directory(
            run_id=run_id, early_stopping=True)
        log_directory = os.path.dirname(early_stopping_log_directory)
        if (os.path.exists(log_directory)
                and os.path.exists(early_stopping_log_directory)):
            validation_losses = mu.load_learning_curves(
                model=self, data_set_kinds="validation", run_id=run_id,
                log_directory=log_directory)["lower_bound"]
            stopped_early, epochs_with_no_improvement = (
                mu.early_stopping_status(
                    validation_losses, self.early_stopping_rounds))
        return stopped_early, epochs_with_no_improvement

    # -- hooks for the subclasses ----------------------------------------------
    def _training_minibatch_size(self, minibatch_size, scenario):
        return minibatch_size

    def _eps_shape(self, samples, cells):
        return (samples, cells, self.latent_size)

    def _loss_tags(self):
        """(scalar index, tag, printed label) of the losses."""
        return [(0, "lower_bound", "ELBO"), (2, "reconstruction_error", "ENRE"),
                (3, "kl_divergence", "KL")]

    def _prior_summary(self):
        """(probabilities, means, variances) of the latent prior, as logged
        under ``prior/cluster_<k>/...``."""
        raise NotImplementedError

    # -- data ------------------------------------------------------------------
    def _device_matrices(self, data_set, noisy=False):
        """(input, target) of a data set on the device.  ``noisy``: the data
        set's noisy preprocessor applied to its values -- a new draw at every
        call -- as both input and target (va:960-976)."""
        from scvae_amd.minibatch import DeviceCSR
        import scipy.sparse

        def upload(values):
            if not scipy.sparse.issparse(values):
                values = scipy.sparse.csr_matrix(
                    numpy.asarray(values, dtype=numpy.float32))
            return DeviceCSR.from_scipy(values, self.engine.device)
        transformed = None     # the host target where it is not data_set.values
        if noisy:
            transformed = data_set.noisy_preprocess(data_set.values)
            x = t = upload(transformed)
        elif self.reconstruction_distribution_name == "bernoulli":
            # the Bernoulli likelihood models the binarised values
            # (va:854-857; "binarise" = values > 0.5, data/processing.py:511-513)
            if data_set.has_binarised_values:
                transformed = data_set.binarised_values
                t = upload(transformed)
            else:
                values = data_set.values
                if scipy.sparse.issparse(values):
                    values = scipy.sparse.csr_matrix(values)
                    binarised = values.copy()
                    binarised.data = (binarised.data > 0.5).astype(
                        numpy.float32)
                    binarised.eliminate_zeros()
                else:
                    binarised = (numpy.asarray(values) > 0.5).astype(
                        numpy.float32)
                transformed = binarised
                t = upload(binarised)
        else:
            t = upload(data_set.values)
        if noisy:
            pass
        elif data_set.has_preprocessed_values:
            x = upload(data_set.preprocessed_values)
        elif self.reconstruction_distribution_name == "bernoulli":
            x = upload(data_set.values)
        else:
            x = t
        t.transformed_values = transformed
        t.decoder_extra = self._decoder_extra_inputs(data_set)
        # N of the constrained Poisson: the count sums of the data set
        # (count_sum_parameter, va:823-826, 1017-1019)
        t.count_sum = None
        if self.use_count_sum_as_parameter:
            t.count_sum = torch.as_tensor(
                numpy.asarray(data_set.count_sum, dtype=numpy.float32)
                .reshape(-1), device=self.engine.device)
        return x, t

    @property
    def decoder_extra_size(self):
        """Columns appended to z at the decoder input: one-hot batch indices
        (batch correction) and the normalised count sum (va:2407-2441)."""
        size = 0
        if self.batch_correction:
            size += int(self.number_of_batches)
        if self.use_count_sum_as_feature:
            size += 1
        return size

    def _decoder_extra_inputs(self, data_set):
        """[N, E] device tensor of the extra decoder inputs of a data set (the
        ``batch_indices`` / ``count_sum_feature`` feeds, va:1012-1023)."""
        if not self.decoder_extra_size:
            return None
        device = self.engine.device
        parts = []
        if self.batch_correction:
            batch_indices = mu.batch_indices_for_subset(data_set)
            indices = torch.as_tensor(
                numpy.asarray(batch_indices).reshape(-1), dtype=torch.int64,
                device=device)
            parts.append(torch.nn.functional.one_hot(
                indices, int(self.number_of_batches)).to(torch.float32))
        if self.use_count_sum_as_feature:
            parts.append(torch.as_tensor(
                numpy.asarray(data_set.normalised_count_sum,
                              dtype=numpy.float32).reshape(-1, 1),
                device=device))
        return torch.cat(parts, dim=1).contiguous()

    # ======================================================================
    # training
    # ======================================================================
    def train(self, training_set, validation_set=None, number_of_epochs=None,
              minibatch_size=None, learning_rate=None, run_id=None,
              new_run=None, reset_training=None, **kwargs):
        """Train model (signature and behaviour of va:640-1599)."""
        dm = defaults["models"]
        if number_of_epochs is None:
            number_of_epochs = dm["number_of_epochs"]
        if minibatch_size is None:
            minibatch_size = dm["minibatch_size"]
        if learning_rate is None:
            learning_rate = dm["learning_rate"]
        if run_id is None:
            run_id = dm["run_id"]
        if new_run is None:
            new_run = dm["new_run"]
        if reset_training is None:
            reset_training = dm["reset_training"]
        analyses_directory = kwargs.get("analyses_directory")
        if analyses_directory is None:
            analyses_directory = defaults["analyses"]["directory"]

        world, rank = _distributed()
        master = rank == 0

        def say(*args, **kw):
            if master:
                print(*args, **kw)

        start_time = time()
        if run_id:
            run_id = mu.check_run_id(run_id)
            new_run = True
        elif new_run:
            run_id = mu.generate_unique_run_id_for_model(
                model=self, timestamp=start_time)
        model_string = ("model for run {}".format(run_id) if run_id
                        else "model")

        permanent_log_directory = self.log_directory(run_id=run_id)
        if (master and reset_training
                and os.path.exists(permanent_log_directory)):
            mu.clear_log_directory(permanent_log_directory)
        if world > 1:   # nobody reads the directories before rank 0 is done with them
            torch.distributed.barrier()

        metadata_log = {
            "epochs trained": None,
            "start time": format_time(start_time),
            "training duration": None,
            "last epoch duration": None,
            "learning rate": learning_rate,
            "minibatch size": minibatch_size
        }

        old_checkpoint = mu.get_checkpoint_state(permanent_log_directory)
        epoch_start = (mu.checkpoint_epoch(old_checkpoint)
                       if old_checkpoint else 0)

        temporary_log_directory = kwargs.get("temporary_log_directory")
        replace_temporary_directory = False
        if temporary_log_directory:
            log_directory = self.log_directory(
                base=temporary_log_directory, run_id=run_id)
            early_stopping_log_directory = self.log_directory(
                base=temporary_log_directory, run_id=run_id,
                early_stopping=True)
            best_model_log_directory = self.log_directory(
                base=temporary_log_directory, run_id=run_id, best_model=True)
            temporary_checkpoint = mu.get_checkpoint_state(log_directory)
            temporary_epoch_start = (
                mu.checkpoint_epoch(temporary_checkpoint)
                if temporary_checkpoint else 0)
            if temporary_epoch_start > epoch_start:
                epoch_start = temporary_epoch_start
            else:
                replace_temporary_directory = True
        else:
            log_directory = self.log_directory(run_id=run_id)
            early_stopping_log_directory = self.log_directory(
                run_id=run_id, early_stopping=True)
            best_model_log_directory = self.log_directory(
                run_id=run_id, best_model=True)

        data_string = mu.build_data_string(
            data_set=training_set,
            reconstruction_distribution_name=(
                self.reconstruction_distribution_name))
        training_string = mu.build_training_string(
            model_string=model_string, epoch_start=epoch_start,
            number_of_epochs=number_of_epochs, data_string=data_string)

        if world > 1:
            # one decision for all ranks (a rank that disagreed would leave the
            # collectives of the others hanging)
            epoch_start = int(_rank_zero_value(epoch_start,
                                               self.engine.device))

        if epoch_start >= number_of_epochs:
            say(training_string)
            return 0

        if (master and temporary_log_directory
                and os.path.exists(permanent_log_directory)
                and replace_temporary_directory):
            say("Copying log directory to temporary directory.")
            copying_time_start = time()
            if os.path.exists(log_directory):
                shutil.rmtree(log_directory)
            shutil.copytree(permanent_log_directory, log_directory)
            say("Log directory copied ({}).".format(
                format_duration(time() - copying_time_start)))
            say()
        if world > 1:
            torch.distributed.barrier()

        minibatch_size = self._training_minibatch_size(
            minibatch_size, "training")
        n_iw = self.number_of_importance_samples["training"]
        n_mc = self.number_of_monte_carlo_samples["training"]

        say("Preparing data.")
        preparing_data_time_start = time()
        engine = self.engine
        if kwargs.get("deterministic"):
            # (not in the reference) the decoder gradient is summed over the
            # gene strips in a fixed order (bit-repeatable steps) instead of
            # with fp32 atomics.  Absent / False leaves the plan's default --
            # scvae_default_dd_atomics, SCVAE_DD_ACCUMULATION -- alone.
            engine.set_dd_atomics(False)
        # (va:840-860: with a noisy preprocessor the matrices are drawn anew at
        #  the head of every epoch; the draw here sizes the buffers)
        noisy_preprocess = training_set.noisy_preprocess is not None
        x_train, t_train = self._device_matrices(
            training_set, noisy=noisy_preprocess)
        n_examples_train = training_set.number_of_examples
        if validation_set:
            x_valid, t_valid = self._device_matrices(
                validation_set,
                noisy=noisy_preprocess
                and validation_set.noisy_preprocess is not None)
            n_examples_valid = validation_set.number_of_examples
        sync = None
        if world > 1:
            from scvae_amd.dataparallel import GradientSynchroniser
            # (bound once, for the largest step of EITHER epoch-end pass: a
            #  later re-bind would replace the buffers the synchroniser holds)
            engine.reserve(max(
                minibatch_size,
                self._evaluation_largest_step(
                    n_examples_train, minibatch_size, n_iw * n_mc),
                self._evaluation_largest_step(
                    n_examples_valid, minibatch_size, n_iw * n_mc)
                if validation_set else 0), n_iw * n_mc)
            sync = GradientSynchroniser(engine)
        say("Data prepared ({}).".format(
            format_duration(time() - preparing_data_time_start)))
        say()

        steps_per_epoch = numpy.ceil(n_examples_train / minibatch_size)
        output_at_step = numpy.round(numpy.linspace(0, steps_per_epoch, 11))

        learning_curves = {"training": {
            tag: [] for _, tag, _ in self._loss_tags()}}
        if validation_set:
            learning_curves["validation"] = {
                tag: [] for _, tag, _ in self._loss_tags()}

        training_writer = validation_writer = None
        if master:
            os.makedirs(log_directory, exist_ok=True)
            training_writer = mu.ScalarWriter(
                os.path.join(log_directory, "training"))
            if validation_set:
                validation_writer = mu.ScalarWriter(
                    os.path.join(log_directory, "validation"))

        # Initialisation
        checkpoint = mu.get_checkpoint_state(log_directory)
        epochs_with_no_improvement = 0
        lower_bound_valid_maximum = -numpy.inf
        lower_bound_valid_early_stopping = -numpy.inf
        if checkpoint:
            say("Restoring earlier model parameters.")
            restoring_time_start = time()
            engine.load_state_dict(mu.load_checkpoint(checkpoint))
            epoch_start = mu.checkpoint_epoch(checkpoint)
            if validation_set:
                curve = mu.load_learning_curves(
                    model=self, data_set_kinds="validation", run_id=run_id,
                    log_directory=log_directory)["lower_bound"]
                if curve is not None and len(curve):
                    lower_bound_valid_maximum = curve.max()
                    self.stopped_early, epochs_with_no_improvement = (
                        self.early_stopping_status(run_id=run_id))
                    back = (0 if numpy.isnan(epochs_with_no_improvement)
                            else int(epochs_with_no_improvement))
                    lower_bound_valid_early_stopping = curve[-1 - back]
                else:
                    self.stopped_early = False
            say("Earlier model parameters restored ({}).".format(
                format_duration(time() - restoring_time_start)))
            say()
        else:
            say("Initialising model parameters.")
            initialising_time_start = time()
            engine.initialise(self.initial_seed)
            epoch_start = 0
            if validation_set:
                self.stopped_early = False
            say("Model parameters initialised ({}).".format(
                format_duration(time() - initialising_time_start)))
            say()
        if sync is not None:
            # rank 0's weights, Adam slots *and* step count / epoch (a rank that
            # saw a stale checkpoint must not keep its own)
            sync.broadcast_state(0)
            epoch_start = int(_rank_zero_value(epoch_start, engine.device))

        metadata_log["epochs trained"] = (epoch_start, number_of_epochs)
        say(training_string)
        say()
        training_time_start = time()
        epoch_duration = 0.0
        device = engine.device
        F = self.feature_size
        samples = n_iw * n_mc

        local_batch = minibatch_size // world if world > 1 else minibatch_size
        # two sets of minibatch buffers: a step carries the fetch and the noise
        # of the NEXT minibatch and -- single process -- its own optimiser update
        # (``Engine.step(learning_rate=, next_minibatch=, next_noise=)``: one
        # call per step), so set i is read while set i ^ 1 fills
        same_matrix = x_train is t_train
        x_buffers = [torch.empty(max(local_batch, 1), F, device=device)
                     for _ in range(2)]
        t_buffers = (x_buffers if same_matrix
                     else [torch.empty_like(b) for b in x_buffers])
        row_consts = [torch.empty(max(local_batch, 1), device=device)
                      for _ in range(2)]
        eps_buffers = [torch.empty(
            int(numpy.prod(self._eps_shape(samples, max(local_batch, 1)))),
            device=device) for _ in range(2)]
        step = int(epoch_start * steps_per_epoch)
        # (the epoch-end passes run steps of several minibatches: bound once,
        #  ahead of the first training step)
        engine.reserve(max(
            local_batch, 1,
            self._evaluation_largest_step(
                n_examples_train, minibatch_size, samples),
            self._evaluation_largest_step(
                n_examples_valid, minibatch_size, samples)
            if validation_set else 0), samples)
        # an integer count matrix that is both input and target: the minibatch
        # is densified as uint16 where the plan takes it (half the bytes for the
        # three kernels that stream it; the step is bit-identical)
        u16_buffers = None
        if (same_matrix and getattr(x_train, "integer_counts", False)
                and hasattr(x_train, "gather_counts_u16")
                and engine.accepts_counts_u16(max(local_batch, 1), True,
                                              n_iw=n_iw)):
            u16_buffers = [torch.empty(
                max(local_batch, 1), x_train.u16_pitch, dtype=torch.uint16,
                device=device) for _ in range(2)]

        def minibatch_buffers(slot, cells):
            """(x, t, row constant, fetch request or None) of a minibatch of
            ``cells`` cells in buffer set ``slot``."""
            rc = row_consts[slot][:cells]
            if (u16_buffers is not None
                    and engine.accepts_counts_u16(cells, True, n_iw=n_iw)):
                xb = tb = u16_buffers[slot][:cells]
            else:
                xb, tb = x_buffers[slot][:cells], t_buffers[slot][:cells]
            return xb, tb, rc

        def noise_buffer(slot, cells):
            return eps_buffers[slot][:int(numpy.prod(
                self._eps_shape(samples, cells)))]
        engine.scalars.zero_()   # incl. the sticky non-finite-step counter [7]

        from scvae_amd.minibatch import philox_normal

        checkpoint_writer = mu.CheckpointWriter()
        first_epoch = True
        for epoch in range(epoch_start, number_of_epochs):
            if noisy_preprocess:
                # (va:960-976; the first epoch uses the draw made above)
                say("Noisily preprocess values.")
                noisy_time_start = time()
                if not first_epoch:
                    x_train, t_train = self._device_matrices(
                        training_set, noisy=True)
                    if (validation_set
                            and validation_set.noisy_preprocess is not None):
                        x_valid, t_valid = self._device_matrices(
                            validation_set, noisy=True)
                say("Values noisily preprocessed ({}).".format(
                    format_duration(time() - noisy_time_start)))
                say()
            first_epoch = False
            epoch_time_start = time()
            if self.number_of_warm_up_epochs:
                warm_up_weight = float(
                    min(epoch / (self.number_of_warm_up_epochs), 1.0))
            else:
                warm_up_weight = 1.0

            shuffled = numpy.random.permutation(n_examples_train)
            if world > 1:   # every rank must see rank 0's permutation
                holder = torch.from_numpy(shuffled).to(device)
                torch.distributed.broadcast(holder, src=0)
                shuffled_indices = holder
            else:
                shuffled_indices = torch.from_numpy(shuffled).to(device)

            # this epoch's minibatches: (rows of this rank, global cells, first
            # global row of this rank)
            batches = []
            for i in range(0, n_examples_train, minibatch_size):
                rows = shuffled_indices[i:(i + minibatch_size)]
                global_cells = int(rows.numel())
                lo = 0
                if world > 1:
                    global_cells -= global_cells % world
                    if global_cells == 0:
                        continue
                    per_rank = global_cells // world
                    lo = rank * per_rank
                    rows = rows[lo:lo + per_rank]
                batches.append((rows, global_cells, lo))

            slot, carried = 0, False
            for index, (rows, global_cells, lo) in enumerate(batches):
                step_time_start = time()
                cells = int(rows.numel())
                xb, tb, rc = minibatch_buffers(slot, cells)
                eps = noise_buffer(slot, cells)
                if not carried:   # (the previous step brought this minibatch)
                    if xb is tb:
                        t_train.request(rows, tb, rc).issue()
                    else:
                        t_train.gather_dense(rows, out=tb, row_const_out=rc)
                        x_train.gather_dense(rows, out=xb)
                    self._draw_noise(eps, samples, cells, global_cells, lo, step)
                # what this step carries for the next one
                next_minibatch = next_noise = None
                carried = False
                if same_matrix and index + 1 < len(batches):
                    n_rows, n_global, n_lo = batches[index + 1]
                    n_cells = int(n_rows.numel())
                    n_xb, _, n_rc = minibatch_buffers(slot ^ 1, n_cells)
                    next_minibatch = t_train.request(n_rows, n_xb, n_rc)
                    next_noise = self._noise_request(
                        noise_buffer(slot ^ 1, n_cells), samples, n_cells,
                        n_global, n_lo, step + 1)
                    carried = True
                de = (t_train.decoder_extra.index_select(0, rows)
                      if t_train.decoder_extra is not None else None)
                cs = (t_train.count_sum.index_select(0, rows)
                      if t_train.count_sum is not None else None)
                scalars = engine.step(
                    xb, tb, eps=eps, row_const=rc, training=True, n_iw=n_iw,
                    n_mc=n_mc, warm_up_weight=warm_up_weight,
                    global_cells=global_cells, decoder_extra=de,
                    count_sum=cs, x_counts=x_train.integer_counts,
                    # (the masks are keyed by the global row: the same for any
                    # sharding, like the noise)
                    dropout_seed=((self.noise_seed * 1000003) << 40)
                    + step + 1, row_offset=lo,
                    # single process: clip + Adam ride with the step; data
                    # parallel: the gradient all-reduce comes first
                    learning_rate=learning_rate if sync is None else None,
                    next_minibatch=next_minibatch, next_noise=next_noise)
                if sync is not None:
                    sync.all_reduce_gradients()
                    engine.adam_step(learning_rate)
                slot ^= 1

                if (step + 1 - steps_per_epoch * epoch) in output_at_step:
                    local = scalars.clone()
                    if sync is not None:
                        sync.all_reduce_scalars(local)
                    local = local.cpu()
                    minibatch_loss = float(local[0])
                    # [7]: steps since the start of train() whose ELBO was not
                    # finite (sticky device-side counter, summed over the ranks)
                    non_finite_steps = float(local[7])
                    step_duration = time() - step_time_start
                    say("Step {:d} ({}): {:.5g}.".format(
                        int(step + 1), format_duration(step_duration),
                        minibatch_loss))
                    if numpy.isnan(minibatch_loss) or non_finite_steps > 0:
                        raise ArithmeticError(
                            "Aborting. The ELBO for the last batch became "
                            "indefinite.")
                step += 1

            say()
            torch.cuda.synchronize(device)
            non_finite = engine.scalars[7:8].clone()
            if sync is not None:
                sync.all_reduce_scalars(non_finite)
            if float(non_finite.item()) > 0:   # any step of the epoch, any rank
                raise ArithmeticError(
                    "Aborting. The ELBO for a batch became indefinite.")
            epoch_duration = time() - epoch_time_start
            say("Epoch {} ({}):".format(
                epoch + 1, format_duration(epoch_duration)))
            if warm_up_weight < 1:
                say("    Warm-up weight: {:.2g}".format(warm_up_weight))
            say("    Evaluating model.")

            prior = self._prior_summary()

            # Training evaluation
            evaluating_time_start = time()
            train_eval = self._evaluation_pass(
                x_train, t_train, training_set, minibatch_size, n_iw, n_mc,
                sync=sync)
            if numpy.isnan(train_eval["lower_bound"]):
                raise ArithmeticError(
                    "Aborting. The ELBO for the training set became "
                    "indefinite.")
            for _, tag, _ in self._loss_tags():
                learning_curves["training"][tag].append(train_eval[tag])
            evaluating_duration = time() - evaluating_time_start
            if master:
                self._write_epoch_summary(
                    training_writer, train_eval, epoch + 1,
                    prior=None if validation_set else prior,
                    kl_neurons=True)
            say("    {} set ({}):".format(
                training_set.kind.capitalize(),
                format_duration(evaluating_duration)),
                self._format_losses(train_eval))
            self._print_extra(say, train_eval, training_set)

            lower_bound_valid = None
            valid_eval = None
            if validation_set:
                evaluating_time_start = time()
                valid_eval = self._evaluation_pass(
                    x_valid, t_valid, validation_set, minibatch_size, n_iw,
                    n_mc, sync=sync)
                lower_bound_valid = valid_eval["lower_bound"]
                if numpy.isnan(lower_bound_valid):
                    raise ArithmeticError(
                        "Aborting. The ELBO for the validation set became "
                        "indefinite.")
                for _, tag, _ in self._loss_tags():
                    learning_curves["validation"][tag].append(valid_eval[tag])
                evaluating_duration = time() - evaluating_time_start
                if master:
                    self._write_epoch_summary(
                        validation_writer, valid_eval, epoch + 1, prior=prior,
                        kl_neurons=False)
                say("    {} set ({}):".format(
                    validation_set.kind.capitalize(),
                    format_duration(evaluating_duration)),
                    self._format_losses(valid_eval))
                self._print_extra(say, valid_eval, validation_set)

            # Early stopping (va:1385-1441)
            if validation_set and not self.stopped_early:
                if lower_bound_valid < lower_bound_valid_early_stopping:
                    if epochs_with_no_improvement == 0:
                        say("    Early stopping:",
                            "Validation loss did not improve",
                            "for this epoch.")
                        say("        "
                            "Saving model parameters for previous epoch.")
                        saving_time_start = time()
                        lower_bound_valid_early_stopping = lower_bound_valid
                        # (queued behind the previous epoch's save and ahead of
                        #  this epoch's: the latest checkpoint then IS the
                        #  previous epoch's)
                        if master:
                            checkpoint_writer.copy_latest(
                                log_directory, early_stopping_log_directory)
                        say("        "
                            "Previous model parameters saved ({})."
                            .format(format_duration(
                                time() - saving_time_start)))
                    else:
                        say("    Early stopping:",
                            "Validation loss has not improved",
                            "for {} epochs.".format(
                                epochs_with_no_improvement + 1))
                    epochs_with_no_improvement += 1
                else:
                    if epochs_with_no_improvement > 0:
                        say("    Early stopping cancelled:",
                            "Validation loss improved.")
                    epochs_with_no_improvement = 0
                    lower_bound_valid_early_stopping = lower_bound_valid
                    if master:
                        checkpoint_writer.remove_tree(
                            early_stopping_log_directory)
                if epochs_with_no_improvement >= self.early_stopping_rounds:
                    say("    Early stopping in effect:",
                        "Previously saved model parameters is available.")
                    self.stopped_early = True
                    epochs_with_no_improvement = numpy.nan

            # Saving model parameters (update checkpoint)
            say("    Saving model parameters.")
            saving_time_start = time()
            if master:   # (written in the background, see CheckpointWriter)
                checkpoint_writer.save(
                    engine.state_dict(
                        non_blocking=self.checkpoint_non_blocking),
                    log_directory,
                    epoch + 1)
            say("    Model parameters saved ({}).".format(
                format_duration(time() - saving_time_start)))

            if (validation_set
                    and lower_bound_valid > lower_bound_valid_maximum):
                say("    Best validation lower_bound yet.",
                    "Saving model parameters as best model parameters.")
                saving_time_start = time()
                lower_bound_valid_maximum = lower_bound_valid
                if master:   # (behind this epoch's save, see CheckpointWriter)
                    checkpoint_writer.copy_latest(
                        log_directory, best_model_log_directory, prune=True)
                say("    Best model parameters saved ({}).".format(
                    format_duration(time() - saving_time_start)))
            say()

            intermediate_analyser = kwargs.get("intermediate_analyser")
            if master and intermediate_analyser:
                source = valid_eval if validation_set else train_eval
                intermediate_analyser(
                    epoch=epoch, learning_curves=learning_curves,
                    epoch_start=epoch_start, model_type=self.type,
                    latent_values=source["latent_values"],
                    data_set=(validation_set if validation_set
                              else training_set),
                    centroids=self._centroids(prior), model_name=self.name,
                    run_id=run_id, analyses_directory=analyses_directory)
                say()

        checkpoint_writer.close()
        training_duration = time() - training_time_start
        say("{} trained for {} epochs ({}).".format(
            capitalise_string(model_string), number_of_epochs,
            format_duration(training_duration)))
        say()

        if master:
            mu.remove_old_checkpoints(log_directory)
            if temporary_log_directory:
                print("Moving log directory to permanent directory.")
                copying_time_start = time()
                if os.path.exists(permanent_log_directory):
                    shutil.rmtree(permanent_log_directory)
                shutil.move(log_directory, permanent_log_directory)
                print("Log directory moved ({}).".format(
                    format_duration(time() - copying_time_start)))
                print()
            metadata_log["training duration"] = format_duration(
                training_duration)
            metadata_log["last epoch duration"] = format_duration(
                epoch_duration)
            metadata_log_filename = "metadata_log"
            epochs_trained = metadata_log.get("epochs trained")
            if epochs_trained:
                metadata_log_filename += "-" + "-".join(
                    map(str, epochs_trained))
            metadata_log_path = os.path.join(
                self.log_directory(run_id=run_id),
                metadata_log_filename + ".log")
            with open(metadata_log_path, "w") as metadata_log_file:
                metadata_log_file.write("\n".join(
                    "{}: {}".format(field, value)
                    for field, value in metadata_log.items() if value))
        if world > 1:
            torch.distributed.barrier()
        return 0

    # -- noise -----------------------------------------------------------------
    def _noise_request(self, eps, samples, cells, global_cells, row_offset,
                       step):
        """The ``philox_normal_blocks`` arguments that fill ``eps`` (shape
        ``_eps_shape``: stacked passes of ``cells`` rows) so that global row g
        of pass s gets the draw keyed by (noise_seed, step, s*global_cells +
        g): the same for any sharding of the rows."""
        blocks = int(numpy.prod(self._eps_shape(samples, cells)[:-2]))
        return dict(out=eps.view(blocks, cells, self.latent_size),
                    block_stride=global_cells, row_offset=row_offset,
                    seed=self.noise_seed, stream_id=step)

    def _draw_noise(self, eps, samples, cells, global_cells, row_offset,
                    step):
        from scvae_amd.minibatch import philox_normal_blocks
        philox_normal_blocks(**self._noise_request(
            eps, samples, cells, global_cells, row_offset, step))

    # -- evaluation pass shared by train (epoch end) and evaluate --------------
    def _evaluation_pass(self, x, t, data_set, minibatch_size, n_iw, n_mc,
                         sync=None, deterministic_z=False, outputs=None):
        """Sequential minibatches, ``is_training=False``; returns the epoch
        averages ``sum(batch means) / (N / B)`` plus per-cell latent means."""
        engine = self.engine
        device = engine.device
        world, rank = _distributed()
        n = data_set.number_of_examples
        F, L = self.feature_size, self.latent_size
        samples = 1 if deterministic_z else n_iw * n_mc
        # whole minibatches share a step where nothing but the averages is
        # asked for (the epoch-end passes of train, a plain evaluate)
        chunks = mu.evaluation_chunks(
            n, minibatch_size,
            0 if outputs else self._evaluation_step_cells(samples))
        starts = [start for start, _, _ in chunks]
        largest = max((cells for _, cells, _ in chunks), default=0)
        weights = torch.tensor([w for _, _, w in chunks], device=device)
        scalars = torch.zeros(len(starts), 8, device=device)
        kl_neurons = torch.zeros(len(starts), L, device=device)
        latent = torch.zeros(n, L, device=device)
        extra = self._allocate_evaluation_outputs(n, len(starts), device)
        x_buffer = torch.empty(largest, F, device=device)
        t_buffer = x_buffer if x is t else torch.empty_like(x_buffer)
        row_const = torch.empty(largest, device=device)
        eps_buffer = None
        if not deterministic_z:
            eps_buffer = torch.empty(
                int(numpy.prod(self._eps_shape(samples, largest))),
                device=device)
        all_rows = torch.arange(n, device=device, dtype=torch.int64)
        # plain passes (no reconstruction statistics requested) over an integer
        # count matrix take the uint16 minibatch where the plan allows it
        u16_buffer = None
        if (not outputs and x is t and getattr(x, "integer_counts", False)
                and hasattr(x, "gather_counts_u16")
                and engine.accepts_counts_u16(largest, False)):
            u16_buffer = torch.empty(
                largest, x.u16_pitch, dtype=torch.uint16, device=device)
        # ... and, where it fits, from a RESIDENT dense copy of the whole set: an
        # evaluation pass walks the set in the same sequential minibatches every
        # epoch (va:1092-1150), so the uint16 rows written by the first pass are
        # the minibatches of every later one -- views, no fetch
        resident = (self._evaluation_resident(x, n)
                    if u16_buffer is not None
                    and data_set.noisy_preprocess is None else None)
        self._evaluation_counter = getattr(
            self, "_evaluation_counter", 0) + 1
        noise_stream = (1 << 40) + self._evaluation_counter * (1 << 20)
        mine = [(j, i, cells) for j, (i, cells, _) in enumerate(chunks)
                if not (world > 1 and j % world != rank)]
        # uint16 steps carry the fetch and the noise of the NEXT step
        # (scvae_side_work: they leave the stream after the input layer's
        # product and run beside the hidden layers); two sets of buffers
        carried = (u16_buffer is not None and len(mine) > 1 and all(
            engine.accepts_counts_u16(cells, False) for _, _, cells in mine))
        if carried:
            u16_sets = [u16_buffer, torch.empty_like(u16_buffer)
                        if resident is None else u16_buffer]
            rc_sets = [row_const, torch.empty_like(row_const)]
            eps_sets = [eps_buffer, torch.empty_like(eps_buffer)
                        if eps_buffer is not None else None]

            def minibatch_of(position):
                """(uint16 rows, row constants, already there) of a step."""
                _, first, count = mine[position]
                if resident is not None:
                    dense, constants, filled = resident
                    return (dense[first:first + count],
                            constants[first:first + count],
                            bool(filled[first:first + count].all()))
                slot = position & 1
                return u16_sets[slot][:count], rc_sets[slot][:count], False

            def carried_request(position):
                _, first, count = mine[position]
                slot = position & 1
                rows_out, constants_out, there = minibatch_of(position)
                request = None
                if not there:
                    request = x.request(all_rows[first:first + count],
                                        rows_out, constants_out)
                    if resident is not None:
                        resident[2][first:first + count] = True
                else:
                    self._evaluation_resident_hits += 1
                noise = None
                if not deterministic_z:
                    noise = self._noise_request(
                        eps_sets[slot][:int(numpy.prod(
                            self._eps_shape(samples, count)))],
                        samples, count, n, first, noise_stream)
                return request, noise
        for position, (j, i, cells) in enumerate(mine):
            rows = all_rows[i:i + cells]
            xb, tb, rc = x_buffer[:cells], t_buffer[:cells], row_const[:cells]
            eps = None
            next_request = next_noise = None
            if carried:
                slot = position & 1
                if position == 0:
                    request, noise = carried_request(0)
                    if request is not None:
                        request.issue()
                    if noise is not None:
                        from scvae_amd.minibatch import philox_normal_blocks
                        philox_normal_blocks(**noise)
                xb, rc, _ = minibatch_of(position)
                tb = xb
                if not deterministic_z:
                    eps = eps_sets[slot][:int(numpy.prod(
                        self._eps_shape(samples, cells)))]
                if position + 1 < len(mine):
                    next_request, next_noise = carried_request(position + 1)
            elif (u16_buffer is not None
                    and engine.accepts_counts_u16(cells, False)):
                if resident is not None:
                    dense, constants, filled = resident
                    xb, rc = dense[i:i + cells], constants[i:i + cells]
                    if filled[i:i + cells].all():
                        self._evaluation_resident_hits += 1
                    else:
                        x.gather_counts_u16(rows, out=xb, row_const_out=rc)
                        filled[i:i + cells] = True
                    tb = xb
                else:
                    xb = tb = x.gather_counts_u16(
                        rows, out=u16_buffer[:cells], row_const_out=rc)
            else:
                t.gather_dense(rows, out=tb, row_const_out=rc)
                if x is not t:
                    x.gather_dense(rows, out=xb)
            if not deterministic_z and not carried:
                eps = eps_buffer[:int(numpy.prod(
                    self._eps_shape(samples, cells)))]
                # (keyed by the cell's row in the set, not by the step: a
                #  cell draws the same noise however the pass is cut into
                #  steps or dealt to ranks)
                self._draw_noise(eps, samples, cells, n, i, noise_stream)
            out = {"q_z_mean": latent[i:i + cells],
                   "kl_neurons": kl_neurons[j]}
            out.update(self._evaluation_step_outputs(
                extra, outputs, i, j, cells))
            de = (t.decoder_extra.index_select(0, rows)
                  if getattr(t, "decoder_extra", None) is not None else None)
            cs = (t.count_sum.index_select(0, rows)
                  if getattr(t, "count_sum", None) is not None else None)
            engine.step(xb, tb, eps=eps, row_const=rc, training=False,
                        n_iw=n_iw, n_mc=n_mc, deterministic_z=deterministic_z,
                        outputs=out, scalars=scalars[j], decoder_extra=de,
                        count_sum=cs, x_counts=x.integer_counts,
                        next_minibatch=next_request, next_noise=next_noise)
        # a step's means count once per minibatch it holds
        scalars *= weights[:, None]
        kl_neurons *= weights[:, None]
        self._weight_evaluation_outputs(extra, weights)
        if sync is not None:
            for tensor in [scalars, kl_neurons, latent] + [
                    v for v in extra.values() if torch.is_tensor(v)]:
                torch.distributed.all_reduce(tensor)
        totals = scalars.sum(dim=0).cpu().numpy().astype(numpy.float64)
        denominator = n / minibatch_size
        result = {}
        for index, tag, _ in self._loss_tags():
            result[tag] = float(totals[index] / denominator)
        result["kl_divergence_neurons"] = (
            kl_neurons.sum(dim=0).cpu().numpy() / denominator)
        result["latent_values"] = latent.cpu().numpy()
        self._finish_evaluation(result, extra, data_set, denominator)
        return result

    # the dense uint16 copy of an evaluation set an epoch-end pass leaves behind for
    # the next epoch's (bytes; 0: none).  68 579 x 32 738 cells x genes: 4.5 GB of
    # the 288; never more than a quarter of what is free when the set is first met
    evaluation_resident_bytes = 64 << 30
    _evaluation_resident_hits = 0     # steps that found their minibatch there

    def _evaluation_resident(self, x, n):
        """(dense uint16 [n, pitch], row constants [n], filled [n] host flags)
        kept on the device matrix ``x``, or None where it does not apply."""
        import numpy
        if not int(self.evaluation_resident_bytes):
            return None
        held = getattr(x, "_evaluation_resident", None)
        if held is not None:
            return held if held[0].shape[0] == n else None
        if getattr(x, "_evaluation_resident_refused", False):
            return None
        need = n * x.u16_pitch * 2 + n * 4
        free, _ = torch.cuda.mem_get_info(self.engine.device)
        if need > min(int(self.evaluation_resident_bytes), free // 4):
            x._evaluation_resident_refused = True
            return None
        device = self.engine.device
        x._evaluation_resident = (
            torch.empty(n, x.u16_pitch, dtype=torch.uint16, device=device),
            torch.empty(n, device=device), numpy.zeros(n, dtype=bool))
        return x._evaluation_resident

    # cells per evaluation step where whole minibatches may share one (0: one
    # step per minibatch, as the reference runs them); the stacked passes of a
    # step stay within what the training workloads run
    evaluation_chunk_cells = 4096
    # the state of an epoch's checkpoint leaves the device on a second stream
    # (Engine.state_dict(non_blocking=True)); False: the blocking copy
    checkpoint_non_blocking = True
    evaluation_chunk_stacked_rows = 16384

    def _evaluation_step_cells(self, samples):
        passes = int(numpy.prod(self._eps_shape(samples, 1)[:-2]))
        cells = min(int(self.evaluation_chunk_cells),
                    int(self.evaluation_chunk_stacked_rows) // max(passes, 1))
        # plans off the fused likelihood kernels (-k > 2, head dropout, ...)
        # keep [rows, F] buffers per head: a step of 4096 cells is then GBs of
        # workspace a step of one minibatch never needed.  Halve the step
        # until its workspace fits half of what is free (below 2 B cells
        # evaluation_chunks runs one step per minibatch, as the reference).
        engine = self._engine
        if engine is not None and cells > 0 and engine.device.type == "cuda":
            free, _ = torch.cuda.mem_get_info(engine.device)
            held = engine.workspace.numel() if engine.workspace is not None else 0
            budget = held + free // 2
            while cells > 1 and engine.lib.scvae_plan_workspace_bytes(
                    engine.handle, cells, max(int(samples), 1)) > budget:
                cells //= 2
        return cells

    def _evaluation_largest_step(self, n, minibatch_size, samples):
        return max((cells for _, cells, _ in mu.evaluation_chunks(
            n, minibatch_size, self._evaluation_step_cells(samples))),
            default=0)

    def _allocate_evaluation_outputs(self, n, n_batches, device):
        return {}

    def _weight_evaluation_outputs(self, extra, weights):
        pass

    def _evaluation_step_outputs(self, extra, outputs, i, j, cells):
        out = {}
        if outputs:
            for key, tensor in outputs.items():
                out[key] = tensor[i:i + cells]
        return out

    def _finish_evaluation(self, result, extra, data_set, denominator):
        pass

    def _print_extra(self, say, evaluation, data_set):
        pass

    def _format_losses(self, evaluation):
        return ", ".join(
            "{}: {:.5g}".format(label, evaluation[tag])
            for _, tag, label in self._loss_tags()) + "."

    def _centroids(self, prior):
        return None

    def _write_epoch_summary(self, writer, evaluation, global_step, prior,
                             kl_neurons):
        scalars = {}
        for _, tag, _ in self._loss_tags():
            scalars["losses/" + tag] = evaluation[tag]
        self._extra_summary(scalars, evaluation)
        if kl_neurons:
            for i, value in enumerate(
                    numpy.atleast_1d(evaluation["kl_divergence_neurons"])):
                scalars["kl_divergence_neurons/{}".format(i)] = value
        if prior is not None:
            probabilities, means, variances = prior
            for k in range(len(probabilities)):
                scalars["prior/cluster_{}/probability".format(k)] = (
                    probabilities[k])
                for l in range(self.latent_size):
                    scalars["prior/cluster_{}/mean/dimension_{}".format(
                        k, l)] = numpy.ravel(means[k])[
                            l if numpy.ndim(means[k]) else 0]
                    scalars["prior/cluster_{}/variance/dimension_{}".format(
                        k, l)] = numpy.ravel(variances[k])[
                            l if numpy.ndim(variances[k]) else 0]
        writer.add_summary(scalars, global_step=global_step)
        writer.flush()

    def _extra_summary(self, scalars, evaluation):
        pass

    # ======================================================================
    # evaluation
    # ======================================================================
    def evaluate(self, evaluation_set, minibatch_size=None, run_id=None,
                 use_early_stopping_model=False, use_best_model=False,
                 **kwargs):
        """Evaluate trained model (signature and returns of va:1781-2217)."""
        from scvae_amd.data import DataSet
        if minibatch_size is None:
            minibatch_size = defaults["models"]["minibatch_size"]
        if run_id is None:
            run_id = defaults["models"]["run_id"]
        if run_id:
            run_id = mu.check_run_id(run_id)
            model_string = "model for run {}".format(run_id)
        else:
            model_string = "model"

        output_versions = kwargs.get("output_versions")
        if output_versions is None:
            output_versions = "all"
        if output_versions == "all":
            output_versions = ["transformed", "reconstructed", "latent"]
        elif not isinstance(output_versions, list):
            output_versions = [output_versions]
        valid = {"transformed", "reconstructed", "latent"}
        if len(output_versions) > 3 or not set(output_versions) <= valid:
            raise ValueError(
                "Can only output at most 3 sets, either the transformed, "
                "the reconstructed, or the latent set.")
        evaluation_subset_indices = kwargs.get("evaluation_subset_indices")
        if evaluation_subset_indices is None:
            evaluation_subset_indices = set()
        log_results = kwargs.get("log_results", True)
        use_deterministic_z = kwargs.get("use_deterministic_z", False)

        minibatch_size = self._training_minibatch_size(
            minibatch_size, "evaluation")
        n_examples_eval = evaluation_set.number_of_examples
        n_features_eval = evaluation_set.number_of_features

        log_directory = self.log_directory(
            run_id=run_id, early_stopping=use_early_stopping_model,
            best_model=use_best_model)
        checkpoint = mu.get_checkpoint_state(log_directory)
        if not checkpoint:
            raise Exception(
                "Cannot evaluate {} when it has not been trained.".format(
                    model_string))
        world, rank = _distributed()
        master = rank == 0
        engine = self.engine
        engine.load_state_dict(mu.load_checkpoint(checkpoint))
        epoch = mu.checkpoint_epoch(checkpoint)
        # (va:1861-1885: a noisy preprocessor draws the evaluation set once)
        noisy_preprocess = evaluation_set.noisy_preprocess is not None
        if noisy_preprocess and master:
            print("Noisily preprocess values.")
        noisy_time_start = time()
        x_eval, t_eval = self._device_matrices(
            evaluation_set, noisy=noisy_preprocess)
        if noisy_preprocess and master:
            print("Values noisily preprocessed ({}).".format(
                format_duration(time() - noisy_time_start)))
            print()

        if log_results and master:
            eval_summary_directory = os.path.join(log_directory, "evaluation")
            if os.path.exists(eval_summary_directory):
                shutil.rmtree(eval_summary_directory)
            eval_summary_writer = mu.ScalarWriter(eval_summary_directory)

        data_string = mu.build_data_string(
            evaluation_set, self.reconstruction_distribution_name)
        if master:
            print("Evaluating trained {} on {}.".format(
                model_string, data_string))
        evaluating_time_start = time()

        device = engine.device
        outputs = {}
        if "reconstructed" in output_versions:
            # the reference's "15 GB dense reconstructed test set"
            # (docs/guide.rst:61) stays in HBM until the single copy below
            # (zeros, not empty: with several ranks each one fills the rows of
            # its own minibatches and the all-reduce(sum) below assembles the rest)
            allocate = torch.zeros if world > 1 else torch.empty
            for key in ("p_x_mean", "p_x_stddev",
                        "stddev_of_p_x_given_z_mean"):
                outputs[key] = allocate(
                    n_examples_eval, n_features_eval, device=device)
        if use_deterministic_z:
            n_iw = n_mc = 1
        else:
            n_iw = self.number_of_importance_samples["evaluation"]
            n_mc = self.number_of_monte_carlo_samples["evaluation"]
        sync = object() if world > 1 else None
        evaluation = self._evaluation_pass(
            x_eval, t_eval, evaluation_set, minibatch_size, n_iw, n_mc,
            sync=sync, deterministic_z=use_deterministic_z, outputs=outputs)
        if world > 1:
            for tensor in outputs.values():
                torch.distributed.all_reduce(tensor)
        evaluating_duration = time() - evaluating_time_start

        if log_results and master:
            self._write_epoch_summary(
                eval_summary_writer, evaluation, epoch,
                prior=self._prior_summary(), kl_neurons=True)
        if master:
            print("    {} set ({}): ".format(
                evaluation_set.kind.capitalize(),
                format_duration(evaluating_duration)),
                self._format_losses(evaluation))
            self._print_extra(print, evaluation, evaluation_set)

        def wrap(values, version, feature_names=None, **extra):
            return DataSet(
                evaluation_set.name, title=evaluation_set.title,
                specifications=evaluation_set.specifications, values=values,
                preprocessed_values=None, labels=evaluation_set.labels,
                example_names=evaluation_set.example_names,
                feature_names=(feature_names if feature_names is not None
                               else evaluation_set.feature_names),
                batch_indices=evaluation_set.batch_indices,
                batch_names=evaluation_set.batch_names,
                features_mapped=evaluation_set.features_mapped,
                feature_selection=evaluation_set.feature_selection,
                example_filter=evaluation_set.example_filter,
                preprocessing_methods=evaluation_set.preprocessing_methods,
                kind=evaluation_set.kind, version=version, **extra)

        output_sets = [None] * len(output_versions)
        if "transformed" in output_versions:
            # (va:2135-2158: the binarised / noisily preprocessed values the
            #  likelihood saw, where those are not the set's own)
            output_sets[output_versions.index("transformed")] = (
                wrap(t_eval.transformed_values, "transformed")
                if t_eval.transformed_values is not None else evaluation_set)
        if "reconstructed" in output_versions:
            import scipy.sparse
            p_x_mean_eval = outputs["p_x_mean"].cpu().numpy()
            p_x_stddev_eval = scipy.sparse.lil_matrix(
                (n_examples_eval, n_features_eval), dtype=numpy.float32)
            stddev_of_p_x_mean_eval = scipy.sparse.lil_matrix(
                (n_examples_eval, n_features_eval), dtype=numpy.float32)
            subset = numpy.array(sorted(evaluation_subset_indices),
                                 dtype=numpy.int64)
            if subset.size > 0:
                index = torch.from_numpy(subset).to(device)
                p_x_stddev_eval[subset] = (
                    outputs["p_x_stddev"][index].cpu().numpy())
                stddev_of_p_x_mean_eval[subset] = (
                    outputs["stddev_of_p_x_given_z_mean"][index]
                    .cpu().numpy())
            output_sets[output_versions.index("reconstructed")] = wrap(
                p_x_mean_eval, "reconstructed",
                total_standard_deviations=p_x_stddev_eval,
                explained_standard_deviations=stddev_of_p_x_mean_eval)
        if "latent" in output_versions:
            latent_names = numpy.array([
                "latent variable {}".format(i + 1)
                for i in range(self.latent_size)])
            latent_sets = self._latent_evaluation_sets(
                evaluation, wrap, latent_names)
            output_sets[output_versions.index("latent")] = latent_sets
        self._attach_predictions(output_sets, evaluation, evaluation_set)
        if len(output_sets) == 1:
            output_sets = output_sets[0]
        return output_sets

    def _attach_predictions(self, output_sets, evaluation, evaluation_set):
        """Models that cluster by themselves (GMVAE, gm:2744-2781) label every
        output set with their cluster assignment."""

    def _latent_evaluation_sets(self, evaluation, wrap, latent_names):
        return {"z": wrap(evaluation["latent_values"], "z",
                          feature_names=latent_names)}

    def _sample_prior(self, count, seed, stream_id):
        """Draw ``count`` latent values from the prior on the device.  Returns
        ``(z, extra)``: z [count, L] is decoded; ``extra`` maps further latent
        set names to [count, ...] tensors."""
        raise NotImplementedError

    def sample(self, sample_size=None, minibatch_size=None, run_id=None,
               use_early_stopping_model=False, use_best_model=False):
        """Sample from trained model: z ~ p(z), x_mean = E[p(x|z)] with the
        decoder in evaluation mode (signature and returns of va:1601-1779 /
        gm:1949-2160)."""
        from scvae_amd.data import DataSet
        from scvae_amd.utilities import normalise_string
        if sample_size is None:
            sample_size = defaults["models"]["sample_size"]
        if minibatch_size is None:
            minibatch_size = defaults["models"]["minibatch_size"]
        if run_id is None:
            run_id = defaults["models"]["run_id"]
        if run_id:
            run_id = mu.check_run_id(run_id)
            model_string = "model for run {}".format(run_id)
        else:
            model_string = "model"
        if self.batch_correction:   # as va:1639-1650
            raise NotImplementedError("Sampling with batch correction.")
        if self.use_count_sum_as_parameter:
            raise NotImplementedError(
                "Sampling with count sum as reconstruction distribution "
                "parameter.")
        if self.use_count_sum_as_feature:
            raise NotImplementedError(
                "Sampling with count sum as additional latent feature.")

        log_directory = self.log_directory(
            run_id=run_id, early_stopping=use_early_stopping_model,
            best_model=use_best_model)
        checkpoint = mu.get_checkpoint_state(log_directory)
        if not checkpoint:
            raise Exception(
                "Cannot evaluate {} when it has not been trained.".format(
                    model_string))
        engine = self.engine
        engine.load_state_dict(mu.load_checkpoint(checkpoint))

        print("Sampling {} examples from {}.".format(
            sample_size, model_string))
        sampling_time_start = time()
        device = engine.device
        x_mean = torch.empty(sample_size, self.feature_size, device=device)
        latent = {}
        for j, i in enumerate(range(0, sample_size, minibatch_size)):
            count = min(minibatch_size, sample_size - i)
            z, extra = self._sample_prior(count, self.noise_seed,
                                          (1 << 41) + j)
            engine.decode(z, out=x_mean[i:i + count])
            extra = dict(extra, z=z)
            for key, value in extra.items():
                if key not in latent:
                    latent[key] = torch.empty(
                        (sample_size,) + tuple(value.shape[1:]),
                        dtype=value.dtype, device=device)
                latent[key][i:i + count] = value
        torch.cuda.synchronize(device)
        print("Examples sampled ({}).".format(format_duration(
            time() - sampling_time_start)))

        title = "Sampled data set"
        name = normalise_string(title)
        sample_names = numpy.array([
            "sample {}".format(i + 1) for i in range(sample_size)])
        feature_names = numpy.array([
            "feature {}".format(i + 1) for i in range(self.feature_size)])

        def data_set(values, version, names):
            return DataSet(
                name, title=title, specifications=dict(),
                values=values.cpu().numpy(), preprocessed_values=None,
                labels=None, example_names=sample_names, feature_names=names,
                batch_indices=None, feature_selection=None,
                example_filter=None, preprocessing_methods=None,
                kind="sample", version=version)
        sample_reconstruction_set = data_set(
            x_mean, "reconstructed", feature_names)
        sample_latent_sets = {
            key: data_set(values, key, numpy.array([
                self._latent_feature_name(key, i)
                for i in range(values.shape[1])]))
            for key, values in latent.items()}
        return sample_reconstruction_set, sample_latent_sets

    def _latent_feature_name(self, key, index):
        return "latent variable {}".format(index + 1)
