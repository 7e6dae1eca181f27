"""Bookkeeping around the training loop: run IDs, checkpoint directories,
the scalar (learning-curve) store and its readers, early-stopping status and
constructor-argument validation.

Restates the non-TensorFlow behaviour of ``scvae/models/utilities.py:140-897``.
The reference keeps its learning curves in TensorFlow event files and reads
them back with ``tf.train.summary_iterator`` (``:908-951``); here the same tags
(``losses/lower_bound`` ...) are appended to ``<log_dir>/<kind>/scalars.jsonl``
and checkpoints are ``model.ckpt-<epoch>.pt`` files indexed by a ``checkpoint``
file, so ``evaluate`` and resumed training find the same information.
"""

import json
import os
import random
import re
import shutil
import time
from datetime import datetime
from string import ascii_uppercase

import numpy

from scvae_amd.utilities import (
    capitalise_string, enumerate_strings)

CHECKPOINT_INDEX = "checkpoint"
CHECKPOINT_PREFIX = "model.ckpt"
SCALARS_FILE = "scalars.jsonl"


# ------------------------------ messages ----------------------------------

def build_training_string(model_string, epoch_start, number_of_epochs,
                          data_string):
    if epoch_start == 0:
        return "Training {} for {} epochs on {}.".format(
            model_string, number_of_epochs, data_string)
    if epoch_start < number_of_epochs:
        return ("Continue training {} for {} additionally epochs (up to {} "
                "epochs) on {}.".format(
                    model_string, number_of_epochs - epoch_start,
                    number_of_epochs, data_string))
    if epoch_start == number_of_epochs:
        return "{} has already been trained for {} epochs on {}.".format(
            capitalise_string(model_string), number_of_epochs, data_string)
    return ("{} has already been trained for more than {} epochs on {}. "
            "Loading model trained for {} epochs.".format(
                capitalise_string(model_string), number_of_epochs,
                data_string, epoch_start))


def build_data_string(data_set, reconstruction_distribution_name):
    if not data_set.noisy_preprocessing_methods:
        if data_set.preprocessing_methods:
            if data_set.preprocessing_methods == ["binarise"]:
                data_string = "binarised values"
            else:
                data_string = "preprocessed values"
        else:
            data_string = "original values"
        if reconstruction_distribution_name == "bernoulli":
            if data_string != "binarised values":
                data_string += " with binarised values as targets"
        elif data_string != "original values":
            data_string += " with original values as targets"
    else:
        if data_set.noisy_preprocessing_methods == ["binarise"]:
            data_string = "new Bernoulli-sampled values"
        else:
            data_string = "new preprocessed values"
        data_string += " at every epoch"
    return data_string


# ------------------------------ run IDs ------------------------------------

def check_run_id(run_id):
    if run_id is None:
        raise TypeError("The run ID has not been set.")
    run_id = str(run_id)
    if not re.fullmatch(r"[\w]+", run_id):
        raise ValueError(
            "`run_id` can only contain letters, numbers, and "
            "underscores ('_').")
    return run_id


def _generate_run_id(timestamp=None, number_of_letters=2):
    if timestamp is None:
        timestamp = time.time()
    stamp = datetime.fromtimestamp(timestamp).strftime("%Y%m%dT%H%M%S")
    letters = "".join(random.choices(ascii_uppercase, k=number_of_letters))
    return stamp + "_" + letters


def generate_unique_run_id_for_model(model, timestamp=None):
    log_directory = model.log_directory()
    existing = []
    if os.path.isdir(log_directory):
        existing = [re.sub(r"^run_", "", d) for d in os.listdir(log_directory)
                    if d.startswith("run_")]
    while True:
        run_id = _generate_run_id(timestamp=timestamp)
        if run_id not in existing:
            return run_id


# ------------------------------ checkpoints --------------------------------

def get_checkpoint_state(log_directory):
    """Path of the latest checkpoint in ``log_directory`` or ``None``
    (stand-in for ``tf.train.get_checkpoint_state``)."""
    index = os.path.join(log_directory, CHECKPOINT_INDEX)
    if not os.path.exists(index):
        return None
    try:
        with open(index) as f:
            name = json.load(f).get("model_checkpoint_path")
    except ValueError:   # a truncated index counts as no checkpoint
        return None
    if not name:
        return None
    path = os.path.join(log_directory, os.path.basename(name))
    return path if os.path.exists(path) else None


def checkpoint_epoch(checkpoint_path):
    stem = os.path.basename(checkpoint_path)
    stem = stem[:-3] if stem.endswith(".pt") else stem
    return int(stem.split("-")[-1])


def save_checkpoint(state, log_directory, epoch):
    import torch
    os.makedirs(log_directory, exist_ok=True)
    name = "{}-{}.pt".format(CHECKPOINT_PREFIX, epoch)
    # write-then-rename, the state before the index that points at it: an
    # interrupted save leaves the previous checkpoint and its index intact
    path = os.path.join(log_directory, name)
    # temporaries of an earlier, interrupted save are full-size state files: drop them
    for stale in os.listdir(log_directory):
        if stale.endswith(".tmp") and stale.startswith(
                (CHECKPOINT_PREFIX, CHECKPOINT_INDEX)):
            try:
                os.remove(os.path.join(log_directory, stale))
            except OSError:
                pass
    # (flushed to the device before the rename: after a power loss the index must
    #  not point at a truncated state file)
    with open(path + ".tmp", "wb") as f:
        torch.save(state, f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(path + ".tmp", path)
    index = os.path.join(log_directory, CHECKPOINT_INDEX)
    with open(index + ".tmp", "w") as f:
        json.dump({"model_checkpoint_path": name}, f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(index + ".tmp", index)
    remove_old_checkpoints(log_directory)  # Saver(max_to_keep=1)
    return os.path.join(log_directory, name)


class CheckpointWriter:
    """The file-system side of an epoch on a background thread, in order: the
    training loop only pays for a snapshot of the state in device memory
    (``engine.state_dict(non_blocking=True)``: the copy to pinned host memory
    runs on a second stream and is waited for by the queue); serialising it,
    the copies into ``early_stopping/`` and ``best/`` (va:1385-1441, 1470-1492)
    and the pruning of old files overlap the next epoch.  Jobs run strictly in the order they
    were queued, so "the latest checkpoint of a directory" means what it means
    in the reference's sequential loop.  Whoever reads the files calls
    ``wait()`` first; at most two states wait in memory."""

    MAX_PENDING_SAVES = 2

    def __init__(self):
        import atexit
        import queue
        atexit.register(self.close)   # (an aborted run still finishes its queued saves)
        self._jobs = queue.Queue()
        self._thread = None
        self._error = None
        self._delivered = False
        self._pending_saves = 0
        import threading
        self._lock = threading.Lock()

    def _worker(self):
        while True:
            job = self._jobs.get()
            try:
                if job is None:
                    return
                if self._error is None:
                    job()
                else:
                    # after a failure the queue is drained without working --
                    # but a skipped save still gives its slot back, or save()
                    # would wait for it for ever
                    skipped = getattr(job, "on_skip", None)
                    if skipped is not None:
                        skipped()
            except BaseException as error:   # raised by the next call from the loop
                if self._error is None:
                    self._error = error
            finally:
                self._jobs.task_done()

    def _raise_if_failed(self):
        """The first failure of a queued job, raised from every later call (it
        stays: nothing queued after it has run, so the files on disk are those of
        before the failure and the run must not carry on as if they were not)."""
        if self._error is not None:
            self._delivered = True
            raise self._error

    def _submit(self, job):
        import threading
        self._raise_if_failed()
        if self._thread is None or not self._thread.is_alive():
            self._thread = threading.Thread(target=self._worker, daemon=True)
            self._thread.start()
        self._jobs.put(job)

    def save(self, state, log_directory, epoch):
        self._raise_if_failed()
        while self._pending_saves >= self.MAX_PENDING_SAVES:
            self.wait()

        def work():
            try:
                # (a state still travelling to the host -- Engine.state_dict(
                #  non_blocking=True) -- is waited for here, off the training
                #  loop, and written as the plain dictionary it then is)
                ready = (dict(state.wait()) if hasattr(state, "wait")
                         else state)
                save_checkpoint(ready, log_directory, epoch)
            finally:
                release()

        def release():
            with self._lock:
                self._pending_saves -= 1
        work.on_skip = release
        with self._lock:
            self._pending_saves += 1
        try:
            self._submit(work)
        except BaseException:
            release()
            raise

    def copy_latest(self, log_directory, output_directory, prune=False):
        """``copy_model_directory`` of the checkpoint that is the latest of
        ``log_directory`` when the job runs (no checkpoint: nothing), with the
        logs as they are NOW."""
        logs = snapshot_logs(log_directory)

        def work():
            latest = get_checkpoint_state(log_directory)
            if latest:
                copy_model_directory(latest, output_directory, logs=logs)
                if prune:
                    remove_old_checkpoints(output_directory)
        self._submit(work)

    def remove_tree(self, directory):
        def work():
            if os.path.exists(directory):
                shutil.rmtree(directory)
        self._submit(work)

    def wait(self):
        if self._thread is not None:
            self._jobs.join()
        self._raise_if_failed()

    def close(self):
        import atexit
        atexit.unregister(self.close)
        try:
            if self._thread is not None:
                self._jobs.join()
            error, self._error = self._error, None
            if error is not None and not self._delivered:   # (nobody has seen it yet)
                raise error
        finally:
            if self._thread is not None:
                self._jobs.put(None)
                self._thread.join()
                self._thread = None


def load_checkpoint(checkpoint_path):
    import torch
    # the state is tensors plus one int: nothing that needs unpickling of
    # arbitrary objects
    return torch.load(checkpoint_path, map_location="cpu",
                      weights_only=True)


SMALL_LOG_FILE_BYTES = 1 << 20


def snapshot_logs(source):
    """The small text files that travel with a checkpoint -- ``*.log`` and the
    scalar stores under ``training/`` and ``validation/`` -- as {relative path:
    bytes}, read NOW (the background copy of the checkpoint must not see the
    records later epochs append)."""
    files = {}
    if not os.path.isdir(source):
        return files
    for entry in os.listdir(source):
        path = os.path.join(source, entry)
        # (every small top-level file travels -- the reference copies the whole
        #  directory, va:1385-1441 --, except the checkpoints themselves and
        #  their index, which copy_model_directory places)
        if (os.path.isfile(path) and entry != CHECKPOINT_INDEX
                and not entry.endswith((".pt", ".tmp"))
                and os.path.getsize(path) <= SMALL_LOG_FILE_BYTES):
            with open(path, "rb") as f:
                files[entry] = f.read()
        elif os.path.isdir(path) and entry in ("training", "validation"):
            for root, _, names in os.walk(path):
                for name in names:
                    full = os.path.join(root, name)
                    with open(full, "rb") as f:
                        files[os.path.relpath(full, source)] = f.read()
    return files


def copy_model_directory(checkpoint_path, output_directory, logs=None):
    """Copy the checkpoint and the scalar logs next to it into
    ``output_directory`` (``early_stopping/`` and ``best/`` are such copies).
    ``logs``: a ``snapshot_logs`` of the source taken earlier (default: now)."""
    source = os.path.dirname(checkpoint_path)
    if logs is None:
        logs = snapshot_logs(source)
    if os.path.exists(output_directory):
        shutil.rmtree(output_directory)
    os.makedirs(output_directory)
    name = os.path.basename(checkpoint_path)
    # INVARIANT: a state file is immutable -- save_checkpoint writes a temporary
    # file and renames it into place, nothing ever rewrites one in place -- so a
    # hard link is a copy that cannot change under `best/` or `early_stopping/`.
    # Anything that edits checkpoints in place (an external tool) must break
    # the link first.  Across file systems: a real copy.
    try:
        os.link(checkpoint_path, os.path.join(output_directory, name))
    except OSError:
        shutil.copy2(checkpoint_path, os.path.join(output_directory, name))
    with open(os.path.join(output_directory, CHECKPOINT_INDEX), "w") as f:
        json.dump({"model_checkpoint_path": name}, f)
    for relative, data in logs.items():
        target = os.path.join(output_directory, relative)
        os.makedirs(os.path.dirname(target), exist_ok=True)
        with open(target, "wb") as f:
            f.write(data)


def remove_old_checkpoints(log_directory):
    latest = get_checkpoint_state(log_directory)
    if latest is None:
        return
    keep = os.path.basename(latest)
    for entry in os.listdir(log_directory):
        if (entry.startswith(CHECKPOINT_PREFIX + "-") and entry != keep
                and not entry.endswith(".tmp")):
            os.remove(os.path.join(log_directory, entry))


def clear_log_directory(log_directory):
    remove_only_if_empty = ("early_stopping", "best")
    for entry in os.listdir(log_directory):
        path = os.path.join(log_directory, entry)
        if os.path.isdir(path) and entry.startswith("run_"):
            continue
        if os.path.isdir(path):
            shutil.rmtree(path)
        else:
            os.remove(path)
    del remove_only_if_empty


# ------------------------------ scalar store -------------------------------

class ScalarWriter:
    """Append-only store of per-epoch scalars (one JSON object per line)."""

    def __init__(self, directory):
        self.directory = directory
        os.makedirs(directory, exist_ok=True)
        self.path = os.path.join(directory, SCALARS_FILE)

    def add_summary(self, scalars, global_step):
        record = {"step": int(global_step), "wall_time": time.time(),
                  "scalars": {k: float(v) for k, v in scalars.items()}}
        with open(self.path, "a") as f:
            f.write(json.dumps(record) + "\n")

    def flush(self):
        pass


def _read_scalars(directory):
    path = os.path.join(directory, SCALARS_FILE)
    if not os.path.exists(path):
        return []
    records = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                try:
                    records.append(json.loads(line))
                except ValueError:   # a torn last line (interrupted append)
                    continue
    # later records for the same step win (resumed / repeated epochs)
    by_step = {}
    for record in records:
        by_step[record["step"]] = record
    return [by_step[s] for s in sorted(by_step)]


def _resolve_log_directory(model, run_id, early_stopping, best_model,
                           log_directory):
    if log_directory is None:
        log_directory = model.log_directory(
            run_id=run_id, early_stopping=early_stopping,
            best_model=best_model)
    return log_directory


def load_number_of_epochs_trained(model, run_id=None, early_stopping=False,
                                  best_model=False):
    log_directory = model.log_directory(
        run_id=run_id, early_stopping=early_stopping, best_model=best_model)
    records = _read_scalars(os.path.join(log_directory, "training"))
    steps = [r["step"] for r in records
             if "losses/lower_bound" in r["scalars"]]
    return max(steps) if steps else None


def load_learning_curves(model, data_set_kinds="all", run_id=None,
                         early_stopping=False, best_model=False,
                         log_directory=None):
    """``{kind: {loss: array over epochs}}`` (or the single kind's dict)."""
    learning_curve_sets = {}
    if data_set_kinds == "all":
        data_set_kinds = ["training", "validation"]
    single = not isinstance(data_set_kinds, list)
    if single:
        data_set_kinds = [data_set_kinds]
    if "AE" in model.type:
        losses = ["lower_bound", "reconstruction_error", "kl_divergence"]
        if model.type == "GMVAE":
            losses += ["kl_divergence_z", "kl_divergence_y"]
    else:
        losses = ["log_likelihood"]
    log_directory = _resolve_log_directory(
        model, run_id, early_stopping, best_model, log_directory)
    for kind in data_set_kinds:
        records = _read_scalars(os.path.join(log_directory, kind))
        curves = {}
        for loss in losses:
            tag = "losses/" + loss
            values = [r["scalars"][tag] for r in records
                      if tag in r["scalars"]]
            curves[loss] = numpy.array(values) if values else None
        learning_curve_sets[kind] = curves
    if single:
        return learning_curve_sets[data_set_kinds[0]]
    return learning_curve_sets


def load_accuracies(model, data_set_kinds="all", superset=False, run_id=None,
                    early_stopping=False, best_model=False):
    if data_set_kinds == "all":
        data_set_kinds = ["training", "validation"]
    single = not isinstance(data_set_kinds, list)
    if single:
        data_set_kinds = [data_set_kinds]
    tag = "superset_accuracy" if superset else "accuracy"
    log_directory = model.log_directory(
        run_id=run_id, early_stopping=early_stopping, best_model=best_model)
    accuracies = {}
    for kind in data_set_kinds:
        records = _read_scalars(os.path.join(log_directory, kind))
        values = [r["scalars"][tag] for r in records if tag in r["scalars"]]
        accuracies[kind] = numpy.array(values) if values else None
    if single:
        return accuracies[data_set_kinds[0]]
    return accuracies


def load_kl_divergences(model, data_set_kind=None, run_id=None,
                        early_stopping=False, best_model=False):
    """[epochs, latent] array of ``kl_divergence_neurons/<i>``."""
    if data_set_kind is None:
        data_set_kind = "training"
    log_directory = model.log_directory(
        run_id=run_id, early_stopping=early_stopping, best_model=best_model)
    records = _read_scalars(os.path.join(log_directory, data_set_kind))
    rows = []
    for r in records:
        tags = sorted(
            (t for t in r["scalars"] if t.startswith("kl_divergence_neurons/")),
            key=lambda t: int(t.split("/")[-1]))
        if tags:
            rows.append([r["scalars"][t] for t in tags])
    return numpy.array(rows) if rows else None


def load_centroids(model, data_set_kinds="all", run_id=None,
                   early_stopping=False, best_model=False):
    """Prior centroids per epoch: ``{kind: {"prior": {"probabilities",
    "means", "covariance_matrices"}}}`` from the ``prior/cluster_k/...`` tags."""
    if data_set_kinds == "all":
        data_set_kinds = ["training", "validation"]
    single = not isinstance(data_set_kinds, list)
    if single:
        data_set_kinds = [data_set_kinds]
    log_directory = model.log_directory(
        run_id=run_id, early_stopping=early_stopping, best_model=best_model)
    K = getattr(model, "number_of_latent_clusters", 1) or 1
    L = model.latent_size
    sets = {}
    for kind in data_set_kinds:
        records = [r for r in _read_scalars(os.path.join(log_directory, kind))
                   if "prior/cluster_0/probability" in r["scalars"]]
        if not records:
            sets[kind] = None
            continue
        E = len(records)
        probabilities = numpy.empty((E, K))
        means = numpy.empty((E, K, L))
        covariances = numpy.zeros((E, K, L, L))
        for e, r in enumerate(records):
            s = r["scalars"]
            for k in range(K):
                probabilities[e, k] = s["prior/cluster_{}/probability".format(k)]
                for l in range(L):
                    means[e, k, l] = s[
                        "prior/cluster_{}/mean/dimension_{}".format(k, l)]
                    covariances[e, k, l, l] = s[
                        "prior/cluster_{}/variance/dimension_{}".format(k, l)]
        sets[kind] = {"prior": {"probabilities": probabilities,
                                "means": means,
                                "covariance_matrices": covariances}}
    if single:
        return sets[data_set_kinds[0]]
    return sets


# ------------------------------ early stopping -----------------------------

def early_stopping_status(losses, early_stopping_rounds):
    """Replay of the early-stopping rule on a validation curve
    (``scvae/models/utilities.py:591-612``)."""
    epochs_without_improvement = 0
    stopped_early = False
    if losses is not None:
        for epoch in range(1, len(losses)):
            if losses[epoch] < losses[epoch - 1]:
                epochs_without_improvement += 1
            else:
                epochs_without_improvement = 0
            if epochs_without_improvement >= early_stopping_rounds:
                stopped_early = True
                epochs_without_improvement = numpy.nan
                break
    return stopped_early, epochs_without_improvement


def better_model_exists(model, run_id=None):
    current = load_number_of_epochs_trained(model, run_id=run_id)
    best = load_number_of_epochs_trained(model, run_id=run_id,
                                         best_model=True)
    return bool(best) and best < current


def model_stopped_early(model, run_id=None):
    stopped_early, _ = model.early_stopping_status(run_id=run_id)
    return stopped_early


# ------------------------------ arguments ----------------------------------

def _parse_number_of_samples(number):
    if not isinstance(number, (int, float)) or isinstance(number, bool):
        raise TypeError("Number of samples should be an integer.")
    if number != int(number):
        raise TypeError("Number of samples should be an integer.")
    return int(number)


def parse_numbers_of_samples(proposed):
    scenarios = ["training", "evaluation"]
    if isinstance(proposed, (int, float)):
        proposed = [proposed]
    if isinstance(proposed, (list, tuple)):
        proposed = list(proposed)
        if len(proposed) == 1:
            proposed = proposed * 2
        elif len(proposed) > 2:
            raise ValueError(
                "List of number of samples can only contain one or two "
                "numbers.")
        return {s: _parse_number_of_samples(n)
                for s, n in zip(scenarios, proposed)}
    if isinstance(proposed, dict):
        try:
            return {s: _parse_number_of_samples(proposed.get(s))
                    for s in scenarios}
        except TypeError:
            raise ValueError(
                "To supply the numbers of samples as a dictionary, the "
                "dictionary must contain the keys {} with the number of "
                "samples for each given as an integer.".format(
                    enumerate_strings(["`{}`".format(s) for s in scenarios],
                                      conjunction="and")))
    raise TypeError("Expected an `int`, `list`, or `dict`; got `{}`.".format(
        type(proposed)))


def validate_model_parameters(reconstruction_distribution=None,
                              number_of_reconstruction_classes=None,
                              model_type=None, latent_distribution=None,
                              parameterise_latent_posterior=None):
    if reconstruction_distribution and number_of_reconstruction_classes:
        if number_of_reconstruction_classes > 0:
            errors = []
            if reconstruction_distribution == "bernoulli":
                errors.append("the Bernoulli distribution")
            if "zero-inflated" in reconstruction_distribution:
                errors.append("zero-inflated distributions")
            if "constrained" in reconstruction_distribution:
                errors.append("constrained distributions")
            if errors:
                raise ValueError(
                    "{} cannot be piecewise categorical.".format(
                        capitalise_string(enumerate_strings(
                            errors, conjunction="or"))))
    if model_type and latent_distribution and parameterise_latent_posterior:
        if "VAE" in model_type:
            if not (model_type in ["VAE"]
                    and latent_distribution == "gaussian mixture"):
                raise ValueError(
                    "Cannot parameterise latent posterior parameters for {} "
                    "or {} distribution.".format(
                        model_type, latent_distribution))


def evaluation_chunks(number_of_examples, minibatch_size, max_cells=0):
    """The steps of an evaluation pass as ``(start, cells, weight)``.

    The reference runs one step per minibatch and reports ``sum(minibatch
    means) / (N / B)`` (va:1092-1150, 1969-2055).  With batch normalisation in
    evaluation mode a cell's terms do not depend on its minibatch, so ``c``
    whole minibatches evaluated in one step of ``c B`` cells contribute ``c``
    times that step's mean -- the sum of their ``c`` means.  The ragged last
    minibatch keeps a step of its own (its mean is over fewer cells).
    ``max_cells`` < 2 B: one step per minibatch, as the reference."""
    n, B = int(number_of_examples), int(minibatch_size)
    per_step = max(1, int(max_cells) // B)
    chunks, start, full = [], 0, n // B
    while full > 0:
        c = min(per_step, full)
        chunks.append((start, c * B, float(c)))
        start += c * B
        full -= c
    if start < n:
        chunks.append((start, n - start, 1.0))
    return chunks


def batch_indices_for_subset(subset):
    if subset.batch_indices is None:
        raise TypeError(
            "No batch indices found in {} set.".format(subset.kind))
    return subset.batch_indices


def not_in_this_build(feature, reference):
    """Uniform error for constructor options outside the built hot path."""
    return NotImplementedError(
        "{} ({}) is not part of this MI355X build of the scVAE hot path "
        "(SURVEY.md section 8f lists it as a later row).".format(
            feature, reference))
