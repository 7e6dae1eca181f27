"""CPU tests of the host-side mirror of the reference interface: names,
directories, defaults, argument parsing, the scalar store, checkpoints, data
containers, CLI parser."""
import os

import numpy as np
import pytest

from scvae_amd.defaults import defaults
from scvae_amd.utilities import format_duration, normalise_string


def test_normalise_string():
    assert normalise_string("Negative Binomial") == "negative_binomial"
    assert normalise_string("zero-inflated negative binomial") == \
        "zero_inflated_negative_binomial"
    assert normalise_string("a/b (c), d") == "a_b_c_d"
    assert normalise_string("GMVAE") == "gmvae"


def test_format_duration():
    assert format_duration(0.0001) == "<1 ms"
    assert format_duration(0.5) == "500 ms"
    assert format_duration(12.345) == "12.3 s"
    assert format_duration(125) == "2m 5s"
    assert format_duration(3725) == "1h 2m 5s"


def test_defaults_are_the_reference_values():
    m = defaults["models"]
    assert m["latent_size"] == 2 and m["hidden_sizes"] == [100]
    assert m["reconstruction_distribution"] == "poisson"
    assert m["minibatch_normalisation"] is True
    assert m["minibatch_size"] == 100 and m["learning_rate"] == 1e-4
    assert m["number_of_epochs"] == 200
    assert defaults["data"]["splitting_fraction"] == 0.9


def test_model_names_and_descriptions():
    from scvae_amd.models import (
        GaussianMixtureVariationalAutoencoder, VariationalAutoencoder)
    vae = VariationalAutoencoder(
        32738, latent_size=25, hidden_sizes=[100, 100],
        reconstruction_distribution="negative_binomial")
    assert vae.type == "VAE"
    assert vae.name == os.path.join(
        "VAE", "gaussian", "negative_binomial-l_25-h_100_100-mc_1-iw_1-kl-bn")
    assert "latent size: 25" in vae.description
    assert "X_TILDE/LOG_R/DENSE/weights" in vae.parameters
    gm = GaussianMixtureVariationalAutoencoder(
        32738, latent_size=100, hidden_sizes=[100, 100],
        reconstruction_distribution="negative binomial",
        number_of_latent_clusters=20, number_of_warm_up_epochs=200)
    assert gm.name == os.path.join(
        "GMVAE", "gaussian_mixture-c_20",
        "negative_binomial-l_100-h_100_100-mc_1-iw_1-bn-wu_200")
    assert gm.number_of_latent_clusters == 20
    plain = VariationalAutoencoder(
        100, minibatch_normalisation=False, kl_weight=0.5,
        number_of_importance_samples=[5, 10])
    assert plain.name.endswith("poisson-l_2-h_100-mc_1-iw_5-kl-klw_0.5")
    assert plain.number_of_importance_samples == {
        "training": 5, "evaluation": 10}
    assert vae.log_directory(base="m", run_id="a1", best_model=True) == \
        os.path.join("m", vae.name, "run_a1", "best")
    with pytest.raises(ValueError):
        vae.log_directory(early_stopping=True, best_model=True)
    with pytest.raises(ValueError):
        vae.log_directory(run_id="not valid!")


def test_constructor_errors():
    from scvae_amd.models import (
        GaussianMixtureVariationalAutoencoder, VariationalAutoencoder)
    with pytest.raises(ValueError, match="not supported"):
        VariationalAutoencoder(10, reconstruction_distribution="lomax")
    with pytest.raises(TypeError, match="number of batches"):
        VariationalAutoencoder(10, batch_correction=True)
    with pytest.raises(ValueError, match="piecewise categorical"):
        VariationalAutoencoder(
            10, reconstruction_distribution="zero-inflated poisson",
            number_of_reconstruction_classes=3)
    # built: the piecewise categorical likelihood, batch correction, count sums
    model = VariationalAutoencoder(10, number_of_reconstruction_classes=3,
                                   batch_correction=True, number_of_batches=2,
                                   count_sum=True)
    assert model.k_max == 3 and model.decoder_extra_size == 3
    # built: dropout, the sampled KL term, the unit-variance posterior
    model = VariationalAutoencoder(10, dropout_keep_probabilities=[0.9, 1, 0.5])
    assert (model.dropout_keep_probability_h, model.dropout_keep_probability_x,
            model.dropout_keep_probability_z) == (0.9, 1, 0.5)
    assert model.name.endswith("dropout_0.9_0.5")
    assert model._engine_arguments()["dropout_keep_probabilities"] == (
        0.9, 1, 0.5)
    model = VariationalAutoencoder(
        10, latent_distribution="unit-variance gaussian")
    assert model.analytical_kl_term is False
    assert not any("LOG_SIGMA" in name
                   for name, _ in model._parameter_shapes())
    model = GaussianMixtureVariationalAutoencoder(
        10, dropout_keep_probabilities=[0.9, 0.8, 0.7, 0.6])
    assert model.dropout_keep_probability_y == 0.6
    learned = GaussianMixtureVariationalAutoencoder(
        10, prior_probabilities_method="learn")
    assert ("Y/P/LOGITS", (learned.n_clusters,)) == learned._parameter_shapes()[0]
    assert "p_learn" in learned.name
    with pytest.raises(NotImplementedError):
        GaussianMixtureVariationalAutoencoder(
            10, prior_probabilities_method="infer")
    with pytest.raises(TypeError):
        GaussianMixtureVariationalAutoencoder(
            10, prior_probabilities_method="custom")


def test_parse_numbers_of_samples():
    from scvae_amd.models.utilities import parse_numbers_of_samples
    assert parse_numbers_of_samples(3) == {"training": 3, "evaluation": 3}
    assert parse_numbers_of_samples([2, 7]) == {"training": 2,
                                                "evaluation": 7}
    assert parse_numbers_of_samples({"training": 1, "evaluation": 4}) == {
        "training": 1, "evaluation": 4}
    with pytest.raises(ValueError):
        parse_numbers_of_samples([1, 2, 3])
    with pytest.raises(TypeError):
        parse_numbers_of_samples("3")


def test_early_stopping_status():
    from scvae_amd.models.utilities import early_stopping_status
    assert early_stopping_status(None, 10) == (False, 0)
    assert early_stopping_status([1, 2, 3, 2.5, 2.4], 10) == (False, 2)
    assert early_stopping_status([1, 2, 1.5, 3], 10) == (False, 0)
    stopped, n = early_stopping_status([5, 4, 3, 2], 3)
    assert stopped and np.isnan(n)


def test_scalar_store_and_checkpoints(tmp_path):
    import torch
    from scvae_amd.models import VariationalAutoencoder
    from scvae_amd.models import utilities as mu
    model = VariationalAutoencoder(10, log_directory=str(tmp_path))
    log_directory = model.log_directory()
    assert mu.get_checkpoint_state(log_directory) is None
    assert not model.has_been_trained()
    writer = mu.ScalarWriter(os.path.join(log_directory, "training"))
    valid = mu.ScalarWriter(os.path.join(log_directory, "validation"))
    for epoch, lb in enumerate([-10.0, -8.0, -9.0], start=1):
        writer.add_summary({"losses/lower_bound": lb,
                            "losses/reconstruction_error": lb + 1,
                            "losses/kl_divergence": 1.0,
                            "kl_divergence_neurons/0": 0.25,
                            "kl_divergence_neurons/1": 0.75,
                            "prior/cluster_0/probability": 1.0,
                            "prior/cluster_0/mean/dimension_0": 0.0,
                            "prior/cluster_0/mean/dimension_1": 0.0,
                            "prior/cluster_0/variance/dimension_0": 1.0,
                            "prior/cluster_0/variance/dimension_1": 1.0},
                           global_step=epoch)
        valid.add_summary({"losses/lower_bound": lb - 1}, global_step=epoch)
    curves = mu.load_learning_curves(model)
    assert np.allclose(curves["training"]["lower_bound"], [-10, -8, -9])
    assert np.allclose(curves["validation"]["lower_bound"], [-11, -9, -10])
    assert mu.load_number_of_epochs_trained(model) == 3
    assert mu.load_kl_divergences(model).shape == (3, 2)
    centroids = mu.load_centroids(model, data_set_kinds="training")
    assert centroids["prior"]["means"].shape == (3, 1, 2)
    state = {"params": torch.arange(4.0), "adam_t": 3}
    path = mu.save_checkpoint(state, log_directory, 3)
    assert mu.get_checkpoint_state(log_directory) == path
    assert mu.checkpoint_epoch(path) == 3 and model.has_been_trained()
    mu.save_checkpoint(state, log_directory, 4)   # max_to_keep=1
    kept = [f for f in os.listdir(log_directory) if f.startswith("model.ckpt")]
    assert kept == ["model.ckpt-4.pt"]
    best = model.log_directory(best_model=True)
    mu.copy_model_directory(mu.get_checkpoint_state(log_directory), best)
    assert mu.checkpoint_epoch(mu.get_checkpoint_state(best)) == 4
    assert os.path.exists(os.path.join(best, "training", "scalars.jsonl"))
    assert torch.equal(mu.load_checkpoint(
        mu.get_checkpoint_state(best))["params"], state["params"])
    assert not mu.better_model_exists(model)


def test_checkpoint_queue_keeps_the_sequential_meaning(tmp_path):
    """``CheckpointWriter``: saves, copies into ``early_stopping/`` / ``best/``
    and prunes run on a background thread in the order they were queued, so a
    copy queued BEFORE an epoch's save gets the previous epoch's checkpoint and
    one queued AFTER it this epoch's -- as in the reference's sequential loop
    (va:1385-1441, 1470-1492); the logs travel as they were when the copy was
    queued; the state file of a copy is a hard link."""
    import torch
    from scvae_amd.models import utilities as mu
    log = str(tmp_path / "log")
    early = str(tmp_path / "log" / "early_stopping")
    best = str(tmp_path / "log" / "best")
    scalars = mu.ScalarWriter(os.path.join(log, "training"))
    writer = mu.CheckpointWriter()

    def state(epoch):
        return {"params": torch.full((1000,), float(epoch)), "adam_t": epoch}
    scalars.add_summary({"lower_bound": -3.0}, 1)
    writer.save(state(1), log, 1)
    writer.copy_latest(log, best, prune=True)          # epoch 1 is the best so far
    scalars.add_summary({"lower_bound": -4.0}, 2)      # (worse: early stopping starts)
    writer.copy_latest(log, early)                     # "previous epoch's parameters"
    writer.save(state(2), log, 2)
    scalars.add_summary({"lower_bound": -2.0}, 3)
    writer.save(state(3), log, 3)
    writer.copy_latest(log, best, prune=True)
    scalars.add_summary({"lower_bound": -9.0}, 4)      # (after the copy was queued)
    writer.close()
    assert mu.checkpoint_epoch(mu.get_checkpoint_state(log)) == 3
    assert mu.checkpoint_epoch(mu.get_checkpoint_state(early)) == 1
    assert mu.checkpoint_epoch(mu.get_checkpoint_state(best)) == 3
    loaded = mu.load_checkpoint(mu.get_checkpoint_state(best))
    assert float(loaded["params"][0]) == 3.0 and loaded["adam_t"] == 3
    # one state file per directory (Saver(max_to_keep=1)), the copy a hard link
    for d in (log, early, best):
        assert len([f for f in os.listdir(d) if f.endswith(".pt")]) == 1
    assert os.stat(mu.get_checkpoint_state(best)).st_ino == os.stat(
        mu.get_checkpoint_state(log)).st_ino
    # the logs of a copy: as they were when it was queued
    steps = lambda d: [r["step"] for r in mu._read_scalars(os.path.join(d, "training"))]
    assert steps(early) == [1, 2] and steps(best) == [1, 2, 3] and steps(log) == [1, 2, 3, 4]
    # a state still on its way to the host (Engine.state_dict(non_blocking=True)) is waited
    # for on the worker and written as a plain dictionary (weights-only loadable)
    from scvae_amd.engine import PendingState

    class Travelling(PendingState):
        waited = 0

        def wait(self):
            Travelling.waited += 1
            return super().wait()
    pending = Travelling(state(7))
    writer = mu.CheckpointWriter()
    writer.save(pending, str(tmp_path / "p"), 7)
    writer.close()
    assert Travelling.waited == 1
    loaded = mu.load_checkpoint(mu.get_checkpoint_state(str(tmp_path / "p")))
    assert type(loaded) is dict and float(loaded["params"][0]) == 7.0 and loaded["adam_t"] == 7
    # a failure on the worker surfaces at the next wait
    writer = mu.CheckpointWriter()
    writer.save({"bad": lambda: None}, str(tmp_path / "x"), 1)   # (not picklable)
    with pytest.raises(Exception):
        writer.wait()
    writer.close()


def test_data_set_and_split():
    from scvae_amd.data import DataSet, SparseRowMatrix
    data_set = DataSet("synthetic_1k")
    data_set.load()
    assert data_set.number_of_examples == 1000
    assert data_set.number_of_features == 100
    assert isinstance(data_set.values, SparseRowMatrix)
    assert data_set.has_labels and not data_set.has_preprocessed_values
    training, validation, test = data_set.split()
    assert (training.number_of_examples, validation.number_of_examples,
            test.number_of_examples) == (810, 90, 100)
    assert (training.kind, validation.kind, test.kind) == (
        "training", "validation", "test")
    permutation = np.random.RandomState(42).permutation(1000)
    assert np.array_equal(data_set.split_indices["training"],
                          permutation[:810])
    assert np.array_equal(data_set.split_indices["test"], permutation[900:])
    dense = data_set.values[permutation[:3]].toarray()
    assert np.array_equal(training.values[:3].toarray(), dense)
    assert np.allclose(data_set.count_sum[:, 0],
                       np.asarray(data_set.values.sum(axis=1)).ravel())
    with pytest.raises(FileNotFoundError):
        DataSet("no_such_data_set").load()


def test_directory_layout_and_cli_parser(tmp_path):
    from scvae_amd import cli
    from scvae_amd.data import DataSet
    from scvae_amd.data.utilities import build_directory_path
    d = DataSet("10x PBMC 68k")
    assert build_directory_path("models", d) == os.path.join(
        "models", "10x_pbmc_68k", "no_split", "no_preprocessing")
    assert build_directory_path("models", d, "random", 0.9) == os.path.join(
        "models", "10x_pbmc_68k", "split-random_0.9", "no_preprocessing")
    with pytest.raises(SystemExit):
        cli.main(["train"])          # data set argument is required
    with pytest.raises(SystemExit):
        cli.main(["analyse", "x"])   # not part of this build
    with pytest.raises(ValueError, match="Model type not found"):
        cli._setup_model(DataSet("synthetic_1k", values=np.ones((4, 3))),
                         model_type="AE")


def test_cluster_accuracy_helpers():
    from scvae_amd.models.gaussian_mixture_variational_autoencoder import (
        accuracy, map_cluster_ids_to_label_ids)
    labels = np.array([0, 0, 1, 1, 2, 2, 2])
    clusters = np.array([5, 5, 5, 3, 3, 3, 3])
    predicted = map_cluster_ids_to_label_ids(labels, clusters)
    assert np.array_equal(predicted, [0, 0, 0, 2, 2, 2, 2])
    assert accuracy(labels, predicted) == pytest.approx(5 / 7)
    predicted = map_cluster_ids_to_label_ids(labels, clusters, [2])
    assert np.array_equal(predicted, [0, 0, 0, 1, 1, 1, 1])
    assert accuracy(labels, predicted, [2]) == pytest.approx(3 / 4)


def test_file_loaders(tmp_path):
    """10x matrix.mtx tarball and tab-separated matrices (loaders.py:651-721,
    391-404) through DataSet / the 81-9-10 split."""
    import gzip
    import io
    import tarfile

    import scipy.io
    import scipy.sparse
    from scvae_amd.data import DataSet
    rng = np.random.default_rng(0)
    genes, cells = 7, 30
    counts = scipy.sparse.random(genes, cells, density=0.3, random_state=1,
                                 data_rvs=lambda n: rng.integers(1, 9, n))
    counts = counts.tocoo().astype(np.int64)
    tar_path = tmp_path / "pbmc_tiny.tar.gz"
    with tarfile.open(tar_path, "w:gz") as tarball:
        def add(name, payload):
            info = tarfile.TarInfo("filtered_matrices/hg19/" + name)
            info.size = len(payload)
            tarball.addfile(info, io.BytesIO(payload))
        buffer = io.BytesIO()
        scipy.io.mmwrite(buffer, counts)
        add("matrix.mtx", buffer.getvalue())
        add("barcodes.tsv", "\n".join(
            "CELL{}-1".format(i) for i in range(cells)).encode())
        add("genes.tsv", "\n".join(
            "ENSG{}\tG{}".format(j, j) for j in range(genes)).encode())
    data = DataSet(str(tar_path))
    assert data.name == "pbmc_tiny"
    data.load()
    assert data.data_format == "10x"
    assert data.values.shape == (cells, genes)
    assert np.array_equal(np.asarray(data.values.todense()),
                          counts.toarray().T)
    assert data.example_names[3] == "CELL3-1"
    assert data.feature_names[2] == "ENSG2\tG2"
    training, validation, test = DataSet(str(tar_path)).split()
    assert (training.number_of_examples + validation.number_of_examples
            + test.number_of_examples) == cells

    dense = counts.toarray().T
    tsv_path = tmp_path / "matrix.tsv.gz"
    with gzip.open(tsv_path, "wt") as handle:
        handle.write("cell\t" + "\t".join(
            "g{}".format(j) for j in range(genes)) + "\n")
        for i in range(cells):
            handle.write("c{}\t".format(i) + "\t".join(
                str(v) for v in dense[i]) + "\n")
    data = DataSet(str(tsv_path), data_format="matrix_ebf")
    data.load()
    assert np.array_equal(np.asarray(data.values.todense()), dense)
    assert data.feature_names[1] == "g1" and data.example_names[4] == "c4"
    flipped = DataSet(str(tsv_path), data_format="matrix_fbe")
    flipped.load()
    assert flipped.values.shape == (genes, cells)
    with pytest.raises(FileNotFoundError):
        DataSet("no_such_data_set").load()


def test_label_prediction_from_latent_values():
    """scvae/analyses/prediction.py: k-means on the latent training values,
    clusters named after their most frequent label."""
    from scvae_amd.analyses.prediction import (
        PREDICTION_METHODS, PredictionSpecifications, clustering_metrics,
        map_cluster_ids_to_label_ids, predict_labels)
    from scvae_amd.data import DataSet
    assert set(PREDICTION_METHODS) == {"k-means", "model"}
    assert PredictionSpecifications("kmeans", 3).method == "k-means"
    assert PredictionSpecifications("K means", 3, "Validation").name == (
        "kmeans_3_validation")
    assert PredictionSpecifications("k-means", 3, "training").name == (
        "kmeans_3")
    with pytest.raises(ValueError):
        PredictionSpecifications("spectral", 3)
    with pytest.raises(TypeError):
        PredictionSpecifications("k-means")
    # majority vote with an excluded class and a tie (smallest id wins)
    labels = np.array([0, 0, 1, 2, 2, 2, 1, 1])
    clusters = np.array([5, 5, 5, 7, 7, 7, 9, 9])
    assert map_cluster_ids_to_label_ids(labels, clusters).tolist() == [
        0, 0, 0, 2, 2, 2, 1, 1]
    assert map_cluster_ids_to_label_ids(labels, clusters, [0]).tolist() == [
        1, 1, 1, 2, 2, 2, 1, 1]
    assert map_cluster_ids_to_label_ids(
        np.array([3, 4]), np.array([0, 0])).tolist() == [3, 3]
    # three well separated blobs in a 2-d "latent space"
    rng = np.random.default_rng(0)
    centres = np.array([[0, 0], [10, 0], [0, 10]], dtype=np.float32)
    ids = rng.integers(0, 3, size=300)
    z = centres[ids] + rng.normal(0, 0.5, size=(300, 2)).astype(np.float32)
    names = np.array(["a", "b", "c"])[ids]

    def latent(rows, kind):
        return DataSet("toy", values=z[rows], labels=names[rows],
                       example_names=np.arange(len(rows)).astype(str),
                       feature_names=np.array(["z1", "z2"]), kind=kind,
                       version="z")
    training, test = latent(np.arange(200), "training"), latent(
        np.arange(200, 300), "test")
    cluster_ids, predicted, superset = predict_labels(
        training, test, method="k-means", number_of_clusters=3)
    assert superset is None and len(np.unique(cluster_ids)) == 3
    assert (predicted == test.labels).all()
    metrics = clustering_metrics(test.labels, cluster_ids, predicted)
    assert metrics["accuracy"] == 1.0
    assert metrics["adjusted Rand index"] == pytest.approx(1.0)
    test.update_predictions(predicted_cluster_ids=cluster_ids,
                            predicted_labels=predicted)
    assert test.has_predictions and test.number_of_predicted_classes == 3
    # "model": whatever the model attached to the evaluation set
    again = predict_labels(training, test, method="model",
                           number_of_clusters=3)
    assert (again[0] == cluster_ids).all() and (again[1] == predicted).all()
    test.reset_predictions()
    assert not test.has_predictions


def test_legacy_gaussian_mixture_scopes():
    """du:349-352 / gm:192-197: the legacy mixture is the same graph under the
    scope MODIFIED_GAUSSIAN, named "gaussian mixture" with the `kl` tag."""
    from scvae_amd.models import GaussianMixtureVariationalAutoencoder
    legacy = GaussianMixtureVariationalAutoencoder(
        10, latent_distribution="legacy gaussian mixture")
    modern = GaussianMixtureVariationalAutoencoder(10)
    assert legacy.latent_distribution_name == "gaussian mixture"
    assert legacy.analytical_kl_term and not modern.analytical_kl_term
    names = [n for n, _ in legacy._parameter_shapes()]
    assert "Z/Q/MODIFIED_GAUSSIAN/MEAN/DENSE/weights" in names
    assert not any("SOFTPLUS_GAUSSIAN" in n for n in names)
    assert [n.replace("MODIFIED", "SOFTPLUS") for n in names] == [
        n for n, _ in modern._parameter_shapes()]
    assert legacy._engine_arguments()["latent_distribution"] == (
        "legacy gaussian mixture")
    assert "-kl" in legacy.name and "-kl" not in modern.name


def test_preprocessing_methods():
    """data/processing.py:305-333, 496-513: log / exp / normalise / binarise,
    sparse stays sparse, applied once when the data set is loaded."""
    import scipy.sparse as sp
    from scvae_amd.data import DataSet
    from scvae_amd.data.processing import build_preprocessor
    rng = np.random.default_rng(0)
    dense = (rng.poisson(1.0, size=(30, 12)) * (rng.random((30, 12)) < 0.5)
             ).astype(np.float32)
    sparse = sp.csr_matrix(dense)
    for methods, want in (
            (["log"], np.log1p(dense)),
            (["exp"], np.expm1(dense)),
            (["binarise"], (dense > 0.5).astype(np.float32)),
            (["log", "binarise"], (np.log1p(dense) > 0.5).astype(np.float32)),
            ([], dense)):
        f = build_preprocessor(methods)
        assert np.allclose(f(dense), want, rtol=1e-6)
        got = f(sparse)
        assert sp.issparse(got) and np.allclose(got.toarray(), want, rtol=1e-6)
    norms = np.sqrt((dense ** 2).sum(axis=0))
    want = dense / np.where(norms > 0, norms, 1)
    assert np.allclose(build_preprocessor(["normalise"])(dense), want, rtol=1e-5)
    assert np.allclose(build_preprocessor(["normalise"])(sparse).toarray(),
                       want, rtol=1e-5)
    with pytest.raises(ValueError, match="not found"):
        build_preprocessor(["square"])
    # noisy: "binarise" is a Bernoulli draw per value (processing.py:311-312, 516-522)
    probabilities = np.clip(dense / dense.max(), 0, 1).astype(np.float32)
    np.random.seed(3)
    want = np.random.binomial(1, probabilities.astype(np.float64)).astype(np.float32)
    np.random.seed(3)
    drawn = build_preprocessor(["binarise"], noisy=True)(probabilities)
    assert np.array_equal(drawn, want) and set(np.unique(drawn)) <= {0.0, 1.0}
    np.random.seed(4)
    sparse_draw = build_preprocessor(["binarise"], noisy=True)(sp.csr_matrix(probabilities))
    assert sp.issparse(sparse_draw) and sparse_draw.nnz == int(sparse_draw.sum())
    assert ((sparse_draw.toarray() > 0) <= (probabilities > 0)).all()
    with pytest.raises(ValueError, match=r"\[0, 1\]"):
        build_preprocessor(["binarise"], noisy=True)(dense + 2)
    noisy_set = DataSet("toy", values=sp.csr_matrix(probabilities),
                        noisy_preprocessing_methods=["binarise"],
                        example_names=np.arange(30).astype(str),
                        feature_names=np.arange(12).astype(str))
    assert noisy_set.noisy_preprocess is not None
    assert DataSet("toy", values=sparse, example_names=np.arange(30).astype(str),
                   feature_names=np.arange(12).astype(str)).noisy_preprocess is None
    data = DataSet("toy", values=sparse, preprocessing_methods=["log"],
                   example_names=np.arange(30).astype(str),
                   feature_names=np.arange(12).astype(str))
    data.preprocess()
    assert np.allclose(data.preprocessed_values.toarray(), np.log1p(dense))
    assert np.array_equal(data.values.toarray(), dense)   # the target stays
    training, validation, test = data.split()
    assert training.has_preprocessed_values
    assert np.allclose(
        training.preprocessed_values.toarray(),
        np.log1p(training.values.toarray()))
    data.binarise()
    assert set(np.unique(data.binarised_values.toarray())) <= {0.0, 1.0}


def test_minibatch_request_checks_and_fills_the_side_work_struct():
    """``DeviceCSR.request`` (the fetch a step carries for the next one): buffer
    checks on the host, and the ``scvae_side_work`` fields it fills -- no GPU
    needed (a stand-in for the device matrix, CPU tensors)."""
    import torch
    from scvae_amd import _lib
    from scvae_amd.minibatch import MinibatchRequest

    class Matrix:
        shape = (50, 12)
        integer_counts = True
        indptr = torch.zeros(51, dtype=torch.int64)
        indices = torch.zeros(4, dtype=torch.int32)
        values = torch.zeros(4)
        row_lgamma1p = torch.zeros(50)

    rows = torch.arange(8)
    out = torch.zeros(8, 16, dtype=torch.uint16)
    rc = torch.zeros(8)
    request = MinibatchRequest(Matrix, rows, out, rc)
    side = _lib.SideWork()
    request.fill(side)
    assert side.fetch_as_u16 == 1 and side.fetch_n == 8
    assert side.fetch_features == 12 and side.fetch_ld == 16
    assert side.fetch_out == out.data_ptr() and side.fetch_rows == rows.data_ptr()
    assert side.fetch_row_values_out == rc.data_ptr()
    assert side.adam_m is None and side.noise_out is None
    fp32 = MinibatchRequest(Matrix, rows, torch.zeros(8, 12))
    fp32.fill(side)
    assert side.fetch_as_u16 == 0 and side.fetch_ld == 12
    assert side.fetch_row_values_out is None
    with pytest.raises(ValueError):      # too few rows in the buffer
        MinibatchRequest(Matrix, rows, torch.zeros(4, 12))
    with pytest.raises(ValueError):      # neither uint16 nor fp32
        MinibatchRequest(Matrix, rows, torch.zeros(8, 12, dtype=torch.float64))
    Matrix.integer_counts = False
    with pytest.raises(ValueError):      # uint16 needs an integer count matrix
        MinibatchRequest(Matrix, rows, out)


def test_evaluation_chunks_keep_the_reference_average():
    """``evaluation_chunks``: contiguous steps over all cells, every step but a
    ragged last one a whole number of minibatches weighted by that number, so
    that sum(weight x step mean) is the reference's sum of minibatch means."""
    from scvae_amd.models.utilities import evaluation_chunks
    for n, B, limit in [(61722, 100, 4096), (6857, 100, 4096), (250, 100, 4096),
                        (99, 100, 4096), (300, 100, 150), (4096, 4096, 4096),
                        (5000, 4096, 4096), (1000, 7, 50), (0, 100, 4096),
                        (250, 100, 0)]:
        chunks = evaluation_chunks(n, B, limit)
        position = 0
        for index, (start, cells, weight) in enumerate(chunks):
            assert start == position and cells > 0
            position += cells
            if cells % B == 0:
                assert weight == cells // B
                assert cells <= max(limit, B)
            else:
                assert index == len(chunks) - 1 and cells < B and weight == 1
        assert position == n
        assert sum(w for _, _, w in chunks) == -(-n // B)
    assert evaluation_chunks(250, 100, 0) == [(0, 100, 1.0), (100, 100, 1.0),
                                              (200, 50, 1.0)]
    # a mean of c B cells times c is the sum of the c minibatch means
    import numpy
    values = numpy.random.default_rng(0).normal(size=61722)
    reference = sum(values[i:i + 100].mean() for i in range(0, 61722, 100))
    chunked = sum(w * values[s:s + c].mean()
                  for s, c, w in evaluation_chunks(61722, 100, 4096))
    assert abs(reference - chunked) < 1e-9 * abs(reference)


def test_checkpoint_writer_failure_is_sticky_and_never_hangs(tmp_path, monkeypatch):
    """A failed job (disk full, ...) must surface at the very next call from the
    loop and must not strand the queue: jobs queued behind it are skipped, a
    skipped save gives its slot back (it used to stay counted, and the next
    save() then waited for ever), later calls raise the same error."""
    import threading
    from scvae_amd.models import utilities as mu
    gate = threading.Event()

    def failing(state, log_directory, epoch):
        gate.wait(10)
        raise OSError("No space left on device")
    monkeypatch.setattr(mu, "save_checkpoint", failing)
    writer = mu.CheckpointWriter()
    log = str(tmp_path / "log")
    writer.save({"a": 1}, log, 1)          # fails once the gate opens
    writer.save({"a": 2}, log, 2)          # queued behind it: skipped
    writer.copy_latest(log, str(tmp_path / "best"))
    gate.set()
    with pytest.raises(OSError):
        writer.wait()
    assert writer._pending_saves == 0
    finished = []

    def later():
        try:
            writer.save({"a": 3}, log, 3)
        except OSError:
            finished.append("raised")
    worker = threading.Thread(target=later, daemon=True)
    worker.start()
    worker.join(10)
    assert finished == ["raised"]          # (not hanging, not silently accepted)
    with pytest.raises(OSError):
        writer.copy_latest(log, str(tmp_path / "best"))
    writer.close()                         # (the error has been delivered: a quiet end)
    unseen = mu.CheckpointWriter()
    unseen.save({"a": 1}, log, 1)
    with pytest.raises(OSError):           # nobody waited: close() is the last chance
        unseen.close()


def test_snapshot_logs_takes_every_small_file_but_the_checkpoints(tmp_path):
    from scvae_amd.models import utilities as mu
    source = tmp_path / "log"
    (source / "training").mkdir(parents=True)
    (source / "training" / "scalars.jsonl").write_text("{}\n")
    (source / "run.log").write_text("x")
    (source / "notes.txt").write_text("kept as before")
    (source / "model.ckpt-3.pt").write_bytes(b"state")
    (source / mu.CHECKPOINT_INDEX).write_text("{}")
    files = mu.snapshot_logs(str(source))
    assert set(files) == {"run.log", "notes.txt",
                          os.path.join("training", "scalars.jsonl")}
