"""The build's host-side helpers against vectors produced by the REFERENCE's own
code (``tests/golden/make_reference_fixtures.py`` executes the NumPy-only
reference functions in the build container; only their inputs / outputs are
stored).  This pins the boundary helpers -- names, directory layout, splits,
the ``development`` data set, early stopping, label mapping -- to the
reference.  It does not pin the TensorFlow arithmetic of the hot path, which
cannot be executed here (oracle/__init__.py: parity unpinned).
"""
import hashlib
import json
import os
import types

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(GOLDEN, "reference_helpers.json")) as f:
        return json.load(f)


def test_string_helpers(ref):
    from scvae_amd import utilities as u
    for s, want in ref["normalise_string"]:
        assert u.normalise_string(s) == want, s
    for seconds, want in ref["format_duration"]:
        assert u.format_duration(seconds) == want, seconds
    for s, want in ref["capitalise_string"]:
        assert u.capitalise_string(s) == want, s
    for strings, conjunction, want in ref["enumerate_strings"]:
        assert u.enumerate_strings(list(strings), conjunction) == want


def test_development_data_set_is_the_references(ref):
    """scvae/data/loaders.py:942-1022 (RandomState(60)): the whole 10 000 x 25
    matrix and every label, by digest, plus readable slices."""
    from scvae_amd.data.synthetic import create_development_data_set
    d = create_development_data_set()
    v = d["values"]
    want = np.load(os.path.join(GOLDEN, "reference_development_data_set.npz"))
    assert tuple(want["shape"]) == v.shape
    assert str(v.dtype) == ref["development_data_set"]["dtype"]
    assert np.array_equal(v[:8], want["first_rows"])
    assert np.array_equal(v.sum(axis=1)[:256], want["row_sums"])
    assert np.array_equal(v.sum(axis=0), want["column_sums"])
    assert int(np.count_nonzero(v)) == int(want["nonzeros"])
    assert list(d["labels"][:32]) == list(want["labels_head"])
    digest = hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()
    assert digest == ref["development_data_set"]["values_sha256"]
    labels = hashlib.sha256(
        "\n".join(np.asarray(d["labels"]).tolist()).encode()).hexdigest()
    assert labels == ref["development_data_set"]["labels_sha256"]
    assert d["feature mapping"] == ref["development_data_set"][
        "feature_mapping"]
    assert list(d["example names"][:3]) == ref["development_data_set"][
        "example_names_head"]
    assert list(d["feature names"][:3]) == ref["development_data_set"][
        "feature_names_head"]


def test_split_indices(ref):
    """scvae/data/processing.py:336-486: RandomState(42) permutation and the
    int(f*n) / int(f*int(f*n)) sizes, the whole index vectors by digest."""
    from scvae_amd.data import DataSet
    for case in ref["split_data_set"]:
        if case["method"] == "indices":
            continue   # no loader of this build provides explicit split indices
        n = case["n"]
        data = DataSet("split", values=np.arange(n, dtype=np.float32)
                       .reshape(n, 1),
                       example_names=np.arange(n),
                       feature_names=np.array(["g"]))
        data.split(method=case["method"], fraction=case["fraction"])
        parts = [np.asarray(data.split_indices[k])
                 for k in ("training", "validation", "test")]
        assert [len(p) for p in parts] == case["sizes"], case
        assert parts[0][:16].tolist() == case["training_head"]
        assert parts[1][:16].tolist() == case["validation_head"]
        assert parts[2][:16].tolist() == case["test_head"]
        digest = hashlib.sha256(
            np.concatenate(parts).astype(np.int64).tobytes()).hexdigest()
        assert digest == case["sha256"], case


def test_directory_layout(ref):
    from scvae_amd.data.utilities import build_directory_path
    for spec, want in ref["build_directory_path"]:
        data_set = types.SimpleNamespace(
            name="pbmc_68k",
            default_splitting_method=spec.get("default_splitting_method"),
            features_mapped=spec.get("features_mapped", False),
            feature_selection_method=spec.get("feature_selection_method"),
            feature_selection_parameters=spec.get(
                "feature_selection_parameters"),
            example_filter_method=spec.get("example_filter_method"),
            example_filter_parameters=spec.get("example_filter_parameters"),
            preprocessing_methods=spec.get("preprocessing_methods"),
            noisy_preprocessing_methods=spec.get(
                "noisy_preprocessing_methods"))
        got = build_directory_path(
            "base", data_set, splitting_method=spec.get("splitting_method"),
            splitting_fraction=spec.get("splitting_fraction"),
            preprocessing=spec.get("preprocessing", True))
        assert got == want, spec


def test_model_utilities(ref):
    from scvae_amd.models import utilities as mu
    for curve, rounds, (stopped, epochs) in ref["early_stopping_status"]:
        losses = None if curve is None else np.array(curve, dtype=float)
        got_stopped, got_epochs = mu.early_stopping_status(losses, rounds)
        assert bool(got_stopped) == stopped, (curve, rounds)
        if epochs is None:
            assert np.isnan(got_epochs), (curve, rounds)
        else:
            assert got_epochs == epochs, (curve, rounds)
    for proposed, want in ref["parse_numbers_of_samples"]:
        assert mu.parse_numbers_of_samples(proposed) == want, proposed
    for proposed, error in ref["parse_numbers_of_samples_errors"]:
        expected = {"ValueError": ValueError, "TypeError": TypeError}[error]
        with pytest.raises(expected):
            mu.parse_numbers_of_samples(proposed)
    for args, want in ref["build_training_string"]:
        assert mu.build_training_string(*args) == want, args
    for kwargs, likelihood, want in ref["build_data_string"]:
        data_set = types.SimpleNamespace(
            has_preprocessed_values=kwargs.get(
                "has_preprocessed_values", False),
            preprocessing_methods=kwargs.get("preprocessing_methods", []),
            noisy_preprocessing_methods=kwargs.get(
                "noisy_preprocessing_methods", []))
        assert mu.build_data_string(data_set, likelihood) == want, kwargs


# fake-``self`` attribute of the fixture -> constructor argument of the build
def _vae_arguments(change):
    kwargs = dict(feature_size=50, latent_size=25, hidden_sizes=[100, 100],
                  reconstruction_distribution="negative binomial",
                  latent_distribution="gaussian")
    rename = {
        "latent_size": "latent_size", "hidden_sizes": "hidden_sizes",
        "reconstruction_distribution_name": "reconstruction_distribution",
        "latent_distribution_name": "latent_distribution",
        "analytical_kl_term": "analytical_kl_term",
        "minibatch_normalisation": "minibatch_normalisation",
        "k_max": "number_of_reconstruction_classes",
        "batch_correction": "batch_correction",
        "number_of_warm_up_epochs": "number_of_warm_up_epochs",
        "kl_weight_value": "kl_weight",
        "inference_architecture": "inference_architecture",
        "generative_architecture": "generative_architecture",
        "parameterise_latent_posterior": "parameterise_latent_posterior",
        "prior_probabilities_method": "prior_probabilities_method",
        "n_clusters": "number_of_latent_clusters",
        "proportion_of_free_nats_for_y_kl_divergence":
            "proportion_of_free_nats_for_y_kl_divergence",
    }
    for key, value in change.items():
        if key == "use_count_sum_as_feature":
            kwargs["count_sum"] = value
        elif key == "dropout_parts":
            kwargs["dropout_keep_probabilities"] = [float(p) for p in value]
        elif key == "number_of_monte_carlo_samples":
            kwargs["number_of_monte_carlo_samples"] = value["training"]
        elif key == "number_of_importance_samples":
            kwargs["number_of_importance_samples"] = value["training"]
        else:
            kwargs[rename[key]] = value
    if kwargs.get("batch_correction"):
        kwargs["number_of_batches"] = 3
    return kwargs


def test_model_names_are_the_references(ref):
    """va:412-469 / gm:441-502: the directory name of a model."""
    from scvae_amd.models import (GaussianMixtureVariationalAutoencoder,
                                  VariationalAutoencoder)
    for change, want in ref["vae_name"]:
        if change.get("parameterise_latent_posterior"):
            continue   # (only meaningful for mixture latents, not built)
        model = VariationalAutoencoder(**_vae_arguments(change))
        assert model.name == want, change
    for change, want in ref["gmvae_name"]:
        kwargs = _vae_arguments(change)
        kwargs.setdefault("number_of_latent_clusters", 20)
        kwargs.setdefault("number_of_warm_up_epochs", 200)
        kwargs["latent_size"] = change.get("latent_size", 100)
        # (gm:192-197: the legacy mixture is stored as "gaussian mixture" with
        # the analytical-KL flag set)
        kwargs["latent_distribution"] = (
            "legacy gaussian mixture" if kwargs.pop("analytical_kl_term", False)
            else "gaussian mixture")
        if kwargs.get("prior_probabilities_method") == "custom":
            kwargs["prior_probabilities"] = [1.0 / 9] * 9
        model = GaussianMixtureVariationalAutoencoder(**kwargs)
        assert model.name == want, change


def test_distribution_registry(ref):
    """du:30-353 registry names and du:356-389 ``parse_distribution``."""
    from scvae_amd.distributions import utilities as du
    names = ref["distribution_names"]
    # every name the build registers is one of the reference's; the count
    # likelihoods of the hot path and the
    # latent parts are all there (Gaussian / log-normal / gamma / Lomax / EMG
    # reconstructions and the full-covariance mixture are out of scope)
    assert set(du.DISTRIBUTIONS) <= set(names["DISTRIBUTIONS"])
    assert {"poisson", "negative binomial", "zero-inflated poisson",
            "zero-inflated negative binomial", "constrained poisson",
            "bernoulli", "gaussian", "softplus gaussian", "modified gaussian",
            "categorical"} <= set(du.DISTRIBUTIONS)
    assert list(du.LATENT_DISTRIBUTIONS) == names["LATENT_DISTRIBUTIONS"]
    assert set(du.GAUSSIAN_MIXTURE_DISTRIBUTIONS) <= set(
        names["GAUSSIAN_MIXTURE_DISTRIBUTIONS"])
    for name in names["DISTRIBUTIONS"]:   # parameter names of the shared ones
        if name in du.DISTRIBUTIONS and name in ref["distribution_parameters"]:
            assert list(du.DISTRIBUTIONS[name]["parameters"]) == ref[
                "distribution_parameters"][name], name
    for (name, model_type), want in ref["parse_distribution"]:
        if want == "ValueError":
            with pytest.raises(ValueError):
                du.parse_distribution(name, model_type)
        else:
            assert du.parse_distribution(name, model_type) == want, name


def test_cluster_label_mapping_and_accuracy(ref):
    """analyses/prediction.py:134-146 (majority vote, scipy.stats.mode ties)
    and analyses/metrics/clustering.py:145-148."""
    from scvae_amd.analyses.prediction import (clustering_metrics,
                                               map_cluster_ids_to_label_ids)
    for case in ref["cluster_label_mapping"]:
        labels = np.array(case["label_ids"])
        clusters = np.array(case["cluster_ids"])
        predicted = map_cluster_ids_to_label_ids(
            labels, clusters, case["excluded_class_ids"])
        assert predicted.tolist() == case["predicted_label_ids"]
        metrics = clustering_metrics(labels, clusters, predicted)
        assert metrics["accuracy"] == pytest.approx(case["accuracy"], abs=0)
        excluded = [str(e) for e in case["excluded_class_ids"]]
        metrics = clustering_metrics(
            labels.astype(str), clusters, predicted.astype(str),
            excluded_classes=excluded)
        assert metrics["accuracy"] == pytest.approx(
            case["accuracy_excluding"], abs=1e-15)
