#!/usr/bin/env python3
"""Golden vectors produced by the REFERENCE's own code (build container only).

TensorFlow 1.15 / TFP 0.7 are not importable here, so the arithmetic of the hot
path stays pinned to independent implementations only (oracle/__init__.py:
"parity unpinned").  But the reference's NumPy-only helpers around the path
*can* be executed: this script pulls their function definitions out of the
reference source files with ``ast`` (no ``import scvae``, which would import
TensorFlow), runs them where they lie under /root/reference, and writes inputs
and outputs to

    tests/golden/reference_helpers.json
    tests/golden/reference_development_data_set.npz

``tests/test_reference_fixtures.py`` compares the build's restatements with
those.  Only the vectors travel; nothing under /root/reference is read at test
time and no reference source text is stored.

Functions executed (file:line in /root/reference/scvae):
    utilities.py:36-76, 93-132     format_duration, normalise_string,
                                   capitalise_string, enumerate_strings
    data/loaders.py:942-1022       _create_development_data_set (RandomState(60))
    data/processing.py:336-486     split_data_set (RandomState(42))
    data/utilities.py:68-142       build_directory_path
    models/utilities.py:591-615    early_stopping_status
    models/utilities.py:795-850    parse_numbers_of_samples, _parse_number_of_samples
    models/utilities.py (build_training_string, build_data_string)
    models/variational_autoencoder.py:412-469                    VAE.name
    models/gaussian_mixture_variational_autoencoder.py:441-502   GMVAE.name
    distributions/utilities.py:356-389   parse_distribution (registry keys by ast)
    analyses/prediction.py:134-146       map_cluster_ids_to_label_ids
    analyses/metrics/clustering.py:145-178   accuracy, _exclude_classes_from_label_set
"""
import ast
import hashlib
import importlib.util
import io
import json
import os
import re
import sys
import time
import types
from contextlib import redirect_stdout

import numpy
import scipy.stats

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scvae"


def _tree(relative):
    with open(os.path.join(REF, relative)) as f:
        return ast.parse(f.read())


def functions(relative, names, namespace, class_name=None):
    """Compile the named function definitions of a reference file into
    ``namespace`` (top level, or methods of ``class_name``)."""
    body = _tree(relative).body
    if class_name is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef)
                    and n.name == class_name).body
    picked = [n for n in body if isinstance(n, ast.FunctionDef)
              and n.name in names]
    found = {n.name for n in picked}
    missing = set(names) - found
    if missing:
        raise KeyError("{}: {} not found".format(relative, sorted(missing)))
    module = ast.Module(body=picked, type_ignores=[])
    exec(compile(module, os.path.join(REF, relative), "exec"), namespace)
    return namespace


def dictionary_keys(relative, variable):
    """String keys of a module-level dict literal (its values may need TF),
    plus keys added afterwards by ``variable["key"] = ...`` statements."""
    keys = None
    for node in _tree(relative).body:
        if not (isinstance(node, ast.Assign) and len(node.targets) == 1):
            continue
        target = node.targets[0]
        if getattr(target, "id", None) == variable:
            keys = [k.value for k in node.value.keys]
        elif (keys is not None and isinstance(target, ast.Subscript)
              and getattr(target.value, "id", None) == variable
              and isinstance(target.slice, ast.Constant)):
            keys.append(target.slice.value)
    if keys is None:
        raise KeyError(variable)
    return keys


def distribution_parameters(relative, variable):
    """name -> {parameter name: support as source text} of the registry's dict
    literal (the order of the parameters is the order of the heads)."""
    for node in _tree(relative).body:
        if (isinstance(node, ast.Assign) and len(node.targets) == 1
                and getattr(node.targets[0], "id", None) == variable):
            result = {}
            for key, value in zip(node.value.keys, node.value.values):
                if not isinstance(value, ast.Dict):
                    continue
                entry = dict(zip((k.value for k in value.keys), value.values))
                parameters = entry.get("parameters")
                if not isinstance(parameters, ast.Dict):
                    continue
                result[key.value] = {}
                for pname, pvalue in zip(parameters.keys, parameters.values):
                    support = None
                    if isinstance(pvalue, ast.Dict):
                        fields = dict(zip((k.value for k in pvalue.keys),
                                          pvalue.values))
                        if "support" in fields:
                            support = ast.unparse(fields["support"])
                    result[key.value][pname.value] = support
            return result
    raise KeyError(variable)


def load_plain_module(relative, name):
    """Reference modules that only import the standard library."""
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(REF, relative))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def jsonable(value):
    if isinstance(value, dict):
        return {str(k): jsonable(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [jsonable(v) for v in value]
    if isinstance(value, numpy.ndarray):
        return jsonable(value.tolist())
    if isinstance(value, (numpy.integer,)):
        return int(value)
    if isinstance(value, (numpy.floating, float)):
        return None if numpy.isnan(value) else float(value)
    if isinstance(value, (numpy.bool_,)):
        return bool(value)
    return value


def main():
    out = {"reference": "scvae/scvae v2.1.4 source under /root/reference, "
                        "functions executed by tests/golden/make_reference_fixtures.py"}
    ru = load_plain_module("utilities.py", "_ref_utilities")

    # ---- strings ----------------------------------------------------------
    strings = ["Negative Binomial", "zero-inflated negative binomial",
               "10x PBMC (68k)", "a/b\\c|d?e*f", "Macosko-MRC", "k-means",
               "unit-variance gaussian", 'x<y>:"z"', "$1,000", "already_ok"]
    out["normalise_string"] = [[s, ru.normalise_string(s)] for s in strings]
    durations = [0.0004, 0.001, 0.0123, 0.9994, 1, 12.3456, 59.99, 60, 61.4,
                 119.6, 3599.6, 3600, 3661, 7199.7, 86399.9]
    out["format_duration"] = [[d, ru.format_duration(d)] for d in durations]
    caps = ["training set", "Training set", "x", "tEST set of things", "z y"]
    out["capitalise_string"] = [[s, ru.capitalise_string(s)] for s in caps]
    lists = [["a"], ["a", "b"], ["a", "b", "c"], ["`x`", "`y`", "`z`", "`w`"]]
    out["enumerate_strings"] = [
        [l, c, ru.enumerate_strings(list(l), conjunction=c)]
        for l in lists for c in ("and", "or")]

    # ---- development data set ----------------------------------------------
    ns = functions("data/loaders.py", ["_create_development_data_set"],
                   {"numpy": numpy})
    d = ns["_create_development_data_set"]()
    v = d["values"]
    numpy.savez(
        os.path.join(HERE, "reference_development_data_set.npz"),
        first_rows=v[:8], row_sums=v.sum(axis=1)[:256],
        column_sums=v.sum(axis=0), total=v.sum(),
        nonzeros=numpy.count_nonzero(v), labels_head=d["labels"][:32],
        shape=numpy.array(v.shape))
    out["development_data_set"] = {
        "values_sha256": hashlib.sha256(
            numpy.ascontiguousarray(v).tobytes()).hexdigest(),
        "labels_sha256": hashlib.sha256(
            "\n".join(d["labels"].tolist()).encode()).hexdigest(),
        "dtype": str(v.dtype),
        "feature_mapping": d["feature mapping"],
        "example_names_head": d["example names"][:3].tolist(),
        "feature_names_head": d["feature names"][:3].tolist(),
    }

    # ---- split -------------------------------------------------------------
    ns = functions(
        "data/processing.py", ["split_data_set"],
        {"numpy": numpy, "time": time.time,
         "format_duration": ru.format_duration,
         "normalise_string": ru.normalise_string,
         "defaults": {"data": {"splitting_method": "default",
                               "splitting_fraction": 0.9}}})
    split = ns["split_data_set"]
    cases = []
    for n, method, fraction in [(100, "random", 0.9), (2700, "random", 0.9),
                                (68579, None, None), (1000, "sequential", 0.8),
                                (37, "random", 0.5), (1000, "default", 0.9)]:
        data = {"values": numpy.arange(n).reshape(n, 1),
                "example names": numpy.arange(n), "feature names": None,
                "class names": None, "labels": None}
        with redirect_stdout(io.StringIO()):
            s = split(data, method=method, fraction=fraction)
        cases.append({
            "n": n, "method": method, "fraction": fraction,
            "sizes": [len(s[k]["example names"]) for k in
                      ("training set", "validation set", "test set")],
            "training_head": s["training set"]["example names"][:16],
            "validation_head": s["validation set"]["example names"][:16],
            "test_head": s["test set"]["example names"][:16],
            "sha256": hashlib.sha256(numpy.concatenate([
                s[k]["example names"] for k in
                ("training set", "validation set", "test set")
            ]).astype(numpy.int64).tobytes()).hexdigest()})
    # explicit split indices (training/test slices, validation carved out)
    n = 1000
    data = {"values": numpy.arange(n).reshape(n, 1),
            "example names": numpy.arange(n), "feature names": None,
            "class names": None, "labels": None,
            "split indices": {"training": slice(0, 900),
                              "test": slice(900, 1000)}}
    with redirect_stdout(io.StringIO()):
        s = split(data, method="default", fraction=0.9)
    cases.append({
        "n": n, "method": "indices",
        "split_indices": {"training": [0, 900], "test": [900, 1000]},
        "sizes": [len(s[k]["example names"]) for k in
                  ("training set", "validation set", "test set")],
        "training_last": int(s["training set"]["example names"][-1]),
        "validation_first": int(s["validation set"]["example names"][0])})
    out["split_data_set"] = cases

    # ---- directory path ------------------------------------------------------
    ns = functions("data/utilities.py", ["build_directory_path"],
                   {"os": os, "normalise_string": ru.normalise_string})
    build = ns["build_directory_path"]
    paths = []
    for spec in [
            dict(),
            dict(splitting_method="random", splitting_fraction=0.9),
            dict(splitting_method="default", splitting_fraction=0.9,
                 default_splitting_method="indices"),
            dict(splitting_method="default", splitting_fraction=0.81,
                 default_splitting_method="random"),
            dict(splitting_method="random", splitting_fraction=0.9,
                 preprocessing_methods=["normalise", "log"]),
            dict(splitting_method="random", splitting_fraction=0.9,
                 preprocessing_methods=["binarise"], preprocessing=False),
            dict(features_mapped=True,
                 feature_selection_method="keep variances above",
                 feature_selection_parameters=[0.5],
                 example_filter_method="keep classes",
                 example_filter_parameters=["B cells", "T cells"],
                 noisy_preprocessing_methods=["binarise"])]:
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
        paths.append([spec, build(
            "base", data_set,
            splitting_method=spec.get("splitting_method"),
            splitting_fraction=spec.get("splitting_fraction"),
            preprocessing=spec.get("preprocessing", True))])
    out["build_directory_path"] = paths

    # ---- model utilities -----------------------------------------------------
    ns = functions(
        "models/utilities.py",
        ["early_stopping_status", "parse_numbers_of_samples",
         "_parse_number_of_samples", "build_training_string",
         "build_data_string"],
        {"numpy": numpy, "enumerate_strings": ru.enumerate_strings,
         "capitalise_string": ru.capitalise_string})
    curves = [None, [], [-10.0], [-10, -9, -8], [-8, -9, -10, -11],
              [-10, -11, -9, -10, -11, -12], [-5, -6, -7, -4, -5],
              list(-numpy.arange(15.0)), [-3, -3, -3, -3]]
    out["early_stopping_status"] = [
        [c, r, jsonable(list(ns["early_stopping_status"](
            None if c is None else numpy.array(c, dtype=float), r)))]
        for c in curves for r in (1, 3, 10)]
    samples = [1, 5, 3.0, [2], [1, 10], {"training": 1, "evaluation": 25}]
    out["parse_numbers_of_samples"] = [
        [jsonable(s), jsonable(ns["parse_numbers_of_samples"](
            dict(s) if isinstance(s, dict) else
            (list(s) if isinstance(s, list) else s)))] for s in samples]
    failures = []
    for bad in ([1, 2, 3], "5", 2.5):
        try:
            ns["parse_numbers_of_samples"](bad)
            failures.append([jsonable(bad), None])
        except Exception as error:   # noqa: BLE001 - record the type
            failures.append([jsonable(bad), type(error).__name__])
    out["parse_numbers_of_samples_errors"] = failures
    out["build_training_string"] = [
        [list(a), ns["build_training_string"](*a)] for a in [
            ("model", 0, 100, "original values"),
            ("model for run r1", 5, 100, "preprocessed values"),
            ("model", 200, 500, "new preprocessed values at every epoch")]]
    data_strings = []
    for kwargs, likelihood in [
            (dict(), "negative binomial"),
            (dict(has_preprocessed_values=True), "negative binomial"),
            (dict(has_preprocessed_values=True), "bernoulli"),
            (dict(has_preprocessed_values=True,
                  preprocessing_methods=["binarise"]), "bernoulli"),
            (dict(noisy_preprocessing_methods=["binarise"]), "bernoulli"),
            (dict(noisy_preprocessing_methods=["normalise"]), "poisson")]:
        data_set = types.SimpleNamespace(
            has_preprocessed_values=kwargs.get(
                "has_preprocessed_values", False),
            preprocessing_methods=kwargs.get("preprocessing_methods", []),
            noisy_preprocessing_methods=kwargs.get(
                "noisy_preprocessing_methods", []))
        data_strings.append([kwargs, likelihood,
                             ns["build_data_string"](data_set, likelihood)])
    out["build_data_string"] = data_strings

    # ---- model names ---------------------------------------------------------
    shared = {"os": os, "normalise_string": ru.normalise_string}
    vae = functions("models/variational_autoencoder.py", ["name"],
                    dict(shared), class_name="VariationalAutoencoder")["name"]
    gmvae = functions(
        "models/gaussian_mixture_variational_autoencoder.py", ["name"],
        dict(shared),
        class_name="GaussianMixtureVariationalAutoencoder")["name"]
    base = dict(
        latent_distribution_name="gaussian", number_of_latent_clusters=1,
        parameterise_latent_posterior=False, inference_architecture="MLP",
        generative_architecture="MLP",
        reconstruction_distribution_name="negative binomial", k_max=None,
        use_count_sum_as_feature=False, latent_size=25,
        hidden_sizes=[100, 100],
        number_of_monte_carlo_samples={"training": 1, "evaluation": 1},
        number_of_importance_samples={"training": 1, "evaluation": 1},
        analytical_kl_term=True, minibatch_normalisation=True,
        batch_correction=False, dropout_parts=[], kl_weight_value=1,
        number_of_warm_up_epochs=0)
    vae_cases = []
    for change in [
            {}, {"latent_size": 100, "reconstruction_distribution_name":
                 "zero-inflated negative binomial"},
            {"latent_distribution_name": "unit-variance gaussian",
             "analytical_kl_term": False, "minibatch_normalisation": False},
            {"k_max": 3, "use_count_sum_as_feature": True,
             "batch_correction": True, "number_of_warm_up_epochs": 200,
             "kl_weight_value": 0.5, "dropout_parts": ["0.9", "0.8"],
             "number_of_monte_carlo_samples": {"training": 5},
             "number_of_importance_samples": {"training": 10},
             "inference_architecture": "LFM", "hidden_sizes": [250]},
            {"generative_architecture": "LFM",
             "parameterise_latent_posterior": True}]:
        fake = types.SimpleNamespace(type="VAE", **{**base, **change})
        vae_cases.append([change, vae.fget(fake)])
    out["vae_name"] = vae_cases
    gm_base = dict(base)
    gm_base.update(latent_distribution_name="gaussian mixture", n_clusters=20,
                   prior_probabilities_method="uniform", latent_size=100,
                   analytical_kl_term=False, number_of_warm_up_epochs=200,
                   proportion_of_free_nats_for_y_kl_divergence=0.0)
    gm_cases = []
    for change in [
            {}, {"prior_probabilities_method": "custom", "n_clusters": 9,
                 "reconstruction_distribution_name":
                     "zero-inflated negative binomial",
                 "proportion_of_free_nats_for_y_kl_divergence": 0.8},
            # what the constructor (gm:192-197) stores for
            # latent_distribution="legacy gaussian mixture"
            {"latent_distribution_name": "gaussian mixture",
             "analytical_kl_term": True,
             "number_of_warm_up_epochs": 0, "k_max": 2,
             "prior_probabilities_method": "learn", "batch_correction": True,
             "dropout_parts": ["0.5"], "kl_weight_value": 2}]:
        fake = types.SimpleNamespace(type="GMVAE", **{**gm_base, **change})
        gm_cases.append([change, gmvae.fget(fake)])
    out["gmvae_name"] = gm_cases

    # ---- distribution registry -----------------------------------------------
    registry = {
        "DISTRIBUTIONS": dictionary_keys(
            "distributions/utilities.py", "DISTRIBUTIONS"),
        "LATENT_DISTRIBUTIONS": dictionary_keys(
            "distributions/utilities.py", "LATENT_DISTRIBUTIONS"),
        "GAUSSIAN_MIXTURE_DISTRIBUTIONS": dictionary_keys(
            "distributions/utilities.py", "GAUSSIAN_MIXTURE_DISTRIBUTIONS")}
    out["distribution_names"] = registry
    parameters = distribution_parameters(
        "distributions/utilities.py", "DISTRIBUTIONS")
    out["distribution_parameters"] = {
        name: list(p) for name, p in parameters.items()}
    out["distribution_supports"] = parameters
    ns = functions(
        "distributions/utilities.py", ["parse_distribution"],
        {"normalise_string": ru.normalise_string,
         "DISTRIBUTIONS": dict.fromkeys(registry["DISTRIBUTIONS"]),
         "LATENT_DISTRIBUTIONS": dict.fromkeys(
             registry["LATENT_DISTRIBUTIONS"]),
         "GAUSSIAN_MIXTURE_DISTRIBUTIONS": dict.fromkeys(
             registry["GAUSSIAN_MIXTURE_DISTRIBUTIONS"])})
    parsed = []
    for args in [("Negative Binomial", None), ("zero_inflated_poisson", None),
                 ("ZERO-INFLATED negative-binomial", None),
                 ("constrained poisson", None), ("gaussian", "VAE"),
                 ("unit_variance_gaussian", "VAE"),
                 ("Gaussian Mixture", "GMVAE"),
                 ("legacy-gaussian-mixture", "GMVAE"),
                 ("student t", None), ("gaussian mixture", "VAE")]:
        try:
            parsed.append([list(args), ns["parse_distribution"](*args)])
        except ValueError:
            parsed.append([list(args), "ValueError"])
    out["parse_distribution"] = parsed

    # ---- cluster -> label mapping and accuracy -------------------------------
    ns = functions("analyses/prediction.py", ["map_cluster_ids_to_label_ids"],
                   {"numpy": numpy, "scipy": scipy})
    ns2 = functions("analyses/metrics/clustering.py",
                    ["accuracy", "_exclude_classes_from_label_set"],
                    {"numpy": numpy})
    rng = numpy.random.RandomState(3)
    mapping_cases = []
    for n, n_labels, n_clusters, excluded in [(50, 4, 5, []), (200, 6, 4, [0]),
                                              (30, 3, 8, [1, 2])]:
        labels = rng.randint(0, n_labels, size=n)
        clusters = rng.randint(0, n_clusters, size=n)
        predicted = ns["map_cluster_ids_to_label_ids"](
            labels, clusters, excluded)
        mapping_cases.append({
            "label_ids": labels, "cluster_ids": clusters,
            "excluded_class_ids": excluded, "predicted_label_ids": predicted,
            "accuracy": ns2["accuracy"](labels, predicted),
            "accuracy_excluding": ns2["accuracy"](
                labels.astype(str), predicted.astype(str),
                excluded_classes=[str(e) for e in excluded] or None)})
    out["cluster_label_mapping"] = mapping_cases

    with open(os.path.join(HERE, "reference_helpers.json"), "w") as f:
        json.dump(jsonable(out), f, indent=1, sort_keys=True)
    print("reference fixtures written to", HERE)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("this script needs the reference checkout at " + REF)
    main()
