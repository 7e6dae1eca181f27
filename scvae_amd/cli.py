"""Command-line interface: ``scvae train`` / ``scvae evaluate`` with the flag
names and defaults of the reference (``scvae/cli.py:698-1239``); the Python
functions ``train`` and ``evaluate`` keep the reference's call order
(``cli.py:111-264, 267-566``): data set -> (split) -> model directory ->
model -> ``model.train`` / ``model.evaluate``.

The ``analyse`` / ``cross-analyse`` commands and the plotting analyses
(``scvae/analyses``) are CPU post-processing outside the built hot path; the
flags that only feed them are accepted and ignored with a notice.
"""

import argparse
import contextlib
import os

from scvae_amd import __version__
from scvae_amd.data import DataSet
from scvae_amd.data.utilities import (
    build_directory_path, indices_for_evaluation_subset)
from scvae_amd.defaults import defaults
from scvae_amd.models import (
    GaussianMixtureVariationalAutoencoder, VariationalAutoencoder)
from scvae_amd.models.utilities import (
    better_model_exists, model_stopped_early)
from scvae_amd.utilities import normalise_string, subtitle, title


def _setup_model(data_set, model_type=None, latent_size=None,
                 hidden_sizes=None, number_of_importance_samples=None,
                 number_of_monte_carlo_samples=None,
                 inference_architecture=None, latent_distribution=None,
                 number_of_classes=None, parameterise_latent_posterior=False,
                 prior_probabilities_method=None, generative_architecture=None,
                 reconstruction_distribution=None,
                 number_of_reconstruction_classes=None, count_sum=None,
                 proportion_of_free_nats_for_y_kl_divergence=None,
                 minibatch_normalisation=None, batch_correction=None,
                 dropout_keep_probabilities=None,
                 number_of_warm_up_epochs=None, kl_weight=None,
                 models_directory=None):
    """Model factory (``cli.py:601-689``)."""
    if model_type is None:
        model_type = defaults["models"]["type"]
    if batch_correction is None:
        batch_correction = defaults["models"]["batch_correction"]
    feature_size = data_set.number_of_features
    number_of_batches = data_set.number_of_batches
    if not data_set.has_batches:
        batch_correction = False

    if normalise_string(model_type) == "vae":
        return VariationalAutoencoder(
            feature_size=feature_size, latent_size=latent_size,
            hidden_sizes=hidden_sizes,
            number_of_monte_carlo_samples=number_of_monte_carlo_samples,
            number_of_importance_samples=number_of_importance_samples,
            inference_architecture=inference_architecture,
            latent_distribution=latent_distribution,
            number_of_latent_clusters=number_of_classes,
            parameterise_latent_posterior=parameterise_latent_posterior,
            generative_architecture=generative_architecture,
            reconstruction_distribution=reconstruction_distribution,
            number_of_reconstruction_classes=number_of_reconstruction_classes,
            minibatch_normalisation=minibatch_normalisation,
            batch_correction=batch_correction,
            number_of_batches=number_of_batches,
            dropout_keep_probabilities=dropout_keep_probabilities,
            count_sum=count_sum,
            number_of_warm_up_epochs=number_of_warm_up_epochs,
            kl_weight=kl_weight, log_directory=models_directory)
    if normalise_string(model_type) == "gmvae":
        prior_probabilities = None
        method = prior_probabilities_method
        if prior_probabilities_method == "infer":
            method = "custom"
            prior_probabilities = getattr(
                data_set, "class_probabilities", None)
        return GaussianMixtureVariationalAutoencoder(
            feature_size=feature_size, latent_size=latent_size,
            hidden_sizes=hidden_sizes,
            number_of_monte_carlo_samples=number_of_monte_carlo_samples,
            number_of_importance_samples=number_of_importance_samples,
            prior_probabilities_method=method,
            prior_probabilities=prior_probabilities,
            latent_distribution=latent_distribution,
            number_of_latent_clusters=number_of_classes,
            proportion_of_free_nats_for_y_kl_divergence=(
                proportion_of_free_nats_for_y_kl_divergence),
            reconstruction_distribution=reconstruction_distribution,
            number_of_reconstruction_classes=number_of_reconstruction_classes,
            minibatch_normalisation=minibatch_normalisation,
            batch_correction=batch_correction,
            number_of_batches=number_of_batches,
            dropout_keep_probabilities=dropout_keep_probabilities,
            count_sum=count_sum,
            number_of_warm_up_epochs=number_of_warm_up_epochs,
            kl_weight=kl_weight, log_directory=models_directory)
    raise ValueError("Model type not found: `{}`.".format(model_type))


def _model_arguments(arguments):
    keys = (
        "model_type", "latent_size", "hidden_sizes",
        "number_of_importance_samples", "number_of_monte_carlo_samples",
        "inference_architecture", "latent_distribution", "number_of_classes",
        "parameterise_latent_posterior", "prior_probabilities_method",
        "generative_architecture", "reconstruction_distribution",
        "number_of_reconstruction_classes", "count_sum",
        "proportion_of_free_nats_for_y_kl_divergence",
        "minibatch_normalisation", "batch_correction",
        "dropout_keep_probabilities", "number_of_warm_up_epochs", "kl_weight")
    return {k: arguments.get(k) for k in keys}


def _load_data(data_set_file_or_name, data_format, data_directory,
               preprocessing_methods, noisy_preprocessing_methods,
               split_data_set, splitting_method, splitting_fraction,
               binarise_values=False):
    data_set = DataSet(
        data_set_file_or_name, data_format=data_format,
        directory=data_directory,
        preprocessing_methods=preprocessing_methods,
        noisy_preprocessing_methods=noisy_preprocessing_methods)
    if binarise_values:   # targets of the Bernoulli likelihood (cli.py:144-160)
        data_set.load()
        data_set.binarise()
    if split_data_set:
        subsets = data_set.split(method=splitting_method,
                                 fraction=splitting_fraction)
    else:
        data_set.load()
        splitting_method = splitting_fraction = None
        subsets = (data_set, None, data_set)
    return data_set, subsets, splitting_method, splitting_fraction


def train(data_set_file_or_name, data_format=None, data_directory=None,
          preprocessing_methods=None, noisy_preprocessing_methods=None,
          split_data_set=None, splitting_method=None, splitting_fraction=None,
          number_of_epochs=None, minibatch_size=None, learning_rate=None,
          run_id=None, new_run=False, reset_training=None,
          models_directory=None, caches_directory=None,
          analyses_directory=None, deterministic=False, **keyword_arguments):
    """Train model on data set (``cli.py:111-264``).  ``deterministic`` (not in
    the reference): bit-repeatable accumulation of the decoder gradient."""
    if split_data_set is None:
        split_data_set = defaults["data"]["split_data_set"]
    if splitting_method is None:
        splitting_method = defaults["data"]["splitting_method"]
    if splitting_fraction is None:
        splitting_fraction = defaults["data"]["splitting_fraction"]
    if models_directory is None:
        models_directory = defaults["models"]["directory"]

    print(title("Data"))
    data_set, subsets, splitting_method, splitting_fraction = _load_data(
        data_set_file_or_name, data_format, data_directory,
        preprocessing_methods, noisy_preprocessing_methods, split_data_set,
        splitting_method, splitting_fraction,
        binarise_values=normalise_string(str(keyword_arguments.get(
            "reconstruction_distribution"))) == "bernoulli")
    training_set, validation_set, _ = subsets

    models_directory = build_directory_path(
        models_directory, data_set=data_set,
        splitting_method=splitting_method,
        splitting_fraction=splitting_fraction)
    model_caches_directory = None
    if caches_directory:
        model_caches_directory = build_directory_path(
            os.path.join(caches_directory, "log"), data_set=data_set,
            splitting_method=splitting_method,
            splitting_fraction=splitting_fraction)

    print(title("Model"))
    model_arguments = _model_arguments(keyword_arguments)
    if (model_arguments["number_of_classes"] is None
            and training_set.has_labels):
        model_arguments["number_of_classes"] = training_set.number_of_classes
    model = _setup_model(data_set=training_set,
                         models_directory=models_directory, **model_arguments)
    print(model.description)
    print()
    print(model.parameters)
    print()

    print(subtitle("Training"))
    if analyses_directory:
        print("Intermediate analyses (plots) are not part of this build; "
              "`--analyses-directory` is ignored.")
    model.train(
        training_set, validation_set, number_of_epochs=number_of_epochs,
        minibatch_size=minibatch_size, learning_rate=learning_rate,
        intermediate_analyser=None, run_id=run_id, new_run=new_run,
        reset_training=reset_training,
        temporary_log_directory=model_caches_directory,
        deterministic=deterministic)
    return 0


def evaluate(data_set_file_or_name, data_format=None, data_directory=None,
             preprocessing_methods=None, noisy_preprocessing_methods=None,
             split_data_set=None, splitting_method=None,
             splitting_fraction=None, minibatch_size=None, run_id=None,
             models_directory=None, evaluation_set_kind=None,
             sample_size=None, model_versions=None, prediction_method=None,
             prediction_training_set_kind=None, **keyword_arguments):
    """Evaluate model on data set (``cli.py:267-566``)."""
    if prediction_method is None:
        prediction_method = defaults["evaluation"]["prediction_method"]
    if prediction_training_set_kind is None:
        prediction_training_set_kind = defaults["evaluation"][
            "prediction_training_set_kind"]
    prediction_training_set_kind = normalise_string(
        prediction_training_set_kind)
    if sample_size is None:
        sample_size = defaults["models"]["sample_size"]
    if split_data_set is None:
        split_data_set = defaults["data"]["split_data_set"]
    if splitting_method is None:
        splitting_method = defaults["data"]["splitting_method"]
    if splitting_fraction is None:
        splitting_fraction = defaults["data"]["splitting_fraction"]
    if models_directory is None:
        models_directory = defaults["models"]["directory"]
    if evaluation_set_kind is None:
        evaluation_set_kind = defaults["evaluation"]["data_set_kind"]
    if model_versions is None:
        model_versions = defaults["evaluation"]["model_versions"]
    if not isinstance(model_versions, list):
        model_versions = [model_versions]
    evaluation_set_kind = normalise_string(evaluation_set_kind)

    print(title("Data"))
    data_set, subsets, splitting_method, splitting_fraction = _load_data(
        data_set_file_or_name, data_format, data_directory,
        preprocessing_methods, noisy_preprocessing_methods, split_data_set,
        splitting_method, splitting_fraction,
        binarise_values=normalise_string(str(keyword_arguments.get(
            "reconstruction_distribution"))) == "bernoulli")
    training_set, validation_set, test_set = subsets
    if split_data_set:
        kinds = {"training": training_set, "validation": validation_set,
                 "test": test_set, "full": data_set}
        if evaluation_set_kind not in kinds:
            raise ValueError(
                "Evaluation set kind `{}` not found.".format(
                    evaluation_set_kind))
        evaluation_set = kinds[evaluation_set_kind]
        prediction_training_set = kinds.get(prediction_training_set_kind)
        if prediction_method and prediction_training_set is None:
            raise ValueError(
                "Prediction training set kind `{}` not found.".format(
                    prediction_training_set_kind))
    else:
        evaluation_set = data_set
        prediction_training_set = data_set
    evaluation_subset_indices = indices_for_evaluation_subset(evaluation_set)

    models_directory = build_directory_path(
        models_directory, data_set=data_set,
        splitting_method=splitting_method,
        splitting_fraction=splitting_fraction)

    print(title("Model"))
    model_arguments = _model_arguments(keyword_arguments)
    if (model_arguments["number_of_classes"] is None
            and training_set.has_labels):
        model_arguments["number_of_classes"] = training_set.number_of_classes
    model = _setup_model(data_set=evaluation_set,
                         models_directory=models_directory, **model_arguments)
    if not model.has_been_trained(run_id=run_id):
        raise Exception(
            "Model not found. Either it has not been trained or "
            "scVAE is looking in the wrong directory. "
            "The models directory resulting from the data set specification "
            "is: \"{}\"".format(models_directory))
    if "all" in model_versions:
        model_versions = ["end_of_training"]
        if better_model_exists(model, run_id=run_id):
            model_versions.append("best_model")
        if model_stopped_early(model, run_id=run_id):
            model_versions.append("early_stopping")
    print(model.description)
    print()

    prediction_specifications = None
    if prediction_method:   # cli.py:450-461
        from scvae_amd.analyses.prediction import PredictionSpecifications
        number_of_clusters = model_arguments["number_of_classes"]
        if number_of_clusters is None:
            number_of_clusters = getattr(
                model, "number_of_latent_clusters", None)
        prediction_specifications = PredictionSpecifications(
            method=prediction_method, number_of_clusters=number_of_clusters,
            training_set_kind=prediction_training_set.kind)
        print("Prediction method: {}.".format(
            prediction_specifications.method))
        print("Number of clusters: {}.".format(
            prediction_specifications.number_of_clusters))
        print("Prediction training set: {} set.".format(
            prediction_specifications.training_set_kind))
        print()

    results = {}
    for model_version in model_versions:
        use_best_model = model_version == "best_model"
        use_early_stopping_model = model_version == "early_stopping"
        print(subtitle("Evaluation ({})".format(
            model_version.replace("_", " "))))
        results[model_version] = model.evaluate(
            evaluation_set=evaluation_set,
            evaluation_subset_indices=evaluation_subset_indices,
            minibatch_size=minibatch_size, run_id=run_id,
            use_best_model=use_best_model,
            use_early_stopping_model=use_early_stopping_model,
            output_versions="all")
        print()
        if sample_size:   # cli.py:494-505
            print(subtitle("Sampling ({})".format(
                model_version.replace("_", " "))))
            sample_reconstruction_set, _ = model.sample(
                sample_size=sample_size, minibatch_size=minibatch_size,
                run_id=run_id, use_best_model=use_best_model,
                use_early_stopping_model=use_early_stopping_model)
            results[model_version] = (
                results[model_version], sample_reconstruction_set)
            print()
        if prediction_specifications is not None:   # cli.py:509-543
            _predict(model, results[model_version], prediction_training_set,
                     prediction_specifications, minibatch_size, run_id,
                     use_best_model, use_early_stopping_model, model_version)
    return results


def _predict(model, version_results, prediction_training_set, specifications,
             minibatch_size, run_id, use_best_model, use_early_stopping_model,
             model_version):
    """Cluster the latent representation and attach the predicted labels to
    every version of the evaluation set (``cli.py:509-543``)."""
    from scvae_amd.analyses.prediction import (
        clustering_metrics, predict_labels)
    print(subtitle("Prediction ({})".format(model_version.replace("_", " "))))
    evaluation_results = (version_results[0]
                          if isinstance(version_results, tuple)
                          else version_results)
    transformed, reconstructed, latent_sets = evaluation_results
    latent_training_sets = model.evaluate(
        evaluation_set=prediction_training_set, minibatch_size=minibatch_size,
        run_id=run_id, use_best_model=use_best_model,
        use_early_stopping_model=use_early_stopping_model,
        output_versions="latent", log_results=False)
    print()
    cluster_ids, predicted_labels, predicted_superset_labels = predict_labels(
        training_set=latent_training_sets["z"],
        evaluation_set=latent_sets["z"], specifications=specifications)
    for version in [transformed, reconstructed] + list(latent_sets.values()):
        version.update_predictions(
            prediction_specifications=specifications,
            predicted_cluster_ids=cluster_ids,
            predicted_labels=predicted_labels)
    if cluster_ids is not None and transformed.has_labels:
        metrics = clustering_metrics(
            transformed.labels, cluster_ids, predicted_labels,
            transformed.excluded_classes)
        print("Clustering of the {} set ({}):".format(
            transformed.kind, specifications.name))
        for name, value in metrics.items():
            print("    {}: {:.4f}".format(name, value))
    print()


def _parse_default(default):
    if not isinstance(default, bool) and default != 0 and not default:
        default = None
    return default


def main(arguments=None):
    parser = argparse.ArgumentParser(
        prog="scvae", description="Model single-cell transcript counts "
        "using deep learning (MI355X build of the training/evaluation path).",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument(
        "--version", "-V", action="version",
        version="%(prog)s {}".format(__version__))
    subparsers = parser.add_subparsers(help="commands", dest="command")
    subparsers.required = True

    parser_train = subparsers.add_parser(
        name="train",
        description="Train model on single-cell transcript counts.",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser_train.set_defaults(func=train)
    parser_evaluate = subparsers.add_parser(
        name="evaluate",
        description="Evaluate model on single-cell transcript counts.",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser_evaluate.set_defaults(func=evaluate)

    dd, dm = defaults["data"], defaults["models"]
    for sub in (parser_train, parser_evaluate):
        sub.add_argument(dest="data_set_file_or_name",
                         help="data set name or path to data set file")
        sub.add_argument("--format", "-f", dest="data_format",
                         metavar="FORMAT",
                         default=_parse_default(dd["format"]),
                         help="format of the data set")
        sub.add_argument("--data-directory", "-D", metavar="DIRECTORY",
                         default=_parse_default(dd["directory"]),
                         help="directory where data are placed or copied")
        sub.add_argument("--preprocessing-methods", "-p", metavar="METHOD",
                         nargs="+",
                         default=_parse_default(dd["preprocessing_methods"]),
                         help="methods for preprocessing data (applied in "
                              "order)")
        sub.add_argument("--noisy-preprocessing-methods", "--np",
                         metavar="METHOD", nargs="+",
                         default=_parse_default(
                             dd["noisy_preprocessing_methods"]),
                         help="methods for noisily preprocessing data at "
                              "every epoch (applied in order)")
        sub.add_argument("--split-data-set", action="store_true",
                         default=_parse_default(dd["split_data_set"]),
                         help="split data set into training, validation, "
                              "and test sets")
        sub.add_argument("--splitting-method", metavar="METHOD",
                         default=_parse_default(dd["splitting_method"]),
                         help="method for splitting data into training, "
                              "validation, and test sets")
        sub.add_argument("--splitting-fraction", metavar="FRACTION",
                         type=float,
                         default=_parse_default(dd["splitting_fraction"]),
                         help="fraction to use when splitting data into "
                              "training, validation, and test sets")
        sub.add_argument("--model-type", "-m", metavar="TYPE",
                         default=_parse_default(dm["type"]),
                         help="type of model; either VAE or GMVAE")
        sub.add_argument("--latent-size", "-l", metavar="SIZE", type=int,
                         default=_parse_default(dm["latent_size"]),
                         help="size of latent space")
        sub.add_argument("--hidden-sizes", "-H", metavar="SIZE", type=int,
                         nargs="+",
                         default=_parse_default(dm["hidden_sizes"]),
                         help="sizes of hidden layers")
        sub.add_argument("--number-of-importance-samples", metavar="NUMBER",
                         type=int, nargs="+",
                         default=_parse_default(dm["number_of_samples"]),
                         help="the number of importance weighted samples "
                              "(if two numbers are given, the first will be "
                              "used for training and the second for "
                              "evaluation)")
        sub.add_argument("--number-of-monte-carlo-samples", metavar="NUMBER",
                         type=int, nargs="+",
                         default=_parse_default(dm["number_of_samples"]),
                         help="the number of Monte Carlo samples (if two "
                              "numbers are given, the first will be used for "
                              "training and the second for evaluation)")
        sub.add_argument("--inference-architecture", metavar="KIND",
                         default=_parse_default(
                             dm["inference_architecture"]),
                         help="architecture of the inference model")
        sub.add_argument("--latent-distribution", "-q",
                         metavar="DISTRIBUTION",
                         help="distribution for the latent variable(s)")
        sub.add_argument("--number-of-classes", "-K", metavar="NUMBER",
                         type=int, help="number of proposed clusters in data "
                                        "set")
        sub.add_argument("--parameterise-latent-posterior",
                         action="store_true",
                         default=_parse_default(
                             dm["parameterise_latent_posterior"]),
                         help="parameterise latent posterior parameters, if "
                              "possible")
        sub.add_argument("--generative-architecture", metavar="KIND",
                         default=_parse_default(
                             dm["generative_architecture"]),
                         help="architecture of the generative model")
        sub.add_argument("--reconstruction-distribution", "-r",
                         metavar="DISTRIBUTION",
                         default=_parse_default(
                             dm["reconstruction_distribution"]),
                         help="distribution for the reconstructions")
        sub.add_argument("--number-of-reconstruction-classes", "-k",
                         metavar="NUMBER", type=int,
                         default=_parse_default(
                             dm["number_of_reconstruction_classes"]),
                         help="the maximum count for which to use "
                              "classification")
        sub.add_argument("--prior-probabilities-method", metavar="METHOD",
                         default=_parse_default(
                             dm["prior_probabilities_method"]),
                         help="method to set prior probabilities")
        sub.add_argument("--number-of-warm-up-epochs", "-w",
                         metavar="NUMBER", type=int,
                         default=_parse_default(
                             dm["number_of_warm_up_epochs"]),
                         help="number of initial epochs with a linear "
                              "weight on the KL divergence")
        sub.add_argument("--kl-weight", metavar="WEIGHT", type=float,
                         default=_parse_default(dm["kl_weight"]),
                         help="weighting of KL divergence")
        sub.add_argument("--proportion-of-free-nats-for-y-kl-divergence",
                         metavar="PROPORTION", type=float,
                         default=_parse_default(dm[
                             "proportion_of_free_nats_for_y_kl_divergence"]),
                         help="proportion of maximum y KL divergence, which "
                              "has constant term and zero gradients, for the "
                              "GMVAE (free-bits method)")
        # as in the reference, a store_true flag whose default is True:
        # batch normalisation cannot be switched off from the command line
        sub.add_argument("--minibatch-normalisation", "-b",
                         action="store_true",
                         default=_parse_default(
                             dm["minibatch_normalisation"]),
                         help="use batch normalisation for minibatches")
        sub.add_argument("--batch-correction", "--bc", action="store_true",
                         default=_parse_default(dm["batch_correction"]),
                         help="use batch correction in models")
        sub.add_argument("--dropout-keep-probabilities", metavar="PROBABILITY",
                         type=float, nargs="+",
                         default=_parse_default(
                             dm["dropout_keep_probabilities"]),
                         help="list of probabilities, p, of keeping "
                              "connections when using dropout (interval: "
                              "]0, 1[, where p in {0, 1, False} means no "
                              "dropout)")
        sub.add_argument("--count-sum", action="store_true",
                         default=_parse_default(dm["count_sum"]),
                         help="use count sum")
        sub.add_argument("--minibatch-size", "-B", metavar="SIZE", type=int,
                         default=_parse_default(dm["minibatch_size"]),
                         help="minibatch size for stochastic optimisation "
                              "algorithm")
        sub.add_argument("--run-id", metavar="ID", type=str,
                         default=_parse_default(dm["run_id"]),
                         help="ID for separate run of the model (can only "
                              "contain alphanumeric characters)")
        sub.add_argument("--models-directory", "-M", metavar="DIRECTORY",
                         default=_parse_default(dm["directory"]),
                         help="directory where models are stored")
        sub.add_argument("--analyses-directory", "-A", metavar="DIRECTORY",
                         default=None,
                         help="directory where analyses are saved (plots are "
                              "not part of this build)")

    parser_train.add_argument(
        "--number-of-epochs", "-e", metavar="NUMBER", type=int,
        default=_parse_default(dm["number_of_epochs"]),
        help="number of epochs for which to train")
    parser_train.add_argument(
        "--learning-rate", metavar="RATE", type=float,
        default=_parse_default(dm["learning_rate"]),
        help="learning rate when training")
    parser_train.add_argument(
        "--new-run", action="store_true",
        default=_parse_default(dm["new_run"]),
        help="train a model anew as a separate run with an automatically "
             "generated ID")
    parser_train.add_argument(
        "--reset-training", action="store_true",
        default=_parse_default(dm["reset_training"]),
        help="reset already trained model")
    parser_train.add_argument(
        "--caches-directory", "-C", metavar="DIRECTORY",
        help="directory for temporary storage")
    parser_train.add_argument(
        "--deterministic", action="store_true", default=False,
        help="(this build) bit-repeatable training steps: sum the decoder "
             "gradient over the gene strips in a fixed order instead of with "
             "fp32 atomics (about 5 %% slower at large minibatches)")

    parser_evaluate.add_argument(
        "--evaluation-set-kind", metavar="KIND",
        default=_parse_default(defaults["evaluation"]["data_set_kind"]),
        help="kind of subset to evaluate and analyse")
    parser_evaluate.add_argument(
        "--sample-size", metavar="SIZE", type=int,
        default=_parse_default(defaults["models"]["sample_size"]),
        help="sample size for sampling model")
    parser_evaluate.add_argument(
        "--prediction-method", "-P", metavar="METHOD",
        default=_parse_default(defaults["evaluation"]["prediction_method"]),
        help="method for predicting labels (k-means, or model for the "
             "clusters of a GMVAE)")
    parser_evaluate.add_argument(
        "--prediction-training-set-kind", metavar="KIND",
        default=_parse_default(
            defaults["evaluation"]["prediction_training_set_kind"]),
        help="kind of subset to fit the prediction method on: training, "
             "validation, test, or full")
    parser_evaluate.add_argument(
        "--model-versions", metavar="VERSION", nargs="+",
        default=_parse_default(defaults["evaluation"]["model_versions"]),
        help="model versions to evaluate: end-of-training, best-model, "
             "early-stopping")

    parsed = parser.parse_args(arguments)
    started = _start_data_parallel()
    try:
        import torch.distributed as dist
        quiet = (dist.is_available() and dist.is_initialized()
                 and dist.get_rank() != 0)
        if quiet:   # one voice: rank 0 prints, writes logs and checkpoints
            with open(os.devnull, "w") as sink, \
                    contextlib.redirect_stdout(sink):
                status = parsed.func(**vars(parsed))
        else:
            status = parsed.func(**vars(parsed))
    finally:
        if started:
            _stop_data_parallel()
    return status


def _start_data_parallel():
    """One process per GPU: when the command was started by
    ``python -m torch.distributed.run --nproc-per-node N -m scvae_amd train ...``
    (``WORLD_SIZE`` / ``RANK`` / ``LOCAL_RANK`` / ``MASTER_*`` in the
    environment), join the process group before any model is built -- the
    model classes shard every minibatch over the group (``models/base.py``,
    ``dataparallel.py``: contiguous row shards, gradient all-reduce over RCCL,
    synchronised batch norm) -- and leave it afterwards.  The reference is a
    single process (va:887); nothing here changes a run without that
    environment.  ``SCVAE_DIST_BACKEND=gloo`` moves the bytes over the host
    (tests on a box with fewer GPUs than ranks).  Returns whether this call
    created the group."""
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    if world <= 1:
        return False
    import torch
    import torch.distributed as dist
    if not dist.is_available() or dist.is_initialized():
        return False
    backend = os.environ.get("SCVAE_DIST_BACKEND", "nccl")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    keyword_arguments = {}
    if torch.cuda.is_available():
        device = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(device)
        if backend == "nccl":
            keyword_arguments["device_id"] = device
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world,
                            **keyword_arguments)
    return True


def _stop_data_parallel():
    import torch.distributed as dist
    if dist.is_initialized():
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()
