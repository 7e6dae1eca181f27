"""``GaussianMixtureVariationalAutoencoder``: drop-in for the class of the same
name in ``scvae/models/gaussian_mixture_variational_autoencoder.py:51``.

y is marginalised with K passes through the shared-weight q(z|x,y) encoder and
decoder; on the GPU the K passes run as one grouped batch
(``scvae_amd/csrc/plan_gmvae.hip``).  Training / evaluation loops, scalar tags
(``losses/kl_divergence_z``, ``losses/kl_divergence_y``, ``accuracy``) and the
cluster accuracy computed from ``argmax q(y|x)`` each epoch (gm:1299-1332)
follow the reference.
"""

import copy
import os

import numpy
import scipy.stats
import torch

from scvae_amd.defaults import defaults
from scvae_amd.distributions import (
    DISTRIBUTIONS, GAUSSIAN_MIXTURE_DISTRIBUTIONS, parse_distribution)
from scvae_amd.models import utilities as mu
from scvae_amd.models.base import ModelBase
from scvae_amd.utilities import normalise_string


def map_cluster_ids_to_label_ids(label_ids, cluster_ids,
                                 excluded_class_ids=()):
    """Majority vote of the labels inside each cluster
    (``scvae/analyses/prediction.py:134-146``)."""
    predicted_label_ids = numpy.zeros_like(cluster_ids)
    for cluster_id in numpy.unique(cluster_ids).tolist():
        indices = cluster_ids == cluster_id
        votes = label_ids[indices]
        for excluded in excluded_class_ids:
            votes = votes[votes != excluded]
        if len(votes) == 0:
            continue
        predicted_label_ids[indices] = numpy.ravel(
            scipy.stats.mode(votes, keepdims=False)[0])[0]
    return predicted_label_ids


def accuracy(labels, predicted_labels, excluded_classes=None):
    """``scvae/analyses/metrics/clustering.py:145-148``."""
    if excluded_classes:
        keep = numpy.ones(len(labels), dtype=bool)
        for excluded in excluded_classes:
            keep &= labels != excluded
        labels, predicted_labels = labels[keep], predicted_labels[keep]
    return numpy.mean(predicted_labels == labels)


class GaussianMixtureVariationalAutoencoder(ModelBase):
    """Gaussian-mixture variational autoencoder class.

    Arguments follow gm:136-326: ``feature_size``, ``latent_size``,
    ``hidden_sizes``, ``reconstruction_distribution``,
    ``number_of_reconstruction_classes``, ``latent_distribution``,
    ``prior_probabilities_method`` (``uniform``, ``custom``, ``learn``),
    ``prior_probabilities``, ``number_of_latent_clusters``,
    ``minibatch_normalisation``, ``batch_correction``, ``number_of_batches``,
    ``number_of_warm_up_epochs``, ``log_directory`` and the keyword arguments
    ``number_of_monte_carlo_samples``, ``number_of_importance_samples``,
    ``proportion_of_free_nats_for_y_kl_divergence``,
    ``dropout_keep_probabilities``, ``count_sum``, ``kl_weight``.
    """

    def __init__(self, feature_size, latent_size=None, hidden_sizes=None,
                 reconstruction_distribution=None,
                 number_of_reconstruction_classes=None,
                 latent_distribution=None, prior_probabilities_method=None,
                 prior_probabilities=None, number_of_latent_clusters=None,
                 minibatch_normalisation=None, batch_correction=None,
                 number_of_batches=None, number_of_warm_up_epochs=None,
                 log_directory=None, **kwargs):
        super().__init__()
        dm = defaults["models"]
        self.type = "GMVAE"
        self.feature_size = feature_size
        self.latent_size = dm["latent_size"] if latent_size is None \
            else latent_size
        self.hidden_sizes = list(
            dm["hidden_sizes"] if hidden_sizes is None else hidden_sizes)

        if reconstruction_distribution is None:
            reconstruction_distribution = dm["reconstruction_distribution"]
        reconstruction_distribution = parse_distribution(
            reconstruction_distribution)
        self.reconstruction_distribution_name = reconstruction_distribution
        self.reconstruction_distribution = DISTRIBUTIONS[
            reconstruction_distribution]

        if number_of_reconstruction_classes is None:
            number_of_reconstruction_classes = dm[
                "number_of_reconstruction_classes"]
        self.number_of_reconstruction_classes = (
            number_of_reconstruction_classes + 1)
        self.k_max = number_of_reconstruction_classes

        if latent_distribution is None:
            latent_distribution = dm["latent_distribution"][self.type]
        latent_distribution = parse_distribution(
            latent_distribution, model_type=self.type)
        self.latent_distribution = copy.deepcopy(
            GAUSSIAN_MIXTURE_DISTRIBUTIONS[latent_distribution])
        analytical_kl_term = False
        if latent_distribution == "legacy gaussian mixture":
            latent_distribution = "gaussian mixture"
            analytical_kl_term = True
        self.latent_distribution_name = latent_distribution
        self.analytical_kl_term = analytical_kl_term
        # variable scope of the z layers: the upper-cased distribution name
        # (gm:2962-2963), MODIFIED_GAUSSIAN for the legacy mixture
        self._z_scope = normalise_string(
            self.latent_distribution["z posterior"]).upper()

        if number_of_latent_clusters is None:
            number_of_latent_clusters = dm["number_of_classes"]
        self.n_clusters = number_of_latent_clusters

        if prior_probabilities_method is None:
            prior_probabilities_method = dm["prior_probabilities_method"]
        if prior_probabilities_method in ["uniform", "learn"]:
            prior_probabilities = None
        elif prior_probabilities_method == "custom":
            if prior_probabilities is None:
                raise TypeError("No custom prior probabilities")
            elif isinstance(prior_probabilities, dict):
                prior_probabilities = list(prior_probabilities.values())
        else:
            raise NotImplementedError(
                "`{}` method for setting prior probabilities not implemented."
                .format(prior_probabilities_method))
        self.prior_probabilities_method = prior_probabilities_method
        self.prior_probabilities = prior_probabilities
        if (self.prior_probabilities
                and len(self.prior_probabilities) != self.n_clusters):
            raise ValueError(
                "The number of provided prior probabilities has to be the "
                "same as the number of latent clusters.")

        mc = kwargs.get("number_of_monte_carlo_samples")
        self.number_of_monte_carlo_samples = (
            dict(dm["number_of_samples"]) if mc is None
            else mu.parse_numbers_of_samples(mc))
        iw = kwargs.get("number_of_importance_samples")
        self.number_of_importance_samples = (
            dict(dm["number_of_samples"]) if iw is None
            else mu.parse_numbers_of_samples(iw))

        if minibatch_normalisation is None:
            minibatch_normalisation = dm["minibatch_normalisation"]
        self.minibatch_normalisation = minibatch_normalisation

        if batch_correction is None:
            batch_correction = dm["batch_correction"]
        self.batch_correction = batch_correction
        if self.batch_correction and number_of_batches is None:
            raise TypeError(
                "The number of batches for batch correction was not "
                "provided.")
        self.number_of_batches = number_of_batches

        free_nats = kwargs.get("proportion_of_free_nats_for_y_kl_divergence")
        if free_nats is None:
            free_nats = dm["proportion_of_free_nats_for_y_kl_divergence"]
        self.proportion_of_free_nats_for_y_kl_divergence = free_nats

        dropout = kwargs.get("dropout_keep_probabilities")
        if dropout is None:
            dropout = dm["dropout_keep_probabilities"]
        self.dropout_keep_probabilities = dropout
        # keep probabilities for 4 kinds of layers: [h, x, z, y] (gm:281-305)
        self.dropout_keep_probability_y = False
        self.dropout_keep_probability_z = False
        self.dropout_keep_probability_x = False
        self.dropout_keep_probability_h = False
        self.dropout_parts = []
        if isinstance(dropout, (list, tuple)):
            if len(dropout) >= 4:
                self.dropout_keep_probability_y = dropout[3]
            if len(dropout) >= 3:
                self.dropout_keep_probability_z = dropout[2]
            if len(dropout) >= 2:
                self.dropout_keep_probability_x = dropout[1]
            if len(dropout) >= 1:
                self.dropout_keep_probability_h = dropout[0]
            self.dropout_parts = [str(p) for p in dropout if p and p != 1]
        else:
            self.dropout_keep_probability_h = dropout
            if dropout and dropout != 1:
                self.dropout_parts = [str(dropout)]

        count_sum = kwargs.get("count_sum")
        if count_sum is None:
            count_sum = dm["count_sum"]
        self.use_count_sum_as_feature = count_sum
        self.use_count_sum_as_parameter = (
            "constrained" in self.reconstruction_distribution_name
            or "multinomial" in self.reconstruction_distribution_name)

        kl_weight = kwargs.get("kl_weight")
        self.kl_weight_value = dm["kl_weight"] if kl_weight is None \
            else kl_weight

        if number_of_warm_up_epochs is None:
            number_of_warm_up_epochs = dm["number_of_warm_up_epochs"]
        self.number_of_warm_up_epochs = number_of_warm_up_epochs

        if log_directory is None:
            log_directory = dm["directory"]
        self.base_log_directory = log_directory

        self.early_stopping_rounds = 10
        self.stopped_early = None

        self._device = kwargs.get("device")
        self.initial_seed = kwargs.get("initial_seed", 0)
        self.noise_seed = kwargs.get("noise_seed", 1)

        mu.validate_model_parameters(
            reconstruction_distribution=self.reconstruction_distribution_name,
            number_of_reconstruction_classes=self.k_max)

        if self.latent_distribution_name != "gaussian mixture":
            raise mu.not_in_this_build(
                "Latent distribution `{}`".format(
                    self.latent_distribution_name), "du:340-353")
        if not self.hidden_sizes:
            raise ValueError("The GMVAE needs at least one hidden layer.")
        if self.reconstruction_distribution_name not in (
                "poisson", "negative binomial", "zero-inflated poisson",
                "zero-inflated negative binomial", "constrained poisson",
                "bernoulli"):
            raise mu.not_in_this_build(
                "Likelihood `{}`".format(
                    self.reconstruction_distribution_name), "du:30-307")

    @property
    def number_of_latent_clusters(self):
        return self.n_clusters

    # -- engine ----------------------------------------------------------------
    def _engine_arguments(self):
        return dict(
            feature_size=self.feature_size, latent_size=self.latent_size,
            hidden_sizes=self.hidden_sizes,
            likelihood=self.reconstruction_distribution_name,
            batch_norm=bool(self.minibatch_normalisation), model_type="GMVAE",
            n_clusters=self.n_clusters, kl_weight=self.kl_weight_value,
            free_nats_proportion=(
                self.proportion_of_free_nats_for_y_kl_divergence),
            decoder_extra=self.decoder_extra_size, k_max=self.k_max,
            prior_probabilities_method=self.prior_probabilities_method,
            prior_probabilities=self.prior_probabilities,
            latent_distribution=(
                "legacy gaussian mixture"
                if self._z_scope == "MODIFIED_GAUSSIAN"
                else "gaussian mixture"),
            dropout_keep_probabilities=(
                self.dropout_keep_probability_h,
                self.dropout_keep_probability_x,
                self.dropout_keep_probability_z,
                self.dropout_keep_probability_y))

    def _parameter_shapes(self):
        table = []
        bn = self.minibatch_normalisation
        H, K, L, F = (self.hidden_sizes, self.n_clusters, self.latent_size,
                      self.feature_size)

        def dense(scope, n_in, n_out, with_bn):
            table.append((scope + "/DENSE/weights", (n_in, n_out)))
            table.append((scope + "/DENSE/biases", (n_out,)))
            if with_bn:
                table.append((scope + "/BATCH_NORM/beta", (n_out,)))
        if self.prior_probabilities_method == "learn":
            table.append(("Y/P/LOGITS", (K,)))
        n_in = F
        for i, h in enumerate(H):
            dense("Y/CATEGORICAL/ENCODER/LAYER_{}".format(i + 1), n_in, h, bn)
            n_in = h
        dense("Y/CATEGORICAL/LOGITS", n_in, K, False)
        n_in = F + K
        for i, h in enumerate(H):
            dense("Z/Q/ENCODER/LAYER_{}".format(i + 1), n_in, h, bn)
            n_in = h
        dense("Z/Q/" + self._z_scope + "/MEAN", n_in, L, False)
        dense("Z/Q/" + self._z_scope + "/SOFTPLUS_SCALE", n_in, L, False)
        dense("Z/P/" + self._z_scope + "/MEAN", K, L, False)
        dense("Z/P/" + self._z_scope + "/SOFTPLUS_SCALE", K, L, False)
        n_in = L + self.decoder_extra_size
        for i, h in enumerate(H[::-1]):
            dense("X/DECODER/LAYER_{}".format(i + 1), n_in, h, bn)
            n_in = h
        for parameter in self.reconstruction_distribution["parameters"]:
            dense("X/DISTRIBUTION/" + parameter.upper(), n_in, F, False)
        if self.k_max:
            dense("X/DISTRIBUTION/P_K", n_in, F * (self.k_max + 1), False)
        return table

    # -- names -------------------------------------------------------------------
    @property
    def name(self):
        """Short name for model used in filenames (gm:441-502)."""
        latent_parts = [normalise_string(self.latent_distribution_name)]
        if "mixture" in self.latent_distribution_name:
            latent_parts.append("c_{}".format(self.n_clusters))
        if self.prior_probabilities_method != "uniform":
            latent_parts.append("p_" + self.prior_probabilities_method)

        parts = [normalise_string(self.reconstruction_distribution_name)]
        if self.k_max:
            parts.append("k_{}".format(self.k_max))
        if self.use_count_sum_as_feature:
            parts.append("sum")
        parts.append("l_{}".format(self.latent_size))
        parts.append("h_" + "_".join(map(str, self.hidden_sizes)))
        parts.append(
            "mc_{}".format(self.number_of_monte_carlo_samples["training"]))
        parts.append(
            "iw_{}".format(self.number_of_importance_samples["training"]))
        if self.analytical_kl_term:
            parts.append("kl")
        if self.minibatch_normalisation:
            parts.append("bn")
        if self.batch_correction:
            parts.append("bc")
        if len(self.dropout_parts) > 0:
            parts.append("dropout_" + "_".join(self.dropout_parts))
        if self.kl_weight_value != 1:
            parts.append("klw_{}".format(self.kl_weight_value))
        if self.number_of_warm_up_epochs:
            parts.append("wu_{}".format(self.number_of_warm_up_epochs))
        if self.proportion_of_free_nats_for_y_kl_divergence:
            parts.append("fn_{}".format(
                self.proportion_of_free_nats_for_y_kl_divergence))
        return os.path.join(self.type, "-".join(latent_parts),
                            "-".join(parts))

    @property
    def description(self):
        """Description of model (gm:504-590)."""
        parts = ["Model setup:"]
        parts.append("type: {}".format(self.type))
        parts.append("feature size: {}".format(self.feature_size))
        parts.append("latent size: {}".format(self.latent_size))
        parts.append("hidden sizes: {}".format(
            ", ".join(map(str, self.hidden_sizes))))
        parts.append("latent distribution: " + self.latent_distribution_name)
        if "mixture" in self.latent_distribution_name:
            parts.append("latent clusters: {}".format(self.n_clusters))
            parts.append("prior probabilities: "
                         + self.prior_probabilities_method)
        parts.append("reconstruction distribution: "
                     + self.reconstruction_distribution_name)
        for label, numbers in (
                ("Monte Carlo samples", self.number_of_monte_carlo_samples),
                ("importance samples", self.number_of_importance_samples)):
            text = "{}: {}".format(label, numbers["training"])
            if numbers["evaluation"] != numbers["training"]:
                text += " (training), {} (evaluation)".format(
                    numbers["evaluation"])
            parts.append(text)
        if self.kl_weight_value != 1:
            parts.append("KL weigth: {}".format(self.kl_weight_value))
        if self.proportion_of_free_nats_for_y_kl_divergence:
            parts.append("free nats for y KL divergence: {}".format(
                self.proportion_of_free_nats_for_y_kl_divergence))
        if self.minibatch_normalisation:
            parts.append("using batch normalisation for minibatches")
        if self.number_of_warm_up_epochs:
            parts.append(
                "using linear warm-up weighting for the first {} epochs"
                .format(self.number_of_warm_up_epochs))
        if self.early_stopping_rounds:
            parts.append(
                "early stopping: after {} epoch with no improvements"
                .format(self.early_stopping_rounds))
        return "\n    ".join(parts)

    # -- loop hooks ---------------------------------------------------------------
    def _eps_shape(self, samples, cells):
        return (self.n_clusters, samples, cells, self.latent_size)

    def _loss_tags(self):
        return [(0, "lower_bound", "ELBO"),
                (2, "reconstruction_error", "ENRE"),
                (3, "kl_divergence_z", "KL_z"),
                (4, "kl_divergence_y", "KL_y")]

    def _sample_prior(self, count, seed, stream_id):
        """y ~ p(y) (one-hot), z = z_y with z_k ~ p(z|y=k) (gm:2816-2826,
        2901-2910): only the component selected by y reaches p_x_mean, so only
        that one is drawn and decoded."""
        from scvae_amd.minibatch import philox_normal
        device = self.engine.device
        probabilities, means, variances = self._prior_summary()
        generator = torch.Generator(device=device).manual_seed(
            (int(seed) << 20) ^ int(stream_id))
        y = torch.multinomial(
            torch.as_tensor(probabilities, dtype=torch.float32,
                            device=device),
            count, replacement=True, generator=generator)
        eps = torch.empty(count, self.latent_size, device=device)
        philox_normal(eps, row_offset=0, seed=seed, stream_id=stream_id)
        mean = torch.as_tensor(means, dtype=torch.float32, device=device)[y]
        std = torch.as_tensor(variances, dtype=torch.float32,
                              device=device)[y].sqrt()
        z = mean + std * eps
        y_one_hot = torch.nn.functional.one_hot(
            y, self.n_clusters).to(torch.int32)
        return z, {"y": y_one_hot}

    def _latent_feature_name(self, key, index):
        return "{} variable {}".format(key, index + 1)

    def _prior_summary(self):
        """p(y) probabilities and p(z|y) means / variances (gm:2879-2882)."""
        engine = self.engine
        K, L = self.n_clusters, self.latent_size
        Wm = engine.parameter("Z/P/" + self._z_scope + "/MEAN/DENSE/weights")
        bm = engine.parameter("Z/P/" + self._z_scope + "/MEAN/DENSE/biases")
        Ws = engine.parameter(
            "Z/P/" + self._z_scope + "/SOFTPLUS_SCALE/DENSE/weights")
        bs = engine.parameter(
            "Z/P/" + self._z_scope + "/SOFTPLUS_SCALE/DENSE/biases")
        means = (Wm + bm).cpu().numpy()
        variances = torch.nn.functional.softplus(Ws + bs).cpu().numpy()
        del L
        prior_logits = engine.prior_logits
        if prior_logits is None:
            probabilities = numpy.full(K, 1.0 / K)
        else:   # p_y_probabilities = softmax(p_y_logits), gm:2815-2816
            probabilities = torch.softmax(
                prior_logits.double(), dim=0).cpu().numpy()
        return (probabilities, means, variances)

    def _centroids(self, prior):
        probabilities, means, variances = prior
        K, L = self.n_clusters, self.latent_size
        covariances = numpy.zeros((K, L, L))
        for k in range(K):
            covariances[k] = numpy.diag(variances[k])
        return {"prior": {"probabilities": numpy.array(probabilities),
                          "means": numpy.stack(means),
                          "covariance_matrices": covariances}}

    def _allocate_evaluation_outputs(self, n, n_batches, device):
        return {
            "q_y_logits": torch.zeros(n, self.n_clusters, device=device),
            "cluster_stats": torch.zeros(
                n_batches, 4, self.n_clusters, self.latent_size,
                device=device),
        }

    def _evaluation_step_outputs(self, extra, outputs, i, j, cells):
        out = super()._evaluation_step_outputs(extra, outputs, i, j, cells)
        out["q_y_logits"] = extra["q_y_logits"][i:i + cells]
        out["cluster_stats"] = extra["cluster_stats"][j]
        return out

    def _weight_evaluation_outputs(self, extra, weights):
        extra["cluster_stats"] *= weights[:, None, None, None]

    def _finish_evaluation(self, result, extra, data_set, denominator):
        result["kl_divergence"] = (result["kl_divergence_z"]
                                   + result["kl_divergence_y"])
        # gm:3401: kl_divergence_neurons is the single total KL
        result["kl_divergence_neurons"] = numpy.array(
            [result["kl_divergence"]])
        logits = extra["q_y_logits"].cpu().numpy()
        result["q_y_logits"] = logits
        stats = extra["cluster_stats"].sum(dim=0).cpu().numpy() / denominator
        result["q_z_means"], result["q_z_variances"] = stats[2], stats[3]
        shifted = logits - logits.max(axis=1, keepdims=True)
        y = numpy.exp(shifted)
        y /= y.sum(axis=1, keepdims=True)
        result["y_mean"] = y
        result["q_y_probabilities"] = y.mean(axis=0)
        cluster_ids = logits.argmax(axis=1)
        result["cluster_ids"] = cluster_ids
        result["accuracy"] = None
        if data_set.has_labels:
            label_ids = numpy.array([
                data_set.class_name_to_class_id[label]
                for label in data_set.labels])
            excluded = [data_set.class_name_to_class_id[c]
                        for c in (data_set.excluded_classes or [])
                        if c in data_set.class_name_to_class_id]
            predicted = map_cluster_ids_to_label_ids(
                label_ids, cluster_ids, excluded)
            result["accuracy"] = float(accuracy(label_ids, predicted,
                                                excluded))
            result["predicted_labels"] = numpy.array([
                data_set.class_id_to_class_name[i] for i in predicted])

    def _attach_predictions(self, output_sets, evaluation, evaluation_set):
        """gm:2744-2781: the argmax cluster of q(y|x) as "model" prediction on
        every returned version of the evaluation set."""
        from scvae_amd.analyses.prediction import PredictionSpecifications
        specifications = PredictionSpecifications(
            method="model", number_of_clusters=self.n_clusters,
            training_set_kind=None)
        for output_set in output_sets:
            members = (list(output_set.values())
                       if isinstance(output_set, dict) else [output_set])
            for member in members:
                if member is None:
                    continue
                member.update_predictions(
                    prediction_specifications=specifications,
                    predicted_cluster_ids=evaluation.get("cluster_ids"),
                    predicted_labels=evaluation.get("predicted_labels"))

    def _print_extra(self, say, evaluation, data_set):
        if evaluation.get("accuracy") is not None:
            say("        Accuracy: {:6.2f} %.".format(
                100 * evaluation["accuracy"]))

    def _extra_summary(self, scalars, evaluation):
        if evaluation.get("accuracy") is not None:
            scalars["accuracy"] = evaluation["accuracy"]
        q_y = evaluation.get("q_y_probabilities")
        if q_y is not None:
            for k in range(self.n_clusters):
                scalars["posterior/cluster_{}/probability".format(k)] = q_y[k]

    def _latent_evaluation_sets(self, evaluation, wrap, latent_names):
        cluster_names = numpy.array([
            "cluster {}".format(k + 1) for k in range(self.n_clusters)])
        return {
            "z": wrap(evaluation["latent_values"], "z",
                      feature_names=latent_names),
            "y": wrap(evaluation["y_mean"], "y",
                      feature_names=cluster_names),
        }
