"""``VariationalAutoencoder``: drop-in for the class of the same name in
``scvae/models/variational_autoencoder.py:47`` (constructor arguments,
``.name``/``.description``/``.parameters``, ``.train``, ``.evaluate``,
``.log_directory``, ``.has_been_trained``, ``.early_stopping_status``), with
the TensorFlow graph replaced by HIP kernels on one or more MI355X GPUs.
"""

import copy
import os

import numpy
import torch

from scvae_amd.defaults import defaults
from scvae_amd.distributions import (
    DISTRIBUTIONS, LATENT_DISTRIBUTIONS, parse_distribution)
from scvae_amd.models import utilities as mu
from scvae_amd.models.base import ModelBase
from scvae_amd.utilities import normalise_string


class VariationalAutoencoder(ModelBase):
    """Variational autoencoder class.

    Arguments:
        feature_size (int): The number of features (genes) of the data.
        latent_size (int, optional): Dimension of the latent variable.
        hidden_sizes (list(int), optional): Units of each hidden layer of the
            encoder (the decoder uses them in reverse order).
        reconstruction_distribution (str, optional): Name of the likelihood
            (``poisson``, ``negative binomial``, ``zero-inflated poisson``,
            ``zero-inflated negative binomial``, ``constrained poisson``,
            ``bernoulli``).
        number_of_reconstruction_classes (int, optional): Piecewise
            categorical classes (``-k``): counts below k are classes of an
            extra ``P_K`` head (Poisson / negative binomial base, va:2507-2532).
        latent_distribution (str, optional): ``gaussian`` or
            ``unit-variance gaussian``.
        minibatch_normalisation (bool, optional): Batch normalisation of the
            hidden layers.
        batch_correction (bool, optional), number_of_batches (int): one-hot
            batch indices appended to the decoder input (va:2407-2441).
        number_of_warm_up_epochs (int, optional): Linear KL warm-up.
        log_directory (str, optional): Where checkpoints and scalars go.
        kwargs: ``parameterise_latent_posterior``, ``analytical_kl_term``,
            ``number_of_monte_carlo_samples``,
            ``number_of_importance_samples``, ``inference_architecture``,
            ``generative_architecture``, ``dropout_keep_probabilities``,
            ``count_sum``, ``kl_weight`` (as in va:114-292) and the build's
            own ``device`` / ``initial_seed`` / ``noise_seed``.
    """

    def __init__(self, feature_size, latent_size=None, hidden_sizes=None,
                 reconstruction_distribution=None,
                 number_of_reconstruction_classes=None,
                 latent_distribution=None, minibatch_normalisation=None,
                 batch_correction=None, number_of_batches=None,
                 number_of_warm_up_epochs=None, log_directory=None,
                 **kwargs):
        super().__init__()
        dm = defaults["models"]
        self.type = "VAE"
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
        self.latent_distribution_name = latent_distribution
        self.latent_distribution = copy.deepcopy(
            LATENT_DISTRIBUTIONS[latent_distribution])

        parameterise = kwargs.get("parameterise_latent_posterior")
        if parameterise is None:
            parameterise = dm["parameterise_latent_posterior"]
        self.parameterise_latent_posterior = parameterise

        clusters = kwargs.get("number_of_latent_clusters")
        if clusters is None:
            clusters = (dm["number_of_classes"]
                        if "mixture" in latent_distribution else 1)
        self.number_of_latent_clusters = clusters

        analytical_kl_term = kwargs.get("analytical_kl_term")
        if analytical_kl_term is None:
            analytical_kl_term = self.latent_distribution_name == "gaussian"
        self.analytical_kl_term = analytical_kl_term

        mc = kwargs.get("number_of_monte_carlo_samples")
        self.number_of_monte_carlo_samples = (
            dict(dm["number_of_samples"]) if mc is None
            else mu.parse_numbers_of_samples(mc))
        iw = kwargs.get("number_of_importance_samples")
        self.number_of_importance_samples = (
            dict(dm["number_of_samples"]) if iw is None
            else mu.parse_numbers_of_samples(iw))

        ia = kwargs.get("inference_architecture")
        self.inference_architecture = (
            dm["inference_architecture"] if ia is None else ia).upper()
        ga = kwargs.get("generative_architecture")
        self.generative_architecture = (
            dm["generative_architecture"] if ga is None else ga).upper()

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

        dropout = kwargs.get("dropout_keep_probabilities")
        if dropout is None:
            dropout = dm["dropout_keep_probabilities"]
        self.dropout_keep_probabilities = dropout
        # keep probabilities for 3 kinds of layers: [h, x, z] (va:245-269)
        self.dropout_keep_probability_z = False
        self.dropout_keep_probability_x = False
        self.dropout_keep_probability_h = False
        self.dropout_parts = []
        if isinstance(dropout, (list, tuple)):
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
            number_of_reconstruction_classes=self.k_max,
            model_type=self.type,
            latent_distribution=self.latent_distribution_name,
            parameterise_latent_posterior=self.parameterise_latent_posterior)

        # options of the reference graph that have no kernels in this build
        for architecture in (self.inference_architecture,
                             self.generative_architecture):
            if architecture not in ("MLP", "LFM"):
                raise ValueError(
                    "The architectures can only be a neural network (MLP) "
                    "or a linear factor model (LFM).")
        # parameterise_latent_posterior (va:2332-2344) only passes the
        # validation above for a mixture latent distribution, which
        # LATENT_DISTRIBUTIONS (du:309-338) does not hold.
        if self.latent_distribution_name not in (
                "gaussian", "unit-variance gaussian"):
            raise mu.not_in_this_build(
                "Latent distribution `{}`".format(
                    self.latent_distribution_name), "du:309-338")
        if self.reconstruction_distribution_name not in (
                "poisson", "negative binomial", "zero-inflated poisson",
                "zero-inflated negative binomial", "constrained poisson",
                "bernoulli"):
            raise mu.not_in_this_build(
                "Likelihood `{}`".format(
                    self.reconstruction_distribution_name), "du:30-307")

    # -- engine ----------------------------------------------------------------
    def _engine_arguments(self):
        return dict(
            feature_size=self.feature_size, latent_size=self.latent_size,
            hidden_sizes=self.hidden_sizes,
            likelihood=self.reconstruction_distribution_name,
            batch_norm=bool(self.minibatch_normalisation), model_type="VAE",
            kl_weight=self.kl_weight_value,
            decoder_extra=self.decoder_extra_size, k_max=self.k_max,
            inference_architecture=self.inference_architecture,
            generative_architecture=self.generative_architecture,
            latent_distribution=self.latent_distribution_name,
            analytical_kl_term=bool(self.analytical_kl_term),
            dropout_keep_probabilities=(
                self.dropout_keep_probability_h,
                self.dropout_keep_probability_x,
                self.dropout_keep_probability_z))

    def _parameter_shapes(self):
        table = []
        bn = self.minibatch_normalisation
        H = self.hidden_sizes

        def dense(scope, n_in, n_out, with_bn):
            table.append((scope + "/DENSE/weights", (n_in, n_out)))
            table.append((scope + "/DENSE/biases", (n_out,)))
            if with_bn:
                table.append((scope + "/BATCH_NORM/beta", (n_out,)))
        n_in = self.feature_size
        for i, h in enumerate(H if self.inference_architecture == "MLP"
                              else []):
            dense("ENCODER/{}".format(i + 1), n_in, h, bn)
            n_in = h
        dense("POSTERIOR/MU", n_in, self.latent_size, False)
        if self.latent_distribution_name != "unit-variance gaussian":
            dense("POSTERIOR/LOG_SIGMA", n_in, self.latent_size, False)
        n_in = self.latent_size + self.decoder_extra_size
        for i, h in enumerate(H[::-1] if self.generative_architecture == "MLP"
                              else []):
            dense("DECODER/{}".format(len(H) - i), n_in, h, bn)
            n_in = h
        for parameter in self.reconstruction_distribution["parameters"]:
            dense("X_TILDE/" + parameter.upper(), n_in, self.feature_size,
                  False)
        if self.k_max:
            dense("X_TILDE/P_K", n_in, self.feature_size * (self.k_max + 1),
                  False)
        return table

    # -- names -------------------------------------------------------------------
    @property
    def name(self):
        """Short name for model used in filenames (va:412-469)."""
        major_parts = [normalise_string(self.latent_distribution_name)]
        if "mixture" in self.latent_distribution_name:
            major_parts.append("c_{}".format(self.number_of_latent_clusters))
        if self.parameterise_latent_posterior:
            major_parts.append("parameterised")
        if self.inference_architecture != "MLP":
            major_parts.append("ia_{}".format(self.inference_architecture))
        if self.generative_architecture != "MLP":
            major_parts.append("ga_{}".format(self.generative_architecture))

        minor_parts = [normalise_string(self.reconstruction_distribution_name)]
        if self.k_max:
            minor_parts.append("k_{}".format(self.k_max))
        if self.use_count_sum_as_feature:
            minor_parts.append("sum")
        minor_parts.append("l_{}".format(self.latent_size))
        minor_parts.append("h_" + "_".join(map(str, self.hidden_sizes)))
        minor_parts.append(
            "mc_{}".format(self.number_of_monte_carlo_samples["training"]))
        minor_parts.append(
            "iw_{}".format(self.number_of_importance_samples["training"]))
        if self.analytical_kl_term:
            minor_parts.append("kl")
        if self.minibatch_normalisation:
            minor_parts.append("bn")
        if self.batch_correction:
            minor_parts.append("bc")
        if len(self.dropout_parts) > 0:
            minor_parts.append("dropout_" + "_".join(self.dropout_parts))
        if self.kl_weight_value != 1:
            minor_parts.append("klw_{}".format(self.kl_weight_value))
        if self.number_of_warm_up_epochs:
            minor_parts.append("wu_{}".format(self.number_of_warm_up_epochs))
        return os.path.join(self.type, "-".join(major_parts),
                            "-".join(minor_parts))

    @property
    def description(self):
        """Description of model (va:471-548)."""
        parts = ["Model setup:"]
        parts.append("type: {}".format(self.type))
        parts.append("feature size: {}".format(self.feature_size))
        parts.append("latent size: {}".format(self.latent_size))
        parts.append("hidden sizes: {}".format(
            ", ".join(map(str, self.hidden_sizes))))
        parts.append("latent distribution: " + self.latent_distribution_name)
        parts.append("reconstruction distribution: "
                     + self.reconstruction_distribution_name)
        if self.k_max > 0:
            parts.append("reconstruction classes: {} (including 0s)".format(
                self.k_max))
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
        if self.analytical_kl_term:
            parts.append("using analytical KL term")
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
    def _training_minibatch_size(self, minibatch_size, scenario):
        # va:807-811 / va:1843-1847: the VAE divides the minibatch by the
        # number of latent samples
        minibatch_size /= (self.number_of_importance_samples[scenario]
                           * self.number_of_monte_carlo_samples[scenario])
        return int(numpy.ceil(minibatch_size))

    def _sample_prior(self, count, seed, stream_id):
        """z ~ N(0, I) (``self.p_z.sample``, va:2393-2394)."""
        from scvae_amd.minibatch import philox_normal
        z = torch.empty(count, self.latent_size, device=self.engine.device)
        philox_normal(z, row_offset=0, seed=seed, stream_id=stream_id)
        return z, {}

    def _prior_summary(self):
        # gaussian prior N(0, 1): one "cluster"; the reference logs the
        # *standard deviation* under the variance tag (va:2391)
        return ([1.0], [numpy.float32(0.0)], [numpy.float32(1.0)])
