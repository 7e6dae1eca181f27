"""Default settings of the command line and the model constructors.

The reference keeps them in one JSON document (``scvae/defaults.py:19-24``
loads ``scvae/defaults.json``); the values are part of the drop-in surface --
``scvae train DATA`` without options must mean the same run -- so every entry
below carries the reference's value.  Sections that only the plots and the
cross-analysis of the reference read are kept for the argument parser's sake.
"""

#: how a data set is found, filtered and split (``data/data_set.py``,
#: ``data/processing.py``): nothing is mapped, selected or preprocessed unless
#: asked for; the random 81 / 9 / 10 split uses fraction 0.9 twice
_DATA = dict(
    format="infer",
    directory="data",
    map_features=False,
    feature_selection=[],
    example_filter=[],
    preprocessing_methods=[],
    noisy_preprocessing_methods=[],
    split_data_set=False,
    splitting_method="default",
    splitting_fraction=0.9,
)

#: the model the reference builds when nothing is said: a VAE with one hidden
#: layer of 100 units, a 2-d gaussian latent space, a Poisson likelihood, batch
#: normalisation on; 200 epochs of minibatches of 100 cells at Adam 1e-4
_MODELS = dict(
    directory="models",
    type="VAE",
    latent_size=2,
    hidden_sizes=[100],
    number_of_samples=dict(training=1, evaluation=1),
    latent_distribution=dict(VAE="gaussian", GMVAE="gaussian mixture"),
    number_of_classes=1,
    parameterise_latent_posterior=False,
    inference_architecture="MLP",
    generative_architecture="MLP",
    reconstruction_distribution="poisson",
    number_of_reconstruction_classes=0,
    prior_probabilities_method="uniform",
    number_of_warm_up_epochs=0,
    kl_weight=1,
    proportion_of_free_nats_for_y_kl_divergence=0.0,
    minibatch_normalisation=True,
    batch_correction=False,
    dropout_keep_probabilities=[],
    count_sum=False,
    number_of_epochs=200,
    minibatch_size=100,
    learning_rate=1e-4,
    sample_size=0,
    run_id="",
    new_run=False,
    reset_training=False,
)

#: ``scvae evaluate``: the test subset, every saved model version, no label
#: prediction unless a method is named
_EVALUATION = dict(
    data_set_kind="test",
    prediction_training_set_kind="training",
    prediction_method="",
    model_versions="all",
)

#: read by the reference's analyses only (not part of this build)
_ANALYSES = dict(
    directory="analyses",
    decomposition_method="PCA",
    decomposition_dimensionality=2,
    highlight_feature_indices=[],
    included_analyses="standard",
    analysis_level="normal",
    export_options=[],
)

defaults = {
    "data": _DATA,
    "analyses": _ANALYSES,
    "models": _MODELS,
    "evaluation": _EVALUATION,
    "cross_analysis": {"log_summary": False},
}
