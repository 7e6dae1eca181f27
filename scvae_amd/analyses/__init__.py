"""The part of ``scvae.analyses`` that ``scvae evaluate`` needs to report
cluster quality: label prediction from the latent representation
(``scvae/analyses/prediction.py``) and the clustering metrics it prints.
Plots and decompositions are not part of this build."""
from scvae_amd.analyses.prediction import (  # noqa: F401
    PREDICTION_METHODS, PredictionSpecifications, predict_labels,
    map_cluster_ids_to_label_ids)
