"""Label prediction for an evaluation set from latent values
(``scvae/analyses/prediction.py:33-262``): cluster the latent representation
of a training set ("k-means"), or take the clusters a GMVAE assigned itself
("model"), and name every cluster after the most frequent label of its cells.
Host-side NumPy / scikit-learn work on the ``[cells, latent]`` outputs of
``model.evaluate``; nothing here runs on the GPU.
"""
from time import time

import numpy

from scvae_amd.defaults import defaults
from scvae_amd.utilities import format_duration, normalise_string

MAXIMUM_SAMPLE_SIZE_FOR_NORMAL_KMEANS = 10000
PREDICTION_METHODS = {}


def _register(name):
    def decorator(function):
        alias = normalise_string(name)
        PREDICTION_METHODS[name] = {
            "aliases": {alias, alias.replace("_", "")},
            "function": function}
        return function
    return decorator


def _method_name(method):
    """The registered name for any of its spellings (``k-means``, ``kmeans``,
    ``K means``: ``proper_string`` in the reference)."""
    key = normalise_string(str(method))
    for name, entry in PREDICTION_METHODS.items():
        if key in entry["aliases"] or key.replace("_", "") in entry["aliases"]:
            return name
    raise ValueError("Prediction method `{}` not found.".format(method))


class PredictionSpecifications:
    def __init__(self, method, number_of_clusters=None,
                 training_set_kind=None):
        self.method = _method_name(method)
        if number_of_clusters is None:
            raise TypeError("Number of clusters not set.")
        self.number_of_clusters = number_of_clusters
        if training_set_kind:
            training_set_kind = normalise_string(training_set_kind)
        self.training_set_kind = training_set_kind

    @property
    def name(self):
        parts = [self.method, self.number_of_clusters]
        if self.training_set_kind and self.training_set_kind != "training":
            parts.append(self.training_set_kind)
        return "_".join(normalise_string(str(p)).replace("_", "")
                        for p in parts)


def map_cluster_ids_to_label_ids(label_ids, cluster_ids,
                                 excluded_class_ids=()):
    """Every cluster gets the most common label id among its cells that is not
    an excluded class (ties: the smallest id, as ``scipy.stats.mode``)."""
    label_ids = numpy.asarray(label_ids)
    cluster_ids = numpy.asarray(cluster_ids)
    predicted_label_ids = numpy.zeros_like(cluster_ids)
    for cluster_id in numpy.unique(cluster_ids).tolist():
        members = cluster_ids == cluster_id
        candidates = label_ids[members]
        for excluded in excluded_class_ids:
            candidates = candidates[candidates != excluded]
        if len(candidates) == 0:
            continue
        values, counts = numpy.unique(candidates, return_counts=True)
        predicted_label_ids[members] = values[numpy.argmax(counts)]
    return predicted_label_ids


def _dense(values):
    return values.toarray() if hasattr(values, "toarray") else numpy.asarray(
        values)


@_register("k-means")
def _predict_using_kmeans(training_set, evaluation_set, number_of_clusters):
    from sklearn.cluster import KMeans, MiniBatchKMeans
    if (training_set.number_of_examples
            <= MAXIMUM_SAMPLE_SIZE_FOR_NORMAL_KMEANS):
        model = KMeans(n_clusters=number_of_clusters, random_state=None,
                       n_init=10)
    else:
        model = MiniBatchKMeans(n_clusters=number_of_clusters,
                                random_state=None, batch_size=100, n_init=3)
    model.fit(_dense(training_set.values))
    return model.predict(_dense(evaluation_set.values)), None, None


@_register("model")
def _predict_using_model(training_set, evaluation_set, number_of_clusters):
    # what the model attached in evaluate() (gm:2744-2781)
    return (evaluation_set.predicted_cluster_ids,
            evaluation_set.predicted_labels,
            getattr(evaluation_set, "predicted_superset_labels", None))


def predict_labels(training_set, evaluation_set, specifications=None,
                   method=None, number_of_clusters=None):
    if specifications is None:
        if method is None:
            method = defaults["evaluation"]["prediction_method"]
        specifications = PredictionSpecifications(
            method=method, number_of_clusters=number_of_clusters,
            training_set_kind=training_set.kind)
    predict = PREDICTION_METHODS[specifications.method]["function"]
    print("Predicting labels for evaluation set using {} with {} components."
          .format(specifications.method, specifications.number_of_clusters))
    start = time()
    cluster_ids, predicted_labels, predicted_superset_labels = predict(
        training_set=training_set, evaluation_set=evaluation_set,
        number_of_clusters=specifications.number_of_clusters)
    if (cluster_ids is not None and predicted_labels is None
            and evaluation_set.has_labels):
        to_id = evaluation_set.class_name_to_class_id
        label_ids = numpy.array([to_id[name] for name in
                                 evaluation_set.labels])
        excluded = [to_id[name] for name in evaluation_set.excluded_classes
                    if name in to_id]
        predicted_ids = map_cluster_ids_to_label_ids(
            label_ids, cluster_ids, excluded)
        to_name = evaluation_set.class_id_to_class_name
        predicted_labels = numpy.array([to_name[i] for i in predicted_ids])
    print("Labels predicted ({}).".format(format_duration(time() - start)))
    return cluster_ids, predicted_labels, predicted_superset_labels


def clustering_metrics(labels, predicted_cluster_ids, predicted_labels=None,
                       excluded_classes=()):
    """What ``scvae evaluate`` prints about a prediction
    (``analyses/metrics/clustering.py``): adjusted Rand index and adjusted
    mutual information of the clusters, accuracy of the mapped labels."""
    from sklearn.metrics import (
        adjusted_mutual_info_score, adjusted_rand_score)
    labels = numpy.asarray(labels)
    keep = numpy.ones(len(labels), dtype=bool)
    for name in excluded_classes:
        keep &= labels != name
    metrics = {
        "adjusted Rand index": float(adjusted_rand_score(
            labels[keep], numpy.asarray(predicted_cluster_ids)[keep])),
        "adjusted mutual information": float(adjusted_mutual_info_score(
            labels[keep], numpy.asarray(predicted_cluster_ids)[keep])),
    }
    if predicted_labels is not None:
        metrics["accuracy"] = float(numpy.mean(
            numpy.asarray(predicted_labels)[keep] == labels[keep]))
    return metrics
