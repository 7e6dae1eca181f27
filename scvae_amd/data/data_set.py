"""Minimal ``DataSet`` container: the attributes and methods of
``scvae/data/data_set.py:50`` that the model classes and the CLI consume
(SURVEY.md section 8b).  Sources: local files (``loaders.py``: 10x ``.h5`` and
``.tar.gz``, sparse ``.h5``, text matrices) with the reference's ``.sparse.h5``
cache of a parsed file (``internal_io.py``), and built-in synthetic count
matrices (``synthetic.py``); downloads and plotting metadata are not provided.
"""

import os
from time import time

import numpy
import scipy.sparse

from scvae_amd.data.processing import build_preprocessor
from scvae_amd.data.sparse import SparseRowMatrix
from scvae_amd.defaults import defaults
from scvae_amd.utilities import normalise_string

# (data_set.py:31-35)
PREPROCESS_SUFFIX = "preprocessed"
PREPROCESSED_EXTENSION = ".sparse.h5"
MINIMUM_NUMBER_OF_SECONDS_BEFORE_SAVING = 30


class DataSet:
    """Cell x gene count matrix with its metadata."""

    def __init__(self, input_file_or_name, data_format=None, title=None,
                 specifications=None, values=None, labels=None,
                 example_names=None, feature_names=None, batch_indices=None,
                 batch_names=None, feature_selection=None,
                 example_filter=None, preprocessing_methods=None,
                 preprocessed_values=None, binarised_values=None,
                 noisy_preprocessing_methods=None,
                 total_standard_deviations=None,
                 explained_standard_deviations=None,
                 features_mapped=False, kind="full", version="original",
                 directory=None, **kwargs):
        # a path to a local file, or the name of a built-in synthetic data set
        self.path = None
        if isinstance(input_file_or_name, str) and os.path.isfile(
                input_file_or_name):
            from scvae_amd.data.loaders import data_set_name_from_path
            self.path = input_file_or_name
            input_file_or_name = data_set_name_from_path(self.path)
        self.name = normalise_string(str(input_file_or_name))
        self.title = title if title is not None else str(input_file_or_name)
        self.data_format = data_format
        self.specifications = specifications or {}
        self.directory = directory or defaults["data"]["directory"]
        self.features_mapped = features_mapped
        self.feature_selection = feature_selection or []
        self.example_filter = example_filter or []
        self.preprocessing_methods = preprocessing_methods or []
        # (data_set.py:362-375: a preprocessor applied anew every epoch of
        #  the training loop, va:960-976)
        self.noisy_preprocessing_methods = noisy_preprocessing_methods or []
        self.noisy_preprocess = (
            build_preprocessor(self.noisy_preprocessing_methods, noisy=True)
            if self.noisy_preprocessing_methods else None)
        self.kind = kind
        self.version = version
        self.split_indices = None
        self.label_superset = None
        self.excluded_classes = []
        self.class_names = None
        self.class_name_to_class_id = None
        self.class_id_to_class_name = None
        self.number_of_classes = None
        self.prediction_specifications = None
        self.predicted_cluster_ids = None
        self.predicted_labels = None
        self.predicted_class_names = None
        self.number_of_predicted_classes = None
        self.values = None
        self.preprocessed_values = None
        self.binarised_values = None
        self.total_standard_deviations = None
        self.explained_standard_deviations = None
        self.count_sum = None
        self.normalised_count_sum = None
        self.labels = None
        self.example_names = None
        self.feature_names = None
        self.batch_indices = None
        self.batch_names = None
        self.number_of_batches = None
        self.number_of_examples = None
        self.number_of_features = None
        self._generator = kwargs.get("generator")
        self.update(values=values, labels=labels, example_names=example_names,
                    feature_names=feature_names, batch_indices=batch_indices,
                    batch_names=batch_names,
                    preprocessed_values=preprocessed_values,
                    binarised_values=binarised_values,
                    total_standard_deviations=total_standard_deviations,
                    explained_standard_deviations=(
                        explained_standard_deviations))

    # -- predictions (data_set.py:682-742) ---------------------------------
    def update_predictions(self, prediction_specifications=None,
                           predicted_cluster_ids=None, predicted_labels=None,
                           predicted_class_names=None, **unused):
        if prediction_specifications is not None:
            self.prediction_specifications = prediction_specifications
        if predicted_cluster_ids is not None:
            self.predicted_cluster_ids = predicted_cluster_ids
        if predicted_labels is not None:
            self.predicted_labels = predicted_labels
            if predicted_class_names is None:
                predicted_class_names = numpy.unique(predicted_labels).tolist()
            self.predicted_class_names = predicted_class_names
            self.number_of_predicted_classes = len(predicted_class_names)

    def reset_predictions(self):
        self.prediction_specifications = None
        self.predicted_cluster_ids = None
        self.predicted_labels = None
        self.predicted_class_names = None
        self.number_of_predicted_classes = None

    @property
    def has_predictions(self):
        return (self.predicted_cluster_ids is not None
                or self.predicted_labels is not None)

    # -- properties used by the models / CLI --------------------------------
    @property
    def has_values(self):
        return self.values is not None

    @property
    def has_preprocessed_values(self):
        return self.preprocessed_values is not None

    @property
    def has_binarised_values(self):
        return self.binarised_values is not None

    @property
    def has_labels(self):
        return self.labels is not None

    @property
    def has_batches(self):
        return self.batch_indices is not None

    @property
    def number_of_values(self):
        return self.number_of_examples * self.number_of_features

    # -- construction ---------------------------------------------------------
    @staticmethod
    def _as_matrix(values):
        if values is None:
            return None
        if scipy.sparse.issparse(values):
            return SparseRowMatrix(values.astype(numpy.float32))
        values = numpy.asarray(values)
        return values

    def update(self, values=None, labels=None, example_names=None,
               feature_names=None, batch_indices=None, batch_names=None,
               preprocessed_values=None, binarised_values=None,
               total_standard_deviations=None,
               explained_standard_deviations=None):
        if values is not None:
            self.values = self._as_matrix(values)
            self.number_of_examples, self.number_of_features = (
                self.values.shape)
            if scipy.sparse.issparse(self.values):
                count_sum = numpy.asarray(
                    self.values.sum(axis=1)).reshape(-1, 1)
            else:
                count_sum = self.values.sum(axis=1).reshape(-1, 1)
            self.count_sum = count_sum.astype(numpy.float32)
            maximum = self.count_sum.max() if self.count_sum.size else 1.0
            self.normalised_count_sum = self.count_sum / max(maximum, 1e-30)
        if labels is not None:
            self.labels = numpy.asarray(labels)
            self.class_names = numpy.unique(self.labels).tolist()
            self.class_name_to_class_id = {
                name: i for i, name in enumerate(self.class_names)}
            self.class_id_to_class_name = {
                i: name for name, i in self.class_name_to_class_id.items()}
            self.number_of_classes = len(self.class_names)
        if example_names is not None:
            self.example_names = numpy.asarray(example_names)
        if feature_names is not None:
            self.feature_names = numpy.asarray(feature_names)
        if batch_indices is not None:
            self.batch_indices = numpy.asarray(batch_indices).reshape(-1, 1)
            self.number_of_batches = int(self.batch_indices.max()) + 1
        if batch_names is not None:
            self.batch_names = batch_names
        if preprocessed_values is not None:
            self.preprocessed_values = self._as_matrix(preprocessed_values)
        if binarised_values is not None:
            self.binarised_values = self._as_matrix(binarised_values)
        if total_standard_deviations is not None:
            self.total_standard_deviations = total_standard_deviations
        if explained_standard_deviations is not None:
            self.explained_standard_deviations = (
                explained_standard_deviations)

    def load(self):
        """Load a local file (``scvae_amd/data/loaders.py``) or materialise a
        built-in synthetic data set; nothing is downloaded."""
        if self.has_values:
            return
        if self.path is not None and self._generator is None:
            from scvae_amd.data.loaders import LOADERS, infer_data_format
            data_format = self.data_format
            if data_format in (None, "infer"):
                data_format = infer_data_format(self.path)
            data_format = normalise_string(data_format)
            if data_format not in LOADERS:
                raise ValueError(
                    "Data format `{}` not recognised (known: {}).".format(
                        data_format, ", ".join(sorted(LOADERS))))
            self.data_format = data_format
            path = self.path
            self._generator = lambda: LOADERS[data_format](path)
        if self._generator is None:
            from scvae_amd.data.synthetic import SYNTHETIC_DATA_SETS
            key = self.name
            if key not in SYNTHETIC_DATA_SETS:
                raise FileNotFoundError(
                    "Data set `{}` is neither a file nor a built-in synthetic "
                    "data set ({}); the catalogue of downloadable data sets "
                    "is outside the scope of this build."
                    .format(self.name, ", ".join(sorted(SYNTHETIC_DATA_SETS))))
            self._generator = SYNTHETIC_DATA_SETS[key]
        # the `.sparse.h5` cache of a parsed file (data_set.py:749-790): loaded
        # when it exists, written when parsing took long enough to be worth it
        sparse_path = self._sparse_cache_path() if self.path else None
        if sparse_path and os.path.isfile(sparse_path):
            from scvae_amd.data import internal_io
            print("Loading data set.")
            dictionary = internal_io.load_data_dictionary(sparse_path)
            dictionary["values"] = scipy.sparse.csr_matrix(
                dictionary["values"], dtype=numpy.float32)
        else:
            loading_time_start = time()
            dictionary = self._generator()
            loading_duration = time() - loading_time_start
            if sparse_path and loading_duration > float(os.environ.get(
                    "SCVAE_CACHE_AFTER_SECONDS",
                    MINIMUM_NUMBER_OF_SECONDS_BEFORE_SAVING)):
                from scvae_amd.data import internal_io
                print("Saving data set.")
                internal_io.save_data_dictionary(
                    {key: dictionary.get(key) for key in (
                        "values", "labels", "example names", "feature names")},
                    sparse_path)
        self.update(values=dictionary["values"],
                    labels=dictionary.get("labels"),
                    example_names=dictionary.get("example names"),
                    feature_names=dictionary.get("feature names"))
        self.preprocess()

    def _sparse_cache_path(self):
        """``<directory>/<name>/preprocessed/<name>.sparse.h5``
        (data_set.py:146-149, 1278-1315 without processing parts)."""
        return os.path.join(self.directory, self.name, PREPROCESS_SUFFIX,
                            self.name + PREPROCESSED_EXTENSION)

    def preprocess(self):
        """``preprocessing_methods`` applied to the values once
        (data_set.py:817-905: the result is the models' input x, the counts
        stay the target t); ``binarise()`` for the Bernoulli likelihood."""
        if self.preprocessing_methods and not self.has_preprocessed_values:
            print("Preprocessing values ({}).".format(
                ", ".join(self.preprocessing_methods)))
            self.update(preprocessed_values=build_preprocessor(
                self.preprocessing_methods)(self.values))

    def binarise(self):
        """data_set.py:984-1024: values > 0.5 as the Bernoulli targets."""
        if not self.has_binarised_values:
            self.update(binarised_values=build_preprocessor(["binarise"])(
                self.values))

    def _subset(self, indices, kind):
        subset = DataSet(
            self.name, title=self.title, specifications=self.specifications,
            values=self.values[indices],
            labels=self.labels[indices] if self.has_labels else None,
            example_names=(self.example_names[indices]
                           if self.example_names is not None else None),
            feature_names=self.feature_names,
            batch_indices=(self.batch_indices[indices]
                           if self.has_batches else None),
            batch_names=self.batch_names,
            preprocessed_values=(self.preprocessed_values[indices]
                                 if self.has_preprocessed_values else None),
            binarised_values=(self.binarised_values[indices]
                              if self.has_binarised_values else None),
            feature_selection=self.feature_selection,
            example_filter=self.example_filter,
            preprocessing_methods=self.preprocessing_methods,
            noisy_preprocessing_methods=self.noisy_preprocessing_methods,
            kind=kind, version=self.version)
        return subset

    def split(self, method=None, fraction=None):
        """Training/validation/test split (``scvae/data/processing.py:336-486``):
        ``RandomState(42)`` permutation, ``int(f*n)`` training+validation of
        which ``int(f*...)`` training (81/9/10 for f = 0.9)."""
        if method is None:
            method = defaults["data"]["splitting_method"]
        if fraction is None:
            fraction = defaults["data"]["splitting_fraction"]
        if not self.has_values:
            self.load()
        method = normalise_string(method)
        if method == "default":
            method = "random"
        n = self.number_of_examples
        random_state = numpy.random.RandomState(42)
        if method == "random":
            indices = random_state.permutation(n)
        elif method == "sequential":
            indices = numpy.arange(n)
        else:
            raise ValueError(
                "Splitting method `{}` not found.".format(method))
        n_training_validation = int(fraction * n)
        n_training = int(fraction * n_training_validation)
        self.split_indices = {
            "training": indices[:n_training],
            "validation": indices[n_training:n_training_validation],
            "test": indices[n_training_validation:],
        }
        return tuple(self._subset(self.split_indices[kind], kind)
                     for kind in ("training", "validation", "test"))

    def clear(self):
        self.values = None
        self.preprocessed_values = None
        self.binarised_values = None
        self.labels = None
        self.count_sum = None
        self.normalised_count_sum = None
