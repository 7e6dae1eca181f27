"""On-disk directory layout shared by the CLI and the models
(``scvae/data/utilities.py:68-142``) and the evaluation-subset picker
(``:145-181``)."""

import os

import numpy

from scvae_amd.utilities import normalise_string

EVALUATION_SUBSET_MAXIMUM_NUMBER_OF_EXAMPLES = 25
EVALUATION_SUBSET_MAXIMUM_NUMBER_OF_EXAMPLES_PER_CLASS = 3


def build_directory_path(base_directory, data_set, splitting_method=None,
                         splitting_fraction=None, preprocessing=True):
    """``<base>/<data set>/<no_split|split-<method>_<fraction>>/<preprocessing>``."""
    if splitting_method:
        if splitting_method == "default":
            splitting_method = (
                getattr(data_set, "default_splitting_method", None)
                or "random")
        if splitting_method == "indices":
            splitting_directory = "split-indices"
        else:
            splitting_directory = "split-{}_{}".format(
                splitting_method, splitting_fraction)
    else:
        splitting_directory = "no_split"

    parts = []
    if getattr(data_set, "features_mapped", False):
        parts.append("features_mapped")
    # (feature selection / example filters are not built -- SURVEY.md section 2
    # OUT OF SCOPE -- but a data set object that carries them still gets the
    # reference's directory name)
    for kind in ("feature_selection", "example_filter"):
        method = getattr(data_set, kind + "_method", None)
        if method:
            part = normalise_string(method)
            for parameter in getattr(data_set, kind + "_parameters",
                                     None) or ():
                part += "_" + normalise_string(str(parameter))
            parts.append(part)
    if preprocessing and data_set.preprocessing_methods:
        parts.extend(map(normalise_string, data_set.preprocessing_methods))
    if preprocessing and data_set.noisy_preprocessing_methods:
        parts.append("noisy")
        parts.extend(map(normalise_string,
                         data_set.noisy_preprocessing_methods))
    preprocessing_directory = "-".join(parts) if parts else "no_preprocessing"
    return os.path.join(base_directory, data_set.name, splitting_directory,
                        preprocessing_directory)


def indices_for_evaluation_subset(
        evaluation_set, maximum_number_of_examples_per_class=None,
        total_maximum_number_of_examples=None):
    if maximum_number_of_examples_per_class is None:
        maximum_number_of_examples_per_class = (
            EVALUATION_SUBSET_MAXIMUM_NUMBER_OF_EXAMPLES_PER_CLASS)
    if total_maximum_number_of_examples is None:
        total_maximum_number_of_examples = (
            EVALUATION_SUBSET_MAXIMUM_NUMBER_OF_EXAMPLES)
    random_state = numpy.random.RandomState(80)
    if evaluation_set.has_labels:
        subset = set()
        for class_name in evaluation_set.class_names:
            indices = numpy.argwhere(evaluation_set.labels == class_name)
            random_state.shuffle(indices)
            subset.update(
                *indices[:maximum_number_of_examples_per_class])
    else:
        subset = set(numpy.random.permutation(
            evaluation_set.number_of_examples)[
                :total_maximum_number_of_examples])
    return subset
