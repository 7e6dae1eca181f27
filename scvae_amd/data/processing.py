"""Value preprocessing of a data set (``scvae/data/processing.py:305-333,
496-522``): the encoder input x may be a transformed copy of the counts, the
likelihood always sees the counts themselves (or, for the Bernoulli likelihood,
their binarisation).  Element-wise NumPy / SciPy work on the host, once per
data set; sparse matrices stay sparse (every method maps 0 to 0)."""
from functools import reduce

import numpy
import scipy.sparse

PREPROCESSERS = {}


def _register(name):
    def decorator(function):
        PREPROCESSERS[name] = function
        return function
    return decorator


def _on_values(values, function):
    if scipy.sparse.issparse(values):
        result = values.copy().astype(numpy.float32)
        result.data = function(result.data).astype(numpy.float32)
        result.eliminate_zeros()
        return result
    return function(numpy.asarray(values, dtype=numpy.float32)).astype(
        numpy.float32)


@_register("log")
def _log(values):
    return _on_values(values, numpy.log1p)


@_register("exp")
def _exp(values):
    return _on_values(values, numpy.expm1)


@_register("normalise")
def _normalise(values):
    """Every feature (column) scaled to unit l2 norm
    (``sklearn.preprocessing.normalize(values, norm="l2", axis=0)``)."""
    if scipy.sparse.issparse(values):
        values = scipy.sparse.csr_matrix(values, dtype=numpy.float32)
        norms = numpy.sqrt(numpy.asarray(
            values.multiply(values).sum(axis=0))).reshape(-1)
    else:
        values = numpy.asarray(values, dtype=numpy.float32)
        norms = numpy.sqrt((values * values).sum(axis=0))
    scale = numpy.where(norms > 0, 1.0 / numpy.maximum(norms, 1e-30), 1.0)
    if scipy.sparse.issparse(values):
        return scipy.sparse.csr_matrix(
            values.multiply(scale.reshape(1, -1)), dtype=numpy.float32)
    return (values * scale).astype(numpy.float32)


@_register("binarise")
def _binarise(values):
    """``sklearn.preprocessing.binarize(values, threshold=0.5)``."""
    return _on_values(values, lambda v: (v > 0.5).astype(numpy.float32))


@_register("bernoulli_sample")
def _bernoulli_sample(values):
    """A Bernoulli draw per value, the value its probability
    (``scvae/data/processing.py:516-522``: what "binarise" means when the
    preprocessing is noisy -- a new sample of the data set every epoch).
    NumPy's global generator, as in the reference: the draws of a run follow
    ``numpy.random.seed``."""
    def draw(probabilities):
        probabilities = numpy.asarray(probabilities, dtype=numpy.float64)
        if probabilities.size and (probabilities.min() < 0
                                   or probabilities.max() > 1):
            raise ValueError(
                "Bernoulli sampling needs values in [0, 1]; found [{:g}, {:g}]."
                .format(probabilities.min(), probabilities.max()))
        return numpy.random.binomial(1, probabilities)
    return _on_values(values, draw)


def build_preprocessor(preprocessing_methods, noisy=False):
    preprocessers = []
    for method in preprocessing_methods or []:
        if noisy and method == "binarise":
            method = "bernoulli_sample"
        if method not in PREPROCESSERS:
            raise ValueError(
                "Preprocessing method `{}` not found.".format(method))
        preprocessers.append(PREPROCESSERS[method])

    def preprocess(values):
        return reduce(lambda v, p: p(v), preprocessers, values)
    return preprocess
