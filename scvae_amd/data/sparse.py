"""``SparseRowMatrix`` (``scvae/data/sparse.py:22-60``): CSR matrix with the
whole-matrix ``mean``/``std``/``var`` the data layer expects."""

import numpy
import scipy.sparse


class SparseRowMatrix(scipy.sparse.csr_matrix):

    @property
    def size(self):
        return self.shape[0] * self.shape[1]

    def mean(self, axis=None):
        if axis is not None:
            return super().mean(axis)
        dtype = self.dtype.type
        if numpy.issubdtype(dtype, numpy.integer):
            dtype = numpy.float64
        return (self.data.sum() / self.size).astype(dtype)

    def var(self, axis=None, ddof=0):
        variance = self.power(2).mean(axis) - numpy.power(self.mean(axis), 2)
        if ddof > 0:
            size = numpy.prod(self.shape)
            variance = variance * size / (size - ddof)
        return variance

    def std(self, axis=None, ddof=0):
        return numpy.sqrt(self.var(axis=axis, ddof=ddof))


def sparsity(a, tolerance=1e-3, batch_size=None):
    if scipy.sparse.issparse(a):
        return 1.0 - a.nnz / (a.shape[0] * a.shape[1])
    return float((numpy.abs(a) < tolerance).mean())
