from scvae_amd.data.data_set import DataSet
from scvae_amd.data.sparse import SparseRowMatrix

__all__ = ["DataSet", "SparseRowMatrix"]
