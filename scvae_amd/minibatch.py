"""Device-resident count matrix and minibatch fetch.

Replaces the host-side ``x_train[minibatch_indices].toarray()`` /
``t_train[minibatch_indices].toarray()`` of the reference loops
(``scvae/models/variational_autoencoder.py:985-998``): the CSR matrix
(``SparseRowMatrix``, ``scvae/data/sparse.py:22``) is uploaded once and every
minibatch is gathered and densified on the GPU by ``scvae_csr_densify``.  When
``x`` and ``t`` are the same matrix (no preprocessing, va:845-861) one dense
buffer serves both.
"""

import ctypes

import numpy
import torch

from scvae_amd import _lib
from scvae_amd.engine import current_stream_handle


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class DeviceCSR:
    """CSR count matrix in HBM (int64 indptr, int32 indices, fp32 values)."""

    def __init__(self, indptr, indices, values, shape, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.shape = (int(shape[0]), int(shape[1]))
        self.indptr = torch.as_tensor(indptr).to(
            device=self.device, dtype=torch.int64).contiguous()
        self.indices = torch.as_tensor(indices).to(
            device=self.device, dtype=torch.int32).contiguous()
        self.values = torch.as_tensor(values).to(
            device=self.device, dtype=torch.float32).contiguous()
        if self.indptr.numel() != self.shape[0] + 1:
            raise ValueError("indptr does not match the number of rows.")
        # precondition of the exact bf16-split input-layer kernels
        # (count_gemm.hip): every value an integer in [0, 65536), verified here
        # once -- preprocessed / noisy matrices fail it and keep the fp32 path
        self.integer_counts = False
        if self.values.numel():
            bad = torch.zeros(1, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.scvae_check_counts(
                _ptr(self.values), self.values.numel(), _ptr(bad),
                current_stream_handle(self.device)), "scvae_check_counts")
            self.integer_counts = int(bad.item()) == 0
        # data-only term of the count likelihoods: sum_f lgamma(1 + t[b, f])
        self.row_lgamma1p = torch.zeros(
            max(self.shape[0], 1), dtype=torch.float32, device=self.device)
        if self.shape[0]:
            _lib.check(self.lib.scvae_csr_row_lgamma1p(
                _ptr(self.indptr), _ptr(self.values), self.shape[0],
                _ptr(self.row_lgamma1p),
                current_stream_handle(self.device)), "scvae_csr_row_lgamma1p")

        self._max_row_entries = None

    @property
    def max_row_entries(self):
        """Most entries a row contributes to a ``CountTiles`` (its non-zeros,
        plus one for every count of more than 8 significant bits): sizes the
        per-group capacity.  Computed on first use (integer count matrices)."""
        if self._max_row_entries is None:
            if not self.integer_counts or not self.shape[0]:
                self._max_row_entries = 0
            else:
                per_row = torch.empty(self.shape[0], dtype=torch.int32,
                                      device=self.device)
                _lib.check(self.lib.scvae_csr_row_entries(
                    _ptr(self.indptr), _ptr(self.values), self.shape[0],
                    _ptr(per_row), current_stream_handle(self.device)),
                    "scvae_csr_row_entries")
                self._max_row_entries = int(per_row.max().item())
        return self._max_row_entries

    @property
    def count_tiles_supported(self):
        """Whether minibatches of this matrix can also be kept as tile-indexed
        non-zeros (``count_tiles``): integer counts, at most 65 536 genes."""
        return bool(self.integer_counts and 0 < self.shape[1] <= 65536
                    and self.max_row_entries > 0)

    def count_tiles(self, n_rows):
        """Buffers for a minibatch of up to ``n_rows`` rows as tile-indexed
        non-zeros (``CountTiles``); fill with ``gather_count_tiles`` or let the
        previous step carry it (``request(..., tiles=)``)."""
        if not self.count_tiles_supported:
            raise ValueError("count tiles: an integer count matrix of at most "
                             "65 536 genes expected")
        return CountTiles(self, int(n_rows))

    def gather_count_tiles(self, rows, tiles):
        """``tiles`` <- the minibatch ``rows`` as tile-indexed non-zeros."""
        n = int(rows.numel())
        if n > tiles.max_rows:
            raise ValueError("more rows than the tiles were sized for")
        _lib.check(self.lib.scvae_csr_count_tiles(
            _ptr(self.indptr), _ptr(self.indices), _ptr(self.values),
            _ptr(rows), n, self.shape[1], ctypes.byref(tiles.struct),
            current_stream_handle(self.device)), "scvae_csr_count_tiles")
        return tiles

    @classmethod
    def from_scipy(cls, matrix, device, rows=None):
        """Upload a scipy matrix -- all of it, or (``rows=(lo, hi)``) only the
        contiguous row shard ``lo .. hi - 1``, whose local row ``i`` is row
        ``lo + i`` of the matrix: a data-parallel rank that draws its cells
        from its own 1 / W of a matrix too large to replicate (SURVEY section
        8e: the 1.3 M-cell matrix is ~20 GB of CSR)."""
        matrix = matrix.tocsr()
        if rows is not None:
            lo, hi = int(rows[0]), int(rows[1])
            if not 0 <= lo <= hi <= matrix.shape[0]:
                raise ValueError("row shard outside the matrix")
            matrix = matrix[lo:hi]
        if not matrix.has_canonical_format:
            # duplicate (row, column) entries are summed, as ``.toarray()`` of
            # the reference's minibatch fetch does (va:997-998); the densify
            # kernel and the lgamma row term take one entry per position
            matrix = matrix.copy()
            matrix.sum_duplicates()
        matrix.sort_indices()
        return cls(matrix.indptr.astype(numpy.int64),
                   matrix.indices.astype(numpy.int32),
                   matrix.data.astype(numpy.float32), matrix.shape, device)

    def row_shard(self, lo, hi):
        """The rows ``lo .. hi - 1`` as a matrix of their own (device-side
        copy of the three arrays; local row ``i`` = row ``lo + i``)."""
        lo, hi = int(lo), int(hi)
        if not 0 <= lo <= hi <= self.shape[0]:
            raise ValueError("row shard outside the matrix")
        first = int(self.indptr[lo].item())
        last = int(self.indptr[hi].item())
        return DeviceCSR(self.indptr[lo:hi + 1] - first,
                         self.indices[first:last].clone(),
                         self.values[first:last].clone(),
                         (hi - lo, self.shape[1]), self.device)

    @property
    def number_of_rows(self):
        return self.shape[0]

    @property
    def nnz(self):
        return int(self.values.numel())

    def gather_dense(self, rows, out=None, row_const_out=None):
        """Dense fp32 ``[len(rows), F]`` minibatch (and its lgamma row term)."""
        n, F = int(rows.numel()), self.shape[1]
        if out is None:
            out = torch.empty((n, F), dtype=torch.float32, device=self.device)
        stream = current_stream_handle(self.device)
        _lib.check(self.lib.scvae_csr_minibatch(
            _ptr(self.indptr), _ptr(self.indices), _ptr(self.values),
            _ptr(rows), n, F, _ptr(out), out.stride(0), 0,
            _ptr(self.row_lgamma1p),
            _ptr(row_const_out) if row_const_out is not None else None,
            stream), "scvae_csr_minibatch")
        return out


    def request(self, rows, out, row_const_out=None, tiles=None):
        """A fetch of ``rows`` into ``out`` (fp32 ``[n, F]`` or uint16
        ``[n, u16_pitch]``) to hand to ``Engine.step(next_minibatch=...)``;
        ``tiles``: a ``CountTiles`` to fill with the same rows (uint16 only)."""
        return MinibatchRequest(self, rows, out, row_const_out, tiles)

    @property
    def u16_pitch(self):
        """Row pitch (elements) of the uint16 minibatch: whole 128-byte lines."""
        return (self.shape[1] + 63) // 64 * 64

    def gather_counts_u16(self, rows, out=None, row_const_out=None):
        """The minibatch as uint16 counts ``[len(rows), u16_pitch]`` (columns past
        F zeroed) for the kernels that stream it -- integer count matrices only
        (``integer_counts``).  Pass it to ``Engine.step`` as both x and t."""
        if not self.integer_counts:
            raise ValueError("not an integer count matrix below 65 536")
        n, F, ld = int(rows.numel()), self.shape[1], self.u16_pitch
        if out is None:
            out = torch.empty((n, ld), dtype=torch.uint16, device=self.device)
        stream = current_stream_handle(self.device)
        _lib.check(self.lib.scvae_csr_minibatch(
            _ptr(self.indptr), _ptr(self.indices), _ptr(self.values),
            _ptr(rows), n, F, _ptr(out), out.stride(0), 1,
            _ptr(self.row_lgamma1p),
            _ptr(row_const_out) if row_const_out is not None else None,
            stream), "scvae_csr_minibatch")
        return out


class CountTiles:
    """A count minibatch as the list of its non-zeros grouped by (16 rows, 32
    genes) -- ``scvae_count_tiles`` (include/scvae_hip.h).  Handed to
    ``Engine.step(count_tiles=)`` next to the uint16 batch of the same rows,
    the input layer's two products (mu:53-59) read it instead of the dense
    batch: same arithmetic, a tenth of the bytes."""

    def __init__(self, matrix, max_rows):
        lib = matrix.lib
        device = matrix.device
        self.max_rows = max_rows
        groups = (max_rows + 15) // 16
        tiles = int(lib.scvae_count_tiles_padded(matrix.shape[1]))
        self.capacity = 16 * matrix.max_row_entries
        if groups * self.capacity >= 2 ** 31:
            raise ValueError("count tiles: minibatch too large")
        self.entries = torch.zeros(max(groups * self.capacity, 1),
                                   dtype=torch.int32, device=device)
        self.tile_ptr = torch.zeros(groups * (tiles + 1), dtype=torch.int32,
                                    device=device)
        self.block_ptr = torch.zeros(groups * (tiles // 16 + 1),
                                     dtype=torch.int32, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.struct = _lib.CountTilesStruct()
        self.struct.entries = self.entries.data_ptr()
        self.struct.tile_ptr = self.tile_ptr.data_ptr()
        self.struct.block_ptr = self.block_ptr.data_ptr()
        self.struct.capacity = self.capacity
        self.struct.status = self.status.data_ptr()

    @property
    def address(self):
        return ctypes.addressof(self.struct)


class MinibatchRequest:
    """The arguments of one minibatch fetch (``scvae_csr_minibatch``), to be
    carried by the step before it (``Engine.step(next_minibatch=...)``: the
    densify then runs under that step's backward pass) or issued directly
    (``issue()``).  Holds references to every tensor involved."""

    def __init__(self, matrix, rows, out, row_const_out=None, tiles=None):
        if tiles is not None and (out.dtype != torch.uint16
                                  or rows.numel() > tiles.max_rows):
            raise ValueError("count tiles go with a uint16 minibatch of at "
                             "most the rows they were sized for")
        self.tiles = tiles
        if out.dtype not in (torch.uint16, torch.float32):
            raise ValueError("uint16 or float32 minibatch buffer expected")
        if out.dtype == torch.uint16 and not matrix.integer_counts:
            raise ValueError("not an integer count matrix below 65 536")
        if out.dim() != 2 or out.stride(1) != 1 or out.shape[0] < rows.numel():
            raise ValueError("row-major [>= len(rows), >= F] buffer expected")
        self.matrix, self.rows, self.out = matrix, rows, out
        self.row_const_out = row_const_out

    def fill(self, side):
        m = self.matrix
        side.fetch_as_u16 = 1 if self.out.dtype == torch.uint16 else 0
        side.fetch_indptr = m.indptr.data_ptr()
        side.fetch_indices = m.indices.data_ptr()
        side.fetch_values = m.values.data_ptr()
        side.fetch_rows = self.rows.data_ptr()
        side.fetch_n = int(self.rows.numel())
        side.fetch_features = m.shape[1]
        side.fetch_out = self.out.data_ptr()
        side.fetch_ld = self.out.stride(0)
        side.fetch_row_values = m.row_lgamma1p.data_ptr()
        side.fetch_row_values_out = (
            self.row_const_out.data_ptr()
            if self.row_const_out is not None else None)
        side.fetch_tiles = (self.tiles.address if self.tiles is not None
                            else None)

    def issue(self):
        if self.out.dtype == torch.uint16:
            self.matrix.gather_counts_u16(self.rows, out=self.out,
                                          row_const_out=self.row_const_out)
            if self.tiles is not None:
                self.matrix.gather_count_tiles(self.rows, self.tiles)
        else:
            self.matrix.gather_dense(self.rows, out=self.out,
                                     row_const_out=self.row_const_out)


def philox_normal(out, row_offset, seed, stream_id):
    """Fill ``out`` ([rows, cols], fp32, device) with N(0,1) draws keyed by
    (seed, stream_id, row_offset + row, col)."""
    lib = _lib.load()
    rows = out.shape[0] if out.dim() > 1 else out.numel()
    cols = out.numel() // max(rows, 1)
    _lib.check(lib.scvae_philox_normal(
        _ptr(out), rows, cols, int(row_offset), int(seed), int(stream_id),
        current_stream_handle(out.device)), "scvae_philox_normal")
    return out


def philox_normal_blocks(out, block_stride, row_offset, seed, stream_id):
    """``out`` [blocks, rows, cols] in ONE launch: block g, row r gets the draw
    of row ``g * block_stride + row_offset + r`` -- the stacked passes (GMVAE
    clusters, importance / Monte-Carlo samples) of a rank's shard of a global
    minibatch of ``block_stride`` cells; equal to ``philox_normal`` block by
    block."""
    lib = _lib.load()
    blocks, rows, cols = out.shape
    if not out.is_contiguous():
        raise ValueError("contiguous [blocks, rows, cols] expected")
    _lib.check(lib.scvae_philox_normal_blocks(
        _ptr(out), blocks, rows, cols, int(block_stride), int(row_offset),
        int(seed), int(stream_id), current_stream_handle(out.device)),
        "scvae_philox_normal_blocks")
    return out


def synthetic_count_matrix(n_cells, n_features, density=0.05, n_clusters=8,
                           seed=60, device="cuda:0", chunk=2048):
    """Synthetic cell x gene counts of a named shape, generated on the GPU.

    Vectorised restatement of the reference's ``development`` generator
    (``scvae/data/loaders.py:942-1022``): per cluster and gene
    ``r ~ 10*U(0,1)``, ``p ~ U(0,1)``, keep probability ``~ U(0,1)``; counts
    are negative binomial (gamma-Poisson) times a Bernoulli keep mask.  The
    keep probabilities are rescaled so that the fraction of nonzeros is about
    ``density`` (10x matrices are 93-98 % zeros).  Returns a ``DeviceCSR`` and
    the cluster label of every cell.
    """
    device = torch.device(device)
    g = torch.Generator(device=device).manual_seed(int(seed))

    def U(*shape):
        return torch.rand(*shape, generator=g, device=device)

    r = 10.0 * U(n_clusters, n_features) + 1e-3
    # NB(r, p) has mean r(1-p)/p; keep p away from 0 so counts stay bounded
    p = 0.05 + 0.95 * U(n_clusters, n_features)
    keep = U(n_clusters, n_features)
    # P(NB > 0) = 1 - p^r ; rescale keep to hit the target density
    p_nonzero = (keep * (1.0 - p ** r)).mean().item()
    keep = torch.clamp(keep * (density / max(p_nonzero, 1e-12)), max=1.0)
    labels = torch.randint(0, n_clusters, (n_cells,), generator=g,
                           device=device)

    indptr = [torch.zeros(1, dtype=torch.int64, device=device)]
    indices, values = [], []
    offset = 0
    for start in range(0, n_cells, chunk):
        lab = labels[start:start + chunk]
        rr, pp, kk = r[lab], p[lab], keep[lab]
        rate = torch._standard_gamma(rr, generator=g) * (1.0 - pp) / pp
        counts = torch.poisson(rate, generator=g)
        counts = counts * (torch.rand(counts.shape, generator=g,
                                      device=device) < kk)
        nz = counts.nonzero(as_tuple=False)
        row_counts = torch.bincount(nz[:, 0], minlength=lab.numel())
        indptr.append(offset + torch.cumsum(row_counts, 0))
        offset += int(nz.shape[0])
        indices.append(nz[:, 1].to(torch.int32))
        values.append(counts[nz[:, 0], nz[:, 1]].to(torch.float32))
    matrix = DeviceCSR(torch.cat(indptr), torch.cat(indices),
                       torch.cat(values), (n_cells, n_features), device)
    return matrix, labels
