"""BASELINE config 5's matrix at its real size on the device (VERDICT round 5,
missing 4): 1 306 127 cells x 27 998 genes at ~6 % non-zeros is 2.2 x 10^9
stored values -- past 2^31 -- where everything else in the suite uses a
16 384-row sample.  The upload (whole, and as one rank's 1 / 8 row shard,
SURVEY section 8e), the one-off passes over the matrix (count check, lgamma row
term, entries per row) and the minibatch fetch in all three forms (fp32,
uint16, tile-indexed non-zeros; va:985-998 on the device) are exercised on rows
from the start, from either side of the 2^31-th stored value and from the end,
against scipy on the host.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu

CELLS, FEATURES = 1306127, 27998


@pytest.fixture(scope="module")
def mouse_brain(cuda_device):
    from scvae_amd.minibatch import synthetic_count_matrix
    matrix, _ = synthetic_count_matrix(CELLS, FEATURES, density=0.06, seed=61,
                                       device=cuda_device, chunk=8192)
    torch.cuda.synchronize()
    return matrix


def _host_rows(matrix, lo, hi):
    """Rows lo .. hi - 1 as a scipy CSR (copied from the device arrays)."""
    first = int(matrix.indptr[lo].item())
    last = int(matrix.indptr[hi].item())
    indptr = (matrix.indptr[lo:hi + 1] - first).cpu().numpy()
    return sp.csr_matrix(
        (matrix.values[first:last].cpu().numpy(),
         matrix.indices[first:last].cpu().numpy().astype(np.int64), indptr),
        shape=(hi - lo, matrix.shape[1]))


def _ranges(matrix):
    """Three row ranges: the start, around the 2^31-th stored value, the end."""
    crossing = int(torch.searchsorted(
        matrix.indptr, torch.tensor([2 ** 31], device=matrix.device)).item())
    assert 1000 < crossing < CELLS - 1000
    return [(0, 700), (crossing - 600, crossing + 600), (CELLS - 1500, CELLS)]


def test_matrix_is_past_two_to_the_31(mouse_brain):
    m = mouse_brain
    assert m.shape == (CELLS, FEATURES)
    assert m.nnz > 2 ** 31
    assert int(m.indptr[-1].item()) == m.nnz
    assert m.integer_counts                      # scvae_check_counts over all of it
    assert m.count_tiles_supported and m.max_row_entries > 0
    # the lgamma row term of the last row, past 2^31 stored values
    last = _host_rows(m, CELLS - 1, CELLS)
    from scipy.special import gammaln
    want = gammaln(1.0 + last.data.astype(np.float64)).sum()
    assert abs(float(m.row_lgamma1p[-1].item()) - want) <= 1e-5 * max(want, 1.0)


def test_minibatches_from_both_sides_of_two_to_the_31(mouse_brain):
    m = mouse_brain
    ranges = _ranges(m)
    host = sp.vstack([_host_rows(m, lo, hi) for lo, hi in ranges]).tocsr()
    ids = np.concatenate([np.arange(lo, hi) for lo, hi in ranges])
    rng = np.random.default_rng(3)
    order = rng.permutation(len(ids))[:3072]
    rows = torch.from_numpy(ids[order]).to(m.device)
    want = np.asarray(host[order].toarray())
    rc = torch.empty(len(order), device=m.device)
    x32 = m.gather_dense(rows, row_const_out=rc)
    x16 = m.gather_counts_u16(rows)
    tiles = m.count_tiles(len(order))
    m.gather_count_tiles(rows, tiles)
    torch.cuda.synchronize()
    assert np.array_equal(x32.cpu().numpy(), want)
    assert np.array_equal(x16[:, :FEATURES].cpu().numpy().astype(np.float32), want)
    assert (x16[:, FEATURES:].to(torch.int32) == 0).all()
    assert torch.equal(rc, m.row_lgamma1p[rows])
    assert int(tiles.status.item()) == 0
    # the tiles hold the same rows: the input layer's product from them is the dense one's
    import ctypes
    from scvae_amd import _lib
    lib = _lib.load()
    n, N = len(order), 64
    g = torch.Generator(device=m.device).manual_seed(1)
    W = torch.randn(FEATURES, N, generator=g, device=m.device) * 0.05
    nbytes = lib.scvae_count_gemm_workspace_bytes(0, n, FEATURES, N)
    ws = torch.empty(nbytes + 16, dtype=torch.uint8, device=m.device)
    outs = [torch.empty(n, N, device=m.device) for _ in range(2)]
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.scvae_count_gemm_u16(0, P(x16), x16.stride(0), n, FEATURES, P(W), N, N,
                                        None, 0, P(outs[0]), N, P(ws), nbytes, st), "dense")
    _lib.check(lib.scvae_count_gemm_tiles(0, ctypes.byref(tiles.struct), P(x16), x16.stride(0),
                                          n, FEATURES, P(W), N, N, None, 0, P(outs[1]), N,
                                          P(ws), nbytes, st), "tiles")
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])


def test_row_shard_of_one_rank_of_eight(mouse_brain):
    """a rank may hold only its 1 / W row shard.  The last of
    eight shards -- every stored value of it past 2^31 -- uploaded from the host
    matrix (scipy, int64 index arrays) and cut on the device: same rows, same
    minibatches as the whole matrix gives."""
    from scvae_amd.minibatch import DeviceCSR
    m = mouse_brain
    lo, hi = 7 * (CELLS // 8), CELLS
    assert int(m.indptr[hi].item()) > 2 ** 31
    on_device = m.row_shard(lo, hi)
    host = _host_rows(m, lo, hi)                      # this rank's part of the file
    from_host = DeviceCSR.from_scipy(host, m.device)
    whole_host = sp.csr_matrix(
        (host.data, host.indices, host.indptr + 0), shape=host.shape)
    sliced = DeviceCSR.from_scipy(whole_host, m.device, rows=(100, hi - lo))
    torch.cuda.synchronize()
    for shard in (on_device, from_host):
        assert shard.shape == (hi - lo, FEATURES)
        assert torch.equal(shard.indptr, on_device.indptr)
        assert torch.equal(shard.indices, on_device.indices)
        assert torch.equal(shard.values, on_device.values)
        assert shard.integer_counts
    local = torch.arange(hi - lo - 2048, hi - lo, device=m.device)
    a = on_device.gather_counts_u16(local)
    b = m.gather_counts_u16(local + lo)
    c = sliced.gather_counts_u16(local - 100)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(on_device.row_lgamma1p[local], m.row_lgamma1p[local + lo])
