#!/usr/bin/env python3
"""HDF5 fixtures for scvae_amd/data/hdf5.py, written by the real library.

Run with an interpreter that has h5py (this container: /opt/conda/bin/python3.9,
h5py 3.3 on libhdf5 1.10) from the repository root:

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py

Writes tests/golden/hdf5_*.h5 and hdf5_expected.npz (what the files hold, as
h5py reads it back).  It also reads back tests/golden/hdf5_written_by_us.h5 --
a file produced by scvae_amd.data.hdf5.Writer (made by
tests/test_hdf5.py::test_writer... when SCVAE_WRITE_HDF5_SAMPLE is set) -- and
checks that libhdf5 sees in it what was put there.
"""
import json
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(7)
expected = {}


def keep(key, value):
    expected[key] = np.asarray(value)


# ---- a CellRanger-2-style count matrix: one genome group, chunked + shuffle + deflate ----
n_genes, n_cells = 300, 120
dense = (rng.poisson(1.5, (n_genes, n_cells)) * (rng.rand(n_genes, n_cells) < 0.1))
dense[5, 7] = 70000          # beyond 16 bits: stays in the file as it is
import scipy.sparse as sp
csc = sp.csc_matrix(dense.astype(np.int32))
barcodes = np.array(["ACGT{:012d}-1".format(i) for i in range(n_cells)], dtype="S18")
gene_names = np.array(["GENE{}".format(i) for i in range(n_genes)], dtype="S")
genes = np.array(["ENSG{:011d}".format(i) for i in range(n_genes)], dtype="S")
with h5py.File(os.path.join(HERE, "hdf5_tenx.h5"), "w") as f:
    g = f.create_group("GRCh38")
    opts = dict(compression="gzip", compression_opts=4, shuffle=True)
    g.create_dataset("data", data=csc.data.astype(np.int32), chunks=(512,), **opts)
    g.create_dataset("indices", data=csc.indices.astype(np.int64), chunks=(512,), **opts)
    g.create_dataset("indptr", data=csc.indptr.astype(np.int64), chunks=(64,), **opts)
    g.create_dataset("shape", data=np.array(csc.shape, dtype=np.int32), chunks=(2,), **opts)
    g.create_dataset("barcodes", data=barcodes, chunks=(50,), **opts)
    g.create_dataset("gene_names", data=gene_names, chunks=(128,), **opts)
    g.create_dataset("genes", data=genes, chunks=(128,), **opts)
keep("tenx/dense", dense.T)
keep("tenx/barcodes", barcodes)
keep("tenx/gene_names", gene_names)

# ---- assorted layouts, types, filters, attributes ----
with h5py.File(os.path.join(HERE, "hdf5_assorted.h5"), "w") as f:
    f.attrs["TITLE"] = np.string_("a title")
    f.attrs["numbers"] = np.arange(5, dtype=np.int16)
    f.attrs["pi"] = np.float64(3.25)
    a = rng.randn(13, 7)
    f.create_dataset("contiguous_f8", data=a)
    keep("assorted/contiguous_f8", a)
    b = np.arange(-6, 6, dtype=np.int8)
    f.create_dataset("compact_i1", data=b, dtype="i1", track_times=False)
    keep("assorted/compact_i1", b)
    c = np.arange(40, dtype=">i4").reshape(5, 8)
    f.create_dataset("big_endian", data=c, dtype=">i4")
    keep("assorted/big_endian", c.astype("<i4"))
    d = rng.randint(0, 1000, (37, 29)).astype(np.uint16)
    f.create_dataset("chunked_edges", data=d, chunks=(16, 10), fletcher32=True,
                     compression="gzip", shuffle=True)
    keep("assorted/chunked_edges", d)
    e = rng.rand(100).astype(np.float32)
    f.create_dataset("chunked_plain", data=e, chunks=(32,))
    keep("assorted/chunked_plain", e)
    s = f.create_dataset("vlen_strings", (4,), dtype=h5py.string_dtype())
    words = ["alpha", "", "gamma delta", "æøå"]
    s[:] = words
    keep("assorted/vlen_strings", np.array(words, dtype="U"))
    flags = np.array([True, False, True])
    f.create_dataset("flags", data=flags)
    keep("assorted/flags", flags.astype(np.uint8))
    f.create_dataset("empty", shape=(0,), dtype="f4")
    f.create_dataset("scalar", data=np.int64(42))
    keep("assorted/scalar", np.int64(42))
    f.create_dataset("unwritten", shape=(6,), dtype="i4")       # no storage: fill value
    keep("assorted/unwritten", np.zeros(6, np.int32))
    nested = f.create_group("outer").create_group("inner")
    nested.attrs["TITLE"] = np.string_("inner group")
    nested.create_dataset("x", data=np.arange(3))
    keep("assorted/outer/inner/x", np.arange(3))
    many = f.create_group("many")          # more members than one symbol table node holds
    for i in range(23):
        many.create_dataset("member_{:02d}".format(i), data=np.full(2, i, dtype=np.int32))
    keep("assorted/many_count", 23)

# ---- the newer file format: version-2 object headers, compact links ----
with h5py.File(os.path.join(HERE, "hdf5_latest.h5"), "w", libver="latest") as f:
    g = f.create_group("group")
    g.attrs["TITLE"] = np.string_("new style")
    v = np.arange(10, dtype=np.float32)
    g.create_dataset("contiguous", data=v)
    keep("latest/contiguous", v)
    w = np.arange(12, dtype=np.int32).reshape(3, 4)
    g.create_dataset("single_chunk", data=w, chunks=(3, 4), compression="gzip")
    keep("latest/single_chunk", w)
    g.create_dataset("fixed_array_index", data=np.arange(100), chunks=(10,))   # unsupported index

np.savez_compressed(os.path.join(HERE, "hdf5_expected.npz"), **expected)

# ---- a file written by our own writer, read back by libhdf5 ----
ours = os.path.join(HERE, "hdf5_written_by_us.h5")
if os.path.exists(ours):
    with open(os.path.join(HERE, "hdf5_written_by_us.json")) as handle:
        want = json.load(handle)
    with h5py.File(ours, "r") as f:
        def check(group, spec, path="/"):
            for key, value in spec.get("attrs", {}).items():
                got = group.attrs[key]
                got = got.decode() if isinstance(got, bytes) else got
                assert np.array_equal(got, value), (path, key, got, value)
            assert sorted(group.keys()) == sorted(spec["members"]), (path, list(group.keys()))
            for name, member in spec["members"].items():
                node = group[name]
                if "members" in member:
                    check(node, member, path + name + "/")
                    continue
                got = node[()]
                if got.dtype.kind == "S":
                    got = np.char.decode(got, "utf-8")
                assert np.array_equal(got, np.asarray(member["data"])), (path, name)
                for key, value in member.get("attrs", {}).items():
                    a = node.attrs[key]
                    a = a.decode() if isinstance(a, bytes) else a
                    assert np.array_equal(a, value), (path, name, key)
        check(f, want)
    print("libhdf5 reads hdf5_written_by_us.h5 as written")
print("fixtures written:", sorted(k for k in os.listdir(HERE) if k.startswith("hdf5_")))
