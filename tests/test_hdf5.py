"""``scvae_amd/data/hdf5.py`` (a NumPy-only reader / writer of the HDF5 subset
10x count matrices and the ``.sparse.h5`` cache use) against files written by
the real library: ``tests/golden/hdf5_*.h5`` were produced with h5py 3.3 on
libhdf5 1.10 by ``tests/golden/make_hdf5_fixtures.py`` together with
``hdf5_expected.npz``, what h5py reads back from them.  Then the loaders on
top (``scvae/data/loaders.py:93-121, 651-676, 725-798``) and the cache
(``scvae/data/internal_io.py``; ``data_set.py:749-790``).  CPU only."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from scvae_amd.data import hdf5

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(GOLDEN, "hdf5_expected.npz"))


def test_reader_on_a_chunked_compressed_count_matrix(expected):
    with hdf5.File(os.path.join(GOLDEN, "hdf5_tenx.h5")) as f:
        assert f.root.keys() == ["GRCh38"]
        group = f.root["GRCh38"]
        assert set(group.keys()) == {"data", "indices", "indptr", "shape",
                                     "barcodes", "gene_names", "genes"}
        table = {k: group[k].read() for k in group.keys()}
        assert group["data"].shape == table["data"].shape
    matrix = sp.csc_matrix((table["data"], table["indices"], table["indptr"]),
                           shape=table["shape"])
    assert np.array_equal(matrix.T.toarray(), expected["tenx/dense"])
    assert np.array_equal(table["barcodes"], expected["tenx/barcodes"])
    assert np.array_equal(table["gene_names"], expected["tenx/gene_names"])


def test_reader_on_layouts_types_filters_and_attributes(expected):
    with hdf5.File(os.path.join(GOLDEN, "hdf5_assorted.h5")) as f:
        attrs = f.root.attributes
        assert attrs["TITLE"] == "a title" and attrs["pi"] == 3.25
        assert np.array_equal(attrs["numbers"], np.arange(5))
        for key in expected.files:
            if not key.startswith("assorted/") or key.endswith("many_count"):
                continue
            got = f.root[key[len("assorted/"):]].read()
            if got.dtype == object:
                got = got.astype("U")
            assert np.array_equal(got, expected[key]), key
        assert f.root["big_endian"].read().dtype.byteorder in "=<"
        assert f.root["empty"].read().shape == (0,)
        # 23 members: several symbol-table nodes under one B-tree node
        many = f.root["many"]
        assert len(many.keys()) == int(expected["assorted/many_count"])
        assert [int(many[k].read()[0]) for k in many.keys()] == list(range(23))
        assert f.root["outer/inner"].attributes["TITLE"] == "inner group"
        names = [node.name for node in f.root.walk()]
        assert "/outer/inner/x" in names and "/many/member_22" in names
        with pytest.raises(KeyError):
            f.root["no/such/node"]


def test_reader_on_the_newer_file_format(expected):
    """libver='latest': version-2 object headers, link messages, layout 4."""
    with hdf5.File(os.path.join(GOLDEN, "hdf5_latest.h5")) as f:
        group = f.root["group"]
        assert group.attributes["TITLE"] == "new style"
        assert np.array_equal(group["contiguous"].read(),
                              expected["latest/contiguous"])
        assert np.array_equal(group["single_chunk"].read(),
                              expected["latest/single_chunk"])
        with pytest.raises(hdf5.Hdf5Error, match="chunk index type"):
            group["fixed_array_index"].read()   # named, not mis-read


def test_not_hdf5(tmp_path):
    path = tmp_path / "x.h5"
    path.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(hdf5.Hdf5Error):
        hdf5.File(str(path))


def _sample_dictionary():
    rng = np.random.RandomState(3)
    dense = (rng.poisson(2.0, (40, 25)) * (rng.rand(40, 25) < 0.2)).astype(
        np.float32)
    return {
        "values": sp.csr_matrix(dense),
        "labels": np.array(["type {}".format(i % 3) for i in range(40)]),
        "example names": np.array(["cell {}".format(i) for i in range(40)]),
        "feature names": np.array(["gène {}".format(i) for i in range(25)]),
        "batch indices": None,
        "class names": ["a", "b", "c"],
        "split indices": {"training": slice(0, 30), "validation": slice(30, 35),
                          "test": [35, 36, 37, 38, 39]},
        "feature mapping": {"g1": ["id1", "id2"], "g2": ["id3"]},
        "training set": {"values": sp.csr_matrix(dense[:30]),
                         "preprocessed values": None},
    }


def _same(a, b, path=""):
    assert type(a) is type(b) or (sp.issparse(a) and sp.issparse(b)), path
    if isinstance(a, dict):
        assert list(a) == list(b) or set(a) == set(b), path
        for key in a:
            _same(a[key], b[key], path + "/" + key)
    elif sp.issparse(a):
        assert a.shape == b.shape and (a != b).nnz == 0, path
        assert a.dtype == b.dtype, path
    elif isinstance(a, np.ndarray):
        assert a.dtype.kind == b.dtype.kind and np.array_equal(a, b), path
    else:
        assert a == b, path


def test_data_dictionary_round_trip(tmp_path, capsys):
    """internal_io.py:29-122: every kind of entry the reference saves."""
    from scvae_amd.data import internal_io
    dictionary = _sample_dictionary()
    path = str(tmp_path / "deep" / "sample.sparse.h5")
    internal_io.save_data_dictionary(dictionary, path)
    loaded = internal_io.load_data_dictionary(path)
    out = capsys.readouterr().out
    assert "Data saved" in out and "Data loaded" in out
    _same(dictionary, loaded)
    # the file as PyTables names things: normalised node names, titles as attributes
    with hdf5.File(path) as f:
        assert f.root.attributes["PYTABLES_FORMAT_VERSION"] == "2.1"
        assert f.root["example_names"].attributes["TITLE"] == "example names"
        assert f.root["class_names_was_list"].attributes["CLASS"] == "ARRAY"
        assert f.root["values"].attributes["CLASS"] == "GROUP"
        assert set(f.root["values"].keys()) == {"data", "indices", "indptr",
                                                "shape"}


def test_written_file_is_what_libhdf5_was_shown(tmp_path):
    """The writer's output byte for byte: the same dictionary must produce the
    file that h5py (libhdf5) opened and compared entry by entry when the
    fixtures were made (``make_hdf5_fixtures.py`` reads
    ``hdf5_written_by_us.h5`` back against ``hdf5_written_by_us.json``)."""
    from scvae_amd.data import internal_io
    path = str(tmp_path / "ours.h5")
    internal_io.save_data_dictionary(_sample_dictionary(), path)
    golden = os.path.join(GOLDEN, "hdf5_written_by_us.h5")
    if os.environ.get("SCVAE_WRITE_HDF5_SAMPLE"):
        import shutil
        shutil.copy(path, golden)
        with hdf5.File(path) as f:
            def spec(group):
                out = {"attrs": {k: v for k, v in group.attributes.items()},
                       "members": {}}
                for key in group.keys():
                    node = group[key]
                    if isinstance(node, hdf5.Group):
                        out["members"][key] = spec(node)
                    else:
                        value = node.read()
                        if value.dtype.kind == "S":
                            value = np.char.decode(value, "utf-8")
                        out["members"][key] = {
                            "data": value.tolist(),
                            "attrs": dict(node.attributes)}
                return out
            with open(golden[:-3] + ".json", "w") as handle:
                json.dump(spec(f.root), handle, ensure_ascii=False)
    with open(path, "rb") as ours, open(golden, "rb") as shown:
        assert ours.read() == shown.read()


def test_10x_h5_loader_and_generic_sparse_h5(expected, tmp_path):
    from scvae_amd.data.loaders import LOADERS, infer_data_format
    path = os.path.join(GOLDEN, "hdf5_tenx.h5")
    assert infer_data_format(path) == "10x"
    d = LOADERS["10x"](path)
    assert d["values"].shape == (120, 300) and d["values"].dtype == np.float32
    assert np.array_equal(d["values"].toarray(), expected["tenx/dense"])
    assert d["example names"][0] == "ACGT000000000000-1"
    assert d["feature names"][299] == "GENE299" and d["genome name"] == "GRCh38"
    # the generic loader finds the name lists by the reference's guesses and
    # orients the matrix by which axis they fit (loaders.py:746-785)
    g = LOADERS["h5"](path)
    assert np.array_equal(g["values"].toarray(), expected["tenx/dense"])
    assert g["example names"][3] == d["example names"][3]
    assert g["feature names"][5] == "ENSG00000000005"      # "genes" is guessed first
    # two groups with arrays: refused as in the reference
    writer = hdf5.Writer()
    writer.root.create_group("a").create_dataset("data", np.arange(3))
    writer.root.create_group("b").create_dataset("data", np.arange(3))
    two = str(tmp_path / "two.h5")
    writer.save(two)
    with pytest.raises(NotImplementedError, match="multiple directories"):
        LOADERS["10x"](two)


def test_data_set_loads_10x_h5_and_uses_the_cache(tmp_path, monkeypatch, capsys):
    """data_set.py:749-790: a parsed file is cached as
    ``<directory>/<name>/preprocessed/<name>.sparse.h5`` when parsing took long
    enough, and loaded from there the next time."""
    import shutil
    from scvae_amd.data import DataSet
    source = str(tmp_path / "pbmc.h5")
    shutil.copy(os.path.join(GOLDEN, "hdf5_tenx.h5"), source)
    directory = str(tmp_path / "data")
    first = DataSet(source, directory=directory)
    first.load()
    cache = os.path.join(directory, "pbmc", "preprocessed", "pbmc.sparse.h5")
    assert not os.path.exists(cache)            # (parsed in well under 30 s)
    monkeypatch.setenv("SCVAE_CACHE_AFTER_SECONDS", "0")
    second = DataSet(source, directory=directory)
    second.load()
    assert os.path.isfile(cache) and "Saving data set." in capsys.readouterr().out
    third = DataSet(source, directory=directory)
    third.load()
    assert "Loading data set." in capsys.readouterr().out
    for other in (second, third):
        assert (first.values != other.values).nnz == 0
        assert np.array_equal(first.example_names, other.example_names)
        assert np.array_equal(first.feature_names, other.feature_names)
    assert first.number_of_examples == 120 and first.number_of_features == 300
