"""Built-in synthetic count data sets.

``development``: restatement of the reference's own self-check data set
(``scvae/data/loaders.py:585-592, 942-1022``): 10 000 x 25 negative-binomial
counts with per-gene Bernoulli dropout drawn from ``numpy.random.RandomState(60)``
in the reference's draw order, so the matrix is reproducible (legacy
``RandomState`` streams are stable across NumPy versions).

``synthetic_<cells>x<genes>``-style named shapes (BASELINE.json configs) are
produced by a vectorised generator of the same family.
"""

import numpy
import scipy.sparse


def create_development_data_set(n_examples=10000, n_features=25, scale=10,
                                update_probability=0.0001):
    random_state = numpy.random.RandomState(60)
    r = numpy.empty((n_examples, n_features))
    p = numpy.empty((n_examples, n_features))
    dropout = numpy.empty((n_examples, n_features))
    labels = numpy.empty(n_examples, numpy.int32)

    def draw():
        # order of draws: r, then p, then dropout (one vector of n_features each)
        return (scale * random_state.rand(n_features),
                random_state.rand(n_features),
                random_state.rand(n_features))

    r_type, p_type, dropout_type = draw()
    label = 1
    for i in range(n_examples):
        if random_state.rand() > 1 - update_probability:
            r_type, p_type, dropout_type = draw()
            label += 1
        r[i], p[i], dropout[i], labels[i] = r_type, p_type, dropout_type, label

    shuffled = random_state.permutation(n_examples)
    r, p, dropout, labels = (r[shuffled], p[shuffled], dropout[shuffled],
                             labels[shuffled])
    no_class = random_state.permutation(n_examples)[:int(0.1 * n_examples)]
    labels[no_class] = 0
    labels = labels.astype(str)

    values = numpy.empty((n_examples, n_features), numpy.float32)
    for i in range(n_examples):
        for j in range(n_features):
            value = random_state.negative_binomial(r[i, j], p[i, j])
            keep = random_state.binomial(1, dropout[i, j])
            values[i, j] = keep * value

    feature_ids = numpy.array(
        ["feature {}".format(j + 1) for j in range(n_features)])
    return {
        "values": values,
        "labels": labels,
        "example names": numpy.array(
            ["example {}".format(i + 1) for i in range(n_examples)]),
        "feature names": feature_ids,
        # feature ids grouped five at a time under "feature A" .. "feature E"
        "feature mapping": {
            "feature " + letter: group.tolist() for letter, group in
            zip("ABCDE", numpy.split(feature_ids, 5))},
    }


def create_count_data_set(n_examples, n_features, density=0.05, n_clusters=8,
                          seed=60, chunk=4096):
    """Vectorised NB x Bernoulli generator (same family as ``development``)
    rescaled to a target nonzero fraction; returns a CSR matrix."""
    rng = numpy.random.RandomState(seed)
    r = 10.0 * rng.rand(n_clusters, n_features) + 1e-3
    p = 0.05 + 0.95 * rng.rand(n_clusters, n_features)
    keep = rng.rand(n_clusters, n_features)
    p_nonzero = (keep * (1.0 - p ** r)).mean()
    keep = numpy.minimum(keep * density / max(p_nonzero, 1e-12), 1.0)
    labels = rng.randint(0, n_clusters, size=n_examples)
    blocks = []
    for start in range(0, n_examples, chunk):
        lab = labels[start:start + chunk]
        counts = rng.negative_binomial(r[lab], p[lab])
        counts = counts * (rng.rand(*counts.shape) < keep[lab])
        blocks.append(scipy.sparse.csr_matrix(counts.astype(numpy.float32)))
    values = scipy.sparse.vstack(blocks).tocsr()
    return {
        "values": values,
        "labels": numpy.array(["cluster {}".format(k) for k in labels]),
        "example names": numpy.array(
            ["cell {}".format(i + 1) for i in range(n_examples)]),
        "feature names": numpy.array(
            ["gene {}".format(j + 1) for j in range(n_features)]),
    }


def _shape(n, f, **kw):
    return lambda: create_count_data_set(n, f, **kw)


#: name -> generator; shapes of the BASELINE.json configs
SYNTHETIC_DATA_SETS = {
    "development": create_development_data_set,
    "synthetic_1k": _shape(1000, 100, density=0.3),
    "synthetic_pbmc_3k": _shape(2700, 32738),
    "synthetic_pbmc_68k": _shape(68579, 32738),
    "synthetic_mbc_1m": _shape(1306127, 27998),
}
