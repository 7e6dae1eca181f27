#!/usr/bin/env python3
"""Where an epoch of VariationalAutoencoder.train goes on the host (cProfile, cumulative):
python tools/profile_train.py [--batch 4096] [--epochs 4]"""
import argparse, contextlib, cProfile, io, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, scipy.sparse, torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=4)
    args = ap.parse_args()
    from scvae_amd.data import DataSet
    from scvae_amd.minibatch import synthetic_count_matrix
    from scvae_amd.models import VariationalAutoencoder
    dev = "cuda:0"
    matrix, _ = synthetic_count_matrix(68579, 32738, density=0.05, seed=60, device=dev)
    n, F = matrix.shape
    host = scipy.sparse.csr_matrix((matrix.values.cpu().numpy(), matrix.indices.cpu().numpy(),
                                    matrix.indptr.cpu().numpy()), shape=(n, F))
    nv = n // 10
    names = numpy.array(["g%d" % j for j in range(F)])
    mk = lambda v, kind, first: DataSet("bench_shaped", values=v, kind=kind, feature_names=names,
                                        example_names=numpy.array(
                                            ["c%d" % i for i in range(first, first + v.shape[0])]))
    training, validation = mk(host[:n - nv], "training", 0), mk(host[n - nv:], "validation", n - nv)
    with tempfile.TemporaryDirectory() as d:
        model = VariationalAutoencoder(feature_size=F, latent_size=25, hidden_sizes=[100, 100],
                                       reconstruction_distribution="negative binomial",
                                       log_directory=d, device=dev)
        with open(os.devnull, "w") as sink, contextlib.redirect_stdout(sink):
            model.train(training, validation, number_of_epochs=1, minibatch_size=args.batch,
                        learning_rate=1e-4)          # warm-up: allocations, first launches
            prof = cProfile.Profile()
            prof.enable()
            model.train(training, validation, number_of_epochs=1 + args.epochs,
                        minibatch_size=args.batch, learning_rate=1e-4)
            torch.cuda.synchronize()
            prof.disable()
    out = io.StringIO()
    pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(45)
    print(out.getvalue())


if __name__ == "__main__":
    main()
