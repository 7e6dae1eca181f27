#!/usr/bin/env python3
"""Throughput of ``model.train`` itself (the drop-in surface: Python loop, CSR gather, noise,
step, Adam, epoch-end evaluation) on a bench-shaped synthetic data set.
Usage: python tools/bench_model_train.py [--cells 32768] [--batch 4096] [--epochs 3]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=32768)
    ap.add_argument("--features", type=int, default=32738)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=3)
    args = ap.parse_args()
    from scvae_amd.data import DataSet
    from scvae_amd.minibatch import synthetic_count_matrix
    from scvae_amd.models import VariationalAutoencoder
    matrix, _ = synthetic_count_matrix(args.cells, args.features, density=0.05,
                                       seed=60, device="cuda:0")
    import scipy.sparse
    values = scipy.sparse.csr_matrix(
        (matrix.values.cpu().numpy(), matrix.indices.cpu().numpy(),
         matrix.indptr.cpu().numpy()), shape=matrix.shape)
    data = DataSet("bench_shaped", values=values,
                   example_names=np.array(["c%d" % i for i in range(args.cells)]),
                   feature_names=np.array(["g%d" % j for j in range(args.features)]))
    with tempfile.TemporaryDirectory() as directory:
        model = VariationalAutoencoder(
            feature_size=args.features, latent_size=25, hidden_sizes=[100, 100],
            reconstruction_distribution="negative binomial",
            log_directory=directory, device="cuda:0")
        t0 = time.perf_counter()
        model.train(data, None, number_of_epochs=args.epochs,
                    minibatch_size=args.batch, learning_rate=1e-4)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    print("model.train: {} epochs of {} cells in {:.2f} s -> {:.0f} cells/s "
          "including the epoch-end evaluation of the training set and checkpoints"
          .format(args.epochs, args.cells, total,
                  args.epochs * args.cells / total))


if __name__ == "__main__":
    main()
