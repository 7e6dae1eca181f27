#!/usr/bin/env python3
"""An epoch of VariationalAutoencoder.train (bench.py's model_train_epoch) with the epoch-end
passes one step per minibatch (the reference's loop) and in steps of several minibatches:
python tools/bench_epoch.py [--batch 100] [--epochs 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[100])
    ap.add_argument("--epochs", type=int, default=5)
    args = ap.parse_args()
    import bench
    from scvae_amd.minibatch import synthetic_count_matrix
    from scvae_amd.models.base import ModelBase as Model
    dev = torch.device("cuda:0")
    matrix, _ = synthetic_count_matrix(bench.N_CELLS, bench.N_FEATURES, density=0.05, seed=60,
                                       device=dev)
    default = Model.evaluation_chunk_cells
    for batch in args.batch:
        for cells in (0, default, 0, default):
            Model.evaluation_chunk_cells = cells
            r = bench.model_train_epoch(matrix, dev, batch, args.epochs)
            print("B = {:5d}, evaluation steps of <= {:4d} cells: {:.4f} s per epoch = {:.0f} "
                  "training cells/s".format(batch, max(cells, batch), r["seconds_per_epoch"],
                                            r["value"]), flush=True)


if __name__ == "__main__":
    main()
