#!/usr/bin/env python3
"""An epoch of VariationalAutoencoder.train (bench.py's model_train_epoch) with the epoch-end
passes one step per minibatch (the reference's loop) and in steps of several minibatches:
python tools/bench_epoch.py [--batch 100] [--epochs 5]
--state: the checkpoint's state copied to the host blocking / non-blocking instead"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[100])
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--state", action="store_true")
    args = ap.parse_args()
    import bench
    from scvae_amd.minibatch import synthetic_count_matrix
    from scvae_amd.models.base import ModelBase as Model
    dev = torch.device("cuda:0")
    matrix, _ = synthetic_count_matrix(bench.N_CELLS, bench.N_FEATURES, density=0.05, seed=60,
                                       device=dev)
    default = Model.evaluation_chunk_cells
    if args.state:        # blocking against non-blocking copy of the checkpoint's state
        for batch in args.batch:
            for nb in (False, True, False, True):
                Model.checkpoint_non_blocking = nb
                r = bench.model_train_epoch(matrix, dev, batch, args.epochs)
                print("B = {:5d}, state copy {}: {:.4f} s per epoch = {:.0f} training cells/s"
                      .format(batch, "non-blocking" if nb else "blocking    ",
                              r["seconds_per_epoch"], r["value"]), flush=True)
        return
    for batch in args.batch:
        for cells in (0, default, 0, default):
            Model.evaluation_chunk_cells = cells
            r = bench.model_train_epoch(matrix, dev, batch, args.epochs)
            print("B = {:5d}, evaluation steps of <= {:4d} cells: {:.4f} s per epoch = {:.0f} "
                  "training cells/s".format(batch, max(cells, batch), r["seconds_per_epoch"],
                                            r["value"]), flush=True)


if __name__ == "__main__":
    main()
