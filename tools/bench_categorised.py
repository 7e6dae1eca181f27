#!/usr/bin/env python3
"""Training step of the headline model with the piecewise categorical likelihood (-k 1 / 2):
fused (two launches of the bf16x9 head kernel) against the unfused kernels.
Usage: python tools/bench_categorised.py [--batch 4096] [--k 1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--features", type=int, default=32738)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import synthetic_count_matrix
    dev = torch.device("cuda:0")
    matrix, _ = synthetic_count_matrix(8192, args.features, density=0.05, seed=60, device=dev)
    B = args.batch
    rows = torch.arange(B, device=dev)
    rc = torch.empty(B, device=dev)
    x = matrix.gather_dense(rows, row_const_out=rc)
    eps = torch.randn(1, B, 25, device=dev)
    for fused in (True, False):
        eng = Engine(args.features, 25, (100, 100), "negative binomial", batch_norm=True,
                     device=dev, seed=0, k_max=args.k)
        eng.set_fused(fused)
        eng.reserve(B, 1)
        for _ in range(3):
            eng.step(x, x, eps=eps, row_const=rc, training=True, x_counts=True)
            eng.adam_step(1e-4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(args.steps):
            eng.step(x, x, eps=eps, row_const=rc, training=True, x_counts=True)
            eng.adam_step(1e-4)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        print("-k {} training step, {} cells x {} genes, NB, {}: {:.3f} ms = {:.0f} cells/s "
              "(fused_categorised = {})".format(args.k, B, args.features,
                                               "fused" if fused else "unfused", ms,
                                               B / ms * 1e3, eng.fused_categorised))
        for _ in range(3):
            eng.step(x, x, eps=eps, row_const=rc, training=False, x_counts=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            eng.step(x, x, eps=eps, row_const=rc, training=False, x_counts=True)
        e1.record()
        torch.cuda.synchronize()
        print("   evaluation step: {:.3f} ms".format(e0.elapsed_time(e1) / args.steps))
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
