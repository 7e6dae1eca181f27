#!/usr/bin/env python3
"""Time the EVALUATION step (is_training = False: the epoch-end passes of model.train and
model.evaluate, va:1092-1150 / 1969-2055) of the headline model on bench-shaped synthetic counts:
minibatch fetch + one graph execution per step, HIP events around the loop.
Usage: python tools/bench_eval.py [--batch 4096] [--steps 100] [--likelihood "negative binomial"]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=16384)
    ap.add_argument("--features", type=int, default=32738)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--latent", type=int, default=25)
    ap.add_argument("--likelihood", default="negative binomial")
    ap.add_argument("--unfused", action="store_true",
                    help="heads as GEMM + element-wise likelihood kernels (A/B)")
    ap.add_argument("--hidden", type=int, nargs="+", default=[100, 100],
                    help="hidden layer sizes (encoder order; the decoder's last layer -- the "
                         "heads' input -- is the first)")
    args = ap.parse_args()
    from scvae_amd.engine import Engine
    from scvae_amd.minibatch import synthetic_count_matrix
    dev = torch.device("cuda:0")
    matrix, _ = synthetic_count_matrix(args.cells, args.features, density=0.05, seed=60,
                                       device=dev)
    B = args.batch
    eng = Engine(args.features, args.latent, tuple(args.hidden), args.likelihood, batch_norm=True,
                 device=dev, seed=0)
    eng.reserve(B, 1)
    if args.unfused:
        eng.set_fused(False)
    u16 = matrix.integer_counts and eng.accepts_counts_u16(B, False)
    carry = os.environ.get("BENCH_EVAL_CARRY", "1") != "0"
    x = [(torch.empty(B, matrix.u16_pitch, dtype=torch.uint16, device=dev) if u16
          else torch.empty(B, args.features, device=dev)) for _ in range(2)]
    rc = [torch.empty(B, device=dev) for _ in range(2)]
    eps = torch.randn(1, B, args.latent, device=dev)
    rows = torch.arange(args.cells, device=dev)

    def request(i):
        r = rows[(i * B) % (args.cells - B + 1):][:B]
        return matrix.request(r, x[i & 1], rc[i & 1])
    request(0).issue()

    def step(i):
        # (BENCH_EVAL_CARRY=0: the fetch in line in front of the step, as before round 6)
        if not carry and i > 0:
            request(i).issue()
        eng.step(x[i & 1], x[i & 1], eps=eps, row_const=rc[i & 1], training=False,
                 x_counts=matrix.integer_counts,
                 next_minibatch=request(i + 1) if carry else None)
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for i in range(args.steps):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    print("evaluation step, {} cells x {} genes, {} H = {} ({} minibatch{}): {:.3f} ms = {:.0f} cells/s"
          .format(B, args.features, args.likelihood, args.hidden, "uint16" if u16 else "fp32",
                  ", unfused heads" if args.unfused else "", ms,
                  B / ms * 1e3))


if __name__ == "__main__":
    main()
