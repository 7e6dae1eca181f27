#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels on the headline shapes (HIP events on the launch stream).
Usage: python tools/bench_kernels.py [--rows 4096] [--kind 1]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scvae_amd import _lib  # noqa: E402


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--features", type=int, default=32738)
    ap.add_argument("--hidden", type=int, default=100)
    ap.add_argument("--kind", type=int, default=1)
    ap.add_argument("--density", type=float, default=0.05)
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    R, F, H = args.rows, args.features, args.hidden
    P = {0: 1, 1: 2, 2: 2, 3: 3}[args.kind]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(0)
    d = torch.relu(torch.randn(R, H, device=dev, generator=g))
    W = [torch.randn(H, F, device=dev, generator=g) * 0.05 for _ in range(P)]
    b = [torch.randn(F, device=dev, generator=g) * 0.1 for _ in range(P)]
    dW = [torch.empty_like(w) for w in W]
    db = [torch.empty_like(x) for x in b]
    t = torch.poisson(torch.full((R, F), 2.0, device=dev), generator=g)
    t = t * (torch.rand(R, F, device=dev, generator=g) < args.density)
    gw = torch.full((R,), -1.0 / R, device=dev)
    rc = torch.lgamma(t + 1).sum(dim=1)
    ll = torch.empty(R, device=dev)
    dd = torch.empty(R, H, device=dev)
    ws_bytes = lib.scvae_decoder_fused_workspace_bytes(R, H, F)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def arr(ts):
        return (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
    aW, ab, adW, adb = arr(W), arr(b), arr(dW), arr(db)

    def fused(train):
        _lib.check(lib.scvae_decoder_fused(
            args.kind, train, d.data_ptr(), R, H, aW, ab, adW, adb, F,
            t.data_ptr(), R, gw.data_ptr(), rc.data_ptr(), ll.data_ptr(),
            dd.data_ptr(), ws.data_ptr(), stream), "fused")
    us = timeit(lambda: fused(1))
    flops = 2.0 * R * F * P * 3 * H
    print("decoder_fused train  : {:9.1f} us  {:6.1f} TFLOP/s algorithmic "
          "({:.1f}% of 157.3)".format(us, flops / us / 1e6,
                                      flops / us / 1e6 / 1.573))
    us = timeit(lambda: fused(0))
    flops = 2.0 * R * F * P * H
    print("decoder_fused forward: {:9.1f} us  {:6.1f} TFLOP/s".format(
        us, flops / us / 1e6))

    x = t
    W1 = torch.randn(F, H, device=dev, generator=g) * 0.01
    b1 = torch.zeros(H, device=dev)
    y = torch.empty(R, H, device=dev)
    gws_bytes = lib.scvae_gemm_workspace_bytes(R, H, F)
    gws = torch.empty(max(gws_bytes, 4), dtype=torch.uint8, device=dev)
    us = timeit(lambda: _lib.check(lib.scvae_gemm(
        0, 0, x.data_ptr(), W1.data_ptr(), b1.data_ptr(), y.data_ptr(), R, H, F,
        F, H, H, 0, 0, gws.data_ptr(), gws_bytes, stream), "gemm"))
    print("encoder-1 forward    : {:9.1f} us  {:6.1f} TFLOP/s".format(
        us, 2.0 * R * F * H / us / 1e6))
    dy = torch.randn(R, H, device=dev, generator=g)
    dW1 = torch.empty(F, H, device=dev)
    dws_bytes = lib.scvae_gemm_workspace_bytes(F, H, R)
    dws = torch.empty(max(dws_bytes, 4), dtype=torch.uint8, device=dev)
    us = timeit(lambda: _lib.check(lib.scvae_gemm(
        1, 0, x.data_ptr(), dy.data_ptr(), None, dW1.data_ptr(), F, H, R, F, H,
        H, 0, 0, dws.data_ptr(), dws_bytes, stream), "gemm"))
    print("encoder-1 dW         : {:9.1f} us  {:6.1f} TFLOP/s".format(
        us, 2.0 * R * F * H / us / 1e6))

    # the same two products on the exact bf16-split kernels (count_gemm.hip): HBM-bound
    # reads of x, priced against the 8 TB/s roof
    for mode, other, out, label in ((0, W1, y, "forward"), (1, dy, dW1, "dW     ")):
        cbytes = lib.scvae_count_gemm_workspace_bytes(mode, R, F, H)
        cws = torch.empty(cbytes + 16, dtype=torch.uint8, device=dev)
        bias = b1.data_ptr() if mode == 0 else None
        us = timeit(lambda: _lib.check(lib.scvae_count_gemm(
            mode, x.data_ptr(), F, R, F, other.data_ptr(), H, H, bias, 0,
            out.data_ptr(), H, cws.data_ptr(), cbytes, stream), "count_gemm"))
        print("encoder-1 {} bf16x3-exact: {:9.1f} us  {:6.2f} TB/s of x "
              "({:.1f}% of 8 TB/s)".format(label, us, 4.0 * R * F / us / 1e6,
                                           4.0 * R * F / us / 1e6 / 8.0 * 100))


if __name__ == "__main__":
    main()
