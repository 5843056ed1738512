#!/usr/bin/env python3
"""Micro-timings of single C-ABI ops with HIP events (development tool, GPU box only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch
from m3asr import ops, _lib

def timeit(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n // 10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n // 10 * 10) * 1e3  # us per call

def main():
    dev = "cuda"
    M = 50
    x = torch.randn(M, 512, device=dev)
    res = torch.randn(M, 512, device=dev)
    # cold-ish: rotate through many weight copies so each call streams from HBM
    for (N, K, name) in [(512, 512, "512x512"), (1024, 512, "1024x512"), (512, 1024, "512x1024"), (1536, 512, "1536x512")]:
        a = torch.randn(M, K, device=dev)
        ws = [torch.randn(N, K, device=dev) for _ in range(64)]
        b = torch.randn(N, device=dev)
        g, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        out = torch.empty(M, N, device=dev)
        i = [0]
        def cold():
            ops.linear(a, ws[i[0] % 64], b, out=out); i[0] += 1
        def warm():
            ops.linear(a, ws[0], b, out=out)
        def cold_ln():
            ops.linear(a, ws[i[0] % 64], b, ln=(g, be, 1e-12), out=out); i[0] += 1
        # graph capture of 10 calls: use distinct weights inside the graph
        def cold10():
            pass
        print("gemm M=50 %-9s  warm %.2f us   warm+LN %.2f us" % (name, timeit(warm), timeit(lambda: ops.linear(a, ws[0], b, ln=(g, be, 1e-12), out=out) if K <= 1024 else None)))
        # cold inside graph: 10 different weights per graph replay, 64 total -> mostly L2-cold but MALL-warm (64 x 1-3 MB)
        print("                       rot64 %.2f us   rot64+LN %.2f us" % (timeit(cold), timeit(cold_ln)))
    # trivial kernel floor
    l = torch.tensor([206], dtype=torch.int32, device=dev)
    print("trivial kernel (mask_conv2d_sample) %.2f us" % timeit(lambda: ops.mask_conv2d_sample(l, 2, 2)))
    y = torch.randn(M, 512, device=dev); gm = torch.ones(512, device=dev); bt = torch.zeros(512, device=dev)
    print("layernorm 50x512 %.2f us" % timeit(lambda: ops.layer_norm(y, gm, bt, 1e-12)))

if __name__ == "__main__":
    main()
