#!/usr/bin/env python3
"""Development probe: a B=1-sized skinny GEMM (50 rows) with L2-hot weights (same matrix every launch) against cold weights
(rotating through 2 GB of matrices): the most a next-layer weight prefetch could buy per launch.  hipGraph of 64 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch
from m3asr import ops
torch.manual_seed(0)
M = 50
for (K, N) in ((512, 2048), (2048, 512), (512, 512), (512, 1536)):
    nW = max(2, int(2e9 // (K * N * 4)))
    nW = min(nW, 512)
    Ws = [torch.randn(N, K, device="cuda") * 0.02 for _ in range(nW)]
    b = torch.zeros(N, device="cuda")
    a = torch.randn(M, K, device="cuda")
    res = {}
    for mode in ("hot", "cold"):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(3): ops.linear(a, Ws[i % nW], b)
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(64):
                    ops.linear(a, Ws[0] if mode == "hot" else Ws[(7 * i + 3) % nW], b)
            for _ in range(3): g.replay()
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(20): g.replay()
            e1.record(s); s.synchronize()
            res[mode] = e0.elapsed_time(e1) * 1e3 / (20 * 64)
    print("M=%d K=%d N=%d (%.1f MB of weights): hot %.2f us / launch, cold %.2f us / launch" % (M, K, N, K * N * 4 / 1e6, res["hot"], res["cold"]))
