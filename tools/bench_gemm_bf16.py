"""Micro-benchmark of the bf16-weight GEMM kernels at long-batch shapes (HIP events, L2-warm).
usage: python tools/bench_gemm_bf16.py [M]   (M3_TILED_MIN_ROWS=<rows> picks the kernel: huge = K-split only)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3m-asr-inference_amd"))
import torch
from m3asr import ops, _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1984
shapes = [("qkv", 1536, 512), ("w1", 1024, 512), ("w2", 512, 1024), ("out", 512, 512), ("sublin", 512, 9728), ("logits", 1434, 512)]
for name, N, K in shapes:
    a = torch.randn(M, K, device="cuda")
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    for _ in range(3):
        ops.linear(a, w, b, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    g = torch.cuda.CUDAGraph()          # a graph of `reps` launches: the Python call cost (~30 us) is not in the timing
    with torch.cuda.graph(g):
        for _ in range(reps):
            ops.linear(a, w, b, out=y)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%-7s M=%d N=%d K=%d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, us, 2.0 * M * N * K / us * 1e-6), flush=True)
