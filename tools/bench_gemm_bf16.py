"""Micro-benchmark of the bf16-weight GEMM kernels at long-batch shapes (HIP events, L2-warm).
usage: python tools/bench_gemm_bf16.py [M ...] [--a16] [--shapes NxK,NxK,...]
  --a16: A given as bf16 (the engine's bf16 activation copies): from 512 rows on this is the LDS-DMA kernel
  (M3_DMA_MIN_ROWS=<rows> moves that threshold; M3_TILED_MIN_ROWS=<rows>: huge = K-split kernel only)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3m-asr-inference_amd"))
import torch
from m3asr import ops, _lib

a16 = "--a16" in sys.argv
Ms = [int(v) for v in sys.argv[1:] if v.isdigit()] or [1984]
custom = [v.split("=", 1)[1] if "=" in v else sys.argv[i + 1] for i, v in enumerate(sys.argv) if v.startswith("--shapes")]
shapes = [("qkv", 1536, 512), ("w1", 1024, 512), ("w2", 512, 1024), ("out", 512, 512), ("w1x2", 2048, 1024), ("logits", 1434, 512)]
if custom:
    shapes = [("NxK", int(t.split("x")[0]), int(t.split("x")[1])) for t in custom[0].split(",")]
for M in Ms:
    for name, N, K in shapes:
        a = torch.randn(M, K, device="cuda")
        if a16:
            a = a.to(torch.bfloat16)
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
        best = 1e9
        for _ in range(5):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        print("%-7s M=%5d N=%4d K=%4d %s  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, "a16" if a16 else "a32", best, 2.0 * M * N * K / best * 1e-6), flush=True)
