"""Development probe: what the epilogue forms of the tiled bf16 GEMM cost at a long-batch shape (graph-timed, L2-warm).
usage: python tools/bench_gemm_epilogue.py [M]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3m-asr-inference_amd"))
import torch
from m3asr import ops, _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1984
B, T = 16, M // 16
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


for name, N, K in [("out 512x512", 512, 512), ("w2 512x1024", 512, 1024), ("w1 1024x512", 1024, 512), ("qkv 1536x512", 1536, 512)]:
    a = torch.randn(M, K, device="cuda")
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    wsum = w.float().sum(1).contiguous()
    y = torch.empty(M, N, device="cuda")
    forms = {
        "plain": lambda: ops.linear(a, w, None, out=y),
        "bias": lambda: ops.linear(a, w, b, out=y),
        "bias+resid": lambda: ops.linear(a, w, b, resid=r, alpha=0.5, out=y),
        "bias+resid+mask": lambda: ops.linear(a, w, b, resid=r, alpha=0.5, lens=lens, rows_per_batch=T, mask_out=True, out=y),
        "ln+bias+silu": lambda: ops.linear(a, w, b, act=_lib.ACT_SILU, ln_folded=(wsum, None, 1e-12), out=y),
    }
    print(name + "  M=%d: " % M + "  ".join("%s %.1f us" % (k, timed(f)) for k, f in forms.items()), flush=True)
