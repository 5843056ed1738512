#!/usr/bin/env python3
"""Development probe: the MoE router product at long-batch sizes (M rows x [embed 512 | x 512] -> N experts, fp32) in its
forms: plain K = 1024, concatenated input, concatenated + LayerNorm prologue (what the engine's moe_router stage runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch
from m3asr import ops
torch.manual_seed(0)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M in (1090, 4400):
    for N in (32, 64):
        emb, x = torch.randn(M, 512, device="cuda"), torch.randn(M, 512, device="cuda")
        cat = torch.cat([emb, x], 1).contiguous()
        w, b = torch.randn(N, 1024, device="cuda") * 0.03, torch.zeros(N, device="cuda")
        g, be = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
        t_plain = timeit(lambda: ops.linear(cat, w, b))
        t_cat = timeit(lambda: ops.linear(emb, w, b, a2=x))
        t_ln = timeit(lambda: ops.linear(x, w[:, 512:].contiguous(), b, ln=(g, be, 1e-12)))
        t_ln512 = timeit(lambda: ops.layer_norm(x, g, be, 1e-12))
        print("M=%d N=%d: plain K=1024 %.1f us | concat %.1f us | LN-prologue GEMM K=512 %.1f us | layer_norm alone %.1f us" % (M, N, t_plain, t_cat, t_ln, t_ln512))
