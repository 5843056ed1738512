#!/usr/bin/env python3
"""Diagnostic (not product): cycle shares of a wave of the fused fp8 expert kernel (needs the -DM3_FUSED_DIAG library)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import numpy as np, torch
from m3asr import ops, _lib
from m3asr.plan import quantize_fp8_rows
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D, F, E = 512, 1024, 32
g = torch.Generator().manual_seed(0)
x = torch.randn(S, D, generator=g).cuda()
gate = (torch.randperm(S, generator=g) % E).to(torch.int32).cuda()
q1, s1 = quantize_fp8_rows(torch.randn(E, F, D, generator=g) * D ** -0.5, dims=(2,))
q2, s2 = quantize_fp8_rows(torch.randn(E, D, F, generator=g) * F ** -0.5, dims=(2,))
b1, b2 = torch.zeros(E, F).cuda(), torch.zeros(E, D).cuda()
a = [q1.cuda(), b1, q2.cuda(), b2]
for _ in range(3):
    ops.moe_expert_ffn(x, gate, *a, w1_scale=s1.cuda(), w2_scale=s2.cuda(), h_scale=0.02)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.m3_debug_fused8_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.m3_debug_fused8_read(buf.ctypes.data, buf.nbytes)
d = buf.reshape(-1, 8).astype(np.float64)
d = d[d[:, 7] > 0]
names = ["barrier", "epilogue (Y out)", "tile end (bias, X flush, xq)", "gemm1+silu+quant", "gemm2", "piece loops", "prologue"]
tot = np.median(d[:, 7])
print("waves %d; whole kernel per wave median %.0f cycles (p90 %.0f); " % (len(d), tot, np.percentile(d[:, 7], 90)))
print("  " + ", ".join("%s %.0f (%.0f%%)" % (n, np.median(d[:, i]), 100 * np.median(d[:, i]) / tot) for i, n in enumerate(names)))
