#!/usr/bin/env python3
"""Diagnostic (not product): where a wave of the fused bf16 expert kernel spends its cycles.  Needs the -DM3_FUSED_DIAG
library variant (M3ASR_LIB=tools/_diag_libm3asr.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import numpy as np, torch
from m3asr import ops, _lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D, F, E = 512, 1024, 32
g = torch.Generator().manual_seed(0)
x = torch.randn(S, D, generator=g).cuda()
gate = (torch.randperm(S, generator=g) % E).to(torch.int32).cuda()
w1 = (torch.randn(E, F, D, generator=g) * D ** -0.5).bfloat16().cuda()
w2 = (torch.randn(E, D, F, generator=g) * F ** -0.5).bfloat16().cuda()
b1, b2 = torch.zeros(E, F).cuda(), torch.zeros(E, D).cuda()
for _ in range(3):
    ops.moe_expert_ffn(x, gate, w1, b1, w2, b2)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 4, dtype=np.uint64)
lib.m3_debug_fused_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.m3_debug_fused_read(buf.ctypes.data, buf.nbytes)
d = buf.reshape(-1, 4).astype(np.float64)
d = d[d[:, 2] > 0]
print("rc", rc, "waves", len(d))
print("cycles per wave: total median %.0f  vmcnt-wait median %.0f (%.1f%%)  barrier median %.0f (%.1f%%)" % (
    np.median(d[:, 2]), np.median(d[:, 0]), 100 * np.median(d[:, 0] / d[:, 2]), np.median(d[:, 1]), 100 * np.median(d[:, 1] / d[:, 2])))
print("total p10/p90 %.0f %.0f" % (np.percentile(d[:, 2], 10), np.percentile(d[:, 2], 90)))
