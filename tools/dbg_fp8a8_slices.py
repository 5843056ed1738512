"""Diagnostic (not product): error of the fp8-arithmetic expert FFN when only one 64-wide slice of F contributes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch, numpy as np
import torch.nn.functional as F
from m3asr import ops
from m3asr.plan import quantize_fp8_rows
S, E, D, Fh = 4096, 32, 512, 1024
g_ = torch.Generator().manual_seed(0)
x = torch.randn(S, D, generator=g_)
gate = (torch.randperm(S, generator=g_) % E).to(torch.int32)
w1 = torch.randn(E, Fh, D, generator=g_) * D ** -0.5; b1 = torch.randn(E, Fh, generator=g_) * 0.1
w2f = torch.randn(E, D, Fh, generator=g_) * Fh ** -0.5; b2 = torch.zeros(E, D)
q1, s1 = quantize_fp8_rows(w1, dims=(2,))
hs = 0.02
def q8(t): return t.float().clamp(-448, 448).to(torch.float8_e4m3fn).double()
for k in range(16):
    w2 = torch.zeros_like(w2f); w2[:, :, 64 * k:64 * k + 64] = w2f[:, :, 64 * k:64 * k + 64]
    q2, s2 = quantize_fp8_rows(w2, dims=(2,))
    y = ops.moe_expert_ffn(x.cuda(), gate.cuda(), q1.cuda(), b1.cuda(), q2.cuda(), b2.cuda(), w1_scale=s1.cuda(), w2_scale=s2.cuda(), h_scale=hs).cpu().double()
    want = torch.zeros(S, D, dtype=torch.float64)
    for e in range(E):
        rows = (gate == e).nonzero().flatten()
        xr = x[rows]; amax = xr.abs().amax(1, keepdim=True)
        xq = q8(xr * (448.0 / amax)); sx = (amax / 448.0).double()
        z = (xq @ q1[e].double().t()) * (s1[e].double() * sx) + b1[e].double()
        hq = q8(F.silu(z).float() * (1.0 / hs))
        want[rows] = (hq @ q2[e].double().t()) * (s2[e].double() * hs)
    err = float((y - want).abs().max() / want.abs().max())
    print("slice %2d (walk position %d of its work-group): max err %.3e" % (k, k % 4, err))
