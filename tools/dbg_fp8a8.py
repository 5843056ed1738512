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
w2 = torch.randn(E, D, Fh, generator=g_) * Fh ** -0.5; b2 = torch.zeros(E, D)
q1, s1 = quantize_fp8_rows(w1, dims=(2,)); q2, s2 = quantize_fp8_rows(w2, dims=(2,))
hs = 0.02
def q8(t): return t.float().clamp(-448, 448).to(torch.float8_e4m3fn).double()
for variant in ("full", "b1=0", "s=1"):
    bb1 = torch.zeros_like(b1) if variant == "b1=0" else b1
    ss1 = torch.ones_like(s1) if variant == "s=1" else s1
    ss2 = torch.ones_like(s2) if variant == "s=1" else s2
    y = ops.moe_expert_ffn(x.cuda(), gate.cuda(), q1.cuda(), bb1.cuda(), q2.cuda(), b2.cuda(), w1_scale=ss1.cuda(), w2_scale=ss2.cuda(), h_scale=hs).cpu().double()
    want = torch.zeros(S, D, dtype=torch.float64)
    for e in range(E):
        rows = (gate == e).nonzero().flatten()
        xr = x[rows]; amax = xr.abs().amax(1, keepdim=True)
        xq = q8(xr * (448.0 / amax)); sx = (amax / 448.0).double()
        z = (xq @ q1[e].double().t()) * (ss1[e].double() * sx) + bb1[e].double()
        hq = q8(F.silu(z).float() * (1.0 / hs))
        want[rows] = (hq @ q2[e].double().t()) * (ss2[e].double() * hs)
    err = (y - want).abs()
    print(variant, "max err %.3e scale %.3e; per-row max err quantiles" % (float(err.max()), float(want.abs().max())),
          np.quantile(err.amax(1).numpy(), [0.5, 0.9, 0.99, 1.0]).round(5), "per-col", np.quantile(err.amax(0).numpy(), [0.5, 0.99, 1.0]).round(5))
    # saturation statistics
    hmax = max(float(F.silu(x[(gate == e).nonzero().flatten()] @ w1[e].t() + bb1[e]).abs().max()) for e in range(E))
    print("   max |H| %.3f, representable up to %.3f" % (hmax, 448 * hs))
