"""Diagnostic (not product): where the fp8-arithmetic expert FFN differs from the fp64 evaluation -- by position of the row in
its expert's tile (wave, token) and by output column block."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch, numpy as np
import torch.nn.functional as F
from m3asr import ops
from m3asr.plan import quantize_fp8_rows
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
E, D, Fh = 32, 512, 1024
g_ = torch.Generator().manual_seed(0)
x = torch.randn(S, D, generator=g_)
gate = (torch.randperm(S, generator=g_) % E).to(torch.int32)
w1 = torch.randn(E, Fh, D, generator=g_) * D ** -0.5; b1 = torch.randn(E, Fh, generator=g_) * 0.1
w2 = torch.randn(E, D, Fh, generator=g_) * Fh ** -0.5; b2 = torch.zeros(E, D)
q1, s1 = quantize_fp8_rows(w1, dims=(2,)); q2, s2 = quantize_fp8_rows(w2, dims=(2,))
hs = 0.02
def q8(t): return t.float().clamp(-448, 448).to(torch.float8_e4m3fn).double()
y = ops.moe_expert_ffn(x.cuda(), gate.cuda(), q1.cuda(), b1.cuda(), q2.cuda(), b2.cuda(), w1_scale=s1.cuda(), w2_scale=s2.cuda(), h_scale=hs).cpu().double()
want = torch.zeros(S, D, dtype=torch.float64)
posn = torch.zeros(S, dtype=torch.long)
for e in range(E):
    rows = (gate == e).nonzero().flatten()
    posn[rows] = torch.arange(rows.numel())
    xr = x[rows]; amax = xr.abs().amax(1, keepdim=True)
    xq = q8(xr * (448.0 / amax)); sx = (amax / 448.0).double()
    z = (xq @ q1[e].double().t()) * (s1[e].double() * sx) + b1[e].double()
    hq = q8(F.silu(z).float() * (1.0 / hs))
    want[rows] = (hq @ q2[e].double().t()) * (s2[e].double() * hs)
err = (y - want).abs() / float(want.abs().max())
rowerr = err.amax(1).numpy()
bad = rowerr > 4e-3
print("rows bad: %d of %d" % (bad.sum(), S))
tok = (posn % 128).numpy(); tile = (posn // 128).numpy()
print("bad by wave  :", [int(bad[(tok // 32) == w].sum()) for w in range(4)])
print("bad by tok%32 :", [int(bad[(tok % 32) == t].sum()) for t in range(32)])
print("bad by tile   :", [int(bad[tile == t].sum()) for t in range(int(tile.max()) + 1)])
print("bad by expert :", [int(bad[(gate == e).numpy()].sum()) for e in range(E)])
colerr = err[torch.from_numpy(bad)].amax(0).numpy() if bad.any() else np.zeros(D)
print("max err per 32-col block (bad rows):", np.round(colerr.reshape(16, 32).max(1), 4))
print("#cols > 4e-3 per bad row (quantiles):", np.quantile((err[torch.from_numpy(bad)] > 4e-3).sum(1).numpy(), [0, .5, 1]) if bad.any() else None)
