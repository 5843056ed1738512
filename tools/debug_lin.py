import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3m-asr-inference_amd"))
import torch
from m3asr import ops, _lib
from m3asr.plan import fold_layernorm
def run(M, N, K, ln, act=_lib.ACT_NONE):
    print("case", M, N, K, ln, act, flush=True)
    a = torch.randn(M, K); w = torch.randn(N, K) * K ** -0.5; b = torch.randn(N)
    f = fold_layernorm(w, b, torch.ones(K), torch.zeros(K))
    w16 = f["ln.weight"].to(torch.bfloat16)
    wsum = w16.double().sum(1).float()
    y = ops.linear(a.cuda(), w16.cuda(), b.cuda(), act=act, ln_folded=(wsum.cuda(), None, 1e-12) if ln else None)
    torch.cuda.synchronize()
    print("  ok", float(y.abs().max()), flush=True)
run(100, 64, 64, True)
run(100, 64, 32, False)
run(100, 128, 32, True)
run(100, 64, 32, True)
run(100, 64, 32, True, _lib.ACT_SILU)
