"""Debug (build with -DM3_XQ_SELFTEST): the fused fp8 operator with in-kernel quantisation vs the XQ form on rows quantised by a
stand-alone kernel -- same process, M3_XQ_SELFTEST_ON toggled between the calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch
from m3asr import ops
from m3asr.plan import quantize_fp8_rows
for S in (8192, 65536):
    D, F, E = 512, 1024, 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(S, D, generator=g).cuda()
    gate = (torch.randperm(S, generator=g) % E).to(torch.int32).cuda()
    w1 = torch.randn(E, F, D, generator=g) * D ** -0.5
    w2 = torch.randn(E, D, F, generator=g) * F ** -0.5
    b1, b2 = (torch.randn(E, F, generator=g) * 0.1).cuda(), (torch.randn(E, D, generator=g) * 0.1).cuda()
    q1, s1 = quantize_fp8_rows(w1, dims=(2,)); q2, s2 = quantize_fp8_rows(w2, dims=(2,))
    a = [t.cuda() for t in (q1, s1, q2, s2)]
    fn = lambda: ops.moe_expert_ffn(x, gate, a[0], b1, a[2], b2, w1_scale=a[1], w2_scale=a[3], h_scale=0.05)
    os.environ.pop("M3_XQ_SELFTEST_ON", None)
    y0 = fn().clone(); y0b = fn().clone()
    os.environ["M3_XQ_SELFTEST_ON"] = "1"
    for var in (1, 2, 3):
        os.environ["M3_XQ_VAR"] = str(var)
        yv = fn().clone()
        print("   variant %d (bit0: image from fp32 rows, bit1: sx from fp32 rows): max |diff| to in-kernel %.3e" % (var, float((yv - y0).abs().max())))
    os.environ["M3_XQ_VAR"] = "0"
    y1 = fn().clone(); y1b = fn().clone()
    torch.cuda.synchronize()
    # fp64 evaluation of the quantised computation on a sample of rows
    idx = torch.arange(0, S, S // 64)
    xe, ge = x[idx].cpu().double(), gate[idx].cpu().long()
    amax = xe.abs().amax(dim=1, keepdim=True)
    sx = amax / 448.0
    xqv = (xe / sx).float().to(torch.float8_e4m3fn).double()
    W1 = q1.view(torch.float8_e4m3fn).double() if q1.dtype == torch.uint8 else q1.double()
    W2 = q2.view(torch.float8_e4m3fn).double() if q2.dtype == torch.uint8 else q2.double()
    z = torch.einsum("sd,sfd->sf", xqv, W1[ge]) * s1[ge].double() * sx + b1.cpu().double()[ge]
    hh = torch.nn.functional.silu(z)
    hq = (hh / 0.05).clamp(-448, 448).float().to(torch.float8_e4m3fn).double()
    want = torch.einsum("sf,sdf->sd", hq, W2[ge]) * s2[ge].double() * 0.05 + b2.cpu().double()[ge]
    e0 = float((y0[idx].cpu().double() - want).abs().max()); e1 = float((y1[idx].cpu().double() - want).abs().max())
    print("   vs fp64 evaluation on 64 rows: in-kernel %.3e, XQ %.3e (max |want| %.3e)" % (e0, e1, float(want.abs().max())))
    d = (y0 - y1).abs()
    rows0 = (gate == 0).nonzero().view(-1)          # expert 0's rows in stable order = its tile rows
    e0m = d[rows0][:128].cpu()
    print("   expert 0 tile: error by column block of 32:", [round(float(e0m[:, 32 * b:32 * b + 32].max()), 3) for b in range(16)])
    print("   error by token group of 8:", [round(float(e0m[8 * t:8 * t + 8].max()), 3) for t in range(16)])
    print("   error by column mod 32:", [round(float(e0m[:, c::32].max()), 3) for c in range(32)])
    print("S=%d: in-kernel vs XQ max |diff| %.3e (max |y| %.3e), rows differing %d; repeat in-kernel %.3e, repeat XQ %.3e" %
          (S, float(d.max()), float(y0.abs().max()), int((d.max(dim=1).values > 0).sum()), float((y0 - y0b).abs().max()), float((y1 - y1b).abs().max())))
