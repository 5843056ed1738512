#!/usr/bin/env python3
"""Development probe: grouped bf16 expert FFN at a saturating size (S tokens, balanced routing) -- whole-op time under a
hipGraph and a correctness check against fp32 torch.  Used with M3ASR_LIB=<variant .so> and rocprofv3 --kernel-trace --stats."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch
from m3asr import ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D, F, E = 512, int(os.environ.get("EXP_F", "1024")), int(os.environ.get("EXP_E", "32"))
g = torch.Generator().manual_seed(0)
dev = "cuda"
x = torch.randn(S, D, generator=g).to(dev)
gate = (torch.randperm(S, generator=g) % E).to(torch.int32).to(dev)
w1 = (torch.randn(E, F, D, generator=g) * D ** -0.5)
w2 = (torch.randn(E, D, F, generator=g) * F ** -0.5)
b1 = torch.randn(E, F, generator=g) * 0.1
b2 = torch.randn(E, D, generator=g) * 0.1
DT = os.environ.get("EXP_DTYPE", "bf16")
WDT = torch.float32 if DT == "f32" else torch.bfloat16
b1d, b2d = b1.to(dev), b2.to(dev)
if DT in ("fp8", "fp8a8"):        # e4m3 weights + per-row scales; fp8a8: activations quantised too (fp8 MFMA)
    from m3asr.plan import quantize_fp8_rows
    q1, s1 = quantize_fp8_rows(w1, dims=(2,))
    q2, s2 = quantize_fp8_rows(w2, dims=(2,))
    q1d, s1d, q2d, s2d = q1.to(dev), s1.to(dev), q2.to(dev), s2.to(dev)
    w1h, w2h = (q1.float() * s1.unsqueeze(-1)).to(WDT).to(dev), (q2.float() * s2.unsqueeze(-1)).to(WDT).to(dev)
    hs = 0.05 if DT == "fp8a8" else None
    fn = lambda: ops.moe_expert_ffn(x, gate, q1d, b1d, q2d, b2d, w1_scale=s1d, w2_scale=s2d, h_scale=hs)
else:
    w1h, w2h = w1.to(WDT).to(dev), w2.to(WDT).to(dev)
    fn = lambda: ops.moe_expert_ffn(x, gate, w1h, b1d, w2h, b2d)
y = fn()
# reference on a sample of rows, fp32 math on the bf16-rounded weights
idx = torch.arange(0, S, max(1, S // 256))
xe, ge = x[idx].cpu(), gate[idx].cpu().long()
h = torch.nn.functional.silu(torch.einsum("sd,sfd->sf", xe.to(WDT).float(), w1h.cpu().float()[ge]) + b1[ge])
want = torch.einsum("sf,sdf->sd", h.to(WDT).float(), w2h.cpu().float()[ge]) + b2[ge]
err = float((y[idx].cpu() - want).abs().max()) / float(want.abs().max())
if os.environ.get("EXP_NO_GRAPH"):          # counter passes: a few plain launches, no graph
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(json.dumps({"S": S, "dtype": str(WDT), "rel_err": round(err, 5), "launches": 6}), flush=True)
    sys.exit(0)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3): fn()
    st.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(10): fn()
    for _ in range(3): gr.replay()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): gr.replay()
    e1.record(st)
    st.synchronize()
us = e0.elapsed_time(e1) / 100 * 1e3
print(json.dumps({"lib": os.environ.get("M3ASR_LIB", "in-tree"), "S": S, "dtype": DT, "op_us": round(us, 2),
                  "TFLOPs": round(4 * D * F * S / us / 1e6, 1), "rel_err": round(err, 5)}), flush=True)
