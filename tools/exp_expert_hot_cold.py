#!/usr/bin/env python3
"""Development probe: the B = 1 expert operator (50 rows, 32 experts, fp32) with its weights resident in the Infinity Cache
(the same 134-MB weight set every call: larger than the L2s' 32 MB, smaller than the 256-MB MALL) against cold weights (rotating
through 18 sets = 2.4 GB, what a forward does): the most a next-layer expert-weight prefetch into the MALL could buy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch
from m3asr import ops
torch.manual_seed(0)
S, D, F, E, L = 50, 512, 1024, 32, 18
x = torch.randn(S, D, device="cuda")
gate = (torch.randperm(S) % E).to(torch.int32).cuda()     # ~25 of 32 experts touched, like the balanced bench routers
sets = [(torch.randn(E, F, D, device="cuda") * 0.03, torch.zeros(E, F, device="cuda"), torch.randn(E, D, F, device="cuda") * 0.03,
         torch.zeros(E, D, device="cuda")) for _ in range(L)]
print("experts touched:", int(torch.unique(gate).numel()))
res = {}
for mode in ("hot", "cold"):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3): ops.moe_expert_ffn(x, gate, *sets[i % L])
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(L):
                ops.moe_expert_ffn(x, gate, *(sets[0] if mode == "hot" else sets[i]))
        for _ in range(3): g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): g.replay()
        e1.record(s); s.synchronize()
        res[mode] = e0.elapsed_time(e1) * 1e3 / (20 * L)
print("index + expert + combine per call: MALL-resident weights %.2f us, cold weights %.2f us" % (res["hot"], res["cold"]))
