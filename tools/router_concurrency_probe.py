#!/usr/bin/env python3
"""Kernel-level probe for DESIGN.md 10.8: launches of m3_moe_router from four streams at once, every result compared with what
the same launch computes alone.  Result on MI355X: 0 of 1600 launches differ, with the in-tree build AND with a library built
with packed-FP32 VALU instructions (make NOPK= OBJDIR=build_pk LIB=../tools/_pk.so; M3ASR_LIB=tools/_pk.so) -- router launches
next to router launches are not the failing combination; the engine-level test (tests/test_concurrent_gpu.py, where the
router's neighbours on a CU are the other contexts' GEMM / conv work-groups) is what shows the difference between the builds.
usage: router_concurrency_probe.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import torch
from m3asr import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
S, D, E, NS = 4480, 512, 64, 4
g = torch.Generator().manual_seed(0)
w = (torch.randn(E, 2 * D, generator=g) * 0.05).cuda()
ln = ((1.0 + 0.2 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda(), 1e-5)
common = 1.5 * torch.randn(1, D, generator=g)
xs = [(torch.randn(S, D, generator=g) + common).cuda() for _ in range(NS)]
es = [torch.randn(S, D, generator=g).cuda() for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
ref = []
for i in range(NS):
    lg, xn = ops.moe_router(es[i], xs[i], w, ln)
    torch.cuda.synchronize()
    ref.append((lg.clone(), xn.clone()))
bad = 0
for r in range(rounds):
    outs = []
    for k in range(3):                       # a few launches per stream so that different phases of the kernel meet on a CU
        for i in range(NS):
            with torch.cuda.stream(streams[i]):
                o = ops.moe_router(es[i], xs[i], w, ln)
                if k == 2:
                    outs.append(o)
    torch.cuda.synchronize()
    for i, (lg, xn) in enumerate(outs):
        if not (torch.equal(lg, ref[i][0]) and torch.equal(xn, ref[i][1])):
            bad += 1
            if bad <= 3:
                rows = (xn != ref[i][1]).any(-1).nonzero().view(-1).tolist()
                print("  round %d stream %d: xn rows %s differ, max |diff| %.3e" % (r, i, rows[:4], float((xn - ref[i][1]).abs().max())))
print("%s: %d of %d concurrent router launches differ from their serial result" % (os.environ.get("M3ASR_LIB", "in-tree library").split("/")[-1], bad, rounds * NS))
