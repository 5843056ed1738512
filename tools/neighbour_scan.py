#!/usr/bin/env python3
"""Which neighbour makes moe_router_kernel return a wrong row?  
Two execution contexts of the 16-bit long-batch engine (B = 64): context A replays ONLY its router stage, context B replays ONE
other stage of its own forward, both in flight together; A's xn / router logits are compared with what A computes alone.  One
line per distinct (kernel, stage) of the forward.  Run once with the in-tree library and once with a library built WITH
packed-FP32 VALU instructions (make -C 3m-asr-inference_amd NOPK= OBJDIR=build_pk LIB=../tools/_pk.so; M3ASR_LIB=tools/_pk.so).
usage: neighbour_scan.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import numpy as np, torch
from m3asr.config import EncoderConfig
from m3asr.weights import make_weights
from m3asr.engine import Engine
B = 64
cfg = EncoderConfig(num_blocks=1, num_experts=64, weight_dtype="bf16")
w = make_weights(cfg, seed=21)
rng = np.random.default_rng(77)
feats, lens = [], []
for c in range(2):
    lengths = rng.integers(50, 501, B); lengths[0] = 500
    feats.append(torch.from_numpy(rng.random((B, 500, cfg.input_dim), dtype=np.float32)).cuda())
    lens.append(torch.from_numpy(lengths.astype(np.int32)).view(1, -1).cuda())
A = Engine.from_state_dict(cfg, w); Bc = A.clone_context()
for e, f, l in zip((A, Bc), feats, lens):
    e(f, l); torch.cuda.synchronize()
names = A.stage_names()
ir = names.index("blocks.0.moe_router")
A.run_stages(0, ir + 1); torch.cuda.synchronize()
ref_xn = A.buffer("xn", torch.float32).clone(); ref_rl = A.buffer("router_logits", torch.float32).clone()
seen = set()
for k, n in enumerate(names):
    kern = [s_["kernel"] for s_ in A.stage_info() if s_["name"] == n][0]
    key = (kern, n.split(".")[-1])
    if key in seen or n.startswith("blocks.0.moe_local") or "moe_top1" in n: continue
    seen.add(key)
    Bc.run_stages(0, k + 1); torch.cuda.synchronize()          # state in front of stage k is valid
    bad = 0; R = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    for r in range(R):
        for _ in range(6): Bc.run_stages(k, k + 1)
        for _ in range(6): A.run_stages(ir, ir + 1)
        torch.cuda.synchronize()
        bad += int(not (torch.equal(A.buffer("xn", torch.float32), ref_xn) and torch.equal(A.buffer("router_logits", torch.float32), ref_rl)))
    print(os.environ.get("M3ASR_LIB", "in-tree").split("/")[-1] + ": router next to %-34s (%-28s): %3d of %d rounds differ" % (n, kern[:28], bad, R), flush=True)
