#!/usr/bin/env python3
"""How long does the host spend enqueuing one graph replay? (development tool, GPU box only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch
from m3asr.config import EncoderConfig
from m3asr.weights import make_weights
from m3asr.engine import Engine
cfg = EncoderConfig()
eng = Engine.from_state_dict(cfg, make_weights(cfg, seed=0))
feat = torch.rand(1, 206, 40).cuda(); fl = torch.tensor([[206]], dtype=torch.int32).cuda()
eng.bind(feat, fl)
for _ in range(5): eng.forward()
eng.stream.synchronize()
for n in (1, 5, 20):
    t0 = time.perf_counter()
    for _ in range(n): eng.forward()
    t1 = time.perf_counter()
    eng.stream.synchronize()
    t2 = time.perf_counter()
    print("n=%d: host enqueue %.3f ms per forward, total %.3f ms per forward" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
