"""InferHelper with the reference's shape (TRTAPI++/python/trt_helper/infer_helper.py:37-161, infer.py:27-103):
load a plan, run it on host arrays, optionally compare with baseline outputs using the reference's own tolerance
torch.allclose(rtol=1e-05, atol=1e-03) (infer_helper.py:93), then a 10-iteration timing loop."""
import time

import numpy as np
import torch

from m3asr.engine import Engine
from m3asr.plan import load_plan
from . import trt


class InferHelper:
    def __init__(self, plan_name, trt_logger=None, device="cuda:0"):
        self.logger = trt_logger if trt_logger is not None else trt.Logger(trt.Logger.INFO)
        cfg, packed, extra = load_plan(plan_name)
        self.cfg, self.extra = cfg, extra
        self.engine = Engine(cfg, packed, device=device)
        self.output_bias = None      # a prior is inside the plan (folded into out_linear or applied after log-softmax)

    def _run(self, feat, feat_len, use_graph):
        out = self.engine.forward(feat, feat_len, use_graph=use_graph)
        self.engine.stream.synchronize()
        return out if self.output_bias is None else out + self.output_bias

    def infer(self, inputs, base_outputs=None):
        """inputs: [feat (B,T,idim) float32, feat_len (1,B) int32] as numpy arrays or torch tensors."""
        dev = self.engine.device
        feat = torch.as_tensor(np.asarray(inputs[0]) if not torch.is_tensor(inputs[0]) else inputs[0]).to(dev, torch.float32).contiguous()
        feat_len = torch.as_tensor(np.asarray(inputs[1]) if not torch.is_tensor(inputs[1]) else inputs[1]).to(dev, torch.int32).contiguous()
        self._run(feat, feat_len, use_graph=True)              # warm up (captures the hipGraph)
        t1 = time.perf_counter()
        out = self._run(feat, feat_len, use_graph=True)
        t2 = time.perf_counter()
        print("time=" + str((t2 - t1) * 1000) + "ms")
        outputs = [out.cpu()]
        if base_outputs is not None:
            for o, base in zip(outputs, base_outputs):
                base = torch.as_tensor(base)
                ok = torch.allclose(base, o, 1e-05, 1e-03)
                print("outputs.shape:" + str(tuple(o.shape)) + " outputs.sum:" + str(float(o.sum())) +
                      " base.sum:" + str(float(base.sum())))
                print("torch.allclose result:" + str(ok))
            t1 = time.perf_counter()
            for _ in range(10):
                self.engine.forward(use_graph=True)
            self.engine.stream.synchronize()
            print("time=" + str((time.perf_counter() - t1) * 100) + "ms per run (10 runs)")
        return [o.numpy() for o in outputs]
