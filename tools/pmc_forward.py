"""Whole-forward HBM-traffic probe for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (bench.py itself segfaults
inside the profiler's counter collection on this pool).  Runs the 18-layer / 32-expert fp32 engine at 1 x 206 frames stage by
stage (no hipGraph), `passes` forwards with calibrated (load-balanced) routing.  usage: pmc_forward.py [passes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch

torch.set_num_threads(4)
from m3asr.config import EncoderConfig
from m3asr.engine import Engine
from m3asr.weights import make_weights

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = EncoderConfig()
eng = Engine.from_state_dict(cfg, make_weights(cfg, seed=0))
feat = torch.rand(1, 206, cfg.input_dim, generator=torch.Generator().manual_seed(1)).cuda()
fl = torch.tensor([[206]], dtype=torch.int32).cuda()
eng.bind(feat, fl)
n = len(eng.stage_names())
for _ in range(passes):
    eng.run_stages(0, n)
eng.stream.synchronize()
print("pmc_forward ok", n, "stages", eng.num_kernels(), "kernels")
