"""HBM-traffic probe for the grouped expert FFN at the headline shape (S=50 rows, 32 experts, D=512, F=1024), made for
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes: the full bench.py segfaults inside the profiler's counter
collection on this pool (tool crash before the first kernel of ours), a small process does not.
18 layers of expert weights (2.4 GB fp32 / 1.2 GB bf16: far beyond the 256 MB Infinity Cache) are visited round-robin
so every launch streams its weights from HBM as in a real forward.  usage: pmc_expert.py [f32|bf16|fp8] [passes]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import numpy as np
import torch

from m3asr import ops

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
S, E, D, F, L = 50, 32, 512, 1024, 18
g = torch.Generator(device="cuda").manual_seed(0)
tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "fp8": torch.float8_e4m3fn}[dtype]
layers = []
for _ in range(L):
    w1 = torch.randn(E, F, D, device="cuda", generator=g) * D ** -0.5
    w2 = torch.randn(E, D, F, device="cuda", generator=g) * F ** -0.5
    sc = None
    if dtype == "fp8":
        s1, s2 = w1.abs().amax(2) / 448.0, w2.abs().amax(2) / 448.0
        w1, w2, sc = (w1 / s1.unsqueeze(-1)).to(tdt), (w2 / s2.unsqueeze(-1)).to(tdt), (s1.contiguous(), s2.contiguous())
    else:
        w1, w2 = w1.to(tdt), w2.to(tdt)
    layers.append((w1, torch.zeros(E, F, device="cuda"), w2, torch.zeros(E, D, device="cuda"), sc))
rng = np.random.default_rng(7)
gates = [torch.from_numpy(rng.integers(0, E, S).astype(np.int32)).cuda() for _ in range(L)]
touched = [int(len(np.unique(gt.cpu().numpy()))) for gt in gates]
x = torch.randn(S, D, device="cuda", generator=g)
ws = torch.empty(ops.moe_expert_workspace_size(S, E, D, F), dtype=torch.uint8, device="cuda")
for _ in range(passes):
    for (w1, b1, w2, b2, sc), gt in zip(layers, gates):
        if sc is None:
            ops.moe_expert_ffn(x, gt, w1, b1, w2, b2, workspace=ws)
        else:
            ops.moe_expert_ffn(x, gt, w1, b1, w2, b2, workspace=ws, w1_scale=sc[0], w2_scale=sc[1])
torch.cuda.synchronize()
wsz = {"f32": 4, "bf16": 2, "fp8": 1}[dtype]
extra = (F + D) * 4 * (2 if dtype == "fp8" else 1)
alg = [t * (2 * D * F * wsz + extra) + S * 2 * D * 4 for t in touched]
print(json.dumps({"dtype": dtype, "launches": passes * L, "experts_touched_mean": float(np.mean(touched)),
                  "alg_bytes_per_launch_mean": float(np.mean(alg))}))
