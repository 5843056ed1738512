import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import numpy as np, torch
from m3asr.config import EncoderConfig
from m3asr.weights import make_weights
from m3asr.engine import Engine
NCTX, B = 4, 64
cfg = EncoderConfig(num_blocks=1, num_experts=64, weight_dtype="bf16")
w = make_weights(cfg, seed=21)
rng = np.random.default_rng(77)
feats, lens = [], []
for c in range(NCTX):
    lengths = rng.integers(50, 501, B); lengths[0] = 500
    feats.append(torch.from_numpy(rng.random((B, 500, cfg.input_dim), dtype=np.float32)).cuda())
    lens.append(torch.from_numpy(lengths.astype(np.int32)).view(1, -1).cuda())
eng0 = Engine.from_state_dict(cfg, w)
ctxs = [eng0] + [eng0.clone_context() for _ in range(NCTX - 1)]
for e, f, l in zip(ctxs, feats, lens):
    e.bind(f, l)
names = ctxs[0].stage_names()
stop = names.index("blocks.0.moe_router") + 1
ref = []
for e in ctxs:
    for _ in range(3):
        e.run_stages(0, stop); torch.cuda.synchronize()
    ref.append(e.buffer("xn", torch.float32).clone())
bad = 0
R = int(os.environ.get("REPS", "400"))
for rep in range(R):
    for e in ctxs:
        e.run_stages(0, stop)
    torch.cuda.synchronize()
    for c, e in enumerate(ctxs):
        bad += int(not torch.equal(e.buffer("xn", torch.float32), ref[c]))
print("%s: %d of %d concurrent runs differ" % (os.environ.get("M3ASR_LIB", "in-tree").split("/")[-1], bad, R * NCTX))
