"""Engine vs CPU oracle at the longest input of the reference's TensorRT profile (builder.py:58-64: up to 6100 frames):
index arithmetic, workspace carving and the long-batch kernels at S ~ 4500 rows (development tool, GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch
from m3asr.config import EncoderConfig
from m3asr.engine import Engine
from m3asr.weights import make_weights
from oracle.encoder_ref import encoder_forward, sub_len
cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
w = make_weights(cfg, seed=1)
g = torch.Generator().manual_seed(0)
B, T = 3, 6100
feat = torch.rand(B, T, cfg.input_dim, generator=g)
fl = torch.tensor([6100, 3001, 777], dtype=torch.int32)
t0 = time.time(); want = encoder_forward(w, cfg, feat, fl); print("oracle s", time.time() - t0, flush=True)
for dt in ("f32", "bf16"):
    c = EncoderConfig(**{**cfg.__dict__, "weight_dtype": dt})
    eng = Engine.from_state_dict(c, w)
    out = eng(feat.cuda(), fl.view(1, -1).cuda()).cpu()
    valid = torch.arange(out.shape[1]).view(1, -1) < sub_len(fl.long()).view(-1, 1)
    err = float((out - want).abs()[valid].max()); sc = float(want.abs()[valid].max())
    print(dt, "ws MB", eng.workspace_size(B, T) / 1e6, "max err", err, "scale", sc, "finite", bool(torch.isfinite(out).all()), flush=True)
