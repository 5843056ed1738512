"""Debug: are the router kernel's e4m3 rows the quantisation of its own fp32 rows (debug_taps engine: both are written)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
import test_full_size_gpu as T
from m3asr.engine import Engine
from m3asr.config import EncoderConfig
from m3asr.weights import make_weights
B = 64
cfg, z = load_golden("cfg5share")
cfg = EncoderConfig(**{**cfg.__dict__, "num_blocks": 1, "embed_blocks": 1})
w = make_weights(cfg, seed=3)
rng = np.random.default_rng(4242)
lengths = rng.integers(50, 501, B); lengths[::64] = 500
feat = torch.from_numpy(rng.random((B, 500, cfg.input_dim), dtype=np.float32))
fl = torch.from_numpy(lengths.astype(np.int32))
cfg8, w8 = T._calibrated_fp8(cfg, w, feat, fl)
eng = Engine.from_state_dict(cfg8, w8, debug_taps=True)
y = eng(feat.cuda().contiguous(), fl.view(1, -1).cuda().contiguous()).cpu()
kern = {s_["name"]: s_["kernel"] for s_ in eng.stage_info()}
print(kern["blocks.0.moe_local.expert"], kern["blocks.0.moe_router"], "packed", eng.packed_rows())
xn = eng.buffer("xn").view(-1, 512).cpu()
xq = eng.buffer("xq", torch.uint8).view(-1, 512).cpu()
sc = eng.buffer("xq_scale").cpu()
gi = eng.buffer("blocks.0.gate_idx", torch.int32).cpu()
live = gi >= 0
print("rows", xn.shape[0], "live", int(live.sum()))
amax = xn.abs().amax(dim=1).clamp_min(1e-30)
print("scale mismatch rows:", int(((sc - amax / 448.0).abs() > 1e-6 * amax)[live].sum()))
q = (xn * (448.0 / amax).unsqueeze(1)).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
neq = (q != xq)
print("byte mismatches on live rows:", int(neq[live].sum()), "of", int(live.sum()) * 512, " rows affected", int(neq[live].any(dim=1).sum()))
i = int(neq[live].any(dim=1).nonzero()[0]) if neq[live].any() else -1
if i >= 0:
    r = live.nonzero().view(-1)[i]
    c = neq[r].nonzero().view(-1)[:8]
    print("row", int(r), "cols", c.tolist(), "got", xq[r][c].tolist(), "want", q[r][c].tolist(), "x", xn[r][c].tolist(), "amax", float(amax[r]), "sc", float(sc[r]))

