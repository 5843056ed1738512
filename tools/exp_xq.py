"""Experiment: the fused fp8 expert kernel fed with rows quantised by the router kernel (M3_ROUTER_XQ=1, default) against the
in-kernel quantisation (M3_ROUTER_XQ=0): logits of one all-local fp8-arithmetic engine saved per setting, compared by a third call.
  python tools/exp_xq.py run B out.npy      |     python tools/exp_xq.py cmp a.npy b.npy"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

if sys.argv[1] == "cmp":
    for suf in (".gi.npy", ".gv.npy", ".e5.npy", ".b0.npy"):
        if os.path.exists(sys.argv[2] + suf):
            a, b = np.load(sys.argv[2] + suf), np.load(sys.argv[3] + suf)
            d = np.abs(a.astype(np.float64) - b.astype(np.float64))
            print(suf, "max |diff| %.4e, differing %d of %d, max |a| %.3e" % (d.max(), int((d > 0).sum()), d.size, np.abs(a).max()))
            if suf == ".b0.npy" and d.max() > 0:
                rows = np.nonzero(d.max(axis=1) > 0)[0]
                print("   rows differing", len(rows), "first", rows[:10].tolist(), "cols of first", np.nonzero(d[rows[0]] > 0)[0][:10].tolist(), d[rows[0]].max())
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b)
    print("max |diff| %.4e  differing elements %d of %d  rows with a difference %d of %d" %
          (d.max(), int((d > 0).sum()), d.size, int((d.reshape(-1, d.shape[-1]).max(axis=1) > 0).sum()), d.size // d.shape[-1]))
    sys.exit(0)

import torch
from conftest import load_golden
import test_full_size_gpu as T
from m3asr.engine import Engine
from m3asr.weights import make_weights

B, out = int(sys.argv[2]), sys.argv[3]
cfg, z = load_golden("cfg5share")
w = make_weights(cfg, seed=int(z["weight_seed"]))
rng = np.random.default_rng(4242)
lengths = rng.integers(50, 501, B); lengths[::64] = 500
feat = torch.from_numpy(rng.random((B, 500, cfg.input_dim), dtype=np.float32))
fl = torch.from_numpy(lengths.astype(np.int32))
cfg8, w8 = T._calibrated_fp8(cfg, w, feat, fl)
eng = Engine.from_state_dict(cfg8, w8, **({"debug_taps": True} if os.environ.get("EXP_TAPS") else {}))
y = eng(feat.cuda().contiguous(), fl.view(1, -1).cuda().contiguous()).cpu()
kern = {s_["name"]: s_["kernel"] for s_ in eng.stage_info()}
print("expert kernel:", kern["blocks.0.moe_local.expert"], " router:", kern["blocks.0.moe_router"], " logits", tuple(y.shape), flush=True)
valid = (torch.arange(y.shape[1]).view(1, -1) < T.sub_len(fl.long()).view(-1, 1)).unsqueeze(-1)
np.save(out, (y * valid).numpy())
if os.environ.get("EXP_TAPS"):
    gi = eng.buffer("blocks.0.gate_idx", torch.int32).cpu()
    np.save(out + ".gi.npy", gi.numpy())
    np.save(out + ".gv.npy", eng.buffer("blocks.0.gate_value").cpu().numpy())
    np.save(out + ".b0.npy", (eng.buffer("blocks.0.out").view(-1, 512).cpu() * (gi >= 0).view(-1, 1)).numpy())
    np.save(out + ".e5.npy", eng.buffer("embed.blocks.5.out").view(-1, 512).cpu().numpy())
