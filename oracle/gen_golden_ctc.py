"""Generates tests/golden/ctc_decode.npz by running the REFERENCE's own CTC searches (container only).

`BaseCTCEncoder.ctc_greedy_search` / `ctc_prefix_beam_search` (trainer_3m_fix/model/encoder.py:156-275) are called
unbound on a stand-in `self` whose `forward` returns a given score matrix (the two methods use nothing else of the
object but `blank_idx`), so the fixture records exactly what the reference's search code produces for those scores.
Inputs are synthetic: (i) random scores, (ii) "peaky" scores shaped like a trained CTC output (mostly blank with runs of
symbols, repeats separated by blanks and not), (iii) a batch with ragged lengths for the greedy search.

Only data is written: inputs, hypotheses and scores.  Run:  python oracle/gen_golden_ctc.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import install_name_modules, import_reference  # noqa: E402


def peaky_scores(rng, T, V, blank, sharp):
    """A plausible CTC posterior: a symbol sequence with repeats laid out over T frames, noise of relative size 1/sharp."""
    x = rng.normal(0, 1, (T, V)).astype(np.float32)
    t = 0
    prev = None
    while t < T:
        r = rng.random()
        if r < 0.45:
            s = blank
        elif r < 0.6 and prev is not None:
            s = prev                      # repeat of the previous symbol (with or without a blank in between)
        else:
            s = int(rng.integers(1, V))
        run = int(rng.integers(1, 4))
        x[t:t + run, s] += sharp
        if s != blank:
            prev = s
        t += run
    return x


def main():
    install_name_modules()
    BaseCTCEncoder = import_reference("model.encoder").BaseCTCEncoder

    def fake_self(scores, lens, blank):
        def forward(xs, xs_lens, *a, **k):
            masks = torch.arange(scores.shape[1]).view(1, -1) < lens.view(-1, 1)
            return {"out_nosm": scores, "out_lens": masks.sum(-1).view(-1), "hidden": scores}
        return types.SimpleNamespace(forward=forward, blank_idx=blank)

    rng = np.random.default_rng(20240607)
    out = {}
    # ---- prefix beam search cases (one utterance each)
    beam_cases = [("rand_b4", 24, 12, 0, 4, None), ("peaky_b5", 40, 30, 0, 5, 4.0), ("peaky_b10", 60, 50, 0, 10, 2.5),
                  ("soft_b8", 32, 20, 0, 8, 1.0), ("blank3_b6", 36, 16, 3, 6, 3.0)]
    names = []
    for name, T, V, blank, beam, sharp in beam_cases:
        x = rng.normal(0, 1, (T, V)).astype(np.float32) if sharp is None else peaky_scores(rng, T, V, blank, sharp)
        scores = torch.from_numpy(x)[None]
        lens = torch.tensor([T])
        hyps, _ = BaseCTCEncoder.ctc_prefix_beam_search(fake_self(scores, lens, blank), scores, lens, beam)
        names.append(name)
        out[name + "_logits"] = x
        out[name + "_meta"] = np.array([blank, beam], dtype=np.int32)
        out[name + "_n"] = np.array([len(hyps)], dtype=np.int32)
        toks = np.full((len(hyps), T), -1, dtype=np.int32)
        for i, (p, _s) in enumerate(hyps):
            toks[i, :len(p)] = p
        out[name + "_hyp_tokens"] = toks
        out[name + "_hyp_len"] = np.array([len(p) for p, _ in hyps], dtype=np.int32)
        out[name + "_hyp_score"] = np.array([s for _, s in hyps], dtype=np.float64)
        print(name, "best:", hyps[0][0][:12], "score %.4f" % hyps[0][1], "n", len(hyps))
    out["beam_cases"] = np.array(names)
    # ---- greedy search on a ragged batch
    B, T, V = 5, 48, 40
    lens = np.array([48, 31, 1, 0, 17], dtype=np.int32)
    x = np.stack([peaky_scores(rng, T, V, 0, 3.0) for _ in range(B)])
    x[1, 5:9] = x[1, 5]                       # a run of identical frames
    scores = torch.from_numpy(x)
    tl = torch.from_numpy(lens).long()
    hyps = BaseCTCEncoder.ctc_greedy_search(fake_self(scores, tl, 0), scores, tl)
    toks = np.full((B, T), -1, dtype=np.int32)
    for i, h in enumerate(hyps):
        toks[i, :len(h)] = h
    out["greedy_logits"], out["greedy_lens"] = x, lens
    out["greedy_tokens"] = toks
    out["greedy_n"] = np.array([len(h) for h in hyps], dtype=np.int32)
    print("greedy lens", [len(h) for h in hyps])
    path = os.path.join(ROOT, "tests", "golden", "ctc_decode.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
