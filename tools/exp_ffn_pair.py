#!/usr/bin/env python3
"""A/B asked for by VERDICT r3 item 3(a): the dense FFN pair of a Conformer block at B = 1 (S = 50 rows, D = 512, F = 1024) as
  A  two skinny GEMM launches (what the engine runs: w_1 + SiLU, then w_2 + 0.5 * residual), and
  B  ONE launch of the slab-form grouped expert kernel with E = 1 (F slices x row tiles, H on chip) + the combine launch that
     sums the F / slice partial outputs (the index launch m3_moe_expert_ffn also issues is not counted: a dense FFN needs none).
Weights rotate through N_SETS x 4.2 MB so that every launch streams them from HBM, as in the 18-layer forward.
Run under `rocprofv3 --kernel-trace --stats`; M3ASR_LIB selects a build with another slice width (make EXTRA=-DM3_EXPERT_SLICE=16).
positionwise_feed_forward.py:79-88."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch

from m3asr import _lib, ops

S, D, F, N_SETS, ITERS = 50, 512, 1024, 160, 480


def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn(N_SETS, F, D, generator=g) * D ** -0.5).to(dev)
    b1 = torch.zeros(N_SETS, F, device=dev)
    w2 = (torch.randn(N_SETS, D, F, generator=g) * F ** -0.5).to(dev)
    b2 = torch.zeros(N_SETS, D, device=dev)
    x = torch.randn(S, D, generator=g).to(dev)
    gate = torch.zeros(S, dtype=torch.int32, device=dev)
    ws = torch.empty(ops.moe_expert_workspace_size(S, 1, D, F), dtype=torch.uint8, device=dev)

    def pair_a(i):
        h = ops.linear(x, w1[i], b1[i], act=_lib.ACT_SILU)
        return ops.linear(h, w2[i], b2[i], resid=x, alpha=0.5)

    def pair_b(i):
        return ops.moe_expert_ffn(x, gate, w1[i:i + 1], b1[i:i + 1], w2[i:i + 1], b2[i:i + 1], resid=x, alpha=0.5, workspace=ws)

    ya, yb = pair_a(3), pair_b(3)
    err = float((ya - yb).abs().max())
    assert err < 1e-4, err
    out = {"max_abs_diff_A_vs_B": err, "slice": _lib.load().m3_moe_expert_slice()}
    for name, fn in (("A_two_gemms", pair_a), ("B_slab_form", pair_b)):
        for i in range(20):
            fn(i % N_SETS)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(ITERS):
            fn(i % N_SETS)
        e1.record()
        torch.cuda.synchronize()
        out[name + "_us_per_pair_wall"] = round(e0.elapsed_time(e1) / ITERS * 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
