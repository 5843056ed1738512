"""Smallest programs for checking that rocprofv3 --pmc works on this box: arg 'torch' = a torch op only,
'lib' = load libm3asr_hip.so and run one scatter, 'expert' = one grouped expert FFN at B=1 size."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd"))
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
x = torch.randn(1 << 20, device="cuda")
y = (x * 2).sum().item()
if mode in ("lib", "expert"):
    from m3asr import ops
    S, E, D, F = 50, 32, 512, 1024
    g = torch.randint(0, E, (S,), dtype=torch.int32, device="cuda")
    m, a, p = ops.moe_scatter_mapping(g, E)
    if mode == "expert":
        xx = torch.randn(S, D, device="cuda")
        w1 = torch.randn(E, F, D, device="cuda") * 0.04; b1 = torch.zeros(E, F, device="cuda")
        w2 = torch.randn(E, D, F, device="cuda") * 0.03; b2 = torch.zeros(E, D, device="cuda")
        for _ in range(3):
            out = ops.moe_expert_ffn(xx, g, w1, b1, w2, b2)
    torch.cuda.synchronize()
print("probe", mode, "ok")
