#!/usr/bin/env python3
"""Roofline micro-benchmarks of the MoE kernels at saturating sizes (the B=1x206 headline is
weight-streaming bound, so HBM GB/s of scatter/gather and MFMA utilisation of the grouped expert FFN are quoted on a
saturating run of the SAME kernels).  Prints one JSON object; run under rocprofv3 --kernel-trace --stats for profiles/.

  scatter / gather : algorithmic bytes = 2*S*D*4 + 4*S          vs HBM peak 8 TB/s
  expert FFN fp32  : algorithmic FLOPs = 4*D*F*S (2 GEMMs)      vs fp32 MFMA peak 157.3 TFLOP/s
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch

from m3asr import ops


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    out = {}
    D, F, E = 512, 1024, 32
    g = torch.Generator().manual_seed(0)
    for S in (65536, 262144):
        gate = torch.randint(0, E, (S,), generator=g, dtype=torch.int32).to(dev)
        x = torch.randn(S, D, generator=g).to(dev)
        mapping, acc, pos = ops.moe_scatter_mapping(gate, E)
        buf = torch.empty_like(x)
        lib_scatter = lambda: ops._lib.load().m3_moe_local_scatter(x.data_ptr(), mapping.data_ptr(), S, D * 4, buf.data_ptr(),
                                                                   torch.cuda.current_stream().cuda_stream)
        y = torch.empty_like(x)
        lib_gather = lambda: ops._lib.load().m3_moe_local_gather(buf.data_ptr(), mapping.data_ptr(), S, D * 4, y.data_ptr(),
                                                                 torch.cuda.current_stream().cuda_stream)
        nbytes = 2 * S * D * 4 + 4 * S
        ts, tg = timed(lib_scatter), timed(lib_gather)
        ti = timed(lambda: ops.moe_scatter_mapping(gate, E))
        assert torch.equal(y, x)
        out["S=%d" % S] = {"scatter_GBps": round(nbytes / ts / 1e9, 1), "gather_GBps": round(nbytes / tg / 1e9, 1),
                           "scatter_frac_of_8TBps": round(nbytes / ts / 8e12, 3), "gather_frac_of_8TBps": round(nbytes / tg / 8e12, 3),
                           "index_kernel_us": round(ti * 1e6, 1), "bytes": nbytes}
    # grouped expert FFN, fp32, balanced routing, saturating M
    w1 = (torch.randn(E, F, D, generator=g) * D ** -0.5).to(dev)
    b1 = torch.zeros(E, F, device=dev)
    w2 = (torch.randn(E, D, F, generator=g) * F ** -0.5).to(dev)
    b2 = torch.zeros(E, D, device=dev)
    for S in (2048, 16384):
        gate = (torch.arange(S, dtype=torch.int32) % E).to(dev)
        x = torch.randn(S, D, generator=g).to(dev)
        ws = torch.empty(ops.moe_expert_workspace_size(S, E, D, F), dtype=torch.uint8, device=dev)
        t = timed(lambda: ops.moe_expert_ffn(x, gate, w1, b1, w2, b2, workspace=ws), iters=10)
        flops = 4.0 * D * F * S
        out["expert_ffn_fp32_S=%d" % S] = {"ms": round(t * 1e3, 3), "TFLOPs": round(flops / t / 1e12, 2),
                                           "frac_of_157TF_fp32_mfma": round(flops / t / 157.3e12, 3),
                                           "rows_per_expert": S // E,
                                           "note": "includes the index + combine launches of m3_moe_expert_ffn"}
    # the same in the bf16-weight mode (bf16 MFMA, fp32 accumulate): dense MFMA peak 2.5 PFLOP/s
    w1h, w2h = w1.to(torch.bfloat16), w2.to(torch.bfloat16)
    for S in (2048, 16384):
        gate = (torch.arange(S, dtype=torch.int32) % E).to(dev)
        x = torch.randn(S, D, generator=g).to(dev)
        ws = torch.empty(ops.moe_expert_workspace_size(S, E, D, F), dtype=torch.uint8, device=dev)
        t = timed(lambda: ops.moe_expert_ffn(x, gate, w1h, b1, w2h, b2, workspace=ws), iters=10)
        flops = 4.0 * D * F * S
        out["expert_ffn_bf16_S=%d" % S] = {"ms": round(t * 1e3, 3), "TFLOPs": round(flops / t / 1e12, 2),
                                           "frac_of_2500TF_bf16_mfma": round(flops / t / 2.5e15, 4),
                                           "rows_per_expert": S // E,
                                           "note": "includes the index + combine launches of m3_moe_expert_ffn_bf16"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
