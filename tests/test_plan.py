"""CPU tests of the host logic: weight packing and the plan file format (no device needed)."""
import torch

from m3asr.config import EncoderConfig
from m3asr.plan import pack_weights, save_plan, load_plan, positional_table
from m3asr.weights import make_weights
from oracle.encoder_ref import positional_table as ref_pe


def test_pack_layouts_and_plan_roundtrip(tmp_path):
    cfg = EncoderConfig.tiny()
    w = make_weights(cfg, seed=1)
    p = pack_weights(w, cfg)
    D, F2 = cfg.attention_dim, cfg.sub_freq
    assert p["blocks.0.self_attn.qkv.ln.weight"].shape == (3 * D, D)
    # LayerNorm affine folded into the projection: Linear(LN(x)) == Linear'(normalise(x))
    x = torch.randn(5, D, generator=torch.Generator().manual_seed(0))
    xn = torch.nn.functional.layer_norm(x, (D,), None, None, 1e-12)
    want = torch.nn.functional.linear(torch.nn.functional.layer_norm(
        x, (D,), w["blocks.0.norm_mha.weight"], w["blocks.0.norm_mha.bias"], 1e-12),
        w["blocks.0.self_attn.linear_k.weight"], w["blocks.0.self_attn.linear_k.bias"])
    got = torch.nn.functional.linear(xn, p["blocks.0.self_attn.qkv.ln.weight"][D:2 * D], p["blocks.0.self_attn.qkv.ln.bias"][D:2 * D])
    assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)
    assert p["subsampling.conv.2.weight_ohwi"].shape == (D, 3, 3, D)
    assert torch.equal(p["subsampling.conv.2.weight_ohwi"][3, 1, 2, 5], w["subsampling.conv.2.weight"][3, 5, 1, 2])
    wl, wr = p["subsampling.out.0.weight"], w["subsampling.out.0.weight"]
    assert torch.equal(wl[:, 2 * D + 7], wr[:, 7 * F2 + 2])          # (f=2, c=7) <- (c=7, f=2)
    assert torch.equal(p["blocks.1.feed_forward.router_weights_t"], w["blocks.1.feed_forward.router_weights"].t())
    assert p["blocks.0.conv_module.depthwise_conv.weight_kc"].shape == (cfg.cnn_module_kernel, D)
    # the positional table is the reference's formula (positional_encoding.py:40-48)
    assert torch.equal(p["pe"][:50], ref_pe(50, D)[0])
    path = str(tmp_path / "tiny.plan")
    save_plan(path, cfg, p, extra={"note": "x"})
    cfg2, p2, extra = load_plan(path)
    assert cfg2 == cfg and extra == {"note": "x"} and list(p2) == list(p)
    assert all(torch.equal(p[k], p2[k]) for k in p)


def test_pack_expert_parallel_slice():
    cfg = EncoderConfig.tiny(num_experts=2, ep_world_size=2, ep_rank=1)
    full = make_weights(EncoderConfig.tiny(num_experts=4), seed=1)
    p = pack_weights(full, cfg)     # load_state_dict_comm semantics: keep experts [rank*E_loc, (rank+1)*E_loc)
    assert torch.equal(p["blocks.0.feed_forward.experts.w_1.weight"], full["blocks.0.feed_forward.experts.w_1.weight"][2:4])
    w2 = full["blocks.0.feed_forward.experts.w_2.weight"][2:4]
    assert torch.equal(p["blocks.0.feed_forward.experts.w_2.weight_sliced"][1, 0, 5, 7], w2[1, 5, 7])
    assert p["blocks.0.feed_forward.router_weights_t"].shape[0] == 4     # the router still scores all experts


def test_batchnorm_fold_matches_eval_bn():
    import torch.nn.functional as F
    cfg = EncoderConfig.tiny(cnn_module_norm="batch_norm", embed_cnn_module_norm="batch_norm")
    w = make_weights(cfg, seed=2)
    g = torch.Generator().manual_seed(0)
    for pfx in ["blocks.%d.conv_module." % i for i in range(cfg.num_blocks)] + \
               ["embed.blocks.%d.conv_module." % i for i in range(cfg.embed_blocks)]:
        w[pfx + "norm.running_mean"] = torch.randn(cfg.attention_dim, generator=g)
        w[pfx + "norm.running_var"] = torch.rand(cfg.attention_dim, generator=g) + 0.5
    p = pack_weights(w, cfg)
    c = "blocks.0.conv_module."
    x = torch.randn(2, cfg.attention_dim, 20, generator=g)
    want = F.batch_norm(F.conv1d(x, w[c + "depthwise_conv.weight"], w[c + "depthwise_conv.bias"], padding=7,
                                 groups=cfg.attention_dim),
                        w[c + "norm.running_mean"], w[c + "norm.running_var"], w[c + "norm.weight"], w[c + "norm.bias"],
                        False, 0.0, 1e-5)
    got = F.conv1d(x, p[c + "depthwise_conv.weight_kc"].t().unsqueeze(1).contiguous(), p[c + "depthwise_conv.bias"],
                   padding=7, groups=cfg.attention_dim)
    assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)


def test_fp8_plan_formats_and_calibration_entries(tmp_path):
    """The 8-bit plan on the host side: e4m3 expert weights with one scale per output row (dequantised value within one
    e4m3 step of the weight), bf16 dense GEMM weights, and -- in the fp8-ARITHMETIC mode -- one h_scale per MoE layer, taken
    from the state dict when the calibrator put it there, else the documented default; all of it survives the plan file."""
    import dataclasses
    from m3asr.plan import DEFAULT_H_SCALE, quantize_fp8_rows
    cfg = dataclasses.replace(EncoderConfig.tiny(), weight_dtype="fp8", fp8_activations=True)
    w = make_weights(cfg, seed=5)
    w["blocks.1.feed_forward.experts.h_scale"] = torch.tensor([0.0123])      # what m3asr.calibrate writes
    p = pack_weights(w, cfg)
    k1 = "blocks.0.feed_forward.experts.w_1."
    assert p[k1 + "weight"].dtype == torch.float8_e4m3fn and p[k1 + "scale"].shape == w[k1 + "weight"].shape[:2]
    deq = p[k1 + "weight"].float() * p[k1 + "scale"].unsqueeze(-1)
    err = (deq - w[k1 + "weight"]).abs()
    assert bool((err <= w[k1 + "weight"].abs() * 2.0 ** -4 + p[k1 + "scale"].unsqueeze(-1) * 2.0 ** -9 + 1e-12).all())   # 3 mantissa bits
    assert p["blocks.0.self_attn.qkv.ln.weight"].dtype == torch.bfloat16                   # dense GEMM weights: bf16
    assert p["blocks.0.feed_forward.router_weights_t"].dtype == torch.float32              # the router stays fp32
    assert abs(float(p["blocks.0.feed_forward.experts.h_scale"]) - DEFAULT_H_SCALE) < 1e-8
    assert abs(float(p["blocks.1.feed_forward.experts.h_scale"]) - 0.0123) < 1e-9
    path = str(tmp_path / "tiny8.plan")
    save_plan(path, cfg, p)
    cfg2, p2, _ = load_plan(path)
    assert cfg2 == cfg and cfg2.fp8_activations and list(p2) == list(p)
    assert all(p[k].dtype == p2[k].dtype and torch.equal(p[k].view(torch.uint8) if p[k].dtype == torch.float8_e4m3fn else p[k],
                                                         p2[k].view(torch.uint8) if p2[k].dtype == torch.float8_e4m3fn else p2[k]) for k in p)
    # rows of zeros quantise to zeros with a finite scale
    q, s = quantize_fp8_rows(torch.zeros(2, 3, 8), dims=(2,))
    assert bool((q.float() == 0).all()) and bool(torch.isfinite(s).all())
