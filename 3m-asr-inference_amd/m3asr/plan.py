"""The "plan": config + packed fp32 weights -- what replaces TensorRT's serialized engine
(TRTAPI++/python/trt_helper/builder_helper.py:146-167 `build_engine`, infer.py:29-36 deserialize).

``pack_weights`` turns a reference-layout state_dict (the keys of ``Net.state_dict()``, builder.py:131-134)
into the layouts the HIP engine consumes (all build-time host work, done once):
  * LayerNorms that feed exactly one Linear (norm_ff_macaron, norm_mha, norm_conv, dense norm_ff, after_norm)
    are folded into it (`*.ln.weight/bias/wsum`): the GEMM runs on the raw rows and normalises its OUTPUT;
  * q/k/v projections fused into one [3D, D] weight + [3D] bias (one GEMM instead of three Linear layers,
    attention.py:334-343);
  * pointwise Conv1d weights (O, I, 1) viewed as [O, I]; depthwise (C,1,K) -> [K, C] (channel-last rows);
  * subsampling conv1 (C,1,3,3) -> [9, C]; conv2 (O,I,3,3) -> [O,3,3,I] (implicit-GEMM K order);
    the Linear after the convs gets its columns permuted from (c, f) to (f, c) because the engine keeps
    activations channel-last (the reference flattens (B,C,T,F)->(B,T,C*F), subsampling.py:141-142);
  * router_weights [D+De, E] -> transposed [E, D+De]; expert w_2 [E,D,F] -> slice-major [E,F/64,D,64];
  * eval BatchNorm in the conv module (cnn_module_norm='batch_norm') folded into the depthwise conv;
  * the sinusoidal table ``pe`` (positional_encoding.py:40-48) for max_len positions;
  * in expert-parallel mode only this rank's expert slice [rank*E_loc, (rank+1)*E_loc) is kept
    (load_state_dict_comm, conformer_fmoe_localComm_catEmbed_domain_acc_hier.py:259-273).

File format (``save_plan`` / ``load_plan``): 8-byte magic, u64 header length, JSON header
{config, tensors: {name: [offset, shape]}}, then raw little-endian fp32 data (256-B aligned).
"""
import json
import re
import math
import struct
from collections import OrderedDict

import numpy as np
import torch

from .config import EncoderConfig

MAGIC = b"M3ASRPL1"
EXPERT_SLICE = 64    # = m3_moe_expert_slice() of libm3asr_hip.so (checked when an Engine is created)


def positional_table(max_len, d):
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _pack_subsampling(sd, p, out):
    w0 = sd[p + "conv.0.weight"]                         # (C,1,3,3)
    C = w0.shape[0]
    out[p + "conv.0.weight_9c"] = w0.reshape(C, 9).t().contiguous()
    out[p + "conv.0.bias"] = sd[p + "conv.0.bias"]
    out[p + "conv.2.weight_ohwi"] = sd[p + "conv.2.weight"].permute(0, 2, 3, 1).contiguous()
    out[p + "conv.2.bias"] = sd[p + "conv.2.bias"]
    wl = sd[p + "out.0.weight"]                          # (D, C*F2), column index c*F2 + f
    F2 = wl.shape[1] // C
    out[p + "out.0.weight"] = wl.view(-1, C, F2).permute(0, 2, 1).reshape(wl.shape[0], F2 * C).contiguous()
    out[p + "out.0.bias"] = sd[p + "out.0.bias"]


def fold_layernorm(weight, bias, gamma, beta):
    """Linear(LayerNorm(x)) = ((x - mean) * rstd) @ (W * gamma)^T + (b + W @ beta): the LayerNorm's affine moves
    into the Linear, so the GEMM prologue only normalises (one fewer dependent load in a latency-bound kernel)."""
    w64, g64, b64 = weight.double(), gamma.double(), beta.double()
    wf = (w64 * g64.unsqueeze(0)).float()
    return {"ln.weight": wf, "ln.bias": (bias.double() + w64 @ b64).float(),
            # output-side LayerNorm (gemm.hip LN_EPI): y = rstd * (x . W'^T - mean * wsum) + bias'
            "ln.wsum": wf.double().sum(1).float(),
            # a frame masked to 0 AFTER the LayerNorm must see the plain bias: bias' - W.beta
            "ln.wbeta": (w64 @ b64).float()}


def _put_folded(out, prefix, folded, keep_wbeta=False):
    for k, v in folded.items():
        if k != "ln.wbeta" or keep_wbeta:
            out[prefix + k] = v


def _pack_block(sd, p, out, norm, moe, cfg):
    folded = ["norm_ff_macaron", "norm_mha", "norm_conv"] + ([] if moe else ["norm_ff"])
    for n in ("norm_ff_macaron", "norm_mha", "norm_conv", "norm_ff", "norm_final"):
        if n not in folded:
            out[p + n + ".weight"] = sd[p + n + ".weight"]
            out[p + n + ".bias"] = sd[p + n + ".bias"]
    for n in ("feed_forward_macaron.w_2", "self_attn.linear_out"):
        out[p + n + ".weight"] = sd[p + n + ".weight"]
        out[p + n + ".bias"] = sd[p + n + ".bias"]
    m = p + "feed_forward_macaron.w_1."
    _put_folded(out, m, fold_layernorm(sd[m + "weight"], sd[m + "bias"],
                                       sd[p + "norm_ff_macaron.weight"], sd[p + "norm_ff_macaron.bias"]))
    a = p + "self_attn."
    wqkv = torch.cat([sd[a + "linear_q.weight"], sd[a + "linear_k.weight"], sd[a + "linear_v.weight"]], 0)
    bqkv = torch.cat([sd[a + "linear_q.bias"], sd[a + "linear_k.bias"], sd[a + "linear_v.bias"]], 0)
    _put_folded(out, a + "qkv.", fold_layernorm(wqkv, bqkv, sd[p + "norm_mha.weight"], sd[p + "norm_mha.bias"]))
    out[a + "pos_bias_u"] = sd[a + "pos_bias_u"]
    out[a + "pos_bias_v"] = sd[a + "pos_bias_v"]
    c = p + "conv_module."
    _put_folded(out, c + "pointwise_conv1.",
                fold_layernorm(sd[c + "pointwise_conv1.weight"].squeeze(-1), sd[c + "pointwise_conv1.bias"],
                               sd[p + "norm_conv.weight"], sd[p + "norm_conv.bias"]), keep_wbeta=True)
    out[c + "pointwise_conv2.weight"] = sd[c + "pointwise_conv2.weight"].squeeze(-1)
    out[c + "pointwise_conv2.bias"] = sd[c + "pointwise_conv2.bias"]
    dw = sd[c + "depthwise_conv.weight"].squeeze(1)      # (C, K)
    db = sd[c + "depthwise_conv.bias"]
    if norm == "layer_norm":
        out[c + "norm.weight"] = sd[c + "norm.weight"]
        out[c + "norm.bias"] = sd[c + "norm.bias"]
    else:  # eval BatchNorm1d folds into the depthwise conv: y = (x - mean) / sqrt(var + eps) * g + b
        s = sd[c + "norm.weight"] / torch.sqrt(sd[c + "norm.running_var"] + 1e-5)
        dw = dw * s.unsqueeze(1)
        db = (db - sd[c + "norm.running_mean"]) * s + sd[c + "norm.bias"]
    out[c + "depthwise_conv.weight_kc"] = dw.t().contiguous()
    out[c + "depthwise_conv.bias"] = db
    if (cfg.causal if moe else cfg.embed_causal):
        # causal module (convolution.py:43-49,118-123): lorder = K - 1 ZERO frames are padded in front of pointwise_conv1, so at
        # the depthwise conv's input a frame left of the utterance is the constant GLU(pointwise_conv1.bias)
        b1 = sd[c + "pointwise_conv1.bias"]
        half = b1.numel() // 2
        out[c + "left_fill"] = b1[:half] * torch.sigmoid(b1[half:])
    f = p + "feed_forward."
    if not moe:
        _put_folded(out, f + "w_1.", fold_layernorm(sd[f + "w_1.weight"], sd[f + "w_1.bias"],
                                                    sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"]))
        out[f + "w_2.weight"] = sd[f + "w_2.weight"]
        out[f + "w_2.bias"] = sd[f + "w_2.bias"]
    else:
        out[f + "router_weights_t"] = sd[f + "router_weights"].t().contiguous()
        # fused route path: the x half of the router with norm_ff folded in (the embed half of all layers is stacked
        # into router_e_all by pack_weights)
        De_ = sd[f + "router_weights"].shape[0] - sd[p + "norm_ff.weight"].shape[0]
        rb = sd[f + "router_bias"] if (f + "router_bias") in sd else torch.zeros(sd[f + "router_weights"].shape[1])
        _put_folded(out, f + "router_x.", fold_layernorm(sd[f + "router_weights"][De_:].t().contiguous(), rb,
                                                         sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"]))
        if (f + "router_bias") in sd:
            out[f + "router_bias"] = sd[f + "router_bias"]
        if getattr(cfg, "fp8_activations", False) and cfg.weight_dtype == "fp8":
            # static scale of the hidden activations for the fp8-arithmetic expert FFN: from the calibrator
            # (m3asr/calibrate.py writes "...experts.h_scale" into the state dict), else the uncalibrated default
            hs = sd.get(f + "experts.h_scale")
            out[f + "experts.h_scale"] = (hs.reshape(1).float() if hs is not None else torch.tensor([DEFAULT_H_SCALE]))
        lo = cfg.ep_rank * cfg.num_experts if cfg.ep_world_size > 1 else 0
        for n in ("experts.w_1.weight", "experts.w_1.bias", "experts.w_2.weight", "experts.w_2.bias"):
            t = sd[f + n]
            if cfg.ep_world_size > 1 and t.shape[0] == cfg.num_experts * cfg.ep_world_size:
                t = t[lo: lo + cfg.num_experts]
            assert t.shape[0] == cfg.num_experts, "%s: %d experts, config says %d" % (f + n, t.shape[0], cfg.num_experts)
            if n == "experts.w_2.weight":
                # [E, D, F] -> slice-major [E, F/S, D, S] (S = EXPERT_SLICE): the bytes a workgroup of the grouped expert FFN
                # streams in its second GEMM become one contiguous run
                E_, D_, F_ = t.shape
                out[f + "experts.w_2.weight_sliced"] = t.view(E_, D_, F_ // EXPERT_SLICE, EXPERT_SLICE).permute(0, 2, 1, 3).contiguous()
            else:
                out[f + n] = t


def pack_weights(state_dict, cfg: EncoderConfig):
    """reference-layout state_dict (CPU fp32) -> OrderedDict of packed contiguous fp32 tensors."""
    sd = {k: v.detach().float() for k, v in state_dict.items()}
    out = OrderedDict()
    _pack_subsampling(sd, "embed.subsampling.", out)
    out["embed.after_norm.weight"] = sd["embed.after_norm.weight"]
    out["embed.after_norm.bias"] = sd["embed.after_norm.bias"]
    for i in range(cfg.embed_blocks):
        _pack_block(sd, "embed.blocks.%d." % i, out, cfg.embed_cnn_module_norm, False, cfg)
    _pack_subsampling(sd, "subsampling.", out)
    for i in range(cfg.num_blocks):
        _pack_block(sd, "blocks.%d." % i, out, cfg.cnn_module_norm, True, cfg)
    _put_folded(out, "out_linear.", fold_layernorm(sd["out_linear.weight"], sd["out_linear.bias"],
                                                   sd["after_norm.weight"], sd["after_norm.bias"]))
    out["router_e_all.weight"] = torch.cat(
        [sd["blocks.%d.feed_forward.router_weights" % i][:cfg.embed_dim].t() for i in range(cfg.num_blocks)], 0).contiguous()
    out["pe"] = positional_table(cfg.max_len, cfg.attention_dim)
    # every block's linear_pos weight stacked: p for all blocks = one GEMM per forward (attention.py:345)
    out["pos_all.weight"] = torch.cat(
        [sd["embed.blocks.%d.self_attn.linear_pos.weight" % i] for i in range(cfg.embed_blocks)] +
        [sd["blocks.%d.self_attn.linear_pos.weight" % i] for i in range(cfg.num_blocks)], 0)
    return cast_gemm_weights(OrderedDict((k, v.contiguous()) for k, v in out.items()), cfg)


def read_cmvn_stats(path):
    """Global CMVN statistics -> (mean, istd) float32 vectors.  Accepts a Kaldi text matrix
    ``[ sum_0 .. sum_{d-1} count \\n sumsq_0 .. sumsq_{d-1} 0 ]`` (what ``Cmvn.read_stats`` loads in
    trainer_3m_fix/loader/cectc_lattice_loader.py:21-24, applied with norm_vars=True) or a 2 x d ``.npy`` of (mean, istd)."""
    if path.endswith(".npy"):
        a = np.load(path, allow_pickle=False).astype(np.float64)
        return torch.from_numpy(a[0].astype(np.float32)), torch.from_numpy(a[1].astype(np.float32))
    txt = open(path).read().replace("[", " ").replace("]", " ")
    vals = np.array([float(v) for v in txt.split()], dtype=np.float64)
    stats = vals.reshape(2, -1)
    count = stats[0, -1]
    mean = stats[0, :-1] / count
    var = np.maximum(stats[1, :-1] / count - mean * mean, 1e-20)
    return torch.from_numpy(mean.astype(np.float32)), torch.from_numpy((1.0 / np.sqrt(var)).astype(np.float32))


def add_front_back_end(packed, cfg, cmvn=None, output_bias=None):
    """Optional front / back end of the acoustic score, attached to a packed weight dict:
    cmvn = (mean, istd) -> fused into the first subsampling conv of both encoders; output_bias [V] (e.g. -log prior,
    builder.py:83-88) -> added after log_softmax when cfg.log_softmax_out, else folded into out_linear's bias."""
    if cmvn is not None:
        packed["cmvn.mean"] = torch.as_tensor(cmvn[0], dtype=torch.float32).contiguous()
        packed["cmvn.istd"] = torch.as_tensor(cmvn[1], dtype=torch.float32).contiguous()
    if output_bias is not None:
        ob = torch.as_tensor(output_bias, dtype=torch.float32).flatten().contiguous()
        assert ob.numel() == cfg.output_dim
        if cfg.log_softmax_out:
            packed["output_bias"] = ob
        else:
            packed["out_linear.ln.bias"] = packed["out_linear.ln.bias"] + ob
    return packed


# GEMM weights that follow cfg.weight_dtype (csrc/engine.hip GETW); everything else -- router, norms, biases,
# conv1, depthwise conv, positional table -- stays fp32 in every mode.
_GEMM_WEIGHT = re.compile(
    r"((feed_forward_macaron|feed_forward)\.w_1|self_attn\.qkv|pointwise_conv1|^out_linear)\.ln\.weight$"
    r"|((feed_forward_macaron|feed_forward)\.w_2|self_attn\.linear_out|pointwise_conv2|subsampling\.out\.0)\.weight$"
    r"|experts\.w_1\.weight$|experts\.w_2\.weight_sliced$|conv\.2\.weight_ohwi$|^pos_all\.weight$")


def is_gemm_weight(name):
    return _GEMM_WEIGHT.search(name) is not None


FP8_MAX = 448.0    # largest finite OCP e4m3 value


# uncalibrated H scale of the fp8-arithmetic expert FFN: |H| up to 448 * 0.05 = 22.4 is representable (H = SiLU(z) of a
# LayerNorm'd input through ~unit-gain weights stays below that); e4m3 is floating point, so a loose scale costs range, not
# precision -- the calibrator (m3asr/calibrate.py) replaces it by 1.25 x the observed maximum / 448
DEFAULT_H_SCALE = 0.05


def quantize_fp8_rows(w, dims):
    """Per-output-row symmetric quantisation to OCP e4m3 (torch.float8_e4m3fn = the format of gfx950's fp8 conversions):
    scale = amax over `dims` / 448, q = round_to_nearest_even(w / scale).  Returns (q fp8, scale fp32 with `dims` removed)."""
    amax = w.abs().amax(dim=dims, keepdim=True).clamp_min(1e-30)
    scale = (amax / FP8_MAX).float()
    q = (w / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).contiguous()
    return q, scale.squeeze(dims).contiguous()


def cast_gemm_weights(packed, cfg: EncoderConfig):
    """Store the GEMM weights in cfg.weight_dtype ("f32" = no-op, "bf16" = round to nearest even).  The reference
    wires --fp16 / plugin_data_type = 1 (builder.py:160, builder_helper.py:47-57,109-123) without finishing it; bf16 is
    the 16-bit type on CDNA4.  "fp8": additionally the expert weights (94 % of the parameters) become e4m3 with one scale per
    output row (W8A16: dequantised to bf16 at the MFMA input).  The folded LayerNorm's column sums are recomputed from the ROUNDED weights, so that
    y = rstd * (a . W'^T - mean * wsum) stays an exact identity for the weights the kernel really multiplies by."""
    if cfg.weight_dtype == "f32":
        return packed
    assert cfg.weight_dtype in ("bf16", "fp8"), cfg.weight_dtype
    out = OrderedDict()
    for k, v in packed.items():
        if cfg.weight_dtype == "fp8" and v.dtype == torch.float32 and k.endswith("experts.w_1.weight"):
            q, sc = quantize_fp8_rows(v, dims=(2,))                       # [E][F][D]: one scale per (e, f)
            out[k], out[k[:-len("weight")] + "scale"] = q, sc
        elif cfg.weight_dtype == "fp8" and v.dtype == torch.float32 and k.endswith("experts.w_2.weight_sliced"):
            q, sc = quantize_fp8_rows(v, dims=(1, 3))                     # [E][F/64][D][64]: one scale per (e, d)
            out[k], out[k[:-len("weight_sliced")] + "scale"] = q, sc
        elif is_gemm_weight(k) and v.dtype == torch.float32:
            out[k] = v.to(torch.bfloat16).contiguous()
        else:
            out[k] = v
    for k in list(out):
        if k.endswith(".ln.weight") and out[k].dtype == torch.bfloat16:
            out[k[:-len("weight")] + "wsum"] = out[k].double().sum(1).float()
    return out


_DTYPES = {"f32": (torch.float32, "<f4", 4), "bf16": (torch.bfloat16, "<u2", 2), "i32": (torch.int32, "<i4", 4),
           "fp8": (torch.float8_e4m3fn, "u1", 1)}
_DTYPE_NAME = {torch.float32: "f32", torch.bfloat16: "bf16", torch.int32: "i32", torch.float8_e4m3fn: "fp8"}


def save_plan(path, cfg: EncoderConfig, packed, extra=None):
    index, off = {}, 0
    for k, v in packed.items():
        name = _DTYPE_NAME[v.dtype]
        index[k] = [off, list(v.shape)] + ([name] if name != "f32" else [])
        off += (v.numel() * _DTYPES[name][2] + 255) // 256 * 256
    header = json.dumps({"config": json.loads(cfg.to_json()), "tensors": index, "extra": extra or {}}).encode()
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(header)))
        f.write(header)
        pad = (-(16 + len(header))) % 256
        f.write(b"\0" * pad)
        for k, v in packed.items():
            if v.dtype == torch.bfloat16:
                b = v.contiguous().view(torch.int16).numpy().astype("<i2", copy=False).tobytes()
            elif v.dtype == torch.float8_e4m3fn:
                b = v.contiguous().view(torch.uint8).numpy().tobytes()
            else:
                b = v.contiguous().numpy().astype(_DTYPES[_DTYPE_NAME[v.dtype]][1], copy=False).tobytes()
            f.write(b)
            f.write(b"\0" * ((-len(b)) % 256))


def load_plan(path):
    """-> (EncoderConfig, OrderedDict name -> CPU tensor (fp32; bf16 for the GEMM weights of a bf16 plan), extra dict)"""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("%s is not an m3asr plan" % path)
        (hl,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hl).decode())
    base = 16 + hl
    base += (-base) % 256
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    packed = OrderedDict()
    for k, ent in header["tensors"].items():
        off, shape = ent[0], ent[1]
        tdt, ndt, _ = _DTYPES[ent[2] if len(ent) > 2 else "f32"]
        n = int(np.prod(shape)) if shape else 1
        if tdt == torch.bfloat16:
            arr = np.frombuffer(mm, dtype="<i2", count=n, offset=base + off).reshape(shape)
            packed[k] = torch.from_numpy(np.array(arr, copy=True)).view(torch.bfloat16)
        elif tdt == torch.float8_e4m3fn:
            arr = np.frombuffer(mm, dtype="u1", count=n, offset=base + off).reshape(shape)
            packed[k] = torch.from_numpy(np.array(arr, copy=True)).view(torch.float8_e4m3fn)
        else:
            arr = np.frombuffer(mm, dtype=ndt, count=n, offset=base + off).reshape(shape)
            packed[k] = torch.from_numpy(np.array(arr, copy=True))
    return EncoderConfig(**header["config"]), packed, header.get("extra", {})
