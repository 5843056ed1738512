"""Conformer-MoE encoder description: emits the network of the reference's
trainer_3m_fix/model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py::Net.forward (:198-234) -- embed encoder
(model/conformer_embed_domain_acc.py:149-181), Conv2dSubsampling4, RelPositionalEncoding, N FmoeConformerLayer blocks,
after_norm, out_linear -- through ``network_helper.add*`` / the ``*PluginDynamic`` operators, op for op.

Unlike the reference this is not an nn.Module tree: the description is a config + a state_dict with the reference's
key names, and each sub-network is a small emitter function.  The op sequence, plugin names, attribute names and
tensor layouts handed to network_helper are the reference's (file:line cited per emitter), so this file doubles as
the op-by-op executor used to validate the fused native engine (trt_helper.BuilderHelper.build_engine).
"""
import math
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

from m3asr.config import EncoderConfig
from m3asr.plan import positional_table
from m3asr.weights import encoder_param_shapes
from trt_helper import trt


def _linear(sd, p):
    return SimpleNamespace(weight=sd[p + "weight"], bias=sd.get(p + "bias"))


def _norm(sd, p, eps):
    return SimpleNamespace(weight=sd[p + "weight"], bias=sd[p + "bias"], eps=eps)


def _conv(sd, p, kernel, stride, padding, groups):
    return SimpleNamespace(weight=sd[p + "weight"], bias=sd.get(p + "bias"), kernel_size=kernel, stride=stride,
                           padding=padding, dilation=tuple(1 for _ in kernel), groups=groups, padding_mode="zeros",
                           out_channels=sd[p + "weight"].shape[0])


def _plugin(nh, name, fields):
    creator = nh.plugin_registry.get_plugin_creator(name, "1", "")
    if not creator:
        raise RuntimeError("Could not find " + name)
    pfc = trt.PluginFieldCollection([
        trt.PluginField(k, np.array([v], dtype=np.int32 if isinstance(v, (int, np.integer)) else np.float32),
                        trt.PluginFieldType.INT32 if isinstance(v, (int, np.integer)) else trt.PluginFieldType.FLOAT32)
        for k, v in fields])
    plugin = creator.create_plugin(name, pfc)
    if not plugin:
        raise RuntimeError("Could not create_plugin " + name)
    return plugin


def _dtype(nh):
    """`data_type` field of the emitted plugins = the type of their ACTIVATION tensors.  The reference passes
    HelperConfig.plugin_data_type; here activations stay fp32 in every mode (the 16-bit mode, --fp16, is a property of
    the packed plan's GEMM weights), so the op-by-op emission -- the fp32 checker of the fused engine -- is always 0."""
    return 0


def emit_subsampling(nh, sd, p, x, x_len):
    """Conv2dSubsampling4.forward (layer/subsampling.py:103-145)."""
    layer = nh.network.add_shuffle(x)                       # (B,T,idim) -> (B,1,T,idim), trans_3d_to_4d_trt :29-36
    layer.reshape_dims = (0, 0, 1, -1)
    layer.second_transpose = (0, 2, 1, 3)
    nh.set_layer_name(layer, "trans_3d_to_4d")
    x = layer.get_output(0)
    x = nh.addReLU(nh.addConv2d(_conv(sd, p + "conv.0.", (3, 3), (2, 2), (0, 0), 1), x))
    x = nh.addReLU(nh.addConv2d(_conv(sd, p + "conv.2.", (3, 3), (2, 2), (0, 0), 1), x))
    sample = _plugin(nh, "MaskConv2dSamplePluginDynamic", [("left_padding", 2), ("stride", 2)])
    for tag in ("first_mask_sample", "second_mask_sample"):
        layer = nh.network.add_plugin_v2([x_len], sample)
        nh.set_layer_name(layer, tag)
        x_len = layer.get_output(0)
    x = nh.addShuffle(x, (0, 2, 1, 3), (0, 0, -1), None, "x.transpose(1, 2).view(b, t, c * f)")
    return nh.addLinear(_linear(sd, p + "out.0."), x), x_len


def emit_pos_enc(nh, x, d_model, max_len):
    """RelPositionalEncoding.forward (layer/positional_encoding.py:101-129): (x*sqrt(d), pe[:, :T'])."""
    plugin = _plugin(nh, "RelPositionalEncodingPluginDynamic",
                     [("data_type", _dtype(nh)), ("scale", float(math.sqrt(d_model))), ("max_len", int(max_len)),
                      ("dim", int(d_model))])
    pe = nh.addConstant(positional_table(max_len, d_model).unsqueeze(0))
    layer = nh.network.add_plugin_v2([x, pe], plugin)
    return layer.get_output(0), layer.get_output(1)


def emit_ffn(nh, sd, p, x):
    """PositionwiseFeedForward.forward (layer/positionwise_feed_forward.py:79-88)."""
    return nh.addLinear(_linear(sd, p + "w_2."), nh.addSiLU(nh.addLinear(_linear(sd, p + "w_1."), x)))


def emit_attention(nh, sd, p, x, x_len, pos_emb, h):
    """RelPositionMultiHeadedAttention.forward (layer/attention.py:320-384) + forward_attention_trt (:199-239)."""
    d_k = x.shape[-1] // h
    k = nh.addLinear(_linear(sd, p + "linear_k."), x)
    v = nh.addLinear(_linear(sd, p + "linear_v."), x)
    q = nh.addLinear(_linear(sd, p + "linear_q."), x)
    pp = nh.addLinear(SimpleNamespace(weight=sd[p + "linear_pos.weight"], bias=None), pos_emb)
    q = nh.addShuffle(q, None, (0, -1, h, d_k), None, "att_q_view")
    v = nh.addShuffle(v, None, (0, -1, h, d_k), (0, 2, 1, 3), "att_v_view_and transpose")
    k = nh.addShuffle(k, None, (0, -1, h, d_k), (0, 2, 3, 1), "att_k_view_and transpose")
    pp = nh.addShuffle(pp, None, (0, -1, h, d_k), (0, 2, 3, 1), "att_p_view")
    u = nh.addConstant(sd[p + "pos_bias_u"].view(1, 1, h, d_k), "pos_bias_u_4d")
    w = nh.addConstant(sd[p + "pos_bias_v"].view(1, 1, h, d_k), "pos_bias_v_4d")
    q_u = nh.addShuffle(nh.addAdd(q, u, "q + self.pos_bias_u"), None, None, (0, 2, 1, 3), "q_with_bias_u_trans")
    q_v = nh.addShuffle(nh.addAdd(q, w, "q + self.pos_bias_v"), None, None, (0, 2, 1, 3), "q_with_bias_v_trans")
    scores = nh.addAdd(nh.addMatMul(q_u, k, "q_with_bias_u_mul_k"), nh.addMatMul(q_v, pp, "q_with_bias_v_mul_p"),
                       "matrix_ac + matrix_bd")                          # no rel_shift, as in the reference
    plugin = _plugin(nh, "AttMaskedSoftmaxPluginDynamic", [("data_type", _dtype(nh)), ("scale", 1.0 / math.sqrt(d_k))])
    layer = nh.network.add_plugin_v2([scores, x_len], plugin)
    nh.set_layer_name(layer, "AttMaskedSoftmaxPluginDynamic")
    ctx = nh.addMatMul(layer.get_output(0), v, "matmul(p_attn, value)")
    ctx = nh.addShuffle(ctx, (0, 2, 1, 3), (0, -1, h * d_k), None, "attn_transpose_and_reshape")
    return nh.addLinear(_linear(sd, p + "linear_out."), ctx)


def emit_conv_module(nh, sd, p, x, x_len, kernel, norm, causal=False):
    """ConvolutionModule.forward (layer/convolution.py:83-167); causal: lorder = kernel - 1 frames of zeros in front of
    pointwise_conv1 through network.add_padding, depthwise conv without padding (:43-49,118-123)."""
    def masked(t):
        plugin = _plugin(nh, "MaskedFillPluginDynamic", [("data_type", _dtype(nh)), ("fill", 0.0)])
        layer = nh.network.add_plugin_v2([t, x_len], plugin)
        nh.set_layer_name(layer, "MaskedFillPluginDynamic")
        return layer.get_output(0)

    C = x.shape[-1]
    x = masked(nh.addShuffle(x, (0, 2, 1), None, None, "conv_trans"))
    x = nh.addShuffle(x, None, (0, 0, 1, -1), None, "conv_trans_3d_to_4d")
    if causal:
        layer = nh.network.add_padding(x, pre_padding=(0, kernel - 1), post_padding=(0, 0))
        nh.set_layer_name(layer, "conv_pad")
        x = layer.get_output(0)
    x = nh.addGLU(nh.addConv1d(_conv(sd, p + "pointwise_conv1.", (1,), (1,), (0,), 1), x), 1)
    x = nh.addConv1d(_conv(sd, p + "depthwise_conv.", (kernel,), (1,), (0 if causal else (kernel - 1) // 2,), C), x)
    if norm != "layer_norm":
        raise RuntimeError("op-by-op emission supports cnn_module_norm='layer_norm' only (the TRT-style forward "
                           "calls addLayerNorm unconditionally, convolution.py:145); batch_norm is folded by the engine")
    x = nh.addShuffle(x, (0, 3, 2, 1), None, None, "use_layer_norm_trans")
    x = nh.addSiLU(nh.addLayerNorm(_norm(sd, p + "norm.", 1e-5), x))
    x = nh.addShuffle(x, (0, 3, 2, 1), None, None, "use_layer_norm_trans")
    x = nh.addConv1d(_conv(sd, p + "pointwise_conv2.", (1,), (1,), (0,), 1), x)
    x = masked(nh.addShuffle(x, None, (0, 0, -1), None, "conv_trans_4d_to_3d"))
    return nh.addShuffle(x, (0, 2, 1), None, None, "use_layer_norm_trans")


def emit_moe(nh, sd, p, x, embed, x_len, cfg):
    """LocalFmoeCatEmbedFeedForward.forward (layer/positionwise_feed_forward.py:169-265)."""
    E, D, F = cfg.num_experts * max(cfg.ep_world_size, 1), x.shape[2], cfg.hidden_units
    router_in = nh.addCat([embed, x], dim=-1)
    logits = nh.addMatMul(router_in, nh.addConstant(sd[p + "router_weights"].view(1, -1, E)))
    if (p + "router_bias") in sd:
        logits = nh.addAdd(logits, nh.addConstant(sd[p + "router_bias"].view(1, 1, E)))
    layer = nh.network.add_plugin_v2([logits, x_len], _plugin(nh, "SoftmaxTopKPluginDynamic", [("data_type", _dtype(nh))]))
    nh.set_layer_name(layer, "SoftmaxTopKPluginDynamic")
    gate_value, gate_idx = layer.get_output(0), layer.get_output(1)
    plugin = _plugin(nh, "FMoEExpertPluginDynamic", [("data_type", _dtype(nh)), ("num_expert", int(cfg.num_experts)),
                                                    ("idim", int(D)), ("hidden_units", int(F))])
    consts = [nh.addConstant(sd[p + n]) for n in ("experts.w_1.weight", "experts.w_1.bias", "experts.w_2.weight",
                                                  "experts.w_2.bias")]
    layer = nh.network.add_plugin_v2([x, gate_idx] + consts, plugin)
    nh.set_layer_name(layer, "FMoEExpertPluginDynamic")
    y = layer.get_output(0)
    return y if cfg.keep_expert_output else nh.addProd(y, gate_value)


def emit_block(nh, sd, p, x, embed, x_len, pos_emb, cfg, heads, norm, moe):
    """FmoeConformerLayer.forward (layer/fmoe_transformer.py:72-170) / ConformerEncoderLayer.forward
    (layer/transformer.py:179-275): macaron FFN, rel-pos MHA, conv module, (MoE | FFN), final LayerNorm."""
    eps = 1e-12
    y = emit_ffn(nh, sd, p + "feed_forward_macaron.", nh.addLayerNorm(_norm(sd, p + "norm_ff_macaron.", eps), x))
    x = nh.addAdd(nh.addScale(y, 0.5), x)
    y = emit_attention(nh, sd, p + "self_attn.", nh.addLayerNorm(_norm(sd, p + "norm_mha.", eps), x), x_len, pos_emb, heads)
    x = nh.addAdd(x, y)
    y = emit_conv_module(nh, sd, p + "conv_module.", nh.addLayerNorm(_norm(sd, p + "norm_conv.", eps), x), x_len,
                         cfg.cnn_module_kernel, norm, bool(cfg.causal if moe else cfg.embed_causal))
    x = nh.addAdd(x, y, "conv_residual_layer")
    xn = nh.addLayerNorm(_norm(sd, p + "norm_ff.", eps), x)
    y = emit_moe(nh, sd, p + "feed_forward.", xn, embed, x_len, cfg) if moe else emit_ffn(nh, sd, p + "feed_forward.", xn)
    x = nh.addAdd(x, nh.addScale(y, 0.5), "residual_layer")
    return nh.addLayerNorm(_norm(sd, p + "norm_final.", eps), x)


class Net:
    """Encoder description with the reference's constructor arguments
    (model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py:31-60) and state_dict key names."""

    def __init__(self, input_dim, output_dim, **encoder_conf):
        self.cfg = EncoderConfig.from_reference_conf(input_dim, output_dim, encoder_conf)
        self._sd = OrderedDict()

    # ---- nn.Module-shaped surface used by builder.py:131-138 ----
    def load_state_dict(self, state_dict, strict=True):
        want = encoder_param_shapes(self.cfg)
        missing = [k for k in want if k not in state_dict and not any(
            u in k for u in ("concat_linear", "after_norm_6", "after_norm_12", "embed.out_linear"))]
        if missing and strict:
            raise RuntimeError("Missing key(s) in state_dict: " + ", ".join(missing[:8]))
        for k, v in state_dict.items():
            if k in want and tuple(v.shape) != tuple(want[k]):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(v.shape), tuple(want[k])))
        self._sd = OrderedDict((k, v.detach().float().cpu()) for k, v in state_dict.items())
        return missing, [k for k in state_dict if k not in want and "running" not in k and "num_batches" not in k]

    def state_dict(self):
        return self._sd

    def parameters(self):
        return list(self._sd.values())

    def eval(self):
        return self

    # ---- graph emission (reference: Net.forward(network_helper, xs, xs_len), :198-234) ----
    def __call__(self, network_helper, xs, xs_len, output_embed=False):
        return self.forward(network_helper, xs, xs_len, output_embed)

    def forward(self, nh, xs, xs_len, output_embed=False):
        cfg, sd = self.cfg, self._sd
        builder = getattr(nh, "_builder", None)
        if builder is not None:
            builder.note_model(sd, cfg)
        # embed encoder (conformer_embed_domain_acc.py:149-181)
        x, x_len = emit_subsampling(nh, sd, "embed.subsampling.", xs, xs_len)
        x, pos = emit_pos_enc(nh, x, cfg.embed_dim, cfg.max_len)
        for i in range(cfg.embed_blocks):
            x = emit_block(nh, sd, "embed.blocks.%d." % i, x, None, x_len, pos, cfg, cfg.embed_heads,
                           cfg.embed_cnn_module_norm, False)
        embed = nh.addLayerNorm(_norm(sd, "embed.after_norm.", 1e-12), x)
        # main encoder
        x, x_len = emit_subsampling(nh, sd, "subsampling.", xs, xs_len)
        x, pos = emit_pos_enc(nh, x, cfg.attention_dim, cfg.max_len)
        for i in range(cfg.num_blocks):
            x = emit_block(nh, sd, "blocks.%d." % i, x, embed, x_len, pos, cfg, cfg.attention_heads,
                           cfg.cnn_module_norm, True)
        x = nh.addLayerNorm(_norm(sd, "after_norm.", 1e-12), x)
        out = nh.addLinear(_linear(sd, "out_linear."), x)
        return (out, embed) if output_embed else out
