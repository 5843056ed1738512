"""Synthetic, name-keyed, per-tensor-seeded weights with the reference's state_dict layout.

The reference's checkpoint is not shipped (README.md:6,13), so every run uses
random-init weights of the reference architecture.  Key names and shapes are those of
``Net.state_dict()`` of model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py (main
encoder) with the ``embed.`` sub-tree of model/conformer_embed_domain_acc.py.  Each tensor is
drawn from its own generator seeded by (seed, crc32(name)), so the dict is reproducible
independently of construction order and identical on every machine.

Expert weight layout follows fmoe/layers.py:34-38: ``w_1.weight [E, F, D]``, ``w_1.bias [E, F]``,
``w_2.weight [E, D, F]``, ``w_2.bias [E, D]``; router ``[D+embed_dim, E]``
(positionwise_feed_forward.py:134-135, zero-initialised there -> randomised here, otherwise
every token routes to expert 0).
"""
import math
import zlib
from collections import OrderedDict

import torch


def _gen(seed, name):
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 0x9E3779B1 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFFFFFF)
    return g


def _uniform(shape, bound, seed, name):
    return torch.empty(shape, dtype=torch.float32).uniform_(-bound, bound, generator=_gen(seed, name))


def _normal(shape, std, seed, name, mean=0.0):
    return torch.empty(shape, dtype=torch.float32).normal_(mean, std, generator=_gen(seed, name))


def encoder_param_shapes(cfg, ep_slice=True):
    """Ordered {name: shape} of the encoder state_dict (concat_linear / after_norm_6/12 /
    embed.out_linear, which the forward never reads, are included for layout parity)."""
    D, F, E, V = cfg.attention_dim, cfg.hidden_units, cfg.num_experts, cfg.output_dim
    K = cfg.cnn_module_kernel
    sh = OrderedDict()

    def subsampling(p, d):
        sh[p + "conv.0.weight"] = (d, 1, 3, 3)
        sh[p + "conv.0.bias"] = (d,)
        sh[p + "conv.2.weight"] = (d, d, 3, 3)
        sh[p + "conv.2.bias"] = (d,)
        sh[p + "out.0.weight"] = (d, d * cfg.sub_freq)
        sh[p + "out.0.bias"] = (d,)

    def norm(p, d):
        sh[p + "weight"] = (d,)
        sh[p + "bias"] = (d,)

    def attn(p, d, h):
        sh[p + "pos_bias_u"] = (h, d // h)
        sh[p + "pos_bias_v"] = (h, d // h)
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            sh[p + n + ".weight"] = (d, d)
            sh[p + n + ".bias"] = (d,)
        sh[p + "linear_pos.weight"] = (d, d)

    def ffn(p, d, f):
        sh[p + "w_1.weight"] = (f, d)
        sh[p + "w_1.bias"] = (f,)
        sh[p + "w_2.weight"] = (d, f)
        sh[p + "w_2.bias"] = (d,)

    def convmod(p, d, norm_type="layer_norm"):
        sh[p + "pointwise_conv1.weight"] = (2 * d, d, 1)
        sh[p + "pointwise_conv1.bias"] = (2 * d,)
        sh[p + "depthwise_conv.weight"] = (d, 1, K)
        sh[p + "depthwise_conv.bias"] = (d,)
        norm(p + "norm.", d)
        if norm_type == "batch_norm":                  # nn.BatchNorm1d buffers (eval mode, convolution.py:60-75)
            sh[p + "norm.running_mean"] = (d,)
            sh[p + "norm.running_var"] = (d,)
        sh[p + "pointwise_conv2.weight"] = (d, d, 1)
        sh[p + "pointwise_conv2.bias"] = (d,)

    def block_tail(p, d):
        for n in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final"):
            norm(p + n + ".", d)
        sh[p + "concat_linear.weight"] = (d, 2 * d)
        sh[p + "concat_linear.bias"] = (d,)

    De = cfg.embed_dim
    subsampling("embed.subsampling.", De)
    norm("embed.after_norm.", De)
    for i in range(cfg.embed_blocks):
        p = "embed.blocks.%d." % i
        attn(p + "self_attn.", De, cfg.embed_heads)
        ffn(p + "feed_forward.", De, cfg.embed_linear_units)
        ffn(p + "feed_forward_macaron.", De, cfg.embed_linear_units)
        convmod(p + "conv_module.", De, cfg.embed_cnn_module_norm)
        block_tail(p, De)
    sh["embed.out_linear.weight"] = (V, De)
    sh["embed.out_linear.bias"] = (V,)

    subsampling("subsampling.", D)
    norm("after_norm.", D)
    norm("after_norm_6.", D)
    norm("after_norm_12.", D)
    for i in range(cfg.num_blocks):
        p = "blocks.%d." % i
        attn(p + "self_attn.", D, cfg.attention_heads)
        sh[p + "feed_forward.router_weights"] = (D + De, E)
        if cfg.router_with_bias:
            sh[p + "feed_forward.router_bias"] = (E,)
        sh[p + "feed_forward.experts.w_1.weight"] = (E, F, D)
        sh[p + "feed_forward.experts.w_1.bias"] = (E, F)
        sh[p + "feed_forward.experts.w_2.weight"] = (E, D, F)
        sh[p + "feed_forward.experts.w_2.bias"] = (E, D)
        ffn(p + "feed_forward_macaron.", D, F)
        convmod(p + "conv_module.", D, cfg.cnn_module_norm)
        block_tail(p, D)
    sh["out_linear.weight"] = (V, D)
    sh["out_linear.bias"] = (V,)
    return sh


_UNUSED = ("concat_linear", "after_norm_6", "after_norm_12", "embed.out_linear")


def make_weights(cfg, seed=0, router_std=0.5, skip_unused=True):
    """Deterministic synthetic state_dict (CPU fp32).  Init law:
    Linear/Conv U(+-1/sqrt(fan_in)); expert w xavier-uniform gain 0.5, expert b N(0,0.1);
    router N(0, router_std); LayerNorm gamma U(0.5,1.5), beta N(0,0.1) (randomised so a
    dropped affine shows up in parity tests); pos_bias xavier-uniform."""
    sd = OrderedDict()
    for name, shape in encoder_param_shapes(cfg).items():
        if skip_unused and any(u in name for u in _UNUSED):
            continue
        leaf = name.rsplit(".", 1)[-1]
        if "experts" in name:
            if leaf == "weight":
                bound = 0.5 * math.sqrt(6.0 / (shape[1] + shape[2]))
                t = _uniform(shape, bound, seed, name)
            else:
                t = _normal(shape, 0.1, seed, name)
        elif leaf == "router_weights":
            t = _normal(shape, router_std, seed, name)
        elif leaf == "router_bias":
            t = _normal(shape, 0.1, seed, name)
        elif leaf in ("pos_bias_u", "pos_bias_v"):
            t = _uniform(shape, math.sqrt(6.0 / (shape[0] + shape[1])), seed, name)
        elif leaf == "running_mean":
            t = _normal(shape, 0.5, seed, name)
        elif leaf == "running_var":
            t = _uniform(shape, 0.5, seed, name) + 1.0
        elif ".norm" in name or name.startswith("after_norm") or "after_norm" in name:
            if leaf == "weight":
                t = _uniform(shape, 0.5, seed, name) + 1.0
            else:
                t = _normal(shape, 0.1, seed, name)
        else:
            # Linear / Conv: fan_in = prod(shape[1:]) for the weight; the bias uses its layer's fan_in
            wname = name[: -len(leaf)] + "weight"
            wshape = encoder_param_shapes_cache(cfg)[wname]
            fan_in = 1
            for s in wshape[1:]:
                fan_in *= s
            t = _uniform(shape, 1.0 / math.sqrt(fan_in), seed, name)
        sd[name] = t
    return sd


_shape_cache = {}


def encoder_param_shapes_cache(cfg):
    key = cfg.to_json()
    if key not in _shape_cache:
        _shape_cache[key] = encoder_param_shapes(cfg)
    return _shape_cache[key]


def count_params(cfg):
    n = 0
    for shape in encoder_param_shapes(cfg).values():
        k = 1
        for s in shape:
            k *= s
        n += k
    return n
