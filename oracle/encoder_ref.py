"""Plain-torch fp32 restatement of the 3M-ASR Conformer-MoE encoder forward (CPU oracle).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference file it
follows (paths relative to /root/reference).  State-dict key names are the reference's own
(``Net.state_dict()`` of trainer_3m_fix/model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py).

Conventions the reference leaves undefined and this oracle (and the HIP path) pin down:
  * padded frames (t >= len'[b]) get gate_idx = -1, gate_value = 0 and a zero expert output
    (the reference's SoftmaxTopK plugin never writes those rows, softmax_topk_kernel.cu:40);
  * the order of rows inside one expert's segment is the stable order (token index ascending),
    one admissible outcome of the reference's atomics (fmoe_expert_kernel.cu:33-37);
  * LayerNorm uses eps as PyTorch does (the reference plugin drops it, layer_norm_kernel.cu:55;
    the stated parity target is PyTorch, infer_helper.py:93).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .moe_index import moe_index_ref


def _cdiv_trunc(a, b):
    """C integer division (truncation toward zero), for python ints and int tensors alike."""
    if isinstance(a, torch.Tensor):
        return torch.div(a, b, rounding_mode="trunc")
    return int(a / b)


def sub_len(l):
    """MaskConv2dSample x2: (l - left_padding - 1)/stride + 1 with left_padding=2, stride=2, in C `int`
    arithmetic as the plugin computes it (TRTAPI++/plugin/mask_conv2d_sample_plugin/mask_conv2d_sample_kernel.cu:34-35).
    The trainer's mask slicing (trainer_3m_fix/layer/subsampling.py:119-137) gives the same value for every
    l >= 7 and for l in {3, 4}; for the degenerate l in {5, 6} the plugin's truncating division yields 1 frame
    where the mask slicing yields 0 -- the plugin is the path being replaced, so it is the one followed."""
    l = _cdiv_trunc(l - 3, 2) + 1
    return _cdiv_trunc(l - 3, 2) + 1


def cmvn(feat, lens, mean, istd):
    """Global CMVN as the reference's unfinished CmvnPlugin states it (TRTAPI++/plugin/incomplete_plugin/cmvn_plugin/
    cmvn_plugin.cu:17-34): out = (in - mean[d]) * var[d] on frames t < len[b], other frames untouched."""
    B, T, _ = feat.shape
    y = (feat - mean.view(1, 1, -1)) * istd.view(1, 1, -1)
    if lens is None:
        return y
    valid = (torch.arange(T).view(1, T) < lens.view(B, 1)).unsqueeze(-1)
    return torch.where(valid, y, feat)


def score(logits, log_softmax=False, output_bias=None):
    """Back end sketched in builder.py:77-88: optional log_softmax, then + (-log prior)."""
    y = torch.log_softmax(logits, dim=-1) if log_softmax else logits
    return y + output_bias.view(1, 1, -1) if output_bias is not None else y


def subsample(feat, w, p):
    """Conv2dSubsampling4.forward (trainer_3m_fix/layer/subsampling.py:103-145):
    (B,T,idim) -> (B,1,T,idim) -> conv3x3 s2 + ReLU -> conv3x3 s2 + ReLU -> (B,T',C*F') -> Linear."""
    x = feat.unsqueeze(1)
    x = F.relu(F.conv2d(x, w[p + "conv.0.weight"], w[p + "conv.0.bias"], stride=2))
    x = F.relu(F.conv2d(x, w[p + "conv.2.weight"], w[p + "conv.2.bias"], stride=2))
    b, c, t, f = x.shape
    x = x.transpose(1, 2).contiguous().view(b, t, c * f)
    return F.linear(x, w[p + "out.0.weight"], w[p + "out.0.bias"])


def positional_table(t, d):
    """PositionalEncoding.__init__ (trainer_3m_fix/layer/positional_encoding.py:40-48):
    pe[pos,2i]=sin(pos*exp(-2i*ln(1e4)/d)), pe[pos,2i+1]=cos(...), positions from 0."""
    pe = torch.zeros(t, d)
    position = torch.arange(0, t, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def rel_pos_enc(x):
    """RelPositionalEncoding plugin (rel_positional_encoding_kernel.cu:62-81;
    positional_encoding.py:101-129): returns (x*sqrt(D), pe[:, :T']); pe is NOT added to x."""
    d = x.shape[-1]
    return x * math.sqrt(d), positional_table(x.shape[1], d)


def layer_norm(x, w, p, eps):
    return F.layer_norm(x, (x.shape[-1],), w[p + "weight"], w[p + "bias"], eps)


def ffn(x, w, p):
    """PositionwiseFeedForward.forward (layer/positionwise_feed_forward.py:79-88): hard-coded SiLU."""
    h = F.linear(x, w[p + "w_1.weight"], w[p + "w_1.bias"])
    h = h * torch.sigmoid(h)
    return F.linear(h, w[p + "w_2.weight"], w[p + "w_2.bias"])


def chunk_mask(size, chunk_size, num_left_chunks=-1):
    """subsequent_chunk_mask (utils/mask.py:42-75), restated: query i sees keys
    [max((i // chunk - left) * chunk, 0), min((i // chunk + 1) * chunk, size)); all left chunks when left < 0.
    Pinned by tests/golden/chunk_mask.npz (generated from the reference's own function, oracle/gen_golden_chunk.py)."""
    i = torch.arange(size).view(-1, 1)
    j = torch.arange(size).view(1, -1)
    c = i // chunk_size
    start = torch.zeros_like(c) if num_left_chunks < 0 else torch.clamp((c - num_left_chunks) * chunk_size, min=0)
    ending = torch.clamp((c + 1) * chunk_size, max=size)
    return (j >= start) & (j < ending)


def rel_pos_mha(x, pos_emb, lens, w, p, h, static_chunk_size=0, num_left_chunks=-1):
    """RelPositionMultiHeadedAttention.forward (layer/attention.py:320-384) +
    forward_attention_trt (:199-239) + AttMaskedSoftmax (common.cuh:264-360):
    scores = ((q+u)k^T + (q+v)p^T) / sqrt(dk)  -- NO rel_shift; softmax over keys j < len[b],
    padded keys get probability 0.  static_chunk_size > 0: additionally the static chunk mask of
    add_optional_chunk_mask (utils/mask.py:127-134): masks & subsequent_chunk_mask(T, chunk, left); a query row with no
    visible key (a padded frame) gets zeros."""
    B, T, D = x.shape
    dk = D // h
    q = F.linear(x, w[p + "linear_q.weight"], w[p + "linear_q.bias"]).view(B, T, h, dk)
    k = F.linear(x, w[p + "linear_k.weight"], w[p + "linear_k.bias"]).view(B, T, h, dk)
    v = F.linear(x, w[p + "linear_v.weight"], w[p + "linear_v.bias"]).view(B, T, h, dk)
    pp = F.linear(pos_emb, w[p + "linear_pos.weight"]).view(1, T, h, dk)
    q_u = (q + w[p + "pos_bias_u"].view(1, 1, h, dk)).transpose(1, 2)
    q_v = (q + w[p + "pos_bias_v"].view(1, 1, h, dk)).transpose(1, 2)
    ac = torch.matmul(q_u, k.permute(0, 2, 3, 1))
    bd = torch.matmul(q_v, pp.permute(0, 2, 3, 1))
    scores = (ac + bd) * (1.0 / math.sqrt(dk))
    key_pad = torch.arange(T).view(1, 1, 1, T) >= lens.view(B, 1, 1, 1)
    if static_chunk_size > 0:
        key_pad = key_pad | ~chunk_mask(T, static_chunk_size, num_left_chunks).view(1, 1, T, T)
    attn = torch.softmax(scores.masked_fill(key_pad, -float("inf")), dim=-1).masked_fill(key_pad, 0.0)
    attn = torch.nan_to_num(attn, nan=0.0)              # rows with no visible key
    ctx = torch.matmul(attn, v.transpose(1, 2))
    ctx = ctx.transpose(1, 2).contiguous().view(B, T, D)
    return F.linear(ctx, w[p + "linear_out.weight"], w[p + "linear_out.bias"])


def conv_module(x, lens, w, p, kernel, norm, causal=False):
    """ConvolutionModule.forward (layer/convolution.py:83-167): transpose, masked_fill(0) on
    padded frames, pw-conv D->2D, GLU(dim=1), depthwise conv k (pad (k-1)/2), LayerNorm over
    channels (eps 1e-5, nn.LayerNorm default; or eval BatchNorm1d), SiLU, pw-conv D->D,
    masked_fill(0), transpose back.
    causal (convolution.py:43-49): lorder = k - 1 zero frames are padded on the LEFT of the module's
    input, in front of pointwise_conv1 (:118-123), and the depthwise conv has no padding of its own."""
    B, T, D = x.shape
    pad = torch.arange(T).view(1, 1, T) >= lens.view(B, 1, 1)
    z = x.transpose(1, 2).masked_fill(pad, 0.0)
    if causal:
        z = F.pad(z, (kernel - 1, 0), "constant", 0.0)
    z = F.conv1d(z, w[p + "pointwise_conv1.weight"], w[p + "pointwise_conv1.bias"])
    z = F.glu(z, dim=1)
    z = F.conv1d(z, w[p + "depthwise_conv.weight"], w[p + "depthwise_conv.bias"],
                 padding=0 if causal else (kernel - 1) // 2, groups=D)
    if norm == "layer_norm":
        z = F.layer_norm(z.transpose(1, 2), (D,), w[p + "norm.weight"], w[p + "norm.bias"], 1e-5).transpose(1, 2)
    else:
        z = F.batch_norm(z, w[p + "norm.running_mean"], w[p + "norm.running_var"],
                         w[p + "norm.weight"], w[p + "norm.bias"], False, 0.0, 1e-5)
    z = z * torch.sigmoid(z)
    z = F.conv1d(z, w[p + "pointwise_conv2.weight"], w[p + "pointwise_conv2.bias"])
    return z.masked_fill(pad, 0.0).transpose(1, 2)


def softmax_top1_tree(logits_row):
    """Exact arg-max rule of SoftmaxAndTop1KernelSmall (softmax_topk_kernel.cu:55-64): a stride
    tree over a power-of-two width with strict '<' (on ties the lower slot of each pair wins)."""
    vals = [float(v) for v in logits_row]
    idx = list(range(len(vals)))
    stride = len(vals) >> 1
    while stride > 0:
        for t in range(stride):
            if vals[t] < vals[t + stride]:
                vals[t] = vals[t + stride]
                idx[t] = idx[t + stride]
        stride >>= 1
    return idx[0]


def softmax_topk(logits, lens):
    """SoftmaxTopK plugin (softmax_topk_kernel.cu:26-120): idx = argmax, value = 1/sum(exp(x-max)).
    Rows t >= len[b] are pinned to idx=-1, value=0 (undefined in the reference)."""
    B, T, E = logits.shape
    m, idx = logits.max(dim=-1)
    # torch.max picks the first maximal index; re-resolve exact ties with the reference's tree rule
    ties = (logits == m.unsqueeze(-1)).sum(-1) > 1
    if bool(ties.any()):
        for b, t in zip(*np.nonzero(ties.numpy())):
            idx[b, t] = softmax_top1_tree(logits[b, t].tolist())
    value = 1.0 / torch.exp(logits - m.unsqueeze(-1)).sum(-1)
    valid = torch.arange(T).view(1, T) < lens.view(B, 1)
    idx = torch.where(valid, idx, torch.full_like(idx, -1)).to(torch.int32)
    value = torch.where(valid, value, torch.zeros_like(value))
    return value.unsqueeze(-1), idx.unsqueeze(-1)


def fmoe_expert(x, gate_idx, w1, b1, w2, b2):
    """FMoEExpert plugin (fmoe_expert_plugin.cpp:36-142): scatter rows by expert, per expert
    H = SiLU(X W1[e]^T + b1[e]), Y = H W2[e]^T + b2[e], gather back.  Rows with gate_idx < 0
    are dropped and produce 0.  Returns (y, mapping, acc_histogram)."""
    B, T, D = x.shape
    E = w1.shape[0]
    g = gate_idx.reshape(-1).numpy().astype(np.int32)
    mapping, acc = moe_index_ref(g, E)
    xf = x.reshape(-1, D)
    n_valid = int(acc[E])
    buf = torch.zeros(max(n_valid, 1), D)
    sel = torch.from_numpy(np.nonzero(mapping >= 0)[0])
    mp = torch.from_numpy(mapping[mapping >= 0].astype(np.int64))
    buf[mp] = xf[sel]                                           # local_scatter
    out = torch.zeros_like(buf)
    for e in range(E):
        lo, hi = int(acc[e]), int(acc[e + 1])
        if hi > lo:
            hid = F.linear(buf[lo:hi], w1[e], b1[e])
            hid = hid * torch.sigmoid(hid)
            out[lo:hi] = F.linear(hid, w2[e], b2[e])
    y = torch.zeros_like(xf)
    y[sel] = out[mp]                                            # local_gather
    return y.view(B, T, D), mapping, acc


def moe_ffn(x, embed, lens, w, p, cfg, taps=None, tag="", route_override=None):
    """LocalFmoeCatEmbedFeedForward.forward (layer/positionwise_feed_forward.py:209-265):
    router_in = cat([embed, x]); logits = router_in @ router_weights (+bias); top-1 softmax gate;
    expert FFN; output * gate_value unless keep_expert_output."""
    router_in = torch.cat([embed, x], dim=-1)
    logits = torch.matmul(router_in, w[p + "router_weights"].unsqueeze(0))
    if (p + "router_bias") in w:
        logits = logits + w[p + "router_bias"]
    gate_value, gate_idx = softmax_topk(logits, lens)
    if route_override is not None and (tag + "gate_idx") in route_override:
        # teacher-forced routing (low-precision tests): take the expert choice from the run under test and the softmax
        # probability of THAT expert, so numeric error is measured without the discrete effect of a flipped arg-max
        gate_idx = route_override[tag + "gate_idx"].view(gate_idx.shape).to(torch.int32)
        prob = torch.softmax(logits, dim=-1)
        gate_value = torch.gather(prob, -1, gate_idx.clamp(min=0).long())
        gate_value = torch.where(gate_idx >= 0, gate_value, torch.zeros_like(gate_value))
    y, mapping, acc = fmoe_expert(x, gate_idx, w[p + "experts.w_1.weight"], w[p + "experts.w_1.bias"],
                                  w[p + "experts.w_2.weight"], w[p + "experts.w_2.bias"])
    if taps is not None:
        taps[tag + "router_logits"] = logits
        taps[tag + "gate_idx"] = gate_idx
        taps[tag + "gate_value"] = gate_value
        taps[tag + "mapping"] = torch.from_numpy(mapping)
        taps[tag + "acc_histogram"] = torch.from_numpy(acc)
        taps[tag + "expert_out"] = y
    if not cfg.keep_expert_output:
        y = y * gate_value
    return y


def conformer_block(x, embed, lens, pos_emb, w, p, cfg, heads, kernel, norm, moe, taps=None, tag="", route_override=None):
    """FmoeConformerLayer.forward (layer/fmoe_transformer.py:72-170) when moe=True,
    ConformerEncoderLayer.forward (layer/transformer.py:179-275) otherwise; ff_scale = 0.5,
    all block LayerNorms eps=1e-12 (fmoe_transformer.py:54-65)."""
    eps = 1e-12
    x = x + 0.5 * ffn(layer_norm(x, w, p + "norm_ff_macaron.", eps), w, p + "feed_forward_macaron.")
    if taps is not None:
        taps[tag + "after_macaron"] = x
    x = x + rel_pos_mha(layer_norm(x, w, p + "norm_mha.", eps), pos_emb, lens, w, p + "self_attn.", heads,
                        getattr(cfg, "static_chunk_size", 0), getattr(cfg, "num_decoding_left_chunks", -1))
    if taps is not None:
        taps[tag + "after_mha"] = x
    causal = bool(getattr(cfg, "causal" if moe else "embed_causal", False))
    x = x + conv_module(layer_norm(x, w, p + "norm_conv.", eps), lens, w, p + "conv_module.", kernel, norm, causal)
    if taps is not None:
        taps[tag + "after_conv"] = x
    xn = layer_norm(x, w, p + "norm_ff.", eps)
    if moe:
        y = moe_ffn(xn, embed, lens, w, p + "feed_forward.", cfg, taps, tag, route_override)
    else:
        y = ffn(xn, w, p + "feed_forward.")
    x = x + 0.5 * y
    x = layer_norm(x, w, p + "norm_final.", eps)
    if taps is not None:
        taps[tag + "out"] = x
    return x


def encoder_forward(w, cfg, feat, feat_len, taps=None, route_override=None):
    """Net.forward (model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py:198-234) with the
    embed encoder of model/conformer_embed_domain_acc.py:149-181.
    feat (B,T,idim) f32, feat_len (1,B) or (B,) int -> logits (B,T',V)."""
    feat = feat.float()
    lens = sub_len(feat_len.reshape(-1).to(torch.int64))
    with torch.no_grad():
        # embed encoder (dense conformer, own weights)
        x = subsample(feat, w, "embed.subsampling.")
        x, pos = rel_pos_enc(x)
        for i in range(cfg.embed_blocks):
            x = conformer_block(x, None, lens, pos, w, "embed.blocks.%d." % i, cfg, cfg.embed_heads,
                                cfg.cnn_module_kernel, cfg.embed_cnn_module_norm, False,
                                taps, "embed.%d." % i)
        embed = layer_norm(x, w, "embed.after_norm.", 1e-12)
        if taps is not None:
            taps["embed"] = embed
        # main MoE encoder
        x = subsample(feat, w, "subsampling.")
        if taps is not None:
            taps["subsample"] = x
        x, pos = rel_pos_enc(x)
        for i in range(cfg.num_blocks):
            x = conformer_block(x, embed, lens, pos, w, "blocks.%d." % i, cfg, cfg.attention_heads,
                                cfg.cnn_module_kernel, cfg.cnn_module_norm, True, taps, "blocks.%d." % i, route_override)
        x = layer_norm(x, w, "after_norm.", 1e-12)
        return F.linear(x, w["out_linear.weight"], w["out_linear.bias"])
