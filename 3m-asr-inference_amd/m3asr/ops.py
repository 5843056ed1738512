"""Torch-tensor front end of the C ABI: device memory and streams come from PyTorch-ROCm, every
computation is a libm3asr_hip.so call (include/m3asr.h).  No op here has a torch fallback."""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "m3asr ops need contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def _f32(t):
    assert t is None or t.dtype == torch.float32, "fp32 tensor expected"
    return _p(t)


def _i32(t):
    assert t is None or t.dtype == torch.int32, "int32 tensor expected"
    return _p(t)


# ---------------------------------------------------------------------------------------- MoE
def moe_scatter_mapping(gate_idx, num_expert, want_pos=True):
    lib = _lib.load()
    g = gate_idx.reshape(-1)
    S = g.numel()
    mapping = torch.empty(S, dtype=torch.int32, device=g.device)
    acc = torch.empty(num_expert + 1, dtype=torch.int32, device=g.device)
    pos = torch.empty(S, dtype=torch.int32, device=g.device) if want_pos else None
    check(lib.m3_moe_scatter_mapping(_i32(g), S, num_expert, _p(mapping), _p(acc), _p(pos), _stream()),
          "m3_moe_scatter_mapping")
    return mapping, acc, pos


def moe_local_scatter(x, mapping, n_rows):
    lib = _lib.load()
    S, row_bytes = x.shape[0], x[0].numel() * x.element_size()
    out = torch.zeros((n_rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    check(lib.m3_moe_local_scatter(_p(x), _i32(mapping), S, row_bytes, _p(out), _stream()), "m3_moe_local_scatter")
    return out


def moe_local_scatter_into(x, mapping, out):
    """out[mapping[s]] = x[s] for mapping[s] >= 0, into a caller-owned buffer (rows of out not hit keep their content)."""
    lib = _lib.load()
    S, row_bytes = x.shape[0], x[0].numel() * x.element_size()
    check(lib.m3_moe_local_scatter(_p(x), _i32(mapping), S, row_bytes, _p(out), _stream()), "m3_moe_local_scatter")
    return out


def ep_send_map(gate_idx, mapping, acc, world, e_loc, capacity, map_send, wire):
    """Expert-parallel send plan on the device (m3_ep_send_map): fills map_send[S] and the header rows of `wire`
    [world, 1 + capacity, D]."""
    lib = _lib.load()
    S = gate_idx.numel()
    row_bytes = wire.shape[-1] * wire.element_size()
    check(lib.m3_ep_send_map(_i32(gate_idx.reshape(-1)), _i32(mapping), _i32(acc), S, world, e_loc, capacity, _p(map_send),
                             _p(wire), row_bytes, _stream()), "m3_ep_send_map")
    return map_send


def ep_recv_gate(wire, world, e_loc, capacity, gate_recv):
    """Local expert id of every received wire row (m3_ep_recv_gate)."""
    lib = _lib.load()
    row_bytes = wire.shape[-1] * wire.element_size()
    check(lib.m3_ep_recv_gate(_p(wire), world, e_loc, capacity, row_bytes, _p(gate_recv), _stream()), "m3_ep_recv_gate")
    return gate_recv


def moe_local_gather(buf, mapping):
    lib = _lib.load()
    S = mapping.numel()
    row_bytes = buf[0].numel() * buf.element_size()
    out = torch.empty((S,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    check(lib.m3_moe_local_gather(_p(buf), _i32(mapping), S, row_bytes, _p(out), _stream()), "m3_moe_local_gather")
    return out


def moe_expert_workspace_size(S, E, D, F):
    return _lib.load().m3_moe_expert_workspace_size(S, E, D, F)


def quantize_rows_e4m3(x):
    """x (S, 512) f32 -> (xq (S, 512) uint8 holding OCP e4m3, scale (S,) f32): scale = amax / 448 per row, round to nearest even,
    saturating -- what the fused fp8 expert kernel does to its input rows (m3_quantize_rows_e4m3)."""
    lib = _lib.load()
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    S, D = x.shape
    xq = torch.empty(S, D, dtype=torch.uint8, device=x.device)
    scale = torch.empty(S, dtype=torch.float32, device=x.device)
    check(lib.m3_quantize_rows_e4m3(_p(x), x.stride(0), S, D, _p(xq), _p(scale), _stream()), "m3_quantize_rows_e4m3")
    return xq, scale


def moe_expert_ffn(x, gate_idx, w1, b1, w2, b2, gate_value=None, resid=None, alpha=1.0, ln=None, workspace=None,
                   w1_scale=None, w2_scale=None, out=None, h_scale=None, xq=None, xq_scale=None):
    """FMoEExpert: x (S,D) f32, gate_idx (S,) i32 -> y (S,D).  Optional fused epilogue (gate, residual, LayerNorm).
    The expert weights pick the kernel family: fp32; bf16 (bf16 MFMA, fp32 accumulate); e4m3 with per-row scales
    w1_scale [E,F] / w2_scale [E,D] (dequantised to bf16 at the MFMA input; with h_scale: fp8 arithmetic, activations
    quantised too -- m3_moe_expert_ffn_fp8a8; xq / xq_scale: the rows already quantised by quantize_rows_e4m3, read instead
    of x where the fused kernel applies -- m3_moe_expert_ffn_fp8a8_xq).  Biases are fp32 in every mode."""
    lib = _lib.load()
    assert w1.dtype == w2.dtype and w1.is_contiguous() and w2.is_contiguous()
    S, D = x.shape
    E, F = w1.shape[0], w1.shape[1]
    if workspace is None:
        workspace = torch.empty(max(moe_expert_workspace_size(S, E, D, F), 1), dtype=torch.uint8, device=x.device)
    y = out if out is not None else torch.empty_like(x)
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    gate = _f32(gate_value.reshape(-1) if gate_value is not None else None)
    tail = (S, E, D, F, gate, _f32(resid), float(alpha), _f32(g), _f32(b), float(eps), _p(y), _p(workspace),
            workspace.numel(), _stream())
    xi, gi = _f32(x), _i32(gate_idx.reshape(-1))
    if w1.dtype == torch.float8_e4m3fn and h_scale is not None:
        # fp8 arithmetic: activations quantised too (rows: per-row dynamic scale; H: the static scale h_scale)
        assert w1_scale is not None and w2_scale is not None, "fp8 expert weights need their per-row scales"
        if xq is not None:
            assert xq.dtype in (torch.uint8, torch.float8_e4m3fn) and xq.is_contiguous() and xq.shape == x.shape and xq_scale is not None
            check(lib.m3_moe_expert_ffn_fp8a8_xq(xi, _p(xq), _f32(xq_scale), gi, _p(w1), _f32(w1_scale), _f32(b1), _p(w2),
                                                 _f32(w2_scale), _f32(b2), float(h_scale), *tail), "m3_moe_expert_ffn_fp8a8_xq")
            return y
        check(lib.m3_moe_expert_ffn_fp8a8(xi, gi, _p(w1), _f32(w1_scale), _f32(b1), _p(w2), _f32(w2_scale), _f32(b2),
                                          float(h_scale), *tail), "m3_moe_expert_ffn_fp8a8")
    elif w1.dtype == torch.float8_e4m3fn:
        assert w1_scale is not None and w2_scale is not None, "fp8 expert weights need their per-row scales"
        check(lib.m3_moe_expert_ffn_fp8(xi, gi, _p(w1), _f32(w1_scale), _f32(b1), _p(w2), _f32(w2_scale), _f32(b2), *tail),
              "m3_moe_expert_ffn_fp8")
    elif w1.dtype == torch.bfloat16:
        check(lib.m3_moe_expert_ffn_bf16(xi, gi, _p(w1), _f32(b1), _p(w2), _f32(b2), *tail), "m3_moe_expert_ffn_bf16")
    else:
        check(lib.m3_moe_expert_ffn(xi, gi, _f32(w1), _f32(b1), _f32(w2), _f32(b2), *tail), "m3_moe_expert_ffn")
    return y


def moe_combine(rows, mapping, gate_value=None, resid=None, alpha=1.0, ln=None, out=None, out_bf16=None):
    """out[s] = LN(resid[s] + alpha * gate[s] * rows[mapping[s]]); rows are in scattered (expert-sorted) order.
    out_bf16 (optional, (S, D) bf16): also receives the result rounded to bf16 (the engine's "xb" copy)."""
    lib = _lib.load()
    S, D = mapping.numel(), rows.shape[-1]
    y = out if out is not None else torch.empty(S, D, dtype=torch.float32, device=rows.device)
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    if out_bf16 is not None:
        assert out_bf16.dtype == torch.bfloat16 and out_bf16.numel() >= S * D and out_bf16.is_contiguous()
        check(lib.m3_moe_combine_bf16(_f32(rows), _i32(mapping), _f32(gate_value.reshape(-1) if gate_value is not None else None),
                                      _f32(resid), float(alpha), _f32(g), _f32(b), float(eps), _p(y), _p(out_bf16), S, D, _stream()),
              "m3_moe_combine_bf16")
        return y
    check(lib.m3_moe_combine(_f32(rows), _i32(mapping), _f32(gate_value.reshape(-1) if gate_value is not None else None),
                             _f32(resid), float(alpha), _f32(g), _f32(b), float(eps), _p(y), S, D, _stream()),
          "m3_moe_combine")
    return y


def moe_router(embed, x, w, ln, bias=None, want_xn=True):
    """logits (S, E) = cat([embed, LayerNorm(x)]) @ w^T (+ bias), xn = LayerNorm(x); w (E, De + D) fp32, ln = (gamma, beta, eps)."""
    lib = _lib.load()
    S, De = embed.shape
    D = x.shape[1]
    E = w.shape[0]
    assert w.shape[1] == De + D and w.is_contiguous() and embed.is_contiguous() and x.is_contiguous()
    logits = torch.empty(S, E, dtype=torch.float32, device=x.device)
    xn = torch.empty_like(x) if want_xn else None
    check(lib.m3_moe_router(_f32(embed), De, De, _f32(x), D, D, _f32(w), _f32(bias), _f32(ln[0]), _f32(ln[1]), float(ln[2]),
                            _f32(xn), D, _p(logits), E, S, E, _stream()), "m3_moe_router")
    return logits, xn


def softmax_top1(logits, lens=None, rows_per_batch=0):
    lib = _lib.load()
    E = logits.shape[-1]
    l2 = logits.reshape(-1, E)
    S = l2.shape[0]
    idx = torch.empty(S, dtype=torch.int32, device=logits.device)
    val = torch.empty(S, dtype=torch.float32, device=logits.device)
    check(lib.m3_softmax_top1(_f32(l2), E, _i32(lens), rows_per_batch, S, E, _p(idx), _p(val), _stream()),
          "m3_softmax_top1")
    return val, idx


# ---------------------------------------------------------------------------------------- dense
def linear(a, w, bias=None, act=_lib.ACT_NONE, a2=None, ln=None, lens=None, rows_per_batch=0, mask_in=False,
           mask_out=False, alpha=1.0, resid=None, out=None, ln_folded=None, split_k=False, out_dtype=torch.float32,
           copy_bf16=None, copy_stats=None, ln_stats=None):
    """y = resid + alpha * mask_out(act(LN(mask_in(cat[a,a2])) @ w^T + bias)); a (M,K1), w (N,K).
    ln = (gamma, beta, eps): affine LayerNorm prologue.  ln_folded = (wsum, wbeta or None, eps): w / bias already
    contain the LayerNorm affine (plan.fold_layernorm) and the kernel normalises its output.
    bf16 activation operands (bf16 weights only): `a` may be a bf16 tensor; out_dtype=torch.bfloat16 writes y as bf16;
    copy_bf16 (M, n_out) bf16 receives an extra bf16 copy of y, copy_stats (M, n_out/128, 2) f32 its per-tile row statistics;
    ln_stats (M, parts, 2): such statistics of a bf16 `a`, which the folded LayerNorm then uses."""
    lib = _lib.load()
    M, K1 = a.shape
    N, K = w.shape
    n_out = N // 2 if act == _lib.ACT_GLU else N
    y = out if out is not None else torch.empty(M, n_out, dtype=out_dtype, device=a.device)
    d = _lib.LinearDesc()
    d.a, d.lda = a.data_ptr(), a.stride(0)
    assert a.dtype in (torch.float32, torch.bfloat16) and y.dtype in (torch.float32, torch.bfloat16)
    d.a_dtype = _lib.BF16 if a.dtype == torch.bfloat16 else _lib.F32
    d.y_dtype = _lib.BF16 if y.dtype == torch.bfloat16 else _lib.F32
    if copy_bf16 is not None:
        assert copy_bf16.dtype == torch.bfloat16 and tuple(copy_bf16.shape) == (M, n_out)
        d.y_copy_bf16, d.ld_copy = copy_bf16.data_ptr(), copy_bf16.stride(0)
    if copy_stats is not None:
        assert copy_bf16 is not None and copy_stats.dtype == torch.float32 and copy_stats.is_contiguous()
        d.y_copy_stats = copy_stats.data_ptr()
    if ln_stats is not None:
        assert ln_stats.dtype == torch.float32 and ln_stats.is_contiguous() and ln_stats.shape[0] == M
        d.ln_stats, d.ln_stat_parts = ln_stats.data_ptr(), ln_stats.shape[1]
    if a2 is not None:
        d.a2, d.lda2, d.k1 = a2.data_ptr(), a2.stride(0), K1
        assert K1 + a2.shape[1] == K
    else:
        assert K1 == K
    assert w.dtype in (torch.float32, torch.bfloat16) and w.is_contiguous()
    d.w, d.bias = w.data_ptr(), (bias.data_ptr() if bias is not None else None)
    d.weight_dtype = _lib.BF16 if w.dtype == torch.bfloat16 else _lib.F32   # bf16 weights -> bf16 MFMA, fp32 accumulate
    d.y, d.ldy = y.data_ptr(), y.stride(0)
    d.M, d.N, d.K = M, N, K
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
    if ln_folded is not None:
        d.ln_wsum, d.ln_eps = ln_folded[0].data_ptr(), float(ln_folded[2])
        if ln_folded[1] is not None:
            d.ln_wbeta = ln_folded[1].data_ptr()
    if lens is not None:
        d.len, d.rows_per_batch = lens.data_ptr(), rows_per_batch
    d.mask_in, d.mask_out = int(mask_in), int(mask_out)
    d.act, d.alpha = act, float(alpha)
    if resid is not None:
        d.resid, d.ldr = resid.data_ptr(), resid.stride(0)
    if split_k:        # deep-K / few-tile problems: split-K kernel + reduce through a scratch workspace
        need = lib.m3_linear_workspace_size(C.byref(d))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=a.device)
        check(lib.m3_linear_ws(C.byref(d), _p(ws), need, _stream()), "m3_linear_ws")
        return y
    check(lib.m3_linear(C.byref(d), _stream()), "m3_linear")
    return y


def layer_norm(x, gamma, beta, eps):
    lib = _lib.load()
    D = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.m3_layer_norm(_f32(x), _f32(gamma), _f32(beta), float(eps), _p(y), x.numel() // D, D, _stream()),
          "m3_layer_norm")
    return y


def relpos_attention(qkv, p, pos_u, pos_v, lens, B, T, H, dk, chunk=0, left_chunks=-1):
    """chunk > 0: static chunk mask (utils/mask.py:42-75): query i sees keys of its chunk and of `left_chunks` chunks to the
    left (< 0: all) -- besides the padding mask."""
    lib = _lib.load()
    D = H * dk
    out = torch.empty(B * T, D, dtype=torch.float32, device=qkv.device)
    if chunk > 0:
        check(lib.m3_relpos_attention_chunk(_f32(qkv), qkv.stride(0), _f32(p), p.stride(0), _f32(pos_u), _f32(pos_v), _i32(lens),
                                            B, T, H, dk, 1.0 / math.sqrt(dk), int(chunk), int(left_chunks), _p(out), D, _stream()),
              "m3_relpos_attention_chunk")
        return out
    check(lib.m3_relpos_attention(_f32(qkv), qkv.stride(0), _f32(p), p.stride(0), _f32(pos_u), _f32(pos_v),
                                  _i32(lens), B, T, H, dk, 1.0 / math.sqrt(dk), _p(out), D, _stream()),
          "m3_relpos_attention")
    return out


def relpos_attention_bf16(qkv, p, pos_u, pos_v, lens, B, T, H, dk, chunk=0, left_chunks=-1):
    """the same on bf16 rows: qkv (B*T, 3*H*dk) bf16 -> ctx (B*T, H*dk) bf16 (T <= 128)"""
    lib = _lib.load()
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    D = H * dk
    out = torch.empty(B * T, D, dtype=torch.bfloat16, device=qkv.device)
    check(lib.m3_relpos_attention_bf16(_p(qkv), 3 * D, _f32(p), p.stride(0), _f32(pos_u), _f32(pos_v), _i32(lens), B, T, H, dk,
                                       1.0 / math.sqrt(dk), int(chunk), int(left_chunks), _p(out), D, _stream()), "m3_relpos_attention_bf16")
    return out


def dwconv_ln_silu(z, w_kc, bias, gamma, beta, eps, B, T):
    lib = _lib.load()
    K, D = w_kc.shape
    out = torch.empty_like(z)
    check(lib.m3_dwconv_ln_silu(_f32(z), _f32(w_kc), _f32(bias), _f32(gamma), _f32(beta), float(eps), B, T, D, K,
                                _p(out), _stream()), "m3_dwconv_ln_silu")
    return out


def subsample_conv1(feat, w9c, bias, act=_lib.ACT_RELU):
    """Conv2d(1, C, 3, stride 2) on (B,T,idim) -> channel-last (B,T1,F1,C); act = ACT_RELU (fused, default) or ACT_NONE."""
    lib = _lib.load()
    B, T, idim = feat.shape
    Cc = w9c.shape[1]
    T1, F1 = (T - 3) // 2 + 1, (idim - 3) // 2 + 1
    out = torch.empty(B, T1, F1, Cc, dtype=torch.float32, device=feat.device)
    check(lib.m3_conv2d_3x3s2_first(_f32(feat), _f32(w9c), _f32(bias), B, T, idim, Cc, int(act), _p(out), _stream()),
          "m3_conv2d_3x3s2_first")
    return out


def cmvn(x, lens, mean, istd):
    lib = _lib.load()
    B, T, D = x.shape
    y = torch.empty_like(x)
    check(lib.m3_cmvn(_f32(x), _i32(lens), _f32(mean), _f32(istd), B, T, D, _p(y), _stream()), "m3_cmvn")
    return y


def log_softmax_bias(x, bias=None):
    lib = _lib.load()
    n = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.m3_log_softmax_bias(_f32(x), _f32(bias), _p(y), x.numel() // n, n, _stream()), "m3_log_softmax_bias")
    return y


# ---------------------------------------------------------------------------------------- CTC search on the logits
def ctc_greedy(logits, lens=None, blank=0):
    """encoder.py:156-180 on the device: (frame_ids (B,T), tokens (B,T) padded with -1, n_tokens (B,)) int32."""
    lib = _lib.load()
    B, T, V = logits.shape
    dev = logits.device
    ids = torch.empty(B, T, dtype=torch.int32, device=dev)
    tokens = torch.empty(B, T, dtype=torch.int32, device=dev)
    n_tokens = torch.empty(B, dtype=torch.int32, device=dev)
    ln = None if lens is None else lens.reshape(-1).contiguous()
    assert ln is None or ln.numel() == B
    check(lib.m3_ctc_greedy(_f32(logits), _i32(ln), B, T, V, int(blank), _p(ids), _p(tokens), _p(n_tokens), _stream()),
          "m3_ctc_greedy")
    return ids, tokens, n_tokens


def ctc_topk(logits, k):
    """per row log_softmax + k best (value desc, index asc): (top_logp (...,k) f32, top_idx (...,k) i32)."""
    lib = _lib.load()
    V = logits.shape[-1]
    rows = logits.numel() // V
    top_logp = torch.empty(*logits.shape[:-1], k, dtype=torch.float32, device=logits.device)
    top_idx = torch.empty(*logits.shape[:-1], k, dtype=torch.int32, device=logits.device)
    check(lib.m3_ctc_topk(_f32(logits), rows, V, int(k), _p(top_logp), _p(top_idx), _stream()), "m3_ctc_topk")
    return top_logp, top_idx


def ctc_prefix_beam_search_host(top_logp, top_idx, beam, blank=0):
    """encoder.py:232-275 over HOST (T,k) arrays of ctc_topk: [(prefix tuple, score)], best first (native host routine)."""
    import numpy as np
    lib = _lib.load()
    lp = np.ascontiguousarray(top_logp, dtype=np.float32)
    ix = np.ascontiguousarray(top_idx, dtype=np.int32)
    T, k = lp.shape
    assert ix.shape == (T, k)
    toks = np.empty((beam, max(T, 1)), dtype=np.int32)
    hlen = np.empty(beam, dtype=np.int32)
    score = np.empty(beam, dtype=np.float32)
    n = C.c_int32(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    if T == 0:
        return [(tuple(), 0.0)]
    check(lib.m3_ctc_prefix_beam_search(vp(lp), vp(ix), T, k, int(beam), int(blank), vp(toks), vp(hlen), vp(score),
                                        C.cast(C.byref(n), C.c_void_p)), "m3_ctc_prefix_beam_search")
    return [(tuple(int(v) for v in toks[i, :hlen[i]]), float(score[i])) for i in range(n.value)]


# ---------------------------------------------------------------------------------------- streaming operators
def cat_split_cache(in_cache, inp):
    """CatSplitCache plugin: (output (B, cache+input), out_cache (B, cache)); f32 or i32 rows."""
    lib = _lib.load()
    assert in_cache.dtype == inp.dtype and in_cache.element_size() == 4
    B, cd = in_cache.shape
    idim = inp.shape[1]
    assert inp.shape[0] == B
    out = torch.empty(B, cd + idim, dtype=inp.dtype, device=inp.device)
    out_cache = torch.empty(B, cd, dtype=inp.dtype, device=inp.device)
    check(lib.m3_cat_split_cache(_p(in_cache), _p(inp), B, cd, idim, _p(out), _p(out_cache), _stream()), "m3_cat_split_cache")
    return out, out_cache


def att_stream_softmax(scores, decode_frame_num, mask_idx, cache_len, scale):
    """AttStreamSoftmax plugin on scores (B, N, ld) (any leading split of N, e.g. (B, h, T, ld))."""
    lib = _lib.load()
    B, ld = scores.shape[0], scores.shape[-1]
    N = scores.numel() // max(B * ld, 1)
    out = torch.empty_like(scores)
    check(lib.m3_att_stream_softmax(_f32(scores), _i32(decode_frame_num), _i32(mask_idx), B, N, ld, int(cache_len),
                                    float(scale), _p(out), _stream()), "m3_att_stream_softmax")
    return out


def rel_positional_encoding(x, pe, scale, frame_num=None, max_offset=0):
    """RelPositionalEncoding plugin: (x * scale, pe[off:off+T] (1,T,D)[, frame_num + T]); off = frame_num[0] or 0."""
    lib = _lib.load()
    B, T, D = x.shape
    pe2 = pe.reshape(-1, D)
    y = torch.empty_like(x)
    pos = torch.empty(1, T, D, dtype=torch.float32, device=x.device)
    fn_out = None if frame_num is None else torch.empty_like(frame_num)
    check(lib.m3_rel_positional_encoding(_f32(x), _f32(pe2), pe2.shape[0], _i32(frame_num), int(max_offset), float(scale),
                                         B, T, D, _p(y), _p(pos), _p(fn_out), _stream()), "m3_rel_positional_encoding")
    return (y, pos) if frame_num is None else (y, pos, fn_out)


def subsample_conv2(x, w, bias, act=_lib.ACT_RELU):
    """Conv2d(C, C, 3, stride 2) on channel-last (B,T1,F1,C) as implicit GEMM; act = ACT_RELU (fused, default) or ACT_NONE."""
    lib = _lib.load()
    B, T1, F1, Cc = x.shape
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    out = torch.empty(B, T2, F2, Cc, dtype=torch.float32, device=x.device)
    check(lib.m3_conv2d_3x3s2(_f32(x), _f32(w), _f32(bias), B, T1, F1, Cc, int(act), _p(out), _stream()),
          "m3_conv2d_3x3s2")
    return out


# ---------------------------------------------------------------------------------------- small plugins
def att_masked_softmax(scores, lens, scale):
    lib = _lib.load()
    B, H, T1, T2 = scores.shape
    out = torch.empty_like(scores)
    check(lib.m3_att_masked_softmax(_f32(scores), _i32(lens), B, H, T1, T2, float(scale), _p(out), _stream()),
          "m3_att_masked_softmax")
    return out


def masked_fill(x, lens, fill):
    lib = _lib.load()
    B, Cc, T = x.shape
    y = torch.empty_like(x)
    check(lib.m3_masked_fill(_f32(x), _i32(lens), B, Cc, T, float(fill), _p(y), _stream()), "m3_masked_fill")
    return y


def glu(x, dim):
    lib = _lib.load()
    dim = dim % x.dim()
    outer = int(math.prod(x.shape[:dim]))
    inner = int(math.prod(x.shape[dim + 1:]))
    Cc = x.shape[dim] // 2
    shape = list(x.shape)
    shape[dim] = Cc
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    check(lib.m3_glu(_f32(x), outer, Cc, inner, _p(y), _stream()), "m3_glu")
    return y


def mask_conv2d_sample(lens, left_padding, stride):
    lib = _lib.load()
    out = torch.empty_like(lens)
    check(lib.m3_mask_conv2d_sample(_i32(lens), lens.numel(), left_padding, stride, _p(out), _stream()),
          "m3_mask_conv2d_sample")
    return out


def scale(x, s):
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.m3_scale(_f32(x), float(s), _p(y), x.numel(), _stream()), "m3_scale")
    return y


def unary(x, act):
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.m3_unary(_f32(x), _p(y), x.numel(), act, _stream()), "m3_unary")
    return y


def binary(a, b, op):
    """Broadcasting element-wise sum / prod (TensorRT ElementWise semantics: equal rank, dims 1 broadcast)."""
    lib = _lib.load()
    assert a.dim() == b.dim(), "elementwise operands must have equal rank"
    shape = [max(x, y) for x, y in zip(a.shape, b.shape)]
    nd = len(shape)

    def strides(t):
        return [0 if t.shape[i] == 1 and shape[i] != 1 else t.stride(i) for i in range(nd)]

    y = torch.empty(shape, dtype=torch.float32, device=a.device)
    arr = C.c_int64 * nd
    check(lib.m3_binary(_f32(a), _f32(b), _p(y), arr(*shape), arr(*strides(a)), arr(*strides(b)), nd, op, _stream()),
          "m3_binary")
    return y


def permute_copy(x, perm):
    """Materialised permutation (TensorRT shuffle's transpose); x may be any strided view."""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32
    nd = x.dim()
    out_shape = [x.shape[p] for p in perm]
    in_strides = [x.stride(p) for p in perm]
    y = torch.empty(out_shape, dtype=torch.float32, device=x.device)
    arr = C.c_int64 * nd
    check(lib.m3_permute(C.c_void_p(x.data_ptr()), _p(y), arr(*out_shape), arr(*in_strides), nd, _stream()), "m3_permute")
    return y


def concat_last(a, b):
    lib = _lib.load()
    da, db = a.shape[-1], b.shape[-1]
    rows = a.numel() // da
    y = torch.empty(tuple(a.shape[:-1]) + (da + db,), dtype=torch.float32, device=a.device)
    check(lib.m3_concat_last(_f32(a), da, _f32(b), db, _p(y), rows, _stream()), "m3_concat_last")
    return y


def softmax_lastdim(x):
    lib = _lib.load()
    n = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.m3_softmax(_f32(x), _p(y), x.numel() // n, n, _stream()), "m3_softmax")
    return y


def batched_matmul(a, b, transpose_b=False):
    """a (..., M, K) @ b (..., K, N) (or b (..., N, K) transposed); leading dims equal or 1 in b/a."""
    lib = _lib.load()
    if b.dim() < a.dim():          # fewer leading dims: broadcast like torch.matmul
        b = b.reshape((1,) * (a.dim() - b.dim()) + tuple(b.shape))
    elif a.dim() < b.dim():
        a = a.reshape((1,) * (b.dim() - a.dim()) + tuple(a.shape))
    M, K = a.shape[-2:]
    N = b.shape[-2] if transpose_b else b.shape[-1]
    lead = [max(x, y) for x, y in zip(a.shape[:-2], b.shape[:-2])]
    batch = int(math.prod(lead)) if lead else 1
    na, nb = int(math.prod(a.shape[:-2])), int(math.prod(b.shape[:-2]))
    if na not in (1, batch):      # partial broadcast (e.g. (1,h,..) against (B,h,..)): materialise with the copy kernel
        a = permute_copy(a.expand(tuple(lead) + tuple(a.shape[-2:])), tuple(range(a.dim())))
        na = batch
    if nb not in (1, batch):
        b = permute_copy(b.expand(tuple(lead) + tuple(b.shape[-2:])), tuple(range(b.dim())))
        nb = batch
    c = torch.empty(tuple(lead) + (M, N), dtype=torch.float32, device=a.device)
    sa = 0 if na == 1 and batch > 1 else M * K
    sb = 0 if nb == 1 and batch > 1 else b.shape[-2] * b.shape[-1]
    check(lib.m3_batched_matmul(_f32(a), _f32(b), _p(c), batch, M, N, K, sa, sb, int(transpose_b), _stream()),
          "m3_batched_matmul")
    return c


def depthwise_conv1d(x, w, bias, pad):
    lib = _lib.load()
    B, Cc, T = x.shape
    K = w.shape[-1]
    y = torch.empty(B, Cc, T + 2 * int(pad) - K + 1, dtype=torch.float32, device=x.device)
    check(lib.m3_depthwise_conv1d(_f32(x), _f32(w.reshape(Cc, K)), _f32(bias), B, Cc, T, K, pad, _p(y), _stream()),
          "m3_depthwise_conv1d")
    return y


def pad2d(x, pre, post):
    """TensorRT IPaddingLayer on the last two dims: pre = (h, w), post = (h, w) zeros."""
    lib = _lib.load()
    H, W = x.shape[-2], x.shape[-1]
    outer = x.numel() // max(H * W, 1)
    y = torch.empty(tuple(x.shape[:-2]) + (H + pre[0] + post[0], W + pre[1] + post[1]), dtype=torch.float32, device=x.device)
    check(lib.m3_pad2d(_f32(x), outer, H, W, int(pre[0]), int(post[0]), int(pre[1]), int(post[1]), _p(y), _stream()), "m3_pad2d")
    return y
