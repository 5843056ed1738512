"""Static chunk mask of the streaming encoders.

CPU tier: the oracle's restatement (oracle/encoder_ref.py:chunk_mask) against tests/golden/chunk_mask.npz -- masks produced by
the reference's OWN utils/mask.py:subsequent_chunk_mask / add_optional_chunk_mask (generator: oracle/gen_golden_chunk.py).
GPU tier: the HIP attention kernels (fp32 core, bf16 core) with (chunk, left_chunks) against torch attention under the FIXTURE's
masks, and the engine with EncoderConfig.static_chunk_size against the oracle's forward.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
from oracle import encoder_ref  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "chunk_mask.npz"))
SQUARE = [tuple(int(v) for v in row) for row in GOLD["square_cases"]]
N_BATCH = int(GOLD["n_batch"])


def _square(n):
    size = SQUARE[n][0]
    return torch.from_numpy(np.unpackbits(GOLD["square_%d" % n])[:size * size].reshape(size, size).astype(bool))


def _batch(n):
    lens = [int(v) for v in GOLD["batch_%d_lens" % n]]
    chunk, left = [int(v) for v in GOLD["batch_%d_params" % n]]
    B, L = len(lens), max(lens)
    m = np.unpackbits(GOLD["batch_%d" % n])[:B * L * L].reshape(B, L, L).astype(bool)
    return lens, chunk, left, torch.from_numpy(m)


# ------------------------------------------------------------------------------------------------ CPU tier
def test_fixture_holds_the_reference_docstring_example():
    assert SQUARE[0] == (4, 2, -1)                                     # utils/mask.py:64-69
    assert _square(0).int().tolist() == [[1, 1, 0, 0], [1, 1, 0, 0], [1, 1, 1, 1], [1, 1, 1, 1]]


@pytest.mark.parametrize("n", range(len(SQUARE)))
def test_oracle_chunk_mask_matches_reference(n):
    size, chunk, left = SQUARE[n]
    assert torch.equal(encoder_ref.chunk_mask(size, chunk, left), _square(n))


@pytest.mark.parametrize("n", range(N_BATCH))
def test_oracle_padding_and_chunk_mask_matches_reference(n):
    lens, chunk, left, want = _batch(n)
    L = max(lens)
    pad = torch.arange(L).view(1, 1, L) < torch.tensor(lens).view(-1, 1, 1)
    got = pad & encoder_ref.chunk_mask(L, chunk, left).unsqueeze(0) if chunk > 0 else pad.expand(-1, L, -1)
    assert torch.equal(got, want)


def _attention_under_mask(qkv, p, u, v, mask, B, T, H, dk, dtype=torch.float64):
    """RelPositionMultiHeadedAttention under an explicit (B, T, T) visibility mask (layer/attention.py:199-239: masked_fill(-inf),
    softmax, masked_fill(0))."""
    D = H * dk
    q, k, vv = [t.to(dtype).view(B, T, H, dk) for t in qkv.view(B, T, 3 * D).split(D, -1)]
    pp = p.to(dtype).view(1, T, H, dk)
    ac = torch.matmul((q + u.to(dtype)).transpose(1, 2), k.permute(0, 2, 3, 1))
    bd = torch.matmul((q + v.to(dtype)).transpose(1, 2), pp.permute(0, 2, 3, 1))
    hide = ~mask.view(B, 1, T, T)
    att = torch.softmax(((ac + bd) / math.sqrt(dk)).masked_fill(hide, -float("inf")), -1)
    att = torch.nan_to_num(att, nan=0.0).masked_fill(hide, 0.0)
    return torch.matmul(att, vv.transpose(1, 2)).transpose(1, 2).reshape(B, T, D)


def _rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("n", range(N_BATCH))
def test_oracle_attention_applies_the_reference_mask(n):
    lens, chunk, left, mask = _batch(n)
    B, T, H, dk = len(lens), max(lens), 2, 8
    D = H * dk
    x, pos = _rnd(B, T, D, seed=1), _rnd(1, T, D, seed=2)
    w = {"a.linear_q.weight": _rnd(D, D, seed=3, scale=0.3), "a.linear_k.weight": _rnd(D, D, seed=4, scale=0.3),
         "a.linear_v.weight": _rnd(D, D, seed=5, scale=0.3), "a.linear_out.weight": torch.eye(D), "a.linear_pos.weight": _rnd(D, D, seed=6, scale=0.3),
         "a.pos_bias_u": _rnd(H, dk, seed=7, scale=0.3), "a.pos_bias_v": _rnd(H, dk, seed=8, scale=0.3)}
    for nme in ("q", "k", "v", "out"):
        w["a.linear_%s.bias" % nme] = torch.zeros(D)
    got = encoder_ref.rel_pos_mha(x, pos, torch.tensor(lens), w, "a.", H, chunk, left)
    qkv = torch.cat([x @ w["a.linear_q.weight"].t(), x @ w["a.linear_k.weight"].t(), x @ w["a.linear_v.weight"].t()], -1)
    p = (pos @ w["a.linear_pos.weight"].t()).view(T, D)
    want = _attention_under_mask(qkv.view(B * T, 3 * D), p, w["a.pos_bias_u"], w["a.pos_bias_v"], mask, B, T, H, dk, torch.float32)
    assert torch.allclose(got, want, atol=2e-5, rtol=2e-5)


# ------------------------------------------------------------------------------------------------ GPU tier
ATT_CASES = [(n, H, dk) for n in range(N_BATCH) for (H, dk) in ((8, 64), (4, 128))]


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,dk", ATT_CASES)
def test_hip_attention_fp32_under_reference_mask(n, H, dk):
    from m3asr import ops
    lens, chunk, left, mask = _batch(n)
    B, T, D = len(lens), max(lens), H * dk
    qkv, p = _rnd(B * T, 3 * D, seed=1), _rnd(T, D, seed=2)
    u, v = _rnd(H, dk, seed=3, scale=0.3), _rnd(H, dk, seed=4, scale=0.3)
    L = torch.tensor(lens, dtype=torch.int32)
    out = ops.relpos_attention(qkv.cuda(), p.cuda(), u.cuda(), v.cuda(), L.cuda(), B, T, H, dk, chunk=chunk, left_chunks=left)
    want = _attention_under_mask(qkv, p, u, v, mask, B, T, H, dk).float()
    got = out.cpu().view(B, T, D)
    assert bool(torch.isfinite(got).all())
    assert torch.allclose(got, want, atol=3e-5, rtol=3e-5), float((got - want).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,dk", ATT_CASES)
def test_hip_attention_bf16_under_reference_mask(n, H, dk):
    from m3asr import ops
    lens, chunk, left, mask = _batch(n)
    B, T, D = len(lens), max(lens), H * dk
    qkv, p = _rnd(B * T, 3 * D, seed=1).to(torch.bfloat16), _rnd(T, D, seed=2)
    u, v = _rnd(H, dk, seed=3, scale=0.3), _rnd(H, dk, seed=4, scale=0.3)
    L = torch.tensor(lens, dtype=torch.int32)
    out = ops.relpos_attention_bf16(qkv.cuda(), p.cuda(), u.cuda(), v.cuda(), L.cuda(), B, T, H, dk, chunk=chunk, left_chunks=left)
    want = _attention_under_mask(qkv.float(), p.to(torch.bfloat16).float(), u, v, mask, B, T, H, dk).float()
    got = out.float().cpu().view(B, T, D)
    assert bool(torch.isfinite(got).all())
    valid = torch.arange(T).view(1, -1) < L.view(-1, 1)
    assert torch.allclose(got[valid], want[valid], atol=2e-2, rtol=2e-2), float((got[valid] - want[valid]).abs().max())


def _engine_case(chunk, left, lens, weight_dtype):
    import dataclasses
    from m3asr.config import EncoderConfig
    from m3asr.engine import Engine
    from m3asr.weights import make_weights
    cfg = EncoderConfig(num_blocks=2, embed_blocks=2, static_chunk_size=chunk, num_decoding_left_chunks=left)
    w = make_weights(cfg, seed=11)
    feat = torch.rand(len(lens), max(lens), cfg.input_dim, generator=torch.Generator().manual_seed(5))
    flen = torch.tensor(lens, dtype=torch.int32)
    eng = Engine.from_state_dict(dataclasses.replace(cfg, weight_dtype=weight_dtype), w)
    got = eng(feat.cuda().contiguous(), flen.view(1, -1).cuda().contiguous()).float().cpu()
    B, Tp = got.shape[0], got.shape[1]
    forced = None
    if weight_dtype != "f32":          # 16-bit arithmetic may flip a near-tie of the router: teacher-forced routing (tests/test_bf16_gpu.py)
        forced = {"blocks.%d.gate_idx" % i: eng.rows_padded("blocks.%d.gate_idx" % i, torch.int32, fill=-1).cpu().view(B, Tp, 1).clone()
                  for i in range(cfg.num_blocks)}
    want = encoder_ref.encoder_forward(w, cfg, feat, flen, route_override=forced)
    full = encoder_ref.encoder_forward(w, dataclasses.replace(cfg, static_chunk_size=0), feat, flen, route_override=forced)
    valid = torch.arange(Tp).view(1, -1) < encoder_ref.sub_len(flen.long()).view(-1, 1)
    return got, want, full, valid


@pytest.mark.gpu
@pytest.mark.parametrize("chunk,left", [(4, -1), (4, 1), (16, 0), (3, 2), (200, -1)])
def test_engine_with_static_chunk_mask_matches_oracle(chunk, left):
    """Whole encoder (embed blocks, MoE blocks, output layer) with the chunk mask in every attention vs the oracle's forward."""
    got, want, full, valid = _engine_case(chunk, left, [206, 150, 97, 333], "f32")
    err = (got - want).abs()[valid]
    assert bool((err <= (2e-4 + 1e-3 * want.abs())[valid]).all()), float(err.max())
    far = float((full - want).abs()[valid].max())
    assert (far > 0.02) if chunk < 80 else (far == 0.0), far            # the mask matters -- unless one chunk covers the utterance


@pytest.mark.gpu
@pytest.mark.parametrize("chunk,left,lens", [
    (16, 1, [206, 150, 97, 333]),                                                                   # fp32 attention core on bf16 weights
    (16, 1, [400, 57, 206, 333, 120, 399, 250, 64, 380, 390, 395, 222, 111, 345, 400, 301]),        # S = 1584: bf16 attention core
    (5, -1, [400, 57, 206, 333, 120, 399, 250, 64, 380, 390, 395, 222, 111, 345, 400, 301])])
def test_engine_bf16_with_static_chunk_mask(chunk, left, lens):
    got, want, full, valid = _engine_case(chunk, left, lens, "bf16")
    scale = float(want.abs()[valid].max())
    err = float((got - want).abs()[valid].max()) / scale
    assert err < 2e-2, err                                            # BF16_REL of tests/test_bf16_gpu.py
    assert float((full - want).abs()[valid].max()) / scale > 3 * err    # the mask matters
