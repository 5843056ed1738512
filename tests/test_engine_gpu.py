"""GPU parity of the whole encoder engine (m3_engine_* through the C ABI) against
 (a) the committed golden vectors (reference's own forward, tests/golden/), and
 (b) the CPU oracle on fresh seeded inputs incl. ragged batches.
Tolerance: north_star asks logits within 1e-3 relative fp32; we hold rtol 1e-3 on |logit| plus a small
absolute floor for near-zero logits (atol 2e-4, ~1e-4 of the typical logit magnitude 2.4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from m3asr.config import EncoderConfig
from m3asr.engine import Engine
from m3asr.weights import make_weights
from oracle.encoder_ref import encoder_forward, sub_len

RTOL, ATOL = 1e-3, 2e-4


def _run(cfg, w, feat, feat_len, **kw):
    eng = Engine.from_state_dict(cfg, w, **kw)
    f, l = feat.cuda().contiguous(), feat_len.to(torch.int32).view(1, -1).cuda().contiguous()
    out = eng(f, l).cpu()
    return eng, out


def _check(out, want, out_len):
    valid = torch.arange(out.shape[1]).view(1, -1) < torch.as_tensor(out_len).view(-1, 1)
    err = (out - want).abs()[valid]
    bound = (ATOL + RTOL * want.abs())[valid]
    assert bool((err <= bound).all()), "max abs err %.3e, max |ref| %.3e" % (float(err.max()), float(want[valid].abs().max()))
    return float(err.max())


@pytest.mark.parametrize("name", ["tiny", "mid", "cfg2", "causal", "causal_mid"])
def test_engine_matches_golden(golden, name):
    cfg, z = golden(name)
    w = make_weights(cfg, seed=int(z["weight_seed"]))
    eng, out = _run(cfg, w, torch.from_numpy(z["feat"]), torch.from_numpy(z["feat_len"]), debug_taps=True)
    assert tuple(out.shape) == z["logits"].shape
    _check(out, torch.from_numpy(z["logits"]), z["out_len"])
    valid = np.arange(out.shape[1])[None, :] < z["out_len"][:, None]
    # routing decisions of the first layers: integer outputs, exact
    for i in range(z["gate_idx"].shape[0]):
        gi = eng.buffer("blocks.%d.gate_idx" % i, torch.int32).cpu().numpy().reshape(valid.shape)
        assert np.array_equal(gi[valid], z["gate_idx"][i][..., 0][valid])
        assert (gi[~valid] == -1).all()
    hist = np.stack([np.diff(eng.buffer("blocks.%d.acc_histogram" % i, torch.int32).cpu().numpy())
                     for i in range(cfg.num_blocks)])
    if bool(valid.all()):
        assert np.array_equal(hist, z["expert_hist"])
    if "block_out" in z.files:
        for i in range(cfg.num_blocks):
            bo = eng.buffer("blocks.%d.out" % i).cpu().view(out.shape[0], out.shape[1], -1)
            _check(bo, torch.from_numpy(z["block_out"][i]), z["out_len"])


def test_engine_ragged_batch_vs_oracle():
    cfg = EncoderConfig(num_blocks=2, embed_blocks=2)
    w = make_weights(cfg, seed=5)
    lengths = [206, 57, 333, 120]
    g = torch.Generator().manual_seed(9)
    feat = torch.rand(len(lengths), max(lengths), cfg.input_dim, generator=g)
    fl = torch.tensor(lengths, dtype=torch.int32)
    want = encoder_forward(w, cfg, feat, fl)
    eng, out = _run(cfg, w, feat, fl)
    _check(out, want, sub_len(fl.long()))


def test_engine_graph_replay_and_fold_are_bit_identical():
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg, seed=2)
    feat = torch.rand(1, 206, cfg.input_dim, generator=torch.Generator().manual_seed(1)).cuda()
    fl = torch.tensor([[206]], dtype=torch.int32).cuda()
    eng = Engine.from_state_dict(cfg, w, fold_pos_proj=False)
    eager = eng(feat, fl).clone()
    for _ in range(3):
        eng.forward(use_graph=True)
    eng.stream.synchronize()
    assert torch.equal(eng._bound[2], eager)
    eng2 = Engine.from_state_dict(cfg, w, fold_pos_proj=True)
    assert torch.equal(eng2(feat, fl), eager)
    assert eng2.num_kernels() < eng.num_kernels()


@pytest.mark.parametrize("B,wdt", [(1, "f32"), (3, "f32"), (2, "bf16")])
def test_engine_forked_embed_branch_is_bit_identical(B, wdt):
    """fork_embed: in the captured graph the embed encoder runs as a second branch beside the main subsampler and block 0 up to
    its router (own scratch buffers, joined where the embedding is first read).  Same kernels on the same data: the logits
    equal the one-chain graph's and the eager run's bit for bit, also on packed ragged rows and after replays."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=2, weight_dtype=wdt)
    w = make_weights(cfg, seed=6)
    T = 206
    feat = torch.rand(B, T, cfg.input_dim, generator=torch.Generator().manual_seed(B)).cuda()
    fl = torch.tensor([[T - 37 * i for i in range(B)]], dtype=torch.int32).cuda()
    chain = Engine.from_state_dict(cfg, w, fork_embed=False)
    eager = chain(feat, fl).clone()
    chain.forward(use_graph=True)
    chain.stream.synchronize()
    assert torch.equal(chain._bound[2], eager)
    forked = Engine.from_state_dict(cfg, w, fork_embed=True)
    assert torch.equal(forked(feat, fl), eager)                       # eager run of the forked plan (one chain, own scratch)
    for _ in range(3):
        forked._bound[2].zero_()
        forked.forward(use_graph=True)
        forked.stream.synchronize()
        assert torch.equal(forked._bound[2], eager)
    # the embed branch's scratch (a single chain that interleaves the two encoders' GEMMs -- see `pairs` below -- has it too)
    assert forked.workspace_size(B, T) >= chain.workspace_size(B, T)
    # the same kernels -- plus, on unpacked rows, the separate length kernel: in one chain the first conv1 forms the subsampled
    # lengths on its way, in a forked graph the main branch reads them while the embed branch's conv1 runs beside it
    # (and, since round 4, minus the launches the single chain saves by running a GEMM of the embed encoder and one of the main
    # encoder's first block as ONE launch -- stages named "a+b"; the forked plan has two streams for that)
    pairs = sum("+" in n for n in chain.stage_names())
    assert forked.num_kernels() - chain.num_kernels() == pairs + (0 if chain.packed_rows() else 1)
    if B == 1 and wdt == "f32":
        assert pairs >= 6, chain.stage_names()


def test_fused_and_staged_route_paths_agree():
    """fuse_route=1 (router + SoftmaxTopK + ScatterMapping in one launch, LayerNorm applied by the expert kernel) vs the
    staged path (router GEMM writes xn, separate gate+index): same routing decisions, logits within fp32 noise."""
    cfg = EncoderConfig(num_blocks=3, embed_blocks=1)
    w = make_weights(cfg, seed=8)
    feat = torch.rand(2, 206, cfg.input_dim, generator=torch.Generator().manual_seed(3)).cuda()
    fl = torch.tensor([[206, 131]], dtype=torch.int32).cuda()
    a = Engine.from_state_dict(cfg, w, fuse_route=True)
    b = Engine.from_state_dict(cfg, w, fuse_route=False, packed_rows=False)      # same (padded) row layout for the taps
    ya, yb = a(feat, fl).clone(), b(feat, fl).clone()
    assert "blocks.0.moe_route" in a.stage_names() and "blocks.0.moe_router" in b.stage_names()
    # (the staged path of short fp32 inputs routes inside the expert launch since round 4: no index launch there either)
    assert not any(n.endswith("moe_gate_index") for n in b.stage_names())
    for i in range(cfg.num_blocks):
        assert torch.equal(a.buffer("blocks.%d.gate_idx" % i, torch.int32), b.buffer("blocks.%d.gate_idx" % i, torch.int32))
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)


def test_split_route_path_agrees_with_staged():
    """fuse_route=2: embed half of every router in one GEMM per forward, x half as a K = D GEMM with norm_ff folded in,
    norm_ff applied by the expert kernel while gathering (xn never materialised): same routing, logits within fp32 noise;
    also on the golden cfg-2 fixture (18 layers, 32 experts)."""
    cfg = EncoderConfig(num_blocks=3, embed_blocks=1)
    w = make_weights(cfg, seed=8)
    feat = torch.rand(2, 206, cfg.input_dim, generator=torch.Generator().manual_seed(3)).cuda()
    fl = torch.tensor([[206, 131]], dtype=torch.int32).cuda()
    a = Engine.from_state_dict(cfg, w, fuse_route=2)
    b = Engine.from_state_dict(cfg, w, fuse_route=0, packed_rows=False)          # same (padded) row layout for the taps
    ya, yb = a(feat, fl).clone(), b(feat, fl).clone()
    assert "router_e_all" in a.stage_names() and "router_e_all" not in b.stage_names()
    for i in range(cfg.num_blocks):
        assert torch.equal(a.buffer("blocks.%d.gate_idx" % i, torch.int32), b.buffer("blocks.%d.gate_idx" % i, torch.int32))
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)


def test_split_route_matches_golden_cfg2(golden):
    cfg, z = golden("cfg2")
    w = make_weights(cfg, seed=int(z["weight_seed"]))
    eng, out = _run(cfg, w, torch.from_numpy(z["feat"]), torch.from_numpy(z["feat_len"]), fuse_route=2)
    _check(out, torch.from_numpy(z["logits"]), z["out_len"])
    valid = np.arange(out.shape[1])[None, :] < z["out_len"][:, None]
    for i in range(z["gate_idx"].shape[0]):
        gi = eng.buffer("blocks.%d.gate_idx" % i, torch.int32).cpu().numpy().reshape(valid.shape)
        assert np.array_equal(gi[valid], z["gate_idx"][i][..., 0][valid])


def test_shape_cache_replays_graphs_across_alternating_shapes():
    """Serving with a few length buckets: Engine.infer keeps static I/O buffers per shape and the native engine parks the
    stage list + captured hipGraph of each bound shape (LRU), so alternating shapes captures each graph once."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg, seed=6)
    eng = Engine.from_state_dict(cfg, w)
    ref = Engine.from_state_dict(cfg, w)
    g = torch.Generator().manual_seed(4)
    shapes = [(1, 206), (2, 120), (1, 77)]
    feats = {s_: torch.rand(s_[0], s_[1], cfg.input_dim, generator=g) for s_ in shapes}
    lens = {s_: torch.tensor([s_[1] - 9 * i for i in range(s_[0])], dtype=torch.int32) for s_ in shapes}
    want = {s_: ref(feats[s_].cuda(), lens[s_].view(1, -1).cuda()).clone() for s_ in shapes}
    for rnd_ in range(4):
        for s_ in shapes:
            out = eng.infer(feats[s_], lens[s_])
            assert torch.equal(out, want[s_]), (rnd_, s_)
    assert eng.num_captures() == len(shapes)            # 12 forwards, 3 captures
    # fresh input values in the same static buffers: still a replay, new result
    f2 = torch.rand(1, 206, cfg.input_dim, generator=g)
    out2 = eng.infer(f2, lens[(1, 206)]).clone()
    assert eng.num_captures() == len(shapes)
    assert torch.equal(out2, ref(f2.cuda(), lens[(1, 206)].view(1, -1).cuda()))


def test_one_shared_workspace_for_every_shape_like_a_trt_execution_context():
    """A C-ABI caller may hand m3_engine_forward ONE max-size workspace for every shape (a TensorRT execution context owns
    one device-memory block).  Shapes A, B, A: the second A revives a parked binding on an exact (shape, pointers) match
    after B has overwritten the whole workspace -- the folded positional projection must not have lived there (it is
    engine-owned memory).  Checked against fresh engines, eager and graph replay, fp32 bit-identical."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg, seed=6)
    eng = Engine.from_state_dict(cfg, w, fold_pos_proj=True)
    g = torch.Generator().manual_seed(12)
    shapes = [(1, 206), (2, 333), (1, 206), (3, 64), (2, 333), (1, 206)]
    big = torch.empty(max(eng.workspace_size(b, t) for b, t in shapes), dtype=torch.uint8, device="cuda")
    eng._workspace = lambda B, T: big                      # every shape binds the same caller-owned block
    io = {}
    for b, t in set(shapes):
        feat = torch.rand(b, t, cfg.input_dim, generator=g).cuda()
        fl = torch.tensor([[t - 11 * i for i in range(b)]], dtype=torch.int32).cuda()
        io[(b, t)] = (feat, fl, torch.empty(eng.output_shape(b, t), device="cuda"),
                      Engine.from_state_dict(cfg, w, fold_pos_proj=False)(feat, fl).clone())
    for use_graph in (False, True):
        for b, t in shapes:
            feat, fl, out, want = io[(b, t)]
            big.fill_(0xFF)                                # whatever the previous shape left behind
            torch.cuda.synchronize()
            eng.forward(feat, fl, out, use_graph=use_graph)
            eng.stream.synchronize()
            assert torch.equal(out, want), (use_graph, b, t)


def test_python_side_shape_caches_are_bounded():
    """Engine.infer on unbucketed lengths: workspaces / static I/O buffers are kept LRU-bounded (max_shapes), not one per
    length ever seen; results stay right after evictions."""
    cfg = EncoderConfig.tiny()
    w = make_weights(cfg, seed=3)
    eng = Engine.from_state_dict(cfg, w, max_shapes=3)
    ref = Engine.from_state_dict(cfg, w)
    g = torch.Generator().manual_seed(2)
    for T in [40, 41, 42, 43, 44, 40, 45, 41]:
        feat = torch.randn(1, T, cfg.input_dim, generator=g)
        fl = torch.tensor([T], dtype=torch.int32)
        out = eng.infer(feat, fl)
        assert len(eng._ws) <= 3 and len(eng._static) <= 3
        assert torch.equal(out, ref(feat.cuda(), fl.view(1, -1).cuda()))


def test_engine_longest_profile_length():
    """The reference profiles its engine up to 6100 frames (builder.py:58-64): long-batch kernels (LDS-tiled GEMMs, grouped
    tiled expert FFN, row-parallel top-1) and 32-bit index arithmetic at S = 3 x 1524 rows, ragged lengths."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg, seed=1)
    feat = torch.rand(3, 6100, cfg.input_dim, generator=torch.Generator().manual_seed(0))
    fl = torch.tensor([6100, 3001, 777], dtype=torch.int32)
    want = encoder_forward(w, cfg, feat, fl)
    eng, out = _run(cfg, w, feat, fl)
    _check(out, want, sub_len(fl.long()))


@pytest.mark.parametrize("kw", [dict(router_with_bias=True), dict(keep_expert_output=True),
                                dict(cnn_module_norm="batch_norm", embed_cnn_module_norm="batch_norm"),
                                dict(router_with_bias=True, keep_expert_output=True, cnn_module_norm="batch_norm",
                                     embed_cnn_module_norm="batch_norm", attention_heads=4, embed_heads=8, num_experts=16)])
def test_engine_config_variants_vs_oracle(kw):
    """The optional pieces of the reference's encoder_conf / moe_conf (router bias, un-gated expert output:
    positionwise_feed_forward.py:169-180,258-262; BatchNorm instead of LayerNorm in the conv module: convolution.py:60-75,
    folded into the depthwise conv by the plan packer) and other head / expert counts, ragged batch."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1, **kw)
    w = make_weights(cfg, seed=13)
    feat = torch.rand(3, 150, cfg.input_dim, generator=torch.Generator().manual_seed(5))
    fl = torch.tensor([150, 96, 33], dtype=torch.int32)
    want = encoder_forward(w, cfg, feat, fl)
    eng, out = _run(cfg, w, feat, fl)
    _check(out, want, sub_len(fl.long()))


@pytest.mark.parametrize("wdt", ["f32", "bf16"])
def test_engine_batch_with_an_empty_utterance(wdt):
    """An utterance shorter than the subsampling receptive field (feat_len 4 -> 0 output frames) inside a batch: all of its
    rows are padding (their values are don't-care, attention over zero keys is NaN there); the other utterances must be
    unaffected -- nothing but the attention keys and the depthwise-conv taps of the SAME utterance couples rows."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1, weight_dtype=wdt)
    cfg32 = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg32, seed=21)
    feat = torch.rand(3, 120, cfg.input_dim, generator=torch.Generator().manual_seed(7))
    fl = torch.tensor([120, 4, 77], dtype=torch.int32)
    eng, out = _run(cfg, w, feat, fl)
    assert int(eng.buffer("lens", torch.int32)[1]) == 0 == int(sub_len(4))
    keep = [0, 2]
    eng2, out2 = _run(cfg, w, feat[keep].contiguous(), fl[keep])
    valid = torch.arange(out.shape[1]).view(1, -1) < sub_len(fl[keep].long()).view(-1, 1)
    assert bool(torch.isfinite(out[keep][valid]).all())
    if wdt == "f32":
        want = encoder_forward(w, cfg32, feat[keep], fl[keep])
        _check(out[keep], want, sub_len(fl[keep].long()))
    assert torch.allclose(out[keep][valid], out2[valid], rtol=1e-4, atol=1e-4)      # same rows with or without the empty one


def test_engine_degenerate_short_utterance_follows_the_plugin_length_rule():
    """feat_len 5/6: the MaskConv2dSample plugin's truncating division gives ONE output frame (the trainer's mask slicing
    would give none; oracle.sub_len documents the difference) -- the engine's lens follow the plugin and the one frame
    matches the oracle."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg, seed=22)
    feat = torch.rand(3, 64, cfg.input_dim, generator=torch.Generator().manual_seed(8))
    fl = torch.tensor([5, 64, 6], dtype=torch.int32)
    eng, out = _run(cfg, w, feat, fl)
    assert eng.buffer("lens", torch.int32).tolist() == [1, 15, 1] == sub_len(fl.long()).tolist()
    _check(out, encoder_forward(w, cfg, feat, fl), sub_len(fl.long()))


@pytest.mark.parametrize("lengths", [[206], [333, 64, 400, 206, 120, 399, 250, 380, 57, 390, 395, 222, 111, 345]])
def test_engine_64_experts_vs_oracle(lengths):
    """BASELINE configs[4] has 64 experts: one utterance (slab expert kernel, fused gate+index) and a long ragged batch
    (moe_top1 + grouped tiled GEMMs), fp32 against the oracle; routing taps exact."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1, num_experts=64)
    w = make_weights(cfg, seed=64)
    feat = torch.rand(len(lengths), max(lengths), cfg.input_dim, generator=torch.Generator().manual_seed(64))
    fl = torch.tensor(lengths, dtype=torch.int32)
    eng, out = _run(cfg, w, feat, fl)
    taps = {}
    want = encoder_forward(w, cfg, feat, fl, taps)
    _check(out, want, sub_len(fl.long()))
    valid = torch.arange(out.shape[1]).view(1, -1) < sub_len(fl.long()).view(-1, 1)
    for i in range(cfg.num_blocks):
        gi = eng.rows_padded("blocks.%d.gate_idx" % i, torch.int32, fill=-1).cpu().view(valid.shape)
        assert torch.equal(gi[valid], taps["blocks.%d.gate_idx" % i].view(valid.shape)[valid].to(torch.int32))
        hist = torch.diff(eng.buffer("blocks.%d.acc_histogram" % i, torch.int32).cpu())
        assert hist.numel() == 64 and int(hist.sum()) == int(valid.sum())


def test_engine_longest_profile_shape():
    """The longest input of the reference's TensorRT profile (builder.py:58-64: up to 6100 frames): index arithmetic,
    workspace carving and the long-batch kernels at S = 3 x 1524 rows, fp32 to the usual bar and bf16 weights to 5e-2."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg, seed=1)
    B, T = 3, 6100
    feat = torch.rand(B, T, cfg.input_dim, generator=torch.Generator().manual_seed(0))
    fl = torch.tensor([6100, 3001, 777], dtype=torch.int32)
    want = encoder_forward(w, cfg, feat, fl)
    out_len = sub_len(fl.long())
    assert out_len.tolist() == [1524, 749, 193]
    eng, out = _run(cfg, w, feat, fl)
    _check(out, want, out_len)
    eng16, out16 = _run(EncoderConfig(**{**cfg.__dict__, "weight_dtype": "bf16"}), w, feat, fl)
    valid = torch.arange(out.shape[1]).view(1, -1) < out_len.view(-1, 1)
    assert bool(torch.isfinite(out16[valid]).all())
    # free-running routing: a frame whose top-1 margin is inside bf16 noise takes another expert and differs by a whole
    # expert FFN, so the bar is on the bulk of the frames, not on the maximum
    err16 = (out16 - want).abs().amax(-1)[valid] / float(want.abs()[valid].max())
    print("bf16 at 3x6100: median %.2e, 99%% %.2e, max %.2e" % (float(err16.median()), float(err16.quantile(0.99)), float(err16.max())))
    assert float(err16.quantile(0.9)) < 5e-2


PACK_CASES = [
    ("two_short", EncoderConfig(num_blocks=2, embed_blocks=1), [120, 77]),                       # 49 rows: skinny kernels
    ("with_degenerate", EncoderConfig(num_blocks=2, embed_blocks=1), [64, 5, 206, 7, 33]),       # 1-frame utterances
    ("all_full", EncoderConfig(num_blocks=1, embed_blocks=1), [97, 97, 97]),                     # no padding at all
    ("long_ragged", EncoderConfig(num_blocks=2, embed_blocks=1),
     [400, 57, 206, 333, 120, 399, 250, 64, 380, 390, 395, 222, 111, 345, 50, 500]),             # 1984 padded rows: tiled
    ("e64_ragged", EncoderConfig(num_blocks=1, embed_blocks=1, num_experts=64), [333, 64, 400, 206, 120, 399, 250, 380]),
]


@pytest.mark.parametrize("wdt", ["f32", "bf16"])
@pytest.mark.parametrize("name,cfg,lengths", PACK_CASES)
def test_engine_packed_rows_equal_padded_rows(name, cfg, lengths, wdt):
    """Ragged batches run the blocks on the packed valid frames (m3_engine_config.packed_rows, automatic for B > 1).  Every
    kernel of a block is row-wise except attention and the depthwise conv, which see exactly the frames of their own
    utterance either way (plus the padded layout's constant pad frame for the conv taps past the end), so the valid logits
    must be BIT-IDENTICAL to the padded engine's; frames past an utterance's end come back as zeros."""
    cfgd = EncoderConfig(**{**cfg.__dict__, "weight_dtype": wdt})
    w = make_weights(cfg, seed=31)
    feat = torch.rand(len(lengths), max(lengths), cfg.input_dim, generator=torch.Generator().manual_seed(len(lengths)))
    fl = torch.tensor(lengths, dtype=torch.int32)
    packed, out_p = _run(cfgd, w, feat, fl)
    padded, out_q = _run(cfgd, w, feat, fl, packed_rows=False)
    assert packed.packed_rows() and not padded.packed_rows()
    out_len = sub_len(fl.long())
    assert packed.buffer("row0", torch.int32).cpu().tolist() == [0] + torch.cumsum(out_len, 0).tolist()
    valid = torch.arange(out_p.shape[1]).view(1, -1) < out_len.view(-1, 1)
    assert torch.equal(out_p[valid], out_q[valid])
    assert bool((out_p[~valid] == 0).all())
    if wdt == "f32":
        _check(out_p, encoder_forward(w, cfg, feat, fl), out_len)
    # replay (hipGraph) with other lengths in the same buffers: the row plan is recomputed on the device
    fl2 = torch.tensor(list(reversed(lengths)), dtype=torch.int32)
    fl2[0] = max(lengths)
    out2 = packed.infer(feat.cuda(), fl2.view(1, -1).cuda()).cpu()
    want2 = padded.infer(feat.cuda(), fl2.view(1, -1).cuda()).cpu()
    valid2 = torch.arange(out2.shape[1]).view(1, -1) < sub_len(fl2.long()).view(-1, 1)
    assert torch.equal(out2[valid2], want2[valid2]) and bool((out2[~valid2] == 0).all())


def test_engine_rejects_bad_input():
    from m3asr._lib import M3Error
    cfg = EncoderConfig.tiny()
    eng = Engine.from_state_dict(cfg, make_weights(cfg, seed=0))
    with pytest.raises(M3Error):
        eng(torch.rand(1, 5, cfg.input_dim).cuda(), torch.tensor([[5]], dtype=torch.int32).cuda())   # T < 7
    w = make_weights(cfg, seed=0)
    del w["blocks.0.norm_ff.weight"]
    with pytest.raises(KeyError):
        Engine.from_state_dict(cfg, w)


def test_front_and_back_end_cmvn_prior_logsoftmax(tmp_path):
    """global CMVN fused into the first conv, -log(prior) and log-softmax on the output."""
    from m3asr.plan import pack_weights, add_front_back_end, read_cmvn_stats
    from m3asr import ops
    from oracle import encoder_ref as ref
    cfg = EncoderConfig(num_blocks=1, embed_blocks=1)
    w = make_weights(cfg, seed=6)
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(2, 90, cfg.input_dim, generator=g) * 3.0 + 5.0
    fl = torch.tensor([[90, 61]], dtype=torch.int32)
    # Kaldi-style stats file: [sum.. count ; sumsq.. 0]
    n, x2 = 1000.0, torch.randn(1000, cfg.input_dim, generator=g).double() * 3.0 + 5.0
    stats = torch.zeros(2, cfg.input_dim + 1, dtype=torch.float64)
    stats[0, :-1], stats[0, -1], stats[1, :-1] = x2.sum(0), n, (x2 * x2).sum(0)
    path = str(tmp_path / "cmvn.txt")
    with open(path, "w") as f:
        f.write(" [\n  " + " ".join("%.10g" % v for v in stats[0]) + "\n  " + " ".join("%.10g" % v for v in stats[1]) + " ]\n")
    mean, istd = read_cmvn_stats(path)
    assert torch.allclose(mean, x2.mean(0).float(), atol=1e-4) and torch.allclose(istd, (1 / x2.std(0, unbiased=False)).float(), rtol=1e-4)
    prior = torch.rand(cfg.output_dim, generator=g) + 0.05
    bias = -torch.log(prior / prior.sum())
    close = lambda a, b: bool(((a - b).abs() <= 3e-4 + 1e-3 * b.abs()).all())
    normed = ref.cmvn(feat, None, mean, istd)
    logits = encoder_forward(w, cfg, normed, fl)
    valid = (torch.arange(logits.shape[1]).view(1, -1) < sub_len(fl.view(-1).long()).view(-1, 1))
    # standalone ops
    assert torch.allclose(ops.cmvn(feat.cuda(), fl.view(-1).cuda(), mean.cuda(), istd.cuda()).cpu(),
                          ref.cmvn(feat, fl.view(-1), mean, istd), atol=1e-5)
    assert torch.allclose(ops.log_softmax_bias(logits.cuda(), bias.cuda()).cpu(), ref.score(logits, True, bias), atol=2e-5)
    for log_softmax in (False, True):
        c2 = EncoderConfig(**{**cfg.__dict__, "log_softmax_out": log_softmax})
        packed = add_front_back_end(pack_weights(w, c2), c2, cmvn=(mean, istd), output_bias=bias)
        out = Engine(c2, packed)(feat.cuda(), fl.cuda()).cpu()
        want = ref.score(logits, log_softmax, bias)
        assert close(out[valid], want[valid]), float((out - want).abs()[valid].max())
        # the same front / back end on a 16-bit plan (bf16 weights): within the bf16 tolerance of the fp32 score
        c16 = EncoderConfig(**{**c2.__dict__, "weight_dtype": "bf16"})
        p16 = add_front_back_end(pack_weights(w, c16), c16, cmvn=(mean, istd), output_bias=bias)
        assert p16["out_linear.ln.bias"].dtype == torch.float32 and p16["cmvn.mean"].dtype == torch.float32
        out16 = Engine(c16, p16)(feat.cuda(), fl.cuda()).cpu()
        assert float((out16 - want).abs()[valid].max()) < 5e-2 * float(want.abs()[valid].max())
