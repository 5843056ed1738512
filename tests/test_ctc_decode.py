"""CTC searches after the encoder and the streaming operators.

Golden: tests/golden/ctc_decode.npz = what the reference's own `ctc_greedy_search` / `ctc_prefix_beam_search`
(trainer_3m_fix/model/encoder.py:156-275) return for synthetic score matrices (oracle/gen_golden_ctc.py, run in the build
container).  CPU tests pin the oracle restatement and the library's HOST beam-search routine against it; the gpu tests run
the device kernels (argmax + collapse, log-softmax + top-k) and the whole chain against the same fixture and the oracle.
Bar: token ids / hypotheses bit-exact; scores to 1e-4 absolute (the reference sums float32 log-probs in float64; the device
log-softmax differs from torch's by a few ulp of float32).  The streaming operators have no reference fixture (their CUDA
sources cannot be built here, no reference test uses them): the oracle follows the kernel text -> "parity unpinned".
"""
import os

import numpy as np
import pytest
import torch

from oracle import ctc_decode as ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ctc_decode.npz")


@pytest.fixture(scope="module")
def z():
    return np.load(GOLDEN, allow_pickle=False)


def _golden_hyps(z, name):
    n = int(z[name + "_n"][0])
    toks, ln, sc = z[name + "_hyp_tokens"], z[name + "_hyp_len"], z[name + "_hyp_score"]
    return [(tuple(int(v) for v in toks[i, :ln[i]]), float(sc[i])) for i in range(n)]


def _same_hyps(got, want, tol=1e-4):
    assert [p for p, _ in got] == [p for p, _ in want]
    np.testing.assert_allclose([s for _, s in got], [s for _, s in want], rtol=0, atol=tol)


# ------------------------------------------------------------------------------------------------ CPU: oracle + host routine
def test_oracle_greedy_matches_reference(z):
    hyps = ref.ctc_greedy_search(z["greedy_logits"], z["greedy_lens"], blank=0)
    for b, h in enumerate(hyps):
        assert h == z["greedy_tokens"][b, :z["greedy_n"][b]].tolist()
    assert [len(h) for h in hyps] == z["greedy_n"].tolist()


def test_oracle_prefix_beam_matches_reference(z):
    for name in z["beam_cases"]:
        name = str(name)
        blank, beam = (int(v) for v in z[name + "_meta"])
        _same_hyps(ref.ctc_prefix_beam_search(z[name + "_logits"], beam, blank), _golden_hyps(z, name))


def test_host_prefix_beam_routine_matches_reference(z):
    """m3_ctc_prefix_beam_search is a host function of the library (no device work): fed with the oracle's top-k it must
    reproduce the reference's n-best lists."""
    from m3asr import ops
    for name in z["beam_cases"]:
        name = str(name)
        blank, beam = (int(v) for v in z[name + "_meta"])
        lp, ix = ref.topk_desc(ref.log_softmax(z[name + "_logits"]), beam)
        _same_hyps(ops.ctc_prefix_beam_search_host(lp, ix, beam, blank), _golden_hyps(z, name))


def test_host_prefix_beam_routine_edge_cases():
    from m3asr import ops
    from m3asr._lib import M3Error
    # one frame, blank best: the empty prefix wins, then the single symbols
    lp = np.log(np.array([[0.6, 0.3, 0.1]], dtype=np.float32))
    ix = np.array([[0, 2, 1]], dtype=np.int32)
    got = ops.ctc_prefix_beam_search_host(lp, ix, 3, 0)
    _same_hyps(got, ref.prefix_beam_search_topk(lp, ix, 3, 0), 1e-6)
    assert got[0][0] == () and got[1][0] == (2,)
    # all-blank utterance; repeated symbol with and without a separating blank
    for seq in ([0, 0, 0, 0], [1, 1, 0, 1, 1], [2, 2, 2]):
        x = np.full((len(seq), 4), -4.0, dtype=np.float32)
        x[np.arange(len(seq)), seq] = 4.0
        lp, ix = ref.topk_desc(ref.log_softmax(x), 3)
        _same_hyps(ops.ctc_prefix_beam_search_host(lp, ix, 3, 0), ref.prefix_beam_search_topk(lp, ix, 3, 0), 1e-6)
    assert ops.ctc_prefix_beam_search_host(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), 3, 0) == [((), 0.0)]
    with pytest.raises(M3Error):
        ops.ctc_prefix_beam_search_host(lp, ix, 0, 0)          # beam 0


def test_oracle_prefix_beam_random_vs_host_routine():
    from m3asr import ops
    rng = np.random.default_rng(5)
    for T, V, beam in ((50, 1434, 10), (17, 9, 9), (80, 64, 3)):
        x = rng.normal(0, 2.0, (T, V)).astype(np.float32)
        lp, ix = ref.topk_desc(ref.log_softmax(x), beam)
        _same_hyps(ops.ctc_prefix_beam_search_host(lp, ix, beam, 0), ref.prefix_beam_search_topk(lp, ix, beam, 0), 1e-5)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_ctc_greedy_kernel_matches_reference(z):
    from m3asr import ops
    logits = torch.from_numpy(z["greedy_logits"]).cuda()
    lens = torch.from_numpy(z["greedy_lens"]).cuda()
    ids, tokens, n = ops.ctc_greedy(logits, lens, blank=0)
    assert torch.equal(ids.cpu().long(), torch.from_numpy(z["greedy_logits"]).argmax(-1))
    assert n.cpu().tolist() == z["greedy_n"].tolist()
    assert np.array_equal(tokens.cpu().numpy(), z["greedy_tokens"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,V,blank", [(1, 50, 1434, 0), (16, 124, 1434, 0), (3, 200, 77, 5), (64, 124, 1434, 0), (2, 1, 3, 0)])
def test_ctc_greedy_kernel_vs_oracle(B, T, V, blank):
    from m3asr import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, T, V, generator=g)
    # make blanks and repeats frequent, and plant exact ties (the first maximum must win)
    fav = torch.randint(0, min(V, 6), (B, T), generator=g)
    x.scatter_(2, fav.unsqueeze(-1), 6.0)
    x[:, ::7, blank] = 9.0
    if V > 40:
        x[:, 3::11, 17] = 9.5
        x[:, 3::11, 33] = 9.5
    lens = torch.randint(0, T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    ids, tokens, n = ops.ctc_greedy(x.cuda(), lens.cuda(), blank)
    assert torch.equal(ids.cpu().long(), x.argmax(-1))
    want = ref.ctc_greedy_search(x.numpy(), lens.numpy(), blank)
    assert n.cpu().tolist() == [len(h) for h in want]
    tok = tokens.cpu().numpy()
    for b, h in enumerate(want):
        assert tok[b, :len(h)].tolist() == h and (tok[b, len(h):] == -1).all()
    # lens = None means every frame
    _, tok2, n2 = ops.ctc_greedy(x.cuda(), None, blank)
    want2 = ref.ctc_greedy_search(x.numpy(), [T] * B, blank)
    assert n2.cpu().tolist() == [len(h) for h in want2]


@pytest.mark.gpu
@pytest.mark.parametrize("rows,V,k", [(50, 1434, 10), (1984, 1434, 10), (7, 12, 12), (33, 100, 1), (5, 64, 20)])
def test_ctc_topk_kernel(rows, V, k):
    from m3asr import ops
    g = torch.Generator().manual_seed(rows + V)
    x = torch.randn(rows, V, generator=g) * 3
    if V >= 64:
        x[:, 5] = x[:, 60]          # a tie inside the candidate set now and then: index order decides
    lp, ix = ops.ctc_topk(x.cuda(), k)
    wlp, wix = ref.topk_desc(torch.log_softmax(x, -1).numpy(), k)
    wv, wi = ref.topk_desc(x.numpy(), k)                      # selection is made on the raw scores
    assert np.array_equal(ix.cpu().numpy(), wi)
    np.testing.assert_allclose(lp.cpu().numpy(), np.take_along_axis(torch.log_softmax(x, -1).numpy(), wi.astype(np.int64), -1),
                               rtol=0, atol=2e-6)
    tv, ti = torch.topk(torch.log_softmax(x, -1), k)          # torch's own top-k agrees wherever values are distinct
    distinct = (tv[:, 1:] != tv[:, :-1]).all(-1) if k > 1 else torch.ones(rows, dtype=torch.bool)
    assert np.array_equal(ix.cpu().numpy()[distinct.numpy()], ti.numpy()[distinct.numpy()].astype(np.int32))


@pytest.mark.gpu
def test_prefix_beam_search_chain_matches_reference(z):
    """device log-softmax + top-k -> host recursion == the reference's n-best on the golden score matrices."""
    from m3asr.decode import CtcDecoder
    for name in z["beam_cases"]:
        name = str(name)
        blank, beam = (int(v) for v in z[name + "_meta"])
        dec = CtcDecoder(engine=None, blank_idx=blank)
        got = dec.prefix_beam_from_logits(torch.from_numpy(z[name + "_logits"]).cuda()[None], beam)
        _same_hyps(got, _golden_hyps(z, name))


@pytest.mark.gpu
def test_decoder_on_engine_logits():
    """feat -> engine -> greedy / prefix beam, against the oracle searches run on the same logits; the best beam
    hypothesis of a peaked posterior is the greedy one."""
    from m3asr.config import EncoderConfig
    from m3asr.weights import make_weights
    from m3asr.engine import Engine
    from m3asr.decode import CtcDecoder
    cfg = EncoderConfig(num_blocks=1, embed_blocks=1)
    eng = Engine.from_state_dict(cfg, make_weights(cfg, seed=3))
    dec = CtcDecoder(eng, blank_idx=0)
    g = torch.Generator().manual_seed(12)
    feat = torch.rand(3, 150, cfg.input_dim, generator=g)
    fl = torch.tensor([150, 64, 97], dtype=torch.int32)
    res = dec.forward(feat, fl)
    logits, out_lens = res["out_nosm"].cpu(), res["out_lens"].cpu()
    assert out_lens.tolist() == [36, 15, 23]
    assert dec.ctc_greedy_search(feat, fl) == ref.ctc_greedy_search(logits.numpy(), out_lens.numpy(), 0)
    hyps, scores = dec.ctc_prefix_beam_search(feat[:1], fl[:1], beam_size=6)
    _same_hyps(hyps, ref.ctc_prefix_beam_search(scores[0].cpu().numpy(), 6, 0))
    with pytest.raises(NotImplementedError):
        dec.ctc_greedy_search(feat, fl, decoding_chunk_size=16)
    # sharpen the scores: beam search's best path collapses to the greedy path
    sharp = (scores * 50).contiguous()
    best = dec.prefix_beam_from_logits(sharp, 4)[0][0]
    assert list(best) == dec.greedy_from_logits(sharp, res["out_lens"][:1])[0]


# ------------------------------------------------------------------------------------------------ streaming operators (gpu)
@pytest.mark.gpu
@pytest.mark.parametrize("B,cd,idim", [(2, 8, 24), (3, 24, 8), (1, 16, 16), (4, 0, 5), (2, 3584, 512), (64, 7 * 512, 4 * 512)])
def test_cat_split_cache(B, cd, idim):
    from m3asr import ops
    g = torch.Generator().manual_seed(cd + idim)
    cache, inp = torch.randn(B, cd, generator=g), torch.randn(B, idim, generator=g)
    out, new_cache = ops.cat_split_cache(cache.cuda(), inp.cuda())
    wout, wcache = ref.cat_split_cache(cache.numpy(), inp.numpy())
    assert np.array_equal(out.cpu().numpy(), wout) and np.array_equal(new_cache.cpu().numpy(), wcache)   # a copy: bit-exact
    ci, ii = torch.randint(-9, 9, (B, cd), generator=g, dtype=torch.int32), torch.randint(-9, 9, (B, idim), generator=g, dtype=torch.int32)
    out, new_cache = ops.cat_split_cache(ci.cuda(), ii.cuda())
    assert torch.equal(out.cpu(), torch.cat([ci, ii], 1)) and torch.equal(new_cache.cpu(), torch.cat([ci, ii], 1)[:, idim:])


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,Tq,ld,cache_len", [(2, 4, 16, 24, 8), (3, 8, 16, 80, 64), (1, 8, 4, 300, 296), (2, 2, 5, 31, 0)])
def test_att_stream_softmax(B, H, Tq, ld, cache_len):
    from m3asr import ops
    g = torch.Generator().manual_seed(ld)
    x = torch.randn(B, H, Tq, ld, generator=g) * 4
    chunk = ld - cache_len
    dfn = torch.randint(1, ld + 8, (B,), generator=g, dtype=torch.int32)     # frames decoded so far (may exceed ld)
    mask = torch.randint(1, chunk + 1, (B,), generator=g, dtype=torch.int32)  # valid frames of this chunk
    dfn[0], mask[0] = ld + 3, chunk                                           # everything valid
    scale = 0.125
    out = ops.att_stream_softmax(x.cuda(), dfn.cuda(), mask.cuda(), cache_len, scale).cpu().numpy()
    want = ref.att_stream_softmax(x.reshape(B, H * Tq, ld).numpy(), dfn.numpy(), mask.numpy(), cache_len, scale).reshape(out.shape)
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out.sum(-1), 1.0, rtol=1e-5)
    assert (out[want == 0] == 0).all()
    # fully valid rows are an ordinary softmax of scale * x
    np.testing.assert_allclose(out[0], torch.softmax(x[0] * scale, -1).numpy(), rtol=1e-5, atol=1e-7)
    # a row with no valid key gives zeros, not NaN
    dfn0 = torch.zeros(B, dtype=torch.int32)
    assert (ops.att_stream_softmax(x.cuda(), dfn0.cuda(), mask.cuda(), cache_len, scale) == 0).all()


@pytest.mark.gpu
def test_rel_positional_encoding_with_offset():
    from m3asr import ops
    from m3asr._lib import M3Error
    from oracle.encoder_ref import positional_table
    B, T, D = 3, 16, 512
    pe = positional_table(200, D)
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(1))
    scale = float(D) ** 0.5
    y, pos = ops.rel_positional_encoding(x.cuda(), pe.cuda(), scale)
    wy, wpos = ref.rel_positional_encoding(x.numpy(), pe.numpy(), scale)
    assert np.array_equal(y.cpu().numpy(), wy) and np.array_equal(pos.cpu().numpy(), wpos)
    fn = torch.tensor([48, 48, 48], dtype=torch.int32)
    y, pos, fn2 = ops.rel_positional_encoding(x.cuda(), pe.cuda(), scale, frame_num=fn.cuda(), max_offset=48)
    wy, wpos, wfn = ref.rel_positional_encoding(x.numpy(), pe.numpy(), scale, fn.numpy())
    assert np.array_equal(y.cpu().numpy(), wy) and np.array_equal(pos.cpu().numpy(), wpos)
    assert fn2.cpu().tolist() == wfn.tolist() == [64, 64, 64]
    with pytest.raises(M3Error):
        ops.rel_positional_encoding(x.cuda(), pe.cuda(), scale, frame_num=fn.cuda(), max_offset=190)   # runs off the table
