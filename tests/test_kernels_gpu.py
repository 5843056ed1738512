"""GPU parity tests: every HIP kernel, called through the C ABI (ctypes), against the CPU oracle.

Bar: bit-exact for integer / index / copy work; fp32 within the tolerance written at each check
(north_star: logits within 1e-3 relative of the PyTorch fp32 reference; per-op bounds are tighter).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from m3asr import ops, _lib
from oracle import encoder_ref as ref
from oracle.moe_index import moe_index_ref, local_scatter_ref, local_gather_ref


def dev(t):
    return t.cuda().contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, want, rtol, atol):
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    assert bool((err <= bound).all()), "max abs err %.3e (max |ref| %.3e), worst excess %.3e" % (
        float(err.max()), float(want.abs().max()), float((err - bound).max()))


# ------------------------------------------------------------------------------------------ index
IDX_CASES = [
    ("one", 1, 32), ("cfg2", 50, 32), ("cfg3", 1090, 32), ("cfg5", 4400, 64), ("chunk_edge", 1024, 32),
    ("chunk_edge+1", 1025, 32), ("wide", 3000, 256), ("E1", 77, 1), ("E4", 15, 4),
    # >= 4096 tokens: one work-group per 1024 tokens (moe_index_multi_kernel)
    ("multi_edge", 4096, 32), ("multi", 16384 + 37, 32), ("multi_e64", 7936, 64), ("multi_wide", 5000, 256), ("multi_E1", 4097, 1),
]


@pytest.mark.parametrize("name,S,E", IDX_CASES)
def test_moe_index_bit_exact(name, S, E):
    rng = np.random.default_rng(S * 131 + E)
    for kind in ("uniform", "all_one", "skewed", "with_dropped", "sorted_desc"):
        if kind == "uniform":
            g = rng.integers(0, E, S)
        elif kind == "all_one":
            g = np.full(S, E - 1)
        elif kind == "skewed":
            g = np.minimum(rng.geometric(0.3, S) - 1, E - 1)
        elif kind == "with_dropped":
            g = rng.integers(-1, E, S)
        else:
            g = np.sort(rng.integers(0, E, S))[::-1].copy()
        g = g.astype(np.int32)
        mapping, acc, pos = ops.moe_scatter_mapping(dev(torch.from_numpy(g)), E)
        m_ref, a_ref = moe_index_ref(g, E)
        assert np.array_equal(mapping.cpu().numpy(), m_ref), (name, kind)
        assert np.array_equal(acc.cpu().numpy(), a_ref), (name, kind)
        nv = int(a_ref[E])
        p = pos.cpu().numpy()[:nv]
        assert np.array_equal(m_ref[p], np.arange(nv)), (name, kind)       # pos is the inverse permutation


def test_moe_index_random_sweep():
    """40 random (S, E, routing law) draws across both forms of the kernel (one work-group below 4096 tokens, one per 1024
    tokens above) against the numpy oracle: mapping / acc_histogram / pos bit-exact, mapping a permutation of the kept rows."""
    rng = np.random.default_rng(2024)
    for _ in range(40):
        S = int(rng.choice([rng.integers(1, 4096), rng.integers(4096, 40000)]))
        E = int(rng.choice([1, 2, 4, 8, 32, 64, 100, 256]))
        law = rng.integers(0, 3)
        if law == 0:
            g = rng.integers(0, E, S)
        elif law == 1:
            g = np.minimum(rng.geometric(0.2, S) - 1, E - 1)
        else:
            g = rng.integers(-1, E, S)                      # dropped rows (padded frames)
        gate = torch.from_numpy(g.astype(np.int32)).cuda()
        mapping, acc, pos = ops.moe_scatter_mapping(gate, E)
        want_map, want_acc = moe_index_ref(g.astype(np.int32), E)
        assert np.array_equal(mapping.cpu().numpy(), want_map), (S, E, law)
        assert np.array_equal(acc.cpu().numpy(), want_acc), (S, E, law)
        kept = int(want_acc[-1])
        assert np.array_equal(want_map[pos.cpu().numpy()[:kept]], np.arange(kept)), (S, E, law)   # pos = mapping^-1
        assert np.array_equal(np.sort(mapping.cpu().numpy()[g >= 0]), np.arange(kept))


@pytest.mark.parametrize("S,D", [(50, 512), (1090, 512), (4400, 512), (333, 32), (7, 4)])
def test_local_scatter_gather_bit_exact(S, D):
    rng = np.random.default_rng(S + D)
    g = rng.integers(-1, 32, S).astype(np.int32)
    m_ref, a_ref = moe_index_ref(g, 32)
    x = rng.standard_normal((S, D)).astype(np.float32)
    mapping = dev(torch.from_numpy(m_ref))
    buf = ops.moe_local_scatter(dev(torch.from_numpy(x)), mapping, int(a_ref[32]))
    assert np.array_equal(buf.cpu().numpy(), local_scatter_ref(x, m_ref, int(a_ref[32])))
    back = ops.moe_local_gather(buf, mapping)
    assert np.array_equal(back.cpu().numpy(), local_gather_ref(buf.cpu().numpy(), m_ref))
    # round trip: valid rows come back bit-identical, dropped rows are zero
    v = g >= 0
    assert np.array_equal(back.cpu().numpy()[v], x[v]) and (back.cpu().numpy()[~v] == 0).all()


def test_scatter_gather_round_trip_full_size():
    """Size-independent property at a size the loop oracle would not finish: gather(scatter(x)) == x."""
    S, D, E = 65536, 512, 32
    g = torch.randint(0, E, (S,), dtype=torch.int32, generator=torch.Generator().manual_seed(1))
    x = dev(rnd(S, D, seed=2))
    mapping, acc, pos = ops.moe_scatter_mapping(dev(g), E)
    assert int(acc[E]) == S
    m = mapping.cpu().numpy()
    assert np.array_equal(np.sort(m), np.arange(S))                    # permutation
    assert np.all(np.diff(g.numpy()[pos.cpu().numpy()]) >= 0)           # rows grouped by expert, ascending
    back = ops.moe_local_gather(ops.moe_local_scatter(x, mapping, S), mapping)
    assert torch.equal(back, x)


# ------------------------------------------------------------------------------------------ gate
@pytest.mark.parametrize("B,T,E", [(1, 50, 32), (3, 17, 64), (2, 9, 4)])
def test_softmax_top1(B, T, E):
    logits = rnd(B, T, E, seed=B * T, scale=4.0)
    logits[0, 0, :] = 0.0                                  # full tie -> reference tree rule
    logits[0, 1, 1] = logits[0, 1, 2] = 9.0
    lens = torch.tensor([T] + [max(1, T - 3 * i) for i in range(1, B)], dtype=torch.int32)
    val, idx = ops.softmax_top1(dev(logits), dev(lens), T)
    v_ref, i_ref = ref.softmax_topk(logits, lens.long())
    assert torch.equal(idx.cpu().view(B, T, 1), i_ref)     # integer output: exact
    close(val.view(B, T, 1), v_ref, 1e-5, 1e-7)


# ------------------------------------------------------------------------------------------ gemm
@pytest.mark.parametrize("M,N,K", [(50, 1024, 512), (50, 512, 1024), (50, 1434, 512), (13, 32, 64),
                                   (100, 1536, 512), (450, 512, 4608), (9, 16, 32), (1090, 512, 512)])
def test_linear_plain(M, N, K):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear(dev(a), dev(w), dev(b))
    close(y, F.linear(a, w, b), 2e-5, 2e-5)


@pytest.mark.parametrize("M,N,K,act", [(50, 512, 9728, _lib.ACT_NONE), (450, 512, 4608, _lib.ACT_RELU),
                                         (7, 64, 4096, _lib.ACT_SILU), (130, 100, 8192, _lib.ACT_NONE)])
def test_linear_split_k(M, N, K, act):
    """Deep-K, few-tile problems through m3_linear_ws: split-K tiled kernel + fixed-order reduce."""
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    d = _lib.LinearDesc()
    d.a, d.lda, d.w, d.M, d.N, d.K, d.ldy = 1, K, 1, M, N, K, N        # sizes only (pointers are not read)
    assert _lib.load().m3_linear_workspace_size(d) > 0
    y = ops.linear(dev(a), dev(w), dev(b), act=act, alpha=0.5, split_k=True)
    want = F.linear(a.double(), w.double(), b.double())
    want = {_lib.ACT_NONE: want, _lib.ACT_RELU: F.relu(want), _lib.ACT_SILU: F.silu(want)}[act] * 0.5
    close(y, want.float(), 3e-5, 3e-5)
    y2 = ops.linear(dev(a), dev(w), dev(b), act=act, alpha=0.5, split_k=True)
    assert torch.equal(y, y2)                                           # deterministic (no atomics)


def test_linear_epilogues():
    M, N, K, T = 100, 1024, 512, 50
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res = rnd(M, N, seed=4)
    lens = torch.tensor([50, 31], dtype=torch.int32)
    pad = (torch.arange(T).view(1, T) >= lens.view(2, 1)).reshape(M, 1)
    want = F.linear(a, w, b)
    close(ops.linear(dev(a), dev(w), dev(b), act=_lib.ACT_SILU), want * torch.sigmoid(want), 2e-5, 2e-5)
    close(ops.linear(dev(a), dev(w), dev(b), act=_lib.ACT_RELU), F.relu(want), 2e-5, 2e-5)
    close(ops.linear(dev(a), dev(w), dev(b), act=_lib.ACT_GLU), F.glu(want, -1), 2e-5, 2e-5)
    close(ops.linear(dev(a), dev(w), dev(b), alpha=0.5, resid=dev(res)), res + 0.5 * want, 2e-5, 2e-5)
    # masked_fill(0) before (input rows) and after (output rows), residual added after the mask
    want_m = F.linear(a.masked_fill(pad, 0.0), w, b)
    close(ops.linear(dev(a), dev(w), dev(b), lens=dev(lens), rows_per_batch=T, mask_in=True), want_m, 2e-5, 2e-5)
    close(ops.linear(dev(a), dev(w), dev(b), lens=dev(lens), rows_per_batch=T, mask_out=True, resid=dev(res)),
          res + want.masked_fill(pad, 0.0), 2e-5, 2e-5)
    # in-place residual update (Y aliases resid), as the engine does
    x = dev(res.clone())
    ops.linear(dev(a), dev(w), dev(b), alpha=0.5, resid=x, out=x)
    close(x, res + 0.5 * want, 2e-5, 2e-5)


@pytest.mark.parametrize("eps", [1e-12, 1e-5])
def test_linear_layernorm_prologue(eps):
    M, N, K = 50, 1536, 512
    a = rnd(M, K, seed=1) * 3.0 + 1.5
    w, b = rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    g, be = rnd(K, seed=4) * 0.2 + 1.0, rnd(K, seed=5, scale=0.1)
    want = F.linear(F.layer_norm(a, (K,), g, be, eps), w, b)
    close(ops.linear(dev(a), dev(w), dev(b), ln=(dev(g), dev(be), eps)), want, 3e-5, 3e-5)
    close(ops.layer_norm(dev(a), dev(g), dev(be), eps), F.layer_norm(a, (K,), g, be, eps), 1e-5, 1e-5)


@pytest.mark.parametrize("mean,std", [(0.0, 1.0), (1.5, 3.0), (10.0, 1.0), (100.0, 1.0), (-40.0, 0.25)])
def test_linear_folded_layernorm(mean, std):
    """Output-side LayerNorm (engine path): Linear(LN(x)) with the affine folded into the weight at pack time.
    Rows with a large common offset (|mean| = 100..160 std) are the adversarial case for the one-pass fp32 statistics
    (var = E[x^2] - mean^2, y = rstd * (a.W' - mean * wsum)): both subtractions cancel, and the error grows linearly with
    |mean| / std -- measured 4e-5 of the output scale per unit of |mean| / std (4e-3 at 100), i.e. still inside the
    north_star's 1e-3 relative at an offset of 100 std.  A trained Conformer's residual stream is LayerNorm'd at every
    block output (|mean| / std of order 1); the bound below states the law instead of hiding it."""
    from m3asr.plan import fold_layernorm
    M, N, K, T = 100, 1024, 512, 50
    a = rnd(M, K, seed=1) * std + mean
    w, b = rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    g, be = rnd(K, seed=4) * 0.2 + 1.0, rnd(K, seed=5, scale=0.1)
    f = fold_layernorm(w, b, g, be)
    tol = max(3e-5, 6e-5 * abs(mean) / std)
    want = F.linear(F.layer_norm(a.double(), (K,), g.double(), be.double(), 1e-12), w.double(), b.double()).float()
    got = ops.linear(dev(a), dev(f["ln.weight"]), dev(f["ln.bias"]), ln_folded=(dev(f["ln.wsum"]), None, 1e-12))
    print("folded LN mean=%g std=%g: max err %.3e" % (mean, std, float((got.cpu() - want).abs().max())))
    close(got, want, tol, tol)
    # conv-module use: LayerNorm -> masked_fill(0) on padded frames -> pointwise conv -> GLU
    lens = torch.tensor([50, 31], dtype=torch.int32)
    pad = (torch.arange(T).view(1, T) >= lens.view(2, 1)).reshape(M, 1)
    want = F.glu(F.linear(F.layer_norm(a.double(), (K,), g.double(), be.double(), 1e-12).masked_fill(pad, 0.0), w.double(), b.double()), -1).float()
    got = ops.linear(dev(a), dev(f["ln.weight"]), dev(f["ln.bias"]), act=_lib.ACT_GLU, lens=dev(lens), rows_per_batch=T,
                     mask_in=True, ln_folded=(dev(f["ln.wsum"]), dev(f["ln.wbeta"]), 1e-12))
    close(got, want, tol, tol)


def test_linear_concat_router():
    M, De, D, E = 50, 512, 512, 32
    emb, x, wr = rnd(M, De, seed=1), rnd(M, D, seed=2), rnd(De + D, E, seed=3, scale=0.5)
    y = ops.linear(dev(emb), dev(wr.t().contiguous()), a2=dev(x))
    close(y, torch.cat([emb, x], -1) @ wr, 2e-5, 1e-4)


@pytest.mark.parametrize("S,E,De,D", [(50, 32, 512, 512), (1, 64, 512, 512), (17, 16, 64, 128), (1090, 32, 512, 512), (4480, 64, 512, 512),
                                      (333, 48, 256, 1024), (50, 8, 512, 512)])
def test_moe_router(S, E, De, D):
    """moe_router.hip: logits = cat([embed, LayerNorm(x)]) @ W_r^T (+ bias) and xn = LayerNorm(x), one work-group per 16 rows
    and all experts (positionwise_feed_forward.py:169-180,225); offsets make the LayerNorm matter."""
    emb, x = rnd(S, De, seed=1), rnd(S, D, seed=2) * 1.7 + 0.4
    wr, b = rnd(E, De + D, seed=3, scale=0.5), rnd(E, seed=4)
    ga, be = rnd(D, seed=5, scale=0.3) + 1.0, rnd(D, seed=6, scale=0.2)
    eps = 1e-12
    want_xn = F.layer_norm(x.double(), (D,), ga.double(), be.double(), eps)
    want = torch.cat([emb.double(), want_xn], -1) @ wr.double().t()
    y, xn = ops.moe_router(dev(emb), dev(x), dev(wr), (dev(ga), dev(be), eps))
    close(xn, want_xn, 2e-5, 2e-5)
    close(y, want, 2e-5, 2e-4)
    yb, _ = ops.moe_router(dev(emb), dev(x), dev(wr), (dev(ga), dev(be), eps), bias=dev(b), want_xn=False)
    close(yb, want + b.double(), 2e-5, 2e-4)
    assert torch.equal(ops.moe_router(dev(emb), dev(x), dev(wr), (dev(ga), dev(be), eps))[0], y)     # run-to-run: bit for bit


# ------------------------------------------------------------------------------------------ expert FFN
@pytest.mark.parametrize("S,E,D,Fh,mode", [(50, 32, 512, 1024, "uniform"), (50, 32, 512, 1024, "all_one"),
                                           (200, 32, 512, 1024, "uniform"), (1090, 32, 512, 1024, "with_dropped"),
                                           (23, 4, 32, 64, "with_dropped"), (600, 64, 512, 1024, "uniform"),
                                           # >= 1024 rows: two grouped LDS-tiled GEMMs (64- and 128-row tiles)
                                           (2048, 32, 512, 1024, "all_one"), (8192, 32, 512, 1024, "uniform"),
                                           (6500, 8, 512, 1024, "with_dropped")])
def test_fmoe_expert(S, E, D, Fh, mode):
    rng = np.random.default_rng(S + E)
    g = {"uniform": rng.integers(0, E, S), "all_one": np.full(S, 3), "with_dropped": rng.integers(-1, E, S)}[mode]
    g = torch.from_numpy(g.astype(np.int32))
    x = rnd(S, D, seed=1)
    w1, b1 = rnd(E, Fh, D, seed=2, scale=D ** -0.5), rnd(E, Fh, seed=3, scale=0.1)
    w2, b2 = rnd(E, D, Fh, seed=4, scale=Fh ** -0.5), rnd(E, D, seed=5, scale=0.1)
    y = ops.moe_expert_ffn(dev(x), dev(g), dev(w1), dev(b1), dev(w2), dev(b2))
    y_ref, m_ref, a_ref = ref.fmoe_expert(x.view(1, S, D), g.view(1, S, 1), w1, b1, w2, b2)
    close(y, y_ref.view(S, D), 3e-5, 3e-5)
    assert bool((y.cpu()[g < 0] == 0).all())
    # fused epilogue: resid + 0.5 * gate * y, then LayerNorm
    gate, res = torch.rand(S, generator=torch.Generator().manual_seed(6)), rnd(S, D, seed=7)
    ga, be = rnd(D, seed=8) * 0.2 + 1.0, rnd(D, seed=9, scale=0.1)
    y2 = ops.moe_expert_ffn(dev(x), dev(g), dev(w1), dev(b1), dev(w2), dev(b2), gate_value=dev(gate), resid=dev(res),
                            alpha=0.5, ln=(dev(ga), dev(be), 1e-12))
    want = F.layer_norm(res + 0.5 * (gate * (g >= 0)).view(S, 1) * y_ref.view(S, D), (D,), ga, be, 1e-12)
    close(y2, want, 5e-5, 5e-5)


def test_fmoe_expert_position_independence():
    """A token's FFN output does not depend on which other tokens are in the batch (what makes
    expert-parallel results bit-identical to single-GPU ones, SURVEY.md §8e 'wire order')."""
    S, E, D, Fh = 64, 32, 512, 1024
    x = rnd(S, D, seed=1)
    w1, b1 = rnd(E, Fh, D, seed=2, scale=D ** -0.5), rnd(E, Fh, seed=3, scale=0.1)
    w2, b2 = rnd(E, D, Fh, seed=4, scale=Fh ** -0.5), rnd(E, D, seed=5, scale=0.1)
    g = torch.randint(0, E, (S,), dtype=torch.int32, generator=torch.Generator().manual_seed(3))
    args = [dev(t) for t in (w1, b1, w2, b2)]
    y_all = ops.moe_expert_ffn(dev(x), dev(g), *args)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(4))
    y_perm = ops.moe_expert_ffn(dev(x[perm]), dev(g[perm]), *args)
    assert torch.equal(y_perm.cpu(), y_all.cpu()[perm])
    y_half = ops.moe_expert_ffn(dev(x[:20]), dev(g[:20]), *args)
    assert torch.equal(y_half.cpu(), y_all.cpu()[:20])


def test_fmoe_expert_position_independence_long_batch():
    """The same property for the two grouped LDS-tiled GEMMs (S >= 1024): a row's result does not depend on which tile of
    its expert it lands in, nor on the other rows of the batch."""
    S, E, D, Fh = 2048, 32, 512, 1024
    x = rnd(S, D, seed=1)
    w1, b1 = rnd(E, Fh, D, seed=2, scale=D ** -0.5), rnd(E, Fh, seed=3, scale=0.1)
    w2, b2 = rnd(E, D, Fh, seed=4, scale=Fh ** -0.5), rnd(E, D, seed=5, scale=0.1)
    g = torch.randint(0, E, (S,), dtype=torch.int32, generator=torch.Generator().manual_seed(3))
    args = [dev(t) for t in (w1, b1, w2, b2)]
    y_all = ops.moe_expert_ffn(dev(x), dev(g), *args)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(4))
    y_perm = ops.moe_expert_ffn(dev(x[perm]), dev(g[perm]), *args)
    assert torch.equal(y_perm.cpu(), y_all.cpu()[perm])
    y_part = ops.moe_expert_ffn(dev(x[:1500]), dev(g[:1500]), *args)       # fewer rows per expert, other tile boundaries
    assert torch.equal(y_part.cpu(), y_all.cpu()[:1500])


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,T,H,dk,lens", [(1, 50, 8, 64, [50]), (2, 50, 8, 64, [50, 36]), (2, 37, 4, 128, [37, 5]),
                                           (3, 9, 2, 16, [9, 6, 1]), (1, 124, 8, 64, [124]),
                                           # >= 128 workgroups of 64 queries: one 16-query tile per wave, no merge
                                           (16, 124, 8, 64, [124, 12, 99, 124, 77, 64, 65, 1, 124, 33, 120, 124, 50, 63, 17, 101]),
                                           (12, 70, 4, 128, [70, 66, 3, 64, 65, 70, 1, 20, 70, 70, 48, 49])])
def test_relpos_attention(B, T, H, dk, lens):
    D = H * dk
    qkv, p = rnd(B * T, 3 * D, seed=1), rnd(T, D, seed=2)
    u, v = rnd(H, dk, seed=3, scale=0.3), rnd(H, dk, seed=4, scale=0.3)
    L = torch.tensor(lens, dtype=torch.int32)
    out = ops.relpos_attention(dev(qkv), dev(p), dev(u), dev(v), dev(L), B, T, H, dk)
    q, k, vv = [t.view(B, T, H, dk) for t in qkv.view(B, T, 3 * D).split(D, -1)]
    pp = p.view(1, T, H, dk)
    ac = torch.matmul((q + u).transpose(1, 2), k.permute(0, 2, 3, 1))
    bd = torch.matmul((q + v).transpose(1, 2), pp.permute(0, 2, 3, 1))
    pad = torch.arange(T).view(1, 1, 1, T) >= L.view(B, 1, 1, 1)
    att = torch.softmax(((ac + bd) / math.sqrt(dk)).masked_fill(pad, -float("inf")), -1).masked_fill(pad, 0.0)
    want = torch.matmul(att, vv.transpose(1, 2)).transpose(1, 2).reshape(B * T, D)
    close(out, want, 3e-5, 3e-5)


@pytest.mark.parametrize("B,T,H,dk,lens", [(1, 50, 8, 64, [50]), (2, 36, 8, 64, [36, 20]), (1, 124, 8, 64, [124]), (1, 128, 8, 64, [128]),
                                           (16, 124, 8, 64, [124, 12, 99, 124, 77, 64, 65, 1, 124, 33, 120, 124, 50, 63, 17, 101]),
                                           (12, 70, 4, 128, [70, 66, 3, 64, 65, 70, 1, 20, 70, 70, 48, 49]), (3, 17, 4, 128, [17, 16, 15])])
def test_relpos_attention_bf16(B, T, H, dk, lens):
    """The attention core on bf16 rows (16-bit modes of long batches): one work-group per (utterance, head), transposed
    scores on the bf16 MFMA, fp32 softmax.  Reference: fp64 evaluation on the bf16 inputs (q + u, q + v rounded to bf16 as
    the kernel does; the probabilities' rounding to bf16 before P.V is the kernel's own error: 2^-9 relative)."""
    D = H * dk
    qkv, p = rnd(B * T, 3 * D, seed=1).to(torch.bfloat16), rnd(T, D, seed=2)
    u, v = rnd(H, dk, seed=3, scale=0.3), rnd(H, dk, seed=4, scale=0.3)
    L = torch.tensor(lens, dtype=torch.int32)
    out = ops.relpos_attention_bf16(dev(qkv), dev(p), dev(u), dev(v), dev(L), B, T, H, dk)
    assert out.dtype == torch.bfloat16
    q, k, vv = [t.float().view(B, T, H, dk) for t in qkv.view(B, T, 3 * D).split(D, -1)]
    r16 = lambda t: t.to(torch.bfloat16).double()
    pp = r16(p).view(1, T, H, dk)
    ac = torch.matmul(r16(q + u).transpose(1, 2), k.double().permute(0, 2, 3, 1))
    bd = torch.matmul(r16(q + v).transpose(1, 2), pp.permute(0, 2, 3, 1))
    pad = torch.arange(T).view(1, 1, 1, T) >= L.view(B, 1, 1, 1)
    att = torch.softmax(((ac + bd) / math.sqrt(dk)).masked_fill(pad, -float("inf")), -1).masked_fill(pad, 0.0)
    want = torch.matmul(att, vv.double().transpose(1, 2)).transpose(1, 2).reshape(B, T, D)
    got = out.float().view(B, T, D)
    valid = torch.arange(T).view(1, -1) < L.view(-1, 1)
    close(got[valid], want[valid], 1.2e-2, 1.2e-2)          # bf16 probabilities and bf16 output: 2^-8 relative each
    assert bool(torch.isfinite(out.float()).all())


# ------------------------------------------------------------------------------------------ conv pieces
@pytest.mark.parametrize("B,T,D,K", [(1, 50, 512, 15), (2, 36, 512, 15), (2, 9, 32, 15), (1, 5, 64, 7),
                                     (16, 124, 512, 15), (7, 99, 512, 7), (130, 5, 64, 15)])   # >= 512 rows: 8 frames per workgroup
def test_dwconv_ln_silu(B, T, D, K):
    z, w, b = rnd(B, T, D, seed=1), rnd(D, 1, K, seed=2, scale=0.3), rnd(D, seed=3, scale=0.1)
    g, be = rnd(D, seed=4) * 0.2 + 1.0, rnd(D, seed=5, scale=0.1)
    out = ops.dwconv_ln_silu(dev(z.view(B * T, D)), dev(w.squeeze(1).t().contiguous()), dev(b), dev(g), dev(be), 1e-5, B, T)
    y = F.conv1d(z.transpose(1, 2), w, b, padding=(K - 1) // 2, groups=D).transpose(1, 2)
    y = F.layer_norm(y, (D,), g, be, 1e-5)
    close(out, (y * torch.sigmoid(y)).reshape(B * T, D), 2e-5, 2e-5)


@pytest.mark.parametrize("B,T,idim,C", [(1, 206, 40, 512), (2, 40, 40, 32), (2, 61, 40, 64)])
def test_subsampling_convs(B, T, idim, C):
    feat = torch.rand(B, T, idim, generator=torch.Generator().manual_seed(1))
    w0, b0 = rnd(C, 1, 3, 3, seed=2, scale=1 / 3), rnd(C, seed=3, scale=0.1)
    w2, b2 = rnd(C, C, 3, 3, seed=4, scale=(9 * C) ** -0.5), rnd(C, seed=5, scale=0.1)
    c1 = ops.subsample_conv1(dev(feat), dev(w0.reshape(C, 9).t().contiguous()), dev(b0))
    y1 = F.relu(F.conv2d(feat.unsqueeze(1), w0, b0, stride=2))
    close(c1, y1.permute(0, 2, 3, 1), 1e-5, 1e-5)
    c2 = ops.subsample_conv2(c1, dev(w2.permute(0, 2, 3, 1).contiguous()), dev(b2))
    y2 = F.relu(F.conv2d(y1, w2, b2, stride=2))
    close(c2, y2.permute(0, 2, 3, 1), 3e-5, 3e-5)


# ------------------------------------------------------------------------------------------ small plugins
def test_small_plugins():
    B, H, T = 2, 4, 19
    lens = torch.tensor([19, 11], dtype=torch.int32)
    s = rnd(B, H, T, T, seed=1, scale=3.0)
    pad = torch.arange(T).view(1, 1, 1, T) >= lens.view(B, 1, 1, 1)
    want = torch.softmax((s * 0.25).masked_fill(pad, -float("inf")), -1).masked_fill(pad, 0.0)
    close(ops.att_masked_softmax(dev(s), dev(lens), 0.25), want, 1e-5, 1e-6)
    x = rnd(B, 24, T, seed=2)
    assert torch.equal(ops.masked_fill(dev(x), dev(lens), 0.0).cpu(),
                       x.masked_fill(torch.arange(T).view(1, 1, T) >= lens.view(B, 1, 1), 0.0))
    x4 = rnd(B, 24, 1, T, seed=3)
    close(ops.glu(dev(x4), 1), F.glu(x4, 1), 1e-6, 1e-6)
    close(ops.glu(dev(x), -1 if T % 2 == 0 else 1), F.glu(x, 1), 1e-6, 1e-6)
    l0 = torch.tensor([206, 150, 7, 500], dtype=torch.int32)
    assert ops.mask_conv2d_sample(dev(l0), 2, 2).cpu().tolist() == [102, 74, 3, 249]
    assert torch.equal(ops.scale(dev(x), 2.0).cpu(), x * 2.0)
    a, b = rnd(2, 5, 8, seed=4), rnd(2, 5, 1, seed=5)
    assert torch.equal(ops.binary(dev(a), dev(b), _lib.OP_PROD).cpu(), a * b)
    assert torch.equal(ops.binary(dev(a), dev(a), _lib.OP_SUM).cpu(), a + a)
    y = rnd(2, 3, 4, 5, seed=6)
    assert torch.equal(ops.permute_copy(dev(y), (0, 2, 3, 1)).cpu(), y.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.concat_last(dev(a), dev(b)).cpu(), torch.cat([a, b], -1))
    close(ops.softmax_lastdim(dev(a)), torch.softmax(a, -1), 1e-6, 1e-7)
    m1, m2 = rnd(2, 4, 7, 16, seed=7), rnd(2, 4, 16, 9, seed=8)
    close(ops.batched_matmul(dev(m1), dev(m2)), m1 @ m2, 1e-5, 1e-5)
    m3 = rnd(1, 4, 9, 16, seed=9)
    close(ops.batched_matmul(dev(m1), dev(m3.expand(1, 4, 9, 16).contiguous()[0:1].repeat(2, 1, 1, 1)), transpose_b=True),
          m1 @ m3.transpose(-1, -2), 1e-5, 1e-5)
    xc, wc, bc = rnd(2, 8, 21, seed=10), rnd(8, 1, 15, seed=11), rnd(8, seed=12)
    close(ops.depthwise_conv1d(dev(xc), dev(wc), dev(bc), 7), F.conv1d(xc, wc, bc, padding=7, groups=8), 1e-5, 1e-5)


# ------------------------------------------------------------------------------------------ long batches (LDS-tiled fp32 GEMM)
@pytest.mark.parametrize("M,N,K", [(496, 1024, 512), (700, 512, 1024), (1984, 1536, 512), (4464, 512, 512), (400, 1434, 512),
                                   (6000, 1024, 512)])
def test_linear_tiled_plain(M, N, K):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear(dev(a), dev(w), dev(b))
    close(y, F.linear(a.double(), w.double(), b.double()).float(), 3e-5, 3e-5)


@pytest.mark.parametrize("B,T", [(8, 97), (40, 124)])       # 776 rows: 64-row tiles; 4960 rows: 128-row tiles
def test_linear_tiled_epilogues(B, T):
    M, N, K = B * T, 512, 512
    a, w, b = rnd(M, K, seed=1), rnd(2 * N, K, seed=2, scale=K ** -0.5), rnd(2 * N, seed=3)
    lens = torch.tensor(([T, 20, 1, T - 1, 50, T, 33, 64] * 5)[:B], dtype=torch.int32)
    pad = (torch.arange(T).view(1, -1) >= lens.view(-1, 1)).reshape(M, 1)
    res = rnd(M, N, seed=4)
    lin = F.linear(a.double(), w.double(), b.double())
    y = ops.linear(dev(a), dev(w[:N]), dev(b[:N]), act=_lib.ACT_SILU, alpha=0.5, resid=dev(res))
    close(y, (res.double() + 0.5 * F.silu(lin[:, :N])).float(), 3e-5, 3e-5)
    lin0 = F.linear(a.masked_fill(pad, 0.0).double(), w.double(), b.double())
    y = ops.linear(dev(a), dev(w), dev(b), act=_lib.ACT_GLU, lens=dev(lens), rows_per_batch=T, mask_in=True)
    close(y, (lin0[:, :N] * torch.sigmoid(lin0[:, N:])).float(), 3e-5, 3e-5)
    y = ops.linear(dev(a), dev(w[:N]), dev(b[:N]), act=_lib.ACT_RELU, lens=dev(lens), rows_per_batch=T, mask_out=True)
    close(y, F.relu(lin[:, :N]).masked_fill(pad, 0.0).float(), 3e-5, 3e-5)


@pytest.mark.parametrize("M,N,mean,std", [(600, 1024, 0.0, 1.0), (1111, 1024, 1.5, 3.0), (5000, 1536, 0.0, 1.0)])
def test_linear_tiled_folded_layernorm(M, N, mean, std):
    from m3asr.plan import fold_layernorm
    K, eps = 512, 1e-12
    a = rnd(M, K, seed=1) * std + mean
    w, b = rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ga, be = rnd(K, seed=4) * 0.2 + 1.0, rnd(K, seed=5, scale=0.1)
    f = fold_layernorm(w, b, ga, be)
    y = ops.linear(dev(a), dev(f["ln.weight"]), dev(f["ln.bias"]), ln_folded=(dev(f["ln.wsum"]), None, eps))
    want = F.linear(F.layer_norm(a.double(), (K,), ga.double(), be.double(), eps), w.double(), b.double())
    close(y, want.float(), 2e-4 * (1 + abs(mean)), 2e-4 * (1 + abs(mean)))
