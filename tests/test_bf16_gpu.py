"""bf16-weight mode (BASELINE.json configs[2] "18-layer 32-expert bf16").

The reference only wires the 16-bit flags (builder.py:160 --fp16, HelperConfig.plugin_data_type, FMoEExpert `data_type`)
and asserts when they are used (fmoe_expert_plugin.cpp:264-266), so there is no 16-bit reference output to match:
"low precision gets its own tolerance" (SURVEY.md §6).  Two levels:
  * kernels: the bf16 GEMM / expert FFN against an fp64 evaluation of EXACTLY the arithmetic they claim (operands
    rounded to bf16 with round-to-nearest-even, exact products, fp32-or-better accumulation) -- tight tolerance;
  * whole encoder: bf16 engine against the fp32 oracle (the reference's numerics), tolerance stated below.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from m3asr import ops, _lib
from m3asr.config import EncoderConfig
from m3asr.engine import Engine
from m3asr.plan import pack_weights, fold_layernorm, save_plan, load_plan, is_gemm_weight
from m3asr.weights import make_weights
from oracle.encoder_ref import encoder_forward, sub_len
from oracle import encoder_ref as ref


def dev(t):
    return t.cuda().contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def r16(t):
    """round-to-nearest-even to bf16, back in fp64 (what the kernel's v_cvt_pk_bf16_f32 / the plan's cast do)"""
    return t.float().to(torch.bfloat16).double()


def close(got, want, rtol, atol):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    assert bool((err <= bound).all()), "max abs err %.3e (max |ref| %.3e), worst excess %.3e" % (
        float(err.max()), float(want.abs().max()), float((err - bound).max()))


@pytest.mark.parametrize("M,N,K", [(50, 1024, 512), (50, 512, 1024), (50, 1434, 512), (13, 32, 64), (200, 1536, 512),
                                   (700, 512, 512), (1500, 1024, 512), (50, 512, 9728), (97, 64, 4608)])
def test_linear_bf16_plain(M, N, K):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear(dev(a), dev(w.to(torch.bfloat16)), dev(b))
    want = r16(a) @ r16(w).t() + b.double()
    close(y, want, 2e-5, 2e-5)


@pytest.mark.parametrize("B,T", [(2, 37), (8, 97)])      # 74 rows: K-split kernel; 776 rows: LDS-tiled kernel
def test_linear_bf16_epilogues(B, T):
    M, N, K = B * T, 512, 512
    a, w, b = rnd(M, K, seed=1), rnd(2 * N, K, seed=2, scale=K ** -0.5), rnd(2 * N, seed=3)
    w16 = dev(w.to(torch.bfloat16))
    lens = torch.tensor([T, 20, 1, T - 1, 50, T, 33, 64][:B], dtype=torch.int32)
    pad = (torch.arange(T).view(1, -1) >= lens.view(-1, 1)).reshape(M, 1)
    res = rnd(M, N, seed=4)
    lin = (r16(a) @ r16(w).t() + b.double())
    # SiLU, scale, residual
    y = ops.linear(dev(a), w16[:N].contiguous(), dev(b[:N]), act=_lib.ACT_SILU, alpha=0.5, resid=dev(res))
    close(y, res.double() + 0.5 * F.silu(lin[:, :N]), 3e-5, 3e-5)
    # GLU with the padded input rows zeroed, ReLU + output mask
    a0 = a.masked_fill(pad, 0.0)
    lin0 = r16(a0) @ r16(w).t() + b.double()
    y = ops.linear(dev(a), w16, dev(b), act=_lib.ACT_GLU, lens=dev(lens), rows_per_batch=T, mask_in=True)
    close(y, lin0[:, :N] * torch.sigmoid(lin0[:, N:]), 3e-5, 3e-5)
    y = ops.linear(dev(a), w16[:N].contiguous(), dev(b[:N]), act=_lib.ACT_RELU, lens=dev(lens), rows_per_batch=T,
                   mask_out=True)
    close(y, F.relu(lin[:, :N]).masked_fill(pad, 0.0), 3e-5, 3e-5)


@pytest.mark.parametrize("M,mean,std", [(50, 0.0, 1.0), (50, 1.5, 3.0), (600, 0.0, 1.0), (1111, 1.5, 3.0)])
def test_linear_bf16_folded_layernorm(M, mean, std):
    """Output-side LayerNorm with bf16 weights: statistics from the fp32 rows, wsum from the ROUNDED folded weight."""
    N, K, eps = 1024, 512, 1e-12
    a = rnd(M, K, seed=1) * std + mean
    w, b = rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ga, be = rnd(K, seed=4) * 0.2 + 1.0, rnd(K, seed=5, scale=0.1)
    f = fold_layernorm(w, b, ga, be)
    w16 = f["ln.weight"].to(torch.bfloat16)
    wsum = w16.double().sum(1).float()
    y = ops.linear(dev(a), dev(w16), dev(f["ln.bias"]), ln_folded=(dev(wsum), None, eps))
    ad = a.double()
    mu, var = ad.mean(1, keepdim=True), ad.var(1, unbiased=False, keepdim=True)
    want_exact_algebra = (r16(a) @ w16.double().t() - mu * wsum.double()) / torch.sqrt(var + eps) + f["ln.bias"].double()
    close(y, want_exact_algebra, 1e-4, 1e-4 * (1 + abs(mean)))
    # and it is a bf16-accurate LayerNorm + Linear: error vs the fp32 layer at the bf16 rounding level
    want = F.linear(F.layer_norm(a, (K,), ga, be, eps), w, b)
    err = (y.cpu() - want).abs().max() / want.abs().max()
    assert err < 2e-2 * (1 + abs(mean) / std), float(err)


@pytest.mark.parametrize("S,E,D,Fh,mode", [(50, 32, 512, 1024, "uniform"), (50, 32, 512, 1024, "all_one"),
                                           (200, 32, 512, 1024, "uniform"), (1090, 32, 512, 1024, "with_dropped"),
                                           (23, 4, 32, 64, "with_dropped"), (600, 64, 512, 1024, "uniform"),
                                           # >= 1024 rows: two grouped GEMMs on the LDS-tiled core (64- and 128-row tiles)
                                           (2048, 32, 512, 1024, "all_one"), (8192, 32, 512, 1024, "uniform"),
                                           (6500, 8, 512, 1024, "with_dropped"),
                                           # ragged row tiles (an expert with 1 row, empty experts, one expert with most
                                           # rows), 64 experts, F not a multiple of 128
                                           (4096, 32, 512, 1024, "skewed"), (16384, 32, 512, 1024, "uniform"),
                                           (33000, 32, 512, 1024, "with_dropped"), (9000, 64, 512, 1024, "uniform"),
                                           (5000, 16, 512, 1088, "skewed"),
                                           # >= 512 rows per expert: the 256 x 256 x 64 LDS-DMA tiles of expert_gemm_g256.hip
                                           # (16384 / 33000 / 6500 above as well): skew with empty and one-row experts, dropped rows
                                           (40000, 32, 512, 1024, "skewed"), (20000, 8, 512, 1024, "skewed"),
                                           (70000, 64, 512, 1024, "with_dropped")])
def test_fmoe_expert_bf16(S, E, D, Fh, mode):
    rng = np.random.default_rng(S + E)
    if mode == "skewed":
        pr = np.ones(E)
        pr[0], pr[2:6] = 0.6 * E, 0.0
        skew = rng.choice(E, size=S, p=pr / pr.sum())
        skew[skew == 1] = 7
        skew[S // 2] = 1                      # expert 1: exactly one row
    g = {"uniform": lambda: rng.integers(0, E, S), "all_one": lambda: np.full(S, 3),
         "with_dropped": lambda: rng.integers(-1, E, S), "skewed": lambda: skew}[mode]()
    g = torch.from_numpy(g.astype(np.int32))
    x = rnd(S, D, seed=1)
    w1, b1 = rnd(E, Fh, D, seed=2, scale=D ** -0.5), rnd(E, Fh, seed=3, scale=0.1)
    w2, b2 = rnd(E, D, Fh, seed=4, scale=Fh ** -0.5), rnd(E, D, seed=5, scale=0.1)
    w1h, w2h = w1.to(torch.bfloat16), w2.to(torch.bfloat16)
    y = ops.moe_expert_ffn(dev(x), dev(g), dev(w1h), dev(b1), dev(w2h), dev(b2))
    want = torch.zeros(S, D, dtype=torch.float64)
    for e in range(E):
        rows = (g == e).nonzero().flatten()
        if rows.numel():
            h = F.silu(r16(x[rows]) @ w1h[e].double().t() + b1[e].double())
            want[rows] = r16(h) @ w2h[e].double().t() + b2[e].double()
    # H is rounded to bf16 from an fp32 accumulation: an element that lands on a rounding boundary may round the other
    # way than in the fp64 evaluation (one bf16 ulp of one of Fh terms), hence not 1e-5 but 1e-3 of the output scale
    close(y, want, 1e-3, 1e-3 * float(want.abs().max()))
    assert bool((y.cpu()[g < 0] == 0).all())
    # within bf16 accuracy of the fp32 expert FFN
    y32, _, _ = ref.fmoe_expert(x.view(1, S, D), g.view(1, S, 1), w1, b1, w2, b2)
    assert float((y.cpu() - y32.view(S, D)).abs().max()) < 3e-2 * float(y32.abs().max())


def test_fmoe_expert_bf16_position_independence_long_batch():
    """Two grouped tiled GEMMs (>= 1024 rows): a row's result depends neither on where it sits in the batch nor on the
    other rows -- a permuted batch gives the permuted result bit for bit (what lets expert-parallel ranks reproduce the
    single-GPU engine), and re-running is bit-reproducible."""
    S, E, D, Fh = 6000, 32, 512, 1024
    x = rnd(S, D, seed=1)
    w1, b1 = rnd(E, Fh, D, seed=2, scale=D ** -0.5).to(torch.bfloat16), rnd(E, Fh, seed=3, scale=0.1)
    w2, b2 = rnd(E, D, Fh, seed=4, scale=Fh ** -0.5).to(torch.bfloat16), rnd(E, D, seed=5, scale=0.1)
    g = torch.randint(0, E, (S,), dtype=torch.int32, generator=torch.Generator().manual_seed(3))
    args = [dev(t) for t in (w1, b1, w2, b2)]
    y_all = ops.moe_expert_ffn(dev(x), dev(g), *args)
    assert torch.equal(ops.moe_expert_ffn(dev(x), dev(g), *args), y_all)                 # run-to-run
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(4))
    y_perm = ops.moe_expert_ffn(dev(x[perm]), dev(g[perm]), *args)
    assert torch.equal(y_perm.cpu(), y_all.cpu()[perm])


def test_fmoe_expert_bf16_position_independence():
    S, E, D, Fh = 64, 32, 512, 1024
    x = rnd(S, D, seed=1)
    w1, b1 = rnd(E, Fh, D, seed=2, scale=D ** -0.5).to(torch.bfloat16), rnd(E, Fh, seed=3, scale=0.1)
    w2, b2 = rnd(E, D, Fh, seed=4, scale=Fh ** -0.5).to(torch.bfloat16), rnd(E, D, seed=5, scale=0.1)
    g = torch.randint(0, E, (S,), dtype=torch.int32, generator=torch.Generator().manual_seed(3))
    args = [dev(t) for t in (w1, b1, w2, b2)]
    y_all = ops.moe_expert_ffn(dev(x), dev(g), *args)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(4))
    y_perm = ops.moe_expert_ffn(dev(x[perm]), dev(g[perm]), *args)
    assert torch.equal(y_perm.cpu(), y_all.cpu()[perm])


# ---------------------------------------------------------------------------------------------- whole encoder
# Tolerance of the bf16 engine against the fp32 oracle: every GEMM input carries 2^-9 relative rounding noise, which
# accumulates in the fp32 residual stream over the blocks.  Routing is discrete: a token whose two best router logits are
# closer than that noise may pick the other expert (synthetic random routers have many near-ties), and then its
# output differs by a whole expert FFN, not by rounding.  So the comparison is teacher-forced: the oracle runs with the
# ENGINE's expert choices (gate value = softmax probability of that expert) and the logits must agree within
# BF16_REL of the largest logit; separately the engine's free-running choices must agree with the fp32 oracle's
# own choices on >= 90 % of the tokens.
BF16_REL = 2e-2     # measured 0.5-1.0e-2 on these cases and 0.8e-2 at 18 layers (tests/test_full_size_gpu.py)


def _bf16_case(cfg32, seed, lengths):
    w = make_weights(cfg32, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    feat = torch.rand(len(lengths), max(lengths), cfg32.input_dim, generator=g)
    fl = torch.tensor(lengths, dtype=torch.int32)
    cfg16 = EncoderConfig(**{**cfg32.__dict__, "weight_dtype": "bf16"})
    eng = Engine.from_state_dict(cfg16, w)
    out = eng(feat.cuda(), fl.view(1, -1).cuda()).cpu()
    B, Tp = out.shape[0], out.shape[1]
    forced = {"blocks.%d.gate_idx" % i: eng.rows_padded("blocks.%d.gate_idx" % i, torch.int32, fill=-1).cpu().view(B, Tp, 1).clone()
              for i in range(cfg32.num_blocks)}
    free_taps = {}
    encoder_forward(w, cfg32, feat, fl, taps=free_taps)
    want = encoder_forward(w, cfg32, feat, fl, route_override=forced)
    return eng, out, want, forced, free_taps, sub_len(fl.long())


@pytest.mark.parametrize("name,cfg,lengths", [
    ("tiny", EncoderConfig.tiny(), [206, 57]),
    ("mid", EncoderConfig(num_blocks=3, embed_blocks=2), [206, 131, 333]),
    ("long_batch", EncoderConfig(num_blocks=2, embed_blocks=1), [400, 57, 206, 333, 120, 399, 250, 64]),   # S = 792: tiled GEMMs
    # S = 1584 rows: every GEMM on the tiled kernel -> bf16 activation operands (xb copy of the residual stream, bf16 h1 /
    # ctx / dw / c1 / c2), grouped tiled expert FFN
    ("bf16_activations", EncoderConfig(num_blocks=2, embed_blocks=2),
     [400, 57, 206, 333, 120, 399, 250, 64, 380, 390, 395, 222, 111, 345, 400, 301]),
])
def test_engine_bf16_vs_fp32_oracle(name, cfg, lengths):
    eng, out, want, forced, free_taps, out_len = _bf16_case(cfg, 11, lengths)
    assert eng.weights["blocks.0.feed_forward.experts.w_1.weight"].dtype == torch.bfloat16
    assert eng.weights["blocks.0.feed_forward.router_weights_t"].dtype == torch.float32
    valid = torch.arange(out.shape[1]).view(1, -1) < out_len.view(-1, 1)
    err = float((out - want).abs()[valid].max()) / float(want.abs()[valid].max())
    print("bf16 %s: max |err| / max |logit| = %.3e (teacher-forced routing)" % (name, err))
    assert err < BF16_REL, err
    agree, total = 0, 0
    for i in range(cfg.num_blocks):
        gi = forced["blocks.%d.gate_idx" % i].view(valid.shape)
        ref_gi = free_taps["blocks.%d.gate_idx" % i].view(valid.shape)
        agree += int((gi[valid] == ref_gi[valid]).sum())
        total += int(valid.sum())
        assert bool((gi[~valid] == -1).all())
    print("bf16 %s: routing agreement %d / %d" % (name, agree, total))
    assert agree >= 0.93 * total, (agree, total)


def test_bf16_plan_round_trip_and_sizes(tmp_path):
    cfg = EncoderConfig(num_blocks=1, embed_blocks=1, weight_dtype="bf16")
    packed = pack_weights(make_weights(cfg, seed=4), cfg)
    n16 = sum(v.numel() for k, v in packed.items() if v.dtype == torch.bfloat16)
    n32 = sum(v.numel() for k, v in packed.items() if v.dtype == torch.float32)
    assert all(is_gemm_weight(k) == (v.dtype == torch.bfloat16) for k, v in packed.items())
    assert n16 > 10 * (n32 - packed["pe"].numel())          # nearly all parameters are GEMM weights
    path = str(tmp_path / "m.plan")
    save_plan(path, cfg, packed)
    cfg2, packed2, _ = load_plan(path)
    assert cfg2.weight_dtype == "bf16"
    for k, v in packed.items():
        assert packed2[k].dtype == v.dtype and torch.equal(packed2[k].view(torch.int16) if v.dtype == torch.bfloat16
                                                           else packed2[k], v.view(torch.int16) if v.dtype == torch.bfloat16 else v)
    feat = torch.rand(1, 206, cfg.input_dim, generator=torch.Generator().manual_seed(1)).cuda()
    fl = torch.tensor([[206]], dtype=torch.int32).cuda()
    a = Engine(cfg, packed)(feat, fl).clone()
    b = Engine(cfg2, packed2)(feat, fl).clone()
    assert torch.equal(a, b)
    # graph replay is bit-identical to eager in bf16 mode too
    e = Engine(cfg, packed)
    eager = e(feat, fl).clone()
    for _ in range(2):
        e.forward(use_graph=True)
    e.stream.synchronize()
    assert torch.equal(e._bound[2], eager)


def test_engine_rejects_dtype_mismatch():
    cfg = EncoderConfig.tiny()
    packed = pack_weights(make_weights(cfg, seed=0), cfg)             # fp32 plan
    cfg16 = EncoderConfig.tiny(weight_dtype="bf16")
    with pytest.raises(_lib.M3Error):
        Engine(cfg16, packed)


# ---------------------------------------------------------------------------------------------- LDS-DMA fed GEMM
# gemm_bf16_dma.hip: bf16 A x bf16 W from 512 rows on (what the engine's block GEMMs run on once it keeps bf16 activation
# copies).  Checked against an fp64 evaluation of exactly its arithmetic: the bf16 operands as they are, exact products,
# fp32-or-better accumulation, fp32 epilogue.
def _tile_stats(yb, n_out):
    """what y_copy_stats must hold: per row and per 128-column tile (sum, sum of squares) of the bf16 values"""
    parts = (n_out + 127) // 128
    f = yb.float().cpu().double()
    out = torch.zeros(f.shape[0], parts, 2, dtype=torch.float64)
    for q in range(parts):
        blk = f[:, 128 * q: 128 * (q + 1)]
        out[:, q, 0], out[:, q, 1] = blk.sum(1), (blk * blk).sum(1)
    return out


@pytest.fixture(scope="module")
def dma_from_512_rows():
    """The LDS-DMA kernel is taken from M3_DMA_MIN_ROWS rows on (default 4096, read once when the library loads): these tests
    run in a child interpreter with the threshold at 512 so that ragged and small shapes reach it too."""
    import os
    import subprocess
    import sys
    if os.environ.get("M3_DMA_MIN_ROWS") == "512":
        return True
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "dma_"],
                       env=dict(os.environ, M3_DMA_MIN_ROWS="512"), cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    return False


@pytest.mark.parametrize("M,N,K", [(512, 512, 512), (1090, 512, 1024), (4480, 1024, 512), (777, 1434, 512), (2000, 1536, 512),
                                   (640, 512, 4608), (8192, 512, 512)])
def test_linear_bf16_dma_plain_and_residual(M, N, K, dma_from_512_rows):
    if not dma_from_512_rows and M < 4096:
        return                                          # covered by the child run (threshold 512)
    a = rnd(M, K, seed=1).to(torch.bfloat16)
    w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
    b, res = rnd(N, seed=3), rnd(M, N, seed=4)
    want = a.double() @ w.double().t() + b.double()
    y = ops.linear(dev(a), dev(w), dev(b))
    close(y, want, 2e-5, 2e-5)
    # residual epilogue with scale, SiLU, a bf16 copy of the result and its per-tile row statistics
    want2 = res.double() + 0.5 * F.silu(want)
    yb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    if N % 128 == 0:
        st = torch.full((M, N // 128, 2), -1.0, device="cuda")
        y2 = ops.linear(dev(a), dev(w), dev(b), act=_lib.ACT_SILU, alpha=0.5, resid=dev(res), copy_bf16=yb, copy_stats=st)
        close(y2, want2, 2e-5, 2e-5)
        assert torch.equal(yb, y2.to(torch.bfloat16))
        ts = _tile_stats(yb, N)
        close(st, ts, 1e-5, 1e-4)
    # bf16 output (row stores of 4 elements: N % 4 == 0)
    if N % 4 == 0:
        y3 = ops.linear(dev(a), dev(w), dev(b), out_dtype=torch.bfloat16)
        assert y3.dtype == torch.bfloat16 and torch.equal(y3, y.to(torch.bfloat16))


@pytest.mark.parametrize("B,T", [(16, 90), (5, 300), (40, 124)])
def test_linear_bf16_dma_epilogues(B, T, dma_from_512_rows):
    if not dma_from_512_rows and B * T < 4096:
        return                                          # covered by the child run (threshold 512)
    """GLU + folded LayerNorm from the producer's row statistics + input mask (the conv module's pointwise_conv1), and the
    output mask (pointwise_conv2) -- the same contracts as the register-staged kernel, on bf16 rows."""
    M, K, Nh = B * T, 512, 512
    lens = torch.tensor([T - (7 * i) % T for i in range(B)], dtype=torch.int32)
    x = (rnd(M, K, seed=1) * 1.5 + 0.3)
    xb = x.to(torch.bfloat16)
    ga, be = rnd(K, seed=5, scale=0.3) + 1.0, rnd(K, seed=6, scale=0.2)
    w, b = rnd(2 * Nh, K, seed=2, scale=K ** -0.5), rnd(2 * Nh, seed=3)
    fo = fold_layernorm(w, b, ga, be)
    wf, bf, wbeta = fo["ln.weight"], fo["ln.bias"], fo["ln.wbeta"]
    wf16 = wf.to(torch.bfloat16)
    wsum16 = wf16.float().sum(1)                       # recomputed from the rounded weights, as plan.cast_gemm_weights does
    stats = torch.zeros(M, 4, 2)
    xf = xb.double()
    for q in range(4):
        stats[:, q, 0], stats[:, q, 1] = xf[:, 128 * q:128 * (q + 1)].sum(1).float(), (xf[:, 128 * q:128 * (q + 1)] ** 2).sum(1).float()
    eps = 1e-12
    y = ops.linear(dev(xb), dev(wf16), dev(bf), act=_lib.ACT_GLU, ln_folded=(dev(wsum16), dev(wbeta), eps), lens=dev(lens),
                   rows_per_batch=T, mask_in=True, ln_stats=dev(stats))
    # reference: LayerNorm statistics of the bf16 rows, normalisation on the output side with the rounded weights
    mean, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
    z = ((xf @ wf16.double().t()) - mean * wsum16.double()) / torch.sqrt(var + eps) + bf.double()
    t = torch.arange(M) % T
    pad = t >= lens.repeat_interleave(T)
    z[pad] = (bf.double() - wbeta.double())           # masked_fill(0) after the LayerNorm: the plain bias
    want = z[:, :Nh] * torch.sigmoid(z[:, Nh:])
    close(y, want, 2e-3, 2e-3)
    # output mask + residual
    w2, b2, res = rnd(Nh, K, seed=7, scale=K ** -0.5).to(torch.bfloat16), rnd(Nh, seed=8), rnd(M, Nh, seed=9)
    y2 = ops.linear(dev(xb), dev(w2), dev(b2), lens=dev(lens), rows_per_batch=T, mask_out=True, resid=dev(res))
    want2 = xf @ w2.double().t() + b2.double()
    want2[pad] = 0.0
    close(y2, want2 + res.double(), 2e-5, 2e-5)
