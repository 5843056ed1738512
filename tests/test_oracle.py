"""CPU tests: the oracle restatement against the committed golden vectors.

The fixtures under tests/golden/ were produced by oracle/gen_golden.py from the reference's
own Net.forward executed in the build container (the reference ships no golden vectors of its
own).  Tolerance: the reference harness's own torch.allclose(rtol=1e-5, atol=1e-3)
(TRTAPI++/python/trt_helper/infer_helper.py:93) is far looser than what we hold here.
"""
import numpy as np
import pytest
import torch

from m3asr.weights import make_weights, count_params
from m3asr.config import EncoderConfig, subsampled_len
from oracle.encoder_ref import encoder_forward, softmax_top1_tree, sub_len
from oracle.moe_index import moe_index_ref, moe_index_loops, local_scatter_ref, local_gather_ref


def _valid_mask(out_len, t):
    return np.arange(t)[None, :] < out_len[:, None]


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_oracle_matches_reference_forward_small(golden, name):
    cfg, z = golden(name)
    w = make_weights(cfg, seed=int(z["weight_seed"]))
    taps = {}
    logits = encoder_forward(w, cfg, torch.from_numpy(z["feat"]), torch.from_numpy(z["feat_len"]), taps).numpy()
    valid = _valid_mask(z["out_len"], logits.shape[1])
    assert logits.shape == z["logits"].shape
    np.testing.assert_allclose(logits[valid], z["logits"][valid], rtol=1e-5, atol=2e-5)
    # routing taps: expert choice must be identical on valid frames, gate prob close
    for i in range(z["gate_idx"].shape[0]):
        gi = taps["blocks.%d.gate_idx" % i].numpy()
        assert np.array_equal(gi[valid], z["gate_idx"][i][valid])
        np.testing.assert_allclose(taps["blocks.%d.gate_value" % i].numpy()[valid], z["gate_value"][i][valid],
                                   rtol=1e-5, atol=1e-6)
        # padded frames are pinned to idx -1 / value 0 by the oracle
        assert (gi[~valid] == -1).all()
    for i in range(z["block_out"].shape[0]):
        np.testing.assert_allclose(taps["blocks.%d.out" % i].numpy()[valid], z["block_out"][i][valid],
                                   rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("name", ["cfg1"])
def test_oracle_matches_reference_forward_cfg1(golden, name):
    """BASELINE.json configs[0]: 12-layer 32-expert, single 206-frame utterance, CPU."""
    cfg, z = golden(name)
    w = make_weights(cfg, seed=int(z["weight_seed"]))
    logits = encoder_forward(w, cfg, torch.from_numpy(z["feat"]), torch.from_numpy(z["feat_len"])).numpy()
    np.testing.assert_allclose(logits, z["logits"], rtol=1e-5, atol=2e-5)


def test_param_count_matches_survey():
    # 711,264,052 parameters at h=8, V=1434, 18 layers
    assert count_params(EncoderConfig()) == 711264052


def test_subsampled_len():
    assert [subsampled_len(t) for t in (206, 50, 500, 7)] == [50, 11, 124, 1]
    assert sub_len(torch.tensor([206, 150])).tolist() == [50, 36]


def test_moe_index_contract_small_and_adversarial():
    rng = np.random.default_rng(0)
    cases = [np.array([], np.int32), np.zeros(50, np.int32), np.full(7, 31, np.int32),
             rng.integers(0, 32, 50), rng.integers(0, 64, 1090), rng.integers(-1, 4, 33),
             np.array([3, -1, 3, 0, -1, 0, 3]), np.arange(32)[::-1].copy()]
    for g in cases:
        E = 64 if (g.size and g.max() >= 32) else 32
        m1, a1 = moe_index_ref(g, E)
        m2, a2 = moe_index_loops(g, E)
        assert np.array_equal(m1, m2) and np.array_equal(a1, a2)
        v = g >= 0
        assert a1[E] == v.sum()
        assert sorted(m1[v].tolist()) == list(range(int(v.sum())))          # permutation of valid rows
        assert (m1[~v] == -1).all()
        # rows of one expert are contiguous, in token order (stable)
        for e in np.unique(g[v]):
            rows = m1[g == e]
            assert np.array_equal(rows, np.arange(a1[e], a1[e + 1]))
        # equals the inverse of a stable argsort (FastMoE pos, fmoe/functions.py:30)
        if v.all() and g.size:
            pos = np.argsort(g, kind="stable")
            inv = np.empty_like(pos)
            inv[pos] = np.arange(g.size)
            assert np.array_equal(inv.astype(np.int32), m1)
        x = rng.standard_normal((g.size, 8)).astype(np.float32)
        back = local_gather_ref(local_scatter_ref(x, m1, int(a1[E])), m1)
        assert np.array_equal(back[v], x[v]) and (back[~v] == 0).all()


def test_top1_tree_rule():
    # reference tie rule (softmax_topk_kernel.cu:55-64): not "first index"
    assert softmax_top1_tree([0.0, 5.0, 5.0, 1.0]) == 2
    assert softmax_top1_tree([5.0, 5.0, 1.0, 1.0]) == 0
    assert softmax_top1_tree([1.0, 2.0, 3.0, 4.0]) == 3


def test_c_oracle_agrees_with_numpy(tmp_path):
    import ctypes, subprocess, os
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "moe_index.c")
    so = str(tmp_path / "libm3oracle.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lm"])
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(5)
    g = rng.integers(-1, 32, 777).astype(np.int32)
    mapping = np.empty(777, np.int32)
    acc = np.empty(33, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.m3o_moe_index(p(g), 777, 32, p(mapping), p(acc)) == 0
    m_ref, a_ref = moe_index_ref(g, 32)
    assert np.array_equal(mapping, m_ref) and np.array_equal(acc, a_ref)
