"""GPU: the drop-in operator surface.  The model description emits the encoder op by op through
network_helper.add* and the *PluginDynamic operators (reference layouts, generic m3_plugin_* C ABI); the result must
equal the golden logits and the fused native engine.  Also the builder.py -> plan -> infer.py round trip."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import trt_helper
from trt_helper import trt
from m3asr.weights import make_weights
from model.conformer_fmoe_localComm_catEmbed_domain_acc_hier import Net


def _reference_conf(cfg):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_synthetic_checkpoint import reference_yaml
    return reference_yaml(cfg)["model_conf"]["encoder_conf"]


@pytest.mark.parametrize("name", ["tiny", "mid", "causal"])
def test_op_by_op_emission_matches_golden(golden, name):
    cfg, z = golden(name)
    net = Net(cfg.input_dim, cfg.output_dim, **_reference_conf(cfg))
    assert net.cfg == cfg
    net.load_state_dict(make_weights(cfg, seed=int(z["weight_seed"])))
    nh = trt_helper.NetworkHelper(config=trt_helper.HelperConfig())
    nh.bind_input("feat", torch.from_numpy(z["feat"]))
    nh.bind_input("feat_len", torch.from_numpy(z["feat_len"]).view(1, -1))
    feat = nh.addInput("feat", trt.float32, (-1, -1, cfg.input_dim))
    feat_len = nh.addInput("feat_len", trt.int32, (1, -1))
    out = net(nh, feat, feat_len)
    nh.markOutput(out)
    got, want = out.cpu(), torch.from_numpy(z["logits"])
    valid = torch.arange(got.shape[1]).view(1, -1) < torch.from_numpy(z["out_len"]).view(-1, 1)
    err = (got - want).abs()[valid]
    assert bool((err <= 2e-4 + 1e-3 * want.abs()[valid]).all()), float(err.max())


def test_plugin_errors_follow_the_reference():
    nh = trt_helper.NetworkHelper(config=trt_helper.HelperConfig())
    assert nh.plugin_registry.get_plugin_creator("NoSuchPluginDynamic", "1", "") is None
    creator = nh.plugin_registry.get_plugin_creator("FMoEExpertPluginDynamic", "1", "")
    bad = trt.PluginFieldCollection([trt.PluginField("data_type", np.array([0], np.int32), trt.PluginFieldType.INT32)])
    assert creator.create_plugin("plugin", bad) is None            # missing num_expert/idim/hidden_units -> null plugin
    with pytest.raises(RuntimeError):
        nh.addGelu(torch.zeros(1, 1, 1, device="cuda"))             # not on the hot path: "not support", as in the reference
    x = torch.zeros(2, 3, 8, device="cuda")
    with pytest.raises(RuntimeError):
        nh.addScale(x.view(6, 8), 2.0)                              # rank < 3 (tensor_network_helper.py:445-447)


def test_builder_and_infer_cli_round_trip(tmp_path):
    d = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "3m-asr-inference_amd"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_checkpoint.py"), "--out-dir", d,
                           "--tiny", "--seed", "3"], env=env)
    plan = os.path.join(d, "enc.plan")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "builder.py"), "-c", os.path.join(d, "config.yaml"),
                                   "-m", os.path.join(d, "model.pt"), "-o", plan, "--opt-shape", "2x64"], env=env, text=True)
    assert "fused engine vs op-by-op emission" in out and os.path.exists(plan)
    feat = np.random.default_rng(0).random((1, 206, 40), dtype=np.float32)
    np.save(os.path.join(d, "feat.npy"), feat)
    # expected output from the CPU oracle (the checker), written as the reference's -o compare file
    from m3asr.config import EncoderConfig
    from oracle.encoder_ref import encoder_forward
    cfg = EncoderConfig.tiny()
    want = encoder_forward(make_weights(cfg, seed=3), cfg, torch.from_numpy(feat), torch.tensor([206])).numpy()
    np.save(os.path.join(d, "want.npy"), want)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "infer.py"), "-p", plan, "-i", os.path.join(d, "feat.npy"),
                                   "-o", os.path.join(d, "want.npy")], env=env, text=True)
    assert "time=" in out and "torch.allclose result:True" in out


def test_builder_fp16_flag_writes_a_bf16_plan(tmp_path):
    """--fp16 (reference builder.py:160, never finished there) = the 16-bit weight mode: bf16 GEMM weights in the plan,
    about half the bytes, and infer.py runs it."""
    d = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "3m-asr-inference_amd"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_checkpoint.py"), "--out-dir", d,
                           "--tiny", "--seed", "3"], env=env)
    p32, p16 = os.path.join(d, "enc32.plan"), os.path.join(d, "enc16.plan")
    common = [sys.executable, os.path.join(ROOT, "builder.py"), "-c", os.path.join(d, "config.yaml"), "-m",
              os.path.join(d, "model.pt"), "--opt-shape", "2x64"]
    subprocess.check_output(common + ["-o", p32], env=env, text=True)
    out = subprocess.check_output(common + ["-o", p16, "--fp16"], env=env, text=True)
    assert "fused engine vs op-by-op emission" in out
    from m3asr.plan import load_plan, is_gemm_weight
    cfg16, packed16, _ = load_plan(p16)
    assert cfg16.weight_dtype == "bf16"
    assert all((v.dtype == torch.bfloat16) == is_gemm_weight(k) for k, v in packed16.items())
    pe_bytes = packed16["pe"].numel() * 4
    assert (os.path.getsize(p16) - pe_bytes) < 0.75 * (os.path.getsize(p32) - pe_bytes)
    feat = np.random.default_rng(0).random((1, 206, 40), dtype=np.float32)
    np.save(os.path.join(d, "feat.npy"), feat)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "infer.py"), "-p", p16, "-i", os.path.join(d, "feat.npy")],
                                  env=env, text=True)
    assert "time=" in out and "outputs.shape:(1, 50, 16)" in out


def test_builder_fp8_flag_writes_an_fp8_plan(tmp_path):
    """--fp8: e4m3 expert weights + per-row scales, bf16 dense weights (needs dims that are multiples of 64)."""
    d = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "3m-asr-inference_amd"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_checkpoint.py"), "--out-dir", d,
                           "--layers", "1", "--seed", "3"], env=env)
    plan = os.path.join(d, "enc8.plan")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "builder.py"), "-c", os.path.join(d, "config.yaml"),
                                   "-m", os.path.join(d, "model.pt"), "-o", plan, "--opt-shape", "2x64", "--fp8"],
                                  env=env, text=True)
    assert "fused engine vs op-by-op emission" in out
    from m3asr.plan import load_plan
    cfg8, packed8, _ = load_plan(plan)
    assert cfg8.weight_dtype == "fp8"
    assert packed8["blocks.0.feed_forward.experts.w_1.weight"].dtype == torch.float8_e4m3fn
    assert packed8["blocks.0.feed_forward.experts.w_1.scale"].dtype == torch.float32


def test_builder_int8_flag_calibrates_and_writes_an_fp8_arithmetic_plan(tmp_path):
    """--int8 (the reference's 8-bit slot: builder.py:39-49 `assert 0`, AsrCalibrator over lists of .npy batches,
    builder_helper.py:109-123 "use_int8 is true, but calibrator is None!"): fp8 expert weights + fp8 arithmetic, with one
    activation scale per MoE layer calibrated from the listed batches, cached like TensorRT's calibration cache."""
    d = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "3m-asr-inference_amd"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_checkpoint.py"), "--out-dir", d,
                           "--layers", "1", "--seed", "3"], env=env)
    rng = np.random.default_rng(1)
    feats, lens = [], []
    for i in range(3):
        np.save(os.path.join(d, "f%d.npy" % i), rng.random((2, 120 + 20 * i, 40), dtype=np.float32))
        np.save(os.path.join(d, "l%d.npy" % i), np.array([120 + 20 * i, 77], dtype=np.int32))
        feats.append("f%d.npy" % i)
        lens.append("l%d.npy" % i)
    open(os.path.join(d, "np_feat.list"), "w").write("\n".join(feats) + "\n")
    open(os.path.join(d, "np_feat_len.list"), "w").write("\n".join(lens) + "\n")
    plan, cache = os.path.join(d, "enc_i8.plan"), os.path.join(d, "conformer.int8.cache")
    cmd = [sys.executable, os.path.join(ROOT, "builder.py"), "-c", os.path.join(d, "config.yaml"), "-m", os.path.join(d, "model.pt"),
           "-o", plan, "--opt-shape", "2x64", "--int8", "--calib-feat-list", os.path.join(d, "np_feat.list"),
           "--calib-feat-len-list", os.path.join(d, "np_feat_len.list"), "--calib-cache", cache]
    out = subprocess.check_output(cmd, env=env, text=True)
    assert "calibrated h_scale per MoE layer" in out and "fused engine vs op-by-op emission" in out
    import json
    from m3asr.plan import load_plan
    cfg8, packed8, _ = load_plan(plan)
    assert cfg8.weight_dtype == "fp8" and cfg8.fp8_activations
    hs = float(packed8["blocks.0.feed_forward.experts.h_scale"])
    assert 1e-4 < hs < 1.0 and abs(json.load(open(cache))["h_scale"][0] - hs) < 1e-7
    out = subprocess.check_output(cmd, env=env, text=True)                 # second build: scales come from the cache
    assert "read from the calibration cache" in out
    # without a calibrator the 8-bit slot refuses, with the reference's message
    cfgh = trt_helper.HelperConfig()
    cfgh.use_int8 = True
    with pytest.raises(RuntimeError, match="calibrator is None"):
        trt_helper.BuilderHelper(cfgh, None, None)


def test_streaming_plugins_behind_the_reference_surface():
    """CatSplitCachePluginDynamic / AttStreamSoftmaxPluginDynamic / RelPositionalEncoding(streaming = 1) through the registry,
    create_plugin, add_plugin_v2 (generic m3_plugin_* C ABI) and network_helper.addCatSplitCache
    (TRTAPI++/python/trt_helper/network_helper.py:80-105), against the restatements of the kernel text
    (oracle/ctc_decode.py; parity unpinned: the reference ships no fixture for them and its CUDA cannot be built here)."""
    from oracle import ctc_decode as ref
    nh = trt_helper.NetworkHelper(config=trt_helper.HelperConfig())
    g = torch.Generator().manual_seed(3)
    # --- addCatSplitCache on a (B, h, T, dk) key cache, along the time axis
    cache, x = torch.randn(2, 4, 6, 8, generator=g), torch.randn(2, 4, 3, 8, generator=g)
    out, new_cache = nh.addCatSplitCache(cache.cuda(), x.cuda(), 2)
    want = torch.cat([cache, x], 2)
    assert torch.equal(out.cpu(), want) and torch.equal(new_cache.cpu(), want[:, :, -6:])
    # input longer than the cache (the reference's other kernel pair), last axis by axis_dim = -1
    cache, x = torch.randn(3, 2, 4, generator=g), torch.randn(3, 2, 9, generator=g)
    out, new_cache = nh.addCatSplitCache(cache.cuda(), x.cuda(), -1)
    want = torch.cat([cache, x], 2)
    assert torch.equal(out.cpu(), want) and torch.equal(new_cache.cpu(), want[:, :, -4:])
    creator = nh.plugin_registry.get_plugin_creator("CatSplitCachePluginDynamic", "1", "")
    assert creator.create_plugin("p", trt.PluginFieldCollection([trt.PluginField("data_type", np.array([0], np.int32), trt.PluginFieldType.INT32)])) is None
    with pytest.raises(RuntimeError):
        nh.addCatSplitCache(torch.zeros(2, 3).cuda(), torch.zeros(2, 3).cuda(), 1)       # nbDims < 3 (cat_split_cache_plugin.cpp:95-98)
    # --- AttStreamSoftmax
    B, H, Tq, ld, cache_len = 2, 4, 5, 21, 16
    scores = torch.randn(B, H, Tq, ld, generator=g)
    dfn, mask = torch.tensor([21, 9], dtype=torch.int32), torch.tensor([5, 3], dtype=torch.int32)
    creator = nh.plugin_registry.get_plugin_creator("AttStreamSoftmaxPluginDynamic", "1", "")
    plugin = creator.create_plugin("AttStreamSoftmaxPluginDynamic", trt.PluginFieldCollection([
        trt.PluginField("data_type", np.array([0], np.int32), trt.PluginFieldType.INT32),
        trt.PluginField("scale", np.array([0.125], np.float32), trt.PluginFieldType.FLOAT32),
        trt.PluginField("cache_len", np.array([cache_len], np.int32), trt.PluginFieldType.INT32)]))
    got = nh.network.add_plugin_v2([scores.cuda(), dfn.cuda(), mask.cuda()], plugin).get_output(0).cpu().numpy()
    want = ref.att_stream_softmax(scores.reshape(B, H * Tq, ld).numpy(), dfn.numpy(), mask.numpy(), cache_len, 0.125).reshape(got.shape)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7)
    # --- RelPositionalEncoding, streaming = 1: third input = frame counter, pos_emb = pe[offset : offset + T]
    D, T = 16, 6
    pe = torch.randn(1, 64, D, generator=g)
    x = torch.randn(2, T, D, generator=g)
    fn = torch.tensor([10, 10], dtype=torch.int32)
    creator = nh.plugin_registry.get_plugin_creator("RelPositionalEncodingPluginDynamic", "1", "")
    plugin = creator.create_plugin("RelPositionalEncodingPluginDynamic", trt.PluginFieldCollection([
        trt.PluginField("data_type", np.array([0], np.int32), trt.PluginFieldType.INT32),
        trt.PluginField("scale", np.array([4.0], np.float32), trt.PluginFieldType.FLOAT32),
        trt.PluginField("max_len", np.array([64], np.int32), trt.PluginFieldType.INT32),
        trt.PluginField("dim", np.array([D], np.int32), trt.PluginFieldType.INT32),
        trt.PluginField("streaming", np.array([1], np.int32), trt.PluginFieldType.INT32)]))
    layer = nh.network.add_plugin_v2([x.cuda(), pe.cuda(), fn.cuda()], plugin)
    assert torch.equal(layer.get_output(0).cpu(), x * 4.0) and torch.equal(layer.get_output(1).cpu(), pe[:, 10:10 + T])
    with pytest.raises(RuntimeError):
        nh.network.add_plugin_v2([x.cuda(), pe.cuda()], plugin)                             # the frame counter is missing
