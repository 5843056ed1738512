"""Activation calibration for the fp8-arithmetic expert FFN (EncoderConfig.fp8_activations).

The reference wires TensorRT's INT8 calibration and never finishes it (builder.py:39-49 `assert 0`; the calibrator hook is
builder_helper.py:109-123: `builder_config.int8_calibrator = calibrator`, an IInt8EntropyCalibrator2 fed from lists of
.npy feature files).  Here the same hook produces the ONE quantity the fp8 path needs from data: a static scale per MoE
layer for the hidden activations H = SiLU(x W1^T + b1) (rows are scaled dynamically per row inside the kernel, weights per
output row at pack time).  e4m3 is a floating-point format, so no histogram / entropy search is needed: the scale only has
to keep H inside the representable range, h_scale = 1.25 * max |H| / 448 over the calibration batches.

Everything runs on the GPU through this repo's own operators (staged engine + m3_linear); nothing here is on the inference
path.
"""
import json
import os

import numpy as np
import torch

from . import _lib, ops
from .engine import Engine

FP8_MAX = 448.0
MARGIN = 1.25


def collect_h_amax(cfg, state_dict, feat, feat_len, device="cuda:0", engine=None):
    """max |H| per MoE layer for one batch (feat (B,T,idim) f32, feat_len (B,) / (1,B) i32, any device).  Uses a staged
    fp32 engine for everything up to each layer's routing, then evaluates H = SiLU(xn W1[e]^T + b1[e]) for the rows of every
    expert with m3_linear.  Returns (list of floats, engine) -- pass the engine back in to reuse it for the next batch."""
    import dataclasses
    if engine is None:
        # (an expert-parallel config names E_loc experts per rank: calibration runs with all of them local)
        c32 = dataclasses.replace(cfg, weight_dtype="f32", fp8_activations=False, ep_world_size=1, ep_rank=0,
                                  num_experts=cfg.num_experts * max(1, cfg.ep_world_size))
        engine = Engine.from_state_dict(c32, state_dict, device=device, fuse_route=False, packed_rows=False)
    eng = engine
    B = feat.shape[0]
    f = feat.to(eng.device, torch.float32).contiguous()
    fl = feat_len.reshape(1, B).to(eng.device, torch.int32).contiguous()
    eng.bind(f, fl)
    names = eng.stage_names()
    D = eng.cfg.attention_dim
    amax, cur = [], 0
    with torch.cuda.stream(eng.stream):
        for li in range(eng.cfg.num_blocks):
            stop = names.index("blocks.%d.moe_local.expert" % li)
            eng.run_stages(cur, stop)
            cur = stop
            xn = eng.buffer("xn").view(-1, D)
            gate = eng.buffer("blocks.%d.gate_idx" % li, torch.int32)
            w1 = state_dict["blocks.%d.feed_forward.experts.w_1.weight" % li]
            b1 = state_dict["blocks.%d.feed_forward.experts.w_1.bias" % li]
            top = 0.0
            for e in range(w1.shape[0]):
                rows = (gate == e).nonzero().flatten()
                if rows.numel() == 0:
                    continue
                h = ops.linear(xn[rows].contiguous(), w1[e].to(eng.device).contiguous(), b1[e].to(eng.device).contiguous(),
                               act=_lib.ACT_SILU)
                top = max(top, float(h.abs().max()))
            amax.append(top)
        eng.run_stages(cur, len(names))
    eng.stream.synchronize()
    return amax, eng


def calibrate_h_scales(cfg, state_dict, batches, device="cuda:0"):
    """h_scale per MoE layer from an iterable of (feat, feat_len) batches; writes "blocks.N.feed_forward.experts.h_scale"
    into `state_dict` (what plan.pack_weights picks up) and returns the list."""
    top, eng = None, None
    for feat, feat_len in batches:
        a, eng = collect_h_amax(cfg, state_dict, feat, feat_len, device=device, engine=eng)
        top = a if top is None else [max(u, v) for u, v in zip(top, a)]
    if top is None:
        raise RuntimeError("calibrate_h_scales: the calibrator produced no batch")
    scales = [max(t, 1e-6) * MARGIN / FP8_MAX for t in top]
    for i, v in enumerate(scales):
        state_dict["blocks.%d.feed_forward.experts.h_scale" % i] = torch.tensor([v], dtype=torch.float32)
    return scales


class AsrCalibrator:
    """The reference's calibrator object, by constructor and method names (builder.py:47:
    ``AsrCalibrator("np_inputs/np_feat.list", "np_inputs/np_feat_len.list", "conformer.int8.cache", 10)``; methods are those
    of TensorRT's IInt8EntropyCalibrator2 that builder_helper.py:109-123 hands to the builder).  feat_list / feat_len_list:
    text files with one .npy path per line (feat (B,T,idim) f32, feat_len (B,) or (1,B) i32), batch_num batches are used;
    cache_file stores the resulting scales (JSON) and short-cuts later builds, like TensorRT's calibration cache."""

    def __init__(self, feat_list=None, feat_len_list=None, cache_file=None, batch_num=10, batches=None):
        self.cache_file, self.batch_num = cache_file, int(batch_num)
        self._batches = list(batches) if batches is not None else None      # in-memory batches (tests, synthetic data)
        self._feat_paths = self._read_list(feat_list) if feat_list else []
        self._len_paths = self._read_list(feat_len_list) if feat_len_list else []
        self._next = 0

    @staticmethod
    def _read_list(path):
        base = os.path.dirname(os.path.abspath(path))
        with open(path) as f:
            return [p if os.path.isabs(p) else os.path.join(base, p) for p in (l.strip() for l in f) if p]

    def get_batch_size(self):
        b = self._peek()
        return int(b[0].shape[0]) if b is not None else 0

    def _peek(self):
        n = len(self._batches) if self._batches is not None else min(len(self._feat_paths), len(self._len_paths))
        if self._next >= min(n, self.batch_num):
            return None
        if self._batches is not None:
            return self._batches[self._next]
        feat = torch.from_numpy(np.load(self._feat_paths[self._next], allow_pickle=False).astype(np.float32))
        fl = torch.from_numpy(np.load(self._len_paths[self._next], allow_pickle=False).astype(np.int32))
        return feat, fl

    def get_batch(self, names=None):
        """Next calibration batch as [feat, feat_len] tensors (TensorRT returns device pointers), None when exhausted."""
        b = self._peek()
        if b is None:
            return None
        self._next += 1
        return [b[0], b[1]]

    def __iter__(self):
        self._next = 0
        while True:
            b = self.get_batch()
            if b is None:
                return
            yield b[0], b[1]

    def read_calibration_cache(self):
        if self.cache_file and os.path.exists(self.cache_file):
            with open(self.cache_file) as f:
                return json.load(f)
        return None

    def write_calibration_cache(self, cache):
        if self.cache_file:
            with open(self.cache_file, "w") as f:
                json.dump(cache, f)
