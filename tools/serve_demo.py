#!/usr/bin/env python3
"""Variable-length serving loop on one MI355X: requests of random length are padded up to a length bucket, each bucket has
static device buffers and its own captured hipGraph (Engine.infer + the native shape cache), so after the first request of
a bucket nothing is re-captured.  Development / illustration tool (GPU box):

    python tools/serve_demo.py [--layers 18] [--requests 200] [--bucket 32] [--weight-dtype f32|bf16|fp8]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import numpy as np
import torch

from m3asr.config import EncoderConfig
from m3asr.engine import Engine
from m3asr.weights import make_weights


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=18)
    ap.add_argument("--requests", type=int, default=200)
    ap.add_argument("--bucket", type=int, default=32, help="frames per length bucket")
    ap.add_argument("--min-frames", type=int, default=100)
    ap.add_argument("--max-frames", type=int, default=300)
    ap.add_argument("--weight-dtype", choices=["f32", "bf16", "fp8"], default="f32")
    a = ap.parse_args()
    cfg = EncoderConfig(num_blocks=a.layers, weight_dtype=a.weight_dtype)
    eng = Engine.from_state_dict(cfg, make_weights(cfg, seed=0))
    rng = np.random.default_rng(0)
    lengths = rng.integers(a.min_frames, a.max_frames + 1, a.requests)
    frames, t0 = 0, None
    for i, n in enumerate(lengths):
        T = int(-(-n // a.bucket) * a.bucket)                      # pad up to the bucket
        feat = torch.zeros(1, T, cfg.input_dim)
        feat[0, :n] = torch.from_numpy(rng.random((int(n), cfg.input_dim), dtype=np.float32))
        out = eng.infer(feat, torch.tensor([int(n)], dtype=torch.int32))   # logits of the valid frames: out[0, :T'(n)]
        if i == a.requests // 4:                                    # every bucket has been seen by now (probably)
            torch.cuda.synchronize()
            t0, frames = time.perf_counter(), 0
        frames += int(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    buckets = len({int(-(-n // a.bucket)) for n in lengths})
    print("requests %d, length buckets %d, graphs captured %d, steady-state %.0f frames/s (%.2f ms per request, one stream, "
          "host copies included)" % (a.requests, buckets, eng.num_captures(), frames / dt,
                                     dt / (a.requests - a.requests // 4) * 1e3))


if __name__ == "__main__":
    main()
