#!/usr/bin/env python3
"""Chunk-by-chunk decoding (m3_engine_forward_chunk) timed: latency of one chunk step and the real-time factor it implies.

  python tools/bench_streaming.py [--chunk 16] [--left-chunks 4] [--batch 1] [--weight-dtype f32] [--seconds 20]

18L x 32e encoder with causal conv modules in both encoders, static_chunk_size = chunk (output frames; one chunk = 4 x chunk
input frames of 10 ms), synthetic weights and features.  Every step after the first is a hipGraph replay (the chunk counter
lives on the device).  Prints one JSON line: ms per chunk (p50 / p99 over all steps, hipEvent pairs on the engine stream),
audio seconds per chunk, real-time factor = compute time / audio time, streams one GPU could serve in real time.
"""
import argparse
import json
import os
import sys

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
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--left-chunks", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--weight-dtype", default="f32")
    ap.add_argument("--layers", type=int, default=18)
    ap.add_argument("--seconds", type=float, default=20.0, help="audio per stream")
    args = ap.parse_args()
    cfg = EncoderConfig(num_blocks=args.layers, causal=True, embed_causal=True, static_chunk_size=args.chunk,
                        num_decoding_left_chunks=args.left_chunks, weight_dtype=args.weight_dtype)
    w = make_weights(cfg, seed=0)
    eng = Engine.from_state_dict(cfg, w, packed_rows=False)
    n_chunks = max(4, int(args.seconds * 100 / (4 * args.chunk)))
    st = eng.streaming(args.batch, n_chunks * args.chunk)
    rng = np.random.default_rng(1234)
    win = torch.from_numpy(rng.random((args.batch, st.window, cfg.input_dim), dtype=np.float32)).to(eng.device)
    valid = torch.full((args.batch,), st.window, dtype=torch.int32, device=eng.device)
    times = []
    for rep in range(3):
        st.reset()
        for n in range(n_chunks):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(eng.stream)
            st.step(win, valid)
            e1.record(eng.stream)
            e1.synchronize()
            if rep > 0:
                times.append(e0.elapsed_time(e1))
    t = np.sort(np.array(times))
    audio_s = 4 * args.chunk * 0.01
    p50 = float(np.median(t))
    out = {"metric": "streaming chunk latency, %dL x %de %s, chunk %d frames (%.2f s of audio), %d left chunks, batch %d" % (
               cfg.num_blocks, cfg.num_experts, args.weight_dtype, args.chunk, audio_s, args.left_chunks, args.batch),
           "ms_per_chunk": {"p50": round(p50, 4), "p99": round(float(t[int(0.99 * (len(t) - 1))]), 4), "min": round(float(t[0]), 4), "n": len(t)},
           "kernels_per_chunk": eng.num_kernels(), "audio_s_per_chunk": audio_s,
           "real_time_factor": round(p50 * 1e-3 / audio_s, 5),
           "streams_in_real_time_one_context": int(args.batch * audio_s / (p50 * 1e-3)),
           "state_MB": round(st.state.numel() / 2 ** 20, 1), "graph_captures": eng.num_captures(), "data": "synthetic"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
