#!/usr/bin/env python3
"""What does a hipGraph of N dependent kernels cost when the kernels do (almost) nothing?  (development tool, GPU box only)

The B=1 forward is 295 dependent launches of 1-30 us each.  This probe replays graphs of N trivial kernels -- (a) one
work-group (m3_mask_conv2d_sample on 1 length), (b) 256 work-groups moving 256 KB (m3_scale) -- on 1..4 streams, the
same way bench.py replays the encoder graphs, and prints the time per graph and per kernel: the floor that the
dependency chain + the command processors put under any 295-launch forward, whatever the kernels do."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))
import torch  # noqa: E402
from m3asr import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 295
REPLAYS = 200


def build(kind, stream):
    lens = torch.tensor([206], dtype=torch.int32, device="cuda")
    x = torch.rand(256 * 256, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for _ in range(3):
            (ops.mask_conv2d_sample(lens, 2, 2) if kind == "1wg" else ops.scale(x, 1.0))
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream):
            y = x
            for _ in range(N):
                if kind == "1wg":
                    lens2 = ops.mask_conv2d_sample(lens, 2, 2)   # independent inputs, but a graph captured on one
                else:                                            # stream serialises its nodes anyway
                    y = ops.scale(y, 1.0)
    return g


def main():
    out = {"kernels_per_graph": N}
    for kind in ("1wg", "256wg"):
        for n_streams in (1, 2, 4):
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            graphs = [build(kind, s) for s in streams]
            for i in range(8 * n_streams):
                with torch.cuda.stream(streams[i % n_streams]):
                    graphs[i % n_streams].replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(REPLAYS):
                with torch.cuda.stream(streams[i % n_streams]):
                    graphs[i % n_streams].replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / REPLAYS
            out["%s_streams%d" % (kind, n_streams)] = {"ms_per_graph": round(dt * 1e3, 4), "us_per_kernel": round(dt * 1e6 / N, 3)}
            print(kind, "streams", n_streams, "-> %.3f ms per graph, %.2f us per kernel" % (dt * 1e3, dt * 1e6 / N), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "dispatch_floor.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
