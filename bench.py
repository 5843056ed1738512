#!/usr/bin/env python3
"""Headline benchmark: encoder frames/sec of the 18L x 32e Conformer-MoE encoder on a 206-frame utterance
(BASELINE.json `metric`; workload = configs[1]: 18-layer 32-expert fp32, batch 1 x 206 frames, all experts
local).  One process per GPU; for N > 1 every rank runs its own utterance on its own replica of the
engine (the path shards over independent utterances, SURVEY.md §8e: "B=1 multi-GPU = replicas"), no
data-path collective, weak scaling.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A step = one encoder forward (feat, feat_len resident in HBM -> logits in HBM), replayed as a hipGraph.
Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : the dominant kernel (grouped expert FFN) -- algorithmic bytes per launch / measured
                 HIP-event duration on the engine's stream vs the 8 TB/s HBM peak;
  cpu_baseline : the oracle's plain-torch fp32 forward of the same workload timed on the host cores
                 (kind "port": the reference's CUDA/TensorRT path cannot be built here).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "3m-asr-inference_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); the execution contexts of --streams need one
# each, so ask for 8 before the runtime initialises (a process-level runtime knob, not a machine setting)
# -- unless a profiler is preloaded: rocprofv3 initialises the runtime before this line runs and intercepts every HW
# queue for its counter collection (see DESIGN.md "The --pmc abort"); the override is then skipped
def _profiler_attached():
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ)


if not _profiler_attached():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
else:
    print("bench: profiler attached, GPU_MAX_HW_QUEUES left at %s" % os.environ.get("GPU_MAX_HW_QUEUES", "<default>"), file=sys.stderr)

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=206)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--layers", type=int, default=18)
    ap.add_argument("--experts", type=int, default=32)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-fold-pos", dest="fold_pos", action="store_false",
                    help="recompute linear_pos(pos_emb[:T']) in every forward (it is input-independent: the engine folds it "
                         "at shape-binding time by default, like a constant-folded initializer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--streams", type=int, default=4,
                    help="concurrent execution contexts per GPU (each its own utterance, stream, workspace and hipGraph, "
                         "weights shared); batch stays 1 per context")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads for the CPU baseline (0 = min(32, cores))")
    ap.add_argument("--routing", choices=["balanced", "random"], default="balanced",
                    help="balanced: calibrate the synthetic router weights so tokens spread over the experts "
                         "(a trained 3M-ASR router is load-balanced by its aux losses); random: raw N(0,0.5) init")
    ap.add_argument("--route-mode", choices=["staged", "fused", "split"], default=None,
                    help="router path of the engine (default: staged; see m3asr/engine.py)")
    ap.add_argument("--fuse-route", action="store_true",
                    help="router + SoftmaxTopK + ScatterMapping as one single-workgroup launch per layer (275 instead of "
                         "292 kernels; measured 2-3 %% slower than the staged path, so off by default)")
    ap.add_argument("--weight-dtype", choices=["f32", "bf16", "fp8"], default="f32",
                    help="storage of the GEMM weights (bf16 = BASELINE.json configs[2]; the headline metric is f32)")
    ap.add_argument("--varlen", default="", help="LO-HI: utterance lengths drawn from U[LO,HI] frames (configs[2]: 50-500)")
    ap.add_argument("--profile-stages", action="store_true", help="print per-stage HIP-event times to stderr")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the N > 1 runs (nccl = RCCL; gloo: rehearsal of the multi-rank path "
                         "on a box with fewer GPUs than ranks -- ranks then share devices)")
    ap.add_argument("--packed-rows", choices=["auto", "on", "off"], default="auto",
                    help="ragged batches: run the blocks on the packed valid frames (auto = for batch > 1)")
    return ap.parse_args()


def stage_times(eng, passes=20):
    """HIP-event duration of every stage, recorded on the engine's own stream (averaged over passes)."""
    names = eng.stage_names()
    n = len(names)
    acc = np.zeros(n)
    st = eng.stream
    for _ in range(passes):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record(st)
        for i in range(n):
            eng.run_stages(i, i + 1)
            evs[i + 1].record(st)
        st.synchronize()
        acc += np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n)])
    return names, acc / passes  # ms


def balance_router(eng, cpu_weights):
    """Synthetic-weight calibration (no effect on the code path being timed): random router weights send
    almost every frame of an utterance to the same 3-4 experts because the frames share a large common
    component.  A trained 3M-ASR router is load-balanced (sparse-L1 + importance losses,
    positionwise_feed_forward.py:155-160), so per layer we project the mean router input out of
    router_weights, layer by layer on the device, using the staged engine API.  The resulting expert
    histogram is reported with the result."""
    names = eng.stage_names()
    first = 0
    for li in range(eng.cfg.num_blocks):
        idx = names.index("blocks.%d.moe_router" % li)
        eng.run_stages(first, idx + 1)
        eng.stream.synchronize()
        lens = eng.buffer("lens", torch.int32)
        Bb = lens.numel()
        if Bb > 1 and eng.packed_rows():       # packed layout: the real frames are the first row0[B] rows
            n_real = int(eng.buffer("row0", torch.int32)[Bb])
            emb, xn = eng.buffer("embed").view(-1, eng.cfg.embed_dim)[:n_real], eng.buffer("xn").view(-1, eng.cfg.attention_dim)[:n_real]
            mu = torch.cat([emb.mean(0), xn.mean(0)])
        else:
            emb, xn = eng.buffer("embed").view(Bb, -1, eng.cfg.embed_dim), eng.buffer("xn").view(Bb, -1, eng.cfg.attention_dim)
            valid = (torch.arange(emb.shape[1], device=lens.device).view(1, -1) < lens.view(-1, 1))   # real (unpadded) frames
            mu = torch.cat([emb[valid].mean(0), xn[valid].mean(0)])
        w = eng.weights["blocks.%d.feed_forward.router_weights_t" % li]          # [E, De + D]
        w -= torch.outer(w @ mu, mu) / (mu @ mu)
        eng.run_stages(idx, idx + 1)                                              # logits with the new weights
        first = idx + 1
        cpu_weights["blocks.%d.feed_forward.router_weights" % li] = w.t().contiguous().cpu()
    eng.run_stages(first, len(names))
    eng.stream.synchronize()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    if args.dist_backend == "gloo":          # rehearsal: more ranks than devices is fine, they share
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:      # host-side weight generation / packing: do not oversubscribe the cores with N ranks x all threads
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend="gloo")

    from m3asr.config import EncoderConfig, subsampled_len
    from m3asr.weights import make_weights
    from m3asr.engine import Engine

    cfg = EncoderConfig(num_blocks=args.layers, num_experts=args.experts, weight_dtype=args.weight_dtype)
    weights = make_weights(cfg, seed=0)

    # synthetic input: U[0,1) features as data/generate_trtexec_inputs.py:7 of the reference; each rank its own utterance
    rng = np.random.default_rng(1234 + rank)
    B, T = args.batch, args.frames
    if args.varlen:
        lo, hi = (int(v) for v in args.varlen.split("-"))
        lengths = rng.integers(lo, hi + 1, B)
        lengths[0] = hi                      # the padded length is always HI
        T = int(lengths.max())
    else:
        lengths = np.full(B, T)
    n_frames = int(lengths.sum())            # real (unpadded) input frames per step per rank
    feat_cpu = torch.from_numpy(rng.random((B, T, cfg.input_dim), dtype=np.float32))
    feat = feat_cpu.to(dev)
    feat_len = torch.from_numpy(lengths.astype(np.int32)).view(1, B).to(dev)
    packed = {"auto": None, "on": True, "off": False}[args.packed_rows]
    # the staged-route engine exposes xn / the router stage, which the synthetic-router calibration needs
    eng = Engine.from_state_dict(cfg, weights, device=dev, fold_pos_proj=args.fold_pos, fuse_route=False, packed_rows=packed)
    if args.routing == "balanced":
        eng.bind(feat, feat_len)
        balance_router(eng, weights)         # updates the device weights in place and the CPU state_dict
    route = {"staged": 0, "fused": 1, "split": 2}[args.route_mode] if args.route_mode else (1 if args.fuse_route else 0)
    if route:                                # rebuild from the calibrated state_dict
        del eng
        torch.cuda.empty_cache()
        eng = Engine.from_state_dict(cfg, weights, device=dev, fold_pos_proj=args.fold_pos, fuse_route=route, packed_rows=packed)
    if not (rank == 0 and world == 1 and not args.no_cpu_baseline):
        weights = None
    eng.bind(feat, feat_len)
    eng.forward(use_graph=False)
    eng.stream.synchronize()
    use_graph = not args.no_graph
    # extra execution contexts: same weights, own utterance / stream / workspace / graph
    ctxs = [eng]
    for si in range(1, args.streams):
        c = eng.clone_context(fold_pos_proj=args.fold_pos, fuse_route=route, packed_rows=packed)
        f2 = torch.from_numpy(np.random.default_rng(5000 + 97 * rank + si).random((B, T, cfg.input_dim), dtype=np.float32)).to(dev)
        c.bind(f2, feat_len.clone())
        ctxs.append(c)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup * len(ctxs)):
        ctxs[i % len(ctxs)].forward(use_graph=use_graph)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ctxs[i % len(ctxs)].forward(use_graph=use_graph)
    for c in ctxs:
        c.stream.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    # latency of one forward when it has the GPU to itself (one context)
    t1 = time.perf_counter()
    for _ in range(50):
        eng.forward(use_graph=use_graph)
    eng.stream.synchronize()
    latency_ms = (time.perf_counter() - t1) / 50 * 1e3
    frames_per_step = world * n_frames
    value = frames_per_step / (dt / args.steps)

    # ---- roofline of the dominant kernel, measured live with HIP events on the engine's stream ----
    roofline = None
    if rank == 0:
        names, ms = stage_times(eng)
        if args.profile_stages:
            for n_, m_ in zip(names, ms):
                if not (n_.startswith("blocks.") or n_.startswith("embed.blocks.")) or ".0." in n_ or ".9." in n_:
                    print("%-40s %8.2f us" % (n_, m_ * 1e3), file=sys.stderr)
            print("sum of stages %.3f ms, %d kernels" % (ms.sum(), eng.num_kernels()), file=sys.stderr)
        idx = [i for i, n_ in enumerate(names) if n_.endswith("moe_local.expert")]
        D, F, E, S = cfg.attention_dim, cfg.hidden_units, cfg.num_experts, B * subsampled_len(T)
        # algorithmic bytes per launch: touched experts x (2DF + F + D) x 4 B  +  S x (D in + D out) x 4 B.
        # acc_histogram of the last layer is live in the workspace; touched counts differ per layer by +-2, so
        # use the per-layer routing recorded in gate_idx.
        touched = []
        for li in range(cfg.num_blocks):
            g = eng.buffer("blocks.%d.gate_idx" % li, torch.int32).cpu().numpy()
            touched.append(len(np.unique(g[g >= 0])))
        wsz = {"f32": 4, "bf16": 2, "fp8": 1}[cfg.weight_dtype]   # expert weights in weight_dtype, biases / scales / rows fp32
        extra = (F + D) * 4 * (2 if cfg.weight_dtype == "fp8" else 1)   # biases (+ per-row scales)
        bytes_alg = np.array([t_ * (2 * D * F * wsz + extra) + S * 2 * D * 4 for t_ in touched], dtype=np.float64)
        # duration of the roofline kernel IN SITU: whole forwards are enqueued stage by stage on the engine stream
        # (the GPU stays the bottleneck: ~3.5 us host cost per launch vs ~8 us per kernel) with HIP events only around
        # each layer's expert launch, so the kernel sees the cache state of a real forward -- repeated in isolation its
        # ~100 MB of weights would sit in the 256 MB Infinity Cache and read 25 % faster
        st = eng.stream
        acc_t = np.zeros(len(idx))
        passes = 20
        for _ in range(passes):
            evs, cur = [], 0
            for i_ in idx:
                eng.run_stages(cur, i_)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                eng.run_stages(i_, i_ + 1)
                e1.record(st)
                evs.append((e0, e1))
                cur = i_ + 1
            eng.run_stages(cur, len(names))
            st.synchronize()
            acc_t += np.array([a.elapsed_time(b) for a, b in evs])
        dur = acc_t / passes * 1e-3
        achieved = float((bytes_alg / dur).mean() / 1e9)
        # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the
        # SAME kernel at the same shape with HBM-cold weights (tools/pmc_expert.py -> profiles/r01_pmc_expert.json; the
        # full bench.py segfaults inside the profiler's counter collection on this pool), scaled by the touched-expert
        # count of this run (the probe's routing touched 25.33 experts per layer)
        traffic = None
        kname = {"f32": "expert_ffn_f32_kernel", "bf16": "expert_ffn_bf16w_kernel", "fp8": "expert_ffn_w8_kernel"}[cfg.weight_dtype]
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_expert.json")
        if os.path.exists(pmc) and B == 1 and T == 206:
            try:
                ent = json.load(open(pmc))[cfg.weight_dtype]
                per = [v for k, v in ent["kernels"].items() if "expert_ffn" in k][0]["traffic_bytes_per_launch"]
                traffic = int(per * float(np.mean(touched)) / ent["meta"]["experts_touched_mean"])
            except Exception:
                traffic = None
        roofline = {"kernel": kname, "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                    "avg_launch_us": round(float(dur.mean() * 1e6), 2),
                    "alg_bytes_per_launch": int(bytes_alg.mean()),
                    "experts_touched_mean": round(float(np.mean(touched)), 2), "experts_touched": touched}

    # ---- CPU baseline: the oracle (plain-torch fp32 restatement) on the host cores, rank 0, N=1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.encoder_ref import encoder_forward
        fl_cpu = torch.from_numpy(lengths.astype(np.int32))
        cores = args.cpu_threads if args.cpu_threads > 0 else min(32, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        times = []
        encoder_forward(weights, cfg, feat_cpu, fl_cpu)
        t_end = time.perf_counter() + args.cpu_seconds
        while time.perf_counter() < t_end and len(times) < 50:
            c0 = time.perf_counter()
            ref_logits = encoder_forward(weights, cfg, feat_cpu, fl_cpu)
            times.append(time.perf_counter() - c0)
        med = float(np.median(times))
        cpu = {"value": round(n_frames / med, 1), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": "%d full forwards of the same %dx%d-frame %dL/%de workload (median %.1f ms), torch %s fp32" % (
                   len(times), B, T, cfg.num_blocks, cfg.num_experts, med * 1e3, torch.__version__)}
        # the checker: GPU logits of the timed workload vs the oracle's
        got = eng._bound[2].cpu()
        vmask = (torch.arange(got.shape[1]).view(1, -1) < torch.tensor([subsampled_len(int(l)) for l in lengths]).view(-1, 1))
        rel = float(((got - ref_logits).abs() / (ref_logits.abs() + 2e-1))[vmask].max())
        cpu["gpu_vs_oracle_max_rel"] = round(rel, 6)
        if cfg.weight_dtype != "f32":
            # 16-bit mode: the calibrated synthetic routers are near-ties by construction, so some tokens pick another
            # expert than in fp32 and differ by a whole expert FFN.  Numeric error is therefore reported with the oracle
            # teacher-forced to the engine's expert choices, next to the fraction of identical choices.
            Tp = got.shape[1]
            forced = {"blocks.%d.gate_idx" % i: eng.rows_padded("blocks.%d.gate_idx" % i, torch.int32, fill=-1).cpu().view(B, Tp, 1).clone()
                      for i in range(cfg.num_blocks)}
            ref_forced = encoder_forward(weights, cfg, feat_cpu, fl_cpu, route_override=forced)
            relf = float(((got - ref_forced).abs()[vmask].max()) / ref_forced.abs()[vmask].max())
            free = {}
            encoder_forward(weights, cfg, feat_cpu, fl_cpu, taps=free)
            same = sum(int((forced[k].view(B, Tp)[vmask] == free[k].view(B, Tp)[vmask]).sum()) for k in forced)
            cpu["gpu_vs_oracle_forced_routing_max_err_over_max_logit"] = round(relf, 6)
            cpu["routing_agreement_with_fp32"] = round(same / float(int(vmask.sum()) * cfg.num_blocks), 4)

    if rank == 0:
        metric = "encoder frames/sec, 18Lx32e Conformer-MoE, 206-frame utterance"
        if args.varlen or B != 1 or T != 206:
            metric = "encoder frames/sec, %dLx%de Conformer-MoE, batch=%d %s" % (
                cfg.num_blocks, cfg.num_experts, B, ("var-len %s frames" % args.varlen) if args.varlen else "%d-frame utterances" % T)
        out = {"metric": metric,
               "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": cfg.weight_dtype, "data": "synthetic",
               "config": {"workload": "%d-layer %d-expert %s, batch=%dx%d frames per GPU%s, all experts local "
                                      "(BASELINE.json configs[%d])" % (
                                          cfg.num_blocks, cfg.num_experts,
                                          {"f32": "fp32", "bf16": "bf16 weights / bf16 MFMA / fp32 accumulate + activations",
                                           "fp8": "fp8 (e4m3) expert weights + bf16 dense weights / bf16 MFMA / fp32 accumulate "
                                                  "+ activations"}[cfg.weight_dtype], B, T,
                                          (" (lengths U[%s], %d real frames)" % (args.varlen, n_frames)) if args.varlen else "",
                                          {"f32": 1, "bf16": 2, "fp8": 4}[cfg.weight_dtype]),
                          "layers": cfg.num_blocks, "experts": cfg.num_experts, "frames": T, "batch_per_gpu": B,
                          "parallelism": "replicas x%d" % world, "streams_per_gpu": len(ctxs),
                          "latency_ms_one_stream": round(latency_ms, 4), "hip_graph": use_graph,
                          "kernels_per_forward": eng.num_kernels(), "fold_pos_proj": bool(args.fold_pos),
                          "routing": args.routing, "route_mode": ["staged", "fused", "split"][route],
                          "packed_rows": bool(B > 1 and eng.packed_rows())},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
