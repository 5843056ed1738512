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
  roofline        : the kernel family that holds the largest share of the forward's GPU time (picked from the in-run
                    per-stage HIP-event timings, m3_engine_stage_info gives each stage's kernel and algorithmic bytes /
                    FLOPs) -- algorithmic bytes (or FLOPs) / measured duration vs the 8 TB/s HBM (or MFMA) peak;
  roofline_expert : the same for the north-star kernel, the grouped expert FFN, timed in situ;
  forward         : whole-forward fractions -- algorithmic bytes / one-stream latency / HBM peak, FLOPs / MFMA peak --
                    and the p50 / p99 latency of >= 50 hipEvent-timed forwards;
  cpu_baseline    : the oracle's plain-torch fp32 forward of the same workload timed on the host cores
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


def _counters_collected():
    """rocprofv3 --pmc / -i counter file (it exports ROCPROF_COUNTER_COLLECTION / ROCPROF_COUNTERS to the profiled process).
    Only counter collection needs the bounded-dispatch mode (--pmc-safe); --kernel-trace / --stats runs keep the real
    configuration (4 contexts, graph replay), so their kernel statistics describe the headline run."""
    v = os.environ.get("ROCPROF_COUNTER_COLLECTION", "")
    return bool(os.environ.get("ROCPROF_COUNTERS")) or v not in ("", "0", "False", "false")


os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

T_START = time.perf_counter()


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
    ap.add_argument("--fp8-activations", action="store_true",
                    help="with --weight-dtype fp8: fp8 ARITHMETIC in the grouped expert FFN (e4m3 activations, fp8 MFMA); the "
                         "per-layer H scales are calibrated on the benchmark batch before timing (m3asr.calibrate)")
    ap.add_argument("--varlen", default="", help="LO-HI: utterance lengths drawn from U[LO,HI] frames (configs[2]: 50-500)")
    ap.add_argument("--profile-stages", action="store_true", help="print per-stage HIP-event times to stderr")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the N > 1 runs (nccl = RCCL; gloo: rehearsal of the multi-rank path "
                         "on a box with fewer GPUs than ranks -- ranks then share devices)")
    ap.add_argument("--ep", action="store_true",
                    help="expert-parallel mode (BASELINE.json configs[3] / [4]): the experts of every layer are sharded "
                         "E / N per rank, every rank keeps --batch utterances, tokens travel by all-to-all (RCCL with "
                         "--dist-backend nccl); value = frames of all ranks / max-over-ranks time")
    ap.add_argument("--no-ep-probe", action="store_true",
                    help="N > 1 replica runs end with a short expert-parallel forward over the same process group, reported "
                         "on stderr and in gpurun_out/ep_probe_nN.json AFTER the JSON line; this switches it off")
    ap.add_argument("--pmc-safe", action="store_true",
                    help="for `rocprofv3 --pmc ... -- python3 bench.py --pmc-safe` (set automatically when counter collection is "
                         "detected): one execution context, plain launches, a device synchronise per forward -- the profiler's "
                         "counter-collection thread aborts when thousands of un-synchronised dispatches are in flight (DESIGN.md 6)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed loop of --steps forwards is run this many times (each bracketed by barrier + synchronize); "
                         "value / ms_per_step are the MEDIAN repeat, every repeat is listed in config.ms_per_step_repeats")
    ap.add_argument("--ep-probe-strict", action="store_true",
                    help="N > 1 replica runs: exit non-zero when the expert-parallel probe after the headline line fails or times "
                         "out (default: the outcome is recorded in gpurun_out/ep_probe_nN.json and on stderr, exit code 0)")
    ap.add_argument("--latency-iters", type=int, default=100, help="hipEvent-timed single forwards for p50 / p99 (>= 50)")
    ap.add_argument("--fork-embed", choices=["auto", "on", "off"], default="auto",
                    help="embed encoder as a second branch of the captured graph beside the main encoder's start (auto = off: it "
                         "shortens one forward by 2.5 %% but eight concurrently active queues collapse the 4-context throughput)")
    ap.add_argument("--packed-rows", choices=["auto", "on", "off"], default="auto",
                    help="ragged batches: run the blocks on the packed valid frames (auto = for batch > 1)")
    ap.add_argument("--ep-full-wire", action="store_true",
                    help="--ep: every wire chunk holds ALL rows of the largest rank (the round-2/3 shape) instead of the bounded wire")
    ap.add_argument("--ep-probe-inject-failure", action="store_true",
                    help="diagnostic (tests): make the expert-parallel probe raise, to show that a failing probe is visible")
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
    # everything below is issued on the ENGINE'S stream: the router stage must see the new weights (with the element-wise
    # ops on torch's default stream they raced with it, and the calibrated routers -- hence the experts touched -- differed
    # from run to run: 21.8 on average under the profiler's serialised dispatch, 25.6 otherwise)
    with torch.cuda.stream(eng.stream):
        for li in range(eng.cfg.num_blocks):
            idx = names.index("blocks.%d.moe_router" % li)
            eng.run_stages(first, idx + 1)
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
            # (element-wise ops + reductions only: no GEMM / GEMV call, so the process never loads rocBLAS / Tensile code objects)
            w -= ((w * mu).sum(1, keepdim=True) * mu) / (mu * mu).sum()
            eng.run_stages(idx, idx + 1)                                              # logits with the new weights
            first = idx + 1
            cpu_weights["blocks.%d.feed_forward.router_weights" % li] = w.t().contiguous().cpu()   # (synchronises the stream)
        eng.run_stages(first, len(names))
    eng.stream.synchronize()


def run_ep(args, rank, world, dev, dist, weights_full, steps, warmup, wdt, B, varlen, frames, balanced=True):
    """Expert-parallel forwards (m3asr/ep.py): experts sharded E / world per rank, batch sharded B per rank, two fixed-shape
    all-to-alls per MoE layer, no host synchronisation inside a forward.  Returns a dict (rank 0: the measurement)."""
    from m3asr.config import EncoderConfig
    from m3asr.engine import Engine
    from m3asr.ep import ExpertParallelEncoder
    E, L = args.experts, args.layers
    assert E % world == 0, "--experts %d must divide over %d ranks" % (E, world)
    full = EncoderConfig(num_blocks=L, num_experts=E, weight_dtype=wdt)
    rng = np.random.default_rng(4321 + rank)
    if varlen:
        lo, hi = (int(v) for v in varlen.split("-"))
        lengths = rng.integers(lo, hi + 1, B)
        lengths[0] = hi
        T = hi
    else:
        lengths, T = np.full(B, frames), frames
    feat = torch.from_numpy(rng.random((B, T, full.input_dim), dtype=np.float32)).to(dev)
    feat_len = torch.from_numpy(lengths.astype(np.int32)).view(1, B).to(dev)
    on_host = world > 1 and dist.get_backend() == "gloo"
    if balanced:
        # load-balanced synthetic routers: calibrated on rank 0's utterances with an all-experts-local engine, then the
        # 18 router matrices are broadcast (dense weights are replicated across expert-parallel ranks)
        keys = ["blocks.%d.feed_forward.router_weights" % i for i in range(L)]
        if rank == 0:
            cal = Engine.from_state_dict(full, weights_full, device=dev, fuse_route=False)
            cal.bind(feat, feat_len)
            balance_router(cal, weights_full)
            del cal
            torch.cuda.empty_cache()
        if world > 1:
            for k in keys:
                t = weights_full[k].contiguous() if on_host else weights_full[k].to(dev).contiguous()
                dist.broadcast(t, src=0)
                weights_full[k] = t.cpu()
    fp8a = bool(getattr(args, "fp8_activations", False)) and wdt == "fp8"
    h_scales = None
    if fp8a:
        # static H scale per MoE layer (m3asr/calibrate.py), from rank 0's batch with all experts local, then broadcast:
        # every rank must quantise H with the same scale as the single-GPU engine would
        from m3asr.calibrate import calibrate_h_scales
        hs = torch.zeros(L, dtype=torch.float32)
        if rank == 0:
            hs = torch.tensor(calibrate_h_scales(full, weights_full, [(feat, feat_len)], device=dev), dtype=torch.float32)
            torch.cuda.empty_cache()
        if world > 1:
            t = hs if on_host else hs.to(dev)
            dist.broadcast(t, src=0)
            hs = t.cpu()
        for i in range(L):
            weights_full["blocks.%d.feed_forward.experts.h_scale" % i] = hs[i:i + 1].clone()
        h_scales = [round(float(hs.min()), 6), round(float(hs.max()), 6)]
    cfg = EncoderConfig(num_blocks=L, num_experts=E // world, ep_world_size=world, ep_rank=rank, weight_dtype=wdt,
                        fp8_activations=fp8a)
    eng = Engine.from_state_dict(cfg, weights_full, device=dev, ep_stages=True)
    # bounded wire: chunks of 2 x (rows / world) instead of all rows; a chunk that overflows is reported on the device and the
    # binding is redone with a capacity that fits BEFORE the timed region (rows are never dropped)
    ep = ExpertParallelEncoder(eng, graph=not args.no_graph, capacity_factor=(None if args.ep_full_wire else 2.0))
    ep.forward(feat, feat_len)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ep.enqueue()          # the first forward of a binding runs eagerly and (by default) captures the whole forward as one graph
    sync()
    for _ in range(warmup):
        ep.enqueue()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ep.enqueue()
    eng.stream.synchronize()
    sync()
    dt = time.perf_counter() - t0
    frames_all = torch.tensor([float(lengths.sum())], dtype=torch.float64, device="cpu" if (on_host or world == 1) else dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=frames_all.device)
    if world > 1:
        dist.all_reduce(frames_all, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    _, S, D, cap = ep._bound
    overflow_after = ep.overflow_needed() if not args.ep_full_wire else 0
    routed_rows = int((torch.stack([eng.buffer("blocks.%d.gate_idx" % li, torch.int32) for li in range(L)]) >= 0).sum().item()) / max(L, 1)
    kern = {st["name"]: st["kernel"] for st in eng.stage_info()}
    touched = []
    for li in range(L):
        g = eng.buffer("blocks.%d.gate_idx" % li, torch.int32).cpu().numpy()
        touched.append(len(np.unique(g[g >= 0])))
    return {"value": float(frames_all.item()) / (dt / steps), "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "frames_per_step_all_ranks": int(frames_all.item()), "batch_per_gpu": B, "padded_frames": T,
            "experts_per_gpu": E // world, "weight_dtype": wdt, "rows_per_rank": S, "packed_rows": bool(B > 1 and eng.packed_rows()),
            "wire": {"capacity_rows": cap, "bytes_per_exchange_per_rank": int(world * (cap + 1) * D * 4), "wire_dtype": "f32",
                     "needed_bytes_per_exchange_per_rank": int(routed_rows * D * 4),
                     "capacity_policy": ("all rows per chunk (cannot overflow)" if args.ep_full_wire else
                                         "2 x rows / world per chunk, repeated with a larger wire on overflow"),
                     "forwards_repeated_for_overflow": ep.reruns, "overflow_after_timed_region": overflow_after,
                     "collectives_per_forward": 2 * L, "host_syncs_per_forward": ep.host_syncs_per_forward(),
                     "backend": (dist.get_backend() if world > 1 else "none (one rank: device copy)")},
            "forward_graph": ep.graph_state, "expert_kernel": kern.get("blocks.0.moe_ep.expert"),
            "fp8_activations": fp8a, "h_scale_min_max": h_scales,
            "global_experts_touched_by_rank0_tokens_mean": round(float(np.mean(touched)), 2),
            "kernels_per_forward_native_stages": eng.num_kernels()}


def build_info():
    """Toolchain + the property the packed-FP32 mitigation rests on, read off the BUILT library."""
    info = {"abi": None, "hipcc": None, "packed_fp32_instructions": None}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import check_device_isa
        from m3asr import _lib
        info["abi"] = _lib.load().m3_abi_version()
        info["hipcc"] = check_device_isa.hipcc_version()
        info["packed_fp32_instructions"] = check_device_isa.scan(_lib.LIB_PATH)["packed_fp32"]
    except Exception as ex:       # noqa: BLE001  (llvm-objdump missing on a box: say so, do not fail the measurement)
        info["error"] = repr(ex)
    return info


def main():
    args = parse()
    downgraded = False
    if args.pmc_safe or _counters_collected():
        # rocprofv3 --pmc serialises and instruments every dispatch; its counter-collection thread aborts (SIGSEGV) once
        # thousands of un-synchronised dispatches are in flight (50 back-to-back forwards x 295 kernels; <= 1 770 survive --
        # DESIGN.md 6, the one root cause on record).  Counter passes therefore run one context, plain launches and a device
        # synchronise per forward; --kernel-trace / --stats runs are NOT downgraded.
        args.streams = 1
        args.no_graph = True
        args.latency_iters = 50
        args.pmc_safe = True
        downgraded = True
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (rank 0 disassembles the built library for the JSON line's `build` entry NOW, ~3 s: not between the timed region and the
    #  headline line, where a peer that has already left would get this rank killed by the launcher before it prints)
    build = build_info() if rank == 0 else None
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    if args.dist_backend == "gloo":          # rehearsal: more ranks than devices is fine, they share
        local_rank %= torch.cuda.device_count()
    dev = "cuda:%d" % local_rank
    if world > 1:      # host-side weight generation / packing: do not oversubscribe the cores with N ranks x all threads
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    prof = _profiler_attached()

    def phase(msg):
        if prof:
            print("bench[%.1fs]: %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)

    from m3asr.config import EncoderConfig, subsampled_len
    from m3asr.weights import make_weights

    # host-side work first (weights are generated on the CPU), the device is touched afterwards
    if args.fp8_activations and args.weight_dtype != "fp8":
        raise SystemExit("--fp8-activations needs --weight-dtype fp8")
    cfg = EncoderConfig(num_blocks=args.layers, num_experts=args.experts, weight_dtype=args.weight_dtype,
                        fp8_activations=bool(args.fp8_activations))
    weights = make_weights(cfg, seed=0)
    phase("weights generated on the host")
    torch.cuda.set_device(local_rank)
    phase("device selected")
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend="gloo")

    from m3asr.engine import Engine

    if args.ep:
        r = run_ep(args, rank, world, dev, dist if world > 1 else None, weights, args.steps, args.warmup, args.weight_dtype,
                   args.batch, args.varlen, args.frames, balanced=args.routing == "balanced")
        if rank == 0:
            full18 = args.layers == 18
            if full18 and args.weight_dtype == "bf16" and args.experts == 32 and world == 8 and args.batch == 2 and args.varlen == "50-500":
                which = "BASELINE.json configs[3]"
            elif full18 and args.weight_dtype == "fp8" and args.experts == 64 and world == 8 and args.batch == 8 and args.varlen == "50-500":
                which = "BASELINE.json configs[4] (%s)" % EncoderConfig(weight_dtype="fp8", fp8_activations=bool(args.fp8_activations)).fp8_label()
            else:
                which = "not a BASELINE.json config (same path at another size)"
            out = {"metric": "encoder frames/sec, %dLx%de Conformer-MoE, expert parallel, batch=%d per GPU %s" % (
                       args.layers, args.experts, args.batch, ("var-len %s frames" % args.varlen) if args.varlen else "%d-frame utterances" % args.frames),
                   "value": round(r["value"], 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(r["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": args.weight_dtype, "data": "synthetic",
                   "config": {"workload": "%d-layer %d-expert %s, expert-parallel %d experts/GPU x %d GPU, batch %d per GPU (%s)" % (
                                  args.layers, args.experts, args.weight_dtype, r["experts_per_gpu"], world, args.batch, which),
                              "parallelism": "ep%d x dp%d" % (world, world), "hip_graph": r["forward_graph"] in ("captured", "engine graph"), **{k: v for k, v in r.items() if k not in ("value", "ms_per_step", "steps", "warmup")}},
                   "roofline": None, "cpu_baseline": None}
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

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
    fork = {"auto": None, "on": True, "off": False}[args.fork_embed]
    # the staged-route engine exposes xn / the router stage, which the synthetic-router calibration needs
    eng = Engine.from_state_dict(cfg, weights, device=dev, fold_pos_proj=args.fold_pos, fuse_route=False, packed_rows=packed, fork_embed=fork)
    phase("engine built (plan packed, weights on the device)")
    if args.routing == "balanced":
        eng.bind(feat, feat_len)
        balance_router(eng, weights)         # updates the device weights in place and the CPU state_dict
        phase("synthetic routers calibrated")
    route = {"staged": 0, "fused": 1, "split": 2}[args.route_mode] if args.route_mode else (1 if args.fuse_route else 0)
    fp8_h_scales = None
    if args.fp8_activations:                 # static H scale per MoE layer, from this batch (untimed set-up, like a builder run)
        from m3asr.calibrate import calibrate_h_scales
        del eng
        torch.cuda.empty_cache()
        h_scales = calibrate_h_scales(cfg, weights, [(feat, feat_len)], device=dev)
        torch.cuda.empty_cache()
        phase("H scales calibrated (%.3g .. %.3g)" % (min(h_scales), max(h_scales)))
        fp8_h_scales = [round(float(min(h_scales)), 6), round(float(max(h_scales)), 6)]
        eng = Engine.from_state_dict(cfg, weights, device=dev, fold_pos_proj=args.fold_pos, fuse_route=0, packed_rows=packed, fork_embed=fork)
    if route:                                # rebuild from the calibrated state_dict
        del eng
        torch.cuda.empty_cache()
        eng = Engine.from_state_dict(cfg, weights, device=dev, fold_pos_proj=args.fold_pos, fuse_route=route, packed_rows=packed, fork_embed=fork)
    if not ((rank == 0 and world == 1 and not args.no_cpu_baseline) or (world > 1 and not args.no_ep_probe)):
        weights = None
    eng.bind(feat, feat_len)
    eng.forward(use_graph=False)
    eng.stream.synchronize()
    phase("first forward done")
    use_graph = not args.no_graph
    # extra execution contexts: same weights, own utterance / stream / workspace / graph
    ctxs = [eng]
    for si in range(1, args.streams):
        c = eng.clone_context(fold_pos_proj=args.fold_pos, fuse_route=route, packed_rows=packed, fork_embed=fork)
        f2 = torch.from_numpy(np.random.default_rng(5000 + 97 * rank + si).random((B, T, cfg.input_dim), dtype=np.float32)).to(dev)
        c.bind(f2, feat_len.clone())
        ctxs.append(c)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    phase("contexts ready")
    for i in range(args.warmup * len(ctxs)):
        ctxs[i % len(ctxs)].forward(use_graph=use_graph)
    phase("warm-up enqueued")
    barrier()
    phase("warm-up done (device synchronised)")
    # the timed region: EXACTLY --steps forwards between barrier + synchronize on both sides, max over ranks; repeated
    # --repeats times so that a short run (the driver's 20 steps = 20 ms) is not a single 20 ms sample: value = median repeat
    rep_dt = []
    for _ in range(max(1, 1 if args.pmc_safe else args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ctxs[i % len(ctxs)].forward(use_graph=use_graph)
            if args.pmc_safe:
                ctxs[i % len(ctxs)].stream.synchronize()
        for c in ctxs:
            c.stream.synchronize()
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        rep_dt.append(d)
    dt = float(np.median(rep_dt))
    ms_per_step = dt / args.steps * 1e3
    phase("timed region done")
    # the driver's default run times 20 forwards per repeat (20 ms samples): a 200-forward region beside it, same contexts, same
    # bracketing, reported as config.value_200_steps (never as `value`)
    long_dt = []
    if args.steps < 200 and not args.pmc_safe and world == 1:
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            for i in range(200):
                ctxs[i % len(ctxs)].forward(use_graph=use_graph)
            for c in ctxs:
                c.stream.synchronize()
            barrier()
            long_dt.append(time.perf_counter() - t0)
    # latency of one forward when it has the GPU to itself (one context): back-to-back mean, and the distribution of
    # individually hipEvent-timed forwards (events recorded on the engine's own stream, one forward in flight at a time)
    t1 = time.perf_counter()
    n_lat = 5 if args.pmc_safe else 50
    for _ in range(n_lat):
        eng.forward(use_graph=use_graph)
        if args.pmc_safe:          # counter collection: keep the dispatches in flight bounded
            eng.stream.synchronize()
    eng.stream.synchronize()
    latency_ms = (time.perf_counter() - t1) / n_lat * 1e3
    phase("back-to-back forwards done")
    lat = []
    for _ in range(max(50, args.latency_iters)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
        eng.forward(use_graph=use_graph)
        e1.record(eng.stream)
        e1.synchronize()
        lat.append(e0.elapsed_time(e1))
    lat = np.sort(np.array(lat))
    phase("hipEvent-timed forwards done")
    frames_per_step = world * n_frames
    value = frames_per_step / (dt / args.steps)

    # ---- rooflines, measured live with HIP events on the engine's stream ----
    roofline = roofline_expert = forward = None
    if rank == 0:
        D, F, E, S = cfg.attention_dim, cfg.hidden_units, cfg.num_experts, B * subsampled_len(T)
        info = eng.stage_info()
        names, ms = stage_times(eng)
        live = S                                   # rows the blocks really work on (packed ragged batch: the valid frames)
        if B > 1 and eng.packed_rows():
            live = int(eng.buffer("row0", torch.int32)[B])
        # experts each layer touched (from the routing taps of the timed workload) -> algorithmic bytes of its launch
        # (SURVEY 8d): touched x (2DF weights + biases [+ scales]) + live rows x (D in + D out) x 4 B
        touched = []
        for li in range(cfg.num_blocks):
            g = eng.buffer("blocks.%d.gate_idx" % li, torch.int32).cpu().numpy()
            touched.append(len(np.unique(g[g >= 0])))
        wsz = {"f32": 4, "bf16": 2, "fp8": 1}[cfg.weight_dtype]   # expert weights in weight_dtype, biases / scales / rows fp32
        extra = (F + D) * 4 * (2 if cfg.weight_dtype == "fp8" else 1)
        exp_bytes = [t_ * (2 * D * F * wsz + extra) + live * 2 * D * 4 for t_ in touched]
        peak_bw = 8000.0                                                        # GB/s (MI355X_MICROARCH.md)
        # dense MFMA TFLOP/s (MI355X_MICROARCH.md): fp32 / bf16 by the instruction's own rate; fp8 against the CHIP's fp8 peak
        # (5 PF, reached only by v_mfma_scale_f32_32x32x64_f8f6f4) when the engine computes in fp8, against the bf16 peak
        # when fp8 is weight storage only (W8A16: the MFMAs are bf16)
        peak_tf = {"f32": 157.3, "bf16": 2500.0, "fp8": 5000.0 if cfg.fp8_activations else 2500.0}[cfg.weight_dtype]
        peak_note = {"f32": "v_mfma_f32_16x16x4_f32 (fp32 in, exact)", "bf16": "bf16 MFMA dense",
                     "fp8": ("chip fp8 dense peak (fp8 arithmetic in the grouped expert FFN)" if cfg.fp8_activations
                             else "bf16 MFMA dense (fp8 is weight storage only: W8A16)")}[cfg.weight_dtype]
        fam, li = {}, 0
        tot_bytes = tot_flops = 0.0
        for st, t_ms in zip(info, ms):
            scale = (live / float(S)) if st["per_row"] else 1.0
            if st["alg_bytes"] < 0:                # the grouped expert FFN of layer li
                by, fl = exp_bytes[li], st["flops"] * scale
                li += 1
            else:
                by, fl = st["alg_bytes"] * scale, st["flops"] * scale
            f_ = fam.setdefault(st["kernel"], [0.0, 0.0, 0.0, 0])
            f_[0] += t_ms; f_[1] += by; f_[2] += fl; f_[3] += st["launches"]
            tot_bytes += by; tot_flops += fl
        t_all = float(sum(v[0] for v in fam.values()))

        def roof(t_ms, by, fl, launches, kernel):
            """bound = whichever roofline leaves less headroom for this kernel at this shape"""
            t_s = t_ms * 1e-3
            f_hbm, f_mfma = by / t_s / (peak_bw * 1e9), fl / t_s / (peak_tf * 1e12)
            d = {"kernel": kernel, "time_share": round(t_ms / t_all, 4), "launches_per_forward": launches,
                 "avg_launch_us": round(t_ms * 1e3 / max(launches, 1), 2),
                 "alg_bytes_per_launch": int(by / max(launches, 1)), "flops_per_launch": int(fl / max(launches, 1))}
            if f_hbm >= f_mfma:
                d.update(bound="hbm", achieved=round(by / t_s / 1e9, 1), peak=peak_bw, unit="GB/s", frac=round(f_hbm, 4))
            else:
                d.update(bound="mfma", achieved=round(fl / t_s / 1e12, 2), peak=peak_tf, unit="TFLOP/s", frac=round(f_mfma, 4))
            return d

        dom = max(fam, key=lambda k: fam[k][0])
        roofline = roof(fam[dom][0], fam[dom][1], fam[dom][2], fam[dom][3], dom)
        roofline["families"] = {k: {"time_share": round(v[0] / t_all, 4), "launches": v[3],
                                    "hbm_frac": round(v[1] / (v[0] * 1e-3) / (peak_bw * 1e9), 4)}
                                for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:8]}
        if args.profile_stages:
            for st, m_ in zip(info, ms):
                n_ = st["name"]
                if not (n_.startswith("blocks.") or n_.startswith("embed.blocks.")) or ".0." in n_ or ".9." in n_:
                    print("%-40s %-34s %8.2f us" % (n_, st["kernel"], m_ * 1e3), file=sys.stderr)
            print("sum of stages %.3f ms, %d kernels" % (ms.sum(), eng.num_kernels()), file=sys.stderr)
        # the grouped expert FFN IN SITU: whole forwards are enqueued stage by stage on the engine stream (the GPU stays
        # the bottleneck: ~3.5 us host cost per launch vs ~8 us per kernel) with HIP events only around each layer's
        # expert launch, so the kernel sees the cache state of a real forward -- repeated in isolation its ~100 MB of
        # weights would sit in the 256 MB Infinity Cache and read 25 % faster
        idx = [i for i, st in enumerate(info) if st["alg_bytes"] < 0]
        st_ = eng.stream
        acc_t = np.zeros(len(idx))
        passes = 20
        for _ in range(passes):
            evs, cur = [], 0
            for i_ in idx:
                eng.run_stages(cur, i_)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st_)
                eng.run_stages(i_, i_ + 1)
                e1.record(st_)
                evs.append((e0, e1))
                cur = i_ + 1
            eng.run_stages(cur, len(names))
            st_.synchronize()
            acc_t += np.array([a.elapsed_time(b_) for a, b_ in evs])
        dur_ms = acc_t / passes
        n_l = sum(info[i_]["launches"] for i_ in idx)
        roofline_expert = roof(float(dur_ms.sum()), float(sum(exp_bytes)), float(sum(info[i_]["flops"] for i_ in idx)) * live / S,
                               n_l, info[idx[0]]["kernel"])
        roofline_expert.update(experts_touched_mean=round(float(np.mean(touched)), 2), experts_touched=touched,
                               rows_per_touched_expert=round(live / max(float(np.mean(touched)), 1.0), 1))
        # HBM bytes per launch from hardware counters: `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over THIS
        # script (bench.py --pmc-safe, same workload), summarised by tools/pmc_summarize.py into profiles/ (counters cannot
        # be read from inside the profiled process); traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch, the gfx950
        # correction of MI355X_MICROARCH.md (HBM).  null when no summary of this exact workload is committed.
        for r_ in (roofline, roofline_expert):
            r_["traffic"] = None
        pmc = next((q_ for q_ in (os.path.join(ROOT, "profiles", r_ + "_pmc_bench.json") for r_ in ("r04", "r03")) if os.path.exists(q_)), "")
        if os.path.exists(pmc):
            try:
                ent = json.load(open(pmc))
                if ent.get("workload") == [cfg.weight_dtype, B, T, cfg.num_blocks, cfg.num_experts]:
                    for r_ in (roofline, roofline_expert):
                        k_ = [v for k, v in ent["kernels"].items() if k.startswith(r_["kernel"].split("<")[0])]
                        if k_:
                            n_l_ = float(sum(v["launches"] for v in k_))      # launch-weighted over the family's template variants
                            r_["traffic"] = int(sum(v["traffic_bytes_per_launch"] * v["launches"] for v in k_) / max(n_l_, 1.0))
                            r_["traffic_source"] = "profiles/%s (rocprofv3 --pmc over bench.py --pmc-safe, timed-workload launches only)" % os.path.basename(pmc)
            except Exception:
                pass
        # The same kernels' durations as rocprofv3 --kernel-trace --stats saw them over this command (the committed summary of the
        # workload: profiles/r03_kernel_stats*.csv).  `avg_launch_us` above is a HIP-event pair around each stage on the engine
        # stream, one forward alone: it contains the dispatch of the launch (~2-3 us); the profiler's figure is the kernel's own
        # begin -> end (at the bench's context count, i.e. under contention).  Both are reported; `frac` uses the larger, in-situ one.
        kt_csv = {("f32", 1, 206, 18, 32): "kernel_stats.csv", ("bf16", 16, 500, 18, 32): "kernel_stats_cfg3.csv",
                  ("fp8", 64, 500, 18, 64): "kernel_stats_cfg5share.csv"}.get((cfg.weight_dtype, B, T, cfg.num_blocks, cfg.num_experts))
        if kt_csv:      # the newest committed summary of this workload
            kt_csv = next((r_ + "_" + kt_csv for r_ in ("r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", r_ + "_" + kt_csv))), None)
        if kt_csv and os.path.exists(os.path.join(ROOT, "profiles", kt_csv)):
            try:
                import csv
                rows_ = list(csv.DictReader(open(os.path.join(ROOT, "profiles", kt_csv))))
                def grp_of(name_):
                    # gemm_bf16w_tiled_kernel<TBM, TBN, TBK, GLU, CONV, LN, GRP, W8, A16IN>: GRP = 1 / 2 are the two GROUPED (per-expert)
                    # GEMMs of the expert FFN, 0 the dense ones -- one template, two families
                    a_ = name_[name_.index("<") + 1:name_.rindex(">")].split(",") if "<" in name_ and ">" in name_ else []
                    return int(a_[6]) if "gemm_bf16w_tiled_kernel<" in name_ and len(a_) > 6 and a_[6].strip().isdigit() else 0
                for r_ in (roofline, roofline_expert):
                    base_ = r_["kernel"].split("<")[0]
                    sel_ = [x for x in rows_ if base_ + "<" in x["Name"] or x["Name"].endswith(base_) or (base_ + "(") in x["Name"]]
                    if base_ == "gemm_bf16w_tiled_kernel":       # grouped instantiations are the expert launches, the rest the dense family
                        want_grp_ = "<grouped" in r_["kernel"]
                        sel_ = [x for x in sel_ if (grp_of(x["Name"]) > 0) == want_grp_]
                    if sel_:
                        calls_ = sum(float(x["Calls"]) for x in sel_)
                        r_["rocprof_avg_launch_us"] = round(sum(float(x["TotalDurationNs"]) for x in sel_) / calls_ / 1e3, 2)
                        r_["rocprof_source"] = "profiles/%s (rocprofv3 --kernel-trace --stats over this command)" % kt_csv
            except Exception:
                pass
        forward = {"alg_bytes": int(tot_bytes), "flops": int(tot_flops),
                   "latency_ms": {"p50": round(float(np.median(lat)), 4), "p99": round(float(lat[int(0.99 * (len(lat) - 1))]), 4),
                                  "min": round(float(lat[0]), 4), "n": int(len(lat)), "timer": "hipEvent pair per forward, engine stream"},
                   "hbm_frac_one_stream": round(tot_bytes / (float(np.median(lat)) * 1e-3) / (peak_bw * 1e9), 4),
                   "hbm_frac_at_value": round(tot_bytes * (value / frames_per_step) / (peak_bw * 1e9), 4),
                   "mfma_frac_at_value": round(tot_flops * (value / frames_per_step) / (peak_tf * 1e12), 4),
                   "mfma_peak_tflops": peak_tf, "mfma_peak_is": peak_note, "live_rows": live, "padded_rows": S}

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
        if args.varlen or B != 1 or T != 206 or cfg.num_blocks != 18 or cfg.num_experts != 32:
            metric = "encoder frames/sec, %dLx%de Conformer-MoE, batch=%d %s" % (
                cfg.num_blocks, cfg.num_experts, B, ("var-len %s frames" % args.varlen) if args.varlen else "%d-frame utterances" % T)
        # which BASELINE.json config the run IS (not which dtype it resembles)
        full = cfg.num_blocks == 18
        if full and cfg.weight_dtype == "f32" and cfg.num_experts == 32 and B == 1 and T == 206 and not args.varlen:
            which = "BASELINE.json configs[1]"
        elif full and cfg.weight_dtype == "bf16" and cfg.num_experts == 32 and B == 16 and args.varlen == "50-500":
            which = "BASELINE.json configs[2]"
        elif full and cfg.weight_dtype == "fp8" and cfg.num_experts == 64 and args.varlen == "50-500":
            which = "one GPU's share of BASELINE.json configs[4]: all 64 experts local, no expert parallelism"
        else:
            which = "not a BASELINE.json config"
        arith = {"f32": "fp32 (v_mfma_f32_16x16x4_f32)",
                 "bf16": "bf16 weights / bf16 MFMA / fp32 accumulate + activations",
                 "fp8": "%s expert weights + bf16 dense weights / fp32 accumulate + activations" % cfg.fp8_label()}[cfg.weight_dtype]
        cpu_model = ""
        try:
            cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        if cpu is not None:
            cpu["cpu_model"] = cpu_model
        out = {"metric": metric,
               "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": cfg.weight_dtype, "data": "synthetic",
               "config": {"workload": "%d-layer %d-expert %s, batch=%dx%d frames per GPU%s, all experts local (%s)" % (
                              cfg.num_blocks, cfg.num_experts, arith, B, T,
                              (" (lengths U[%s], %d real frames)" % (args.varlen, n_frames)) if args.varlen else "", which),
                          "layers": cfg.num_blocks, "experts": cfg.num_experts, "frames": T, "batch_per_gpu": B,
                          "parallelism": "replicas x%d" % world, "streams_per_gpu": len(ctxs),
                          "value_is": "%d concurrent batch-%d requests per GPU (execution contexts sharing one copy of the "
                                      "weights); one request alone: latency_ms_one_stream" % (len(ctxs), B),
                          "latency_ms_one_stream": round(latency_ms, 4),
                          "frames_per_s_one_stream": round(n_frames / (latency_ms * 1e-3), 1), "hip_graph": use_graph,
                          "h_scale_min_max": fp8_h_scales, "kernels_per_forward": eng.num_kernels(), "fold_pos_proj": bool(args.fold_pos),
                          "routing": args.routing, "route_mode": ["staged", "fused", "split"][route],
                          "packed_rows": bool(B > 1 and eng.packed_rows()), "fork_embed": args.fork_embed,
                          "timed_region": "median of %d repeats of %d forwards" % (len(rep_dt), args.steps),
                          "ms_per_step_repeats": [round(d_ / args.steps * 1e3, 4) for d_ in rep_dt],
                          "value_200_steps": (round(world * n_frames / (float(np.median(long_dt)) / 200), 1) if long_dt else None),
                          "build": build,
                          "profiler_downgraded": downgraded,
                          "h_scale_calibrated_on": ("the benchmark batch itself (untimed set-up)" if args.fp8_activations else None)},
               "roofline": roofline, "roofline_expert": roofline_expert, "forward": forward, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1 and not args.no_ep_probe and weights is not None:
        # After the headline line: a short expert-parallel run over the same process group (configs[3]-shaped: bf16, 2
        # ragged utterances per GPU, experts sharded E / N), so that a multi-GPU run also exercises the all-to-all path.
        # The outcome -- status ok / failed / timeout, the measurement or the exception -- goes to stderr and to
        # gpurun_out/ep_probe_nN.json.  The headline line is already out; the exit code stays 0 unless --ep-probe-strict.
        import threading
        probe_path = os.path.join(ROOT, "gpurun_out", "ep_probe_n%d.json" % world)

        def record(status, payload):
            payload = dict(payload, status=status, n_gpus=world, rank=rank)
            print("ep_probe status=%s rank=%d %s" % (status, rank, json.dumps(payload)), file=sys.stderr, flush=True)
            if rank == 0 or status != "ok":
                try:
                    os.makedirs(os.path.dirname(probe_path), exist_ok=True)
                    with open(probe_path if rank == 0 else probe_path.replace(".json", "_rank%d.json" % rank), "w") as f:
                        json.dump(payload, f)
                except OSError:
                    pass

        def on_timeout():                # a hung collective cannot be unwound: say so, then leave
            record("timeout", {"what": "expert-parallel probe did not finish within 150 s (hung collective or device)"})
            sys.stdout.flush()
            os._exit(3 if args.ep_probe_strict else 0)

        timer = threading.Timer(150.0, on_timeout)
        timer.daemon = True
        timer.start()
        rc = 0
        try:
            if args.ep_probe_inject_failure:
                raise RuntimeError("injected failure (--ep-probe-inject-failure)")
            if args.experts % world:
                record("skipped", {"what": "%d experts do not divide over %d ranks" % (args.experts, world)})
            else:
                r = run_ep(args, rank, world, dev, dist, weights, 10, 2, "bf16", 2, "50-500", 0, balanced=True)
                if rank == 0:
                    r["what"] = "expert-parallel probe after the replica benchmark: %dL/%de bf16, %d experts/GPU, 2 utterances U[50,500] per GPU" % (
                        args.layers, args.experts, args.experts // world)
                    record("ok", r)
        except BaseException as ex:      # noqa: BLE001 -- recorded, not swallowed: status "failed" with the exception
            record("failed", {"what": "expert-parallel probe raised", "exception": repr(ex)})
            rc = 4 if args.ep_probe_strict else 0
        timer.cancel()
        sys.stdout.flush()
        sys.stderr.flush()
        if rc:
            os._exit(rc)                 # a failed collective may have left the group unusable: do not wait on it
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
