"""GPU: the expert-parallel driver on the HIP backend.
 * world 1: ExpertParallelEncoder over an engine with the expert-parallel stage list (ep_stages: index on global ids, wire,
   exchange as a device copy, receive-side grouped FFN, combine) == the all-experts-local engine, bit for bit.
 * world 2 on ONE GPU: two processes share cuda:0, experts sharded 2+2, gloo transport staged through the host
   (RCCL refuses two ranks on one device); each rank's logits must match the CPU oracle with all experts local."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from m3asr.config import EncoderConfig
from m3asr.engine import Engine
from m3asr.ep import ExpertParallelEncoder
from m3asr.weights import make_weights
from oracle.encoder_ref import encoder_forward, sub_len


def test_ep_world1_equals_fused_engine():
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg, seed=4)
    feat = torch.rand(2, 120, cfg.input_dim, generator=torch.Generator().manual_seed(2)).cuda()
    fl = torch.tensor([[120, 77]], dtype=torch.int32).cuda()
    # the EP driver replaces the moe_local.* stages of the staged (unfused-route) engine: same kernels, same order
    staged = Engine.from_state_dict(cfg, w, fuse_route=False, packed_rows=False)(feat, fl).clone()
    ep = ExpertParallelEncoder(Engine.from_state_dict(cfg, w, ep_stages=True, packed_rows=False))
    assert torch.equal(ep.forward(feat, fl), staged)
    # fuse_route engines do router + top-1 + index in one launch (different summation order in the router)
    fused = Engine.from_state_dict(cfg, w, fuse_route=True)(feat, fl)
    assert torch.allclose(fused, staged, rtol=1e-4, atol=1e-4)


def test_ep_world1_on_packed_rows_equals_engine():
    """The device-side exchange works on the packed row layout of ragged batches (rows past the live count carry
    gate_idx -1 and never reach the wire): with one rank it reproduces the packed engine bit for bit, fp32."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1)
    w = make_weights(cfg, seed=4)
    feat = torch.rand(4, 206, cfg.input_dim, generator=torch.Generator().manual_seed(2)).cuda()
    fl = torch.tensor([[206, 77, 150, 33]], dtype=torch.int32).cuda()
    eng = Engine.from_state_dict(cfg, w, fuse_route=False)
    want = eng(feat, fl).clone()
    assert eng.packed_rows()
    ep = ExpertParallelEncoder(Engine.from_state_dict(cfg, w, ep_stages=True))
    got = ep.forward(feat, fl)
    assert ep.eng.packed_rows()
    assert torch.equal(got, want)
    # later forwards of the binding replay one graph (stages + exchanges): nothing is allocated or synchronised per layer
    assert torch.equal(ep.forward(feat, fl), want)
    assert ep.graph_state == "engine graph"
    assert torch.equal(ep.forward(feat, fl), want)


def test_bench_expert_parallel_rehearsal_two_ranks_one_gpu(tmp_path):
    """bench.py --ep with 2 ranks sharing this GPU (gloo transport staged through the host; RCCL refuses two ranks on one
    device): the expert-parallel benchmark path end to end -- sharded plan, router broadcast, fixed-shape exchange, JSON
    line -- on a 2-layer model; and the replica mode's expert-parallel probe after the headline line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def base_cmd():      # a fresh rendezvous port per invocation (a port just released may still be in TIME_WAIT)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--layers", "2",
                "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run(base_cmd() + ["--ep", "--weight-dtype", "bf16", "--batch", "2", "--varlen", "50-500"], capture_output=True,
                       text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["experts_per_gpu"] == 16
    # (the gloo rehearsal stages every exchange through the host: 2 synchronisations per layer; RCCL: none)
    assert line["config"]["wire"]["host_syncs_per_forward"] == 4 and line["config"]["wire"]["collectives_per_forward"] == 4
    assert line["config"]["forward_graph"].startswith("eager: the gloo transport")
    r = subprocess.run(base_cmd() + ["--streams", "2"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "replicas x2"
    assert "ep_probe status=ok" in r.stderr, r.stderr[-2000:]
    probe = json.load(open(os.path.join(root, "gpurun_out", "ep_probe_n2.json")))
    assert probe["status"] == "ok" and probe["value"] > 0
    # a failing probe is visible: status "failed" + the exception on stderr and in the file, exit code non-zero when strict
    r = subprocess.run(base_cmd() + ["--streams", "2", "--ep-probe-inject-failure", "--ep-probe-strict"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode != 0
    assert [l for l in r.stdout.splitlines() if l.startswith("{")], "the headline line must still be printed; stderr: " + r.stderr[-3000:]
    assert "ep_probe status=failed" in r.stderr and "injected failure" in r.stderr
    assert json.load(open(os.path.join(root, "gpurun_out", "ep_probe_n2.json")))["status"] == "failed"


def test_ep_world1_bf16_equals_engine():
    """16-bit mode (BASELINE.json configs[3]: bf16 expert parallel): the EP driver feeds bf16 expert weights to
    m3_moe_expert_ffn_bf16; with one rank it must reproduce the bf16 engine bit for bit (same kernels, same order),
    for the slab form (S < 1024) and for the two grouped tiled GEMMs (S >= 1024)."""
    cfg = EncoderConfig(num_blocks=2, embed_blocks=1, weight_dtype="bf16")
    w = make_weights(cfg, seed=4)
    for B, T in ((2, 120), (16, 400)):                      # 58 rows (slab form); 1584 rows, 1200 routed (tiled form on both sides)
        feat = torch.rand(B, T, cfg.input_dim, generator=torch.Generator().manual_seed(2)).cuda()
        fl = torch.tensor([[T - 13 * i for i in range(B)]], dtype=torch.int32).cuda()
        eng = Engine.from_state_dict(cfg, w, bf16_activations=False, packed_rows=False)
        want = eng(feat, fl).clone()
        ep = ExpertParallelEncoder(Engine.from_state_dict(cfg, w, bf16_activations=False, packed_rows=False, ep_stages=True))
        assert torch.equal(ep.forward(feat, fl), want)
        if B == 16:      # the default engine keeps bf16 activation operands at this size (a bf16 copy of x that every kernel
            # writing x maintains): the driver's combine (m3_moe_combine_bf16) maintains it too -- bit for bit again
            eng16 = Engine.from_state_dict(cfg, w, packed_rows=False)
            want16 = eng16(feat, fl).clone()
            assert eng16.buffer("xb", torch.bfloat16) is not None
            ep16 = ExpertParallelEncoder(Engine.from_state_dict(cfg, w, packed_rows=False, ep_stages=True))
            assert torch.equal(ep16.forward(feat, fl), want16)
            assert not torch.equal(want16, want)        # (the two modes do differ: bf16 activation operands are in use)


def _worker(rank, world, port, out_dir, wdt, capacity_factor=None):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e_loc = 2
    full = EncoderConfig.tiny(num_experts=e_loc * world)          # all experts of the model
    cfg = EncoderConfig.tiny(num_experts=e_loc, ep_world_size=world, ep_rank=rank, weight_dtype=wdt)
    w = make_weights(full, seed=9)                                # whole-model state_dict; pack_weights slices the experts
    g = torch.Generator().manual_seed(100 + rank)
    T = 90 - 7 * rank
    feat = torch.randn(2, T, cfg.input_dim, generator=g)
    fl = torch.tensor([[T, T - 20]], dtype=torch.int32)
    eng = Engine.from_state_dict(cfg, w, device="cuda:0")          # ep_world_size > 1 -> staged (unfused) route path; B = 2: packed rows
    ep = ExpertParallelEncoder(eng, capacity_factor=capacity_factor)
    out = ep.forward(feat.cuda(), fl.cuda()).cpu()
    np.save(os.path.join(out_dir, "reruns%d.npy" % rank), np.array([ep.reruns, ep._bound[3], ep._full_cap]))
    # reference: all experts local (fp32: the CPU oracle; bf16: the single-rank engine of the same precision, whose row
    # results are position independent)
    if wdt == "f32":
        want = encoder_forward(w, full, feat, fl)
    else:
        full16 = EncoderConfig.tiny(num_experts=e_loc * world, weight_dtype=wdt)
        want = Engine.from_state_dict(full16, w, device="cuda:0", bf16_activations=False)(feat.cuda(), fl.cuda()).cpu()
    valid = torch.arange(out.shape[1]).view(1, -1) < sub_len(fl.view(-1).long()).view(-1, 1)
    err = float((out - want).abs()[valid].max())
    np.save(os.path.join(out_dir, "err%d.npy" % rank), np.array([err, float(want[valid].abs().max())]))
    dist.barrier()
    dist.destroy_process_group()


def test_ep_bounded_wire_repeats_on_overflow(tmp_path):
    """capacity_factor bounds a wire chunk at f x rows / world instead of all rows; a chunk that needs more is reported on the
    device and the forward is repeated with a capacity that fits -- no row is dropped (the fixed-shape wire moved
    world x the bytes needed).  f = 0.3 at world 2 cannot hold a balanced routing: at least one repeat, same logits as ever."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), "f32", 0.3), nprocs=2, join=True)
    for r in range(2):
        err, scale = np.load(os.path.join(str(tmp_path), "err%d.npy" % r))
        reruns, cap, full = np.load(os.path.join(str(tmp_path), "reruns%d.npy" % r))
        assert err <= 2e-4 + 1e-3 * scale, (r, err)
        assert reruns >= 1 and cap <= full, (reruns, cap, full)


@pytest.mark.parametrize("world,wdt", [(2, "f32"), (4, "f32"), (4, "bf16")])
def test_ep_ranks_on_one_gpu(tmp_path, world, wdt):
    """2 / 4 processes share cuda:0, 2 experts each (4 / 8 in total), gloo transport staged through the host (RCCL refuses
    several ranks on one device).  fp32: each rank's logits match the CPU oracle with all experts local; bf16
    (BASELINE.json configs[3]): equal to the single-rank bf16 engine up to fp32 rounding of the gate value."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), wdt), nprocs=world, join=True)
    for r in range(world):
        err, scale = np.load(os.path.join(str(tmp_path), "err%d.npy" % r))
        if wdt == "f32":
            assert err <= 2e-4 + 1e-3 * scale, (r, err)
        else:      # same expert arithmetic; the multi-rank engine takes top-1 and index in two kernels, the single-rank one in a
            assert err <= 1e-5 * scale + 1e-6, (r, err)   # fused kernel (gate value summed in another order: last-ulp noise)


@pytest.mark.parametrize("mode", ["bf16", "fp8_arithmetic"])
def test_ep_world8_real_dims_one_process(mode):
    """BASELINE.json configs[3] / configs[4] at REAL dimensions and world 8, on one GPU: 8 rank engines in one process
    (m3asr.ep.InProcessRanks: every native stage and wire format is the real one, the all-to-all is a device copy between
    the ranks' wire buffers -- a GPU box admits neither 8 processes on its card nor RCCL ranks sharing a device).
      bf16:           32 experts, 4 per rank, 2 ragged utterances U[50,500] per rank   (configs[3], 4 of its 18 layers)
      fp8 arithmetic: 64 experts, 8 per rank, 64 ragged utterances per rank, calibrated H scales (configs[4], 4 layers);
                      the receive side must run the fused fp8 kernel (e4m3 x e4m3 MFMA), not the weight-only form
    Reference = ONE engine of the same precision with all experts local, on batches that take the SAME kernels as a rank
    does (the kernels of a stage are chosen by row count -- bf16 activation copies, LDS-DMA GEMM, router GEMM, bf16 attention
    core from some row count on -- and another arithmetic moves near-ties of the first router already):
      bf16: each rank's own two utterances, one rank after the other (248 rows on both sides);
      fp8:  the union batch of 512 utterances (63 488 rows) against ranks of 64 (7 936 rows: configs[4]'s batch per GPU) --
            both above every threshold, both on the fused fp8 expert kernel.
    What differs is then only the form of the grouped expert FFN (rows per launch) and the order of rows inside an expert:
    rounding-level differences in H."""
    from m3asr.ep import InProcessRanks
    world = 8
    fp8 = mode == "fp8_arithmetic"
    E, per_rank = (64, 64) if fp8 else (32, 2)
    wdt = "fp8" if fp8 else "bf16"
    full = EncoderConfig(num_blocks=4, num_experts=E, weight_dtype=wdt, fp8_activations=fp8)
    w = make_weights(full, seed=21)
    rng = np.random.default_rng(77)
    B = world * per_rank
    lengths = rng.integers(50, 501, B)
    lengths[::per_rank] = 500                                    # every rank's padded length is 500 (one shape for all)
    feat = torch.from_numpy(rng.random((B, 500, full.input_dim), dtype=np.float32))
    fl = torch.from_numpy(lengths.astype(np.int32))
    if fp8:
        from m3asr.calibrate import calibrate_h_scales
        calibrate_h_scales(full, w, [(feat[:16], fl[:16])])
    ref_eng = Engine.from_state_dict(full, w)
    chunks = [(0, B)] if fp8 else [(r * per_rank, (r + 1) * per_rank) for r in range(world)]
    want, ref_gate = [], []
    for lo, hi in chunks:
        want.append(ref_eng(feat[lo:hi].cuda().contiguous(), fl[lo:hi].view(1, -1).cuda().contiguous()).cpu())
        ref_gate.append(torch.stack([ref_eng.rows_padded("blocks.%d.gate_idx" % i, torch.int32, fill=-1).cpu().view(hi - lo, -1)
                                     for i in range(full.num_blocks)]))
    want, ref_gate = torch.cat(want), torch.cat(ref_gate, dim=1)
    kern = {s_["name"]: s_["kernel"] for s_ in ref_eng.stage_info()}
    if fp8:
        assert kern["blocks.0.moe_local.expert"] == "expert_ffn_fused_fp8_kernel"
    del ref_eng
    torch.cuda.empty_cache()
    engines = []
    for r in range(world):
        cfg = EncoderConfig(num_blocks=4, num_experts=E // world, ep_world_size=world, ep_rank=r, weight_dtype=wdt, fp8_activations=fp8)
        engines.append(Engine.from_state_dict(cfg, w))
    feats = [feat[r * per_rank:(r + 1) * per_rank].cuda().contiguous() for r in range(world)]
    lens = [fl[r * per_rank:(r + 1) * per_rank].view(1, -1).cuda().contiguous() for r in range(world)]
    ranks = InProcessRanks(engines)
    outs = [o.clone() for o in ranks.forward(feats, lens)]
    # eight engines in flight at once: the same forward again must give the same bits (the symptom by which the router kernel's
    # dependence on its CU's other residents was found -- DESIGN.md 10.8)
    again = ranks.forward(feats, lens)
    assert all(torch.equal(a, b) for a, b in zip(outs, again)), "two runs of the same expert-parallel forward differ"
    # bounded wire: chunks of 2 x rows / world instead of all rows.  These synthetic routers are far from balanced
    # (raw random router weights send most frames of an utterance to a few experts), so the first try may overflow: the device
    # reports the rows the fullest chunk needed, the forward is repeated with that capacity (what ExpertParallelEncoder.forward
    # does) and must then be clean; in fp8 arithmetic (a row's result does not depend on how rows are grouped) with the same
    # bits as the full wire.  (bf16: the receive side's row count picks another form of the grouped FFN: no bit comparison.)
    full_cap = max(int(f.shape[0]) * e.output_shape(int(f.shape[0]), int(f.shape[1]))[1] for e, f in zip(engines, feats))
    cap = -(-2 * full_cap // world // 16) * 16
    bounded = [o.clone() for o in ranks.forward(feats, lens, capacity=cap)]
    if ranks.overflow > cap:
        cap = min(full_cap, -(-ranks.overflow // 16) * 16)
        bounded = [o.clone() for o in ranks.forward(feats, lens, capacity=cap)]
    assert ranks.overflow == 0
    print("bounded wire: %d rows per chunk (full wire %d)" % (cap, full_cap))
    if fp8:
        assert all(torch.equal(a, b) for a, b in zip(outs, bounded))
    ranks.forward(feats, lens, capacity=16)
    assert ranks.overflow > 16
    ranks.forward(feats, lens)                 # (back to the full wire for the checks below)
    ek = {s_["name"]: s_["kernel"] for s_ in engines[0].stage_info()}["blocks.0.moe_ep.expert"]
    if fp8:
        assert ek == "expert_ffn_fused_fp8_kernel", ek
    got = torch.cat([o.cpu() for o in outs])
    Tp = got.shape[1]
    valid = torch.arange(Tp).view(1, -1) < sub_len(fl.long()).view(-1, 1)
    gate = torch.cat([torch.stack([e.rows_padded("blocks.%d.gate_idx" % i, torch.int32, fill=-1).cpu().view(per_rank, -1)
                                   for i in range(full.num_blocks)]) for e in engines], dim=1)
    agree = float((gate[:, valid] == ref_gate[:, valid]).float().mean())
    # a near-tie of one router may fall the other way (the receive side sums H in another row order): such an utterance takes
    # another expert from there on and is compared by routing agreement only; at most 2 of the 16 / 64 utterances may do so
    flipped = ((gate != ref_gate) & valid.unsqueeze(0)).any(dim=2).any(dim=0)
    same = valid & ~flipped.view(-1, 1)
    err = float((got - want).abs()[same].max()) / float(want.abs()[valid].max())
    print("EP world 8 (%s, %d experts, %d per rank, %d utterances per rank, receive-side kernel %s): max |err| / max |logit| "
          "= %.3e vs the all-experts-local engine on %d of %d utterances, routing agreement %.5f (flips at layer/utterance/frame %s)"
          % (mode, E, E // world, per_rank, ek, err, int((~flipped).sum()), B, agree,
             ((gate != ref_gate) & valid.unsqueeze(0)).nonzero().tolist()[:8]))
    assert bool((got[~valid] == 0).all())
    assert agree >= 0.999, agree
    assert int(flipped.sum()) <= 2, flipped.nonzero().view(-1).tolist()
    assert err <= 5e-3, err          # two 16-bit evaluations that round H in different kernels; 16-bit vs fp32 is held to 2e-2
    if fp8:                          # the fused fp8 kernel's result per row does not depend on how rows are grouped: exact
        assert err == 0.0 and agree == 1.0, (err, agree)
