"""Expert parallelism for the MoE feed-forward: the 32 (or 64) experts of every layer are sharded contiguously over the
ranks of one node, rank r owning global experts [r*E_loc, (r+1)*E_loc); each rank keeps its own utterances (data
parallel) and tokens travel to the rank that owns their expert and back.

Semantics = the reference's training-time FastMoE path (inference in the reference is single-GPU):
  moe_prepare_forward  trainer_3m_fix/fmoe/functions.py:13-52   local_expert_count over GLOBAL expert ids, count exchange,
                                                                 fwd_expert_count = global_expert_count.view(world, E_loc).sum(0)
  MOEScatter.forward   fmoe/functions.py:63-86                  local_scatter (sort rows by global expert) + global_scatter
  MOEGather.forward    fmoe/functions.py:175-199                global_gather + local_gather
  expert ownership     model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py:251-254,272
Wire order: the send buffer is sorted by global expert id, so the slice for rank j is contiguous; what rank j receives
from rank i is already sorted by its local expert id.  Rows are processed by the grouped expert FFN, whose result for a
row does not depend on the other rows in the launch, so expert-parallel output equals single-GPU output bit for bit.

Transport: torch.distributed all_to_all_single -- backend "nccl" is RCCL over xGMI on the GPU box (messages here are
10s-100s of KB: latency-bound, 2 row exchanges + 1 count exchange per layer); with the "gloo" backend (CPU tests, or
several ranks sharing one GPU) device tensors are staged through host memory.  Compute is a pluggable backend: the
product backend is ``HipBackend`` (libm3asr_hip.so through m3asr.ops); tests may plug the CPU oracle.
"""
import torch

from . import _lib
import torch.distributed as dist


def _all_to_all(send, in_splits, out_splits, group=None):
    """all_to_all_single with row splits; stages through the host when the backend cannot move device memory."""
    out_rows = int(sum(out_splits))
    recv = torch.empty((out_rows,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        recv.copy_(send[:out_rows])
        return recv
    if send.is_cuda and dist.get_backend(group) == "gloo":
        s_cpu, r_cpu = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r_cpu, s_cpu, list(out_splits), list(in_splits), group=group)
        recv.copy_(r_cpu)
    else:
        dist.all_to_all_single(recv, send.contiguous(), list(out_splits), list(in_splits), group=group)
    return recv


class HipBackend:
    """Compute steps of one expert-parallel MoE layer on the MI355X (C ABI through m3asr.ops)."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def index(self, gate_idx, n_expert_total):
        return self.ops.moe_scatter_mapping(gate_idx, n_expert_total)        # mapping, acc_histogram, pos

    def scatter(self, x, mapping, n_rows):
        return self.ops.moe_local_scatter(x, mapping, n_rows)

    def expert_ffn(self, rows, gate_local, w):
        return self.ops.moe_expert_ffn(rows, gate_local, w["w1"], w["b1"], w["w2"], w["b2"],
                                       w1_scale=w.get("s1"), w2_scale=w.get("s2"))

    def combine(self, rows_sorted, mapping, gate_value, resid, alpha, ln, out=None):
        return self.ops.moe_combine(rows_sorted, mapping, gate_value, resid, alpha, ln, out=out)


def ep_moe_layer(x, gate_idx, gate_value, weights, e_loc, resid=None, alpha=1.0, ln=None, out=None, backend=None,
                 group=None):
    """One expert-parallel MoE feed-forward on this rank's tokens.

    x (S,D) f32: LayerNorm'd MoE input; gate_idx (S,) i32 GLOBAL expert id or -1; gate_value (S,) f32 or None;
    weights {"w1","b1","w2","b2"}: THIS rank's experts [E_loc,...]; returns LN(resid + alpha*gate*expert(x)) (S,D).
    """
    be = backend if backend is not None else HipBackend()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_total = e_loc * world
    S, D = x.shape
    mapping, acc, _ = be.index(gate_idx, n_total)
    local_count = (acc[1:] - acc[:-1]).to(torch.int64)                          # [world * E_loc], by global expert id
    n_valid = int(acc[n_total])                                                  # host sync (FastMoE does the same)
    send = be.scatter(x, mapping, max(n_valid, 1))[:n_valid]
    # --- count exchange: global_count[j][i] = rows arriving from rank j for my local expert i
    lc = local_count.view(world, e_loc)
    if world > 1:
        gc = torch.empty_like(lc)
        if lc.is_cuda and dist.get_backend(group) == "gloo":
            g_cpu = torch.empty(lc.shape, dtype=lc.dtype)
            dist.all_to_all_single(g_cpu, lc.cpu(), group=group)
            gc.copy_(g_cpu)
        else:
            dist.all_to_all_single(gc, lc.contiguous(), group=group)
    else:
        gc = lc
    in_splits = lc.sum(1).tolist()                                               # rows I send to each rank
    gc_host = gc.cpu()
    out_splits = gc_host.sum(1).tolist()                                         # rows I receive from each rank
    # --- global_scatter
    recv = _all_to_all(send, in_splits, out_splits, group)
    # local expert id of every received row: per source rank the rows are sorted by my local expert
    recv_gate = torch.repeat_interleave(torch.arange(e_loc, dtype=torch.int32).repeat(world),
                                        gc_host.reshape(-1)).to(x.device)
    # --- this rank's experts on everything it received (grouped FFN; row results are position independent)
    if recv.shape[0] > 0:
        y_recv = be.expert_ffn(recv, recv_gate, weights)
    else:
        y_recv = recv
    # --- global_gather (splits swapped), then local_gather + gate + residual + LayerNorm
    back = _all_to_all(y_recv, out_splits, in_splits, group)
    if back.shape[0] == 0:
        back = torch.zeros(1, D, dtype=x.dtype, device=x.device)
    return be.combine(back, mapping, gate_value, resid, alpha, ln, out)


class ExpertParallelEncoder:
    """Drives a staged native engine (m3asr.engine.Engine built with ep_world_size > 1) across ranks: every stage runs
    on the engine's stream; the ``blocks.N.moe_local.*`` stages are replaced by ``ep_moe_layer``."""

    def __init__(self, engine, group=None):
        self.eng, self.group = engine, group
        # needs the staged route path (router GEMM materialises xn; gate / index / expert / combine are separate stages)
        cfg = engine.cfg
        self.e_loc = cfg.num_experts
        self.backend = HipBackend()
        self.layers = []
        for i in range(cfg.num_blocks):
            p = "blocks.%d." % i
            w = engine.weights
            E, D, F = cfg.num_experts, cfg.attention_dim, cfg.hidden_units
            # the plan keeps w_2 slice-major [E, F/S, D, S] for the fused engine; the C-ABI op takes the reference
            # layout [E, D, F] -> undo once at set-up
            w2 = w[p + "feed_forward.experts.w_2.weight_sliced"].permute(0, 2, 1, 3).reshape(E, D, F).contiguous()
            wd = {"w1": w[p + "feed_forward.experts.w_1.weight"], "b1": w[p + "feed_forward.experts.w_1.bias"],
                  "w2": w2, "b2": w[p + "feed_forward.experts.w_2.bias"]}
            if (p + "feed_forward.experts.w_1.scale") in w:        # fp8 experts: per-row scales travel with the weights
                wd["s1"], wd["s2"] = w[p + "feed_forward.experts.w_1.scale"], w[p + "feed_forward.experts.w_2.scale"]
            self.layers.append({"w": wd,
                                "ln": (w[p + "norm_final.weight"], w[p + "norm_final.bias"], 1e-12)})

    def forward(self, feat, feat_len):
        eng, cfg = self.eng, self.eng.cfg
        logits = eng.bind(feat, feat_len)
        names = eng.stage_names()
        if "blocks.0.moe_router" not in names or "router_e_all" in names:   # fused / split route engines never write xn
            raise RuntimeError("ExpertParallelEncoder needs an engine built with fuse_route=False (or ep_world_size > 1)")
        try:
            eng.buffer("xb")
            raise RuntimeError("ExpertParallelEncoder needs an engine built with bf16_activations=False "
                               "(the driver replaces the combine stage, which maintains the bf16 copy of x)")
        except _lib.M3Error:
            pass
        try:
            eng.buffer("row0")
            raise RuntimeError("ExpertParallelEncoder needs an engine built with packed_rows=False "
                               "(the exchange is written for the padded (B, T') row layout)")
        except _lib.M3Error:
            pass
        S, D = eng.buffer("x").numel() // cfg.attention_dim, cfg.attention_dim
        cur = 0
        with torch.cuda.stream(eng.stream):
            for i in range(cfg.num_blocks):
                # world == 1 engines fuse gate + index ("moe_gate_index", runs before): then only expert/combine are replaced
                key = "blocks.%d.moe_local.index" % i
                first = names.index(key) if key in names else names.index("blocks.%d.moe_local.expert" % i)
                last = names.index("blocks.%d.moe_local.combine" % i)
                eng.run_stages(cur, first)
                x = eng.buffer("x").view(S, D)
                xn = eng.buffer("xn").view(S, D)
                gidx = eng.buffer("blocks.%d.gate_idx" % i, torch.int32)
                gval = None if cfg.keep_expert_output else eng.buffer("blocks.%d.gate_value" % i)
                L = self.layers[i]
                ep_moe_layer(xn, gidx, gval, L["w"], self.e_loc, resid=x, alpha=0.5, ln=L["ln"], out=x,
                             backend=self.backend, group=self.group)
                cur = last + 1
            eng.run_stages(cur, len(names))
        eng.stream.synchronize()
        return logits
