"""Expert parallelism for the MoE feed-forward: the 32 (or 64) experts of every layer are sharded contiguously over the
ranks of one node, rank r owning global experts [r*E_loc, (r+1)*E_loc); each rank keeps its own utterances (data
parallel) and tokens travel to the rank that owns their expert and back.

Semantics = the reference's training-time FastMoE path (inference in the reference is single-GPU):
  moe_prepare_forward  trainer_3m_fix/fmoe/functions.py:13-52   local_expert_count over GLOBAL expert ids, count exchange,
                                                                 fwd_expert_count = global_expert_count.view(world, E_loc).sum(0)
  MOEScatter.forward   fmoe/functions.py:63-86                  local_scatter (sort rows by global expert) + global_scatter
  MOEGather.forward    fmoe/functions.py:175-199                global_gather + local_gather
  expert ownership     model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py:251-254,272

FastMoE reads the counts back to the host (`.cpu()`) to size every all-to-all-v.  Here NOTHING returns to the host inside
a layer: the exchange has a fixed shape, so a forward is one uninterrupted stream of enqueues.

  wire buffer [world][1 + C][D] f32   chunk j = what this rank sends to rank j (after the all-to-all with EQUAL splits:
                                      what it received from rank j): one header row carrying the E_loc row counts of the
                                      chunk (the count exchange rides in the payload -> two collectives per layer instead
                                      of three) and up to C rows sorted by rank j's local expert id.  C = rows per rank
                                      (max over ranks, agreed once per bound shape): a rank may route everything to one peer.
  per layer   index (global ids) -> ep_send_map (wire row of every token + headers) -> local_scatter into the wire ->
              all_to_all_single -> ep_recv_gate (local expert id of every received wire row) -> grouped expert FFN on the
              received wire (m3_moe_expert_ffn: its own stable index puts the rows in FastMoE's receive order: by local
              expert, then source rank, then wire order) -> all_to_all_single back -> combine (gate, residual, LayerNorm)
              reading each token's result at the wire row it was sent from.

A row's expert FFN result does not depend on the other rows of the launch (fp32 / slab forms: bit for bit), so
expert-parallel output equals the single-GPU output.  Cost of the fixed shape: world x C rows on the wire instead of C
(2-9 MB per exchange at the BASELINE configs, microseconds on xGMI) against two host synchronisations per layer.

Transport: torch.distributed all_to_all_single -- backend "nccl" is RCCL over xGMI on the GPU box; with the "gloo"
backend (CPU tests, or several ranks sharing one GPU) device tensors are staged through host memory.  Compute is a
pluggable backend: the product backend is ``HipBackend`` (libm3asr_hip.so through m3asr.ops); tests may plug the CPU oracle.
"""
import torch

from . import _lib
import torch.distributed as dist


def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _all_to_all_equal(recv, send, group=None):
    """all_to_all_single with equal splits (dim 0 = world chunks); stages through the host when the backend cannot move
    device memory (gloo rehearsal)."""
    if _world(group) == 1:
        recv.copy_(send)
        return recv
    if send.is_cuda and dist.get_backend(group) == "gloo":
        s_cpu, r_cpu = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r_cpu, s_cpu, group=group)
        recv.copy_(r_cpu)
    else:
        dist.all_to_all_single(recv, send, group=group)
    return recv


class HipBackend:
    """Compute steps of one expert-parallel MoE layer on the MI355X (C ABI through m3asr.ops); every step only enqueues."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def index(self, gate_idx, n_expert_total):
        return self.ops.moe_scatter_mapping(gate_idx, n_expert_total)        # mapping, acc_histogram, pos

    def send_map(self, gate_idx, mapping, acc, world, e_loc, cap, map_send, wire):
        return self.ops.ep_send_map(gate_idx, mapping, acc, world, e_loc, cap, map_send, wire)

    def scatter_into(self, x, map_send, wire):
        return self.ops.moe_local_scatter_into(x, map_send, wire.view(-1, wire.shape[-1]))

    def recv_gate(self, wire, world, e_loc, cap, gate_recv):
        return self.ops.ep_recv_gate(wire, world, e_loc, cap, gate_recv)

    def expert_ffn(self, rows, gate_local, w, out, workspace=None):
        return self.ops.moe_expert_ffn(rows, gate_local, w["w1"], w["b1"], w["w2"], w["b2"],
                                       w1_scale=w.get("s1"), w2_scale=w.get("s2"), out=out, workspace=workspace)

    def combine(self, rows, mapping, gate_value, resid, alpha, ln, out=None, out_bf16=None):
        return self.ops.moe_combine(rows, mapping, gate_value, resid, alpha, ln, out=out, out_bf16=out_bf16)


class EpBuffers:
    """Per-shape buffers of the exchange (allocated once, reused by every layer and every forward)."""

    def __init__(self, S, D, world, e_loc, capacity, device, F=None, n_expert=None):
        self.S, self.D, self.world, self.e_loc, self.cap = S, D, world, e_loc, capacity
        rows = world * (capacity + 1)
        self.wire_a = torch.zeros(world, capacity + 1, D, dtype=torch.float32, device=device)
        self.wire_b = torch.zeros(world, capacity + 1, D, dtype=torch.float32, device=device)
        self.map_send = torch.empty(S, dtype=torch.int32, device=device)
        self.gate_recv = torch.empty(rows, dtype=torch.int32, device=device)
        self.workspace = None
        if F is not None and str(device) != "cpu":
            from . import ops
            self.workspace = torch.empty(max(ops.moe_expert_workspace_size(rows, e_loc, D, F), 1), dtype=torch.uint8, device=device)


def agree_capacity(S, device, group=None):
    """Rows per wire chunk: the largest row count of any rank (one all-reduce per bound shape, outside the layer loop)."""
    if _world(group) == 1:
        return int(S)
    on_host = dist.get_backend(group) == "gloo"
    t = torch.tensor([int(S)], dtype=torch.int64, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def ep_moe_layer(x, gate_idx, gate_value, weights, e_loc, resid=None, alpha=1.0, ln=None, out=None, backend=None,
                 group=None, buffers=None, out_bf16=None):
    """One expert-parallel MoE feed-forward on this rank's tokens; only enqueues work (no host read-back).

    x (S,D) f32: LayerNorm'd MoE input; gate_idx (S,) i32 GLOBAL expert id or -1; gate_value (S,) f32 or None;
    weights {"w1","b1","w2","b2"}: THIS rank's experts [E_loc,...]; returns LN(resid + alpha*gate*expert(x)) (S,D).
    buffers: EpBuffers of this shape (created on the fly when None -- that path costs one capacity all-reduce).
    out_bf16 (S,D) bf16: also receives the result (engines with bf16 activation operands keep such a copy of x).
    """
    be = backend if backend is not None else HipBackend()
    world = _world(group)
    S, D = x.shape
    if buffers is None:
        buffers = EpBuffers(S, D, world, e_loc, agree_capacity(S, x.device, group), x.device,
                            F=weights["w1"].shape[1], n_expert=e_loc)
    bf = buffers
    mapping, acc, _ = be.index(gate_idx, e_loc * world)
    # wire row of every token (+ the count headers), payload rows scattered straight into the send wire
    be.send_map(gate_idx, mapping, acc, world, e_loc, bf.cap, bf.map_send, bf.wire_a)
    be.scatter_into(x, bf.map_send, bf.wire_a)
    # global_scatter: chunk j of wire_a -> rank j (equal splits: sizes are shape constants)
    _all_to_all_equal(bf.wire_b, bf.wire_a, group)
    # this rank's experts on everything it received; the result goes back into wire_a at the same wire rows
    be.recv_gate(bf.wire_b, world, e_loc, bf.cap, bf.gate_recv)
    rows = bf.wire_b.view(-1, D)
    be.expert_ffn(rows, bf.gate_recv, weights, out=bf.wire_a.view(-1, D), workspace=bf.workspace)
    # global_gather, then local_gather + gate + residual + LayerNorm: token s reads wire row map_send[s]
    _all_to_all_equal(bf.wire_b, bf.wire_a, group)
    if out_bf16 is not None:
        return be.combine(bf.wire_b.view(-1, D), bf.map_send, gate_value, resid, alpha, ln, out, out_bf16=out_bf16)
    return be.combine(bf.wire_b.view(-1, D), bf.map_send, gate_value, resid, alpha, ln, out)


class ExpertParallelEncoder:
    """Drives a staged native engine (m3asr.engine.Engine built with ep_world_size > 1) across ranks: every stage runs
    on the engine's stream; the ``blocks.N.moe_local.*`` stages are replaced by ``ep_moe_layer``.  Works on padded and
    on packed rows (rows past the live count carry gate_idx -1 and never travel)."""

    def __init__(self, engine, group=None):
        self.eng, self.group = engine, group
        # needs the staged route path (router GEMM materialises xn; gate / index / expert / combine are separate stages)
        cfg = engine.cfg
        self.e_loc = cfg.num_experts
        self.backend = HipBackend()
        self.layers = []
        self._buffers = {}
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

    def bind(self, feat, feat_len):
        """Bind the input buffers, check the engine is usable and agree on the wire capacity (the only collective outside
        the layers; once per shape)."""
        eng, cfg = self.eng, self.eng.cfg
        logits = eng.bind(feat, feat_len)
        names = eng.stage_names()
        if "blocks.0.moe_router" not in names or "router_e_all" in names:   # fused / split route engines never write xn
            raise RuntimeError("ExpertParallelEncoder needs an engine built with fuse_route=False (or ep_world_size > 1)")
        try:       # engines with bf16 activation operands keep a bf16 copy of x: the driver's combine maintains it
            self._xb = eng.buffer("xb", torch.bfloat16)
        except _lib.M3Error:
            self._xb = None
        S, D = eng.buffer("x").numel() // cfg.attention_dim, cfg.attention_dim
        key = (tuple(feat.shape), S)
        if key not in self._buffers:
            cap = agree_capacity(S, eng.device, self.group)
            self._buffers[key] = EpBuffers(S, D, _world(self.group), self.e_loc, cap, eng.device, F=cfg.hidden_units)
        # per layer: (first replaced stage, stage after the last replaced one)
        plan, cur = [], 0
        for i in range(cfg.num_blocks):
            # world == 1 engines fuse gate + index ("moe_gate_index", runs before): then only expert/combine are replaced
            k = "blocks.%d.moe_local.index" % i
            first = names.index(k) if k in names else names.index("blocks.%d.moe_local.expert" % i)
            plan.append((cur, first))
            cur = names.index("blocks.%d.moe_local.combine" % i) + 1
        self._plan, self._tail, self._bound = plan, (cur, len(names)), (logits, S, D, self._buffers[key])
        return logits

    def enqueue(self):
        """Enqueue one forward on the engine's stream: native stages + exchange, no host synchronisation."""
        eng, cfg = self.eng, self.eng.cfg
        logits, S, D, bufs = self._bound
        x = eng.buffer("x").view(S, D)
        xn = eng.buffer("xn").view(S, D)
        with torch.cuda.stream(eng.stream):
            for i, (a, b) in enumerate(self._plan):
                eng.run_stages(a, b)
                gidx = eng.buffer("blocks.%d.gate_idx" % i, torch.int32)
                gval = None if cfg.keep_expert_output else eng.buffer("blocks.%d.gate_value" % i)
                L = self.layers[i]
                ep_moe_layer(xn, gidx, gval, L["w"], self.e_loc, resid=x, alpha=0.5, ln=L["ln"], out=x,
                             backend=self.backend, group=self.group, buffers=bufs,
                             out_bf16=None if self._xb is None else self._xb.view(-1)[:S * D].view(S, D))
            eng.run_stages(*self._tail)
        return logits

    def forward(self, feat, feat_len):
        b = getattr(self, "_bound_io", None)
        if b is None or b[0] is not feat or b[1] is not feat_len:
            self.bind(feat, feat_len)
            self._bound_io = (feat, feat_len)
        logits = self.enqueue()
        self.eng.stream.synchronize()
        return logits
