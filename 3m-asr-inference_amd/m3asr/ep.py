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
                                       w1_scale=w.get("s1"), w2_scale=w.get("s2"), out=out, workspace=workspace,
                                       h_scale=w.get("h_scale"))

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
    """Drives an engine whose stage list holds the expert-parallel stages (m3asr.engine.Engine built with cfg.ep_world_size
    > 1, or with ep_stages=True for a one-rank rehearsal): every kernel -- the exchange bookkeeping, the grouped expert FFN
    on the received rows (plan layouts, fp8 arithmetic with the calibrated H scale included), the combine -- is a native
    stage on the engine's stream; this class only puts the two all-to-alls of every MoE layer between them
    (``blocks.N.moe_ep.send`` -> exchange -> ``.moe_ep.expert`` -> exchange -> ``.moe_ep.combine``).  Works on padded and on
    packed rows (rows past the live count carry gate_idx -1 and never travel).

    graph=True: the whole forward -- native stages and collectives -- is captured once per binding into one graph on the
    engine's stream and replayed (RCCL collectives are capturable; with one rank the engine's own hipGraph is used).  A
    transport that stages through the host (gloo with device tensors) or a failed capture falls back to eager enqueueing;
    ``self.graph_state`` says which ("engine graph" / "captured" / "eager: <reason>")."""

    def __init__(self, engine, group=None, graph=True, capacity_factor=None):
        """capacity_factor: None = every wire chunk can hold ALL rows of the largest rank (nothing can overflow; world x the bytes
        a balanced routing needs).  A number f bounds the chunk at ceil(f x rows / world): the wire shrinks by world / f; a chunk
        that would need more rows reports it on the device ("ep.overflow"), `forward` then repeats the forward with a capacity
        that fits (all ranks agree through one MAX all-reduce per forward) -- rows are never dropped, unlike FastMoE's
        capacity_factor (the reference's moe_conf default is -1: no dropping, ...domain_acc_hier.py:98-113)."""
        if not getattr(engine, "ep_stages", False):
            raise RuntimeError("ExpertParallelEncoder needs an engine with the expert-parallel stages (cfg.ep_world_size > 1, or "
                               "Engine(..., ep_stages=True) for a one-rank rehearsal)")
        self.eng, self.group = engine, group
        self.world = _world(group)
        if self.world != max(1, engine.cfg.ep_world_size):
            raise RuntimeError("process group of %d ranks, engine built for ep_world_size=%d" % (self.world, engine.cfg.ep_world_size))
        self.e_loc = engine.cfg.num_experts
        self.want_graph = bool(graph)
        self.graph_state = "eager: not bound"
        self._graph = None
        self._caps = {}
        self.capacity_factor = None if capacity_factor is None else float(capacity_factor)
        self.reruns = 0               # forwards repeated because a chunk overflowed its bounded capacity

    def on_host(self):
        return self.world > 1 and dist.get_backend(self.group) == "gloo"

    def bind(self, feat, feat_len):
        """Bind the input buffers and agree on the wire capacity (the only collective outside the layers; once per shape)."""
        eng, cfg = self.eng, self.eng.cfg
        B, T = int(feat.shape[0]), int(feat.shape[1])
        S = B * eng.output_shape(B, T)[1]
        key = (B, T)
        if key not in self._caps:
            full = agree_capacity(S, eng.device, self.group)
            cap = full
            if self.capacity_factor is not None and self.world > 1:
                cap = min(full, -(-int(self.capacity_factor * full / self.world + 0.999) // 16) * 16)    # whole 16-row steps
            self._caps[key] = max(1, cap)
            self._full_cap = full
        eng.set_ep_capacity(self._caps[key])
        logits = eng.bind(feat, feat_len)
        names = eng.stage_names()
        D = cfg.attention_dim
        wa, wb = eng.buffer("ep.wire_a"), eng.buffer("ep.wire_b")
        self._wire = (wa.view(self.world, -1, D), wb.view(self.world, -1, D))
        # stage ranges between collectives: one rank -> the engine holds the exchanges itself (device copies)
        cuts = []
        if self.world > 1:
            for i in range(cfg.num_blocks):
                cuts.append(names.index("blocks.%d.moe_ep.send" % i) + 1)
                cuts.append(names.index("blocks.%d.moe_ep.expert" % i) + 1)
        self._segments = list(zip([0] + cuts, cuts + [len(names)]))
        self._bound = (logits, S, D, self._caps[key])
        self._graph = None
        self.graph_state = "eager: not captured yet"
        return logits

    def _enqueue_eager(self):
        eng = self.eng
        wa, wb = self._wire
        with torch.cuda.stream(eng.stream):
            for k, (a, b) in enumerate(self._segments):
                eng.run_stages(a, b)
                if k + 1 < len(self._segments):
                    _all_to_all_equal(wb, wa, self.group)

    def _capture(self):
        """One graph for the whole forward.  Returns the reason it cannot be used, or None."""
        eng = self.eng
        if self.world == 1:
            return None                                    # the native engine captures its own stage list (exchanges = copies)
        if self.on_host():
            return "the gloo transport stages device tensors through the host"
        try:
            eng.stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=eng.stream, capture_error_mode="thread_local"):
                wa, wb = self._wire
                for k, (a, b) in enumerate(self._segments):
                    eng.run_stages(a, b)
                    if k + 1 < len(self._segments):
                        dist.all_to_all_single(wb, wa, group=self.group)
            self._graph = g
            return None
        except Exception as ex:      # noqa: BLE001 -- a transport that cannot be captured still runs eagerly
            self._graph = None
            return "capture failed: %r" % (ex,)

    def enqueue(self):
        """Enqueue one forward on the engine's stream: native stages + exchanges, no host synchronisation on the RCCL
        path (the gloo rehearsal stages every exchange through the host: 2 synchronisations per layer)."""
        eng = self.eng
        logits = self._bound[0]
        if self.want_graph and self._graph is None and self.graph_state.startswith("eager: not captured"):
            self._enqueue_eager()                           # first call of a binding runs eagerly (kernel attributes, RCCL set-up)
            why = self._capture()
            self.graph_state = ("engine graph" if self.world == 1 else "captured") if why is None else "eager: " + why
            return logits
        if self.want_graph and self.world == 1 and self.graph_state == "engine graph":
            eng.forward(use_graph=True)
        elif self._graph is not None:
            with torch.cuda.stream(eng.stream):
                self._graph.replay()
        else:
            self._enqueue_eager()
        return logits

    def overflow_needed(self):
        """After a forward (stream synchronised): 0, or the rows per chunk the largest chunk of any rank would have needed
        (MAX over the ranks: the capacity is a shape every rank shares)."""
        need = int(self.eng.buffer("ep.overflow", torch.int32)[0].item())
        if self.world > 1:
            t = torch.tensor([need], dtype=torch.int64, device="cpu" if self.on_host() else self.eng.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            need = int(t.item())
        return need

    def host_syncs_per_forward(self):
        return 2 * self.eng.cfg.num_blocks if self.on_host() else 0

    def forward(self, feat, feat_len):
        b = getattr(self, "_bound_io", None)
        if b is None or b[0] is not feat or b[1] is not feat_len:
            self.bind(feat, feat_len)
            self._bound_io = (feat, feat_len)
        logits = self.enqueue()
        self.eng.stream.synchronize()
        if self.capacity_factor is not None:
            need = self.overflow_needed()
            while need > self._bound[3]:           # a chunk overflowed: this forward's result is invalid -> larger wire, again
                B, T = int(feat.shape[0]), int(feat.shape[1])
                self._caps[(B, T)] = min(self._full_cap, -(-int(need * 1.25) // 16) * 16)
                self.reruns += 1
                self.bind(feat, feat_len)
                logits = self.enqueue()
                self.eng.stream.synchronize()
                need = self.overflow_needed()
        return logits


class InProcessRanks:
    """Rehearsal of an N-rank expert-parallel forward inside ONE process on ONE device: N engines (rank r built with
    cfg.ep_world_size = N, cfg.ep_rank = r: its own expert shard, its own utterances, its own stream and workspace) are
    stepped segment by segment and the all-to-all is performed by device copies between their wire buffers
    (chunk j of rank i's send wire -> chunk i of rank j's receive wire: exactly what all_to_all_single with equal splits
    does).  Every native stage, every wire format and the capacity agreement are the real ones; only the transport is
    replaced.  Used to check an 8-rank configuration at real dimensions on a box with one GPU (a GPU box admits few
    processes per device, and RCCL refuses several ranks on one device)."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.world = len(self.engines)
        for r, e in enumerate(self.engines):
            if e.cfg.ep_world_size != self.world or e.cfg.ep_rank != r:
                raise RuntimeError("engine %d was built for rank %d of %d" % (r, e.cfg.ep_rank, e.cfg.ep_world_size))

    def forward(self, feats, feat_lens, capacity=None):
        """capacity: rows per wire chunk (None = the largest row count of any rank: nothing can overflow).  After the call
        ``self.overflow`` = rows the fullest chunk of any rank needed beyond a bounded capacity (0 = the result is valid)."""
        engs, W = self.engines, self.world
        rows = [int(f.shape[0]) * e.output_shape(int(f.shape[0]), int(f.shape[1]))[1] for e, f in zip(engs, feats)]
        cap = max(rows) if capacity is None else int(capacity)    # agree_capacity: the largest row count of any rank
        logits, wires, segs = [], [], None
        for e, f, l in zip(engs, feats, feat_lens):
            e.set_ep_capacity(cap)
            logits.append(e.bind(f, l))
            D = e.cfg.attention_dim
            wires.append((e.buffer("ep.wire_a").view(W, -1, D), e.buffer("ep.wire_b").view(W, -1, D)))
            names = e.stage_names()
            cuts = []
            for i in range(e.cfg.num_blocks):
                cuts.append(names.index("blocks.%d.moe_ep.send" % i) + 1)
                cuts.append(names.index("blocks.%d.moe_ep.expert" % i) + 1)
            s = list(zip([0] + cuts, cuts + [len(names)]))
            assert segs is None or len(s) == len(segs)
            segs = s if segs is None else segs
            e._segs = s
        for k in range(len(segs)):
            for e in engs:
                a, b = e._segs[k]
                e.run_stages(a, b)
            for e in engs:
                e.stream.synchronize()
            if k + 1 < len(segs):
                for i in range(W):                                # rank i's chunk j -> rank j's chunk i
                    for j in range(W):
                        wires[j][1][i].copy_(wires[i][0][j])
                torch.cuda.synchronize()
        self.overflow = max(int(e.buffer("ep.overflow", torch.int32)[0].item()) for e in engs)
        return logits
