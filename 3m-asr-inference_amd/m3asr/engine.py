"""Python handle on the native encoder engine (m3_engine_* in include/m3asr.h).

Plays the role of TensorRT's ICudaEngine + IExecutionContext in the reference
(infer.py:27-103, TRTAPI++/python/trt_helper/infer_helper.py:37-161): weights live in device
memory owned by this object (torch tensors), activations in a caller-visible workspace tensor, the
forward is one C call that replays a hipGraph.
"""
import collections
import ctypes as C

import torch

from . import _lib
from ._lib import check
from .config import EncoderConfig, subsampled_len
from .plan import pack_weights


_TORCH_DTYPE = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.int32: _lib.I32, torch.float8_e4m3fn: _lib.FP8}


def _engine_config(cfg: EncoderConfig, fold_pos_proj, debug_taps, fuse_route=False, bf16_activations=True, packed_rows=None,
                   ep_stages=False, fork_embed=None):
    ec = _lib.EngineConfig()
    ec.input_dim, ec.output_dim = cfg.input_dim, cfg.output_dim
    ec.attention_dim, ec.attention_heads, ec.num_blocks = cfg.attention_dim, cfg.attention_heads, cfg.num_blocks
    ec.embed_dim, ec.embed_heads = cfg.embed_dim, cfg.embed_heads
    ec.embed_linear_units, ec.embed_blocks = cfg.embed_linear_units, cfg.embed_blocks
    ec.num_experts, ec.hidden_units = cfg.num_experts, cfg.hidden_units
    ec.cnn_module_kernel = cfg.cnn_module_kernel
    ec.cnn_layer_norm = int(cfg.cnn_module_norm == "layer_norm")
    ec.embed_cnn_layer_norm = int(cfg.embed_cnn_module_norm == "layer_norm")
    ec.router_with_bias, ec.keep_expert_output = int(cfg.router_with_bias), int(cfg.keep_expert_output)
    ec.ep_world_size, ec.ep_rank = cfg.ep_world_size, cfg.ep_rank
    ec.fold_pos_proj, ec.debug_taps, ec.fuse_route = int(fold_pos_proj), int(debug_taps), int(fuse_route)
    ec.log_softmax_out = int(cfg.log_softmax_out)
    ec.bf16_activations = 0 if bf16_activations else -1
    ec.weight_dtype = {"f32": _lib.F32, "bf16": _lib.BF16, "fp8": _lib.FP8}[cfg.weight_dtype]
    ec.packed_rows = 0 if packed_rows is None else (1 if packed_rows else -1)
    ec.fp8_activations = int(bool(getattr(cfg, "fp8_activations", False)) and cfg.weight_dtype == "fp8")
    ec.ep_stages = int(bool(ep_stages))
    ec.fork_embed = 0 if fork_embed is None else (1 if fork_embed else -1)
    ec.static_chunk_size = int(getattr(cfg, "static_chunk_size", 0))
    ec.num_left_chunks = int(getattr(cfg, "num_decoding_left_chunks", -1))
    ec.causal, ec.embed_causal = int(bool(getattr(cfg, "causal", False))), int(bool(getattr(cfg, "embed_causal", False)))
    return ec


class Engine:
    def __init__(self, cfg: EncoderConfig, packed, device="cuda:0", fold_pos_proj=True, debug_taps=False,
                 fuse_route=False, bf16_activations=True, packed_rows=None, max_shapes=8, ep_stages=False, fork_embed=None):
        """packed: output of plan.pack_weights / plan.load_plan (CPU tensors; GEMM weights in cfg.weight_dtype), or the ``weights`` dict of another
        Engine on the same device (several execution contexts sharing one copy of the weights, like TensorRT's
        multiple IExecutionContexts per engine).
        fold_pos_proj (default on): p = linear_pos(pos_emb[:T']) of every block depends on the weights and on T' only, not
        on the input, so it is computed once when a shape is bound (constant folding) instead of in every forward.
        bf16_activations (16-bit modes): on long batches keep GEMM-only activations and a copy of the residual stream as bf16
        (automatic); False disables it (needed when stages are replaced from the host, e.g. ExpertParallelEncoder).
        fuse_route: 0 / False = staged route (router GEMM on cat([embed, x]) with a LayerNorm prologue that writes xn);
        1 / True = router + top-1 + index in one single-workgroup launch (S <= 256); 2 = split route (embed half of all
        routers in one GEMM per forward, x half as a folded-LayerNorm GEMM, norm_ff applied by the expert kernel).
        ep_stages: build the expert-parallel stage list ("blocks.N.moe_ep.*") although cfg.ep_world_size is 1 -- a one-rank
        rehearsal of the exchange (m3asr/ep.py); cfg.ep_world_size > 1 always builds it.
        fork_embed: the embed encoder as a second branch of the captured graph beside the main encoder's start (None =
        automatic: short inputs; True / False force it)."""
        self.lib = _lib.load()
        from .plan import EXPERT_SLICE
        assert self.lib.m3_moe_expert_slice() == EXPERT_SLICE, "plan.EXPERT_SLICE out of sync with libm3asr_hip.so"
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.weights = {k: (v if v.is_cuda else v.to(self.device, non_blocking=False)).contiguous() for k, v in packed.items()}
        names = list(self.weights)
        table = (_lib.WeightEntry * len(names))()
        self._keep = [n.encode() for n in names]
        for i, n in enumerate(names):
            table[i].name = self._keep[i]
            table[i].data = self.weights[n].data_ptr()
            table[i].numel = self.weights[n].numel()
            table[i].dtype = _TORCH_DTYPE[self.weights[n].dtype]
        self.ep_stages = bool(ep_stages) or cfg.ep_world_size > 1
        ec = _engine_config(cfg, fold_pos_proj, debug_taps, int(fuse_route) if not self.ep_stages else 0, bf16_activations, packed_rows,
                            ep_stages=ep_stages, fork_embed=fork_embed)
        self.handle = self.lib.m3_engine_create(C.byref(ec), table, len(names))
        if not self.handle:
            raise _lib.M3Error("m3_engine_create failed: " + _lib.last_error())
        self.stream = torch.cuda.Stream(device=self.device)
        # (B, T) -> workspace / static I/O buffers, LRU-bounded like the native shape cache (current binding + 7 parked):
        # a server fed unbucketed lengths must not keep a workspace per length ever seen.  Dropping a workspace that a
        # parked native binding still names is harmless: a binding is revived only on an exact pointer match, everything
        # in a workspace is rewritten by the forward, and the folded positional projection lives in engine-owned memory.
        self.max_shapes = max(1, int(max_shapes))
        self._ws = collections.OrderedDict()
        self._static = collections.OrderedDict()
        self._bound = None
        self._ep_capacity = 0

    def set_ep_capacity(self, rows_per_chunk):
        """Expert parallel: rows per wire chunk for the bindings made from now on (the largest B*T' of any rank)."""
        check(self.lib.m3_engine_set_ep_capacity(self.handle, int(rows_per_chunk)), "m3_engine_set_ep_capacity")
        self._ep_capacity = int(rows_per_chunk)

    @classmethod
    def from_state_dict(cls, cfg, state_dict, **kw):
        return cls(cfg, pack_weights(state_dict, cfg), **kw)

    def clone_context(self, **kw):
        """Another execution context (own stream, workspace, hipGraph) over the SAME device weights."""
        return Engine(self.cfg, self.weights, device=str(self.device), **kw)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.m3_engine_destroy(h)

    def weight_bytes(self):
        return sum(v.numel() * v.element_size() for v in self.weights.values())

    def output_shape(self, B, T):
        return (B, subsampled_len(T), self.cfg.output_dim)

    def workspace_size(self, B, T):
        return self.lib.m3_engine_workspace_size(self.handle, B, T)

    def _lru(self, cache, key, make):
        v = cache.get(key)
        if v is None:
            v = cache[key] = make()
            while len(cache) > self.max_shapes:
                cache.popitem(last=False)
        else:
            cache.move_to_end(key)
        return v

    def _workspace(self, B, T):
        return self._lru(self._ws, (B, T, self._ep_capacity),
                         lambda: torch.empty(self.workspace_size(B, T), dtype=torch.uint8, device=self.device))

    def bind(self, feat, feat_len, logits=None):
        """Bind device buffers (feat (B,T,idim) f32, feat_len (1,B)/(B,) i32); returns logits tensor."""
        assert feat.is_cuda and feat.dtype == torch.float32 and feat.is_contiguous()
        assert feat_len.is_cuda and feat_len.dtype == torch.int32 and feat_len.is_contiguous()
        B, T, _ = feat.shape
        if logits is None:
            logits = torch.empty(self.output_shape(B, T), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, T)
        n = self.lib.m3_engine_prepare(self.handle, feat.data_ptr(), feat_len.data_ptr(), B, T, logits.data_ptr(),
                                       ws.data_ptr(), ws.numel())
        if n < 0:
            raise _lib.M3Error("m3_engine_prepare failed: " + _lib.last_error())
        self._bound = (feat, feat_len, logits, ws, B, T)
        return logits

    def forward(self, feat=None, feat_len=None, logits=None, use_graph=True, stream=None):
        """Enqueue one encoder forward on `stream` (default: the engine's own stream).  With no
        arguments the last bound buffers are reused (steady-state replay)."""
        if feat is not None:
            b = self._bound
            if b is None or b[0].data_ptr() != feat.data_ptr() or b[1].data_ptr() != feat_len.data_ptr() or \
                    tuple(b[0].shape) != tuple(feat.shape) or (logits is not None and logits.data_ptr() != b[2].data_ptr()):
                self.bind(feat, feat_len, logits)
        feat, feat_len, logits, ws, B, T = self._bound
        st = stream if stream is not None else self.stream
        check(self.lib.m3_engine_forward(self.handle, feat.data_ptr(), feat_len.data_ptr(), B, T, logits.data_ptr(),
                                         ws.data_ptr(), ws.numel(), int(use_graph), C.c_void_p(st.cuda_stream)),
              "m3_engine_forward")
        return logits

    def infer(self, feat, feat_len):
        """Serving entry point: static per-shape device buffers + graph replay.  feat (B,T,idim) / feat_len (B,) or (1,B) on
        any device are copied into this shape's input buffers, the shape's hipGraph is replayed on the engine stream
        (captured on first use; the native engine keeps the stage lists and graphs of the last few shapes, so alternating
        between length buckets re-captures nothing) and the shape's logits buffer is returned -- valid until the next
        infer() of the same shape."""
        B, T = int(feat.shape[0]), int(feat.shape[1])
        f, l, out = self._lru(self._static, (B, T), lambda: (
            torch.empty(B, T, self.cfg.input_dim, dtype=torch.float32, device=self.device),
            torch.empty(1, B, dtype=torch.int32, device=self.device),
            torch.empty(self.output_shape(B, T), dtype=torch.float32, device=self.device)))
        with torch.cuda.stream(self.stream):
            f.copy_(feat, non_blocking=True)
            l.copy_(feat_len.reshape(1, B).to(torch.int32), non_blocking=True)
        self.forward(f, l, out, use_graph=True)
        self.stream.synchronize()
        return out

    def num_captures(self):
        return self.lib.m3_engine_num_captures(self.handle)

    def __call__(self, feat, feat_len):
        """Synchronous convenience: waits for prior work on the current stream, runs, syncs."""
        self.stream.wait_stream(torch.cuda.current_stream())
        out = self.forward(feat, feat_len, use_graph=False)
        self.stream.synchronize()
        return out

    def streaming(self, B, max_frames, history_frames=None):
        """A set of B streams decoded chunk by chunk (see StreamingEncoder)."""
        return StreamingEncoder(self, B, max_frames, history_frames)

    # ---- staged execution / taps ---------------------------------------------------------
    def stage_names(self):
        n = self.lib.m3_engine_num_stages(self.handle)
        return [self.lib.m3_engine_stage_name(self.handle, i).decode() for i in range(n)]

    def num_kernels(self):
        return self.lib.m3_engine_num_kernels(self.handle)

    def stage_info(self):
        """Per stage of the bound shape: dict(name, kernel, launches, per_row, alg_bytes, flops) -- what the stage launches
        and its algorithmic HBM bytes / FLOPs (alg_bytes < 0: data-dependent, the grouped expert FFN)."""
        out = []
        for i, name in enumerate(self.stage_names()):
            si = _lib.StageInfo()
            check(self.lib.m3_engine_stage_info(self.handle, i, C.byref(si)), "m3_engine_stage_info")
            out.append(dict(name=name, kernel=(si.kernel or b"").decode(), launches=si.launches, per_row=bool(si.per_row),
                            alg_bytes=si.alg_bytes, flops=si.flops))
        return out

    def run_stages(self, first, last, stream=None):
        st = stream if stream is not None else self.stream
        check(self.lib.m3_engine_run(self.handle, first, last, C.c_void_p(st.cuda_stream)), "m3_engine_run")

    def packed_rows(self):
        """True when the bound shape runs its blocks on packed (padding-free) rows: "x" / "xn" / "embed" then hold the valid
        frames of all utterances back to back, buffer("row0", int32) [B+1] gives each utterance's first row."""
        try:
            self.buffer("row0", torch.int32)
            return True
        except _lib.M3Error:
            return False

    def rows_padded(self, name, dtype=torch.float32, fill=0):
        """A per-row intermediate ("x", "xn", "embed", "blocks.N.gate_idx", ...) as a (B, T', width) tensor whatever the row
        layout of the bound shape: a view for padded rows, a copy (frames past an utterance's end = `fill`) for packed rows."""
        B, T = self._bound[0].shape[0], self._bound[0].shape[1]
        Tp = self.output_shape(B, T)[1]
        rows = self.buffer(name, dtype).view(B * Tp, -1)
        if not self.packed_rows():
            return rows.view(B, Tp, -1)
        row0 = self.buffer("row0", torch.int32).tolist()
        out = torch.full((B, Tp, rows.shape[1]), fill, dtype=dtype, device=rows.device)
        for b in range(B):
            n = row0[b + 1] - row0[b]
            out[b, :n] = rows[row0[b]:row0[b + 1]]
        return out

    def buffer(self, name, dtype=torch.float32, ws=None):
        """Zero-copy view of a named intermediate inside the bound workspace (ws: the workspace of the binding that ran last
        when it is not this object's own, e.g. a StreamingEncoder's)."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        check(self.lib.m3_engine_buffer(self.handle, name.encode(), C.byref(ptr), C.byref(nbytes)), "m3_engine_buffer")
        ws = self._bound[3] if ws is None else ws
        off = ptr.value - ws.data_ptr()
        assert 0 <= off and off + nbytes.value <= ws.numel()
        return ws[off: off + nbytes.value].view(dtype)


class StreamingEncoder:
    """Chunk-by-chunk decoding of B utterances side by side (m3_engine_forward_chunk): the decoding-chunk semantics of the
    reference's encoders (model/encoder.py:100-140: decoding_chunk_size, num_decoding_left_chunks) with per-layer K / V
    history and depthwise-conv caches carried in device memory -- what the reference's CatSplitCache / AttStreamSoftmax /
    streaming RelPositionalEncoding plugins were written for.  The engine's config must have static_chunk_size = c > 0 and
    causal = embed_causal = True.

        st = engine.streaming(B, max_frames)         # max_frames: longest stream in OUTPUT frames (T')
        for n in range(n_chunks):
            logits_chunk = st.step(window_n, valid_n)      # (B, c, V); window_n (B, 4c+3, idim), valid_n (B,) real frames in it

    `decode(feat, feat_len)` cuts whole utterances into windows itself and returns (B, T', V) like Engine.__call__."""

    def __init__(self, engine, B, max_frames, history_frames=None):
        self.eng, cfg = engine, engine.cfg
        self.c = int(cfg.static_chunk_size)
        if self.c <= 0 or not (cfg.causal and cfg.embed_causal):
            raise _lib.M3Error("StreamingEncoder needs static_chunk_size > 0 and causal conv modules in both encoders")
        left = int(cfg.num_decoding_left_chunks)
        max_frames = -(-int(max_frames) // self.c) * self.c                 # whole chunks
        if history_frames is None:
            history_frames = max_frames if left < 0 else (left + 1) * self.c
        self.desc = _lib.StreamDesc(int(B), int(history_frames), int(max_frames))
        self.window = engine.lib.m3_engine_chunk_input_frames(engine.handle)
        n = engine.lib.m3_engine_stream_state_size(engine.handle, C.byref(self.desc))
        if n == 0:
            raise _lib.M3Error("m3_engine_stream_state_size failed: " + _lib.last_error())
        dev = engine.device
        self.state = torch.empty(n, dtype=torch.uint8, device=dev)
        self.feat = torch.zeros(B, self.window, cfg.input_dim, dtype=torch.float32, device=dev)
        self.valid = torch.zeros(B, dtype=torch.int32, device=dev)
        self.logits = torch.empty(B, self.c, cfg.output_dim, dtype=torch.float32, device=dev)
        self.ws = torch.empty(engine.workspace_size(B, self.window), dtype=torch.uint8, device=dev)
        self.chunks = 0
        self.reset()

    def reset(self):
        e = self.eng
        check(e.lib.m3_engine_stream_reset(e.handle, C.byref(self.desc), self.state.data_ptr(), self.state.numel(),
                                           C.c_void_p(e.stream.cuda_stream)), "m3_engine_stream_reset")
        self.chunks = 0

    def step(self, window, valid, use_graph=True):
        """One chunk: window (B, 4c+3, idim) feature frames starting at input frame 4 c n, valid (B,) how many of them are
        real (pass 0 for an utterance with fewer than 7 frames left: no output frame fits).  Returns this object's logits
        buffer (B, c, V), valid until the next step."""
        e = self.eng
        with torch.cuda.stream(e.stream):
            self.feat.copy_(window, non_blocking=True)
            self.valid.copy_(valid.to(torch.int32).reshape(-1), non_blocking=True)
        check(e.lib.m3_engine_forward_chunk(e.handle, C.byref(self.desc), self.state.data_ptr(), self.state.numel(),
                                            self.feat.data_ptr(), self.valid.data_ptr(), self.logits.data_ptr(),
                                            self.ws.data_ptr(), self.ws.numel(), self.chunks, int(use_graph),
                                            C.c_void_p(e.stream.cuda_stream)), "m3_engine_forward_chunk")
        self.chunks += 1
        return self.logits

    def buffer(self, name, dtype=torch.float32):
        """A named intermediate of the chunk that ran last (rows = B x c frames of that chunk)."""
        return self.eng.buffer(name, dtype, ws=self.ws)

    def decode(self, feat, feat_len, use_graph=True):
        """Whole utterances through the chunked path: feat (B, T, idim), feat_len (B,) -> logits (B, T', V) (frames past an
        utterance's end zeroed), the concatenation of the chunk outputs."""
        e, c = self.eng, self.c
        B, T = int(feat.shape[0]), int(feat.shape[1])
        lens = feat_len.reshape(-1).to("cpu", torch.int64)
        Tp = subsampled_len(T)
        n_chunks = -(-Tp // c)
        out = torch.zeros(B, n_chunks * c, e.cfg.output_dim, dtype=torch.float32, device=e.device)
        self.reset()
        e.stream.wait_stream(torch.cuda.current_stream())
        feat = feat.to(e.device)
        padded = torch.zeros(B, max(T, 4 * c * n_chunks + 3), feat.shape[2], dtype=torch.float32, device=e.device)
        padded[:, :T] = feat
        for n in range(n_chunks):
            left = (lens - 4 * c * n).clamp(min=0, max=self.window)
            left = torch.where(left >= 7, left, torch.zeros_like(left))
            lg = self.step(padded[:, 4 * c * n: 4 * c * n + self.window], left, use_graph=use_graph)
            with torch.cuda.stream(e.stream):
                out[:, n * c:(n + 1) * c] = lg
        e.stream.synchronize()
        out = out[:, :Tp]
        valid = torch.arange(Tp).view(1, -1) < torch.tensor([subsampled_len(int(v)) if v >= 7 else 0 for v in lens]).view(-1, 1)
        out[~valid.to(out.device)] = 0
        return out
