"""CTC searches behind the encoder engine -- host mirror of the reference's decoding entry points
(trainer_3m_fix/model/encoder.py:156-275: `ctc_greedy_search`, `ctc_prefix_beam_search`; same names, argument meaning
and return types), running on the device through libm3asr_hip.so:

  greedy        logits stay on the GPU; argmax + repeat/blank collapse are kernels (m3_ctc_greedy); only the token
                lists (B x T' int32 + counts) are copied back.
  prefix beam   per-frame log-softmax + top-`beam` on the GPU (m3_ctc_topk), T' x beam pairs copied back, the prefix
                recursion in the library's host routine (m3_ctc_prefix_beam_search).

There is no CPU fallback: without the HIP library every call raises.
"""
from typing import List, Tuple

import torch

from . import ops


class CtcDecoder:
    """decoder = CtcDecoder(engine, blank_idx=0); engine: m3asr.engine.Engine (feat (B,T,idim), feat_len -> logits (B,T',V))."""

    def __init__(self, engine, blank_idx: int = 0):
        self.engine = engine
        self.blank_idx = int(blank_idx)

    def forward(self, xs: torch.Tensor, xs_lens: torch.Tensor):
        """-> {"out_nosm": logits (B,T',V) on the device, "out_lens": (B,) int32 on the device} (encoder.py:140-147)."""
        dev = self.engine.device
        feat = xs.to(dev, torch.float32).contiguous()
        lens = xs_lens.reshape(1, -1).to(dev, torch.int32).contiguous()
        logits = self.engine(feat, lens)
        out_lens = self.engine.buffer("lens", torch.int32)[:feat.shape[0]].clone()
        return {"out_nosm": logits, "out_lens": out_lens}

    @staticmethod
    def _full_context(decoding_chunk_size, num_decoding_left_chunks):
        # the Conformer-MoE encoder is full-context (non-causal conv, unmasked attention): only the reference's
        # "use full chunk" setting (< 0) describes what the engine computes
        assert decoding_chunk_size != 0, "decoding does not support dynamic chunks"
        if decoding_chunk_size > 0:
            raise NotImplementedError("chunked decoding: the full-context encoder has no chunk mask")

    def ctc_greedy_search(self, xs: torch.Tensor, xs_lens: torch.Tensor, decoding_chunk_size: int = -1,
                          num_decoding_left_chunks: int = -1) -> List[List[int]]:
        self._full_context(decoding_chunk_size, num_decoding_left_chunks)
        res = self.forward(xs, xs_lens)
        return self.greedy_from_logits(res["out_nosm"], res["out_lens"])

    def greedy_from_logits(self, logits: torch.Tensor, out_lens: torch.Tensor) -> List[List[int]]:
        _, tokens, n_tokens = ops.ctc_greedy(logits, out_lens, self.blank_idx)
        tokens, n_tokens = tokens.cpu(), n_tokens.cpu().tolist()
        return [tokens[b, :n].tolist() for b, n in enumerate(n_tokens)]

    def ctc_prefix_beam_search(self, xs: torch.Tensor, xs_lens: torch.Tensor, beam_size: int,
                               decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1
                               ) -> Tuple[List[Tuple[Tuple[int, ...], float]], torch.Tensor]:
        """-> (n-best [(prefix, ctc score)], logits (1,T',V)); batch size 1 as in the reference (encoder.py:213-214).
        The reference returns the encoder's hidden states for attention rescoring as the second item; the AED decoder
        is out of scope here, the scores the search ran on are returned instead."""
        assert xs.shape[0] == xs_lens.reshape(-1).shape[0] == 1, "prefix beam search supports batch size 1"
        self._full_context(decoding_chunk_size, num_decoding_left_chunks)
        res = self.forward(xs, xs_lens)
        return self.prefix_beam_from_logits(res["out_nosm"], beam_size), res["out_nosm"]

    def prefix_beam_from_logits(self, logits: torch.Tensor, beam_size: int):
        """logits (1,T',V) or (T',V) on the device; all T' frames are searched (max_len, encoder.py:222)."""
        x = logits.reshape(-1, logits.shape[-1])
        top_logp, top_idx = ops.ctc_topk(x, beam_size)
        return ops.ctc_prefix_beam_search_host(top_logp.cpu().numpy(), top_idx.cpu().numpy(), beam_size, self.blank_idx)
