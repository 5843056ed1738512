"""CPU restatement of the reference's CTC searches and streaming operators -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(libm3asr_hip.so through m3asr.ops / m3asr.decode) never does.

Pinned by tests/golden/ctc_decode.npz: outputs of the reference's own `BaseCTCEncoder.ctc_greedy_search` /
`ctc_prefix_beam_search` (trainer_3m_fix/model/encoder.py:156-275) run in the build container on synthetic score
matrices (generator: oracle/gen_golden_ctc.py).  The streaming operators have no reference test or fixture and their
CUDA sources cannot be built here: their restatements below follow the kernel text and are "parity unpinned".
"""
import math

import numpy as np

NEG_INF = -float("inf")


def ctc_greedy_search(logits, lens, blank=0):
    """model/encoder.py:156-180: argmax per frame (first maximum), then per utterance skip repeats and blanks over the
    first lens[b] frames.  logits (B,T,V) array-like -> list of B python lists."""
    x = np.asarray(logits)
    ids = x.argmax(axis=-1)
    hyps = []
    for b in range(x.shape[0]):
        n = int(lens[b])
        row = ids[b, :n]
        prev = np.concatenate([[-1], row[:-1]]) if n else row
        hyps.append([int(v) for v in row[(row != prev) & (row != blank)]])
    return hyps


def log_add(*terms):
    """utils/common.py:148-156 (python floats = double precision)."""
    m = max(terms)
    if m == NEG_INF:
        return NEG_INF
    return m + math.log(sum(math.exp(t - m) for t in terms))


def log_softmax(x):
    x = np.asarray(x, dtype=np.float32)
    m = x.max(axis=-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True, dtype=np.float32))).astype(np.float32)


def topk_desc(logp, k):
    """k best entries of each row, (value desc, index asc) -- torch.topk's order whenever the k+1 best values are distinct."""
    logp = np.asarray(logp)
    order = np.lexsort((np.broadcast_to(np.arange(logp.shape[-1]), logp.shape), -logp), axis=-1)[..., :k]
    return np.take_along_axis(logp, order, -1), order.astype(np.int32)


def prefix_beam_search_topk(top_logp, top_idx, beam, blank=0):
    """The recursion of model/encoder.py:232-275 over per-frame candidate lists (top_logp, top_idx of shape (T,k)).

    State per prefix: (pb, pnb) = log prob of all alignments of the prefix ending in blank / not in blank.
    Per frame, every candidate symbol s with log prob ps extends every kept prefix:
      s blank           -> same prefix:  pb  (+)= pb+ps, pnb+ps
      s == last symbol  -> same prefix:  pnb (+)= pnb+ps            (repeat collapses)
                           prefix+s:     pnb (+)= pb+ps             (repeat after a blank is a new symbol)
      otherwise         -> prefix+s:     pnb (+)= pb+ps, pnb+ps
    then the `beam` best by log_add(pb, pnb) survive; ties keep first-touch order (stable sort of an insertion-ordered
    dict, :266-269)."""
    beams = {(): (0.0, NEG_INF)}
    for lp_t, ix_t in zip(np.asarray(top_logp), np.asarray(top_idx)):
        grown = {}

        for ps, s in zip((float(v) for v in lp_t), (int(v) for v in ix_t)):
            for prefix, (pb, pnb) in beams.items():
                if s == blank:
                    n_pb, n_pnb = grown.get(prefix, (NEG_INF, NEG_INF))
                    grown[prefix] = (log_add(n_pb, pb + ps, pnb + ps), n_pnb)
                elif prefix and s == prefix[-1]:
                    n_pb, n_pnb = grown.get(prefix, (NEG_INF, NEG_INF))
                    grown[prefix] = (n_pb, log_add(n_pnb, pnb + ps))
                    ext = prefix + (s,)
                    n_pb, n_pnb = grown.get(ext, (NEG_INF, NEG_INF))
                    grown[ext] = (n_pb, log_add(n_pnb, pb + ps))
                else:
                    ext = prefix + (s,)
                    n_pb, n_pnb = grown.get(ext, (NEG_INF, NEG_INF))
                    grown[ext] = (n_pb, log_add(n_pnb, pb + ps, pnb + ps))
        ranked = sorted(grown.items(), key=lambda kv: log_add(*kv[1]), reverse=True)
        beams = dict(ranked[:beam])
    return [(p, log_add(*v)) for p, v in beams.items()]


def ctc_prefix_beam_search(logits, beam, blank=0):
    """model/encoder.py:182-275 for one utterance: logits (T,V) -> [(prefix, score)] best first."""
    lp, ix = topk_desc(log_softmax(logits), beam)
    return prefix_beam_search_topk(lp, ix, beam, blank)


# ------------------------------------------------------------------------------------------ streaming operators
def cat_split_cache(in_cache, inp):
    """cat_split_cache_kernel.cu:30-107: output = cache ++ input along the last axis; new cache = its last cache_dim values."""
    out = np.concatenate([in_cache, inp], axis=1)
    cd = in_cache.shape[1]
    return out, out[:, out.shape[1] - cd:].copy()


def att_stream_softmax(scores, decode_frame_num, mask_idx, cache_len, scale):
    """att_stream_softmax_kernel.cu:28-71,136-191 on scores (B,N,ld): valid keys [max(0, ld-dfn[b]), min(ld, mask[b])+cache_len)
    (clamped to ld), softmax of (x - max)*scale there, 0 elsewhere."""
    x = np.asarray(scores, dtype=np.float32)
    B, N, ld = x.shape
    out = np.zeros_like(x)
    for b in range(B):
        first = max(0, ld - int(decode_frame_num[b]))
        last = min(ld, min(ld, int(mask_idx[b])) + cache_len)
        if last <= first:
            continue
        v = x[b, :, first:last]
        e = np.exp((v - v.max(axis=-1, keepdims=True)) * np.float32(scale))
        out[b, :, first:last] = e / e.sum(axis=-1, keepdims=True)
    return out


def rel_positional_encoding(x, pe, scale, frame_num=None):
    """rel_positional_encoding_kernel.cu:62-69 and the streaming contract stated at :108-111."""
    T = x.shape[1]
    off = 0 if frame_num is None else int(frame_num[0])
    y = np.asarray(x, dtype=np.float32) * np.float32(scale)
    pos = np.asarray(pe).reshape(-1, x.shape[2])[off:off + T][None]
    return (y, pos) if frame_num is None else (y, pos, np.asarray(frame_num) + T)
