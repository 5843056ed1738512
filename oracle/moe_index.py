"""numpy restatement of the MoE indexing contract (integer work, bit-exact).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ScatterMappingKernel (TRTAPI++/plugin/fmoe_expert_plugin/fmoe_expert_kernel.cu:25-90):
  his[e+1] = #{i : g_i = e}; inclusive scan -> acc_histogram[0..E] (:54-56,72);
  mapping[i] = acc_histogram[g_i] + rank_i (:33-37,67-70).
The reference takes rank_i from shared-memory atomicAdd arrival order, which is not
deterministic; the pinned (canonical) outcome is the STABLE rank  rank_i = #{j<i : g_j = g_i},
i.e. the inverse permutation of FastMoE's ``pos = argsort(gate)`` taken stably
(trainer_3m_fix/fmoe/functions.py:30).  ``acc_histogram`` and the permutation property are
order-independent and identical to the reference's.

Extension pinned here (not in the reference): g_i < 0 marks a dropped row (padded frame);
it gets mapping -1 and is not counted.
"""
import numpy as np


def moe_index_ref(gate_idx, num_expert):
    g = np.asarray(gate_idx, dtype=np.int32).reshape(-1)
    acc = np.zeros(num_expert + 1, dtype=np.int32)
    valid = g >= 0
    cnt = np.bincount(g[valid], minlength=num_expert).astype(np.int32)
    acc[1:] = np.cumsum(cnt, dtype=np.int64).astype(np.int32)
    mapping = np.full(g.shape[0], -1, dtype=np.int32)
    order = np.argsort(np.where(valid, g, num_expert), kind="stable")   # FastMoE pos (stable)
    nv = int(valid.sum())
    mapping[order[:nv]] = np.arange(nv, dtype=np.int32)
    return mapping, acc


def moe_index_loops(gate_idx, num_expert):
    """The same contract as literal loops (small cases; cross-checks the vectorised form)."""
    g = [int(v) for v in np.asarray(gate_idx).reshape(-1)]
    cnt = [0] * num_expert
    rank = [-1] * len(g)
    for i, e in enumerate(g):
        if e >= 0:
            rank[i] = cnt[e]
            cnt[e] += 1
    acc = [0] * (num_expert + 1)
    for e in range(num_expert):
        acc[e + 1] = acc[e] + cnt[e]
    mapping = [acc[e] + rank[i] if e >= 0 else -1 for i, e in enumerate(g)]
    return np.array(mapping, dtype=np.int32), np.array(acc, dtype=np.int32)


def local_scatter_ref(x, mapping, n_rows=None):
    """ScatterMappingCopyKernel (fmoe_expert_kernel.cu:92-118): out[mapping[s]] = x[s]."""
    x = np.asarray(x)
    mapping = np.asarray(mapping)
    n = int(n_rows if n_rows is not None else (mapping.max() + 1 if mapping.size else 0))
    out = np.zeros((n,) + x.shape[1:], dtype=x.dtype)
    sel = mapping >= 0
    out[mapping[sel]] = x[sel]
    return out


def local_gather_ref(buf, mapping):
    """GatherrMappingCopyKernel (fmoe_expert_kernel.cu:191-217): y[s] = buf[mapping[s]];
    dropped rows (mapping < 0) read as 0."""
    buf = np.asarray(buf)
    mapping = np.asarray(mapping)
    out = np.zeros((mapping.shape[0],) + buf.shape[1:], dtype=buf.dtype)
    sel = mapping >= 0
    out[sel] = buf[mapping[sel]]
    return out


def ep_exchange_counts_ref(local_counts_per_rank, world, e_loc):
    """moe_prepare_forward (trainer_3m_fix/fmoe/functions.py:13-52) in numpy, for all ranks at
    once: local_expert_count[r][g] (g = global expert id = owner*e_loc + local id) ->
    global_expert_count[r][j*e_loc+i] = rows arriving at rank r from rank j for local expert i,
    fwd_expert_count[r] = global_expert_count[r].view(world, e_loc).sum(0) (:43-44)."""
    lc = np.asarray(local_counts_per_rank).reshape(world, world, e_loc)   # [src][owner][i]
    gc = lc.transpose(1, 0, 2).copy()                                     # [owner][src][i]
    fwd = gc.sum(1)
    return gc.reshape(world, world * e_loc), fwd


def ep_send_map_ref(gate_idx, mapping, acc, world, e_loc, capacity, row_words):
    """CPU statement of m3_ep_send_map (3m-asr-inference_amd/csrc/ep_exchange.hip): wire row of every token and the
    header rows (int32 counts per local expert of the destination rank) of a [world, 1 + capacity, row_words] wire."""
    S = len(gate_idx)
    map_send = np.full(S, -1, dtype=np.int32)
    headers = np.zeros((world, 1 + capacity, row_words), dtype=np.int32)
    for j in range(world):
        for i in range(e_loc):
            headers[j, 0, i] = acc[j * e_loc + i + 1] - acc[j * e_loc + i]
    for s in range(S):
        g = int(gate_idx[s])
        if 0 <= g < world * e_loc:
            j = g // e_loc
            off = int(mapping[s]) - int(acc[j * e_loc])
            map_send[s] = j * (capacity + 1) + 1 + off if off < capacity else -1
    return map_send, headers


def ep_recv_gate_ref(headers, world, e_loc, capacity):
    """CPU statement of m3_ep_recv_gate: local expert id of every wire row of the received chunks (-1: header row /
    unused capacity); headers[j, 0, :e_loc] = rows from rank j per local expert."""
    gate = np.full((world, 1 + capacity), -1, dtype=np.int32)
    for j in range(world):
        t = 0
        for i in range(e_loc):
            c = max(int(headers[j, 0, i]), 0)
            c = min(c, capacity - t)
            gate[j, 1 + t: 1 + t + c] = i
            t += c
    return gate.reshape(-1)
