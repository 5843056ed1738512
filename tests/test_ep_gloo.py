"""Expert-parallel exchange on CPU with the gloo backend, world_size 2 (the N>1 path of §⑤): every rank routes its own
tokens over experts sharded 4 per rank; the result must equal the single-process result with all experts local.
Compute steps use the CPU oracle as the backend (the product backend is HipBackend = libm3asr_hip.so); what is under
test here is the host logic and the wire format: fixed-shape chunks [world][1 + capacity][D] with the row counts in a
header row (the count exchange rides in the payload), equal-split all-to-all both ways, wire order, reassembly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from m3asr.ep import ep_moe_layer
from oracle.encoder_ref import fmoe_expert
from oracle.moe_index import moe_index_ref, local_scatter_ref, ep_exchange_counts_ref, ep_send_map_ref, ep_recv_gate_ref


class OracleBackend:
    """numpy / torch statements of the device steps (oracle/moe_index.py, oracle/encoder_ref.py)."""

    def index(self, gate_idx, n_total):
        m, a = moe_index_ref(gate_idx.numpy(), n_total)
        return torch.from_numpy(m), torch.from_numpy(a), None

    def send_map(self, gate_idx, mapping, acc, world, e_loc, cap, map_send, wire):
        m, hdr = ep_send_map_ref(gate_idx.numpy(), mapping.numpy(), acc.numpy(), world, e_loc, cap, wire.shape[-1])
        map_send.copy_(torch.from_numpy(m))
        wire.view(torch.int32)[:, 0, :e_loc] = torch.from_numpy(hdr[:, 0, :e_loc])
        return map_send

    def scatter_into(self, x, map_send, wire):
        flat = wire.view(-1, wire.shape[-1])
        keep = map_send >= 0
        flat[map_send[keep].long()] = x[keep]
        return wire

    def recv_gate(self, wire, world, e_loc, cap, gate_recv):
        gate_recv.copy_(torch.from_numpy(ep_recv_gate_ref(wire.view(torch.int32).numpy(), world, e_loc, cap)))
        return gate_recv

    def expert_ffn(self, rows, gate_local, w, out, workspace=None):
        y, _, _ = fmoe_expert(rows.unsqueeze(0), gate_local.view(1, -1, 1), w["w1"], w["b1"], w["w2"], w["b2"])
        out.copy_(y[0])
        return out

    def combine(self, rows_sorted, mapping, gate_value, resid, alpha, ln, out=None):
        m = mapping.long()
        y = torch.zeros(m.numel(), rows_sorted.shape[1])
        y[m >= 0] = rows_sorted[m[m >= 0]]
        if gate_value is not None:
            y = y * gate_value.view(-1, 1)
        y = resid + alpha * y if resid is not None else alpha * y
        if ln is not None:
            y = F.layer_norm(y, (y.shape[1],), ln[0], ln[1], ln[2])
        return y


def _problem(world, e_loc, D, Fh):
    g = torch.Generator().manual_seed(11)
    E = world * e_loc
    w = {"w1": torch.randn(E, Fh, D, generator=g) * D ** -0.5, "b1": torch.randn(E, Fh, generator=g) * 0.1,
         "w2": torch.randn(E, D, Fh, generator=g) * Fh ** -0.5, "b2": torch.randn(E, D, generator=g) * 0.1}
    toks = []
    for r, S in enumerate([37, 5] if world == 2 else [23] * world):
        x = torch.randn(S, D, generator=g)
        gate = torch.randint(-1 if r == 0 else 0, E, (S,), generator=g).to(torch.int32)
        if r == 1:
            gate[:] = 6                                     # a rank whose tokens all go to one remote-or-local expert
        toks.append((x, gate, torch.rand(S, generator=g), torch.randn(S, D, generator=g)))
    ln = (torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1, 1e-12)
    return w, toks, ln


def _worker(rank, world, port, e_loc, D, Fh, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, toks, ln = _problem(world, e_loc, D, Fh)
    mine = {k: v[rank * e_loc:(rank + 1) * e_loc] for k, v in w.items()}      # load_state_dict_comm slice
    x, gate, gval, resid = toks[rank]
    y = ep_moe_layer(x, gate, gval, mine, e_loc, resid=resid, alpha=0.5, ln=ln, backend=OracleBackend())
    np.save(os.path.join(out_dir, "y%d.npy" % rank), y.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_expert_parallel_world2_matches_single_process(tmp_path):
    world, e_loc, D, Fh = 2, 4, 32, 64
    mp.spawn(_worker, args=(world, _free_port(), e_loc, D, Fh, str(tmp_path)), nprocs=world, join=True)
    w, toks, ln = _problem(world, e_loc, D, Fh)
    for r, (x, gate, gval, resid) in enumerate(toks):
        y_full, _, _ = fmoe_expert(x.unsqueeze(0), gate.view(1, -1, 1), w["w1"], w["b1"], w["w2"], w["b2"])
        want = F.layer_norm(resid + 0.5 * gval.view(-1, 1) * y_full[0], (D,), ln[0], ln[1], ln[2])
        got = torch.from_numpy(np.load(os.path.join(str(tmp_path), "y%d.npy" % r)))
        assert torch.allclose(got, want, atol=2e-6, rtol=1e-5), float((got - want).abs().max())


def test_count_exchange_contract():
    """moe_prepare_forward (fmoe/functions.py:37-44): global_expert_count[r][j*E_loc+i] = rows from rank j for my expert i."""
    world, e_loc = 4, 2
    rng = np.random.default_rng(3)
    gates = [rng.integers(0, world * e_loc, 20 + 3 * r) for r in range(world)]
    local = np.stack([np.bincount(g, minlength=world * e_loc) for g in gates])
    gc, fwd = ep_exchange_counts_ref(local, world, e_loc)
    for r in range(world):
        for j in range(world):
            for i in range(e_loc):
                assert gc[r, j * e_loc + i] == np.sum(gates[j] == r * e_loc + i)
        assert np.array_equal(fwd[r], gc[r].reshape(world, e_loc).sum(0))
    assert fwd.sum() == sum(len(g) for g in gates)


def test_wire_format_contract():
    """m3_ep_send_map / m3_ep_recv_gate (oracle statements): headers carry the per-local-expert counts, every token gets a
    distinct wire row inside its owner's chunk in sorted order, dropped tokens get none; the receiver labels exactly the
    announced rows, in local-expert order, and nothing beyond the capacity whatever the header says."""
    world, e_loc, S, words = 4, 2, 23, 8
    rng = np.random.default_rng(5)
    gate = rng.integers(-1, world * e_loc, S).astype(np.int32)
    mapping, acc = moe_index_ref(gate, world * e_loc)
    cap = S
    map_send, hdr = ep_send_map_ref(gate, mapping, acc, world, e_loc, cap, words)
    live = gate >= 0
    assert (map_send[~live] == -1).all() and len(set(map_send[live])) == int(live.sum())
    for s_ in np.nonzero(live)[0]:
        j, row = divmod(int(map_send[s_]), cap + 1)
        assert j == gate[s_] // e_loc and row >= 1
    for j in range(world):
        for i in range(e_loc):
            assert hdr[j, 0, i] == np.sum(gate == j * e_loc + i)
    # a receiver that got chunk j from every rank r: here simply feed the headers back
    g = ep_recv_gate_ref(hdr, world, e_loc, cap).reshape(world, cap + 1)
    for j in range(world):
        assert g[j, 0] == -1
        want = np.repeat(np.arange(e_loc), hdr[j, 0, :e_loc])
        assert np.array_equal(g[j, 1:1 + len(want)], want) and (g[j, 1 + len(want):] == -1).all()
    bad = hdr.copy()
    bad[0, 0, 0] = 10 ** 6                       # a corrupt count must not label rows past the chunk
    g2 = ep_recv_gate_ref(bad, world, e_loc, cap).reshape(world, cap + 1)
    assert (g2[0, 1:] == 0).all() and g2.shape[1] == cap + 1
