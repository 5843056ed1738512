"""Expert-parallel exchange on CPU with the gloo backend, world_size 2 (the N>1 path of §⑤): every rank routes its own
tokens over experts sharded 4 per rank; the result must equal the single-process result with all experts local.
Compute steps use the CPU oracle as the backend (the product backend is HipBackend = libm3asr_hip.so); what is under
test here is the host logic: counts, split sizes, wire order, reassembly."""
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
from oracle.moe_index import moe_index_ref, local_scatter_ref, ep_exchange_counts_ref


class OracleBackend:
    def index(self, gate_idx, n_total):
        m, a = moe_index_ref(gate_idx.numpy(), n_total)
        return torch.from_numpy(m), torch.from_numpy(a), None

    def scatter(self, x, mapping, n_rows):
        return torch.from_numpy(local_scatter_ref(x.numpy(), mapping.numpy(), n_rows))

    def expert_ffn(self, rows, gate_local, w):
        y, _, _ = fmoe_expert(rows.unsqueeze(0), gate_local.view(1, -1, 1), w["w1"], w["b1"], w["w2"], w["b2"])
        return y[0]

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
