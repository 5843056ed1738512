#!/usr/bin/env python3
"""Golden vectors for the static chunk mask (BUILD CONTAINER ONLY -- needs /root/reference, never runs on the GPU box).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Imports the reference's own ``utils/mask.py`` (plain torch, importable as it lies) and records what its
``subsequent_chunk_mask`` (:42-75) and ``add_optional_chunk_mask`` (:80-145, static branch :127-134 and the fixed-size
decoding branch :112-126) return for a list of (size, chunk, left) cases and (lengths, chunk, left) batches.  The
fixture holds only parameters and bit-packed boolean masks -- data, no reference source.

Usage:  python oracle/gen_golden_chunk.py [--check]
"""
import argparse
import importlib.util
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MASK = "/root/reference/trainer_3m_fix/utils/mask.py"
OUT = os.path.join(ROOT, "tests", "golden", "chunk_mask.npz")

# (size, chunk_size, num_left_chunks): the docstring example, chunk == 1, chunk >= size, ragged last chunk, left = 0 / 1 / 3 / all,
# and the BASELINE sequence length after subsampling (T' = 124) with the chunk sizes the streaming recipes use (16 / 25)
SQUARE = [(4, 2, -1), (1, 1, -1), (7, 1, 0), (7, 1, 2), (8, 8, -1), (8, 16, 0), (13, 4, -1), (13, 4, 0), (13, 4, 1), (13, 4, 3),
          (50, 16, -1), (50, 16, 1), (51, 25, 0), (124, 16, -1), (124, 16, 2), (124, 25, 1), (124, 1, 0), (124, 124, -1), (61, 5, 4)]
# (lengths, static_chunk_size, num_decoding_left_chunks): padding mask (B, 1, L) & chunk mask -> (B, L, L)
BATCH = [((13, 9, 1), 4, 1), ((50, 37), 16, -1), ((124, 124, 100, 3), 16, 2), ((31, 17, 31), 25, 0), ((12, 7), 0, -1)]


def load_reference_mask():
    spec = importlib.util.spec_from_file_location("ref_utils_mask", REF_MASK)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build():
    ref = load_reference_mask()
    out = {"square_cases": np.asarray(SQUARE, np.int32), "n_batch": np.asarray(len(BATCH), np.int32)}
    for n, (size, chunk, left) in enumerate(SQUARE):
        m = ref.subsequent_chunk_mask(size, chunk, left)
        assert m.shape == (size, size) and m.dtype == torch.bool
        out["square_%d" % n] = np.packbits(m.numpy().reshape(-1))
    for n, (lens, chunk, left) in enumerate(BATCH):
        lens_t = torch.tensor(lens)
        L = int(lens_t.max())
        xs = torch.zeros(len(lens), L, 8)
        masks = (torch.arange(L).view(1, -1) < lens_t.view(-1, 1)).unsqueeze(1)          # (B, 1, L), ~make_pad_mask
        static = ref.add_optional_chunk_mask(xs, masks, False, False, 0, chunk, left)
        if static.shape[1] == 1:
            static = static.expand(-1, L, -1)
        out["batch_%d_lens" % n] = np.asarray(lens, np.int32)
        out["batch_%d_params" % n] = np.asarray([chunk, left], np.int32)
        out["batch_%d" % n] = np.packbits(static.contiguous().numpy().reshape(-1))
        if chunk > 0:      # the decoding branch with a fixed chunk size gives the same mask (use_dynamic_chunk, decoding_chunk_size > 0)
            dyn = ref.add_optional_chunk_mask(xs, masks, True, False, chunk, 0, left)
            assert torch.equal(dyn, static)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="compare with the committed fixture instead of writing it")
    a = ap.parse_args()
    got = build()
    if a.check:
        have = np.load(OUT)
        assert sorted(have.files) == sorted(got), "fixture keys differ"
        for k in got:
            assert np.array_equal(have[k], got[k]), k
        print("chunk_mask.npz matches the reference (%d arrays)" % len(got))
    else:
        np.savez_compressed(OUT, **got)
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
