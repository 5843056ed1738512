"""CPU oracle for the 3M-ASR Conformer-MoE encoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the CPU baseline.
The product path (``3m-asr-inference_amd/``) never imports this package and
fails loudly when the HIP library is missing.

Contents
  encoder_ref.py   plain-torch fp32 restatement of the encoder forward
                   (follows /root/reference/trainer_3m_fix model/ + layer/ files,
                   cited per function)
  moe_index.py     numpy restatement of the MoE indexing contract (integer, bit-exact)
  moe_index.c      the same contract in plain C (built by __graft_entry__.build())
  gen_golden.py    container-only: drives the reference's own Net.forward through a
                   torch-eager helper and writes tests/golden/*.npz

Pinning: the reference ships no tests, fixtures or golden vectors (SURVEY.md §4,
§8c) and its CUDA/TensorRT plugins cannot be built here.  The restatement is
pinned against outputs of the reference's *own Python forward call graph*
executed in the build container (gen_golden.py); the fixtures are committed
under tests/golden/.  Op-level arithmetic of the closed TensorRT/cuDNN/cuBLAS
layers is unpinned by the reference and follows PyTorch semantics, which is the
reference's own stated parity target (infer_helper.py:93).
"""
