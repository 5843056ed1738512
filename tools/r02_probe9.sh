#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p9; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "fmoe_expert" > $O/pytest_fused.log 2>&1; echo "pytest fused rc=$?"; tail -3 $O/pytest_fused.log
for v in BASE DEPHASE NO_MFMA NO_FILL; do
  echo "== $v"; M3ASR_LIB=$PWD/tools/_diag_$v.so timeout -k 10 200 python tools/diag_fused.py 65536 2>&1 | tail -2
done
echo "== BASE 16384"; M3ASR_LIB=$PWD/tools/_diag_BASE.so timeout -k 10 200 python tools/diag_fused.py 16384 2>&1 | tail -2
