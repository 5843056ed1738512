#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p17; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_fp8_gpu.py -m gpu -x -q -k "fp8_arithmetic" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for S in 16384 65536; do
  EXP_DTYPE=fp8a8 timeout -k 10 200 python tools/exp_expert_ffn.py $S 2>&1 | tail -4
  echo "== diag $S"; M3ASR_LIB=$PWD/tools/_diag8.so timeout -k 10 200 python tools/diag_fused8.py $S 2>&1 | tail -3
done
