#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for v in OLDMAX UNSPLIT; do
  echo "== $v"; M3ASR_LIB=$PWD/tools/_diag_$v.so timeout -k 10 300 python -m pytest tests/test_fp8_gpu.py -m gpu -x -q -k "fp8_arithmetic and 16384" 2>&1 | grep -E "S=16384|passed|failed" | cut -c1-220
done
