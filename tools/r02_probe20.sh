#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for v in _diag8 _diag8_NO_SILU _diag8_NO_STAGE _diag8_NO_XPF _diag8_NO_READS _diag8_NO_MFMA; do
  echo "== $v"; M3ASR_LIB=$PWD/tools/$v.so timeout -k 10 200 python tools/diag_fused8.py 65536 2>&1 < /dev/null | tail -2
done
