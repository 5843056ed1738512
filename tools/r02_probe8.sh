#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p8; mkdir -p $O
for v in DEPHASE DEPHASE_NO_MFMA; do
  echo "== $v"; M3ASR_LIB=$PWD/tools/_diag_$v.so timeout -k 10 200 python tools/diag_fused.py 65536 2>&1 | tail -2
done
echo "== DEPHASE 16384"; M3ASR_LIB=$PWD/tools/_diag_DEPHASE.so timeout -k 10 200 python tools/diag_fused.py 16384 2>&1 | tail -2
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1 ; echo "pmc write rc=$?" )
