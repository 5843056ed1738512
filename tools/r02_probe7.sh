#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p7; mkdir -p $O
for v in libm3asr NO_READS NO_MFMA NO_FILL ASM; do
  echo "== $v"; M3ASR_LIB=$PWD/tools/_diag_$v.so timeout -k 10 200 python tools/diag_fused.py 65536 2>&1 | tail -2
done
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1 ; echo "pmc rc=$?" )
grep -E "bench\[|Aborted|\"value\"" $O/pmc_fetch.log | cut -c1-160 | tail -14
