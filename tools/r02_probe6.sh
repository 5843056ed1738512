#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "fmoe_expert" > $O/pytest_fused.log 2>&1; echo "pytest fused rc=$?"; tail -3 $O/pytest_fused.log
M3ASR_LIB=$PWD/tools/_diag_libm3asr.so timeout -k 10 200 python tools/diag_fused.py 65536 2>&1 | tail -4
M3ASR_LIB=$PWD/tools/_diag_libm3asr.so timeout -k 10 200 python tools/diag_fused.py 16384 2>&1 | tail -4
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1 ; echo "pmc rc=$?" )
grep -E "bench\[|Aborted" $O/pmc_fetch.log | cut -c1-200 | tail -12
