#!/bin/bash
# round-2 probe 3: fused bf16 expert FFN (parity + microbench), pmc experiment C (no rocBLAS in the process)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -s -k "fmoe_expert" > $O/pytest_fused.log 2>&1; echo "pytest fused rc=$?"; tail -8 $O/pytest_fused.log
for S in 16384 65536; do
  timeout -k 10 300 python tools/exp_expert_ffn.py $S > $O/exp_$S.json 2> $O/exp_$S.err; echo "exp $S rc=$?"; cat $O/exp_$S.json
  M3_EXPERT_FUSED_MIN_ROWS=100000000 timeout -k 10 300 python tools/exp_expert_ffn.py $S > $O/exp_${S}_tiled.json 2>> $O/exp_$S.err; cat $O/exp_${S}_tiled.json
done
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py 65536 > $GRAFT_REPO_ROOT/$O/kt.log 2>&1; echo "kt rc=$?" )
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {} | cut -c1-200'
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1 ; echo "pmc_fetch rc=$?" )
tail -3 $O/pmc_fetch.log
