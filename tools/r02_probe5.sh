#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -s -k "fmoe_expert" > $O/pytest_fused.log 2>&1; echo "pytest fused rc=$?"; tail -4 $O/pytest_fused.log
for S in 16384 65536; do
  timeout -k 10 300 python tools/exp_expert_ffn.py $S > $O/exp_$S.json 2> $O/exp_$S.err; echo "exp $S rc=$?"; cat $O/exp_$S.json
done
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py 65536 > $GRAFT_REPO_ROOT/$O/kt.log 2>&1; echo "kt rc=$?" )
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -4 {} | cut -c1-120,300-420'
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt16 -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py 16384 > $GRAFT_REPO_ROOT/$O/kt16.log 2>&1; echo "kt16 rc=$?" )
find $O/kt16 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -4 {} | cut -c1-120,300-420'
for C in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_$C.log 2>&1 ; echo "pmc_$C rc=$?" )
grep -E "bench\[|Aborted" $O/pmc_$C.log | cut -c1-200 | tail -8
done
