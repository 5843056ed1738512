#!/bin/bash
# round-2 evidence run: full GPU test suite, the default bench line, rocprofv3 kernel stats of the same command and the
# two PMC passes over bench.py --pmc-safe; summaries are copied into profiles/ by the caller.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02prof; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python bench.py --steps 200 --warmup 20 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/kt.log 2>&1; echo "kernel-trace rc=$?" )
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r02_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_$C.log 2>&1 ; echo "pmc_$C rc=$?" )
done
python3 tools/pmc_summarize.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/r02_pmc_bench.json f32 1 206 18 32
rm -rf $O/kt/*/*kernel_trace.csv
