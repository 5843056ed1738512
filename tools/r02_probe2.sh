#!/bin/bash
# round-2 probe 2: full GPU test suite, default bench, and the --pmc abort experiment B (4 host threads, 1 context)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p2; mkdir -p $O
python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err && echo "bench done" && tail -c 2500 $O/bench.json
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1 ; echo "pmc_fetch rc=$?" )
tail -3 $O/pmc_fetch.log
timeout -k 10 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
