#!/bin/bash
# final check of the round: build entry point, smoke, full GPU suite, default bench
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; echo "bench rc=$?"; cut -c1-260 $O/bench.json
