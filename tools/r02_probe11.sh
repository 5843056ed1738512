#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p11; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ep_gpu.py -m gpu -x -q > $O/pytest_ep.log 2>&1; echo "pytest ep rc=$?"; tail -12 $O/pytest_ep.log
timeout -k 10 400 python bench.py --ep --weight-dtype bf16 --batch 16 --varlen 50-500 --steps 20 --warmup 3 > $O/ep1.json 2> $O/ep1.err; echo "ep world1 rc=$?"; tail -c 1500 $O/ep1.json; tail -3 $O/ep1.err
