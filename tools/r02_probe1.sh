#!/bin/bash
# round-2 probe 1: baseline, context sweep, per-stage times, and the --pmc abort experiment (one run per configuration)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p1; mkdir -p $O
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-stages > $O/s4.json 2> $O/s4.err && echo "s4 done" && tail -c 600 $O/s4.json
for s in 6 8; do
  python bench.py --steps 240 --warmup 24 --no-cpu-baseline --streams $s > $O/s$s.json 2> $O/s$s.err && echo "s$s done" && tail -c 300 $O/s$s.json
done
GPU_MAX_HW_QUEUES=16 python bench.py --steps 240 --warmup 24 --no-cpu-baseline --streams 12 > $O/s12q16.json 2> $O/s12q16.err && tail -c 300 $O/s12q16.json
# pmc abort experiment A: bench.py as is (override now skipped under the profiler), 1 stream
( cd /tmp && timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmcA -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmcA.log 2>&1 ; echo "pmcA rc=$?" ) 
tail -5 $O/pmcA.log
env | grep -i -E "rocp|preload" > $O/env.txt || true
