#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for hq in 4 8 16 24; do
  for st in 4 5 6 8; do
    GPU_MAX_HW_QUEUES=$hq timeout -k 10 300 python bench.py --no-cpu-baseline --steps 240 --warmup 24 --streams $st --latency-iters 50 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq=$hq streams=$st value %.0f ms/step %.3f'%(d['value'],d['ms_per_step']))"
  done
done
