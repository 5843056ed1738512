#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for v in 0 300 600; do
  M3_TILED_THIN_BELOW=$v timeout -k 10 400 python bench.py --weight-dtype bf16 --batch 16 --varlen 50-500 --streams 4 --steps 80 --warmup 8 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('thin_below=$v value %.0f ms/step %.3f latency p50 %.3f'%(d['value'],d['ms_per_step'],d['forward']['latency_ms']['p50']))"
done
