#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_full_size_gpu.py tests/test_ep_gpu.py tests/test_network_helper_gpu.py -m gpu -x -q 2>&1 < /dev/null | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline_expert']; print('value %.0f ms/step %.4f latency %.3f expert avg us %.2f frac %.3f'%(d['value'],d['ms_per_step'],d['forward']['latency_ms']['p50'],e['avg_launch_us'],e['frac']))"
