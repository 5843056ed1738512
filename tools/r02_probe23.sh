#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02p23; mkdir -p $O
for v in 512 2048; do
  M3_GATE_INDEX_FUSED_MAX=$v timeout -k 10 400 python bench.py --weight-dtype bf16 --batch 16 --varlen 50-500 --streams 4 --steps 60 --warmup 6 --no-cpu-baseline > $O/cfg3_$v.json 2> $O/cfg3_$v.err < /dev/null; echo "fused_max=$v rc=$?"
  python - <<EOF
import json
d=json.load(open("$O/cfg3_$v.json")); print("value %.0f  ms/step %.3f  latency p50 %.3f  kernels %d"%(d["value"], d["ms_per_step"], d["forward"]["latency_ms"]["p50"], d["config"]["kernels_per_forward"]))
EOF
done
M3_GATE_INDEX_FUSED_MAX=2048 timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -3
