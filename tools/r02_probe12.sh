#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p12; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "linear or engine" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "packed" >> $O/pytest.log 2>&1; echo "pytest packed rc=$?"; tail -3 $O/pytest.log
for thin in 600 0; do
  M3_TILED_THIN_BELOW=$thin timeout -k 10 300 python bench.py --weight-dtype bf16 --batch 16 --varlen 50-500 --streams 4 --steps 80 --warmup 8 --no-cpu-baseline > $O/cfg3_thin$thin.json 2> $O/cfg3_thin$thin.err; echo "thin=$thin rc=$?"
  python3 -c "
import json;d=json.loads(open('$O/cfg3_thin$thin.json').read().strip().splitlines()[-1]);print('thin=$thin value',d['value'],'lat',d['config']['latency_ms_one_stream'],'p50',d['forward']['latency_ms']['p50'],d['roofline']['kernel'],d['roofline']['time_share'],d['roofline']['frac'])"
done
