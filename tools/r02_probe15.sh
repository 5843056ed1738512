#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p15; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_fp8_gpu.py -m gpu -x -q -s -k "fp8_arithmetic" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "fp8 arithmetic|vs the unq|passed|failed|Error" $O/pytest.log | head -20
M3ASR_LIB=$PWD/tools/_diag8.so timeout -k 5 200 python tools/diag_fused8.py 65536 2>&1 | tail -1
for S in 16384 65536; do
  EXP_DTYPE=fp8a8 timeout -k 10 300 python tools/exp_expert_ffn.py $S 2>/dev/null | tail -1
  ( cd /tmp && EXP_DTYPE=fp8a8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$S -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py $S > $GRAFT_REPO_ROOT/$O/kt$S.log 2>&1 )
  python3 - $(find $O/kt$S -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]: print(r["Name"][:48], r["Calls"], r["AverageNs"])
PY
done
