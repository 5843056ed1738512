#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p16; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_network_helper_gpu.py -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "fp8 arithmetic engine|passed|failed|Error|error" $O/pytest.log | head -20; tail -5 $O/pytest.log
