#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02p21; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 < /dev/null; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for S in 65536 16384; do
  ( cd /tmp && EXP_DTYPE=fp8a8 EXP_NO_GRAPH=1 timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/mfma$S -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py $S > $O/mfma$S.log 2>&1 < /dev/null; echo "mfma$S rc=$?" )
done
