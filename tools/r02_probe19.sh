#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02p19; mkdir -p $O
for S in 16384 65536; do
  cd /tmp && EXP_DTYPE=fp8a8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$S -o t -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py $S > $O/run_$S.log 2>&1 < /dev/null; echo "rc=$?"; tail -1 $O/run_$S.log
  cd $GRAFT_REPO_ROOT
  f=$(find $O/prof_$S -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -8 "$f" | cut -c1-160; else echo "no stats file"; ls -R $O/prof_$S | head; fi
done
