#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02p26; mkdir -p $O
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/kt.log 2>&1 < /dev/null; echo "kt rc=$?" )
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -4 "$f" | cut -c1-60,200-330; fi
tail -1 $O/kt.log | cut -c1-200
rm -f $O/kt/*kernel_trace.csv $O/kt/*/*kernel_trace.csv
