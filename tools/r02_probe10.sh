#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p10; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > $O/pytest_bf16.log 2>&1; echo "pytest bf16 rc=$?"; tail -3 $O/pytest_bf16.log
for S in 16384 65536; do
  timeout -k 10 300 python tools/exp_expert_ffn.py $S > $O/exp_$S.json 2> $O/exp_$S.err; cat $O/exp_$S.json
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$S -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py $S > $GRAFT_REPO_ROOT/$O/kt$S.log 2>&1; echo "kt$S rc=$?" )
  find $O/kt$S -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -4 {} | cut -c1-60,290-400'
  ( cd /tmp && EXP_NO_GRAPH=1 timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O/mfma$S -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py $S > $GRAFT_REPO_ROOT/$O/mfma$S.log 2>&1; echo "mfma$S rc=$?" )
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r02p10/mfma*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fused" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in acc.items()}
    if m:
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
        print(d, {k: "%.4g" % v for k, v in m.items()}, "MFMA util %.3f" % util)
PY
