#!/bin/bash
# round-2 probe 4: why is the fused expert kernel slow -- memory-side counters
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02p4; mkdir -p $O
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $C | cut -d' ' -f1)
  ( cd /tmp && EXP_NO_GRAPH=1 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/$O/$tag -- python3 $GRAFT_REPO_ROOT/tools/exp_expert_ffn.py 65536 > $GRAFT_REPO_ROOT/$O/$tag.log 2>&1; echo "$tag rc=$?" )
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r02p4/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fused" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, "n=%d mean=%.4g" % (len(v), sum(v) / len(v)))
PY
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1 ; echo "pmc_fetch rc=$?" )
grep -E "bench\[|Aborted|value" $O/pmc_fetch.log | cut -c1-300 | tail -12
