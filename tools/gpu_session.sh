#!/bin/bash
# One parameterised script for the GPU box (replaces the per-experiment probe scripts of round 2):
#   gpurun -- 'bash tools/gpu_session.sh <out-subdir> <step> [<step> ...]'
# Steps write under gpurun_out/<out-subdir>/; the summaries worth keeping are copied into profiles/ by hand afterwards.
#   tests [pytest args]   not a step list: everything after `tests` goes to pytest (-m gpu); must be the last step
#   bench                 default bench line (configs[1])                           -> bench_n1.json
#   bench_cfg3            configs[2]  (bf16, 16 x U[50,500])                        -> bench_cfg3_bf16.json
#   bench_cfg5            one GPU's share of configs[4], fp8 arithmetic, B = 64     -> bench_cfg5share_fp8a8.json
#   bench_ep              world-1 rehearsal of the expert-parallel path (configs[3] shape per GPU) -> bench_ep_world1.json
#   kt                    rocprofv3 --kernel-trace --stats over the default bench   -> kernel_stats.csv
#   kt_cfg3 / kt_cfg5     the same over the configs[2] / configs[4]-share runs
#   pmc                   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py --pmc-safe + summary -> pmc_bench.json
#   calib                 FETCH_SIZE calibration microbenchmark under --pmc         -> fetch_calib.json
#   stages                bench.py --profile-stages (per-stage HIP-event times) for configs[1], [2], [4]-share
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p "$O"
R=$GRAFT_REPO_ROOT
CFG3="--weight-dtype bf16 --batch 16 --varlen 50-500 --streams 4"
CFG5="--weight-dtype fp8 --fp8-activations --experts 64 --batch 64 --varlen 50-500 --streams 4"
while [ $# -gt 0 ]; do
  step=$1; shift
  echo "== $step ($(date +%T))"
  case $step in
    tests)
      timeout -k 10 1700 python -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
      exit $rc ;;
    bench)      python bench.py --steps 200 --warmup 20 > $O/bench_n1.json 2> $O/bench_n1.err; echo "rc=$?"; tail -c 600 $O/bench_n1.json ;;
    bench_cfg3) python bench.py $CFG3 --steps 100 --warmup 10 > $O/bench_cfg3_bf16.json 2> $O/bench_cfg3.err; echo "rc=$?"; tail -c 400 $O/bench_cfg3_bf16.json ;;
    bench_cfg5) python bench.py $CFG5 --steps 60 --warmup 6 > $O/bench_cfg5share_fp8a8.json 2> $O/bench_cfg5.err; echo "rc=$?"; tail -c 400 $O/bench_cfg5share_fp8a8.json ;;
    bench_ep)
      python bench.py --ep --weight-dtype bf16 --batch 16 --varlen 50-500 --steps 60 --warmup 6 > $O/bench_ep_world1.json 2> $O/bench_ep.err; echo "rc=$?"; tail -c 700 $O/bench_ep_world1.json
      python bench.py --ep --no-graph --weight-dtype bf16 --batch 16 --varlen 50-500 --steps 60 --warmup 6 > $O/bench_ep_world1_eager.json 2>> $O/bench_ep.err; echo "rc=$?"
      python bench.py --weight-dtype bf16 --batch 16 --varlen 50-500 --streams 1 --steps 60 --warmup 6 --no-cpu-baseline > $O/bench_cfg3_one_stream.json 2>> $O/bench_ep.err; echo "rc=$?" ;;
    kt|kt_cfg3|kt_cfg5)
      case $step in kt) A="--steps 200 --warmup 20";; kt_cfg3) A="$CFG3 --steps 60 --warmup 6";; kt_cfg5) A="$CFG5 --steps 40 --warmup 4";; esac
      ( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$step -- python3 $R/bench.py $A --no-cpu-baseline > $R/$O/$step.json 2> $R/$O/$step.err; echo "rc=$?" )
      f=$(find $O/$step -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${step}_kernel_stats.csv && head -12 $O/${step}_kernel_stats.csv
      find $O/$step -name "*kernel_trace.csv" -delete ;;
    pmc)
      for C in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmc_$C -- python3 $R/bench.py --pmc-safe --steps 6 --warmup 1 --no-cpu-baseline > $R/$O/pmc_$C.json 2> $R/$O/pmc_$C.err; echo "pmc_$C rc=$?" )
      done
      python3 tools/pmc_summarize.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_bench.json f32 1 206 18 32 --bench-line $O/pmc_FETCH_SIZE.json
      find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*.csv" -size +20M -delete ;;
    calib)
      ( cd tools/ubench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib 2> /dev/null )
      ( cd /tmp && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/calib -- $R/tools/ubench/fetch_calib > $R/$O/calib.log 2>&1; echo "rc=$?" )
      python3 - $O <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(os.path.join(O, "calib", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
region = 512 << 20
out = {k: {"launches": len(v), "fetch_size_kb_mean": sum(v) / len(v), "fetch_size_over_bytes_read": sum(v) / len(v) * 1024 / region} for k, v in acc.items()}
json.dump({"bytes_read_per_launch": region, "kernels": out}, open(os.path.join(O, "fetch_calib.json"), "w"), indent=1)
for k, v in out.items():
    print("%-28s FETCH_SIZE x 1024 / bytes = %.3f" % (k, v["fetch_size_over_bytes_read"]))
PY
      ;;
    stages)
      python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-stages > $O/stages_cfg1.json 2> $O/stages_cfg1.txt; echo "rc=$?"
      python bench.py $CFG3 --steps 20 --warmup 3 --no-cpu-baseline --profile-stages > $O/stages_cfg3.json 2> $O/stages_cfg3.txt; echo "rc=$?"
      python bench.py $CFG5 --steps 10 --warmup 2 --no-cpu-baseline --profile-stages > $O/stages_cfg5.json 2> $O/stages_cfg5.txt; echo "rc=$?" ;;
    ab_headline)
      # A/B of the round-3 headline levers at configs[1], all in this one call on this one device: forked embed branch on / off,
      # non-temporal weight loads on / off (second library built here with -DM3_NT_WEIGHTS=0)
      ( cd 3m-asr-inference_amd && make -j16 EXTRA=-DM3_NT_WEIGHTS=0 OBJDIR=build_nt0 LIB=../tools/_ab_nt0.so > /dev/null 2>&1; echo "nt0 build rc=$?" )
      for rep in 1 2; do
        for v in base fork_off nt0 nt0_fork_off; do
          case $v in base) E=""; A="";; fork_off) E=""; A="--fork-embed off";; nt0) E="M3ASR_LIB=$R/tools/_ab_nt0.so"; A="";; nt0_fork_off) E="M3ASR_LIB=$R/tools/_ab_nt0.so"; A="--fork-embed off";; esac
          env $E python bench.py --steps 200 --warmup 20 --no-cpu-baseline $A > $O/ab_${v}_$rep.json 2> $O/ab_${v}_$rep.err
          python3 -c "import json,sys; d=json.loads([l for l in open('$O/ab_${v}_$rep.json') if l.startswith('{')][-1]); print('$v rep $rep: value %.0f  one-stream %.4f ms  p50 %.4f' % (d['value'], d['config']['latency_ms_one_stream'], d['forward']['latency_ms']['p50']))"
        done
      done ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
