#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02p22; mkdir -p $O
timeout -k 10 400 python bench.py --weight-dtype bf16 --batch 16 --varlen 50-500 --streams 4 --steps 60 --warmup 6 --no-cpu-baseline --profile-stages > $O/cfg3.json 2> $O/cfg3.err < /dev/null; echo "cfg3 rc=$?"; tail -c 600 $O/cfg3.json
timeout -k 10 400 python bench.py --weight-dtype fp8 --fp8-activations --batch 64 --varlen 50-500 --streams 2 --experts 64 --steps 40 --warmup 4 --no-cpu-baseline > $O/cfg5_fp8a8.json 2> $O/cfg5_fp8a8.err < /dev/null; echo "cfg5 fp8a8 rc=$?"; tail -c 300 $O/cfg5_fp8a8.err; cut -c1-700 $O/cfg5_fp8a8.json
timeout -k 10 400 python bench.py --weight-dtype fp8 --batch 64 --varlen 50-500 --streams 2 --experts 64 --steps 40 --warmup 4 --no-cpu-baseline > $O/cfg5_fp8.json 2> $O/cfg5_fp8.err < /dev/null; echo "cfg5 fp8 rc=$?"; cut -c1-300 $O/cfg5_fp8.json
