#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02p28; mkdir -p $O
timeout -k 10 400 python bench.py --weight-dtype fp8 --fp8-activations --batch 64 --varlen 50-500 --streams 2 --experts 64 --steps 30 --warmup 4 --no-cpu-baseline --profile-stages > $O/cfg5.json 2> $O/cfg5.err < /dev/null; echo "rc=$?"
grep -E "^(lens|pack_plan|embed.subsample|subsample|blocks.9\.|logits|unpack|sum of)" $O/cfg5.err | head -40
python - <<EOF
import json
d=json.load(open("$O/cfg5.json")); print(d["value"], d["ms_per_step"], d["forward"]["latency_ms"]["p50"]); 
for k,v in sorted(d["roofline"]["families"].items(), key=lambda kv:-kv[1]["time_share"])[:10]: print("  %-42s %.3f %d"%(k,v["time_share"],v["launches"]))
EOF
