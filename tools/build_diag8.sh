#!/bin/bash
# diagnostic library: the product objects + moe_expert_fused_fp8.hip rebuilt with -DM3_FUSED_DIAG (cycle stamps); not shipped
set -e
cd "$(dirname "$0")/../3m-asr-inference_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 -DM3_FUSED_DIAG -c csrc/moe_expert_fused_fp8.hip -o /tmp/diag8_fused.o
objs=$(ls build/*.o | grep -v moe_expert_fused_fp8.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/_diag8.so $objs /tmp/diag8_fused.o
