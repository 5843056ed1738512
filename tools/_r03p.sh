cd $GRAFT_REPO_ROOT
for kv in 0 1; do
  echo "=== HIP_FORCE_DEV_KERNARG=$kv"
  HIP_FORCE_DEV_KERNARG=$kv M3ASR_LIB=$PWD/tools/_diag_gemm.so timeout -k 10 120 python tools/diag_gemm_f32.py 50 1024 512 ln cold 2>&1 | grep -v amdgpu.ids | head -4
done
for rep in 1 2; do for kv in 0 1; do
  HIP_FORCE_DEV_KERNARG=$kv python bench.py --steps 200 --warmup 20 --no-cpu-baseline > /tmp/n1_$kv.json 2> /tmp/n1_$kv.err
  python3 -c "import json; d=json.loads([l for l in open('/tmp/n1_$kv.json') if l.startswith('{')][-1]); print('DEV_KERNARG=$kv cfg1: value %.0f one-stream %.4f p50 %.4f' % (d['value'], d['config']['latency_ms_one_stream'], d['forward']['latency_ms']['p50']))"
done; done
