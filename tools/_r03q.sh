cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
for args in "50 1024 512 ln cold" "50 512 1024 cold" "50 512 512 cold" "50 1536 512 ln cold"; do
  M3ASR_LIB=$PWD/tools/_diag_gemm.so timeout -k 10 120 python tools/diag_gemm_f32.py $args 2>&1 | grep -v amdgpu.ids
done
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/n1.json 2> $O/n1.err
  python3 -c "import json; d=json.loads([l for l in open('$O/n1.json') if l.startswith('{')][-1]); print('cfg1: value %.0f one-stream %.4f p50 %.4f' % (d['value'], d['config']['latency_ms_one_stream'], d['forward']['latency_ms']['p50']))"
done
