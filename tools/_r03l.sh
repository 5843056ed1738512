cd $GRAFT_REPO_ROOT; O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/pytest_k.log 2>&1; echo "pytest attention rc=$?"; tail -5 $O/pytest_k.log
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py tests/test_engine_gpu.py tests/test_full_size_gpu.py -x -q > $O/pytest_engine.log 2>&1; echo "pytest engine rc=$?"; tail -5 $O/pytest_engine.log
python bench.py --weight-dtype fp8 --fp8-activations --experts 64 --batch 64 --varlen 50-500 --streams 2 --steps 20 --warmup 3 --no-cpu-baseline --profile-stages > $O/stages_cfg5.json 2> $O/stages_cfg5.txt; grep "blocks.9\.\|embed.blocks.0.att" $O/stages_cfg5.txt
python3 -c "import json; d=json.loads([l for l in open('$O/stages_cfg5.json') if l.startswith('{')][-1]); print('cfg5: value %.0f one-stream %.4f' % (d['value'], d['config']['latency_ms_one_stream']))"
python bench.py --weight-dtype bf16 --batch 16 --varlen 50-500 --streams 4 --steps 40 --warmup 5 --profile-stages > $O/stages_cfg3.json 2> $O/stages_cfg3.txt; grep "blocks.9\.att\|embed.blocks.0.att" $O/stages_cfg3.txt
python3 -c "import json; d=json.loads([l for l in open('$O/stages_cfg3.json') if l.startswith('{')][-1]); print('cfg3: value %.0f one-stream %.4f' % (d['value'], d['config']['latency_ms_one_stream']), d['cpu_baseline'])"
