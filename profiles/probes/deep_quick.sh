cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/hq
timeout -k 10 600 python -m pytest tests/test_gpu_deep.py tests/test_gpu_parity.py tests/test_gpu_bench_modes.py -x -q -m gpu > gpurun_out/hq/pytest.log 2>&1; echo rc=$?; tail -2 gpurun_out/hq/pytest.log
timeout -k 10 300 python3 bench.py --adna --reads 1000000 --steps 2 --warmup 1 --no-cpu --no-e2e > gpurun_out/hq/adna.json 2> gpurun_out/hq/adna.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/hq/adna.json'));print('adna 1M', d['value'], d['roofline']['search_kernel_ms'], d['roofline']['deep_kernel_ms'], d['config']['checksum'])"
