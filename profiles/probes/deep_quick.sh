cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/hq
timeout -k 10 600 python -m pytest tests/test_gpu_deep.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/hq/pytest.log 2>&1; echo rc=$?; tail -2 gpurun_out/hq/pytest.log | cut -c1-200
timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/hq/pe.json 2> gpurun_out/hq/pe.err || { tail -3 gpurun_out/hq/pe.err; exit 1; }
python3 -c "import json;d=json.load(open('gpurun_out/hq/pe.json'));print('pe', d['value'], d['roofline']['deep_kernel_ms'], d['roofline']['search_kernel_ms'], d['config']['checksum'])"
timeout -k 10 300 python3 bench.py --adna --reads 6250000 --steps 2 --warmup 1 --no-cpu --no-e2e > gpurun_out/hq/adna.json 2> gpurun_out/hq/adna.err || { tail -3 gpurun_out/hq/adna.err; exit 1; }
python3 -c "import json;d=json.load(open('gpurun_out/hq/adna.json'));print('adna 6M', d['value'], d['roofline']['deep_kernel_ms'], d['config']['checksum'], d['config']['instrumented_run_same_rows'], d['roofline']['bucket_touches_per_read'])"
