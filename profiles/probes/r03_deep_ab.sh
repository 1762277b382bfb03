# kernel D after a change: parity (deep tests, bench sample check) and the three deep workloads' kernel times, no profiler attached
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/dab
timeout -k 10 600 python -m pytest tests/test_gpu_deep.py tests/test_gpu_parity.py -x -q > gpurun_out/dab/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/dab/pytest.log
timeout -k 10 400 python3 bench.py --adna --reads 6250000 --steps 2 --warmup 1 --cpu-seconds 6 --no-e2e --extras off > gpurun_out/dab/adna.json 2> gpurun_out/dab/adna.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/dab/adna.json'));print('adna 6.25M', d['value'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], 'frac', d['roofline']['frac'], d['config']['bit_exact_vs_cpu_sample'], d['config']['checksum'])"
timeout -k 10 400 python3 bench.py --adna --reads 1000000 --steps 2 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/dab/adna1.json 2> gpurun_out/dab/adna1.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/dab/adna1.json'));print('adna 1M', d['value'], 'D ms', d['roofline']['deep_kernel_ms'], d['config']['checksum'])"
timeout -k 10 400 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --cpu-seconds 6 --extras off > gpurun_out/dab/pe.json 2> gpurun_out/dab/pe.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/dab/pe.json'));print('pe 1M', d['value'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['bit_exact_vs_cpu_sample'], d['config']['checksum'])"
timeout -k 10 400 python3 bench.py --repeats --steps 2 --warmup 1 --cpu-seconds 6 --no-e2e --extras off > gpurun_out/dab/rep.json 2> gpurun_out/dab/rep.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/dab/rep.json'));print('repeats 10M', d['value'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['bit_exact_vs_cpu_sample'], d['config']['checksum'])"
