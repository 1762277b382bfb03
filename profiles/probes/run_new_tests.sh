cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 900 python -m pytest tests/test_gpu_se.py tests/test_gpu_pe.py tests/test_gpu_records.py tests/test_gpu_bam.py tests/test_gpu_bam2bam_cli.py tests/test_gpu_bench_modes.py -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/t1/pytest.log
NABWA_TIMING=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 > gpurun_out/t1/pe.json 2> gpurun_out/t1/pe.err; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/t1/pe.json'));print('pe', d['value'], d['config']['stage_ms'], d['config']['bit_exact_vs_cpu_sample'])"
grep "se_posn" gpurun_out/t1/pe.err | tail -1
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/t1/head.json 2> gpurun_out/t1/head.err; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/t1/head.json'));print('e2e', d['e2e']['reads_per_s'], d['e2e']['stage_ms'], d['e2e']['bit_exact_vs_reference_sample'])"
