cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 800 python -m pytest tests/test_gpu_se.py tests/test_gpu_pe.py tests/test_gpu_records.py tests/test_gpu_bam.py -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -4 gpurun_out/t1/pytest.log | cut -c1-220
NABWA_TIMING=1 timeout -k 10 400 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 > gpurun_out/t1/pe.json 2> gpurun_out/t1/pe.err; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/t1/pe.json'));print('pe', d['value'], d['ms_per_step'], d['config']['stage_ms'], d['config']['bit_exact_vs_cpu_sample'])"
grep "pe_finish" gpurun_out/t1/pe.err | tail -2 | cut -c1-330
