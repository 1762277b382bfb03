cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 600 python -m pytest tests/test_gpu_bam2bam_cli.py tests/test_gpu_bam.py -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -4 gpurun_out/t1/pytest.log | cut -c1-220
NABWA_TIMING=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/t1/head.json 2> gpurun_out/t1/head.err; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/t1/head.json'));print('e2e', d['e2e']['reads_per_s'], d['e2e']['first_batch_reads_per_s'], d['e2e']['stage_ms'])"
grep -n "se_posn\|bam_batch\|se_refine" gpurun_out/t1/head.err | tail -6 | cut -c1-300
