cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 900 python -m pytest tests/test_gpu_se.py tests/test_gpu_pe.py tests/test_gpu_records.py tests/test_gpu_bam.py tests/test_gpu_bam2bam_cli.py -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/t1/pytest.log
bash profiles/probes/e2e_timing.sh 2>&1 | grep "^e2e\|pass2" | tail -3
