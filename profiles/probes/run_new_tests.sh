cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 gpurun_out/cli && timeout -k 10 600 python -m pytest tests/test_gpu_bam2bam_cli.py -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -12 gpurun_out/t1/pytest.log | cut -c1-220
timeout -k 10 500 python3 profiles/probes/cli_rate.py 4000000 /tmp/cli_rate > gpurun_out/cli/rate.log 2>&1; echo rc=$?; cat gpurun_out/cli/rate.log | cut -c1-400
