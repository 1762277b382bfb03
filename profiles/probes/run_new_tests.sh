cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 600 python -m pytest tests/test_gpu_se.py -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -2 gpurun_out/t1/pytest.log
