cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -4 gpurun_out/t1/pytest.log
