cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 600 python -m pytest tests/test_gpu_se.py tests/test_gpu_pe.py tests/test_gpu_records.py -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -15 gpurun_out/t1/pytest.log | cut -c1-250
bash profiles/probes/pe_timing2.sh
