cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 900 python -m pytest tests/test_gpu_se.py tests/test_gpu_pe.py tests/test_gpu_records.py tests/test_gpu_bam.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/t1/pytest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/dp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/dp/prof -o run --output-format csv -- python3 profiles/probes/dp_rate.py > gpurun_out/dp/dp_rate.json 2> gpurun_out/dp/dp_rate.err; echo rc=$?
cat gpurun_out/dp/dp_rate.json
f=$(find gpurun_out/dp/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/dp/dp_kernel_stats.csv; head -5 gpurun_out/dp/dp_kernel_stats.csv
