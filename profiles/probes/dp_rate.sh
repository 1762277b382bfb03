cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/dp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/dp/prof -o run --output-format csv -- python3 profiles/probes/dp_rate.py > gpurun_out/dp/dp_rate.json 2> gpurun_out/dp/dp_rate.err; echo rc=$?
cat gpurun_out/dp/dp_rate.json; tail -3 gpurun_out/dp/dp_rate.err
f=$(find gpurun_out/dp/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/dp/dp_kernel_stats.csv; head -8 gpurun_out/dp/dp_kernel_stats.csv
