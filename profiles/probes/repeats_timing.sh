cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/rep_t
NABWA_TIMING=1 timeout -k 10 500 python3 bench.py --repeats --steps 1 --warmup 0 --no-cpu --no-e2e > gpurun_out/rep_t/out.json 2> gpurun_out/rep_t/err.log; echo rc=$?
grep "kernel D" gpurun_out/rep_t/err.log | head -4 | cut -c1-600
python3 -c "import json;d=json.load(open('gpurun_out/rep_t/out.json'));print(d['value'], d['roofline']['search_kernel_ms'], d['roofline']['deep_kernel_ms'])"
