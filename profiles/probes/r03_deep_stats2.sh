# statistics build of kernel D (no histogram) on the paired-end and the repeat-genome workloads
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/st
NABWA_TIMING=1 NABWA_BENCH_QUICK=1 timeout -k 10 400 python3 bench.py --pe --pairs 500000 --steps 1 --warmup 0 --no-cpu --extras off > gpurun_out/st/pe.json 2> gpurun_out/st/pe.err; echo pe rc=$?
grep -h "kernel D" gpurun_out/st/pe.err | tail -3 | cut -c1-900
NABWA_TIMING=1 NABWA_BENCH_QUICK=1 timeout -k 10 500 python3 bench.py --repeats --steps 1 --warmup 0 --no-cpu --no-e2e --extras off > gpurun_out/st/rep.json 2> gpurun_out/st/rep.err; echo rep rc=$?
grep -h "kernel D" gpurun_out/st/rep.err | tail -3 | cut -c1-900
NABWA_BENCH_QUICK=1 timeout -k 10 500 python3 bench.py --repeats --steps 2 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/st/rep2.json 2> gpurun_out/st/rep2.err
python3 -c "import json;d=json.load(open('gpurun_out/st/rep2.json'));print('repeats', d['value'], 'S', d['roofline']['search_kernel_ms'], 'D', d['roofline']['deep_kernel_ms'], d['config']['checksum'])"
