cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1 && timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t1/pytest.log 2>&1; echo rc=$?; tail -4 gpurun_out/t1/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/t1/bench.json 2> gpurun_out/t1/bench.err; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/t1/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['e2e']['reads_per_s'], d['config']['bit_exact_vs_cpu_sample'], d['config']['instrumented_run_same_rows'])"
