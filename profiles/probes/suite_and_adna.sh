cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/t2
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t2/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/t2/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 bench.py --adna --reads 6250000 --steps 2 --warmup 1 --no-e2e > gpurun_out/t2/adna_6m.json 2> gpurun_out/t2/adna_6m.log; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/t2/adna_6m.json'));print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'], d['config']['bit_exact_vs_cpu_sample'])"
