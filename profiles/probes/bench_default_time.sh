cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/bd
s=$(date +%s); timeout -k 10 900 python3 bench.py > gpurun_out/bd/out.json 2> gpurun_out/bd/err.log; rc=$?; e=$(date +%s); echo "rc=$rc wall=$((e-s)) s"
python3 -c "import json;d=json.load(open('gpurun_out/bd/out.json'));print(d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['e2e']['reads_per_s'], d['config']['bit_exact_vs_cpu_sample'])"
