cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/ab
export NABWA_LIB=$PWD/gpurun_ab/libnabwa_w4.so
timeout -k 10 300 python3 bench.py --adna --reads 1000000 --steps 2 --warmup 1 --no-cpu --no-e2e > gpurun_out/ab/adna.json 2> gpurun_out/ab/adna.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/ab/adna.json'));print('adna w4', d['value'], d['roofline']['deep_kernel_ms'], d['config']['checksum'])"
timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/ab/pe.json 2> gpurun_out/ab/pe.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/ab/pe.json'));print('pe w4', d['value'], d['roofline']['deep_kernel_ms'], d['config']['checksum'])"
