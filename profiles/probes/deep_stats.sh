cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/ds
for cfg in "48" "24"; do
export NABWA_DEEP_STAGE=$cfg
echo "== records per chain and round $cfg"
NABWA_TIMING=1 timeout -k 10 300 python3 bench.py --adna --reads 1000000 --steps 1 --warmup 0 --no-cpu --no-e2e > gpurun_out/ds/adna.json 2> gpurun_out/ds/adna.err || exit 1
grep "kernel D" gpurun_out/ds/adna.err | sed -n 3,4p | cut -c1-420
timeout -k 10 300 python3 bench.py --adna --reads 1000000 --steps 2 --warmup 1 --no-cpu --no-e2e > gpurun_out/ds/adna2.json 2> gpurun_out/ds/adna2.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/ds/adna2.json'));print('adna no-prof', d['value'], d['roofline']['deep_kernel_ms'], d['config']['checksum'])"
NABWA_TIMING=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 1 --warmup 0 --no-cpu > gpurun_out/ds/pe.json 2> gpurun_out/ds/pe.err || exit 1
grep "kernel D" gpurun_out/ds/pe.err | sed -n 1,2p | cut -c1-420
timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/ds/pe2.json 2> gpurun_out/ds/pe2.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/ds/pe2.json'));print('pe no-prof', d['value'], d['roofline']['deep_kernel_ms'], d['config']['checksum'])"
done
