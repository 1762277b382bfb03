cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/dv
for dv in 0 4 3; do
export NABWA_DIVERT_CLS=$dv
timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/dv/pe.json 2> gpurun_out/dv/pe.err || { tail -3 gpurun_out/dv/pe.err; exit 1; }
python3 -c "import json;d=json.load(open('gpurun_out/dv/pe.json'));print('divert $dv pe', d['value'], d['config']['stage_ms']['search (kernels W, S, D)'], d['roofline']['search_kernel_ms'], d['roofline']['deep_kernel_ms'], d['config']['second_pass_reads'], d['config']['checksum'])"
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-e2e > gpurun_out/dv/h.json 2> gpurun_out/dv/h.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/dv/h.json'));r=d['roofline'];print('divert $dv headline', d['value'], r['search_kernel_ms'], r['deep_kernel_ms'], d['config']['second_pass_reads'], d['config']['checksum'])"
done
