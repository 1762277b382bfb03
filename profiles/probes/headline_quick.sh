cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/hq
timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-e2e > gpurun_out/hq/out.json 2> gpurun_out/hq/err.log || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/hq/out.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], 'S', r['search_kernel_ms'], 'D', r['deep_kernel_ms'], 'W', r['width_kernel']['kernel_ms'], d['config']['second_pass_reads'], d['config']['checksum'])"
