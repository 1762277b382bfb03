cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/hb
for tb in 400 700 1000 1400 2000; do
NABWA_TRIP_BUDGET=$tb timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-e2e > gpurun_out/hb/out_$tb.json 2> gpurun_out/hb/err_$tb.log || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/hb/out_$tb.json'));r=d['roofline'];print($tb, d['value'], d['ms_per_step'], r['search_kernel_ms'], r['deep_kernel_ms'], r['width_kernel']['kernel_ms'], d['config']['second_pass_reads'], d['config']['checksum'])"
done
