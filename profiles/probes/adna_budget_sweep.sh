cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/ab2
for tb in 300 1000 2000 5000; do
NABWA_TRIP_BUDGET=$tb timeout -k 10 300 python3 bench.py --adna --reads 6250000 --steps 2 --warmup 1 --no-cpu --no-e2e > gpurun_out/ab2/out_$tb.json 2> gpurun_out/ab2/err_$tb.log || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/ab2/out_$tb.json'));r=d['roofline'];print($tb, d['value'], d['ms_per_step'], 'S', r['search_kernel_ms'], 'D', r['deep_kernel_ms'], d['config']['second_pass_reads'], d['config']['checksum'])"
done
