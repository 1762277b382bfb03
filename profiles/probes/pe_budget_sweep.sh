cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/pe_b
for tb in 600 1000 1400 2000 3000; do
NABWA_TRIP_BUDGET=$tb timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/pe_b/out_$tb.json 2> gpurun_out/pe_b/err_$tb.log || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/pe_b/out_$tb.json'));print($tb, d['value'], d['config']['stage_ms']['search (kernels W, S, D)'], d['roofline']['search_kernel_ms'], d['roofline']['deep_kernel_ms'], d['config']['second_pass_reads'], d['config']['checksum'])"
done
