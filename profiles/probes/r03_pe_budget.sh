# kernel S's hand-over budget on the paired-end workload after kernel D's wave-wide chains
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/wb3
for b in 300 600 1000 1500 2000; do
  NABWA_TRIP_BUDGET=$b NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu --extras off > gpurun_out/wb3/pe_b$b.json 2> gpurun_out/wb3/pe_b$b.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/wb3/pe_b$b.json'));print('budget $b: pe 1M value', d['value'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['checksum'])"
done
