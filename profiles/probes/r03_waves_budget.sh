# kernel D at 3 / 4 / 5 waves per SIMD, and kernel S's hand-over budget on the ancient-DNA workload, after this round's changes to D
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/wb2
bash profiles/probes/r03_variants.sh w3 w5 cur
for b in 100 300 1000; do
  NABWA_TRIP_BUDGET=$b NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --adna --reads 6250000 --steps 2 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/wb2/adna_b$b.json 2> gpurun_out/wb2/adna_b$b.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/wb2/adna_b$b.json'));print('budget $b: adna 6.25M value', d['value'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['checksum'])"
done
for b in 1000 2000 4000; do
  NABWA_TRIP_BUDGET=$b NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu --extras off > gpurun_out/wb2/pe_b$b.json 2> gpurun_out/wb2/pe_b$b.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/wb2/pe_b$b.json'));print('budget $b: pe 1M value', d['value'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['checksum'])"
done
