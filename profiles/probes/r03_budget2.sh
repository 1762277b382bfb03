# kernel S's hand-over budget on the headline, paired-end and repeat-genome workloads after this round's kernel D
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/wb4
for b in 100 200 300; do
  NABWA_TRIP_BUDGET=$b NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu --extras off > gpurun_out/wb4/pe_b$b.json 2> gpurun_out/wb4/pe_b$b.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/wb4/pe_b$b.json'));print('budget $b: pe 1M value', d['value'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['checksum'])"
done
for b in 300 600 1000 2000; do
  NABWA_TRIP_BUDGET=$b NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/wb4/head_b$b.json 2> gpurun_out/wb4/head_b$b.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/wb4/head_b$b.json'));print('budget $b: headline value', d['value'], 'ms/step', d['ms_per_step'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['checksum'])"
done
for b in 300 1000 2000; do
  NABWA_TRIP_BUDGET=$b NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --repeats --steps 2 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/wb4/rep_b$b.json 2> gpurun_out/wb4/rep_b$b.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/wb4/rep_b$b.json'));print('budget $b: repeats value', d['value'], 'ms/step', d['ms_per_step'], 'D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['checksum'])"
done
