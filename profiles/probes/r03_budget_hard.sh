# hand-over budget by class: reads without an exact occurrence on either strand go to kernel D after NABWA_TRIP_BUDGET_HARD trips, the others after the usual 2000
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/wb5
for h in 2000 300 150; do
  export NABWA_TRIP_BUDGET_HARD=$h
  NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/wb5/head_$h.json 2> gpurun_out/wb5/head_$h.err || exit 1
  NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu --extras off > gpurun_out/wb5/pe_$h.json 2> gpurun_out/wb5/pe_$h.err || exit 1
  NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --repeats --steps 2 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/wb5/rep_$h.json 2> gpurun_out/wb5/rep_$h.err || exit 1
  python3 -c "
import json
for w in ('head','pe','rep'):
    d=json.load(open('gpurun_out/wb5/%s_$h.json' % w)); r=d['roofline']
    print('hard $h:', w, 'value', d['value'], 'ms/step', d['ms_per_step'], 'W', r.get('width_kernel_ms', r.get('width_kernel',{}).get('kernel_ms')), 'S', r['search_kernel_ms'], 'D', r['deep_kernel_ms'], d['config']['checksum'])"
done
