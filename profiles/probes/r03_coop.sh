# the wave-wide expansion of one-row chains (DF_COOP) at several thresholds (chains still running in the round), against the build without it
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/coop
bash profiles/probes/r03_variants.sh fw
for c in 0 2 4 8 64; do
  export NABWA_DEEP_COOP=$c
  NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --adna --reads 6250000 --steps 2 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/coop/adna_$c.json 2> gpurun_out/coop/adna_$c.err || exit 1
  NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu --extras off > gpurun_out/coop/pe_$c.json 2> gpurun_out/coop/pe_$c.err || exit 1
  python3 -c "
import json
a=json.load(open('gpurun_out/coop/adna_$c.json')); p=json.load(open('gpurun_out/coop/pe_$c.json'))
print('coop $c: adna D ms', a['roofline']['deep_kernel_ms'], a['config']['checksum'], '| pe D ms', p['roofline']['deep_kernel_ms'], p['config']['checksum'])"
done
