# how many chain steps of kernel D stand on one-row intervals (what a text step could serve), per workload; statistics build (NABWA_TIMING)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/d1
export NABWA_BENCH_QUICK=1
NABWA_TIMING=1 timeout -k 10 400 python3 bench.py --adna --reads 2000000 --steps 1 --warmup 0 --no-cpu --no-e2e > gpurun_out/d1/adna.json 2> gpurun_out/d1/adna.err || exit 1
grep "kernel D" gpurun_out/d1/adna.err | cut -c1-520
NABWA_TIMING=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 1 --warmup 0 --no-cpu > gpurun_out/d1/pe.json 2> gpurun_out/d1/pe.err || exit 1
grep "kernel D" gpurun_out/d1/pe.err | cut -c1-520
NABWA_TIMING=1 timeout -k 10 400 python3 bench.py --repeats --reads 10000000 --steps 1 --warmup 0 --no-cpu --no-e2e > gpurun_out/d1/rep.json 2> gpurun_out/d1/rep.err || exit 1
grep "kernel D" gpurun_out/d1/rep.err | cut -c1-520
