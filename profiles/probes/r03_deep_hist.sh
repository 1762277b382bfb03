# statistics build of kernel D on the three deep workloads: lane-steps, expansions by form and depth, records, children
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/hist
export NABWA_TIMING=1 NABWA_DEEP_HIST=1 NABWA_BENCH_QUICK=1
timeout -k 10 400 python3 bench.py --adna --reads 1000000 --steps 1 --warmup 0 --no-cpu --no-e2e --extras off > gpurun_out/hist/adna.json 2> gpurun_out/hist/adna.err; echo adna rc=$?
timeout -k 10 400 python3 bench.py --pe --pairs 500000 --steps 1 --warmup 0 --no-cpu --extras off > gpurun_out/hist/pe.json 2> gpurun_out/hist/pe.err; echo pe rc=$?
grep -h "kernel D" gpurun_out/hist/adna.err | head -8
grep -h "kernel D" gpurun_out/hist/pe.err | head -12
