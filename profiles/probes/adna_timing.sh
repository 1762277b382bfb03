cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/adna_t
NABWA_TIMING=1 timeout -k 10 500 python3 bench.py --adna --reads 1000000 --steps 1 --warmup 0 --no-cpu --no-e2e > gpurun_out/adna_t/out.json 2> gpurun_out/adna_t/err.log; echo rc=$?
grep "kernel D" gpurun_out/adna_t/err.log | head -5
