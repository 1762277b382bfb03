cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/do
NABWA_DEEP_DUMP=$PWD/gpurun_out/do/dump.bin timeout -k 10 300 python3 bench.py --adna --reads 1000000 --steps 1 --warmup 0 --no-cpu --no-e2e > gpurun_out/do/out.json 2> gpurun_out/do/err.log || exit 1
ls -la gpurun_out/do/dump.bin
python3 profiles/probes/deep_order_probe.py gpurun_out/do/dump.bin 4096 | tee gpurun_out/do/probe.txt
