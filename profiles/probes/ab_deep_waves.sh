cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/ab
for w in 2; do
  NABWA_LIB=$PWD/gpurun_ab/libnabwa_w$w.so NABWA_TIMING=1 timeout -k 10 400 python3 bench.py --adna --reads 1000000 --steps 2 --warmup 1 --no-cpu --no-e2e > gpurun_out/ab/w$w.json 2> gpurun_out/ab/w$w.err || exit 1
  echo "waves/SIMD $w: $(python3 -c "import json;d=json.load(open('gpurun_out/ab/w$w.json'));print(d['value'], d['roofline']['deep_kernel_ms'], d['config']['checksum'])")"
  grep "kernel D" gpurun_out/ab/w$w.err | tail -1
done
