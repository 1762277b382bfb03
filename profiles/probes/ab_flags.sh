cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/abf
for v in own2 own4 own8 own16; do
export NABWA_LIB=$PWD/gpurun_ab/lib_$v.so
timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-cpu --no-e2e > gpurun_out/abf/$v.json 2> gpurun_out/abf/$v.err || { echo "$v failed"; tail -3 gpurun_out/abf/$v.err; continue; }
python3 -c "import json;d=json.load(open('gpurun_out/abf/$v.json'));r=d['roofline'];print('$v', d['value'], 'S', r['search_kernel_ms'], 'D', r['deep_kernel_ms'], 'W', r['width_kernel']['kernel_ms'], d['config']['checksum'])"
done
