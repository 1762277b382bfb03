# kernel times of several builds of the library on the deep workloads (NABWA_LIB picks the build); no CPU legs
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/var
for v in "$@"; do
  export NABWA_LIB=$GRAFT_REPO_ROOT/network-aware-bwa_amd/libnabwa_$v.so
  [ "$v" = cur ] && export NABWA_LIB=$GRAFT_REPO_ROOT/network-aware-bwa_amd/libnabwa.so
  NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --adna --reads 6250000 --steps 2 --warmup 1 --no-cpu --no-e2e --extras off > gpurun_out/var/adna_$v.json 2> gpurun_out/var/adna_$v.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/var/adna_$v.json'));print('$v adna 6.25M D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['checksum'])"
  NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu --extras off > gpurun_out/var/pe_$v.json 2> gpurun_out/var/pe_$v.err || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/var/pe_$v.json'));print('$v pe 1M D ms', d['roofline']['deep_kernel_ms'], 'S ms', d['roofline']['search_kernel_ms'], d['config']['checksum'])"
done
