# statistics build of kernel D (no histogram) on the ancient-DNA workload, 1 M reads, for the builds named
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/st
for v in "$@"; do
  export NABWA_LIB=$GRAFT_REPO_ROOT/network-aware-bwa_amd/libnabwa_$v.so
  [ "$v" = cur ] && export NABWA_LIB=$GRAFT_REPO_ROOT/network-aware-bwa_amd/libnabwa.so
  NABWA_TIMING=1 NABWA_BENCH_QUICK=1 timeout -k 10 400 python3 bench.py --adna --reads 1000000 --steps 1 --warmup 0 --no-cpu --no-e2e --extras off > gpurun_out/st/adna_$v.json 2> gpurun_out/st/adna_$v.err; echo $v rc=$?
  grep -h "kernel D" gpurun_out/st/adna_$v.err | tail -3 | cut -c1-900
done
