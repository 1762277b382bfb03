#!/bin/bash
# Round-2 records (run on the GPU box through gpurun):  bash profiles/r02_collect.sh
# bench lines + rocprofv3 kernel statistics of the four workloads, then the counter passes of the headline kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
run() {   # name, bench args
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o run --output-format csv -- python3 bench.py "$@" > $O/${name}_bench.json 2> $O/${name}_bench.log
  echo "$name rc=$?"
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${name}_kernel_stats.csv
  rm -rf $O/prof_$name
}
run headline --steps 5 --warmup 1
run repeats --repeats --steps 3 --warmup 1 --no-e2e
run adna --adna --reads 1000000 --steps 3 --warmup 1 --cpu-seconds 12
run pe --pe --pairs 1000000 --steps 2 --warmup 1 --cpu-seconds 10
bash profiles/collect_pmc.sh r02 > $O/pmc.log 2>&1
cp gpurun_out/pmc_r02/summary.json $O/pmc_summary.json
bash profiles/collect_pmc_deep.sh r02_deep 1000000 > $O/pmc_deep.log 2>&1
cp gpurun_out/pmc_r02_deep/summary.json $O/pmc_deep_summary.json
ls -la $O
