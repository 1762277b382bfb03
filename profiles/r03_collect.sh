#!/bin/bash
# Round-3 records (run on the GPU box through gpurun):  bash profiles/r03_collect.sh [stats|pmc|all]
#   stats: bench line + rocprofv3 --kernel-trace --stats of each workload on its own (the driver's default command runs all four in one
#          process; per-kernel averages would mix them), at the sizes the driver's line uses
#   pmc  : HBM traffic (FETCH_SIZE, WRITE_SIZE; separate passes, no tracing flags alongside) of the dominant kernel of each workload,
#          and the SQ / request counters of kernel D on the ancient-DNA workload
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WHAT=${1:-all}
O=gpurun_out/r03
mkdir -p $O
run() {   # name, bench args
  local name=$1; shift
  timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o run --output-format csv -- python3 bench.py --extras off "$@" > $O/${name}_bench.json 2> $O/${name}_bench.log
  echo "$name rc=$?"
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${name}_kernel_stats.csv
  rm -rf $O/prof_$name
}
pmc() {   # name, counter groups..., then -- bench args
  local name=$1; shift
  local groups=()
  while [ "$1" != "--" ]; do groups+=("$1"); shift; done; shift
  local i=0
  for g in "${groups[@]}"; do
    d=$O/pmc_$name/g$i; i=$((i+1)); mkdir -p $O/pmc_$name
    NABWA_BENCH_QUICK=1 timeout -k 10 600 rocprofv3 --pmc $g -d "$d" -o run --output-format csv -- python3 bench.py --extras off --no-cpu --no-e2e --steps 1 --warmup 0 "$@" > "$d.json" 2> "$d.err" || echo "group '$g' of $name failed (see $d.err)"
  done
  python3 profiles/summarize_pmc.py $O/pmc_$name > $O/pmc_$name.json
  # the launch size the pass was taken at and its dominant kernel: bench.py quotes a pass only for the same size (tests/test_profiles.py)
  python3 - "$O/pmc_$name.json" "$name" <<'PY'
import json, sys
p, name = sys.argv[1], sys.argv[2]
d = json.load(open(p))
d["units"], d["dominant"] = {"headline": (10_000_000, "S"), "adna": (6_250_000, "D"), "pe": (1_000_000, "D"), "repeats": (10_000_000, "D")}[name]
d["what"] = "rocprofv3 --pmc passes (profiles/r03_collect.sh pmc) of bench.py --extras off at the size of the driver line; per-launch means of the first-pass launches"
json.dump(d, open(p, "w"), indent=1)
PY
  echo "pmc $name:"; cat $O/pmc_$name.json
}
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  run headline --steps 5 --warmup 1
  run adna --adna --reads 6250000 --steps 2 --warmup 1 --cpu-seconds 8 --no-e2e
  run pe --pe --pairs 1000000 --steps 2 --warmup 1 --cpu-seconds 8
  run repeats --repeats --steps 2 --warmup 1 --cpu-seconds 8 --no-e2e
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  pmc headline "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" --
  pmc adna "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_FLAT" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" -- --adna --reads 6250000
  pmc pe "FETCH_SIZE" "WRITE_SIZE" -- --pe --pairs 1000000
  pmc repeats "FETCH_SIZE" "WRITE_SIZE" -- --repeats
fi
ls -la $O
