#!/bin/bash
# Counter passes for kernel D (fm_deep_kernel) on the ancient-DNA workload (run on the GPU box through gpurun):
#   bash profiles/collect_pmc_deep.sh <tag> [reads]
# One rocprofv3 --pmc pass per counter group (no tracing flags alongside), each on a short bench run.
set -e
TAG=${1:-run}
READS=${2:-250000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_FLAT" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS")
i=0
for g in "${GROUPS_[@]}"; do
  d=$OUT/g$i; i=$((i+1))
  NABWA_BENCH_QUICK=1 timeout -k 10 300 rocprofv3 --pmc $g -d "$d" -o run --output-format csv -- python3 bench.py --adna --reads $READS --steps 1 --warmup 0 --no-cpu > "$d.json" 2> "$d.err" || echo "group '$g' failed (see $d.err)"
done
python3 profiles/summarize_pmc.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
