#!/bin/bash
# Counter passes for the two search kernels (run on the GPU box through gpurun):
#   bash profiles/collect_pmc.sh <tag>
# One rocprofv3 --pmc pass per counter group (no tracing flags alongside, per the pool's rules), each on a
# short bench run; results under gpurun_out/pmc_<tag>/<group>/ and reduced by profiles/summarize_pmc.py.
set -e
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum")
i=0
for g in "${GROUPS_[@]}"; do
  d=$OUT/g$i; i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $g -d "$d" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-e2e > "$d.json" 2> "$d.err" || echo "group '$g' failed (see $d.err)"
done
python3 profiles/summarize_pmc.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
