#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the two search kernels:  bash profiles/collect_pmc_traffic.sh <tag>
set -e
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  d=$OUT/g$i; i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $g -d "$d" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu > "$d.json" 2> "$d.err" || echo "group '$g' failed"
done
python3 profiles/summarize_pmc.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
