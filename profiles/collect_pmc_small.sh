set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_v7
mkdir -p "$OUT"
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT")
i=0
for g in "${GROUPS_[@]}"; do
  d=$OUT/g$i; i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $g -d "$d" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu > "$d.json" 2> "$d.err" || echo "group '$g' failed"
done
python3 profiles/summarize_pmc.py "$OUT" > "$OUT/summary.json"
python3 -c "
import json; d=json.load(open('$OUT/summary.json'))
for k in ('W','S'):
    print(k, {a: (round(b/1e9,3) if isinstance(b,float) and b>1e6 else b) for a,b in sorted(d[k].items())})"
