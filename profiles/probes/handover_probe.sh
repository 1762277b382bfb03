cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/ho
NABWA_DEEP_DUMP=$PWD/gpurun_out/ho/pe.bin NABWA_BENCH_QUICK=1 timeout -k 10 300 python3 bench.py --pe --pairs 1000000 --steps 1 --warmup 0 --no-cpu > gpurun_out/ho/pe.json 2> gpurun_out/ho/pe.err || { tail -3 gpurun_out/ho/pe.err; exit 1; }
python3 - <<'PY'
import numpy as np
a=np.fromfile('gpurun_out/ho/pe.bin.all',np.uint8).reshape(-1,4)
mn=np.minimum(a[:,0],a[:,1]); mx=np.maximum(a[:,0],a[:,1])
print("reads", len(a), "handed on", int(a[:,2].sum()))
for c in range(5):
    m=mn==c
    print("min class", c, "reads", int(m.sum()), "handed on", int(a[m,2].sum()), "%.3f"%(a[m,2].mean() if m.any() else 0))
for c in range(5):
    for d in range(c,5):
        m=(mn==c)&(mx==d)
        if m.sum()>1000: print("classes", (c,d), "reads", int(m.sum()), "handed on %.3f"%a[m,2].mean())
PY
