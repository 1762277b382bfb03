"""Would reads with ONE substitution run well in lockstep waves if the waves held reads whose substitution sits at a similar
position?  10 M exact synthetic reads; a fraction gets one substitution at a position drawn from [lo, hi); the search runs
with every wave in lockstep (NABWA_SYNC_REFILL=1, work order: exact reads first) and in the production mode.
   python profiles/exp_lockstep_class1.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nabwa = importlib.import_module("network-aware-bwa_amd")
synth = importlib.import_module("network-aware-bwa_amd.synth")
n = 3099734149; R = 10_000_000; L = 100
d_text = synth.synth_text(n, 20261004, n_dup=2000, dup_len=5000)
parts = [synth.build_index(d_text, n, rev, 32, True) for rev in (0, 1)]
ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]), (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]), device_ptrs=True)
for p in parts:
    p[0].free(); p[2].free()
seq0, rseq0, off = synth.synth_reads(d_text, n, R, L, 0, 0, 2)
d_text.free()
opt = nabwa.gap_init_opt()
rng = np.random.default_rng(1)
for lo, hi in ((0, 100), (40, 60), (50, 51), (0, 16), (84, 100)):
    seq = seq0.copy().reshape(R, L); rseq = rseq0.copy().reshape(R, L)
    pick = np.nonzero(rng.random(R) < 0.18)[0]
    pos = rng.integers(lo, hi, len(pick))                     # index into seq (the reversed read)
    add = rng.integers(1, 4, len(pick)).astype(np.uint8)
    seq[pick, pos] = (seq[pick, pos] + add) & 3
    rseq[pick, pos] = 3 - seq[pick, pos]
    res = []
    for sync in ("0", "1"):
        os.environ["NABWA_SYNC_REFILL"] = sync
        b = nabwa.Batch(ix, opt, seq.reshape(-1), rseq.reshape(-1), off, per_read=True)
        b.run(); b.sync(); b.run(); b.sync()
        res.append((b.last_kernel_ms(), b.last_width_ms()))
        b.close()
    print("substitution in seq[%d, %d): production S %.1f ms (W %.1f); all waves lockstep S %.1f ms" % (lo, hi, res[0][0], res[0][1], res[1][0]), flush=True)
