"""Experiment (not part of the product): how much does the search kernel gain when the reads of a wave are
alike?  Reads are re-ordered by (strand, n_mm, n_hits) of their first-run result, then timed again."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nabwa = importlib.import_module("network-aware-bwa_amd")
synth = importlib.import_module("network-aware-bwa_amd.synth")
n = int(os.environ.get("G", 3099734149)); R = int(os.environ.get("R", 10000000))
d_text = synth.synth_text(n, 20261004, n_dup=2000, dup_len=5000)
parts = [synth.build_index(d_text, n, rev, 32, False) for rev in (0, 1)]
ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]), device_ptrs=True)
for p in parts: p[0].free()
seq, rseq, off = synth.synth_reads(d_text, n, R, 100, 2000, 0, 2)
d_text.free()
opt = nabwa.gap_init_opt()
def timeit(seq, rseq, off, tag):
    b = nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)
    b.run(); b.sync()
    ks = []
    for _ in range(3):
        b.run(); b.sync(); ks.append((b.last_kernel_ms(), b.last_width_ms()))
    print(tag, "S/W ms", np.mean([k[0] for k in ks]), np.mean([k[1] for k in ks]), flush=True)
    return b
b = timeit(seq, rseq, off, "original order")
hits, _ = b.fetch(); b.close()
key = np.array([(int(h[0]["info"]) >> 24 & 1) * 100 + (int(h[0]["info"]) & 0xff) * 10 + min(len(h), 9) if len(h) else 999 for h in hits])
order = np.argsort(key, kind="stable")
print("classes:", {int(k): int(c) for k, c in zip(*np.unique(key, return_counts=True))})
S = seq.reshape(R, 100)[order].reshape(-1); RS = rseq.reshape(R, 100)[order].reshape(-1)
os.environ["NABWA_SYNC_REFILL"] = "0"; timeit(S, RS, off, "sorted, async refill").close()
os.environ["NABWA_SYNC_REFILL"] = "1"; timeit(S, RS, off, "sorted, wave-sync refill").close()
