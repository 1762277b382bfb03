"""Throughput of the paired-end chain (nabwa_pe_posn + nabwa_pe_finish) on synthetic FR pairs, reduced genome.
   G=<genome bp> R=<pairs> NABWA_TIMING=1 python profiles/exp_pe_finish_rate.py"""
import ctypes as C
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nabwa = importlib.import_module("network-aware-bwa_amd")
synth = importlib.import_module("network-aware-bwa_amd.synth")
n = int(os.environ.get("G", 500000000)); R = int(os.environ.get("R", 1000000)); L = 100
d_text = synth.synth_text(n, 20261004, n_dup=2000, dup_len=5000)
parts = [synth.build_index(d_text, n, rev, 32, True) for rev in (0, 1)]
ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]), (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]), device_ptrs=True)
text = d_text.to_host(np.uint8, n)
pad = (-n) % 4
t4 = np.concatenate([text, np.zeros(pad, np.uint8)]).reshape(-1, 4)
pac = (t4[:, 0] << 6 | t4[:, 1] << 4 | t4[:, 2] << 2 | t4[:, 3]).astype(np.uint8)
os.makedirs("/tmp/syn", exist_ok=True)
pac.tofile("/tmp/syn/g.pac")
with open("/tmp/syn/g.pac", "ab") as f:
    f.write(bytes([0, n % 4]) if n % 4 == 0 else bytes([n % 4]))
open("/tmp/syn/g.ann", "w").write("%d 1 11\n0 chrS synthetic\n0 %d 0\n" % (n, n))
open("/tmp/syn/g.amb", "w").write("%d 1 0\n" % n)
ix.attach_reference("/tmp/syn/g")
# FR pairs: insert ~N(300, 25); 0.5 % substitutions; 3 % of second ends replaced by a far-away read (discordant)
rng = np.random.default_rng(5)
isz = np.clip(rng.normal(300, 25, R), 210, 420).astype(np.int64)
p0 = rng.integers(0, n - 500, R)
idx = np.arange(L)
r1 = text[p0[:, None] + idx]                                   # forward read
r2f = text[(p0 + isz - L)[:, None] + idx]                       # fragment's right end, forward strand
far = rng.random(R) < 0.03
pf = rng.integers(0, n - L, R)
r2f[far] = text[pf[far][:, None] + idx]
for r in (r1, r2f):
    m = rng.random(r.shape) < 0.005
    r[m] = (r[m] + rng.integers(1, 4, int(m.sum()))) & 3
r2 = 3 - r2f[:, ::-1]                                           # second read = reverse complement of the right end
reads = np.empty((2 * R, L), np.uint8)
reads[0::2] = r1; reads[1::2] = r2
seq = reads[:, ::-1].reshape(-1).copy()                         # bwa_seq_t.seq: the read reversed
rseq = (3 - reads[:, ::-1]).reshape(-1).copy()                  # bwa_seq_t.rseq: the complement of seq (bwaseqio.c:225-232)
off = np.arange(2 * R + 1, dtype=np.int64) * L
opt = nabwa.gap_init_opt()
t = time.time(); hits, _ = ix.cal_sa_reg_gap(opt, seq, rseq, off, per_read=True); t1 = time.time() - t
full = np.full(2 * R, L, np.int32)
t = time.time(); recs, st = ix.pe_posn(opt, off, full, hits, nabwa.srand48_state(11)); t2 = time.time() - t
h = np.zeros(100000, np.uint16)
for i in range(0, R):
    a, b = recs[2 * i].se, recs[2 * i + 1].se
    if a.type and b.type:
        d = nabwa.lib().nabwa_isize_bin(2, a.mapQ, b.mapQ, a.pos, a.len, b.pos, b.len)
        if d >= 0 and h[d] < 65535:
            h[d] += 1
    if h.sum() > 200000:
        break
rc, ii = nabwa.isize_infer(h, 1e-5, n)
print("isize: rc %d avg %.1f std %.1f high_bayesian %d" % (rc, ii.avg, ii.std, ii.high_bayesian))
t = time.time(); tot, mp = ix.pe_finish(opt, nabwa.pe_opt_default(), ii, seq, rseq, off, hits, recs); t3 = time.time() - t
pp = sum(1 for i in range(0, 2 * R, 199) if recs[i].se.flag & 2)
sw = sum(1 for i in range(0, 2 * R, 199) if recs[i].se.type == 3)
print("search %.2f s; pe_posn %.2f s = %.2f M pairs/s; pe_finish %.2f s = %.2f M pairs/s (python wrappers included); rescue tried on %d pairs, fixed %d; sample: proper %d of %d, rescued %d"
      % (t1, t2, R / t2 / 1e6, t3, R / t3 / 1e6, tot[0], mp[0], pp, len(range(0, 2 * R, 199)), sw))
