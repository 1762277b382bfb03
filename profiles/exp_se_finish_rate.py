"""Throughput of the SE finishing chain (nabwa_se_finish) on the bench workload, reduced genome."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nabwa = importlib.import_module("network-aware-bwa_amd")
synth = importlib.import_module("network-aware-bwa_amd.synth")
n = int(os.environ.get("G", 500000000)); R = int(os.environ.get("R", 2000000))
d_text = synth.synth_text(n, 20261004, n_dup=2000, dup_len=5000)
parts = [synth.build_index(d_text, n, rev, 32, True) for rev in (0, 1)]
ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]), (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]), device_ptrs=True)
# reference annotations for the synthetic genome: one contig, no holes, 2-bit packed text
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
seq, rseq, off = synth.synth_reads(d_text, n, R, 100, 2000, 20000, 2)
opt = nabwa.gap_init_opt()
t = time.time(); hits, _ = ix.cal_sa_reg_gap(opt, seq, rseq, off, per_read=True); t1 = time.time() - t
full = np.full(R, 100, np.int32)
t = time.time(); recs, st = ix.se_finish(opt, seq, rseq, off, full, hits, 3, nabwa.srand48_state(11)); t2 = time.time() - t
mapped = sum(1 for i in range(0, R, 997) if recs[i].type)
gapped = sum(1 for i in range(0, R, 97) if recs[i].n_cigar)
print("search (one-shot API incl. transfers) %.2f s = %.2f M reads/s; se_finish %.2f s = %.2f M reads/s; sample mapped %d gapped %d" % (t1, R / t1 / 1e6, t2, R / t2 / 1e6, mapped, gapped))
