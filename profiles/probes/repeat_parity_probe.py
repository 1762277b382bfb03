#!/usr/bin/env python3
"""Probe: on a small genome with the repeat families of bench.py --repeats, compare the GPU search rows of several kernel
configurations with the compiled reference, read by read (run on the GPU box).  Usage: python profiles/probes/repeat_parity_probe.py [genome_len] [reads]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nabwa_testlib as T

nabwa = importlib.import_module("network-aware-bwa_amd")
synth = importlib.import_module("network-aware-bwa_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
d_text = synth.synth_text_repeats(n, 20261004)
parts = [synth.build_index(d_text, n, rev, 32, True) for rev in (0, 1)]
host_bwt = [p[0].to_host(np.uint32, p[1]) for p in parts]
seq, rseq, off = synth.synth_reads(d_text, n, n_reads, 100, 2000, 0, 2)
opt = nabwa.gap_init_opt()
ref = T.load_ref()
ref.ref_index_wrap.restype = C.c_void_p
ref.ref_index_wrap.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
ref.ref_cal_sa_reg_gap_mt.restype = C.c_long
ref.ref_cal_sa_reg_gap_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
rix = C.c_void_p(ref.ref_index_wrap(T.ptr(host_bwt[0]), len(host_bwt[0]), T.ptr(host_bwt[1]), len(host_bwt[1])))
copt = T.GapOpt(); C.memmove(C.byref(copt), C.byref(opt), 64)
na = np.zeros(n_reads, np.int32); rows = np.zeros(256 * n_reads, T.ALN_DT)
tot = ref.ref_cal_sa_reg_gap_mt(rix, C.byref(copt), n_reads, T.ptr(off), T.ptr(seq), T.ptr(rseq), 16, T.ptr(na), T.ptr(rows), len(rows))
assert tot >= 0
bnd = np.concatenate([[0], np.cumsum(na)])
print("reference: %d reads, %d rows, reads with > 1 row: %d, with > 16 rows: %d" % (n_reads, tot, (na > 1).sum(), (na > 16).sum()), flush=True)
CONFIGS = ({}, {"NABWA_TRIP_BUDGET": "0"}, {"NABWA_TEXT_KERNELS": "0"}, {"NABWA_TEXT_KERNELS": "5"}, {"NABWA_TEXT_KERNELS": "3"}, {"NABWA_TEXT_KERNELS": "1"}, {"NABWA_CAP1": "16", "NABWA_TEXT_KERNELS": "0"}, {"NABWA_CAP1": "16", "NABWA_TEXT_KERNELS": "2"})
if len(sys.argv) > 3 and sys.argv[3] == "big":
    CONFIGS = ({}, {"NABWA_TEXT_KERNELS": "1"}, {"NABWA_TEXT_KERNELS": "2"}, {"NABWA_TEXT_KERNELS": "4"}, {"NABWA_TEXT_KERNELS": "6"}, {"NABWA_TRIP_BUDGET": "0"}, {"NABWA_KMER_T": "15"})
    xx = ({"NABWA_KMER_T": "0", "NABWA_TEXT_KERNELS": "0"}, {"NABWA_KMER_T": "0", "NABWA_TEXT_KERNELS": "0", "NABWA_TRIP_BUDGET": "0"}, {"NABWA_KMER_T": "0"}, {"NABWA_TEXT_KERNELS": "0"}, {"NABWA_KMER_T": "0", "NABWA_TEXT_KERNELS": "0", "NABWA_CAP1": "64"})
for env in CONFIGS:
    for k in ("NABWA_TRIP_BUDGET", "NABWA_TEXT_KERNELS", "NABWA_CAP1", "NABWA_KMER_T", "COUNT", "NABWA_DEEP_LANES", "NABWA_TIER_A"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]), (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]), device=0, device_ptrs=True)
    b = nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)
    b.run(); n2 = b.sync()
    if env.get("COUNT"):
        print("touches", b.count_touches(), flush=True)      # leaves the instrumented run's rows in the batch
    g_na, g_rows, _ = b.fetch_flat()
    gb = np.concatenate([[0], np.cumsum(g_na)])
    bad = [i for i in range(n_reads) if g_na[i] != na[i] or g_rows[gb[i]:gb[i + 1]].tobytes() != rows[bnd[i]:bnd[i + 1]].tobytes()]
    print(env, "second pass", n2, "differ", len(bad), bad[:5], flush=True)
    for i in bad[:3]:
        print("  read", i, "n ref/gpu", na[i], g_na[i], "ref", rows[bnd[i]:bnd[i + 1]][:3], "... gpu", g_rows[gb[i]:gb[i + 1]][:3])
        r0, g0 = rows[bnd[i]:bnd[i + 1]], g_rows[gb[i]:gb[i + 1]]
        m = min(len(r0), len(g0)); d = [j for j in range(m) if r0[j] != g0[j]]
        print("   first differing row", d[:1], (r0[d[0]], g0[d[0]]) if d else None)
    b.close(); ix.close()
