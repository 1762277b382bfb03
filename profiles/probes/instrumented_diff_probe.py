#!/usr/bin/env python3
"""Probe: rows of the timed run vs rows of the instrumented (touch counting) run on the full bench workload, read by read;
the reads that differ are then searched by the compiled reference.  python profiles/probes/instrumented_diff_probe.py [reads]"""
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
n = 3099734149
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d_text = synth.synth_text_repeats(n, 20261004)
parts = [synth.build_index(d_text, n, rev, 32, True) for rev in (0, 1)]
host_bwt = [p[0].to_host(np.uint32, p[1]) for p in parts]
seq, rseq, off = synth.synth_reads(d_text, n, n_reads, 100, 2000, 0, 2)
opt = nabwa.gap_init_opt()
ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]), (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]), device=0, device_ptrs=True)
b = nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)
b.run(); n2 = b.sync()
na1, rows1, me1 = b.fetch_flat()
print("timed run: second pass", n2, "rows", len(rows1), flush=True)
print("touches", b.count_touches(), flush=True)
na2, rows2, me2 = b.fetch_flat()
print("instrumented run: rows", len(rows2), flush=True)
bad = np.flatnonzero(na1 != na2)
print("reads whose row count differs:", len(bad), bad[:20], flush=True)
print("max_entries differ:", int((me1 != me2).sum()))
if len(bad):
    ref = T.load_ref()
    ref.ref_index_wrap.restype = C.c_void_p
    ref.ref_index_wrap.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    ref.ref_cal_sa_reg_gap_mt.restype = C.c_long
    ref.ref_cal_sa_reg_gap_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    rix = C.c_void_p(ref.ref_index_wrap(T.ptr(host_bwt[0]), len(host_bwt[0]), T.ptr(host_bwt[1]), len(host_bwt[1])))
    copt = T.GapOpt(); C.memmove(C.byref(copt), C.byref(opt), 64)
    sel = bad[:64]
    s2 = np.concatenate([seq[off[i]:off[i + 1]] for i in sel]); r2 = np.concatenate([rseq[off[i]:off[i + 1]] for i in sel])
    o2 = np.concatenate([[0], np.cumsum([off[i + 1] - off[i] for i in sel])]).astype(np.int64)
    na = np.zeros(len(sel), np.int32); rows = np.zeros(1 << 20, T.ALN_DT)
    tot = ref.ref_cal_sa_reg_gap_mt(rix, C.byref(copt), len(sel), T.ptr(o2), T.ptr(s2), T.ptr(r2), 16, T.ptr(na), T.ptr(rows), len(rows))
    print("reference rows for the first differing reads:", list(na), "timed:", list(na1[sel]), "instrumented:", list(na2[sel]), "max_entries timed/instr:", list(me1[sel][:8]), list(me2[sel][:8]))
