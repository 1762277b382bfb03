"""Helpers for the CPU wave emulation of kernel D (tests/emu/deep_emu.cpp): build, bind, run.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

import nabwa_testlib as T

EMU_DIR = os.path.join(T.ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libdeep_emu.so")
CSRC = os.path.join(T.ROOT, "network-aware-bwa_amd", "csrc")


def build(asan=False):
    out = os.path.join(EMU_DIR, "libdeep_emu_asan.so" if asan else "libdeep_emu.so")
    srcs = [os.path.join(EMU_DIR, "deep_emu.cpp"), os.path.join(EMU_DIR, "emu_hip.hpp")] + [
        os.path.join(CSRC, f) for f in ("fm_deep_body.hpp", "wave_spmd.hpp", "nabwa_dev.hpp", "fm_search.hpp")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if asan else ["-O2"]
    subprocess.run(["g++", "-std=c++17", "-fPIC", "-shared", "-I" + EMU_DIR, "-Wall", "-Wno-unused-function",
                    "-Wno-unused-variable", "-Wno-unknown-pragmas"] + flags + [srcs[0], "-o", out], check=True)
    return out


def load(asan=False):
    lib = C.CDLL(build(asan))
    lib.emu_deep_search.restype = C.c_int
    lib.emu_deep_search.argtypes = [C.c_void_p] * 4 + [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 6
    return lib


def toy_words():
    """.bwt / .rbwt / .sa / .rsa of the toy index as u32 words"""
    return [np.fromfile(T.TOY + ext, np.uint32) for ext in (".bwt", ".rbwt", ".sa", ".rsa")]


def run(lib, words, opt, seq, rseq, off, per_read=0, max_lanes=64, careful=0, stage_k=32, n_pages=1 << 14, own_cap=1 << 14,
        per_wave=0, aln_cap=1024, text=0, lds=1, table=0, coop=64):
    """-> (rows per read, max_entries, status, stats)"""
    n = len(off) - 1
    knobs = np.array([max_lanes, careful, stage_k, n_pages, own_cap, per_wave, aln_cap, text, lds, table, coop], np.int32)
    n_aln = np.zeros(max(n, 1), np.int32)
    maxe = np.zeros(max(n, 1), np.int32)
    status = np.zeros(max(n, 1), np.uint8)
    rows = np.zeros((max(n, 1), aln_cap), T.ALN_DT)
    stats = np.zeros(16, np.uint64)
    rc = lib.emu_deep_search(T.ptr(words[0]), T.ptr(words[1]), T.ptr(words[2]), T.ptr(words[3]), C.byref(opt), n, T.ptr(off), T.ptr(seq), T.ptr(rseq), per_read,
                             T.ptr(knobs), T.ptr(n_aln), T.ptr(rows), T.ptr(maxe), T.ptr(status), T.ptr(stats))
    assert rc == 0
    return [rows[i, :n_aln[i]] for i in range(n)], maxe[:n], status[:n], stats
