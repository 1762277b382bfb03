"""CPU-side checks of the drop-in boundary: libnabwa.so loads and exports every symbol that
include/nabwa.h declares; struct layouts match the reference's; host-only entry points work.
No compute call is made here (no GPU in this container)."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

import nabwa_testlib as T

nabwa = importlib.import_module("network-aware-bwa_amd")


def declared_symbols():
    txt = open(os.path.join(T.ROOT, "include", "nabwa.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nabwa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(nabwa.LIB_PATH):
        nabwa.build()
    L = C.CDLL(nabwa.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), "libnabwa.so does not export %s" % s


def test_struct_layouts_match_reference():
    assert C.sizeof(nabwa.GapOpt) == 64            # gap_opt_t, bwtaln.h:143-153
    assert nabwa.ALN_DT.itemsize == 16             # bwt_aln1_t, bwtaln.h:41-45
    o = nabwa.gap_init_opt()                       # defaults, bwtaln.c:19-35
    assert (o.s_mm, o.s_gapo, o.s_gape) == (3, 11, 4)
    assert (o.max_diff, o.max_gapo, o.max_gape) == (-1, 1, 6)
    assert (o.indel_end_skip, o.max_del_occ, o.max_entries) == (5, 10, 2000000)
    assert o.mode == 3 and o.seed_len == 32 and o.max_seed_diff == 2 and o.max_top2 == 30
    assert abs(o.fnr - 0.04) < 1e-7
    # the header of a reference-written .sai is exactly this block
    raw = open(os.path.join(T.GOLDEN, "se_default.sai"), "rb").read(64)
    assert bytes(o) == raw


def test_maxdiff_matches_reference_table():
    vec = np.load(os.path.join(T.GOLDEN, "vectors.npz"))
    for l in range(1, 400):
        assert nabwa.cal_maxdiff(l, 0.02, 0.04) == vec["maxdiff_004"][l - 1]
        assert nabwa.cal_maxdiff(l, 0.02, 0.01) == vec["maxdiff_001"][l - 1]


def test_no_cpu_fallback_without_gpu():
    """Without a device the product path must fail loudly, not compute on the CPU."""
    if nabwa.lib().nabwa_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nabwa.NabwaError) as e:
        nabwa.Index.load(T.TOY)
    assert e.value.code == nabwa.ENODEV


def test_aln_tool_is_built_and_has_no_cpu_path():
    """nabwa_aln (the `bwa aln` command line) is built with the library; without a device it writes
    nothing and says why."""
    import subprocess
    tool = os.path.join(T.ROOT, "network-aware-bwa_amd", "nabwa_aln")
    if not os.path.exists(tool):
        nabwa.build()
    r = subprocess.run([tool], capture_output=True)
    assert r.returncode == 1 and b"Usage:   nabwa_aln [options] <prefix> <in.fq>" in r.stderr
    if nabwa.lib().nabwa_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([tool, T.TOY, os.path.join(T.GOLDEN, "reads_se_head.fq")], capture_output=True)
    assert r.returncode == 2 and r.stdout == b"" and b"cannot set up the index" in r.stderr


def test_encode_read_matches_reference_encoding():
    """nabwa_encode_read == what bam1_to_seq / bwa_read_seq produce (checked against the test helper whose
    output reproduces the reference's .sai files bit for bit, incl. -q trimming and the reverse-flag undo)"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    for trim in (0, 20):
        seq, rseq, off, full = T.encode_reads(reads, trim)
        for i in (0, 1, 7, 100, 590, 595, 600, 605):
            name, s, q = reads[i]
            codes = T.NT4[np.frombuffer(s.encode(), np.uint8)]
            qual = np.frombuffer(q.encode(), np.uint8) - 33
            a, b = nabwa.encode_read(codes, qual, False, trim)
            assert np.array_equal(a, seq[off[i]:off[i + 1]]) and np.array_equal(b, rseq[off[i]:off[i + 1]]), name
            # a record stored reverse-complemented with the reverse flag set decodes to the same read
            rc = np.where(codes < 4, 3 - codes, codes)[::-1]
            a2, b2 = nabwa.encode_read(rc, qual[::-1], True, trim)
            assert np.array_equal(a2, a) and np.array_equal(b2, b), name
    assert C.sizeof(nabwa.BwaSeq) == 200
