"""Kernel D on the GPU (fm_deep.hip: one deep search per wavefront): the code that tests/test_deep_emu.py checks on a CPU
wave emulation, now as it ships -- through the C ABI, against the reference's .sai goldens and the oracle."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import nabwa_testlib as T
from test_gpu_parity import random_reads, to_gap_opt, toy_genome

pytestmark = pytest.mark.gpu
nabwa = importlib.import_module("network-aware-bwa_amd")

SAI_SETS = ["default", "adna", "n3", "e3", "loggap", "k1R5", "i2", "q20", "m64", "nonstop"]


@pytest.fixture(scope="module")
def gix():
    ix = nabwa.Index.load(T.TOY, 0, True)
    yield ix
    ix.close()


@pytest.fixture(scope="module")
def orc():
    lib = T.load_oracle()
    return lib, T.OracleIndex(lib)


@pytest.mark.parametrize("keyform,coop", [("1", None), ("0", None), ("1", "64"), ("0", "0")])
@pytest.mark.parametrize("name", SAI_SETS)
def test_every_golden_option_set_through_kernel_d(gix, orc, monkeypatch, name, keyform, coop):
    """a first-pass arena of 16 entries sends nearly every read on to kernel D: rows = the reference's .sai, max_entries = the oracle's;
    with entries in key form while the interval table reaches (the default) and as rows throughout; with the wave-wide expansion of
    one-row chains as the library picks it (on: these reads have 100 bases), for every such chain, and never"""
    monkeypatch.setenv("NABWA_CAP1", "16")
    monkeypatch.setenv("NABWA_DEEP_KEYFORM", keyform)
    if coop is not None:
        monkeypatch.setenv("NABWA_DEEP_COOP", coop)
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se_head.fq" if name == "nonstop" else "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads, opt.trim_qual)
    _, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off, n_threads=8)
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    b.run()
    n2 = b.sync()
    got, maxe = b.fetch()
    b.close()
    assert n2 > len(reads) // 3
    bad = [reads[i][0] for i in range(len(reads)) if got[i].tobytes() != gold[i].tobytes()]
    assert not bad, "kernel D differs from the reference .sai for %d reads, e.g. %s" % (len(bad), bad[:5])
    assert np.array_equal(maxe, wmaxe)


@pytest.mark.parametrize("depth", ["3", "6"])
def test_key_form_with_shallow_tables(orc, monkeypatch, depth):
    """interval tables of 3 and 6 levels: the entries of kernel D change from key form to rows that close to the roots; and reads shorter
    than the table is deep, whose hits and exact tails take their rows from the level of their own string"""
    monkeypatch.setenv("NABWA_KMER_T", depth)
    monkeypatch.setenv("NABWA_CAP1", "16")
    ix = nabwa.Index.load(T.TOY, 0, True)
    try:
        for name in ("adna", "default", "nonstop", "e3"):
            opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
            reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se_head.fq" if name == "nonstop" else "reads_se.fq"))
            seq, rseq, off, _ = T.encode_reads(reads, opt.trim_qual)
            _, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off, n_threads=8)
            b = nabwa.Batch(ix, to_gap_opt(opt), seq, rseq, off, False)
            b.run(); b.sync()
            got, maxe = b.fetch()
            b.close()
            bad = [reads[i][0] for i in range(len(reads)) if got[i].tobytes() != gold[i].tobytes()]
            assert not bad and np.array_equal(maxe, wmaxe), (name, bad[:5])
        rng = np.random.default_rng(15)
        g = toy_genome()
        reads = []
        for i in range(2000):
            L = int(rng.integers(3, 14))
            p = int(rng.integers(0, len(g) - L))
            sq = list(g[p:p + L])
            for _ in range(int(rng.integers(0, 3))):
                sq[int(rng.integers(0, L))] = "ACGTN"[int(rng.integers(0, 5))]
            reads.append(("s%d" % i, "".join(sq), "I" * L))
        seq, rseq, off, _ = T.encode_reads(reads)
        opt = deep_opt()
        opt.max_diff, opt.fnr, opt.seed_len = 2, -1.0, 5
        monkeypatch.setenv("NABWA_CAP1", "4")
        monkeypatch.setenv("NABWA_ALNCAP2", "8192")
        want, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off, n_threads=8)
        b = nabwa.Batch(ix, to_gap_opt(opt), seq, rseq, off, False)
        b.run()
        assert b.sync() > 500
        got, maxe = b.fetch()
        b.close()
        bad = [i for i in range(len(reads)) if got[i].tobytes() != want[i].tobytes()]
        assert not bad and np.array_equal(maxe, wmaxe), bad[:5]
    finally:
        ix.close()


def deep_opt():
    o = T.default_opt()
    o.fnr, o.max_gapo, o.seed_len = 0.01, 2, 16500
    return o


def test_many_waves_share_the_pool(gix, orc, monkeypatch):
    """20 000 noisy reads with the deep option set: thousands of waves draw pages from one pool at the same time"""
    monkeypatch.setenv("NABWA_CAP1", "128")
    rng = np.random.default_rng(5)
    reads = random_reads(rng, 20000, toy_genome(), lens=(50, 60, 76), err=0.04, indel=0.2)
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = deep_opt()
    want, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off, n_threads=16)
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    for _ in range(2):                                  # the second run re-uses the pool and the per-wave page lists
        b.run()
        n2 = b.sync()
        got, maxe = b.fetch()
        assert n2 > 4000
        bad = [i for i in range(len(reads)) if got[i].tobytes() != want[i].tobytes()]
        assert not bad, "%d reads differ, e.g. %s" % (len(bad), bad[:5])
        assert np.array_equal(maxe, wmaxe)
    b.close()


@pytest.mark.parametrize("max_entries", [40, 2500])
def test_max_entries_cutoff_on_the_gpu(gix, orc, monkeypatch, max_entries):
    """bwtgap.c:140 inside a round: the lanes before the cut-off are committed, the search goes on pop by pop"""
    monkeypatch.setenv("NABWA_CAP1", "16")
    rng = np.random.default_rng(6)
    reads = random_reads(rng, 400, toy_genome(), lens=(76, 100), err=0.05, indel=0.2)
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = deep_opt()
    opt.max_entries = max_entries
    want, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off, n_threads=8)
    assert (wmaxe > max_entries).any()
    got, maxe = gix.cal_sa_reg_gap(to_gap_opt(opt), seq, rseq, off)
    bad = [i for i in range(len(reads)) if got[i].tobytes() != want[i].tobytes()]
    assert not bad and np.array_equal(maxe, wmaxe)


def test_hit_lists_beyond_the_wide_rows_are_searched_again_with_longer_lists(gix, orc, monkeypatch):
    """NABWA_ALNCAP2 = 1 row per read in the wide result arrays: the reference's hit list grows without bound (bwtgap.c:186-190), so
    the reads with more hits are searched again with 16 x the rows (then 256 x ...), and every read gets its rows"""
    monkeypatch.setenv("NABWA_CAP1", "16")
    monkeypatch.setenv("NABWA_ALNCAP2", "1")
    for name in ("se_adna", "se_nonstop"):
        opt, gold = T.read_sai(os.path.join(T.GOLDEN, name + ".sai"))
        reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se_head.fq" if name == "se_nonstop" else "reads_se.fq"))
        seq, rseq, off, _ = T.encode_reads(reads)
        assert max(len(g) for g in gold) > 1
        for _ in range(2):                                           # the same batch run twice: the grown blocks of the first run are released
            b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
            b.run(); b.sync()
            b.run(); b.sync()
            got, _ = b.fetch()
            b.close()
            for i in range(len(reads)):
                assert got[i].tobytes() == gold[i].tobytes(), (name, reads[i][0])
        got, _ = gix.cal_sa_reg_gap(to_gap_opt(opt), seq, rseq, off)
        for i in range(len(reads)):
            assert got[i].tobytes() == gold[i].tobytes(), (name, reads[i][0])


def test_more_hit_rows_than_the_grown_lists_is_an_error_with_the_other_reads_intact(gix, orc, monkeypatch):
    monkeypatch.setenv("NABWA_HIT_GROW", "0")                        # no second search: the error path
    """NABWA_ALNCAP2 = 1 row per read in the wide result arrays: reads with more hits cannot be answered.  That is its own
    error code (never an empty answer that looks like "unmapped"), the resolved reads are still handed out, and the
    bwa_seq_t-level entry reports the error too."""
    monkeypatch.setenv("NABWA_CAP1", "16")
    monkeypatch.setenv("NABWA_ALNCAP2", "1")
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_adna.sai"))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads)
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    b.run()
    with pytest.raises(nabwa.NabwaError) as e:
        b.sync()
    assert e.value.code == nabwa.EHITS
    got, _ = b.fetch()
    b.close()
    multi = [i for i in range(len(reads)) if len(gold[i]) > 1]
    assert multi
    for i in range(len(reads)):
        assert got[i].tobytes() == gold[i].tobytes() or (len(gold[i]) > 1 and len(got[i]) == 0)
    with pytest.raises(nabwa.NabwaError) as e:
        gix.cal_sa_reg_gap(to_gap_opt(opt), seq, rseq, off)
    assert e.value.code == nabwa.EHITS
    n = 50
    arr = (nabwa.BwaSeq * n)()
    keep = []
    for i in range(n):
        s = np.ascontiguousarray(seq[off[i]:off[i + 1]]); r = np.ascontiguousarray(rseq[off[i]:off[i + 1]])
        keep += [s, r]
        arr[i].seq = s.ctypes.data; arr[i].rseq = r.ctypes.data; arr[i].bits0 = len(s)
    g = to_gap_opt(opt)
    rc = nabwa.lib().nabwa_bwa_cal_sa_reg_gap(gix._h, n, arr, C.byref(g))
    assert rc == (nabwa.EHITS if any(len(gold[i]) > 1 for i in range(n)) else nabwa.OK)
