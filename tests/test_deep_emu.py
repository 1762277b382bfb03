"""Kernel D (fm_deep_body.hpp: one deep search per wavefront, speculative rounds + ordered commit) on the CPU.

The kernel body is compiled by g++ as an emulation of one 64-lane wave (tests/emu/, wave_spmd.hpp) and must reproduce
the reference's `.sai` rows (tests/golden/, written by the compiled reference) and the oracle's rows and `max_entries`
for every option set, with any number of lanes per round, with the smallest staging buffers (chains that continue over
rounds), in careful mode (one pop per round), across the `max_entries` cut-off, with several waves sharing a page pool
and with a pool that runs dry.  The same code then runs on the GPU (tests/test_gpu_deep.py)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import emu_deep as E
import nabwa_testlib as T

SAI_SETS = ["default", "adna", "n3", "e3", "loggap", "k1R5", "i2", "q20", "m64", "nonstop"]


@pytest.fixture(scope="module")
def emu():
    return E.load()


@pytest.fixture(scope="module")
def words():
    return E.toy_words()


@pytest.fixture(scope="module")
def orc():
    lib = T.load_oracle()
    return lib, T.OracleIndex(lib)


def golden(name):
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se_head.fq" if name == "nonstop" else "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads, opt.trim_qual)
    return opt, gold, reads, seq, rseq, off


def check(got, maxe, st, want, wmaxe, what):
    bad = [i for i in range(len(want)) if got[i].tobytes() != want[i].tobytes() or st[i] != 0 or (wmaxe is not None and maxe[i] != wmaxe[i])]
    assert not bad, "%s: %d reads differ, e.g. %s" % (what, len(bad), bad[:5])


@pytest.mark.parametrize("name", SAI_SETS)
def test_golden_sai(emu, words, orc, name):
    """rows = the reference's .sai; max_entries = the oracle's (the .sai does not hold it)"""
    opt, gold, reads, seq, rseq, off = golden(name)
    ctr = T.Counters()
    _, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off, counters=ctr)
    for lanes in (64, 5) if name in ("default", "adna", "m64") else (64,):
        got, maxe, st, stats = E.run(emu, words, opt, seq, rseq, off, max_lanes=lanes)
        check(got, maxe, st, gold, wmaxe, "%s, %d lanes" % (name, lanes))
        # the instrumented kernel counts the reference algorithm's bucket touches (the roofline's algorithmic bytes): only the
        # chains that are committed count, so speculation must not change the total
        assert stats[8] + stats[9] == ctr.n_bucket, (name, lanes, stats[8:10], ctr.n_bucket)
    # text mode: an exact tail that has narrowed to one row is finished by comparing the read with the text
    got, maxe, st, stats = E.run(emu, words, opt, seq, rseq, off, text=1)
    check(got, maxe, st, gold, wmaxe, "%s, text mode" % name)
    assert stats[7] > 0
    assert stats[13] > 0, "no forced levels were walked on the text"
    assert stats[14] > 0 and stats[15] >= stats[14], "no chain on one row was taken by the wave: %s" % stats[14:16]
    got, maxe, st, stats = E.run(emu, words, opt, seq, rseq, off, text=1, coop=3, stage_k=2)
    check(got, maxe, st, gold, wmaxe, "%s, text mode, few records per chain, the wave takes over below 4 chains" % name)
    # key form: entries whose strings are shorter than the interval table is deep carry the string, not its rows (fm_deep.hpp); a shallow
    # table moves the change to rows close to the roots, a deep one to where intervals are a few rows wide
    for table, kn in ((2, dict(coop=0)), (5, dict(max_lanes=7)), (9, dict(text=1)), (9, dict(text=1, coop=2, stage_k=3))):
        got, maxe, st, stats = E.run(emu, words, opt, seq, rseq, off, table=table, **kn)
        check(got, maxe, st, gold, wmaxe, "%s, table of depth %d" % (name, table))
        # (chains in key form are expanded in chain steps, or -- few chains left in the round -- taken by the whole wave down to the table's depth)
        assert (stats[10] > 0 or stats[14] > 0) and (stats[11] > 0 or table < 5), "no entry in key form was expanded / finished: %s" % stats[10:15]
        assert stats[10] > 0 or kn.get("coop", 64), "without the hand-over the chain steps expand the entries in key form"
        assert stats[12] > 0 and (stats[13] > 0 or not kn.get("text")), "no forced levels were walked through the table / on the text: %s" % stats[12:14]


def noisy_reads(seed, n, lens, err):
    from test_gpu_parity import random_reads, toy_genome
    rng = np.random.default_rng(seed)
    return random_reads(rng, n, toy_genome(), lens=lens, err=err, indel=0.2)


def deep_opt():
    o = T.default_opt()
    o.fnr, o.max_gapo, o.seed_len = 0.01, 2, 16500          # what ancient-DNA pipelines run with: deep searches
    return o


@pytest.mark.parametrize("knobs", [dict(), dict(max_lanes=1), dict(max_lanes=7), dict(careful=1), dict(stage_k=1), dict(stage_k=3, max_lanes=64),
                                   dict(stage_k=48, text=1), dict(per_wave=25), dict(per_read=1), dict(text=1, max_lanes=9), dict(lds=0), dict(lds=0, text=1),
                                   dict(table=1), dict(table=4, careful=1), dict(table=7, stage_k=1), dict(table=8, text=1, per_wave=25), dict(table=10, lds=0, text=1),
                                   dict(table=6, per_read=1, max_lanes=3), dict(text=1, coop=0), dict(text=1, coop=2, table=7), dict(text=1, coop=64, careful=1)])
def test_noisy_reads_vs_oracle(emu, words, orc, knobs):
    reads = noisy_reads(11, 150, (50, 63, 76, 100), 0.04)
    seq, rseq, off, _ = T.encode_reads(reads)
    for opt in (T.default_opt(), deep_opt()):
        kn = dict(knobs)
        per_read = kn.pop("per_read", 0)
        want, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off, per_read=per_read)
        got, maxe, st, stats = E.run(emu, words, opt, seq, rseq, off, per_read=per_read, **kn)
        check(got, maxe, st, want, wmaxe, str(knobs))
        if not knobs:
            assert stats[0] * 4 < stats[2], "rounds should pop several entries each: %s" % stats[:4]


@pytest.mark.parametrize("max_entries", [3, 40, 300, 2500])
def test_max_entries_cutoff(emu, words, orc, max_entries):
    """bwtgap.c:140: the search ends when more than max_entries are live; the statistic and the rows found until then must
    be the reference's, wherever in a round the cut-off falls"""
    reads = noisy_reads(12, 120, (76, 100), 0.05)
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = deep_opt()
    opt.max_entries = max_entries
    want, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off)
    assert (wmaxe > max_entries).any()
    for lanes in (64, 2):
        got, maxe, st, stats = E.run(emu, words, opt, seq, rseq, off, max_lanes=lanes)
        check(got, maxe, st, want, wmaxe, "max_entries %d, %d lanes" % (max_entries, lanes))


def test_pool_runs_dry_and_hit_rows_run_out(emu, words, orc):
    reads = noisy_reads(13, 60, (100,), 0.05)
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = deep_opt()
    want, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off)
    # a pool of 6 pages: the reads that fit are exact, the others are flagged for the guaranteed pass, none is wrong
    got, maxe, st, stats = E.run(emu, words, opt, seq, rseq, off, n_pages=6, own_cap=6)
    assert (st == 3).any() and (st == 0).any() and stats[5] == (st == 3).sum()
    for i in range(len(reads)):
        assert st[i] == 3 or (got[i].tobytes() == want[i].tobytes() and maxe[i] == wmaxe[i])
    # one row per read: reads with more hits say so (status 4), the others are exact
    got, maxe, st, _ = E.run(emu, words, opt, seq, rseq, off, aln_cap=1)
    multi = np.array([len(w) > 1 for w in want])
    assert multi.any() and ((st == 4) == multi).all()
    for i in np.flatnonzero(~multi):
        assert got[i].tobytes() == want[i].tobytes()


def test_reads_shorter_than_the_table(emu, words, orc):
    """a hit, an exact tail and the cut-off inside the table's depth: the rows come from the level of the entry's own string"""
    from test_gpu_parity import toy_genome
    rng = np.random.default_rng(15)
    g = toy_genome()
    reads = []
    for i in range(300):
        L = int(rng.integers(3, 14))
        p = int(rng.integers(0, len(g) - L))
        sq = list(g[p:p + L])
        for _ in range(int(rng.integers(0, 3))):
            sq[int(rng.integers(0, L))] = "ACGTN"[int(rng.integers(0, 5))]
        reads.append(("s%d" % i, "".join(sq), "I" * L))
    seq, rseq, off, _ = T.encode_reads(reads)
    for opt in (T.default_opt(), deep_opt()):
        opt.max_diff, opt.fnr, opt.seed_len = 2, -1.0, 5
        for mode_bits in (0, 0x10):                      # the default, and -N: every hit within max_diff
            opt.mode = (opt.mode & ~0x10) | mode_bits
            want, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off)
            for table in (0, 3, 8, 11):
                got, maxe, st, _ = E.run(emu, words, opt, seq, rseq, off, table=table, text=1 if table == 8 else 0, aln_cap=4096)
                check(got, maxe, st, want, wmaxe, "short reads, table %d" % table)


def test_empty_and_all_n_reads(emu, words, orc):
    reads = [("e", "", ""), ("n", "N" * 60, "I" * 60), ("a", "ACGT" * 15, "I" * 60), ("n2", "ACGTN" * 12, "I" * 60)]
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = T.default_opt()
    want, wmaxe = T.oracle_cal_sa_reg_gap(orc[0], orc[1].h, opt, seq, rseq, off)
    for table in (0, 6):
        got, maxe, st, _ = E.run(emu, words, opt, seq, rseq, off, table=table)
        check(got, maxe, st, want, wmaxe, "edge reads")


def test_under_address_sanitizer():
    """the same emulation, built with -fsanitize=address,undefined, on the deep option set (run in a child: ASan must be loaded first)"""
    so = E.build(asan=True)
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, nabwa_testlib as T, emu_deep as E, test_deep_emu as D\n"
            "lib = E.load(asan=True); words = E.toy_words(); o = T.load_oracle(); ox = T.OracleIndex(o)\n"
            "reads = D.noisy_reads(14, 40, (50, 76, 100), 0.04); seq, rseq, off, _ = T.encode_reads(reads)\n"
            "for kn in (dict(), dict(stage_k=2), dict(n_pages=8, own_cap=8), dict(per_wave=7), dict(text=1), dict(text=1, stage_k=48), dict(table=7), dict(table=9, text=1, stage_k=2), dict(text=1, coop=2)):\n"
            "    opt = D.deep_opt(); want, wm = T.oracle_cal_sa_reg_gap(o, ox.h, opt, seq, rseq, off)\n"
            "    got, maxe, st, _ = E.run(lib, words, opt, seq, rseq, off, **kn)\n"
            "    assert all(st[i] == 3 or got[i].tobytes() == want[i].tobytes() for i in range(len(reads)))\n"
            "print('asan ok')\n" % os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", PYTHONMALLOC="malloc")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "asan ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert os.path.exists(so)
