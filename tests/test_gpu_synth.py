"""The GPU index builder used by bench.py (bench infrastructure) must write exactly what the
reference's `bwa index` writes: checked byte-for-byte on the toy genome whose index files were
produced by the reference itself."""
import importlib
import os

import numpy as np
import pytest

import nabwa_testlib as T

pytestmark = pytest.mark.gpu
nabwa = importlib.import_module("network-aware-bwa_amd")
synth = importlib.import_module("network-aware-bwa_amd.synth")


def toy_text():
    pac = np.fromfile(T.TOY + ".pac", np.uint8)
    n = (len(pac) - 1) * 4 - (4 - int(pac[-1])) if pac[-1] else (len(pac) - 2) * 4
    n = 103000
    codes = np.zeros(n, np.uint8)
    idx = np.arange(n)
    codes[:] = (pac[idx >> 2] >> ((~idx & 3) << 1)) & 3
    return codes


@pytest.mark.parametrize("reverse,ext_bwt,ext_sa", [(0, ".bwt", ".sa"), (1, ".rbwt", ".rsa")])
def test_builder_reproduces_reference_index_files(reverse, ext_bwt, ext_sa):
    codes = toy_text()
    d_text = synth.DevArray.from_host(np.concatenate([codes, np.zeros(64, np.uint8)]))
    bw, nb, sw, ns = synth.build_index(d_text, len(codes), reverse, 32, True)
    got_bwt = bw.to_host(np.uint32, nb)
    got_sa = sw.to_host(np.uint32, ns)
    want_bwt = np.fromfile(T.TOY + ext_bwt, np.uint32)
    want_sa = np.fromfile(T.TOY + ext_sa, np.uint32)
    assert len(got_bwt) == len(want_bwt) and np.array_equal(got_bwt, want_bwt)
    # words 1..4 of a .sa file are "skipped" by the reader (bwtio.c:170); the reference writes L2 there
    assert len(got_sa) == len(want_sa) and np.array_equal(got_sa, want_sa)
    for d in (d_text, bw, sw):
        d.free()


def test_synthetic_genome_end_to_end_vs_oracle():
    """1 Mbp synthetic genome with planted repeats -> GPU-built index -> GPU search == CPU oracle search
    on the same arrays (the path bench.py takes, at a size the oracle finishes in seconds)"""
    n = 1_000_003
    d_text = synth.synth_text(n, 99, n_dup=40, dup_len=700)
    parts = []
    for rev in (0, 1):
        parts.append(synth.build_index(d_text, n, rev, 32, True))
    ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]),
                                 (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]), device_ptrs=True)
    seq, rseq, off = synth.synth_reads(d_text, n, 4000, 100, 8000, 60000, 5)
    opt = nabwa.gap_init_opt()
    got, gmax = ix.cal_sa_reg_gap(opt, seq, rseq, off)
    h0 = parts[0][0].to_host(np.uint32, parts[0][1])
    h1 = parts[1][0].to_host(np.uint32, parts[1][1])
    olib = T.load_oracle()
    oh = olib.orc_index_wrap(T.ptr(h0), len(h0), T.ptr(h1), len(h1))
    want, wmax = T.oracle_cal_sa_reg_gap(olib, oh, T.default_opt(), seq, rseq, off, n_threads=8)
    bad = [i for i in range(len(got)) if got[i].tobytes() != want[i].tobytes()]
    assert not bad, "first differing read %d" % bad[0]
    assert np.array_equal(gmax, wmax)
    n_hit = sum(1 for g in got if len(g))
    assert n_hit > 3500          # the reads really come from this genome
    # SA samples are consistent with the text: suffix SA[k] starts with the read-independent k-th smallest suffix
    text = d_text.to_host(np.uint8, n)
    k = np.arange(1, 2000, dtype=np.uint32)
    sa = ix.sa_lookup(np.zeros(len(k), np.uint8), k)
    pre = [bytes(text[p:p + 40]) for p in sa]
    assert pre == sorted(pre)
    ix.close()
    d_text.free()
    for p in parts:
        p[0].free()
        p[2].free()


def test_repeat_genome_with_the_deepest_interval_table(monkeypatch):
    """A genome with the repeat families of `bench.py --repeats` (homopolymer and short-unit tandem repeats among them), searched
    with the interval table at its full depth T = 16, where a path key uses all 32 bits: the all-T key 0xffffffff must survive the
    "interval not empty" test of the key-form expansion (round 2: it did not -- reads from poly-T stretches lost a hit; the uniform
    genome has no such reads, and smaller genomes pick a smaller T).  Rows and max_entries against the oracle."""
    monkeypatch.setenv("NABWA_KMER_T", "16")
    n = 40_000_003
    d_text = synth.synth_text_repeats(n, 7)
    text = d_text.to_host(np.uint8, n)
    run = np.flatnonzero(np.convolve((text == 3).astype(np.int32), np.ones(16, np.int32), "valid") == 16)
    assert len(run) > 100, "the genome must hold poly-T stretches"
    parts = [synth.build_index(d_text, n, rev, 32, True) for rev in (0, 1)]
    ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]),
                                 (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]), device_ptrs=True)
    assert ix.export(0, 4, 0, 1)[0] == 16
    seq, rseq, off = synth.synth_reads(d_text, n, 60000, 100, 3000, 0, 9)
    # plus reads cut straight out of homopolymer loci (both strands end up starting with sixteen T)
    extra = []
    for p in run[:: max(1, len(run) // 400)][:400]:
        w = text[max(0, p - 50):max(0, p - 50) + 100]
        if len(w) == 100:
            extra.append(w[::-1].copy())                      # bwa_seq_t.seq: the read reversed
    es = np.concatenate(extra)
    seq = np.concatenate([seq, es]); rseq = np.concatenate([rseq, (3 - es).astype(np.uint8)])
    off = np.concatenate([off, off[-1] + 100 * np.arange(1, len(extra) + 1)]).astype(np.int64)
    opt = nabwa.gap_init_opt()
    got, gmax = ix.cal_sa_reg_gap(opt, seq, rseq, off)
    h0 = parts[0][0].to_host(np.uint32, parts[0][1])
    h1 = parts[1][0].to_host(np.uint32, parts[1][1])
    olib = T.load_oracle()
    oh = olib.orc_index_wrap(T.ptr(h0), len(h0), T.ptr(h1), len(h1))
    want, wmax = T.oracle_cal_sa_reg_gap(olib, oh, T.default_opt(), seq, rseq, off, n_threads=16)
    bad = [i for i in range(len(got)) if got[i].tobytes() != want[i].tobytes()]
    assert not bad, "%d reads differ, first %d" % (len(bad), bad[0])
    assert np.array_equal(gmax, wmax)
    assert sum(len(w) > 1 for w in want) > 200              # repeats: reads with several hit rows
    ix.close()
    d_text.free()
    for p in parts:
        p[0].free()
        p[2].free()
