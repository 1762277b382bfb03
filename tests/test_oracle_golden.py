"""The CPU oracle (oracle/nabwa_oracle.c) against golden vectors produced by the reference's own
code (tests/golden/make_golden.py).  This is what pins the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import nabwa_testlib as T

SAI_SETS = ["default", "adna", "n3", "e3", "loggap", "k1R5", "i2", "q20", "m64"]


@pytest.fixture(scope="module")
def lib():
    return T.load_oracle()


@pytest.fixture(scope="module")
def ix(lib):
    return T.OracleIndex(lib)


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(T.GOLDEN, "vectors.npz"))


def test_rank_primitives(lib, ix, vec):
    buf = (C.c_uint32 * 4)()
    b2 = (C.c_uint32 * 4)()
    for which in (0, 1):
        b = ix.bwt(which)
        ks, occ, occ4 = vec["occ_k%d" % which], vec["occ_v%d" % which], vec["occ4_v%d" % which]
        for i, k in enumerate(ks):
            for c in range(4):
                assert lib.orc_occ(b, int(k), c) == occ[i, c]
            lib.orc_occ4(b, int(k), buf)
            assert list(buf) == list(occ4[i])
        for i in range(len(vec["p_k%d" % which])):
            lib.orc_2occ4(b, int(vec["p_k%d" % which][i]), int(vec["p_l%d" % which][i]), buf, b2)
            assert list(buf) == list(vec["p_ck%d" % which][i])
            assert list(b2) == list(vec["p_cl%d" % which][i])


def test_sa_lookup(lib, ix, vec):
    for which in (0, 1):
        for k, v in zip(vec["sa_k%d" % which], vec["sa_v%d" % which]):
            assert lib.orc_sa(ix.bwt(which), int(k)) == v


def test_maxdiff(lib, vec):
    for l in range(1, 400):
        assert lib.orc_maxdiff(l, 0.02, 0.04) == vec["maxdiff_004"][l - 1]
        assert lib.orc_maxdiff(l, 0.02, 0.01) == vec["maxdiff_001"][l - 1]


@pytest.mark.parametrize("name", SAI_SETS + ["nonstop"])
def test_sai_parity(lib, ix, name):
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se_head.fq" if name == "nonstop" else "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads, opt.trim_qual)
    assert len(gold) == len(reads)
    got, _ = T.oracle_cal_sa_reg_gap(lib, ix.h, opt, seq, rseq, off, per_read=0)
    bad = [reads[i][0] for i in range(len(reads)) if got[i].tobytes() != gold[i].tobytes()]
    assert not bad, "oracle differs from reference .sai for %d reads, e.g. %s" % (len(bad), bad[:5])


def test_threads_do_not_change_results(lib, ix):
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads)
    got, _ = T.oracle_cal_sa_reg_gap(lib, ix.h, opt, seq, rseq, off, per_read=0, n_threads=4)
    assert all(got[i].tobytes() == gold[i].tobytes() for i in range(len(reads)))


def test_global_dp(lib, vec):
    sm = [np.array([11, -19, -19, -19, -13, -19, 11, -19, -19, -13, -19, -19, 11, -19, -13,
                    -19, -19, -19, 11, -13, -13, -13, -13, -13, -13], np.int32),
          np.array([1, -3, -3, -3, -2, -3, 1, -3, -3, -2, -3, -3, 1, -3, -2,
                    -3, -3, -3, 1, -2, -2, -2, -2, -2, -2], np.int32)]
    cig = (C.c_uint32 * 1024)()
    ncig = C.c_int()
    for t in range(int(vec["dp_n"])):
        r = np.ascontiguousarray(vec["dp_ref"][vec["dp_ref_off"][t]:vec["dp_ref_off"][t + 1]])
        q = np.ascontiguousarray(vec["dp_qry"][vec["dp_qry_off"][t]:vec["dp_qry_off"][t + 1]])
        go, ge, gend, band, smid = [int(x) for x in vec["dp_params"][vec["dp_pid"][t]]]
        sc = lib.orc_global(T.ptr(r), len(r), T.ptr(q), len(q), go, ge, gend, T.ptr(sm[smid]), 5, band,
                            cig, C.byref(ncig))
        want = vec["dp_cig"][vec["dp_cig_off"][t]:vec["dp_cig_off"][t + 1]]
        assert sc == vec["dp_score"][t], t
        assert list(cig[:ncig.value]) == list(want), t


def test_global_dp_on_the_references_own_demo(lib):
    """the one known answer the reference ships for this path: the global alignment of its stdaln demo
    (stdaln.c:1048,1056-1058, built with -DSTDALN_MAIN, under aln_param_blast = {5, 2, 2, aln_sm_blast, 5, 50},
    stdaln.c:214-226) prints `>1,34 1,32 1D16M1D16M`"""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}

    def enc(s):
        return np.array([code.get(c.upper(), 4) for c in s], np.uint8)
    sm = np.array([1, -3, -3, -3, -2, -3, 1, -3, -3, -2, -3, -3, 1, -3, -2, -3, -3, -3, 1, -2, -2, -2, -2, -2, -2], np.int32)
    r, q = enc("CGTGCGATGCactgCATACGGCTCGCCTAGATCA"), enc("AAGGGATGCTCTGCATCGgCTCGGCTAGCTGT")
    assert (len(r), len(q)) == (34, 32)
    cig = (C.c_uint32 * 64)()
    ncig = C.c_int()
    lib.orc_global(T.ptr(r), len(r), T.ptr(q), len(q), 5, 2, 2, T.ptr(sm), 5, 50, cig, C.byref(ncig))
    assert "".join("%d%s" % (c >> 4, "MID"[c & 15]) for c in cig[:ncig.value]) == "1D16M1D16M"


@pytest.mark.parametrize("name", ["default", "adna", "q20"])
def test_sam_parity(lib, ix, name):
    """aln2seq (RNG in record order) -> SA lookup -> mapQ -> gap refinement -> MD/NM/XA, against samse output."""
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    sam = T.parse_sam(os.path.join(T.GOLDEN, "se_%s.sam" % name))
    seq, rseq, off, full = T.encode_reads(reads, opt.trim_qual)
    names = ["chr1", "chr2", "chr3"]
    rng = (C.c_uint64 * 1)()
    lib.orc_srand48(rng, 11)
    rec = T.SeRec()
    for i, (rd, g) in enumerate(zip(reads, sam)):
        L = int(off[i + 1] - off[i])
        a = np.ascontiguousarray(gold[i])
        s0 = np.ascontiguousarray(seq[off[i]:off[i + 1]])
        s1 = np.ascontiguousarray(rseq[off[i]:off[i + 1]])
        lib.orc_se_finish(ix.h, C.byref(opt), rng, L, int(full[i]), T.ptr(s0), T.ptr(s1), len(a), T.ptr(a), 3,
                          C.byref(rec))
        assert g["name"] == rd[0]
        assert rec.flag == g["flag"], rd[0]
        if rec.type == 0:
            assert g["rname"] == "*"
            continue
        assert names[rec.seqid] == g["rname"], rd[0]
        assert rec.rpos == g["pos"], rd[0]
        assert rec.mapQ == g["mapq"], rd[0]
        cig = T.cigar16_str(rec.cigar[:rec.n_cigar]) if rec.n_cigar else "%dM" % rec.len
        assert cig == g["cigar"], rd[0]
        tg = g["tags"]
        assert rec.xt.decode() == tg["XT"], rd[0]
        assert rec.nm == tg["NM"] and rec.md.decode() == tg["MD"], rd[0]
        assert rec.c1 == tg["X0"], rd[0]
        if rec.c1 <= opt.max_top2:
            assert rec.c2 == tg["X1"], rd[0]
        assert (rec.n_mm, rec.n_gapo, rec.n_gapo + rec.n_gape) == (tg["XM"], tg["XO"], tg["XG"]), rd[0]
        if rec.clip_len < rec.full_len:
            assert tg["XC"] == rec.clip_len
        xa = ""
        for j in range(rec.n_multi):
            m = rec.multi[j]
            mc = T.cigar16_str(m.cigar[:m.n_cigar]) if m.n_cigar else "%dM" % rec.len  # p->len after bwa_correct_trimmed (bwase.c:558)
            end = m.pos + (sum(x & 0x3fff for x in m.cigar[:m.n_cigar] if (x >> 14) in (0, 2)) if m.n_cigar else L)
            sid = max(k for k, o in enumerate([0, 60000, 100000]) if m.pos >= o)
            xa += "%s,%s%d,%s,%d;" % (names[sid], "-" if m.strand else "+", m.pos - [0, 60000, 100000][sid] + 1, mc,
                                      m.gap + m.mm)
        assert xa == tg.get("XA", ""), rd[0]
