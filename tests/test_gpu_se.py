"""GPU tests of the single-end finishing chain through the product's C ABI:
 * nabwa_global_align (HIP aln_global_core) against the DP vectors the reference produced;
 * FM search -> nabwa_se_finish (host RNG in record order, GPU bwt_sa batch, GPU gap refinement, MD/NM, flags)
   against the reference's own `samse` output for the same reads: position, CIGAR, MAPQ, NM, MD, X0/X1/XM/XO/XG,
   XA, XT, flags (contig bridging, trimming)."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import nabwa_testlib as T

pytestmark = pytest.mark.gpu
nabwa = importlib.import_module("network-aware-bwa_amd")

SM = [np.array([11, -19, -19, -19, -13, -19, 11, -19, -19, -13, -19, -19, 11, -19, -13,
                -19, -19, -19, 11, -13, -13, -13, -13, -13, -13], np.int32),
      np.array([1, -3, -3, -3, -2, -3, 1, -3, -3, -2, -3, -3, 1, -3, -2,
                -3, -3, -3, 1, -2, -2, -2, -2, -2, -2], np.int32)]


@pytest.mark.parametrize("wave,small", [("1", "4096"), ("0", "4096"), ("0", "0")])
def test_global_align_golden(monkeypatch, wave, small):
    # every form of the kernel: one wavefront per task (rows and directions in LDS: what runs unless a task is too large for it), and one pair
    # per lane with the rows in LDS for a handful of tasks / in HBM for many
    monkeypatch.setenv("NABWA_DP_WAVE", wave)
    monkeypatch.setenv("NABWA_DP_SMALL", small)
    vec = np.load(os.path.join(T.GOLDEN, "vectors.npz"))
    n = int(vec["dp_n"])
    for pid in range(len(vec["dp_params"])):
        go, ge, gend, band, smid = [int(x) for x in vec["dp_params"][pid]]
        idx = [t for t in range(n) if vec["dp_pid"][t] == pid]
        refs = [vec["dp_ref"][vec["dp_ref_off"][t]:vec["dp_ref_off"][t + 1]] for t in idx]
        qrys = [vec["dp_qry"][vec["dp_qry_off"][t]:vec["dp_qry_off"][t + 1]] for t in idx]
        ro = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.int64)
        qo = np.concatenate([[0], np.cumsum([len(q) for q in qrys])]).astype(np.int64)
        score, cigs = nabwa.global_align(np.concatenate(refs), ro, np.concatenate(qrys), qo, go, ge, gend, SM[smid], band,
                                         max_cigar=512)
        for j, t in enumerate(idx):
            want = vec["dp_cig"][vec["dp_cig_off"][t]:vec["dp_cig_off"][t + 1]]
            assert score[j] == vec["dp_score"][t], (pid, t)
            assert list(cigs[j]) == list(want), (pid, t)
        if pid == 0:                                   # the working memory kept between calls can be given back at any time
            nabwa.lib().nabwa_dp_scratch_release(0)


@pytest.mark.parametrize("wave", ["1", "0"])
def test_global_align_random_vs_oracle(monkeypatch, wave):
    """fresh random pairs incl. length-1 and very unequal lengths, several parameter blocks; one wavefront per task, and one pair per lane"""
    monkeypatch.setenv("NABWA_DP_WAVE", wave)
    olib = T.load_oracle()
    rng = np.random.default_rng(5)
    refs, qrys = [], []
    for _ in range(700):
        l2 = int(rng.integers(1, 160))
        q = rng.integers(0, 5, l2).astype(np.uint8)
        if rng.random() < 0.6:
            r = list(q)
            for _ in range(int(rng.integers(0, 5))):
                p = int(rng.integers(0, len(r) + 1))
                if rng.random() < 0.5 and len(r) > 1:
                    del r[min(p, len(r) - 1)]
                else:
                    r.insert(p, int(rng.integers(0, 4)))
            r = np.array(r if r else [0], np.uint8)
        else:
            r = rng.integers(0, 5, int(rng.integers(1, 200))).astype(np.uint8)
        refs.append(r)
        qrys.append(q)
    ro = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.int64)
    qo = np.concatenate([[0], np.cumsum([len(q) for q in qrys])]).astype(np.int64)
    cig = (C.c_uint32 * 1024)()
    ncig = C.c_int()
    for go, ge, gend, sm, band in ((26, 9, 5, SM[0], 50), (26, 9, -1, SM[0], 50), (5, 2, 2, SM[1], 7), (8, 2, 2, SM[1], 3)):
        score, cigs = nabwa.global_align(np.concatenate(refs), ro, np.concatenate(qrys), qo, go, ge, gend, sm, band,
                                         max_cigar=512)
        for i, (r, q) in enumerate(zip(refs, qrys)):
            r = np.ascontiguousarray(r)
            q = np.ascontiguousarray(q)
            sc = olib.orc_global(T.ptr(r), len(r), T.ptr(q), len(q), go, ge, gend, T.ptr(sm), 5, band, cig, C.byref(ncig))
            assert score[i] == sc, (i, band)
            assert list(cigs[i]) == list(cig[:ncig.value]), (i, band)


@pytest.mark.parametrize("name", ["default", "adna", "q20"])
def test_se_chain_matches_reference_sam(name):
    opt, _ = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    sam = T.parse_sam(os.path.join(T.GOLDEN, "se_%s.sam" % name))
    seq, rseq, off, full = T.encode_reads(reads, opt.trim_qual)
    g = nabwa.GapOpt()
    C.memmove(C.byref(g), C.byref(opt), 64)
    ix = nabwa.Index.load(T.TOY, 0, True, True)
    hits, _ = ix.cal_sa_reg_gap(g, seq, rseq, off, per_read=False)          # bwa aln semantics (one call for the file)
    # two batches, to check that the RNG stream continues across calls exactly as the process-global one does
    cut = 301
    st = nabwa.srand48_state(11)                                           # srand48(bns->seed), seed 11 (bntseq.c:181)
    recs = []
    for lo, hi in ((0, cut), (cut, len(reads))):
        o = off[lo:hi + 1] - off[lo]
        out, st = ix.se_finish(g, seq[off[lo]:off[hi]], rseq[off[lo]:off[hi]], o, full[lo:hi], hits[lo:hi], 3, st)
        recs.extend(out[i] for i in range(hi - lo))
    names = ["chr1", "chr2", "chr3"]
    offs = [0, 60000, 100000]
    for rd, r, gsam in zip(reads, recs, sam):
        assert gsam["name"] == rd[0]
        assert r.flag == gsam["flag"], rd[0]
        if r.type == 0:
            continue
        assert names[r.seqid] == gsam["rname"] and r.rpos == gsam["pos"] and r.mapQ == gsam["mapq"], rd[0]
        cig = T.cigar16_str(r.cigar[:r.n_cigar]) if r.n_cigar else "%dM" % r.len
        assert cig == gsam["cigar"], rd[0]
        tg = gsam["tags"]
        assert r.xt.decode() == tg["XT"] and r.nm == tg["NM"] and r.md.decode() == tg["MD"], rd[0]
        assert r.c1 == tg["X0"], rd[0]
        if r.c1 <= opt.max_top2:
            assert r.c2 == tg["X1"], rd[0]
        assert (r.n_mm, r.n_gapo, r.n_gapo + r.n_gape) == (tg["XM"], tg["XO"], tg["XG"]), rd[0]
        xa = ""
        for j in range(r.n_multi):
            m = r.multi[j]
            mc = T.cigar16_str(m.cigar[:m.n_cigar]) if m.n_cigar else "%dM" % r.len
            sid = max(k for k, o in enumerate(offs) if m.pos >= o)
            xa += "%s,%s%d,%s,%d;" % (names[sid], "-" if m.strand else "+", m.pos - offs[sid] + 1, mc, m.gap + m.mm)
        assert xa == tg.get("XA", ""), rd[0]
    ix.close()


FORMS = [("lds", "diag"), ("hbm", "diag"), ("lds", "rows"), ("hbm", "rows")]


def set_form(monkeypatch, rows, forward="diag"):
    """the switches of dp_wave.hip: the two rows of a task in LDS or in HBM (NABWA_DP_ROWS), and the forward pass of the local alignment
    along the anti-diagonals or row by row -- the form that carries the reference's 16-bit drop (NABWA_DP_FORWARD)"""
    monkeypatch.setenv("NABWA_DP_ROWS", rows)
    monkeypatch.setenv("NABWA_DP_FORWARD", forward)


def flat(v, tag, idx=None):
    ro, qo = v[tag + "_ref_off"], v[tag + "_qry_off"]
    idx = range(len(ro) - 1) if idx is None else idx
    refs = [v[tag + "_ref"][ro[t]:ro[t + 1]] for t in idx]
    qrys = [v[tag + "_qry"][qo[t]:qo[t + 1]] for t in idx]
    r_o = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.int64)
    q_o = np.concatenate([[0], np.cumsum([len(q) for q in qrys])]).astype(np.int64)
    return np.concatenate(refs), r_o, np.concatenate(qrys), q_o


@pytest.mark.parametrize("rows", ["lds", "hbm"])
def test_extend_align_golden(monkeypatch, rows):
    """aln_extend_core (named by the north star; reached from bwasw in the reference): known answers from the reference"""
    set_form(monkeypatch, rows)
    v = np.load(os.path.join(T.GOLDEN, "vectors_sw.npz"))
    n = len(v["pid"])
    for pid in range(len(v["params"])):
        go, ge, gend, band, smid = [int(x) for x in v["params"][pid]]
        idx = [t for t in range(n) if v["pid"][t] == pid]
        refs = [v["ref"][v["ref_off"][t]:v["ref_off"][t + 1]] for t in idx]
        qrys = [v["qry"][v["qry_off"][t]:v["qry_off"][t + 1]] for t in idx]
        ro = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.int64)
        qo = np.concatenate([[0], np.cumsum([len(q) for q in qrys])]).astype(np.int64)
        score, cigs = nabwa.extend_align(np.concatenate(refs), ro, np.concatenate(qrys), qo, go, ge, SM[smid], band,
                                         v["g0"][idx], max_cigar=512)
        for j, t in enumerate(idx):
            assert score[j] == v["ext_score"][t], (pid, t)
            want = v["ext_cig"][v["ext_cig_off"][t]:v["ext_cig_off"][t + 1]]
            assert list(cigs[j]) == list(want), (pid, t)


@pytest.mark.parametrize("rows", ["lds", "hbm"])
@pytest.mark.parametrize("tag", ["ext", "extlong"])
def test_extend_align_at_size(monkeypatch, rows, tag):
    """extensions of 100 - 400 bases under two bands, and of 3000 - 3600 bases whose scores pass 32000 -- where the reference's 16-bit rows
    drop by 16000 (stdaln.c:919-932); answers of the compiled reference (make_golden.py sw_rescue)"""
    set_form(monkeypatch, rows)
    v = np.load(os.path.join(T.GOLDEN, "vectors_sw_rescue.npz"))
    n = len(v[tag + "_score"])
    for band in sorted(set(v[tag + "_band"].tolist())):
        idx = [t for t in range(n) if v[tag + "_band"][t] == band]
        ref, ro, qry, qo = flat(v, tag, idx)
        score, cigs = nabwa.extend_align(ref, ro, qry, qo, 26, 9, SM[0], int(band), v[tag + "_g0"][idx], max_cigar=512)
        for j, t in enumerate(idx):
            assert score[j] == v[tag + "_score"][t], (tag, t)
            assert list(cigs[j]) == list(v[tag + "_cig"][v[tag + "_cig_off"][t]:v[tag + "_cig_off"][t + 1]]), (tag, t)
    if tag == "extlong":
        assert v[tag + "_score"].max() > 32000


@pytest.mark.parametrize("rows,forward", FORMS)
def test_local_align_on_rescue_sized_tasks(monkeypatch, rows, forward):
    """400 tasks shaped like mate rescue (a window of 300 - 620 bases, a read of 70 - 250 with substitutions, an indel now and
    then, some reads that are not in their window at all, N runs) against the compiled reference's answers (make_golden.py
    sw_rescue): score, first and last cell of the path, sub-optimal score, CIGAR -- in every form of the kernel"""
    set_form(monkeypatch, rows, forward)
    v = np.load(os.path.join(T.GOLDEN, "vectors_sw_rescue.npz"))
    ref, ro, qry, qo = flat(v, "loc")
    score, coords, subo, cigs = nabwa.local_align(ref, ro, qry, qo, 26, 9, SM[0], 50, 1, max_cigar=62)
    assert sum(1 for x in score if x > 0) > 300
    for t in range(len(score)):
        assert score[t] == v["loc_score"][t], t
        assert list(cigs[t]) == list(v["loc_cig"][v["loc_cig_off"][t]:v["loc_cig_off"][t + 1]]), t
        assert tuple(coords[t]) == tuple(v["loc_coords"][t]), (t, coords[t], v["loc_coords"][t])
        assert subo[t] == v["loc_subo"][t], t


@pytest.mark.parametrize("rows", ["lds", "hbm"])
def test_local_align_long_reads_take_the_16_bit_drop(monkeypatch, rows):
    """reads of 3000 - 3600 bases: the forward score passes 32000 and the reference's rows drop by 16000, forward (stdaln.c:583-602) and
    reverse (:654-666); such tasks go row by row here of themselves"""
    set_form(monkeypatch, rows)
    v = np.load(os.path.join(T.GOLDEN, "vectors_sw_rescue.npz"))
    ref, ro, qry, qo = flat(v, "loclong")
    score, coords, subo, cigs = nabwa.local_align(ref, ro, qry, qo, 26, 9, SM[0], 50, 1, max_cigar=512)
    assert v["loclong_score"].max() > 32000
    for t in range(len(score)):
        assert score[t] == v["loclong_score"][t], t
        assert list(cigs[t]) == list(v["loclong_cig"][v["loclong_cig_off"][t]:v["loclong_cig_off"][t + 1]]), t
        assert tuple(coords[t]) == tuple(v["loclong_coords"][t]) and subo[t] == v["loclong_subo"][t], t


@pytest.mark.parametrize("rows,forward", FORMS)
def test_local_align_golden(monkeypatch, rows, forward):
    """aln_local_core (mate-rescue Smith-Waterman, bwape.c:456): scores, sub-optimal scores and CIGARs from the reference, in every form
    of the kernel (a wave per task; rows in LDS or HBM; forward pass along the anti-diagonals or row by row)"""
    set_form(monkeypatch, rows, forward)
    v = np.load(os.path.join(T.GOLDEN, "vectors_sw.npz"))
    n = len(v["pid"])
    for pid in range(len(v["params"])):
        go, ge, gend, band, smid = [int(x) for x in v["params"][pid]]
        idx = [t for t in range(n) if v["pid"][t] == pid]
        refs = [v["ref"][v["ref_off"][t]:v["ref_off"][t + 1]] for t in idx]
        qrys = [v["qry"][v["qry_off"][t]:v["qry_off"][t + 1]] for t in idx]
        ro = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.int64)
        qo = np.concatenate([[0], np.cumsum([len(q) for q in qrys])]).astype(np.int64)
        score, coords, subo, cigs = nabwa.local_align(np.concatenate(refs), ro, np.concatenate(qrys), qo, go, ge, SM[smid],
                                                      band, 1, max_cigar=512)
        for j, t in enumerate(idx):
            assert score[j] == v["loc_score"][t], (pid, t)
            want = v["loc_cig"][v["loc_cig_off"][t]:v["loc_cig_off"][t + 1]]
            assert list(cigs[j]) == list(want), (pid, t)
            if len(want):
                assert subo[j] == v["loc_subo"][t], (pid, t)
