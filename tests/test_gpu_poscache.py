"""finish_pair's position cache (my_hash, reference bam2bam.c:741-757): hit rows of 1000 suffixes or more get their text positions once
per file, keyed by (k, l) alone, with the strand and LENGTH of the read that brought the row first.  Reads of other lengths that end on
the same base of a repeat share the row (same suffixes of the reversed text) and, in `bam2bam -t 1`, pair on the first read's positions
-- shifted by the difference of the lengths.  nabwa_pe_finish_cached must write what the reference's chain writes with one cache over
the file; without a cache (NULL) it must write what the chain writes without one.

The index is built by the compiled reference's `index` command; the expected records come from oracle/_ref: our restatement of
finish_pair's glue (ref_harness.c) around the reference's own functions and its own hash map."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

import nabwa_testlib as T

nabwa = importlib.import_module("network-aware-bwa_amd")
pytestmark = pytest.mark.gpu

N_COPIES, ELEM, SPACER = 1150, 140, 260


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    refbin = os.path.join(T.ROOT, "oracle", "_ref", "bwa_ref")
    ref = T.load_ref()
    if ref is None or not os.path.exists(refbin):
        pytest.skip("the compiled reference (oracle/_ref) did not travel")
    tmp = tmp_path_factory.mktemp("poscache")
    rng = np.random.default_rng(77)
    elem = "".join("ACGT"[c] for c in rng.integers(0, 4, ELEM))
    parts, starts = [], []
    at = 0
    for _ in range(N_COPIES):                       # 1150 exact copies of one element, each between spacers of its own
        sp = "".join("ACGT"[c] for c in rng.integers(0, 4, SPACER))
        parts.append(sp); at += SPACER
        starts.append(at)
        parts.append(elem); at += ELEM
    parts.append("".join("ACGT"[c] for c in rng.integers(0, 4, 500)))
    text = "".join(parts)
    fa = str(tmp / "rep.fa")
    with open(fa, "w") as f:
        f.write(">rep\n")
        for o in range(0, len(text), 70):
            f.write(text[o:o + 70] + "\n")
    r = subprocess.run([refbin, "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # pairs: end 0 lies inside the element and ENDS on its base 130 whatever its length (50..100); end 1 is 90 unique bases of the spacer
    # 150..240 bases further on, so the pair belongs to one copy.  Both orders of the strands occur.
    reads = []
    for i in range(240):
        c = int(rng.integers(0, N_COPIES))
        L = int(rng.integers(50, 101)) if i % 4 else 70
        a_end = starts[c] + 130
        r0 = text[a_end - L:a_end]
        m0 = starts[c] + ELEM + 60 + int(rng.integers(0, 60))
        r1 = revcomp(text[m0:m0 + 90])
        if i % 2:                                   # the same fragment read from the other side
            r0, r1 = revcomp(r0), revcomp(r1)
            reads.append((r1, r0) if i % 4 == 1 else (r0, r1))
        else:
            reads.append((r0, r1))
    inter = []
    for i, (x, y) in enumerate(reads):
        inter.append(("p%d" % i, x, "I" * len(x)))
        inter.append(("p%d" % i, y, "I" * len(y)))
    seq, rseq, off, full = T.encode_reads(inter)
    ix = nabwa.Index.load(fa, 0, True, True)
    opt = nabwa.gap_init_opt()
    hits, _ = ix.cal_sa_reg_gap(opt, seq, rseq, off, per_read=True)
    ref.ref_index_load.restype = C.c_void_p
    ref.ref_index_load.argtypes = [C.c_char_p, C.c_int]
    rix = C.c_void_p(ref.ref_index_load(fa.encode(), 1))
    yield dict(ix=ix, ref=ref, rix=rix, opt=opt, seq=seq, rseq=rseq, off=off, full=full, hits=hits, n=len(reads))
    ix.close()


def reference_chain(w, cached):
    """posn_pair in order, then finish_pair over all pairs with one cache (or none), through the reference's functions"""
    ref, rix, off, seq, rseq, hits, n = w["ref"], w["rix"], w["off"], w["seq"], w["rseq"], w["hits"], w["n"]
    P = C.c_void_p
    ref.ref_pe_new.restype = P
    ref.ref_pe_new.argtypes = [C.c_int]
    ref.ref_pe_set.argtypes = [P, C.c_int, C.c_int, C.c_int, P, P, C.c_int, P]
    ref.ref_pe_posn.argtypes = [P, P, P]
    ref.ref_pe_finish_cached.argtypes = [P, P, P, P, P]
    ref.ref_poscache_new.restype = P
    ref.ref_poscache_free.argtypes = [P]
    ref.ref_pe_get.argtypes = [P, C.c_int, C.c_int, P, P, C.c_char_p, C.c_int, P]
    ref.ref_pe_free.argtypes = [P]
    copt = T.GapOpt()
    C.memmove(C.byref(copt), C.byref(w["opt"]), 64)
    b = P(ref.ref_pe_new(n))
    for i in range(2 * n):
        r = np.ascontiguousarray(np.asarray(hits[i], nabwa.ALN_DT))
        ref.ref_pe_set(b, i // 2, i % 2, int(off[i + 1] - off[i]), T.ptr(np.ascontiguousarray(seq[off[i]:off[i + 1]])),
                       T.ptr(np.ascontiguousarray(rseq[off[i]:off[i + 1]])), len(r), T.ptr(r))
    ref.ref_seed48(11)
    ref.ref_pe_posn(b, rix, C.byref(copt))
    iiv = (C.c_double * 6)(400.0, 40.0, 1e-5, 200, 600, 700)
    cache = P(ref.ref_poscache_new()) if cached else None
    ref.ref_pe_finish_cached(b, rix, C.byref(copt), iiv, cache)
    out = []
    f = np.zeros(17, np.int64); cg = np.zeros(256, np.uint16); mdb = C.create_string_buffer(1024); mu = np.zeros(21 * 16, np.int64)
    for i in range(2 * n):
        ref.ref_pe_get(b, i // 2, i % 2, T.ptr(f), T.ptr(cg), mdb, 1024, T.ptr(mu))
        out.append((f.copy(), cg[:int(f[13])].copy(), mdb.value))
    if cached:
        ref.ref_poscache_free(cache)
    ref.ref_pe_free(b)
    return out


def ours(w, cached, pieces):
    """nabwa_pe_posn over all pairs, then nabwa_pe_finish_cached piece by piece in record order with one cache (or NULL)"""
    ix, opt, off, n = w["ix"], w["opt"], w["off"], w["n"]
    L = nabwa.lib()
    P = C.c_void_p
    L.nabwa_poscache_create.restype = P
    L.nabwa_poscache_destroy.argtypes = [P]
    L.nabwa_poscache_size.restype = C.c_int64
    L.nabwa_poscache_size.argtypes = [P]
    L.nabwa_pe_finish_cached.argtypes = [P, P, P, P, C.c_int, P, P, P, P, P, P, P, P, P]
    recs, _ = ix.pe_posn(opt, off, w["full"], w["hits"], nabwa.srand48_state(11))
    ii = nabwa.IsizeInfo(400.0, 40.0, 1e-5, 200, 600, 700)
    po = nabwa.pe_opt_default()
    cache = P(L.nabwa_poscache_create()) if cached else None
    n_aln = np.array([len(h) for h in w["hits"]], np.int32)
    bounds = [n * k // pieces for k in range(pieces + 1)]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        o = np.ascontiguousarray(off[2 * lo:2 * hi + 1] - off[2 * lo])
        pad = np.zeros(16, np.uint8)
        sq = np.ascontiguousarray(np.concatenate([w["seq"][off[2 * lo]:off[2 * hi]], pad]))
        rq = np.ascontiguousarray(np.concatenate([w["rseq"][off[2 * lo]:off[2 * hi]], pad]))
        rows = np.ascontiguousarray(np.concatenate([np.asarray(h, nabwa.ALN_DT) for h in w["hits"][2 * lo:2 * hi]] + [np.zeros(0, nabwa.ALN_DT)]))
        na = np.ascontiguousarray(n_aln[2 * lo:2 * hi])
        sub = (nabwa.PeRec * (2 * (hi - lo))).from_buffer(recs, C.sizeof(nabwa.PeRec) * 2 * lo)
        rc = L.nabwa_pe_finish_cached(ix._h, C.byref(opt), C.byref(po), C.byref(ii), hi - lo, T.ptr(o), T.ptr(sq), T.ptr(rq), T.ptr(na), T.ptr(rows),
                                      sub, None, None, cache)
        assert rc == 0, L.nabwa_last_error()
    size = L.nabwa_poscache_size(cache) if cached else 0
    if cached:
        L.nabwa_poscache_destroy(cache)
    return recs, size


def compare(recs, want):
    diff = 0
    for r, (f, cg, md) in enumerate(want):
        s = recs[r].se
        assert s.type == f[0], r
        if s.type == 0:
            continue
        bridging = bool(s.flag & 4)
        got = [s.type, s.strand, s.n_mm, s.n_gapo, s.n_gape, s.score, s.sa, s.c1, s.c2, s.pos, s.mapQ if not bridging else int(f[10]), s.seQ]
        assert got == [int(x) for x in f[:12]], (r, got, f[:12].tolist())
        assert s.n_cigar == f[13] and list(s.cigar[:s.n_cigar]) == list(cg), r
        assert s.nm == f[14] and s.md == md, r
        assert (recs[r].extra_flag & 0xff) == (f[12] & 0xff), r


@pytest.mark.parametrize("pieces", [1, 3])
def test_one_cache_over_the_file_gives_the_sequential_reference_records(world, pieces):
    want = reference_chain(world, True)
    recs, size = ours(world, True, pieces)
    compare(recs, want)
    assert size >= 1                                   # wide rows did enter the cache


def test_without_a_cache_every_row_stands_for_itself(world):
    compare(ours(world, False, 2)[0], reference_chain(world, False))


def test_the_cache_changes_records(world):
    """the fixture is worth its name: with reads of several lengths on one wide row, cached and uncached chains of the REFERENCE differ"""
    a, b = reference_chain(world, True), reference_chain(world, False)
    n_diff = sum(1 for x, y in zip(a, b) if x[0][9] != y[0][9] or x[2] != y[2])
    assert n_diff >= 5, n_diff
