"""The batching front-end (bam_batch.hip): BAM records in -> BAM records out, checked field by field against the SAM the
reference's `samse` / `sampe` printed for the same reads (tests/golden/*.sam) -- flags, contig, position, MAPQ, CIGAR, mate
fields, template length, SEQ / QUAL as stored (reverse-complemented for reverse-strand hits), and every tag.

bam2bam.c itself cannot be compiled in the build container (it needs <zmq.h>), so bytes only bam2bam writes -- the `bin`
field, tag order and integer tag types, the erased input tags -- are PARITY UNPINNED: they follow a reading of
bam2bam.c:430-593 / bwaseqio.c:413-464 and are asserted here as that reading says, not against a run of the reference."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import bamlib as B
import nabwa_testlib as T

nabwa = importlib.import_module("network-aware-bwa_amd")
pytestmark = pytest.mark.gpu
P = C.c_void_p


def bind():
    L = nabwa.lib()
    L.nabwa_isize_table_create.restype = P
    L.nabwa_isize_table_create.argtypes = [C.c_double, C.c_int64]
    L.nabwa_isize_table_destroy.argtypes = [P]
    L.nabwa_isize_table_infer_all.argtypes = [P]
    L.nabwa_isize_table_get.argtypes = [P, C.c_char_p, P]
    L.nabwa_isize_table_encode.restype = C.c_int64
    L.nabwa_isize_table_encode.argtypes = [P, P, C.c_int64]
    L.nabwa_isize_table_decode.argtypes = [P, P, C.c_int64]
    L.nabwa_isize_table_merge.argtypes = [P, P]
    L.nabwa_bam_batch_create.argtypes = [P, P, P, C.c_int, P, P, P]
    L.nabwa_bam_batch_pass1.argtypes = [P, P, P]
    L.nabwa_bam_batch_pass2.argtypes = [P, P, P, P]
    L.nabwa_bam_batch_output.argtypes = [P, P, C.c_int64, P, P]
    L.nabwa_bam_batch_destroy.argtypes = [P]
    return L


def chk(L, rc):
    assert rc == 0, L.nabwa_last_error().decode()


def toy_ann():
    lines = open(T.TOY + ".ann").read().split("\n")
    return int(lines[0].split()[0]), [l.split()[1] for l in lines[1::2] if l]


def run_batches(L, ix, opt, records_per_batch, table_blob=None):
    """the two passes over several batches with one RNG stream and one insert-size table, as a front-end drives them"""
    l_pac, contigs = toy_ann()
    g = nabwa.GapOpt(); C.memmove(C.byref(g), C.byref(opt), 64)
    po = nabwa.pe_opt_default()
    tab = P(L.nabwa_isize_table_create(po.ap_prior, l_pac))
    st = C.c_uint64(nabwa.srand48_state(11))
    batches = []
    for recs in records_per_batch:
        buf, off = B.pack(recs)
        h = P()
        chk(L, L.nabwa_bam_batch_create(ix._h, C.byref(g), C.byref(po), len(recs), T.ptr(buf), T.ptr(off), C.byref(h)))
        chk(L, L.nabwa_bam_batch_pass1(h, C.byref(st), tab))
        batches.append((h, buf, off))
    chk(L, L.nabwa_isize_table_infer_all(tab))                       # the barrier between the passes (infer_all_isizes)
    if table_blob is not None:
        chk(L, L.nabwa_isize_table_decode(tab, T.ptr(table_blob), len(table_blob)))
    out = []
    tot, mp = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
    for h, buf, off in batches:
        chk(L, L.nabwa_bam_batch_pass2(h, tab, tot, mp))
        nb = C.c_int64()
        oo = np.zeros(len(off), np.int64)
        L.nabwa_bam_batch_output(h, None, 0, T.ptr(oo), C.byref(nb))
        ob = np.zeros(max(nb.value, 1), np.uint8)
        chk(L, L.nabwa_bam_batch_output(h, T.ptr(ob), nb.value, T.ptr(oo), C.byref(nb)))
        out += B.decode(ob, oo, contigs)
        L.nabwa_bam_batch_destroy(h)
    ii = nabwa.IsizeInfo()
    L.nabwa_isize_table_get(tab, b"", C.byref(ii))
    L.nabwa_isize_table_destroy(tab)
    return out, ii, (list(tot), list(mp))


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


@pytest.mark.parametrize("name", ["default", "adna", "q20"])
def test_se_bam_records_match_the_reference_sam(name):
    L = bind()
    opt, _ = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    sam = T.parse_sam(os.path.join(T.GOLDEN, "se_%s.sam" % name))
    ix = nabwa.Index.load(T.TOY, 0, True, True)
    junk = B.tag_z("RG", "lib1") + B.tag_a("XT", "U") + B.tag_i("NM", 7) + B.tag_z("MD", "100") + B.tag_z("ZZ", "kept") + B.tag_i("X0", 1) + B.tag_z("XA", "x;")
    recs = []
    for i, (n, s, q) in enumerate(reads):
        if i % 5 == 3:       # an input record that carries the reverse-strand flag holds the reverse complement (undone by bam1_to_seq)
            recs.append(B.make_record(n, revcomp(s), q[::-1], 4 | 16, junk))
        else:
            recs.append(B.make_record(n, s, q, 4, junk if i % 2 else b""))
    out, _, _ = run_batches(L, ix, opt, [recs[:250], recs[250:]])          # two batches: the RNG stream runs through
    assert len(out) == len(sam)
    for i, (g, w) in enumerate(zip(out, sam)):
        assert g["name"] == w["name"]
        if i % 5 == 3 and w["rname"] == "*":
            # an unmapped read keeps what the input record said about its strand, and its bases as they were stored
            # (bam2bam.c:573-592 clears the pairing flags only); `samse` never saw that flag: its input was FASTQ
            w = dict(w, flag=w["flag"] | 16, seq=revcomp(w["seq"]), qual=w["qual"][::-1])
        assert g["flag"] == w["flag"], (w["name"], g["flag"], w["flag"])
        assert (g["rname"], g["pos"], g["mapq"], g["cigar"]) == (w["rname"], w["pos"], w["mapq"], w["cigar"]), w["name"]
        assert (g["seq"], g["qual"]) == (w["seq"], w["qual"]), w["name"]
        assert (g["rnext"], g["pnext"], g["tlen"]) == ("*", 0, 0)
        t = dict(g["tags"])
        kept = {k: t.pop(k) for k in ("RG", "ZZ") if k in t}
        assert kept == ({"RG": "lib1", "ZZ": "kept"} if (i % 5 == 3 or i % 2) else {}), w["name"]       # the caller's tags stay, the aligner's are regenerated
        assert t == w["tags"], (w["name"], t, w["tags"])
        if w["rname"] == "*":
            assert (g["tid"], g["bin"]) == (-1, 0)
    ix.close()


def pe_records():
    fq = [T.read_fastq(os.path.join(T.GOLDEN, "reads_pe_%d.fq" % e)) for e in (1, 2)]
    recs = []
    for i in range(len(fq[0])):
        a, b = fq[0][i], fq[1][i]
        name = a[0][:-2] if a[0].endswith("/1") else a[0]
        r1 = B.make_record(name, a[1], a[2], 1 | 4 | 8 | 64, B.tag_i("AM", 3))
        r2 = B.make_record(name, b[1], b[2], 1 | 4 | 8 | 128)
        recs += [r2, r1] if i % 7 == 2 else [r1, r2]                # read 2 first now and then: swapped back (bwaseqio.c:362-365)
    return recs, len(fq[0])


def test_pe_bam_records_match_the_reference_sam():
    L = bind()
    opt, _ = T.read_sai(os.path.join(T.GOLDEN, "pe_1.sai"))
    sam = T.parse_sam(os.path.join(T.GOLDEN, "pe_default.sam"))
    v = np.load(os.path.join(T.GOLDEN, "vectors_pe_chain.npz"))
    ix = nabwa.Index.load(T.TOY, 0, True, True)
    recs, n_pairs = pe_records()
    # (1) bam2bam's own route: histogram over the positioned pairs -> infer_all_isizes; the estimate is the reference's
    out, ii, (tot, mp) = run_batches(L, ix, opt, [recs[:300], recs[300:]])
    iv = v["ii_hist"]
    assert (ii.avg, ii.std, ii.ap_prior, ii.low, ii.high, ii.high_bayesian) == tuple(iv)
    # (2) under the estimate `sampe` made for these pairs (sent the way a worker gets it: decode_iinfo's blob), every field of
    #     the SAM `sampe` printed
    iv = v["ii_sampe"]
    blob = np.frombuffer(b"\0" + bytes(8) + bytes(nabwa.IsizeInfo(iv[0], iv[1], iv[2], int(iv[3]), int(iv[4]), int(iv[5]))), np.uint8).copy()
    out, ii, (tot, mp) = run_batches(L, ix, opt, [recs[:300], recs[300:]], table_blob=blob)
    assert len(out) == len(sam) == 2 * n_pairs
    for r, (g, w) in enumerate(zip(out, sam)):
        assert g["name"] == w["name"].split("/")[0]
        assert g["flag"] == w["flag"], (r, g["flag"], w["flag"])
        assert (g["rname"], g["pos"], g["rnext"], g["pnext"], g["tlen"]) == (w["rname"], w["pos"], w["rnext"], w["pnext"], w["tlen"]), r
        assert (g["mapq"], g["cigar"]) == (w["mapq"], w["cigar"]), r
        assert (g["seq"], g["qual"]) == (w["seq"], w["qual"]), r
        assert g["tags"] == w["tags"], (r, g["tags"], w["tags"])
    assert mp[0] == sum(1 for w in sam if w["tags"].get("XT") == "M") and mp[1] == 0
    ix.close()


def test_pairs_of_two_read_groups_each_under_its_own_estimate():
    """pairs of two read groups interleaved in one batch: pass 2 gathers each group's pairs and finishes them under that group's
    estimate (finish_pair looks it up by bam_get_rg, bam2bam.c:712-715).  Group "a" gets the estimate `sampe` made, group "b" none:
    every pair must come out as in a run of its own kind (pass 1 and its random stream do not depend on the group)"""
    L = bind()
    opt, _ = T.read_sai(os.path.join(T.GOLDEN, "pe_1.sai"))
    v = np.load(os.path.join(T.GOLDEN, "vectors_pe_chain.npz"))
    ix = nabwa.Index.load(T.TOY, 0, True, True)
    fq = [T.read_fastq(os.path.join(T.GOLDEN, "reads_pe_%d.fq" % e)) for e in (1, 2)]

    def records(rg_of):
        recs = []
        for i in range(len(fq[0])):
            a, b = fq[0][i], fq[1][i]
            name = a[0][:-2] if a[0].endswith("/1") else a[0]
            tag = B.tag_z("RG", rg_of(i))
            recs += [B.make_record(name, a[1], a[2], 1 | 4 | 8 | 64, tag), B.make_record(name, b[1], b[2], 1 | 4 | 8 | 128, tag)]
        return recs

    iv = v["ii_sampe"]
    est = bytes(nabwa.IsizeInfo(iv[0], iv[1], iv[2], int(iv[3]), int(iv[4]), int(iv[5])))
    blob = np.frombuffer(b"a\0" + bytes(8) + est + b"b\0" + bytes(8) + bytes(nabwa.IsizeInfo()), np.uint8).copy()
    fields = lambda out: [(r["name"], r["flag"], r["rname"], r["pos"], r["mapq"], r["cigar"], r["rnext"], r["pnext"], r["tlen"],
                           {k: x for k, x in r["tags"].items() if k != "RG"}) for r in out]
    all_a = fields(run_batches(L, ix, opt, [records(lambda i: "a")], table_blob=blob)[0])
    all_b = fields(run_batches(L, ix, opt, [records(lambda i: "b")], table_blob=blob)[0])
    mixed = fields(run_batches(L, ix, opt, [records(lambda i: "ab"[i % 2])], table_blob=blob)[0])
    assert sum(x != y for x, y in zip(all_a, all_b)) > 20                      # the estimate matters
    for i in range(len(fq[0])):
        want = all_a if i % 2 == 0 else all_b
        assert mixed[2 * i:2 * i + 2] == want[2 * i:2 * i + 2], i
    ix.close()


def test_isize_table_blob_and_merge():
    L = bind()
    a, b = P(L.nabwa_isize_table_create(1e-5, 100000)), P(L.nabwa_isize_table_create(1e-5, 100000))
    ii = nabwa.IsizeInfo(401.5, 39.25, 1e-5, 250, 560, 641)
    blob = np.frombuffer(b"grp1\0" + bytes(8) + bytes(ii) + b"\0" + bytes(8) + bytes(nabwa.IsizeInfo()), np.uint8).copy()
    chk(L, L.nabwa_isize_table_decode(a, T.ptr(blob), len(blob)))
    got = nabwa.IsizeInfo()
    assert L.nabwa_isize_table_get(a, b"grp1", C.byref(got)) == 0 and bytes(got) == bytes(ii)
    assert L.nabwa_isize_table_get(a, b"nope", C.byref(got)) == 1 and got.avg == 0
    n = L.nabwa_isize_table_encode(a, None, 0)
    back = np.zeros(n, np.uint8)
    assert L.nabwa_isize_table_encode(a, T.ptr(back), n) == n and sorted(back.tobytes().split(b"\0")[0:1]) in ([b""], [b"grp1"])
    chk(L, L.nabwa_isize_table_decode(b, T.ptr(back), n))
    assert L.nabwa_isize_table_get(b, b"grp1", C.byref(got)) == 0 and bytes(got) == bytes(ii)
    chk(L, L.nabwa_isize_table_merge(a, b))
    L.nabwa_isize_table_destroy(a); L.nabwa_isize_table_destroy(b)


def test_awkward_records_go_through_both_passes():
    """an empty batch; a record without bases; a read of nothing but N; a read shorter than any seed: none of them may stop a batch,
    and the ordinary reads around them keep the answers the reference's samse gives them.  A record flagged as paired whose mate is not
    the next record is the reference's error too (read_bam_pair returns -2 unless broken input is allowed, bwaseqio.c:378-396)."""
    L = bind()
    opt, _ = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:40]
    sam = T.parse_sam(os.path.join(T.GOLDEN, "se_default.sam"))[:40]
    ix = nabwa.Index.load(T.TOY, 0, True, True)
    normal = [B.make_record(n, s, q, 4) for n, s, q in reads]
    odd = [B.make_record("empty", "", "", 4), B.make_record("all_n", "N" * 60, "I" * 60, 4), B.make_record("tiny", "ACGTACG", "IIIIIII", 4)]
    out, _, _ = run_batches(L, ix, opt, [[], normal[:20] + odd[:2], odd[2:] + normal[20:], []])
    names = [o["name"] for o in out]
    assert names == [r[0] for r in reads[:20]] + ["empty", "all_n", "tiny"] + [r[0] for r in reads[20:]]
    by = {o["name"]: o for o in out}
    for nm in ("empty", "all_n"):
        assert by[nm]["flag"] & 4 and by[nm]["rname"] == "*" and by[nm]["cigar"] == "*", (nm, by[nm])
    assert by["tiny"]["seq"] in ("ACGTACG", "CGTACGT") and by["tiny"]["mapq"] == 0          # seven bases occur in many places: placed somewhere, mapping quality 0
    assert by["empty"]["seq"] in ("", "*") and by["all_n"]["seq"] == "N" * 60
    for g, s in zip([o for o in out if o["name"] not in ("empty", "all_n", "tiny")], sam):
        # the drand48 stream differs from samse's run once extra records sit between the reads: only reads with one best place are compared
        if s["tags"].get("X0", 1) == 1 and not (s["flag"] & 4):
            assert (g["rname"], g["pos"], g["cigar"], g["tags"].get("NM")) == (s["rname"], s["pos"], s["cigar"], s["tags"].get("NM")), s["name"]
    # the lone mate
    g = nabwa.GapOpt(); C.memmove(C.byref(g), C.byref(opt), 64)
    po = nabwa.pe_opt_default()
    buf, off = B.pack([B.make_record("widow", reads[0][1], reads[0][2], 1 | 4 | 8 | 64), normal[1]])
    h = P()
    assert L.nabwa_bam_batch_create(ix._h, C.byref(g), C.byref(po), 2, T.ptr(buf), T.ptr(off), C.byref(h)) == nabwa.EINVAL
    assert b"lone mate" in L.nabwa_last_error()
    ix.close()
