"""The index AT SIZE from files in the reference's on-disk layout: the synthetic GRCh38-sized
FM-indexes bench.py builds on the GPU are written out as <prefix>.bwt / .rbwt / .sa / .rsa / .pac / .ann / .amb (bwtio.c:161-204,
bntseq.c:63-117,240-250) -- 240 contigs of unequal lengths, 3000 ambiguity holes -- and loaded with nabwa_index_load(prefix, with_sa,
with_pac): the search of the headline reads gives the rows (checksum) the index made from the same arrays in memory gives, the
annotation comes back as written, and a sample of reads placed on contig borders and holes goes through the finishing chain against
the compiled reference loading the SAME files (bns_restore, bwt_restore_*).  Falls back to 256 Mbp where /tmp is short of room."""
import ctypes as C
import importlib
import os
import shutil

import numpy as np
import pytest

import nabwa_testlib as T

nabwa = importlib.import_module("network-aware-bwa_amd")
synth = importlib.import_module("network-aware-bwa_amd.synth")
pytestmark = pytest.mark.gpu


def pack_pac(d_text, n):
    t = d_text.to_host(np.uint8)[:n]
    pad = (-n) % 4
    if pad:
        t = np.concatenate([t, np.zeros(pad, np.uint8)])
    t = t.reshape(-1, 4)
    return np.ascontiguousarray((t[:, 0] << 6) | (t[:, 1] << 4) | (t[:, 2] << 2) | t[:, 3]).astype(np.uint8)


def test_genome_sized_index_from_reference_format_files(tmp_path_factory):
    n = int(os.environ.get("NABWA_TEST_GENOME", 3_099_734_149))
    root = os.environ.get("NABWA_TEST_TMP", "/tmp")
    if shutil.disk_usage(root).free < 3 * n:                      # ~1.5 bytes per base on disk, with room to spare
        n = 256_000_000
    d = os.path.join(root, "nabwa_index_files_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, "synth")
    try:
        run(n, prefix)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run(n, prefix):
    n_reads = 2_000_000 if n > 1_000_000_000 else 300_000
    d_text = synth.synth_text(n, 20261004, n_dup=2000, dup_len=5000, device=0)
    parts = [synth.build_index(d_text, n, rev, 32, True, device=0) for rev in (0, 1)]
    seq, rseq, off = synth.synth_reads(d_text, n, n_reads, 100, 2000, 0, 2, device=0)
    opt = nabwa.gap_init_opt()
    # ---- the index from the arrays in memory: the rows to expect
    ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]), (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]),
                                 device=0, device_ptrs=True)
    b = nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)
    b.run(); b.sync()
    want_sum = b.checksum()
    b.close(); ix.close()
    # ---- the same arrays as files of the reference's formats
    for t, (bw, nbw, sa, nsa) in enumerate(parts):
        bw.to_host(np.uint32, nbw).tofile(prefix + (".rbwt" if t else ".bwt"))        # primary, L2[1..4], then the Occ-interleaved BWT words (bwtio.c:184-204)
        sa.to_host(np.uint32, nsa).tofile(prefix + (".rsa" if t else ".sa"))          # primary, 4 skipped words, sa_intv, seq_len, then the samples (bwtio.c:161-182)
        bw.free(); sa.free()
    pac = pack_pac(d_text, n)
    d_text.free()
    with open(prefix + ".pac", "wb") as f:                                            # bntseq.c:240-250: the packed bases, a zero byte when they end on a byte border, the count of bases in the last byte
        f.write(pac.tobytes())
        if n % 4 == 0:
            f.write(b"\0")
        f.write(bytes([n % 4]))
    rng = np.random.default_rng(8)
    n_ctg = 240
    cuts = np.sort(rng.choice(np.arange(1000, n - 1000), n_ctg - 1, replace=False))
    offs = np.concatenate([[0], cuts]).astype(np.int64)
    lens = np.diff(np.concatenate([offs, [n]])).astype(np.int64)
    assert lens.max() < 2**31
    names = ["ctg%03d" % i for i in range(n_ctg)]
    hole_off = np.sort(rng.choice(np.arange(5000, n - 5000), 3000, replace=False)).astype(np.int64)
    hole_len = rng.integers(1, 2000, 3000).astype(np.int64)
    hole_len = np.minimum(hole_len, np.diff(np.concatenate([hole_off, [n]])) - 1)    # holes do not overlap
    ctg_of_hole = np.searchsorted(offs, hole_off, side="right") - 1
    with open(prefix + ".ann", "w") as f:                                             # bns_dump (bntseq.c:63-75)
        f.write("%d %d %u\n" % (n, n_ctg, 11))
        for i in range(n_ctg):
            f.write("%d %s a synthetic contig\n" % (i, names[i]) if i % 3 else "%d %s\n" % (i, names[i]))
            f.write("%d %d %d\n" % (offs[i], lens[i], int((ctg_of_hole == i).sum())))
    with open(prefix + ".amb", "w") as f:
        f.write("%d %d %d\n" % (n, n_ctg, 3000))
        for o, l in zip(hole_off, hole_len):
            f.write("%d %d N\n" % (o, l))
    # ---- loaded from the files
    ix = nabwa.Index.load(prefix, 0, True, True)
    L = nabwa.lib()
    assert ix.seq_len(0) == n and ix.seq_len(1) == n
    assert L.nabwa_index_n_contigs(ix._h) == n_ctg
    L.nabwa_index_contig.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    L.nabwa_index_reference_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    for i in (0, 1, 119, n_ctg - 1):
        nm, o, l = C.create_string_buffer(64), C.c_int64(), C.c_int32()
        assert L.nabwa_index_contig(ix._h, i, nm, 64, C.byref(o), C.byref(l)) == 0
        assert (nm.value.decode(), o.value, l.value) == (names[i], int(offs[i]), int(lens[i]))
    lp, sd = C.c_int64(), C.c_uint32()
    assert L.nabwa_index_reference_info(ix._h, C.byref(lp), C.byref(sd)) == 0 and (lp.value, sd.value) == (n, 11)
    b = nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)
    b.run(); b.sync()
    assert b.checksum() == want_sum                          # the same rows for every one of the reads
    hits_head = b.fetch_flat()
    b.close()
    # ---- a sample through the finishing chain against the reference loading the same files: reads cut where contigs meet and where holes lie
    ref = T.load_ref()
    if ref is None:
        ix.close()
        pytest.skip("oracle/_ref did not travel: rows and annotation checked, the chain against the reference not")
    text_at = lambda p, l: np.array([(pac[(p + j) >> 2] >> ((~(p + j) & 3) << 1)) & 3 for j in range(l)], np.uint8)
    starts = []
    for c in cuts[:60]:
        starts += [int(c) - 100, int(c) - 37, int(c)]          # inside a contig's end, across the border, at the next one's start
    for o, l in list(zip(hole_off, hole_len))[:120]:
        starts += [int(o) - 50, int(o) + int(l) - 30]           # into a hole, out of one
    starts += [int(x) for x in rng.integers(0, n - 100, 400)]
    reads = []
    for k, p in enumerate(starts):
        r = text_at(p, 100)
        if k % 4 == 1:
            r = r.copy(); r[int(rng.integers(10, 90))] ^= 1      # a substitution
        if k % 9 == 2:
            c0 = int(rng.integers(20, 80))
            r = np.concatenate([r[:c0], r[c0 + 1:], text_at(p + 100, 1)])      # a deleted base
        if k % 2:
            r = (3 - r)[::-1]                                    # the other strand
        reads.append(r)
    s_seq = np.concatenate([r[::-1] for r in reads]).astype(np.uint8)              # bwa_seq_t.seq: the read reversed; rseq: its complement
    s_rseq = (3 - s_seq).astype(np.uint8)
    s_off = np.arange(len(reads) + 1, dtype=np.int64) * 100
    hits, _ = ix.cal_sa_reg_gap(opt, s_seq, s_rseq, s_off, per_read=True)
    full = np.full(len(reads), 100, np.int32)
    recs, _ = ix.se_finish(opt, s_seq, s_rseq, s_off, full, hits, 3, nabwa.srand48_state(11))
    ref.ref_index_load.restype = C.c_void_p
    ref.ref_index_load.argtypes = [C.c_char_p, C.c_int]
    rix = C.c_void_p(ref.ref_index_load(prefix.encode(), 1))
    P = C.c_void_p
    ref.ref_se_chain_mt.argtypes = [P, P, C.c_int, C.c_int, P, P, P, P, P, C.c_int, P, P, P, C.c_int, P]
    ref.ref_pac2real.argtypes = [P, C.c_int64, C.c_int, P, P]
    copt = T.GapOpt(); C.memmove(C.byref(copt), C.byref(opt), 64)
    na = np.array([len(h) for h in hits], np.int32)
    rows = np.ascontiguousarray(np.concatenate([np.asarray(h, nabwa.ALN_DT) for h in hits] + [np.zeros(0, nabwa.ALN_DT)]))
    f = np.zeros((len(reads), 16), np.int64); cg = np.zeros((len(reads), 64), np.uint16); md = np.zeros((len(reads), 256), np.uint8)
    secs = (C.c_double * 2)()
    ref.ref_seed48(11)
    ref.ref_se_chain_mt(rix, C.byref(copt), 3, len(reads), T.ptr(s_off), T.ptr(s_seq), T.ptr(s_rseq), T.ptr(na), T.ptr(rows), 4, T.ptr(f), T.ptr(cg), T.ptr(md), 256, secs)
    n_map = n_hole = n_bridge = 0
    for i in range(len(reads)):
        s, w = recs[i], f[i]
        assert s.type == w[0], i
        if s.type == 0:
            continue
        n_map += 1
        bridging = bool(s.flag & 4)
        n_bridge += bridging
        assert [s.strand, s.n_mm, s.n_gapo, s.n_gape, s.score, s.sa, s.c1, s.c2, s.pos] == [int(x) for x in w[1:10]], i
        assert bridging or s.mapQ == w[10], i
        assert s.n_cigar == w[12] and list(s.cigar[:s.n_cigar]) == list(cg[i, :s.n_cigar]) and s.nm == w[13], i
        assert s.md == bytes(md[i]).split(b"\0", 1)[0], i           # N's of the holes restored in MD (bwase.c:243-268)
        sid, o = C.c_int32(), C.c_int64()
        ln = 100 if s.n_cigar == 0 else sum((c & 0x3fff) for c in s.cigar[:s.n_cigar] if (c >> 14) in (0, 2))
        nn = ref.ref_pac2real(rix, int(s.pos), ln, C.byref(sid), C.byref(o))
        assert (s.seqid, s.rpos, s.nn) == (sid.value, int(s.pos) - o.value + 1, nn), (i, s.seqid, s.rpos, s.nn, sid.value, o.value, nn)
        n_hole += nn > 0
    assert n_map > 0.9 * len(reads) and n_hole >= 100 and n_bridge >= 20
    ix.close()
