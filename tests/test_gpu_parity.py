"""GPU parity (run with -m gpu on an MI355X): the HIP path through the C ABI against
 (a) the golden vectors the reference itself produced and (b) the CPU oracle on fresh seeded inputs.
Integer work: everything is compared bit-exact."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import nabwa_testlib as T

pytestmark = pytest.mark.gpu
nabwa = importlib.import_module("network-aware-bwa_amd")

SAI_SETS = ["default", "adna", "n3", "e3", "loggap", "k1R5", "i2", "q20", "m64", "nonstop"]


@pytest.fixture(scope="module")
def gix():
    ix = nabwa.Index.load(T.TOY, 0, True)
    yield ix
    ix.close()


@pytest.fixture(scope="module")
def olib():
    return T.load_oracle()


@pytest.fixture(scope="module")
def oix(olib):
    return T.OracleIndex(olib)


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(T.GOLDEN, "vectors.npz"))


def to_gap_opt(o):
    g = nabwa.GapOpt()
    C.memmove(C.byref(g), C.byref(o), 64)
    return g


def test_occ4_golden(gix, vec):
    for which in (0, 1):
        got = gix.occ4(which, vec["occ_k%d" % which])
        assert np.array_equal(got, vec["occ4_v%d" % which])
        assert np.array_equal(gix.occ4(which, vec["p_k%d" % which]), vec["p_ck%d" % which])
        assert np.array_equal(gix.occ4(which, vec["p_l%d" % which]), vec["p_cl%d" % which])


def test_occ4_every_row(gix, olib, oix):
    """all rows of both toy indexes, incl. -1, primary and seq_len"""
    buf = (C.c_uint32 * 4)()
    for which in (0, 1):
        n = gix.seq_len(which)
        ks = np.concatenate([np.arange(0, n + 1, dtype=np.uint32), np.array([0xFFFFFFFF], np.uint32)])
        got = gix.occ4(which, ks)
        step = 97
        for i in list(range(0, len(ks), step)) + [len(ks) - 2, len(ks) - 1]:
            olib.orc_occ4(oix.bwt(which), int(ks[i]), buf)
            assert list(got[i]) == list(buf), (which, int(ks[i]))
        # differences between consecutive rows are 0/1 and sum to one base per row (except '$')
        d = np.diff(got[:-1].astype(np.int64), axis=0)
        assert d.min() >= 0 and d.sum(axis=1).max() <= 1


def test_sa_lookup_golden(gix, vec):
    for which in (0, 1):
        k = vec["sa_k%d" % which]
        got = gix.sa_lookup(np.full(len(k), which, np.uint8), k)
        assert np.array_equal(got, vec["sa_v%d" % which])


def test_sa_is_a_permutation(gix):
    """size-independent property: SA over all rows 1..n is a permutation of 0..n-1"""
    for which in (0, 1):
        n = gix.seq_len(which)
        k = np.arange(1, n + 1, dtype=np.uint32)
        got = gix.sa_lookup(np.full(n, which, np.uint8), k)
        assert np.array_equal(np.sort(got), np.arange(n, dtype=np.uint32))


@pytest.mark.parametrize("name", SAI_SETS)
def test_sai_parity(gix, name):
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se_head.fq" if name == "nonstop" else "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads, opt.trim_qual)
    got, _ = gix.cal_sa_reg_gap(to_gap_opt(opt), seq, rseq, off, per_read=False)
    bad = [reads[i][0] for i in range(len(reads)) if got[i].tobytes() != gold[i].tobytes()]
    assert not bad, "GPU differs from the reference .sai for %d reads, e.g. %s" % (len(bad), bad[:5])


def random_reads(rng, n, genome, lens=(100,), err=0.01, indel=0.05):
    reads = []
    G = len(genome)
    for i in range(n):
        L = int(rng.choice(lens))
        if rng.random() < 0.05:
            s = "".join("ACGT"[x] for x in rng.integers(0, 4, L))
        else:
            p = int(rng.integers(0, G - L - 8))
            s = list(genome[p:p + L + 8])
            for j in range(L):
                if rng.random() < err:
                    s[j] = "ACGTN"[int(rng.integers(0, 5))]
            if rng.random() < indel:
                q = int(rng.integers(8, L - 8))
                if rng.random() < 0.5:
                    del s[q]
                else:
                    s.insert(q, "ACGT"[int(rng.integers(0, 4))])
            s = "".join(s[:L])
        if rng.random() < 0.5:
            s = s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))
        reads.append(("q%d" % i, s, "I" * len(s)))
    return reads


def toy_genome():
    seqs = []
    cur = []
    for line in open(T.TOY + ".fa"):
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur))
            cur = []
        else:
            cur.append(line.strip())
    seqs.append("".join(cur))
    return "".join(seqs)


@pytest.mark.parametrize("per_read", [False, True])
def test_random_reads_vs_oracle(gix, olib, oix, per_read):
    """fresh seeded reads of mixed length (ragged batch), both option-derivation modes, max_entries too"""
    rng = np.random.default_rng(7 + per_read)
    reads = random_reads(rng, 3000, toy_genome(), lens=(36, 50, 76, 100, 100, 100, 150), err=0.02)
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = T.default_opt()
    want, wmax = T.oracle_cal_sa_reg_gap(olib, oix.h, opt, seq, rseq, off, per_read=int(per_read), n_threads=8)
    got, gmax = gix.cal_sa_reg_gap(to_gap_opt(opt), seq, rseq, off, per_read=per_read)
    bad = [i for i in range(len(reads)) if got[i].tobytes() != want[i].tobytes()]
    assert not bad, "GPU differs from the oracle for %d reads, first %s" % (len(bad), reads[bad[0]])
    assert np.array_equal(gmax, wmax)


def test_edge_batches(gix):
    opt = nabwa.gap_init_opt()
    # empty batch
    got, _ = gix.cal_sa_reg_gap(opt, np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.int64))
    assert got == []
    # zero-length read between two normal ones; single read
    reads = [("a", "ACGTACGTACGTAGCTAGCTAGCATCGATCGATCGACTAGCTAGC", "I" * 45), ("e", "", ""),
             ("n", "N" * 50, "I" * 50)]
    seq, rseq, off, _ = T.encode_reads(reads)
    got, maxe = gix.cal_sa_reg_gap(opt, seq, rseq, off)
    assert len(got) == 3 and len(got[1]) == 0 and len(got[2]) == 0 and maxe[1] == 0 and maxe[2] == 0


def test_wide_pass_is_bit_exact(gix, olib, oix, monkeypatch):
    """force the first pass to overflow (tiny arena / hit list): the slot-reusing second pass must
    give the same answers"""
    monkeypatch.setenv("NABWA_CAP1", "48")
    monkeypatch.setenv("NABWA_ALNCAP1", "1")
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads)
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    b.run()
    n2 = b.sync()
    got, _ = b.fetch()
    cs = b.checksum()
    b.close()
    assert n2 > 100
    assert all(got[i].tobytes() == gold[i].tobytes() for i in range(len(reads)))
    monkeypatch.delenv("NABWA_CAP1")
    monkeypatch.delenv("NABWA_ALNCAP1")
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    b.run()
    assert b.sync() < 300           # only the reads with > NABWA_CAP1 pushes, > 16 hits or more trips than the hand-over budget (200 here: most of these reads occur exactly on neither strand)
    assert b.checksum() == cs          # device checksum is independent of which pass produced a row
    assert b.checksum()[1] == sum(len(g) for g in gold)
    b.close()


@pytest.mark.parametrize("env", [
    {"NABWA_CAP1": "48"},                                                    # straight to kernel D (one search per wavefront)
    {"NABWA_CAP1": "48", "NABWA_TIER_A": "1"},                               # the first-pass kernel once more with its largest arena, kernel D for the rest
    {"NABWA_CAP1": "16", "NABWA_DEEP_LANES": "3"},                           # rounds of three chains
    {"NABWA_CAP1": "16", "NABWA_DEEP_STAGE": "1"},                           # the smallest staging buffers: chains continue over rounds
    {"NABWA_CAP1": "16", "NABWA_DEEP_CAREFUL": "1"},                         # one pop per round
    {"NABWA_CAP1": "16", "NABWA_DEEP_PAGES": "6000", "NABWA_DEEP_WAVES_PER_CU": "1"},   # a pool that runs dry: the guaranteed pass finishes them
    {"NABWA_CAP1": "48", "NABWA_ALNCAP1": "1"},                              # the first pass fails on the hit lists too
], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_flagged_reads_through_kernel_d_are_bit_exact(gix, monkeypatch, env):
    """reads that outgrow the first pass go to kernel D (fm_deep_body.hpp; nabwa_api.hip: nabwa_batch_sync): speculative rounds
    of up to 64 chains, ordered commit, paged arenas -- the rows are the reference's, on the option set ancient-DNA pipelines
    use (deep searches), under every knob that changes how the rounds are cut"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_adna.sai"))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads)
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    b.run()
    n2 = b.sync()
    got, _ = b.fetch()
    assert b.last_deep_ms() > 0
    b.close()
    assert n2 > 100
    assert all(got[i].tobytes() == gold[i].tobytes() for i in range(len(reads)))


def test_rerun_is_idempotent(gix):
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads)
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    sums = []
    for _ in range(3):
        b.run()
        b.sync()
        sums.append(b.checksum())
    b.close()
    assert sums[0] == sums[1] == sums[2]


def test_unsupported_options_fail_loudly(gix):
    opt = nabwa.gap_init_opt()
    opt.max_gape = 300          # the reference's own score field (11 bits, bwtgap.c:58) could not hold this either
    reads = [("a", "ACGT" * 20, "I" * 80)]
    seq, rseq, off, _ = T.encode_reads(reads)
    with pytest.raises(nabwa.NabwaError) as e:
        gix.cal_sa_reg_gap(opt, seq, rseq, off)
    assert e.value.code == nabwa.EINVAL


def test_touch_counter_matches_oracle(gix, olib, oix):
    """the instrumented kernel counts the reference algorithm's Occ-bucket touches (roofline bytes)
    exactly as the CPU restatement's counters do"""
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    seq, rseq, off, _ = T.encode_reads(reads)
    ctr = T.Counters()
    T.oracle_cal_sa_reg_gap(olib, oix.h, opt, seq, rseq, off, per_read=0, counters=ctr)
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    ts, tw = b.count_touches()
    assert ts + tw == ctr.n_bucket and ts > 0 and tw > 0
    b.close()


def test_drop_in_on_reference_records(gix):
    """bwa_cal_sa_reg_gap on the reference's own bwa_seq_t records: the same array goes through the
    compiled reference (when oracle/_ref travelled) and through nabwa_bwa_cal_sa_reg_gap; every field the
    reference's post-conditions name (bwtaln.c:82-91,113) must agree, rows byte for byte."""
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:300]
    seq, rseq, off, _ = T.encode_reads(reads)
    n = len(reads)

    def make():
        arr = (nabwa.BwaSeq * n)()
        keep = []
        for i in range(n):
            s = np.ascontiguousarray(seq[off[i]:off[i + 1]]); r = np.ascontiguousarray(rseq[off[i]:off[i + 1]])
            keep += [s, r]
            L = len(s)
            arr[i].seq = s.ctypes.data; arr[i].rseq = r.ctypes.data
            arr[i].bits0 = L | (1 << 20) | (2 << 21)          # len, strand=1, type=REPEAT: must be reset
            arr[i].lenbits = L; arr[i].clip_len = L
            arr[i].sa = 12345; arr[i].c1c2seq = (7 << 56) | (5 << 28) | 9
        return arr, keep

    ours, k1 = make()
    gix.bwa_cal_sa_reg_gap(ours, n, to_gap_opt(opt))
    for i in range(n):
        rows = np.frombuffer(C.string_at(ours[i].aln, 16 * ours[i].n_aln), T.ALN_DT) if ours[i].n_aln else np.zeros(0, T.ALN_DT)
        assert rows.tobytes() == gold[i].tobytes(), reads[i][0]
        assert ours[i].sa == 0 and (ours[i].bits0 >> 21 & 3) == 0 and ours[i].c1c2seq == (7 << 56)
        assert (ours[i].bits0 & 0xfffff) == off[i + 1] - off[i] and (ours[i].bits0 >> 20 & 1) == 1
    ref = T.load_ref()
    if ref is not None:
        ref.ref_index_load.restype = C.c_void_p
        rix = ref.ref_index_load(T.TOY.encode(), 0)
        theirs, k2 = make()
        ref.bwa_cal_sa_reg_gap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        ref.bwa_cal_sa_reg_gap(C.c_void_p(rix), n, theirs, C.byref(opt))   # ref_index_t starts with bwt_t *bwt[2]
        for i in range(n):
            assert ours[i].n_aln == theirs[i].n_aln and ours[i].max_entries == theirs[i].max_entries, reads[i][0]
            assert ours[i].bits0 == theirs[i].bits0 and ours[i].sa == theirs[i].sa and ours[i].c1c2seq == theirs[i].c1c2seq
            if ours[i].n_aln:
                assert C.string_at(ours[i].aln, 16 * ours[i].n_aln) == C.string_at(theirs[i].aln, 16 * theirs[i].n_aln)


def test_text_mode_companions(gix):
    """what the index derives at load time (fm_index.hip): the text of both indexes equals the toy genome's .pac
    (forward / reversed), the inverse SA inverts the full SA, the full SA equals the reference's bwt_sa known answers,
    and the interval table holds the oracle's backward-search intervals"""
    n = gix.seq_len(0)
    pac = np.fromfile(T.TOY + ".pac", np.uint8)
    bases = ((pac[:, None] >> np.array([6, 4, 2, 0])) & 3).reshape(-1)[:n]        # first base in the top bits (bwtaln.h:33)
    assert (gix.export(0, 2, 0, n) == bases).all()
    assert (gix.export(1, 2, 0, n) == bases[::-1]).all()                          # .rbwt indexes the reversed text (bwtindex.c:116-143)
    v = np.load(os.path.join(T.GOLDEN, "vectors.npz"))
    for which in (0, 1):
        sa = gix.export(which, 0, 0, n + 1)
        isa = gix.export(which, 1, 0, n + 1)
        assert sa[0] == 0xffffffff and isa[n] == 0
        assert (isa[sa[1:]] == np.arange(1, n + 1)).all()
        assert (sa[v["sa_k%d" % which]] == v["sa_v%d" % which]).all()
    # interval table against a plain backward search with the reference's own rank answers (oracle)
    lib = T.load_oracle()
    oix = T.OracleIndex(lib)
    Tdepth = int(gix.export(0, 4, 0, 1)[0])
    assert Tdepth >= 1
    rng = np.random.default_rng(5)
    keys = np.unique(np.concatenate([rng.integers(0, 4 ** Tdepth, 300), [0, 4 ** Tdepth - 1]]))
    for which in (0, 1):
        bw = oix.bwt(which)
        hdr = np.fromfile(T.TOY + (".rbwt" if which else ".bwt"), np.uint32, 5)   # primary, C(C), C(G), C(T), seq_len (bwtio.c)
        L2 = [0, int(hdr[1]), int(hdr[2]), int(hdr[3])]
        for key in keys:
            k, l = 0, n
            for t in range(Tdepth):
                c = int(key) >> (2 * (Tdepth - 1 - t)) & 3
                ok, ol = lib.orc_occ(bw, (k - 1) & 0xffffffff, c), lib.orc_occ(bw, l, c)
                k, l = L2[c] + ok + 1, L2[c] + ol
                if k > l:
                    break
            got = gix.export(which, 3, int(key), 1)[0]
            if k > l:
                assert got[0] > got[1], (which, key)
            else:
                assert (int(got[0]), int(got[1])) == (k, l), (which, key)


@pytest.mark.parametrize("block", ["e15", "e40", "n16", "o20"])
def test_option_blocks_beyond_the_first_pass_go_to_kernel_d(gix, olib, oix, block):
    """option blocks the first-pass kernel's compact entries cannot hold -- more than 64 score levels (`aln -e 15`), max_gape
    > 31, max_diff > 14, max_gapo > 15 -- are not refused: the whole batch goes to kernel D, and the answers are the oracle's"""
    rng = np.random.default_rng(99)
    reads = random_reads(rng, 300 if block == "e15" else 60, toy_genome(), lens=(50, 76, 100), err=0.02)
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = T.default_opt()
    if block in ("e15", "e40"):
        opt.max_gape = 15 if block == "e15" else 40
        opt.mode &= ~1                              # BWA_MODE_GAPE off, as `aln -e` does
    elif block == "n16":
        opt.fnr, opt.max_diff, opt.max_entries = 0.0, 16, 3000      # (the cut-off keeps the oracle's run short)
    else:
        opt.fnr, opt.max_diff, opt.max_gapo, opt.max_entries = 0.0, 20, 20, 3000
    want, wmax = T.oracle_cal_sa_reg_gap(olib, oix.h, opt, seq, rseq, off, per_read=0, n_threads=8)
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    b.run()
    n2 = b.sync()
    got, gmax = b.fetch()
    b.close()
    assert n2 == len(reads)                         # every read went through the wide instantiation
    bad = [i for i in range(len(reads)) if got[i].tobytes() != want[i].tobytes()]
    assert not bad, "GPU differs from the oracle for %d reads" % len(bad)
    assert np.array_equal(gmax, wmax)



@pytest.mark.parametrize("per_read", [False, True])
def test_one_shot_entry_flat(gix, olib, oix, per_read):
    """nabwa_cal_sa_reg_gap on host buffers, as a C caller sees it: counts, rows in read order, max_entries; a row buffer
    that is too small is answered with NABWA_ECAP and the size needed"""
    rng = np.random.default_rng(21)
    reads = random_reads(rng, 1000, toy_genome(), lens=(36, 50, 76, 100, 100, 150), err=0.02)
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = T.default_opt()
    want, wmax = T.oracle_cal_sa_reg_gap(olib, oix.h, opt, seq, rseq, off, per_read=int(per_read), n_threads=8)
    n_aln, rows, maxe = gix.cal_sa_reg_gap_flat(to_gap_opt(opt), seq, rseq, off, per_read=per_read, cap_rows=10)   # too small first: resized by the answer
    assert (n_aln == np.array([len(w) for w in want])).all()
    assert rows.tobytes() == np.concatenate([np.asarray(w, nabwa.ALN_DT) for w in want]).tobytes()
    assert np.array_equal(maxe, wmax)


def test_staged_upload_and_pooled_buffers_on_a_large_batch(gix, olib, oix, monkeypatch):
    """Host buffers of >= 256 MiB go up through four threads and pinned slots (nabwa_api.hip: staged_upload), and the
    working buffers of a call are handed to the next one (pool).  2.2 M reads x 100 bases x {seq, rseq} = 440 MB: every
    thread re-uses its slots.  Same answer as the plain hipMemcpy path on fresh buffers, and as the oracle on a sample."""
    rng = np.random.default_rng(99)
    code = np.full(256, 4, np.uint8)
    for i, ch in enumerate("ACGT"):
        code[ord(ch)] = i
    g = code[np.frombuffer(toy_genome().upper().encode(), np.uint8)]
    n, L = 2_200_000, 100
    pos = rng.integers(0, len(g) - L, n)
    reads = g[pos[:, None] + np.arange(L)[None, :]]
    mut = rng.random(reads.shape) < 0.004
    reads[mut] = rng.integers(0, 4, int(mut.sum())).astype(np.uint8)
    flip = rng.random(n) < 0.5                                   # half of them from the other strand
    reads[flip] = np.where(reads[flip] < 4, 3 - reads[flip], 4)[:, ::-1]
    seq = np.ascontiguousarray(reads[:, ::-1]).reshape(-1)
    rseq = np.where(seq < 4, 3 - seq, seq).astype(np.uint8)
    off = (np.arange(n + 1, dtype=np.int64) * L)
    opt = to_gap_opt(T.default_opt())
    monkeypatch.setenv("NABWA_STAGED_MIN_MB", "1000000")         # plain hipMemcpy
    a = gix.cal_sa_reg_gap_flat(opt, seq, rseq, off, per_read=True)
    monkeypatch.setenv("NABWA_STAGED_MIN_MB", "0")               # staged, into the buffers the first call released
    b = gix.cal_sa_reg_gap_flat(opt, seq, rseq, off, per_read=True)
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    m = 3000
    want, wmax = T.oracle_cal_sa_reg_gap(olib, oix.h, T.default_opt(), seq[:m * L], rseq[:m * L], off[:m + 1], per_read=1, n_threads=8)
    n_aln, rows, maxe = b
    assert (n_aln[:m] == np.array([len(w) for w in want])).all()
    assert rows[:int(n_aln[:m].sum())].tobytes() == np.concatenate([np.asarray(w, nabwa.ALN_DT) for w in want]).tobytes()
    assert np.array_equal(maxe[:m], wmax)


@pytest.mark.parametrize("env", [
    {"NABWA_TEXT_MODE": "0"},                                   # no full SA / inverse / text: every interval stays in row form
    {"NABWA_KMER_T": "0"},                                      # no interval table: every tail is walked
    {"NABWA_KMER_T": "0", "NABWA_TEXT_MODE": "0"},              # the plain 2 GB index
    {"NABWA_KMER_T": "5"},                                      # a table shallower than the width passes would like
    {"NABWA_CLASS_SORT": "0", "NABWA_W_SYNC": "0"},             # batch order, no lockstep waves
    {"NABWA_TEXT_KERNELS": "1"}, {"NABWA_TEXT_KERNELS": "2"},   # text mode in one kernel only
], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_sai_parity_in_every_index_configuration(monkeypatch, env):
    """the accelerating structures are optional: each way of switching them off gives the reference's .sai (default and
    gap-heavy option sets) through the paths that remain"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ix = nabwa.Index.load(T.TOY, 0, True)
    try:
        for name, n_reads in (("default", 606), ("e3", 606), ("loggap", 150)):      # (-L -o 2 -e 8 is slow: a sample)
            opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_%s.sai" % name))
            reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:n_reads]
            seq, rseq, off, _ = T.encode_reads(reads, opt.trim_qual)
            # per_read=False sizes the options by the longest read of the batch: keep the full batch's longest read in
            per_read = False
            if n_reads < 606:
                per_read = None
            if per_read is None:
                full = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
                longest = max(range(len(full)), key=lambda i: len(full[i][1]))
                reads = reads + [full[longest]]
                gold = list(gold[:n_reads]) + [gold[longest]]
                seq, rseq, off, _ = T.encode_reads(reads, opt.trim_qual)
            got, _ = ix.cal_sa_reg_gap(to_gap_opt(opt), seq, rseq, off, per_read=False)
            bad = [reads[i][0] for i in range(len(reads)) if got[i].tobytes() != gold[i].tobytes()]
            assert not bad, "%s: GPU differs from the reference .sai for %d reads, e.g. %s" % (name, len(bad), bad[:5])
    finally:
        ix.close()


def test_width_kernel_output_matches_bwt_cal_width(gix, olib, oix):
    """kernel W directly (not through the search's rows): interval widths, lower bounds and the "same width as before" flag of
    the full passes, lower bounds of the seed passes, for both strands of reads of several lengths incl. N and reads shorter
    than the seed -- against the oracle's bwt_cal_width (bwtaln.c:52-76)"""
    rng = np.random.default_rng(31)
    reads = random_reads(rng, 400, toy_genome(), lens=(20, 32, 33, 50, 76, 100), err=0.03)
    seq, rseq, off, _ = T.encode_reads(reads)
    opt = T.default_opt()
    b = nabwa.Batch(gix, to_gap_opt(opt), seq, rseq, off, False)
    n, W, SW = len(reads), 101, opt.seed_len + 1
    w = np.zeros((n, 2, W), np.uint32); bid = np.zeros((n, 2, W), np.uint8); sbid = np.zeros((n, 2, SW), np.uint8)
    L = nabwa.lib()
    L.nabwa_batch_width_records.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.nabwa_batch_width_records(b._h, 0, n, T.ptr(w), T.ptr(bid), T.ptr(sbid)) == 0, L.nabwa_last_error()
    olib.orc_cal_width.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    ow = np.zeros(W, np.uint32); ob = np.zeros(W, np.int32)
    for i in range(n):
        ln = int(off[i + 1] - off[i])
        for x, arr in ((0, seq), (1, rseq)):
            s = np.ascontiguousarray(arr[off[i]:off[i + 1]])
            olib.orc_cal_width(oix.bwt(x), ln, T.ptr(s), T.ptr(ow), T.ptr(ob))
            assert np.array_equal(w[i, x, :ln + 1], ow[:ln + 1]), (i, x)
            assert np.array_equal(bid[i, x, :ln + 1] & 127, np.minimum(ob[:ln + 1], 127)), (i, x)
            same = np.concatenate([[False], ow[1:ln + 1] == ow[:ln]])
            assert np.array_equal(bid[i, x, :ln + 1] >> 7 != 0, same), (i, x)
            if ln > opt.seed_len:
                t = np.ascontiguousarray(s[ln - opt.seed_len:])
                olib.orc_cal_width(oix.bwt(x), opt.seed_len, T.ptr(t), T.ptr(ow), T.ptr(ob))
                assert np.array_equal(sbid[i, x, :SW] & 127, np.minimum(ob[:SW], 127)), (i, x)
    b.close()
