#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs oracle/_ref built by `make -C oracle ref`,
which compiles the reference's own sources in place from /root/reference).
Everything written here is *data*: a seeded synthetic genome, its index files as
written by the reference's `index` command, seeded synthetic reads, and the
reference's outputs for them (.sai from `aln`, SAM from `samse`, and function-level
vectors for the rank / SA / DP primitives).  No reference source text is stored.

Usage:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "bwa_ref")
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libbwaref.so")
PREFIX = os.path.join(HERE, "toy")

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def make_genome(rng):
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    c1 = list(rnd(60000))
    # exact 5 kb duplicate inside chr1, and a diverged (1 %) copy on chr2
    dup = c1[5000:10000]
    c1[30000:35000] = dup
    c2 = list(rnd(40000))
    div = list(dup[:2000])
    for p in rng.choice(2000, 20, replace=False):
        div[p] = "ACGT"[("ACGT".index(div[p]) + 1 + rng.integers(0, 3)) % 4]
    c2[8000:10000] = div
    # N run, short N, tandem repeats, homopolymer
    c1[45000:45200] = "N" * 200
    c1[52000:52003] = "NNN"
    c1[20000:20300] = list("ACG" * 100)
    c2[20000:20080] = "A" * 80
    c2[25000:25400] = list("ACGTTGCA" * 50)
    c2[30000:30001] = "N"
    # a short third contig so that contig-bridging alignments exist
    c3 = list(rnd(3000))
    return [("chr1", "".join(c1)), ("chr2", "".join(c2)), ("chr3", "".join(c3))]


def mutate(rng, s, n_sub=0, ins=0, dele=0, n_n=0, margin=8):
    s = list(s)
    L = len(s)
    pos = rng.choice(np.arange(margin, L - margin), size=n_sub + n_n + (1 if ins else 0) + (1 if dele else 0),
                     replace=False)
    pos = list(pos)
    for _ in range(n_sub):
        p = pos.pop()
        if s[p] in "ACGT":
            s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + rng.integers(0, 3)) % 4]
    for _ in range(n_n):
        s[pos.pop()] = "N"
    if ins:
        p = pos.pop()
        s[p:p] = list("".join("ACGT"[i] for i in rng.integers(0, 4, ins)))
    if dele:
        p = pos.pop()
        del s[p:p + dele]
    return "".join(s)


def make_reads(rng, contigs):
    g = {n: s for n, s in contigs}
    cat = "".join(s for _, s in contigs)
    reads = []

    def sample(L, lo=0, hi=None, contig=None):
        src = g[contig] if contig else cat
        hi = (len(src) - L) if hi is None else hi
        while True:
            p = int(rng.integers(lo, hi))
            w = src[p:p + L + 8]
            if "N" not in w:
                return w

    def add(tag, s, qual=None):
        if rng.integers(0, 2):
            s = revcomp(s)
            if qual:
                qual = qual[::-1]
        reads.append(("r%04d_%s" % (len(reads), tag), s, qual or "I" * len(s)))

    for L, n in ((100, 60), (150, 12), (76, 12), (50, 12), (36, 8), (32, 6), (20, 4), (250, 4)):
        for _ in range(n):
            add("exact%d" % L, sample(L)[:L])
    for k in (1, 2, 3, 4, 5, 6):
        for _ in range(20):
            add("sub%d" % k, mutate(rng, sample(100)[:100], n_sub=k))
    for k in (1, 2, 3):
        for _ in range(10):
            add("sub%d_76" % k, mutate(rng, sample(76)[:76], n_sub=k))
            add("sub%d_50" % k, mutate(rng, sample(50)[:50], n_sub=k))
    for ins in (1, 2, 3, 5):
        for _ in range(12):
            add("ins%d" % ins, mutate(rng, sample(100)[:100], ins=ins, margin=12)[:100])
    for d in (1, 2, 3, 5, 7):
        for _ in range(12):
            add("del%d" % d, mutate(rng, sample(100)[:100 + d], dele=d, margin=12))
    for _ in range(20):
        add("indelsub", mutate(rng, sample(100)[:101], n_sub=int(rng.integers(1, 3)), dele=1, margin=12))
        add("inssub", mutate(rng, sample(100)[:100], n_sub=int(rng.integers(1, 3)), ins=1, margin=12)[:100])
    for k in (1, 2, 3, 6):
        for _ in range(6):
            add("N%d" % k, mutate(rng, sample(100)[:100], n_n=k))
    add("allN", "N" * 60)
    add("halfN", "N" * 30 + sample(40)[:40])
    # repeats: the exact duplicate, the diverged copy, tandem repeats, homopolymer
    for _ in range(20):
        p = int(rng.integers(5000, 9900))
        add("dup", g["chr1"][p:p + 100])
        add("dupsub", mutate(rng, g["chr1"][p:p + 100], n_sub=1))
    for _ in range(10):
        p = int(rng.integers(8000, 9900))
        add("div", g["chr2"][p:p + 100])
    for _ in range(6):
        p = int(rng.integers(19950, 20250))
        add("tandem3", g["chr1"][p:p + 100])
        p = int(rng.integers(24950, 25350))
        add("tandem8", g["chr2"][p:p + 100])
        add("tandem8gap", mutate(rng, g["chr2"][25000 + int(rng.integers(0, 200)):][:101], dele=1, margin=20))
    for _ in range(4):
        p = int(rng.integers(19960, 20050))
        add("polyA", g["chr2"][p:p + 60])
    add("polyA70", "A" * 70)
    # contig bridging, ends of the concatenated text, N-run overlap
    l1 = len(g["chr1"])
    for off in (50, 30, 70, 99, 1):
        add("bridge", cat[l1 - off:l1 - off + 100])
    add("start", cat[:100])
    add("end", cat[-100:])
    add("endsub", mutate(rng, cat[-100:], n_sub=2))
    for off in (40, 80, 95):
        add("nrun", g["chr1"][45000 - off:45000 - off + 100].replace("N", "A"))
    # unmappable
    for L in (100, 50, 36):
        for _ in range(10):
            add("random%d" % L, "".join("ACGT"[i] for i in rng.integers(0, 4, L)))
    # low-quality tails for the trimming option set (-q 20)
    for _ in range(20):
        s = mutate(rng, sample(100)[:100], n_sub=int(rng.integers(0, 3)))
        t = int(rng.integers(5, 50))
        s = s[:100 - t] + "".join("ACGT"[i] for i in rng.integers(0, 4, t))
        q = "I" * (100 - t) + "".join(chr(33 + int(x)) for x in rng.integers(2, 15, t))
        reads.append(("r%04d_lowq%d" % (len(reads), t), s, q))
    return reads


OPTION_SETS = {
    # name: (aln args, run samse?)
    "default": ([], True),
    "adna": (["-n", "0.01", "-o", "2", "-l", "16500"], True),
    "n3": (["-n", "3"], False),
    "e3": (["-e", "3", "-o", "2"], False),
    "loggap": (["-L", "-o", "2", "-e", "8", "-d", "3"], False),
    "k1R5": (["-k", "1", "-R", "5", "-l", "25"], False),
    "i2": (["-i", "2", "-M", "2", "-O", "7", "-E", "3"], False),
    "q20": (["-q", "20"], True),
    "m64": (["-m", "64"], False),
}
NONSTOP_SUBSET = 120  # -N is slow; run it on the first reads only


def run(cmd, stdout):
    with open(stdout, "wb") as fo:
        subprocess.run(cmd, check=True, stdout=fo, stderr=subprocess.DEVNULL)


def main():
    if not (os.path.exists(REFBIN) and os.path.exists(REFLIB)):
        sys.exit("build the reference oracle first: make -C oracle ref")
    rng = np.random.default_rng(20261004)
    contigs = make_genome(rng)
    with open(PREFIX + ".fa", "w") as f:
        for n, s in contigs:
            f.write(">%s synthetic\n" % n)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    subprocess.run([REFBIN, "index", "-a", "is", "-p", PREFIX, PREFIX + ".fa"], check=True,
                   stderr=subprocess.DEVNULL)
    for ext in (".rpac",):
        if os.path.exists(PREFIX + ext):
            os.remove(PREFIX + ext)
    reads = make_reads(rng, contigs)
    fq = os.path.join(HERE, "reads_se.fq")
    with open(fq, "w") as f:
        for n, s, q in reads:
            f.write("@%s\n%s\n+\n%s\n" % (n, s, q))
    fqn = os.path.join(HERE, "reads_se_head.fq")
    with open(fqn, "w") as f:
        for n, s, q in reads[:NONSTOP_SUBSET]:
            f.write("@%s\n%s\n+\n%s\n" % (n, s, q))
    for name, (args, sam) in OPTION_SETS.items():
        sai = os.path.join(HERE, "se_%s.sai" % name)
        run([REFBIN, "aln"] + args + [PREFIX, fq], sai)
        if sam:
            run([REFBIN, "samse", PREFIX, sai, fq], os.path.join(HERE, "se_%s.sam" % name))
    run([REFBIN, "aln", "-N", PREFIX, fqn], os.path.join(HERE, "se_nonstop.sai"))

    # ---- function-level vectors through the ctypes harness -------------------------------
    lib = C.CDLL(REFLIB)
    lib.ref_index_load.restype = C.c_void_p
    lib.ref_index_load.argtypes = [C.c_char_p, C.c_int]
    ix = C.c_void_p(lib.ref_index_load(PREFIX.encode(), 1))
    lib.ref_seq_len.restype = C.c_uint32
    lib.ref_seq_len.argtypes = [C.c_void_p, C.c_int]
    lib.ref_primary.restype = C.c_uint32
    lib.ref_primary.argtypes = [C.c_void_p, C.c_int]
    lib.ref_occ.restype = C.c_uint32
    lib.ref_occ.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int]
    lib.ref_occ4.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    lib.ref_2occ4.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.ref_sa.restype = C.c_uint32
    lib.ref_sa.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    out = {}
    for which in (0, 1):
        n = lib.ref_seq_len(ix, which)
        prim = lib.ref_primary(ix, which)
        ks = np.concatenate([
            np.array([0, 1, 2, 126, 127, 128, 129, 255, 256, prim - 1, prim, prim + 1, n - 2, n - 1, n,
                      0xFFFFFFFF], dtype=np.uint64),
            rng.integers(0, n + 1, 400).astype(np.uint64)]).astype(np.uint32)
        occ = np.zeros((len(ks), 4), np.uint32)
        occ4 = np.zeros((len(ks), 4), np.uint32)
        buf = (C.c_uint32 * 4)()
        for i, k in enumerate(ks):
            for c in range(4):
                occ[i, c] = lib.ref_occ(ix, which, int(k), c)
            if k != n or True:
                lib.ref_occ4(ix, which, int(k), buf)
                occ4[i] = list(buf)
        # pairs (k-1, l) as the search issues them, incl. k-1 == -1 and same-bucket cases
        kk = rng.integers(0, n, 300).astype(np.uint64)
        ll = np.minimum(kk + rng.integers(0, 300, 300).astype(np.uint64), n).astype(np.uint32)
        km1 = (kk - 1).astype(np.uint32)  # wraps 0 -> 0xffffffff
        ck = np.zeros((300, 4), np.uint32)
        cl = np.zeros((300, 4), np.uint32)
        b2 = (C.c_uint32 * 4)()
        for i in range(300):
            lib.ref_2occ4(ix, which, int(km1[i]), int(ll[i]), buf, b2)
            ck[i] = list(buf)
            cl[i] = list(b2)
        sak = np.concatenate([np.array([1, 2, 31, 32, 33, prim, n], np.uint64),
                              rng.integers(1, n + 1, 300).astype(np.uint64)]).astype(np.uint32)
        sav = np.array([lib.ref_sa(ix, which, int(k)) for k in sak], np.uint32)
        out.update({"occ_k%d" % which: ks, "occ_v%d" % which: occ, "occ4_v%d" % which: occ4,
                    "p_k%d" % which: km1, "p_l%d" % which: ll, "p_ck%d" % which: ck, "p_cl%d" % which: cl,
                    "sa_k%d" % which: sak, "sa_v%d" % which: sav,
                    "seq_len%d" % which: np.uint32(n), "primary%d" % which: np.uint32(prim)})
    lib.ref_maxdiff.restype = C.c_int
    lib.ref_maxdiff.argtypes = [C.c_int, C.c_double, C.c_double]
    Ls = np.arange(1, 400)
    out["maxdiff_004"] = np.array([lib.ref_maxdiff(int(l), 0.02, 0.04) for l in Ls], np.int32)
    out["maxdiff_001"] = np.array([lib.ref_maxdiff(int(l), 0.02, 0.01) for l in Ls], np.int32)

    # ---- DP vectors: aln_global_core under aln_param_bwa-like and blast-like blocks ------
    lib.ref_global.restype = C.c_int
    sm_maq = np.array([11, -19, -19, -19, -13, -19, 11, -19, -19, -13, -19, -19, 11, -19, -13,
                       -19, -19, -19, 11, -13, -13, -13, -13, -13, -13], np.int32)
    sm_blast = np.array([1, -3, -3, -3, -2, -3, 1, -3, -3, -2, -3, -3, 1, -3, -2,
                         -3, -3, -3, 1, -2, -2, -2, -2, -2, -2], np.int32)
    params = [(26, 9, 5, sm_maq, 50), (26, 9, -1, sm_maq, 50), (5, 2, 2, sm_blast, 50),
              (26, 9, 5, sm_maq, 6), (8, 2, 2, sm_blast, 3)]
    dp_rows = []
    cases = []
    for t in range(260):
        l2 = int(rng.integers(1, 140))
        q = rng.integers(0, 4, l2).astype(np.uint8)
        r = list(q)
        # derive the reference window from the query with a few edits
        for _ in range(int(rng.integers(0, 4))):
            if len(r) > 2:
                r[int(rng.integers(0, len(r)))] = int(rng.integers(0, 4))
        kind = int(rng.integers(0, 6))
        if kind == 1 and len(r) > 12:
            p = int(rng.integers(4, len(r) - 4)); d = int(rng.integers(1, 8)); del r[p:p + d]
        elif kind == 2:
            p = int(rng.integers(0, len(r) + 1)); d = int(rng.integers(1, 8))
            r[p:p] = list(rng.integers(0, 4, d))
        elif kind == 3:
            r = list(rng.integers(0, 4, int(rng.integers(1, 150))))
        elif kind == 4:
            r = r + list(rng.integers(0, 4, int(rng.integers(1, 70))))
        if not r:
            r = [0]
        if rng.integers(0, 8) == 0:
            r[int(rng.integers(0, len(r)))] = 4
        if rng.integers(0, 8) == 0:
            q[int(rng.integers(0, l2))] = 4
        r = np.array(r, np.uint8)
        cases.append((r, q, params[t % len(params)], t % len(params)))
    cig = (C.c_uint32 * 1024)()
    ncig = C.c_int()
    plen = C.c_int()
    for r, q, (go, ge, gend, sm, band), pid in cases:
        sc = lib.ref_global(r.ctypes.data_as(C.c_void_p), len(r), q.ctypes.data_as(C.c_void_p), len(q),
                            go, ge, gend, sm.ctypes.data_as(C.c_void_p), 5, band, cig, C.byref(ncig), None,
                            C.byref(plen))
        dp_rows.append((r, q, pid, sc, np.array(cig[:ncig.value], np.uint32)))
    out["dp_n"] = np.int32(len(dp_rows))
    out["dp_ref"] = np.concatenate([x[0] for x in dp_rows])
    out["dp_ref_off"] = np.cumsum([0] + [len(x[0]) for x in dp_rows]).astype(np.int64)
    out["dp_qry"] = np.concatenate([x[1] for x in dp_rows])
    out["dp_qry_off"] = np.cumsum([0] + [len(x[1]) for x in dp_rows]).astype(np.int64)
    out["dp_pid"] = np.array([x[2] for x in dp_rows], np.int32)
    out["dp_score"] = np.array([x[3] for x in dp_rows], np.int32)
    out["dp_cig"] = np.concatenate([x[4] for x in dp_rows])
    out["dp_cig_off"] = np.cumsum([0] + [len(x[4]) for x in dp_rows]).astype(np.int64)
    out["dp_params"] = np.array([[p[0], p[1], p[2], p[4], 0 if p[3] is sm_maq else 1] for p in params], np.int32)
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **out)
    make_sw_vectors(lib, sm_maq, sm_blast)
    make_pe_vectors(lib)
    make_pe_chain(lib)
    print("golden fixtures written to", HERE, "reads:", len(reads))


def make_sw_vectors(lib, sm_maq, sm_blast):
    """aln_extend_core (stdaln.c:862) and aln_local_core (stdaln.c:529) known answers -> vectors_sw.npz"""
    rng = np.random.default_rng(424242)
    lib.ref_extend.restype = C.c_int
    lib.ref_local.restype = C.c_int
    params = [(26, 9, 5, sm_maq, 50, 0), (5, 2, 2, sm_blast, 50, 1), (26, 9, 5, sm_maq, 8, 0), (5, 2, 2, sm_blast, 4, 1)]
    refs, qrys, pids, g0s, e_score, e_cig, l_score, l_cig, l_subo = [], [], [], [], [], [], [], [], []
    cig = (C.c_uint32 * 4096)()
    ncig = C.c_int()
    plen = C.c_int()
    subo = C.c_int()
    for t in range(240):
        l2 = int(rng.integers(1, 150))
        q = rng.integers(0, 4, l2).astype(np.uint8)
        r = list(q)
        for _ in range(int(rng.integers(0, 6))):
            pos = int(rng.integers(0, len(r) + 1))
            k = int(rng.integers(0, 3))
            if k == 0 and len(r) > 1:
                del r[min(pos, len(r) - 1)]
            elif k == 1:
                r.insert(pos, int(rng.integers(0, 4)))
            elif len(r):
                r[min(pos, len(r) - 1)] = int(rng.integers(0, 4))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            r = list(rng.integers(0, 4, int(rng.integers(1, 200))))
        elif kind == 1:
            r = list(rng.integers(0, 4, int(rng.integers(0, 60)))) + r + list(rng.integers(0, 4, int(rng.integers(0, 60))))
        elif kind == 2:
            r = r + list(rng.integers(0, 4, int(rng.integers(1, 80))))
        if not r:
            r = [0]
        if rng.integers(0, 10) == 0:
            r[int(rng.integers(0, len(r)))] = 4
        r = np.array(r, np.uint8)
        pid = t % len(params)
        go, ge, gend, sm, band, _ = params[pid]
        g0 = int(rng.integers(1, 60))
        sc = lib.ref_extend(r.ctypes.data_as(C.c_void_p), len(r), q.ctypes.data_as(C.c_void_p), len(q), go, ge, gend,
                            sm.ctypes.data_as(C.c_void_p), 5, band, g0, cig, C.byref(ncig), None, C.byref(plen))
        refs.append(r); qrys.append(q); pids.append(pid); g0s.append(g0)
        e_score.append(sc); e_cig.append(np.array(cig[:ncig.value], np.uint32))
        sc = lib.ref_local(r.ctypes.data_as(C.c_void_p), len(r), q.ctypes.data_as(C.c_void_p), len(q), go, ge, gend,
                           sm.ctypes.data_as(C.c_void_p), 5, band, 1, cig, C.byref(ncig), None, C.byref(plen), C.byref(subo))
        l_score.append(sc); l_cig.append(np.array(cig[:ncig.value], np.uint32)); l_subo.append(subo.value)
    cat = lambda xs, dt: np.concatenate([np.asarray(x, dt) for x in xs]) if xs else np.zeros(0, dt)
    offs = lambda xs: np.cumsum([0] + [len(x) for x in xs]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "vectors_sw.npz"),
                        ref=cat(refs, np.uint8), ref_off=offs(refs), qry=cat(qrys, np.uint8), qry_off=offs(qrys),
                        pid=np.array(pids, np.int32), g0=np.array(g0s, np.int32),
                        params=np.array([[p[0], p[1], p[2], p[4], p[5]] for p in params], np.int32),
                        ext_score=np.array(e_score, np.int32), ext_cig=cat(e_cig, np.uint32), ext_cig_off=offs(e_cig),
                        loc_score=np.array(l_score, np.int32), loc_cig=cat(l_cig, np.uint32), loc_cig_off=offs(l_cig),
                        loc_subo=np.array(l_subo, np.int32))




def rescue_tasks(rng, n=400):
    """tasks shaped like mate rescue: a window of 300 - 620 bases, a read of 70 - 250 with substitutions, an indel now and then, some
    reads that are not in their window at all, N runs"""
    refs, qrys = [], []
    for t in range(n):
        lw, lr = int(rng.integers(300, 620)), int(rng.integers(70, 250))
        w = rng.integers(0, 4, lw).astype(np.uint8)
        if t % 9 == 0:
            r = rng.integers(0, 4, lr).astype(np.uint8)                    # not there
        else:
            p = int(rng.integers(0, lw - lr)) if lw > lr else 0
            r = w[p:p + lr].copy()
            sub = rng.random(len(r)) < 0.04
            r[sub] = rng.integers(0, 4, int(sub.sum()))
            if t % 4 == 0 and len(r) > 40:
                c = int(rng.integers(20, len(r) - 20))
                r = np.concatenate([r[:c], r[c + int(rng.integers(1, 4)):]]) if t % 8 == 0 else np.concatenate([r[:c], rng.integers(0, 4, int(rng.integers(1, 4))).astype(np.uint8), r[c:]])
        if t % 13 == 0:
            w[10:14] = 4; r[5:7] = 4
        refs.append(w); qrys.append(r.astype(np.uint8))
    return refs, qrys


def long_tasks(rng, n, anchored):
    """reads of 3000 - 3600 bases whose scores pass 32000: the reference's 16-bit rows drop by 16000 there (stdaln.c:252-253, :583-602,
    :919-932).  anchored: the read starts where the window starts (extension); else it lies inside a longer window (local)."""
    refs, qrys = [], []
    for t in range(n):
        lr = int(rng.integers(3000, 3600))
        q = rng.integers(0, 4, lr).astype(np.uint8)
        r = list(q)
        for _ in range(int(rng.integers(0, 12))):                            # a few edits; two tasks stay exact
            if t < 2:
                break
            pos = int(rng.integers(50, len(r) - 50))
            k = int(rng.integers(0, 3))
            if k == 0:
                del r[pos]
            elif k == 1:
                r.insert(pos, int(rng.integers(0, 4)))
            else:
                r[pos] = (r[pos] + 1) % 4
        if anchored:
            r = r + list(rng.integers(0, 4, int(rng.integers(0, 80))))
        else:
            r = list(rng.integers(0, 4, int(rng.integers(0, 150)))) + r + list(rng.integers(0, 4, int(rng.integers(0, 150))))
        refs.append(np.array(r, np.uint8)); qrys.append(q)
    return refs, qrys


def make_sw_rescue_vectors(lib=None):
    """aln_local_core / aln_extend_core known answers at the sizes the kernels are used at -> vectors_sw_rescue.npz:
    400 rescue-shaped local tasks under aln_param_bwa, 6 local and 6 extension tasks long enough for the 16-bit drop, 40 extension tasks
    of 100 - 400 bases (band 50 and 12).  For every local task: score, sub-optimal score, the path's first and last cell, the CIGAR."""
    if lib is None:
        lib = C.CDLL(REFLIB)
    lib.ref_extend.restype = C.c_int
    lib.ref_local.restype = C.c_int
    sm_maq = np.array([11, -19, -19, -19, -13, -19, 11, -19, -19, -13, -19, -19, 11, -19, -13, -19, -19, -19, 11, -13, -13, -13, -13, -13, -13], np.int32)
    rng = np.random.default_rng(77)
    cap = 1 << 15
    cig = (C.c_uint32 * cap)()
    path = np.zeros(3 * cap, np.int32)
    ncig, plen, subo = C.c_int(), C.c_int(), C.c_int()
    out = {}
    cat = lambda xs, dt: np.concatenate([np.asarray(x, dt) for x in xs]) if xs else np.zeros(0, dt)
    offs = lambda xs: np.cumsum([0] + [len(x) for x in xs]).astype(np.int64)

    def run_local(refs, qrys, tag):
        sc, su, co, cg = [], [], [], []
        for r, q in zip(refs, qrys):
            s_ = lib.ref_local(r.ctypes.data_as(C.c_void_p), len(r), q.ctypes.data_as(C.c_void_p), len(q), 26, 9, 5, sm_maq.ctypes.data_as(C.c_void_p), 5, 50, 1,
                               cig, C.byref(ncig), path.ctypes.data_as(C.c_void_p), C.byref(plen), C.byref(subo))
            sc.append(s_); su.append(subo.value); cg.append(np.array(cig[:ncig.value], np.uint32))
            n_p = plen.value
            # the path runs from the end cell back to the start cell (1-based i on the window, j on the read)
            co.append([path[3 * (n_p - 1)], path[3 * (n_p - 1) + 1], path[0], path[1]] if n_p > 0 else [0, 0, 0, 0])
        out[tag + "_ref"], out[tag + "_ref_off"], out[tag + "_qry"], out[tag + "_qry_off"] = cat(refs, np.uint8), offs(refs), cat(qrys, np.uint8), offs(qrys)
        out[tag + "_score"], out[tag + "_subo"], out[tag + "_coords"] = np.array(sc, np.int32), np.array(su, np.int32), np.array(co, np.int32)
        out[tag + "_cig"], out[tag + "_cig_off"] = cat(cg, np.uint32), offs(cg)

    def run_extend(refs, qrys, g0s, bands, tag):
        sc, cg = [], []
        for r, q, g0, bw in zip(refs, qrys, g0s, bands):
            s_ = lib.ref_extend(r.ctypes.data_as(C.c_void_p), len(r), q.ctypes.data_as(C.c_void_p), len(q), 26, 9, 5, sm_maq.ctypes.data_as(C.c_void_p), 5, int(bw), int(g0),
                                cig, C.byref(ncig), None, C.byref(plen))
            sc.append(s_); cg.append(np.array(cig[:ncig.value], np.uint32))
        out[tag + "_ref"], out[tag + "_ref_off"], out[tag + "_qry"], out[tag + "_qry_off"] = cat(refs, np.uint8), offs(refs), cat(qrys, np.uint8), offs(qrys)
        out[tag + "_g0"], out[tag + "_band"], out[tag + "_score"] = np.array(g0s, np.int32), np.array(bands, np.int32), np.array(sc, np.int32)
        out[tag + "_cig"], out[tag + "_cig_off"] = cat(cg, np.uint32), offs(cg)

    refs, qrys = rescue_tasks(rng)
    run_local(refs, qrys, "loc")
    refs, qrys = long_tasks(rng, 6, False)
    run_local(refs, qrys, "loclong")
    assert out["loclong_score"].max() > 32000
    refs, qrys = long_tasks(rng, 6, True)
    run_extend(refs, qrys, [int(rng.integers(1, 60)) for _ in refs], [50] * len(refs), "extlong")
    assert out["extlong_score"].max() > 32000
    refs, qrys = [], []
    for t in range(40):
        lr = int(rng.integers(100, 400))
        q = rng.integers(0, 4, lr).astype(np.uint8)
        r = list(q)
        for _ in range(int(rng.integers(0, 8))):
            pos = int(rng.integers(0, len(r)))
            k = int(rng.integers(0, 3))
            if k == 0 and len(r) > 1:
                del r[pos]
            elif k == 1:
                r.insert(pos, int(rng.integers(0, 4)))
            else:
                r[pos] = (r[pos] + 1) % 4
        if t % 5 == 0:                                                      # the read runs off into unrelated sequence half-way
            r[len(r) // 2:] = list(rng.integers(0, 4, len(r) - len(r) // 2))
        refs.append(np.array(r + list(rng.integers(0, 4, int(rng.integers(0, 60)))), np.uint8)); qrys.append(q)
    run_extend(refs, qrys, [int(rng.integers(1, 80)) for _ in refs], [50 if t % 2 else 12 for t in range(40)], "ext")
    np.savez_compressed(os.path.join(HERE, "vectors_sw_rescue.npz"), **out)
    print("vectors_sw_rescue.npz: %d local tasks (%d found), long local scores %s, long extension scores %s"
          % (len(out["loc_score"]), int((out["loc_score"] > 0).sum()), out["loclong_score"].tolist(), out["extlong_score"].tolist()))


def make_pe_vectors(lib):
    """insert-size inference (insert_size.c) and pairing (bwape.c:180-293) known answers -> vectors_pe.npz"""
    rng = np.random.default_rng(777)
    out = {}
    # ---- insert-size histograms: normal, skewed, bimodal, too few, saturated bin
    hists, res = [], []
    for t in range(24):
        h = np.zeros(100000, np.int64)
        kind = t % 6
        n = int(rng.integers(30, 60000)) if kind != 3 else int(rng.integers(0, 19))
        if kind in (0, 3, 5):
            x = rng.normal(rng.integers(150, 600), rng.integers(10, 80), n)
        elif kind == 1:
            x = rng.gamma(4.0, rng.integers(30, 90), n)
        elif kind == 2:
            x = np.concatenate([rng.normal(250, 20, n // 2), rng.normal(3000, 300, n - n // 2)])
        else:
            x = rng.normal(400, 40, n)
            x[: max(1, n // 50)] = rng.uniform(0, 99999, max(1, n // 50))
        x = np.clip(np.round(x), 0, 99999).astype(np.int64)
        np.add.at(h, x, 1)
        if kind == 5:
            h[int(np.argmax(h))] = 65535
        h = np.minimum(h, 65535).astype(np.uint16)
        o = np.zeros(6, np.float64)
        L = int(rng.choice([4641652, 3099734149, 103000]))
        ap = float(rng.choice([1e-5, 1e-3]))
        lib.ref_infer_isize(h.ctypes.data_as(C.c_void_p), C.c_double(ap), C.c_int64(L), o.ctypes.data_as(C.c_void_p))
        hists.append(np.concatenate([[L], np.nonzero(h)[0], [-1], h[np.nonzero(h)[0]]]).astype(np.int64))
        res.append(np.concatenate([[ap], o]))
    out["is_hist"] = np.concatenate(hists)
    out["is_hist_off"] = np.cumsum([0] + [len(x) for x in hists]).astype(np.int64)
    out["is_res"] = np.array(res, np.float64)
    # ---- pairing cases
    lib.ref_pairing.restype = C.c_int
    pin, pout, cnts, alns, aoff, hits, hoff, misc = [], [], [], [], [0], [], [0], []
    for t in range(400):
        na = [int(rng.integers(1, 5)), int(rng.integers(1, 5))]
        base = int(rng.integers(2000, 90000))
        rows = [[], []]
        hp, hr, he = [], [], []
        for e in range(2):
            for r in range(na[e]):
                a = int(rng.integers(0, 2)); nmm = int(rng.integers(0, 4)); go = int(rng.integers(0, 2)); ge = int(rng.integers(0, 3)) if go else 0
                score = 3 * nmm + 11 * go + 4 * ge
                k = int(rng.integers(1, 100000)); w = int(rng.choice([1, 1, 1, 2, 3]))
                rows[e].append([nmm | go << 8 | ge << 16 | a << 24, k, k + w - 1, score])
                for _ in range(w):
                    far = rng.random() < 0.3
                    pos = int(rng.integers(0, 100000)) if far else base + int(rng.integers(-700, 700))
                    hp.append(max(pos, 0)); hr.append(r); he.append(e)
        ii = np.array([0, 0, 1e-5, 0, 0, 0], np.float64)
        if t % 3:
            avg = float(rng.integers(200, 500)); sd = float(rng.integers(10, 60))
            ii = np.array([avg, sd, 1e-5, max(1, avg - 4 * sd), avg + 4 * sd, avg + float(rng.integers(3, 7)) * sd], np.float64)
        p_in = np.zeros(22, np.int64)
        for e in range(2):
            j = int(rng.integers(0, len([1 for x in he if x == e])))
            idx = [i for i, x in enumerate(he) if x == e][j]
            r = rows[e][hr[idx]]
            ln = int(rng.choice([100, 100, 76, 150]))
            p_in[11 * e: 11 * e + 11] = [hp[idx] if rng.random() < 0.8 else hp[idx] + 3, r[0] >> 24 & 1, int(rng.choice([0, 0, 10, 23, 25, 37])),
                                          int(rng.choice([0, 23, 37])), ln, ln, r[0] & 0xff, r[0] >> 8 & 0xff, r[0] >> 16 & 0xff, r[3], 1 | (64 if e == 0 else 128)]
        a0 = np.array(rows[0], np.uint32).reshape(-1); a1 = np.array(rows[1], np.uint32).reshape(-1)
        hp_ = np.array(hp, np.uint32); hr_ = np.array(hr, np.int32); he_ = np.array(he, np.int32)
        p_out = np.zeros(22, np.int64)
        nn = np.array(na, np.int32)
        max_isize = int(rng.choice([500, 1000]))
        cnt = lib.ref_pairing(nn.ctypes.data_as(C.c_void_p), a0.ctypes.data_as(C.c_void_p), a1.ctypes.data_as(C.c_void_p),
                              len(hp), hp_.ctypes.data_as(C.c_void_p), hr_.ctypes.data_as(C.c_void_p), he_.ctypes.data_as(C.c_void_p),
                              p_in.ctypes.data_as(C.c_void_p), max_isize, 1, 3, ii.ctypes.data_as(C.c_void_p),
                              p_out.ctypes.data_as(C.c_void_p))
        pin.append(p_in); pout.append(p_out); cnts.append(cnt)
        alns.append(np.concatenate([a0, a1])); aoff.append(aoff[-1] + len(a0) + len(a1))
        hits.append(np.stack([hp_.astype(np.int64), hr_, he_], 1).reshape(-1)); hoff.append(hoff[-1] + 3 * len(hp))
        misc.append(np.concatenate([[na[0], na[1], max_isize], ii]))
    out.update(pr_in=np.array(pin), pr_out=np.array(pout), pr_cnt=np.array(cnts, np.int32), pr_aln=np.concatenate(alns),
               pr_aln_off=np.array(aoff, np.int64), pr_hit=np.concatenate(hits), pr_hit_off=np.array(hoff, np.int64),
               pr_misc=np.array(misc, np.float64))
    np.savez_compressed(os.path.join(HERE, "vectors_pe.npz"), **out)


def read_genome():
    out, name, buf = [], None, []
    for line in open(PREFIX + ".fa"):
        if line.startswith(">"):
            if name:
                out.append((name, "".join(buf)))
            name, buf = line[1:].split()[0], []
        else:
            buf.append(line.strip())
    out.append((name, "".join(buf)))
    return out


def make_pe_reads(rng, contigs, L=100, mu=300, sd=25, heavy=False):
    """Read pairs of the toy genome: proper FR pairs (insert ~N(300,25)), pairs whose second end is too diverged for
    `aln` (singletons), far-apart and cross-contig pairs (discordant), ends inside the exact duplicate (pairing
    has to choose), and pairs whose second end carries the chr1 version of the diverged chr2 copy (so `aln` places it
    on chr1 and the mate rescue finds it next to read 1 on chr2).  L / mu / sd: read length and insert sizes; heavy: the
    error load of BASELINE config 3 (about 2 % substitutions on every read)."""
    g = dict(contigs)
    pairs = []

    def subs(s, n, lo=0, hi=None):
        s = list(s)
        hi = len(s) if hi is None else hi
        for p in rng.choice(np.arange(lo, hi), n, replace=False):
            s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + int(rng.integers(0, 3))) % 4] if s[p] in "ACGT" else "A"
        return "".join(s)

    def frag_pair(contig, start, isize, L1, L2):
        f = g[contig][start:start + isize]
        return f[:L1], revcomp(f[-L2:])

    def add(tag, r1, r2):
        if rng.integers(0, 2):                       # which end is read 1 is arbitrary
            r1, r2 = r2, r1
        pairs.append(("p%04d_%s" % (len(pairs), tag), r1, r2))

    def clean(contig, start, isize):
        return "N" not in g[contig][start:start + isize]

    n_kind = {"proper": 230, "sub": 40, "diverged": 24, "far": 24, "cross": 16, "dup": 30, "rescue": 30, "short": 16, "gap": 20, "junk": 6}
    for kind, n in n_kind.items():
        made = 0
        while made < n:
            contig = "chr1" if rng.random() < 0.6 else "chr2"
            isize = int(np.clip(rng.normal(mu, sd), 2 * L + 10, mu + 5 * sd))
            L1 = L2 = L
            if kind == "short":
                L1, L2 = int(rng.choice([76, 50, 100])), int(rng.choice([76, 50]))
            start = int(rng.integers(0, len(g[contig]) - isize))
            if kind == "dup":
                contig, start = "chr1", int(rng.choice([5000, 30000])) + int(rng.integers(-250, 4900))
            if kind == "rescue":
                contig, start = "chr2", int(rng.integers(7700, 7990))
                isize = int(rng.integers(mu + 30, mu + 120))
            if not clean(contig, start, isize):
                continue
            r1, r2 = frag_pair(contig, start, isize, L1, L2)
            if kind == "proper":
                hi = 6 if heavy else 3
                r1, r2 = subs(r1, int(rng.integers(0, hi))), subs(r2, int(rng.integers(0, hi)))
            elif kind == "sub":
                r1, r2 = subs(r1, int(rng.integers(2, 5)), 34), subs(r2, int(rng.integers(2, 5)), 34)
            elif kind == "diverged":                  # too many differences for aln: the end stays unmapped
                r2 = subs(r2, int(rng.integers(9, 14)))
            elif kind == "far":
                start2 = int(rng.integers(0, len(g[contig]) - L))
                if not clean(contig, start2, L) or abs(start2 - start) < 1500:
                    continue
                r2 = revcomp(g[contig][start2:start2 + L]) if rng.random() < 0.7 else g[contig][start2:start2 + L]
            elif kind == "cross":
                other = "chr2" if contig == "chr1" else ("chr3" if rng.random() < 0.3 else "chr1")
                start2 = int(rng.integers(0, len(g[other]) - L))
                if not clean(other, start2, L):
                    continue
                r2 = revcomp(g[other][start2:start2 + L])
            elif kind == "rescue":
                # second end: chr1 version (chr1[5000+x]) of the chr2 copy window, with differences outside the seed
                e2 = start + isize                    # fragment end on chr2, inside the copy 8000..10000
                lo = e2 - L
                if lo < 8000 or e2 > 10000:
                    continue
                w1 = g["chr1"][5000 + lo - 8000: 5000 + e2 - 8000]
                w2 = g["chr2"][lo:e2]
                ndiv = sum(a != b for a, b in zip(w1, w2))
                if ndiv < 1:
                    continue
                r2f = list(w1)
                cand = [i for i in range(0, L - 34) if w1[i] == w2[i]]        # read 2 is the reverse complement: its seed is the window's tail
                for p in rng.choice(cand, int(rng.integers(4, 6)), replace=False):
                    r2f[p] = "ACGT"[("ACGT".index(r2f[p]) + 1 + int(rng.integers(0, 3))) % 4]
                r2 = revcomp("".join(r2f))
            elif kind == "gap":
                p = int(rng.integers(36, 80)); d = int(rng.integers(1, 4))
                if rng.random() < 0.5:
                    r2 = r2[:p] + r2[p + d:] + "ACGTACGT"[:d]
                else:
                    r2 = (r2[:p] + "TGCA"[:d] + r2[p:])[:len(r2)]
                r1 = subs(r1, int(rng.integers(0, 2)), 34)
            elif kind == "junk":
                r1 = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
                r2 = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
            if kind in ("proper", "sub") and rng.random() < 0.04:
                r1 = r1[:50] + "N" + r1[51:]
            add(kind, r1, r2)
            made += 1
    order = rng.permutation(len(pairs))
    return [pairs[i] for i in order]


PE_F = ("type", "strand", "n_mm", "n_gapo", "n_gape", "score", "sa", "c1", "c2", "pos", "mapQ", "seQ", "extra_flag", "n_cigar",
        "nm", "n_multi", "len")


def make_pe_chain(lib=None, tag="", L=100, mu=300, sd=25, heavy=False, seed=4242):
    """The paired-end chain (bam2bam.c:683-811 / bwape.c:295-425,519-633 / bwase.c:356-423) on toy read pairs:
    reads_pe_[12].fq, pe_[12].sai, pe_default.sam (the reference's `sampe`), vectors_pe_chain.npz (state of every end
    after pass 1 and after pass 2 under three insert-size estimates: sampe's own, bam2bam's histogram one, none).
    tag "150": the same for 2 x 150 bp pairs with config 3's error load (files reads_pe150_[12].fq, pe150_[12].sai,
    pe150_default.sam, vectors_pe150_chain.npz)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nabwa_testlib as T
    if lib is None:
        lib = C.CDLL(REFLIB)
    rng = np.random.default_rng(seed)
    pairs = make_pe_reads(rng, read_genome(), L=L, mu=mu, sd=sd, heavy=heavy)
    fq = [os.path.join(HERE, "reads_pe%s_%d.fq" % (tag, e)) for e in (1, 2)]
    sai = [os.path.join(HERE, "pe%s_%d.sai" % (tag, e)) for e in (1, 2)]
    for e in range(2):
        with open(fq[e], "w") as f:
            for n, r1, r2 in pairs:
                s = (r1, r2)[e]
                f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
        run([REFBIN, "aln", PREFIX, fq[e]], sai[e])
    sam_path = os.path.join(HERE, "pe%s_default.sam" % tag)
    run([REFBIN, "sampe", PREFIX, sai[0], sai[1], fq[0], fq[1]], sam_path)

    P = C.c_void_p
    lib.ref_index_load.restype = P
    lib.ref_index_load.argtypes = [C.c_char_p, C.c_int]
    lib.ref_pe_new.restype = P
    lib.ref_pe_new.argtypes = [C.c_int]
    lib.ref_pe_set.argtypes = [P, C.c_int, C.c_int, C.c_int, P, P, C.c_int, P]
    lib.ref_pe_posn.argtypes = [P, P, P]
    lib.ref_pe_finish.argtypes = [P, P, P, P]
    lib.ref_pe_get.argtypes = [P, C.c_int, C.c_int, P, P, C.c_char_p, C.c_int, P]
    lib.ref_pe_isize_pairs.argtypes = [P, C.c_double, C.c_int64, P]
    lib.ref_pe_isize_pairs.restype = C.c_int
    lib.ref_pe_free.argtypes = [P]
    lib.ref_seed48.argtypes = [C.c_long]
    ix = lib.ref_index_load(PREFIX.encode(), 1)
    opt, alns = [], []
    enc = []
    for e in range(2):
        o, a = T.read_sai(sai[e])
        opt.append(o); alns.append(a)
        enc.append(T.encode_reads(T.read_fastq(fq[e])))
    n = len(pairs)
    l_pac = sum(len(s) for _, s in read_genome())

    def snapshot(b):
        f = np.zeros((2 * n, 17), np.int64); cg = np.zeros((2 * n, 64), np.uint16); mu = np.zeros((2 * n, 16 * 21), np.int64)
        md = []
        buf = C.create_string_buffer(1024)
        for i in range(n):
            for e in range(2):
                r = 2 * i + e
                lib.ref_pe_get(b, i, e, f[r].ctypes.data_as(P), cg[r].ctypes.data_as(P), buf, 1024, mu[r].ctypes.data_as(P))
                md.append(buf.value.decode())
        return f, cg, mu, md

    def chain(ii_mode):
        b = lib.ref_pe_new(n)
        for i in range(n):
            for e in range(2):
                seq, rseq, off, _ = enc[e]
                a = alns[e][i]
                s = np.ascontiguousarray(seq[off[i]:off[i + 1]]); rs = np.ascontiguousarray(rseq[off[i]:off[i + 1]])
                av = np.ascontiguousarray(a).view(np.uint32) if len(a) else np.zeros(4, np.uint32)
                lib.ref_pe_set(b, i, e, len(s), s.ctypes.data_as(P), rs.ctypes.data_as(P), len(a), av.ctypes.data_as(P))
        lib.ref_seed48(11)
        lib.ref_pe_posn(b, ix, C.byref(opt[1]))
        posn = snapshot(b)
        ii = np.zeros(6, np.float64)
        if ii_mode == "sampe":
            lib.ref_pe_isize_pairs(b, C.c_double(1e-5), C.c_int64(l_pac), ii.ctypes.data_as(P))
        elif ii_mode == "hist":
            h = np.zeros(100000, np.uint16)
            f = posn[0]
            for i in range(n):
                a, c = f[2 * i], f[2 * i + 1]
                if a[0] in (1, 2) and c[0] in (1, 2) and a[10] >= 20 and c[10] >= 20:      # insert_size.c:141-165
                    d = c[9] + c[16] - a[9] if a[9] < c[9] else a[9] + a[16] - c[9]
                    if 0 <= d < 100000:
                        h[d] += 1
            lib.ref_infer_isize(h.ctypes.data_as(P), C.c_double(1e-5), C.c_int64(l_pac), ii.ctypes.data_as(P))
        lib.ref_pe_finish(b, ix, C.byref(opt[1]), ii.ctypes.data_as(P))
        fin = snapshot(b)
        lib.ref_pe_free(b)
        return posn, ii, fin

    out = {}
    for mode in ("sampe", "hist", "null"):
        posn, ii, fin = chain(mode)
        if mode == "sampe":
            out["posn_f"] = posn[0]
        out["ii_" + mode] = ii
        out["f_" + mode], out["cig_" + mode], out["multi_" + mode] = fin[0], fin[1], fin[2]
        out["md_" + mode] = np.array(fin[3])
        print("pe chain [%s] ii=%s  types=%s  FPP=%d  MATESW=%d" % (mode, ii, np.bincount(fin[0][:, 0], minlength=4),
              int((fin[0][:, 12] & 2).astype(bool).sum()), int((fin[0][:, 0] == 3).sum())))
    check_against_sampe(T, out, read_genome(), pairs, sam_path)
    np.savez_compressed(os.path.join(HERE, "vectors_pe%s_chain.npz" % tag), **out)


def check_against_sampe(T, out, contigs, pairs, sam_path):
    """The harness chain under sampe's own insert-size estimate must reproduce what the reference's `sampe` printed."""
    sam = T.parse_sam(sam_path)
    f, cg, md = out["f_sampe"], out["cig_sampe"], out["md_sampe"]
    offs = np.cumsum([0] + [len(s) for _, s in contigs])
    assert len(sam) == len(f), (len(sam), len(f))
    bad = 0
    for r, rec in enumerate(sam):
        if f[r][0] == 0:
            continue
        pos = int(f[r][9])
        sid = int(np.searchsorted(offs, pos, side="right") - 1)
        cig = "".join("%d%s" % (c & 0x3fff, "MIDS"[c >> 14]) for c in cg[r][:f[r][13]]) or "%dM" % f[r][16]
        ok = (rec["rname"] == contigs[sid][0] and rec["pos"] == pos - offs[sid] + 1 and rec["cigar"] == cig
              and rec["tags"].get("MD") == md[r] and rec["tags"].get("NM") == f[r][14] and bool(rec["flag"] & 2) == bool(f[r][12] & 2)
              and rec["tags"].get("SM") == f[r][11])
        if not (rec["flag"] & 4):
            ok = ok and rec["mapq"] == f[r][10]
        if not ok:
            bad += 1
            print("MISMATCH", r, rec, f[r], cig, md[r])
    assert bad == 0, "%d records differ from the reference's sampe output" % bad
    print("harness chain == reference sampe output on", len(sam), "records")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pe_chain":
        make_pe_chain()
    elif len(sys.argv) > 1 and sys.argv[1] == "pe_chain150":
        make_pe_chain(tag="150", L=150, mu=400, sd=40, heavy=True, seed=150150)
    elif len(sys.argv) > 1 and sys.argv[1] == "sw_rescue":
        make_sw_rescue_vectors()
    else:
        main()
