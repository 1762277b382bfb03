"""Shared helpers for the tests: fixture parsing, read encoding, ctypes bindings of the
CPU oracle (oracle/liboracle.so) and -- when built -- the compiled reference
(oracle/_ref/libbwaref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
TOY = os.path.join(GOLDEN, "toy")
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libbwaref.so")

NT4 = np.full(256, 4, np.uint8)
for _i, _c in enumerate("ACGT"):
    NT4[ord(_c)] = _i
    NT4[ord(_c.lower())] = _i


class GapOpt(C.Structure):
    """gap_opt_t, reference bwtaln.h:143-153 (64 bytes)."""
    _fields_ = [("s_mm", C.c_int), ("s_gapo", C.c_int), ("s_gape", C.c_int), ("mode", C.c_int),
                ("indel_end_skip", C.c_int), ("max_del_occ", C.c_int), ("max_entries", C.c_int),
                ("fnr", C.c_float), ("max_diff", C.c_int), ("max_gapo", C.c_int), ("max_gape", C.c_int),
                ("max_seed_diff", C.c_int), ("seed_len", C.c_int), ("n_threads", C.c_int),
                ("max_top2", C.c_int), ("trim_qual", C.c_int)]


assert C.sizeof(GapOpt) == 64


class Counters(C.Structure):
    _fields_ = [("n_bucket", C.c_uint64), ("n_sa", C.c_uint64), ("n_pop", C.c_uint64), ("n_push", C.c_uint64)]


class SeMulti(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("gap", C.c_int), ("mm", C.c_int), ("strand", C.c_int),
                ("n_cigar", C.c_int), ("cigar", C.c_uint16 * 64)]


class SeRec(C.Structure):
    _fields_ = [("type", C.c_int), ("strand", C.c_int), ("n_mm", C.c_int), ("n_gapo", C.c_int),
                ("n_gape", C.c_int), ("score", C.c_int), ("sa", C.c_uint32), ("pos", C.c_uint32),
                ("c1", C.c_uint32), ("c2", C.c_uint32), ("mapQ", C.c_int), ("seQ", C.c_int),
                ("len", C.c_int), ("full_len", C.c_int), ("clip_len", C.c_int),
                ("n_cigar", C.c_int), ("cigar", C.c_uint16 * 64), ("nm", C.c_int), ("md", C.c_char * 512),
                ("n_multi", C.c_int), ("multi", SeMulti * 16),
                ("flag", C.c_int), ("seqid", C.c_int), ("nn", C.c_int), ("rpos", C.c_int64), ("xt", C.c_char)]


def default_opt():
    o = GapOpt()
    o.s_mm, o.s_gapo, o.s_gape = 3, 11, 4
    o.max_diff, o.max_gapo, o.max_gape = -1, 1, 6
    o.indel_end_skip, o.max_del_occ, o.max_entries = 5, 10, 2000000
    o.mode = 0x01 | 0x02
    o.seed_len, o.max_seed_diff = 32, 2
    o.fnr = 0.04
    o.n_threads, o.max_top2, o.trim_qual = 1, 30, 0
    return o


def read_fastq(path):
    out = []
    with open(path) as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().strip()
            f.readline()
            q = f.readline().strip()
            out.append((h[1:].split()[0], s, q))
    return out


def read_fasta(path):
    """[(name, sequence)] of a FASTA file"""
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                out.append([line[1:].split()[0], []])
            elif out:
                out[-1][1].append(line.strip())
    return [(n, "".join(p)) for n, p in out]


def trim_len(qual, trim_qual, min_len=35):
    """bwa_trim_read, reference bwaseqio.c:110-123."""
    L = len(qual)
    if trim_qual < 1:
        return L
    s = 0
    mx = 0
    max_l = L - 1
    for l in range(L - 1, min_len - 2, -1):
        s += trim_qual - (ord(qual[l]) - 33)
        if s < 0:
            break
        if s > mx:
            mx = s
            max_l = l
    return max_l + 1


def encode_reads(reads, trim_qual=0, comp=True):
    """(name, bases, qual) -> flat seq (reversed read) / rseq (reverse complement) code arrays
    plus offsets, the way the reference prepares bwa_seq_t (bwaseqio.c:225-232, :272-297)."""
    seqs, rseqs, off, full = [], [], [0], []
    for _, s, q in reads:
        codes = NT4[np.frombuffer(s.encode(), np.uint8)]
        L = trim_len(q, trim_qual) if trim_qual >= 1 else len(codes)
        c = codes[:L]
        seq = c[::-1].copy()
        rseq = np.where(seq < 4, 3 - seq, seq).astype(np.uint8) if comp else seq.copy()
        seqs.append(seq)
        rseqs.append(rseq)
        off.append(off[-1] + L)
        full.append(len(codes))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.uint8)
    return cat(seqs), cat(rseqs), np.array(off, np.int64), np.array(full, np.int32)


ALN_DT = np.dtype([("info", "<u4"), ("k", "<u4"), ("l", "<u4"), ("score", "<i4")])


def read_sai(path):
    """.sai = gap_opt_t header + per read {int32 n_aln, n_aln x bwt_aln1_t} (bwtaln.c:242-246,387)."""
    raw = open(path, "rb").read()
    opt = GapOpt.from_buffer_copy(raw[:64])
    p = 64
    recs = []
    while p < len(raw):
        n = int(np.frombuffer(raw, "<i4", 1, p)[0])
        p += 4
        recs.append(np.frombuffer(raw, ALN_DT, n, p).copy())
        p += 16 * n
    return opt, recs


def parse_sam(path):
    recs = []
    for line in open(path):
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        tags = {}
        for t in f[11:]:
            k, ty, v = t.split(":", 2)
            tags[k] = int(v) if ty == "i" else v
        recs.append(dict(name=f[0], flag=int(f[1]), rname=f[2], pos=int(f[3]), mapq=int(f[4]), cigar=f[5],
                         rnext=f[6], pnext=int(f[7]), tlen=int(f[8]), seq=f[9], qual=f[10], tags=tags))
    return recs


def cigar16_str(c):
    return "".join("%d%s" % (x & 0x3fff, "MIDS"[x >> 14]) for x in c)


def cigar32_str(c):
    return "".join("%d%s" % (x >> 4, "MIDS"[x & 0xf]) for x in c)


def ensure_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "nabwa_oracle.c")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True,
                       stdout=subprocess.DEVNULL)
    return ORACLE_SO


_P = C.c_void_p


def load_oracle():
    lib = C.CDLL(ensure_oracle())
    lib.orc_index_load.restype = _P
    lib.orc_index_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    lib.orc_index_wrap.restype = _P
    lib.orc_index_wrap.argtypes = [_P, C.c_uint64, _P, C.c_uint64]
    lib.orc_index_free.argtypes = [_P]
    lib.orc_occ.restype = C.c_uint32
    lib.orc_occ.argtypes = [_P, C.c_uint32, C.c_int]
    lib.orc_occ4.argtypes = [_P, C.c_uint32, _P]
    lib.orc_2occ4.argtypes = [_P, C.c_uint32, C.c_uint32, _P, _P]
    lib.orc_sa.restype = C.c_uint32
    lib.orc_sa.argtypes = [_P, C.c_uint32]
    lib.orc_maxdiff.restype = C.c_int
    lib.orc_maxdiff.argtypes = [C.c_int, C.c_double, C.c_double]
    lib.orc_cal_sa_reg_gap.restype = C.c_long
    lib.orc_cal_sa_reg_gap.argtypes = [_P, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_long, _P, C.c_int, _P]
    lib.orc_srand48.argtypes = [_P, C.c_long]
    lib.orc_drand48.restype = C.c_double
    lib.orc_drand48.argtypes = [_P]
    lib.orc_se_finish.argtypes = [_P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P]
    lib.orc_global.restype = C.c_int
    lib.orc_global.argtypes = [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P]
    return lib


class OracleIndex:
    """orc_index_t; bwt(which) gives the address of the embedded orc_bwt_t."""
    BWT_SIZE = 64  # sizeof(orc_bwt_t) on LP64: checked in tests

    def __init__(self, lib, prefix=TOY, with_sa=1, with_pac=1):
        self.lib = lib
        self.h = lib.orc_index_load(prefix.encode(), with_sa, with_pac)

    def bwt(self, which):
        return C.c_void_p(self.h + which * self.BWT_SIZE)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def oracle_cal_sa_reg_gap(lib, ixh, opt, seq, rseq, off, per_read=0, n_threads=1, counters=None):
    n = len(off) - 1
    n_aln = np.zeros(n, np.int32)
    maxe = np.zeros(n, np.int32)
    cap = max(64 * n, 4096)
    while True:
        rows = np.zeros(cap, ALN_DT)
        tot = lib.orc_cal_sa_reg_gap(ixh, C.byref(opt), n, ptr(off), ptr(seq), ptr(rseq), per_read,
                                     ptr(n_aln), ptr(rows), cap, ptr(maxe), n_threads,
                                     C.byref(counters) if counters is not None else None)
        if tot >= 0:
            break
        cap *= 4
    bounds = np.concatenate([[0], np.cumsum(n_aln)])
    return [rows[bounds[i]:bounds[i + 1]] for i in range(n)], maxe


def load_ref():
    """The compiled reference (only in the build container / when oracle/_ref travelled)."""
    if not os.path.exists(REF_SO):
        return None
    lib = C.CDLL(REF_SO)
    lib.ref_index_load.restype = _P
    lib.ref_index_load.argtypes = [C.c_char_p, C.c_int]
    lib.ref_cal_sa_reg_gap.restype = C.c_long
    lib.ref_cal_sa_reg_gap.argtypes = [_P, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_long, _P]
    lib.ref_global.restype = C.c_int
    lib.ref_sa.restype = C.c_uint32
    lib.ref_sa.argtypes = [_P, C.c_int, C.c_uint32]
    lib.ref_seed48.argtypes = [C.c_long]
    lib.ref_aln2pos_se.argtypes = [_P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P]
    return lib
