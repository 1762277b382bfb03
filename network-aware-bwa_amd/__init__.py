"""network-aware-bwa_amd -- MI355X-native per-read alignment hot path of mpieva/network-aware-bwa.

This package is only the thin Python face (ctypes) of ``libnabwa.so`` -- the C-ABI library declared
in ``include/nabwa.h`` whose HIP kernels do the work.  It mirrors the reference's own operator
interface for the path (``bwa_cal_sa_reg_gap``, ``bwt_sa``; reference bwtaln.h:187, bwt.h:100) in
flat-array form.  There is no CPU fallback: if the library or a GPU is missing, calls raise.

The directory name contains a hyphen; import it with
``importlib.import_module("network-aware-bwa_amd")``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NABWA_LIB", os.path.join(_HERE, "libnabwa.so"))   # NABWA_LIB: A/B builds

OK, ENODEV, EINVAL, EIO, ENOMEM, ECAP, EHITS = 0, -1, -2, -3, -4, -5, -6

ALN_DT = np.dtype([("info", "<u4"), ("k", "<u4"), ("l", "<u4"), ("score", "<i4")])  # bwt_aln1_t, bwtaln.h:41-45


class GapOpt(C.Structure):
    """gap_opt_t (reference bwtaln.h:143-153), 64 bytes."""
    _fields_ = [("s_mm", C.c_int), ("s_gapo", C.c_int), ("s_gape", C.c_int), ("mode", C.c_int),
                ("indel_end_skip", C.c_int), ("max_del_occ", C.c_int), ("max_entries", C.c_int),
                ("fnr", C.c_float), ("max_diff", C.c_int), ("max_gapo", C.c_int), ("max_gape", C.c_int),
                ("max_seed_diff", C.c_int), ("seed_len", C.c_int), ("n_threads", C.c_int),
                ("max_top2", C.c_int), ("trim_qual", C.c_int)]


MAX_CIGAR, MAX_MD, MAX_MULTI = 64, 512, 16


class SeMulti(C.Structure):
    """nabwa_multi_t"""
    _fields_ = [("pos", C.c_uint32), ("gap", C.c_int32), ("mm", C.c_int32), ("strand", C.c_int32),
                ("n_cigar", C.c_int32), ("cigar", C.c_uint16 * MAX_CIGAR)]


class SeRec(C.Structure):
    """nabwa_se_t: a finished single-end alignment"""
    _fields_ = [("type", C.c_int32), ("strand", C.c_int32), ("n_mm", C.c_int32), ("n_gapo", C.c_int32),
                ("n_gape", C.c_int32), ("score", C.c_int32), ("sa", C.c_uint32), ("pos", C.c_uint32),
                ("c1", C.c_uint32), ("c2", C.c_uint32), ("mapQ", C.c_int32), ("seQ", C.c_int32),
                ("len", C.c_int32), ("full_len", C.c_int32), ("clip_len", C.c_int32),
                ("n_cigar", C.c_int32), ("cigar", C.c_uint16 * MAX_CIGAR), ("nm", C.c_int32), ("md", C.c_char * MAX_MD),
                ("n_multi", C.c_int32), ("multi", SeMulti * MAX_MULTI),
                ("flag", C.c_int32), ("seqid", C.c_int32), ("nn", C.c_int32), ("rpos", C.c_int64), ("xt", C.c_char)]


class IsizeInfo(C.Structure):
    """nabwa_isize_t = isize_info_t without the histogram pointer (reference bwape.h:16-20)"""
    _fields_ = [("avg", C.c_double), ("std", C.c_double), ("ap_prior", C.c_double), ("low", C.c_uint32),
                ("high", C.c_uint32), ("high_bayesian", C.c_uint32)]


class PeEnd(C.Structure):
    """nabwa_pe_end_t: the bwa_seq_t fields pairing() touches"""
    _fields_ = [("pos", C.c_uint32), ("strand", C.c_int32), ("mapQ", C.c_int32), ("seQ", C.c_int32), ("len", C.c_int32),
                ("full_len", C.c_int32), ("n_mm", C.c_int32), ("n_gapo", C.c_int32), ("n_gape", C.c_int32),
                ("score", C.c_int32), ("extra_flag", C.c_int32)]


class PeOpt(C.Structure):
    """nabwa_pe_opt_t = pe_opt_t (reference bwtaln.h:158-164)"""
    _fields_ = [("max_isize", C.c_int32), ("force_isize", C.c_int32), ("max_occ", C.c_int32), ("max_occ_se", C.c_int32),
                ("n_multi", C.c_int32), ("N_multi", C.c_int32), ("type", C.c_int32), ("is_sw", C.c_int32),
                ("is_preload", C.c_int32), ("ap_prior", C.c_double)]


class PeRec(C.Structure):
    """nabwa_pe_t: one end of a finished pair"""
    _fields_ = [("se", SeRec), ("extra_flag", C.c_int32), ("m_seqid", C.c_int32), ("am", C.c_int32), ("mapQ_paired", C.c_int32),
                ("m_rpos", C.c_int64), ("isize", C.c_int64)]


def pe_opt_default():
    """bwa_init_pe_opt (reference bwape.c:27-41)"""
    po = PeOpt()
    lib().nabwa_pe_opt_default(C.byref(po))
    return po


def isize_infer(hist, ap_prior, L):
    """infer_isize_hist (reference insert_size.c:50-139) -> (rc, IsizeInfo)"""
    h = np.ascontiguousarray(hist, np.uint16)
    ii = IsizeInfo()
    rc = lib().nabwa_isize_infer(_ptr(h), float(ap_prior), int(L), C.byref(ii))
    return rc, ii


def isize_add_pairs(recs, n_pairs, hist):
    """improve_isize_est (reference insert_size.c:141-165) over a batch of positioned pairs; hist: 100000 uint16 bins"""
    assert hist.dtype == np.uint16 and hist.flags.c_contiguous
    lib().nabwa_isize_add_pairs.argtypes = [C.c_int, _P, _P]
    _chk(lib().nabwa_isize_add_pairs(int(n_pairs), C.cast(recs, _P), _ptr(hist)))


def pairing(ends, hits, rows0, rows1, max_isize, s_mm, ii):
    """pairing (reference bwape.c:180-293); ends: (PeEnd * 2), modified in place; returns cnt_chg"""
    hits = np.ascontiguousarray(hits, np.uint64)
    r0 = np.ascontiguousarray(rows0, ALN_DT)
    r1 = np.ascontiguousarray(rows1, ALN_DT)
    return lib().nabwa_pairing(ends, len(hits), _ptr(hits), _ptr(r0), _ptr(r1), int(max_isize), int(s_mm), C.byref(ii))


class BwaSeq(C.Structure):
    """bwa_seq_t (reference bwtaln.h:64-90), 200 bytes, as nabwa_bwa_seq_t declares it"""
    _fields_ = [("name", C.c_void_p), ("seq", C.c_void_p), ("rseq", C.c_void_p), ("qual", C.c_void_p),
                ("bits0", C.c_uint32), ("bits1", C.c_uint32), ("score", C.c_int32), ("clip_len", C.c_int32),
                ("n_aln", C.c_int32), ("pad0", C.c_int32), ("aln", C.c_void_p), ("n_multi", C.c_int32), ("pad1", C.c_int32),
                ("multi", C.c_void_p), ("sa", C.c_uint32), ("pos", C.c_uint32), ("c1c2seq", C.c_uint64),
                ("n_cigar", C.c_int32), ("pad2", C.c_int32), ("cigar", C.c_void_p), ("tid", C.c_int32), ("bc", C.c_char * 64),
                ("lenbits", C.c_uint32), ("md", C.c_void_p), ("max_entries", C.c_int32), ("pad3", C.c_int32)]


def encode_read(codes, qual=None, reverse=False, trim_qual=0, is_comp=True):
    """bam1_to_seq's encoding (reference bwaseqio.c:272-307): -> (seq, rseq) of the trimmed length"""
    codes = np.ascontiguousarray(codes, np.uint8)
    q = np.ascontiguousarray(qual, np.uint8) if qual is not None else None
    s = np.zeros(max(len(codes), 1), np.uint8)
    r = np.zeros(max(len(codes), 1), np.uint8)
    L = lib().nabwa_encode_read(len(codes), _ptr(codes), _ptr(q), int(reverse), int(trim_qual), int(is_comp), _ptr(s), _ptr(r))
    return s[:L], r[:L]


def srand48_state(seed):
    """state of the reference's process-global drand48 stream right after srand48(seed)"""
    return ((seed & 0xffffffff) << 16) | 0x330E


def global_align(ref, ref_off, qry, qry_off, gap_open, gap_ext, gap_end, matrix25, band, device=0, max_cigar=MAX_CIGAR):
    """aln_global_core + aln_path2cigar32 (reference stdaln.c:345-525, :1009-1039) for a batch of pairs"""
    n = len(ref_off) - 1
    ref = np.ascontiguousarray(ref, np.uint8)
    qry = np.ascontiguousarray(qry, np.uint8)
    ref_off = np.ascontiguousarray(ref_off, np.int64)
    qry_off = np.ascontiguousarray(qry_off, np.int64)
    mat = np.ascontiguousarray(matrix25, np.int32)
    score = np.zeros(max(n, 1), np.int32)
    ncig = np.zeros(max(n, 1), np.int32)
    cig = np.zeros((max(n, 1), max_cigar), np.uint32)
    _chk(lib().nabwa_global_align(device, n, _ptr(ref_off), _ptr(ref), _ptr(qry_off), _ptr(qry), gap_open, gap_ext,
                                  gap_end, _ptr(mat), band, _ptr(score), _ptr(ncig), _ptr(cig), max_cigar))
    return score[:n], [cig[i, :ncig[i]] for i in range(n)]


def local_align(ref, ref_off, qry, qry_off, gap_open, gap_ext, matrix25, band, thres=1, device=0, max_cigar=MAX_CIGAR):
    """aln_local_core (reference stdaln.c:529-761) for a batch of pairs -> (score, coords[n,4], subo, cigars)"""
    n = len(ref_off) - 1
    ref = np.ascontiguousarray(ref, np.uint8)
    qry = np.ascontiguousarray(qry, np.uint8)
    ref_off = np.ascontiguousarray(ref_off, np.int64)
    qry_off = np.ascontiguousarray(qry_off, np.int64)
    mat = np.ascontiguousarray(matrix25, np.int32)
    score = np.zeros(max(n, 1), np.int32)
    coords = np.zeros((max(n, 1), 4), np.int32)
    subo = np.zeros(max(n, 1), np.int32)
    ncig = np.zeros(max(n, 1), np.int32)
    cig = np.zeros((max(n, 1), max_cigar), np.uint32)
    _chk(lib().nabwa_local_align(device, n, _ptr(ref_off), _ptr(ref), _ptr(qry_off), _ptr(qry), gap_open, gap_ext, _ptr(mat),
                                 band, thres, _ptr(score), _ptr(coords), _ptr(subo), _ptr(ncig), _ptr(cig), max_cigar))
    return score[:n], coords[:n], subo[:n], [cig[i, :ncig[i]] for i in range(n)]


def extend_align(ref, ref_off, qry, qry_off, gap_open, gap_ext, matrix25, band, g0, device=0, max_cigar=MAX_CIGAR):
    """aln_extend_core (reference stdaln.c:862-1007) for a batch of pairs"""
    n = len(ref_off) - 1
    ref = np.ascontiguousarray(ref, np.uint8)
    qry = np.ascontiguousarray(qry, np.uint8)
    ref_off = np.ascontiguousarray(ref_off, np.int64)
    qry_off = np.ascontiguousarray(qry_off, np.int64)
    mat = np.ascontiguousarray(matrix25, np.int32)
    g0 = np.ascontiguousarray(g0, np.int32)
    score = np.zeros(max(n, 1), np.int32)
    ncig = np.zeros(max(n, 1), np.int32)
    cig = np.zeros((max(n, 1), max_cigar), np.uint32)
    _chk(lib().nabwa_extend_align(device, n, _ptr(ref_off), _ptr(ref), _ptr(qry_off), _ptr(qry), gap_open, gap_ext,
                                  _ptr(mat), band, _ptr(g0), _ptr(score), _ptr(ncig), _ptr(cig), max_cigar))
    return score[:n], [cig[i, :ncig[i]] for i in range(n)]


class NabwaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libnabwa error %d: %s" % (code, msg))
        self.code = code


def build(verbose=False):
    """Compile libnabwa.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "all"], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libnabwa.so failed:\n%s\n%s" % (r.stdout, r.stderr))


_lib = None
_P = C.c_void_p


def lib():
    """The loaded library.  Raises if it has not been built -- there is no other implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libnabwa.so is missing (%s): run __graft_entry__.build() / make -C %s"
                           % (LIB_PATH, os.path.join(_HERE, "csrc")))
    L = C.CDLL(LIB_PATH)
    L.nabwa_last_error.restype = C.c_char_p
    L.nabwa_device_count.restype = C.c_int
    L.nabwa_gap_init_opt.argtypes = [_P]
    L.nabwa_cal_maxdiff.restype = C.c_int
    L.nabwa_cal_maxdiff.argtypes = [C.c_int, C.c_double, C.c_double]
    L.nabwa_index_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, _P]
    L.nabwa_index_from_arrays.argtypes = [C.c_int, C.c_int, _P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint64, _P,
                                          C.c_uint64, _P]
    L.nabwa_index_destroy.argtypes = [_P]
    L.nabwa_index_destroy.restype = None
    L.nabwa_index_seq_len.restype = C.c_uint32
    L.nabwa_index_seq_len.argtypes = [_P, C.c_int]
    L.nabwa_index_device_bytes.restype = C.c_uint64
    L.nabwa_index_device_bytes.argtypes = [_P]
    L.nabwa_cal_sa_reg_gap.argtypes = [_P, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_int64, _P, _P]
    L.nabwa_batch_create.argtypes = [_P, _P, C.c_int, _P, _P, _P, C.c_int, _P]
    L.nabwa_batch_run.argtypes = [_P]
    L.nabwa_batch_sync.argtypes = [_P, _P]
    L.nabwa_batch_last_kernel_ms.restype = C.c_float
    L.nabwa_batch_last_kernel_ms.argtypes = [_P]
    L.nabwa_batch_last_width_ms.restype = C.c_float
    L.nabwa_batch_last_width_ms.argtypes = [_P]
    L.nabwa_batch_last_deep_ms.restype = C.c_float
    L.nabwa_batch_last_deep_ms.argtypes = [_P]
    L.nabwa_batch_fetch.argtypes = [_P, _P, _P, C.c_int64, _P, _P]
    L.nabwa_batch_checksum.argtypes = [_P, _P, _P]
    L.nabwa_batch_count_touches.argtypes = [_P, _P, _P]
    L.nabwa_batch_destroy.argtypes = [_P]
    L.nabwa_batch_destroy.restype = None
    L.nabwa_sa_lookup.argtypes = [_P, C.c_int, _P, _P, _P]
    L.nabwa_occ4.argtypes = [_P, C.c_int, C.c_int, _P, _P]
    L.nabwa_global_align.argtypes = [C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P,
                                     C.c_int]
    L.nabwa_extend_align.argtypes = [C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P, C.c_int]
    L.nabwa_local_align.argtypes = [C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P,
                                    C.c_int]
    L.nabwa_index_attach_reference.argtypes = [_P, C.c_char_p]
    L.nabwa_bwa_cal_sa_reg_gap.argtypes = [_P, C.c_int, _P, _P]
    L.nabwa_isize_infer.argtypes = [_P, C.c_double, C.c_int64, _P]
    L.nabwa_isize_bin.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int]
    L.nabwa_pairing.argtypes = [_P, C.c_int, _P, _P, _P, C.c_int, C.c_int, _P]
    L.nabwa_encode_read.restype = C.c_int
    L.nabwa_encode_read.argtypes = [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]
    L.nabwa_se_finish.argtypes = [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]
    L.nabwa_index_export.argtypes = [_P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _P]
    L.nabwa_pe_opt_default.argtypes = [_P]
    L.nabwa_pe_opt_default.restype = None
    L.nabwa_pe_posn.argtypes = [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P]
    L.nabwa_pe_finish.argtypes = [_P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]
    _lib = L
    return L


def _chk(rc):
    if rc != OK:
        raise NabwaError(rc, lib().nabwa_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(_P) if a is not None else None


def gap_init_opt():
    """gap_init_opt (reference bwtaln.c:19-35)."""
    o = GapOpt()
    lib().nabwa_gap_init_opt(C.byref(o))
    return o


def cal_maxdiff(length, err=0.02, thres=0.04):
    """bwa_cal_maxdiff (reference bwtaln.c:37-49)."""
    return lib().nabwa_cal_maxdiff(int(length), float(err), float(thres))


class Index:
    """Both FM-indexes (forward and reversed text) resident in HBM; replaces the globals set up by
    init_genome_index (reference bam2bam.c:844-858)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, prefix, device=0, with_sa=True, with_ref=False):
        h = _P()
        _chk(lib().nabwa_index_load(prefix.encode(), device, int(with_sa), int(with_ref), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, bwt0, bwt1, sa0=None, sa1=None, device=0, device_ptrs=False):
        """bwt0/bwt1: .bwt/.rbwt file content as uint32 words (numpy arrays, or (ptr, n_words) tuples of
        device memory when device_ptrs)."""
        h = _P()

        def pn(x):
            if x is None:
                return None, 0
            if isinstance(x, tuple):
                return C.c_void_p(x[0]), int(x[1])
            return _ptr(x), x.size
        p0, n0 = pn(bwt0)
        p1, n1 = pn(bwt1)
        s0, m0 = pn(sa0)
        s1, m1 = pn(sa1)
        _chk(lib().nabwa_index_from_arrays(device, int(device_ptrs), p0, n0, p1, n1, s0, m0, s1, m1, C.byref(h)))
        return cls(h)

    def bwa_cal_sa_reg_gap(self, seqs, n_seqs, opt):
        """drop-in for bwa_cal_sa_reg_gap on an array of the reference's bwa_seq_t records (BwaSeq * n)"""
        _chk(lib().nabwa_bwa_cal_sa_reg_gap(self._h, n_seqs, seqs, C.byref(opt)))

    def attach_reference(self, prefix):
        """.ann/.amb/.pac of the index (reference bns_restore + bwt_restore_pac)"""
        _chk(lib().nabwa_index_attach_reference(self._h, prefix.encode()))

    def set_reference(self, l_pac, seed, pac, n_contigs=16, name="synth"):
        """nabwa_index_set_reference: n_contigs equal contigs "synth1".. over the reference (a contig's length is an int32 in the
        reference's bntann1_t), no ambiguity holes, .pac bytes from memory.  Returns the contig (names, offsets, lengths)."""
        L = lib()
        L.nabwa_index_set_reference.argtypes = [_P, C.c_int64, C.c_uint32, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, _P]
        offs = np.array([l_pac * i // n_contigs for i in range(n_contigs)], np.int64)
        lens = np.diff(np.append(offs, l_pac)).astype(np.int32)
        nm = ["%s%d" % (name, i + 1) for i in range(n_contigs)]
        names = (C.c_char_p * n_contigs)(*[x.encode() for x in nm])
        pac = np.ascontiguousarray(pac, np.uint8)
        _chk(L.nabwa_index_set_reference(self._h, int(l_pac), int(seed), n_contigs, names, _ptr(offs), _ptr(lens), 0, None, None, None, _ptr(pac)))
        return nm, offs, lens

    def pe_posn_flat(self, opt, off, full_len, n_aln, rows, rng_state, out=None):
        """nabwa_pe_posn on flat arrays (n_aln per read, rows back to back) -> (PeRec array, new rng state)"""
        n = len(off) - 1
        if out is None:
            out = (PeRec * max(n, 1))()
        st = C.c_uint64(rng_state)
        _chk(lib().nabwa_pe_posn(self._h, C.byref(opt), n // 2, _ptr(np.ascontiguousarray(off, np.int64)), _ptr(np.ascontiguousarray(full_len, np.int32)),
                                 _ptr(np.ascontiguousarray(n_aln, np.int32)), _ptr(np.ascontiguousarray(rows)), C.byref(st), out))
        return out, st.value

    def pe_finish_flat(self, opt, popt, ii, seq, rseq, off, n_aln, rows, recs):
        """nabwa_pe_finish on flat arrays; recs (from pe_posn_flat) are updated in place -> (n_tot, n_mapped)"""
        n = len(off) - 1
        tot, mp = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        _chk(lib().nabwa_pe_finish(self._h, C.byref(opt), C.byref(popt), C.byref(ii), n // 2, _ptr(np.ascontiguousarray(off, np.int64)),
                                   _ptr(np.ascontiguousarray(seq, np.uint8)), _ptr(np.ascontiguousarray(rseq, np.uint8)),
                                   _ptr(np.ascontiguousarray(n_aln, np.int32)), _ptr(np.ascontiguousarray(rows)), recs, tot, mp))
        return list(tot), list(mp)

    def se_finish(self, opt, seq, rseq, off, full_len, hits, n_occ, rng_state):
        """aln2seq + positions + mapQ + gap refinement + MD/NM for a batch, in record order.
        hits: per-read arrays of bwt_aln1_t rows.  Returns (array of SeRec, new rng state)."""
        n = len(off) - 1
        n_aln = np.array([len(h) for h in hits], np.int32)
        rows = np.concatenate([np.asarray(h, ALN_DT) for h in hits] + [np.zeros(0, ALN_DT)]) if n else np.zeros(0, ALN_DT)
        rows = np.ascontiguousarray(rows)
        out = (SeRec * max(n, 1))()
        st = C.c_uint64(rng_state)
        fl = np.ascontiguousarray(full_len, np.int32)
        _chk(lib().nabwa_se_finish(self._h, C.byref(opt), n, _ptr(np.ascontiguousarray(off, np.int64)),
                                   _ptr(np.ascontiguousarray(seq, np.uint8)), _ptr(np.ascontiguousarray(rseq, np.uint8)),
                                   _ptr(fl), _ptr(n_aln), _ptr(rows), n_occ, C.byref(st), out))
        return out, st.value

    def pe_posn(self, opt, off, full_len, hits, rng_state):
        """posn_pair (reference bam2bam.c:683-703) for interleaved ends (index 2*pair + end), in record order.
        Returns (array of PeRec, new rng state)."""
        n = len(off) - 1
        assert n % 2 == 0
        n_aln = np.array([len(h) for h in hits], np.int32)
        rows = np.ascontiguousarray(np.concatenate([np.asarray(h, ALN_DT) for h in hits] + [np.zeros(0, ALN_DT)]))
        out = (PeRec * max(n, 1))()
        st = C.c_uint64(rng_state)
        fl = np.ascontiguousarray(full_len, np.int32)
        _chk(lib().nabwa_pe_posn(self._h, C.byref(opt), n // 2, _ptr(np.ascontiguousarray(off, np.int64)), _ptr(fl), _ptr(n_aln),
                                 _ptr(rows), C.byref(st), out))
        return out, st.value

    def pe_finish(self, opt, popt, ii, seq, rseq, off, hits, recs):
        """finish_pair (reference bam2bam.c:705-811) on the records pe_posn returned (updated in place).
        Returns (n_tot, n_mapped) of the mate rescue."""
        n = len(off) - 1
        n_aln = np.array([len(h) for h in hits], np.int32)
        rows = np.ascontiguousarray(np.concatenate([np.asarray(h, ALN_DT) for h in hits] + [np.zeros(0, ALN_DT)]))
        tot = (C.c_uint64 * 2)()
        mp = (C.c_uint64 * 2)()
        _chk(lib().nabwa_pe_finish(self._h, C.byref(opt), C.byref(popt), C.byref(ii), n // 2, _ptr(np.ascontiguousarray(off, np.int64)),
                                   _ptr(np.ascontiguousarray(seq, np.uint8)), _ptr(np.ascontiguousarray(rseq, np.uint8)),
                                   _ptr(n_aln), _ptr(rows), recs, tot, mp))
        return list(tot), list(mp)

    def export(self, which, what, first, n):
        """derived index parts for tests (nabwa_index_export): 0 full SA, 1 inverse SA, 2 text bases, 3 interval table, 4 its depth"""
        out = np.zeros(int(n) * (2 if what == 3 else 1), np.uint32)
        _chk(lib().nabwa_index_export(self._h, int(which), int(what), int(first), int(n), _ptr(out)))
        return out.reshape(-1, 2) if what == 3 else out

    def seq_len(self, which=0):
        return lib().nabwa_index_seq_len(self._h, which)

    def device_bytes(self):
        return lib().nabwa_index_device_bytes(self._h)

    def close(self):
        if self._h:
            lib().nabwa_index_destroy(self._h)
            self._h = None

    def sa_lookup(self, which, k):
        """bwt_sa (reference bwt.c:72-81) for many rows."""
        which = np.ascontiguousarray(which, np.uint8)
        k = np.ascontiguousarray(k, np.uint32)
        out = np.zeros(len(k), np.uint32)
        _chk(lib().nabwa_sa_lookup(self._h, len(k), _ptr(which), _ptr(k), _ptr(out)))
        return out

    def occ4(self, which, k):
        """bwt_occ4 (reference bwt.c:159-176) for many rows."""
        k = np.ascontiguousarray(k, np.uint32)
        out = np.zeros((len(k), 4), np.uint32)
        _chk(lib().nabwa_occ4(self._h, which, len(k), _ptr(k), _ptr(out)))
        return out

    def cal_sa_reg_gap_flat(self, opt, seq, rseq, off, per_read=False, cap_rows=None):
        """The C one-shot entry (nabwa_cal_sa_reg_gap) on host buffers, results as flat arrays:
        (n_aln int32[n], rows ALN_DT[total], max_entries int32[n]) -- what a C caller of the boundary gets."""
        n = len(off) - 1
        seq = np.ascontiguousarray(seq, np.uint8); rseq = np.ascontiguousarray(rseq, np.uint8)
        off = np.ascontiguousarray(off, np.int64)
        n_aln = np.empty(max(n, 1), np.int32); maxe = np.empty(max(n, 1), np.int32)
        cap = int(cap_rows if cap_rows is not None else n + n // 8 + 1024)
        while True:
            buf = np.empty(max(cap, 1), ALN_DT)
            rows = C.c_int64()
            rc = lib().nabwa_cal_sa_reg_gap(self._h, C.byref(opt), n, _ptr(off), _ptr(seq), _ptr(rseq), int(per_read),
                                            _ptr(n_aln), _ptr(buf), cap, C.byref(rows), _ptr(maxe))
            if rc == ECAP and rows.value > cap:
                cap = rows.value
                continue
            _chk(rc)
            return n_aln[:n], buf[:rows.value], maxe[:n]

    def cal_sa_reg_gap(self, opt, seq, rseq, off, per_read=False):
        """bwa_cal_sa_reg_gap (reference bwtaln.c:93-142) over a flat batch.
        Returns (list of per-read hit arrays, max_entries)."""
        b = Batch(self, opt, seq, rseq, off, per_read)
        try:
            b.run()
            b.sync()
            return b.fetch()
        finally:
            b.close()


class Batch:
    """Device-resident batch of reads: upload once, run the FM search many times."""

    def __init__(self, index, opt, seq, rseq, off, per_read=False):
        self.n = len(off) - 1
        self._seq = np.ascontiguousarray(seq, np.uint8)
        self._rseq = np.ascontiguousarray(rseq, np.uint8)
        self._off = np.ascontiguousarray(off, np.int64)
        self._h = _P()
        _chk(lib().nabwa_batch_create(index._h, C.byref(opt), self.n, _ptr(self._off), _ptr(self._seq),
                                      _ptr(self._rseq), int(per_read), C.byref(self._h)))

    def run(self):
        _chk(lib().nabwa_batch_run(self._h))

    def sync(self):
        n2 = C.c_int()
        _chk(lib().nabwa_batch_sync(self._h, C.byref(n2)))
        return n2.value

    def last_kernel_ms(self):
        return float(lib().nabwa_batch_last_kernel_ms(self._h))

    def last_width_ms(self):
        return float(lib().nabwa_batch_last_width_ms(self._h))

    def last_deep_ms(self):
        """HIP-event time of kernel D (the deep searches the first pass handed on) in the most recent run; 0 if it did not run"""
        return float(lib().nabwa_batch_last_deep_ms(self._h))

    def checksum(self):
        s = C.c_uint64()
        r = C.c_int64()
        _chk(lib().nabwa_batch_checksum(self._h, C.byref(s), C.byref(r)))
        return s.value, r.value

    def count_touches(self):
        """Occ-bucket touches of the reference algorithm on this batch (untimed instrumented run)."""
        v, w = C.c_uint64(), C.c_uint64()
        _chk(lib().nabwa_batch_count_touches(self._h, C.byref(v), C.byref(w)))
        return v.value, w.value   # (search kernel, width kernel)

    def fetch(self):
        n_aln = np.zeros(max(self.n, 1), np.int32)
        maxe = np.zeros(max(self.n, 1), np.int32)
        rows = C.c_int64()
        rc = lib().nabwa_batch_fetch(self._h, _ptr(n_aln), None, 0, C.byref(rows), _ptr(maxe))
        if rc not in (OK, ECAP):
            _chk(rc)
        buf = np.zeros(max(rows.value, 1), ALN_DT)
        if rows.value:
            _chk(lib().nabwa_batch_fetch(self._h, _ptr(n_aln), _ptr(buf), rows.value, C.byref(rows), _ptr(maxe)))
        n_aln = n_aln[:self.n]
        bounds = np.concatenate([[0], np.cumsum(n_aln)])
        return [buf[bounds[i]:bounds[i + 1]] for i in range(self.n)], maxe[:self.n]

    def fetch_flat(self, keep=None):
        """-> (n_aln per read, all rows back to back, max_entries per read).  keep: a dict the caller holds from batch to batch -- the three
        arrays live in it and are used again (a streaming caller's buffers: no fresh pages per batch, one fetch where the rows fit)"""
        if keep is not None and "n_aln" in keep and len(keep["n_aln"]) >= max(self.n, 1):
            n_aln, maxe, buf = keep["n_aln"], keep["maxe"], keep["rows"]
        else:
            n_aln = np.zeros(max(self.n, 1), np.int32)
            maxe = np.zeros(max(self.n, 1), np.int32)
            buf = None
        rows = C.c_int64()
        rc = lib().nabwa_batch_fetch(self._h, _ptr(n_aln), _ptr(buf) if buf is not None else None, len(buf) if buf is not None else 0, C.byref(rows), _ptr(maxe))
        if rc not in (OK, ECAP):
            _chk(rc)
        if rc == ECAP or buf is None:
            buf = np.zeros(max(rows.value + rows.value // 8, 1), ALN_DT)
            if rows.value:
                _chk(lib().nabwa_batch_fetch(self._h, _ptr(n_aln), _ptr(buf), len(buf), C.byref(rows), _ptr(maxe)))
        if keep is not None:
            keep["n_aln"], keep["maxe"], keep["rows"] = n_aln, maxe, buf
        return n_aln[:self.n], buf[:rows.value], maxe[:self.n]

    def close(self):
        if self._h:
            lib().nabwa_batch_destroy(self._h)
            self._h = None
