"""Bench / test infrastructure: synthetic genome, reads and GPU-built FM-indexes (libnabwa_synth.so).

Not part of the drop-in ABI.  The arrays it produces have exactly the byte layout of the reference's
.bwt / .sa files (reference bwtio.c:161-204) and enter the product through
``nabwa_index_from_arrays`` like a real index would.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnabwa_synth.so")
_P = C.c_void_p
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libnabwa_synth.so is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.nabwa_synth_last_error.restype = C.c_char_p
        L.nabwa_synth_text.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, _P]
        L.nabwa_synth_reads.argtypes = [C.c_int, _P, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64,
                                        _P, _P, _P]
        L.nabwa_synth_build_index.argtypes = [C.c_int, _P, C.c_uint64, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int]
        L.nabwa_synth_free.argtypes = [_P]
        L.nabwa_synth_gather_bench.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]
        L.nabwa_synth_malloc.argtypes = [C.c_int, C.c_uint64, _P]
        L.nabwa_synth_d2h.argtypes = [_P, _P, C.c_uint64]
        L.nabwa_synth_h2d.argtypes = [_P, _P, C.c_uint64]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError("libnabwa_synth: " + lib().nabwa_synth_last_error().decode())


class DevArray:
    """A hipMalloc'd array owned by this object."""

    def __init__(self, ptr, nbytes, device=0):
        self.ptr, self.nbytes, self.device = ptr, nbytes, device

    @classmethod
    def empty(cls, nbytes, device=0):
        p = _P()
        _chk(lib().nabwa_synth_malloc(device, max(int(nbytes), 16), C.byref(p)))
        return cls(p.value, int(nbytes), device)

    @classmethod
    def from_host(cls, a, device=0):
        a = np.ascontiguousarray(a)
        d = cls.empty(a.nbytes, device)
        _chk(lib().nabwa_synth_h2d(_P(d.ptr), a.ctypes.data_as(_P), a.nbytes))
        return d

    def to_host(self, dtype, count=None, offset_bytes=0):
        dt = np.dtype(dtype)
        count = (self.nbytes - offset_bytes) // dt.itemsize if count is None else count
        out = np.empty(count, dt)
        _chk(lib().nabwa_synth_d2h(out.ctypes.data_as(_P), _P(self.ptr + offset_bytes), count * dt.itemsize))
        return out

    def free(self):
        if self.ptr:
            lib().nabwa_synth_free(_P(self.ptr))
            self.ptr = 0


def synth_text(n, seed, n_dup=0, dup_len=0, device=0):
    """uniform-random ACGT codes (one byte per base) with planted repeats, on the device"""
    p = _P()
    _chk(lib().nabwa_synth_text(device, int(n), int(seed), int(n_dup), int(dup_len), C.byref(p)))
    return DevArray(p.value, int(n), device)


def synth_text_repeats(n, seed, device=0):
    """random ACGT with planted repeat families (LINE-like, Alu-like with a young subfamily, tandem repeats; synth_index.hip)"""
    p = _P()
    lib().nabwa_synth_text_repeats.argtypes = [C.c_int, C.c_uint64, C.c_uint64, _P]
    _chk(lib().nabwa_synth_text_repeats(device, int(n), int(seed), C.byref(p)))
    return DevArray(p.value, int(n), device)


def build_index(d_text, n, reverse, sa_intv=32, with_sa=True, device=0, verbose=False):
    """-> (bwt_words DevArray, n_words, sa_words DevArray|None, n_sa_words): content of .bwt/.sa files"""
    bw, sw = _P(), _P()
    nb, ns = C.c_uint64(), C.c_uint64()
    _chk(lib().nabwa_synth_build_index(device, _P(d_text.ptr), int(n), int(reverse), int(sa_intv), C.byref(bw),
                                       C.byref(nb), C.byref(sw) if with_sa else None,
                                       C.byref(ns) if with_sa else None, int(verbose)))
    return (DevArray(bw.value, nb.value * 4, device), nb.value,
            DevArray(sw.value, ns.value * 4, device) if with_sa else None, ns.value if with_sa else 0)


def synth_reads(d_text, n, n_reads, length, sub_ppm, indel_ppm, seed, device=0):
    """reads sampled from the text -> host arrays (seq, rseq, off) in bwa_seq_t encoding"""
    ds = DevArray.empty(n_reads * length, device)
    dr = DevArray.empty(n_reads * length, device)
    do = DevArray.empty((n_reads + 1) * 8, device)
    _chk(lib().nabwa_synth_reads(device, _P(d_text.ptr), int(n), int(n_reads), int(length), int(sub_ppm),
                                 int(indel_ppm), int(seed), _P(ds.ptr), _P(dr.ptr), _P(do.ptr)))
    seq, rseq, off = ds.to_host(np.uint8), dr.to_host(np.uint8), do.to_host(np.int64)
    ds.free()
    dr.free()
    do.free()
    return seq, rseq, off


def gather_ceiling(table_bytes=2 << 30, bytes_per_access=64, chains=1, n_blocks=1024, steps=2000, device=0):
    """measured throughput of dependent random gathers (GB/s, M accesses/s): the attainable bound of the FM search"""
    g, m = C.c_double(), C.c_double()
    _chk(lib().nabwa_synth_gather_bench(device, int(table_bytes), int(bytes_per_access), int(chains), int(n_blocks),
                                        int(steps), C.byref(g), C.byref(m)))
    return g.value, m.value


def synth_pairs(d_text, n, n_pairs, length, sub_ppm, indel_ppm, isize_mean, isize_sd, seed, device=0):
    """pairs sampled from the text (FR, insert ~ N(mean, sd)), reads interleaved 2*pair + end -> host arrays (seq, rseq, off)"""
    m = 2 * n_pairs
    ds = DevArray.empty(m * length, device)
    dr = DevArray.empty(m * length, device)
    do = DevArray.empty((m + 1) * 8, device)
    lib().nabwa_synth_pairs.argtypes = [C.c_int, _P, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_uint64, _P, _P, _P]
    _chk(lib().nabwa_synth_pairs(device, _P(d_text.ptr), int(n), int(n_pairs), int(length), int(sub_ppm), int(indel_ppm),
                                 float(isize_mean), float(isize_sd), int(seed), _P(ds.ptr), _P(dr.ptr), _P(do.ptr)))
    seq, rseq, off = ds.to_host(np.uint8), dr.to_host(np.uint8), do.to_host(np.int64)
    ds.free()
    dr.free()
    do.free()
    return seq, rseq, off
