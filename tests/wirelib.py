"""ctypes face of the wire record (include/nabwa.h: nabwa_wire_*) for the tests, and a builder of messages BY HAND from the reference's
description (bam2bam.c:951-1006) that the codec is checked against."""
import ctypes as C
import struct

P = C.c_void_p


class WireRead(C.Structure):
    _fields_ = [("core", C.c_uint8 * 32), ("data_len", C.c_int32), ("data", P),
                ("strand", C.c_uint8), ("type", C.c_uint8), ("n_mm", C.c_uint8), ("n_gapo", C.c_uint8), ("n_gape", C.c_uint8), ("seQ", C.c_uint8), ("mapQ", C.c_uint8),
                ("len", C.c_int32), ("clip_len", C.c_int32), ("score", C.c_int32), ("sa", C.c_uint32), ("c1", C.c_uint32), ("c2", C.c_uint32), ("pos", C.c_uint32),
                ("n_multi", C.c_int32), ("multi", P), ("max_entries", C.c_int32), ("n_aln", C.c_int32), ("aln", P)]


class WireRec(C.Structure):
    _fields_ = [("recno", C.c_uint64), ("kind", C.c_uint8), ("phase", C.c_uint8), ("read", WireRead * 2)]


def bind(L):
    L.nabwa_wire_size.restype = C.c_int64
    L.nabwa_wire_size.argtypes = [P]
    L.nabwa_wire_encode.restype = C.c_int64
    L.nabwa_wire_encode.argtypes = [P, P, C.c_int64]
    L.nabwa_wire_decode.argtypes = [P, C.c_int64, P]
    L.nabwa_wire_core_from_bam.argtypes = [P, P]
    L.nabwa_wire_core_to_bam.argtypes = [P, P]
    L.nabwa_wire_config_encode.restype = C.c_int64
    L.nabwa_wire_config_encode.argtypes = [P, P, C.c_char_p, P, C.c_int64]
    L.nabwa_wire_config_decode.argtypes = [P, C.c_int64, P, P, P, C.c_int]
    return L


def host_core(bam_core):
    """bam1_core_t as bamlite.c's reader fills it (bin:16 | qual:8 | l_qname:8, flag:16 | n_cigar:16) from the 32 bytes of a BAM file's record"""
    w = list(struct.unpack("<8I", bam_core))
    b, q, l = w[2] >> 16, w[2] >> 8 & 0xff, w[2] & 0xff
    f, n = w[3] >> 16, w[3] & 0xffff
    w[2] = b | q << 16 | l << 24
    w[3] = f | n << 16
    return struct.pack("<8I", *w)


def message(recno, kind, phase, reads):
    """msg_init_from_pair by hand.  reads: dicts with bam (the record as it stands in a BAM stream: block_size, core, data) and, by phase,
    strand type n_mm n_gapo n_gape seQ mapQ len clip_len score sa c1 c2 pos multi (list of 16-byte strings) / max_entries aln (list of 16-byte strings)"""
    m = struct.pack("<QBB", recno, kind, phase)
    for r in reads[:kind]:
        bam = r["bam"]
        m += host_core(bam[4:36]) + struct.pack("<i", len(bam) - 36) + bam[36:]
        if phase == 2:
            m += bytes([r["strand"] << 4 | r["type"], r["n_mm"], r["n_gapo"], r["n_gape"], r["seQ"], r["mapQ"]])
            m += struct.pack("<iiiIIIIi", r["len"], r["clip_len"], r["score"], r["sa"], r["c1"], r["c2"], r["pos"], len(r["multi"]))
            m += b"".join(r["multi"])
        if phase in (1, 2):
            m += struct.pack("<ii", r["max_entries"], len(r["aln"])) + b"".join(r["aln"])
    return m


def decode(L, msg):
    rec = WireRec()
    buf = (C.c_uint8 * len(msg)).from_buffer_copy(msg)
    rc = L.nabwa_wire_decode(buf, len(msg), C.byref(rec))
    return rc, rec, buf


def encode(L, rec):
    n = L.nabwa_wire_size(C.byref(rec))
    assert n > 0
    out = (C.c_uint8 * n)()
    assert L.nabwa_wire_encode(C.byref(rec), out, n) == n
    return bytes(out)
