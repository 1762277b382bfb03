"""Minimal BAM record encoder / decoder for the tests (uncompressed records as they stand in a BAM stream)."""
import struct

import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"
CIG = "MIDNSHP=X"


def make_record(name, seq, qual, flag, tags=b""):
    """an unaligned record: refID -1, pos -1, no CIGAR; qual as a phred+33 string"""
    l = len(seq)
    nm = name.encode() + b"\0"
    packed = bytearray((l + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= NT16.index(c if c in NT16 else "N") << (4 if i % 2 == 0 else 0)
    q = bytes(ord(c) - 33 for c in qual)
    core = struct.pack("<iiIIiiii", -1, -1, (4680 << 16) | len(nm), flag << 16, l, -1, -1, 0)
    body = core + nm + bytes(packed) + q + tags
    return struct.pack("<I", len(body)) + body


def tag_z(k, v):
    return k.encode() + b"Z" + v.encode() + b"\0"


def tag_i(k, v):
    return k.encode() + b"i" + struct.pack("<i", v)


def tag_a(k, v):
    return k.encode() + b"A" + v.encode()


def pack(records):
    off = np.zeros(len(records) + 1, np.int64)
    np.cumsum([len(r) for r in records], out=off[1:])
    return np.frombuffer(b"".join(records), np.uint8).copy(), off


def decode(buf, off, contigs):
    """-> list of dicts with the SAM fields of every record"""
    out = []
    raw = bytes(buf)
    for i in range(len(off) - 1):
        r = raw[off[i]:off[i + 1]]
        bs, tid, pos, y, z, l_seq, mtid, mpos, tlen = struct.unpack_from("<IiiIIiiii", r, 0)
        assert bs + 4 == len(r)
        l_name, mapq, b_in = y & 0xff, y >> 8 & 0xff, y >> 16
        n_cig, flag = z & 0xffff, z >> 16
        p = 36
        name = r[p:p + l_name - 1].decode(); p += l_name
        cig = "".join("%d%s" % (c >> 4, CIG[c & 15]) for c in struct.unpack_from("<%dI" % n_cig, r, p)); p += 4 * n_cig
        seq = "".join(NT16[r[p + (j >> 1)] >> (4 if j % 2 == 0 else 0) & 15] for j in range(l_seq)); p += (l_seq + 1) // 2
        qual = "".join(chr(c + 33) for c in r[p:p + l_seq]); p += l_seq
        tags, order = {}, []
        while p < len(r):
            k, ty = r[p:p + 2].decode(), chr(r[p + 2]); p += 3
            if ty == "A":
                v = chr(r[p]); p += 1
            elif ty == "i":
                v = struct.unpack_from("<i", r, p)[0]; p += 4
            elif ty == "Z":
                e = r.index(b"\0", p); v = r[p:e].decode(); p = e + 1
            else:
                raise ValueError("tag type " + ty)
            tags[k] = v; order.append(k + ":" + ty)
        out.append(dict(name=name, flag=flag, rname=contigs[tid] if tid >= 0 else "*", pos=pos + 1, mapq=mapq, cigar=cig or "*",
                        rnext="*" if mtid < 0 else ("=" if mtid == tid else contigs[mtid]), pnext=mpos + 1, tlen=tlen, seq=seq, qual=qual,
                        tags=tags, order=order, bin=b_in, tid=tid))
    return out
