"""The record `bwa bam2bam -p` and `bwa worker` exchange (msg_init_from_pair / pair_init_from_msg, reference bam2bam.c:951-1097) and the
configuration reply (:1255-1266), through the library's codec (csrc/wire_worker.cpp).  PARITY UNPINNED: libzmq is absent and bam2bam.c
cannot be compiled, so there is no reference message to compare with; the expected bytes are built by hand, field by field, from the
cited lines (tests/wirelib.py), and the struct sizes the raw blocks rely on are checked against the reference's headers."""
import ctypes as C
import importlib
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import bamlib as B
import nabwa_testlib as T
import wirelib as W

nabwa = importlib.import_module("network-aware-bwa_amd")


@pytest.fixture(scope="module")
def L():
    return W.bind(nabwa.lib())


def sample_reads():
    r1 = B.make_record("pair/1", "ACGTACGTACGTAACCGGTT", "IIIIIIIIIIIIIIIIIIII", 1 | 64 | 4 | 8, B.tag_z("RG", "lib1"))
    r2 = B.make_record("pair/1", "TTGGCCAATTGGCCAATTGG", "IIIIIHHHHHGGGGGFFFFF", 1 | 128 | 4 | 8, B.tag_z("RG", "lib1") + B.tag_i("ZZ", 7))
    aln = [struct.pack("<IIIi", 1 | 0 << 8 | 0 << 16 | 1 << 24, 100, 102, 3), struct.pack("<IIIi", 2, 5000, 5000, 6)]
    multi = [struct.pack("<II", 777, 3 << 15 | 2 << 23 | 1 << 31) + b"\xde\xad\xbe\xef\0\0\0\0"]
    d1 = dict(bam=r1, strand=1, type=2, n_mm=1, n_gapo=0, n_gape=0, seQ=23, mapQ=25, len=20, clip_len=20, score=3, sa=101, c1=3, c2=1, pos=123456789,
              multi=multi, max_entries=417, aln=aln)
    d2 = dict(bam=r2, strand=0, type=1, n_mm=0, n_gapo=1, n_gape=2, seQ=37, mapQ=37, len=20, clip_len=20, score=19, sa=4000000000, c1=1, c2=0, pos=4100000000,
              multi=[], max_entries=90, aln=aln[:1])
    return d1, d2


@pytest.mark.parametrize("kind,phase", [(0, 0), (1, 0), (2, 0), (1, 1), (2, 1), (1, 2), (2, 2), (2, 3), (0, 3)])
def test_decode_then_encode_gives_the_bytes_the_reference_lines_describe(L, kind, phase):
    d1, d2 = sample_reads()
    msg = W.message(0x0123456789abcdef, kind, phase, [d1, d2])
    # the sizes msg_init_from_pair adds up (bam2bam.c:954-966)
    want = 10
    for d in (d1, d2)[:kind]:
        want += 32 + 4 + len(d["bam"]) - 36 + (38 + 16 * len(d["multi"]) if phase == 2 else 0) + (8 + 16 * len(d["aln"]) if phase in (1, 2) else 0)
    assert len(msg) == want
    rc, rec, keep = W.decode(L, msg)
    assert rc == 0, nabwa.lib().nabwa_last_error()
    assert (rec.recno, rec.kind, rec.phase) == (0x0123456789abcdef, kind, phase)
    for e, d in enumerate((d1, d2)[:kind]):
        x = rec.read[e]
        assert bytes(x.core) == W.host_core(d["bam"][4:36]) and x.data_len == len(d["bam"]) - 36
        assert C.string_at(x.data, x.data_len) == d["bam"][36:]
        if phase == 2:
            got = (x.strand, x.type, x.n_mm, x.n_gapo, x.n_gape, x.seQ, x.mapQ, x.len, x.clip_len, x.score, x.sa, x.c1, x.c2, x.pos, x.n_multi)
            assert got == tuple(d[k] for k in ("strand", "type", "n_mm", "n_gapo", "n_gape", "seQ", "mapQ", "len", "clip_len", "score", "sa", "c1", "c2", "pos")) + (len(d["multi"]),)
            assert C.string_at(x.multi, 16 * x.n_multi) == b"".join(d["multi"])
        if phase in (1, 2):
            assert (x.max_entries, x.n_aln) == (d["max_entries"], len(d["aln"])) and C.string_at(x.aln, 16 * x.n_aln) == b"".join(d["aln"])
    assert L.nabwa_wire_size(C.byref(rec)) == len(msg)
    assert W.encode(L, rec) == msg


def test_field_offsets_of_a_positioned_single_read(L):
    """the positioned block byte by byte (bam2bam.c:982-998): strand << 4 | type, five bytes, eight little-endian words, the other hits, then
    max_entries, n_aln, the rows"""
    d1, _ = sample_reads()
    msg = W.message(5, 1, 2, [d1])
    o = 10 + 32 + 4 + len(d1["bam"]) - 36
    assert msg[:8] == struct.pack("<Q", 5) and msg[8] == 1 and msg[9] == 2
    assert msg[o] == (1 << 4 | 2) and list(msg[o + 1:o + 6]) == [1, 0, 0, 23, 25]
    assert struct.unpack_from("<iiiIIIIi", msg, o + 6) == (20, 20, 3, 101, 3, 1, 123456789, 1)
    assert msg[o + 38:o + 54] == d1["multi"][0]
    assert struct.unpack_from("<ii", msg, o + 54) == (417, 2)
    rc, rec, keep = W.decode(L, msg)
    assert rc == 0 and W.encode(L, rec) == msg


def test_a_message_must_be_consumed_exactly(L):
    d1, d2 = sample_reads()
    msg = W.message(1, 2, 2, [d1, d2])
    for bad in (msg[:-1], msg + b"\0", msg[:9], msg[:40], msg[:8] + bytes([3]) + msg[9:], msg[:9] + bytes([4]) + msg[10:]):
        rc, _, _ = W.decode(L, bad)
        assert rc == nabwa.EINVAL
    grown = bytearray(W.message(1, 1, 1, [d1]))                 # a row count that claims more than the message holds
    struct.pack_into("<i", grown, len(grown) - 16 * 2 - 4, 1 << 30)
    assert W.decode(L, bytes(grown))[0] == nabwa.EINVAL


def test_core_between_file_order_and_host_order(L):
    core = struct.pack("<iiIIiiii", 3, 1234567, 4681 << 16 | 37 << 8 | 9, 0x63 << 16 | 5, 100, 3, 1234999, 532)
    host = (C.c_uint8 * 32)()
    L.nabwa_wire_core_from_bam(core, host)
    assert bytes(host) == W.host_core(core)
    w = struct.unpack("<8I", bytes(host))
    assert w[2] == 4681 | 37 << 16 | 9 << 24 and w[3] == 0x63 | 5 << 16          # bin:16 qual:8 l_qname:8 / flag:16 n_cigar:16 from bit 0 up
    back = (C.c_uint8 * 32)()
    L.nabwa_wire_core_to_bam(host, back)
    assert bytes(back) == core


def test_config_reply(L):
    opt = nabwa.gap_init_opt()
    opt.max_gapo, opt.fnr = 2, 0.01
    po = nabwa.pe_opt_default()
    po.max_isize = 777
    buf = (C.c_uint8 * 512)()
    n = L.nabwa_wire_config_encode(C.byref(opt), C.byref(po), b"/data/hg19/whole_genome", buf, 512)
    assert n == 64 + 48 + len(b"/data/hg19/whole_genome")                   # gap_opt_t . pe_opt_t . prefix without a terminator (bam2bam.c:1260-1263)
    raw = bytes(buf[:n])
    assert raw[:64] == bytes(opt) and raw[64:112] == bytes(po) and raw[112:] == b"/data/hg19/whole_genome"
    o2, p2, pre = nabwa.GapOpt(), nabwa.PeOpt(), C.create_string_buffer(256)
    assert L.nabwa_wire_config_decode(raw, n, C.byref(o2), C.byref(p2), pre, 256) == 0
    assert bytes(o2) == bytes(opt) and bytes(p2) == bytes(po) and pre.value == b"/data/hg19/whole_genome"
    assert L.nabwa_wire_config_decode(raw, 100, C.byref(o2), C.byref(p2), pre, 256) == nabwa.EINVAL
    assert L.nabwa_wire_config_decode(raw, n, C.byref(o2), C.byref(p2), pre, 8) == nabwa.ECAP
    assert L.nabwa_wire_config_encode(C.byref(opt), C.byref(po), b"x", buf, 10) == nabwa.ECAP


def test_raw_block_sizes_are_the_reference_structs():
    """the message carries bam1_core_t, bwt_multi1_t, bwt_aln1_t, gap_opt_t and pe_opt_t as they lie in memory: their sizes (and the bit
    order of bam1_core_t) asked of the C compiler with the reference's own headers"""
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference's headers are only in the build container")
    src = r'''
#include <stdio.h>
#include <string.h>
#include "bamlite.h"
#include "bwtaln.h"
int main(void) {
    bam1_core_t c; unsigned w[8]; bwt_multi1_t m; unsigned mw[4];
    memset(&c, 0, sizeof c); c.bin = 4681; c.qual = 37; c.l_qname = 9; c.flag = 0x63; c.n_cigar = 5; memcpy(w, &c, 32);
    memset(&m, 0, sizeof m); m.pos = 777; m.gap = 3; m.mm = 2; m.strand = 1; memcpy(mw, &m, 16);
    printf("%zu %zu %zu %zu %zu %u %u %u %u\n", sizeof(bam1_core_t), sizeof(bwt_multi1_t), sizeof(bwt_aln1_t), sizeof(gap_opt_t), sizeof(pe_opt_t), w[2], w[3], mw[0], mw[1]);
    return 0; }
'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I/root/reference", "-o", os.path.join(td, "t"), os.path.join(td, "t.c")], check=True)
        out = subprocess.run([os.path.join(td, "t")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [32, 16, 16, 64, 48, 4681 | 37 << 16 | 9 << 24, 0x63 | 5 << 16, 777, 3 << 15 | 2 << 23 | 1 << 31]
