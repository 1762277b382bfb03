"""The input side of nabwa_bam2bam (csrc/bgzf_in.hpp) on the CPU: BGZF blocks (inflated many at a time), gzip members without the
BGZF field, one gzip stream and a file that is not compressed all give the same bytes -- what the reference's bamlite gets from
gzread (bamlite.h:7-11); a cut or damaged BGZF file is an error, not a short read."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EOF_BLOCK = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bgzf") / "bgzf_in_test")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", out, os.path.join(ROOT, "tests", "emu", "bgzf_in_main.cpp"), "-lz", "-lpthread"], check=True)
    return out


def bgzf(raw, block=0xff00, extra_first=False):
    out = b""
    for o in range(0, len(raw), block):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = c.compress(raw[o:o + block]) + c.flush()
        if extra_first:                       # another subfield in front of BC: the header is longer, BSIZE counts it
            xtra = b"XY" + struct.pack("<H", 3) + b"abc" + b"BC" + struct.pack("<HH", 2, len(d) + 12 + 13 + 8 - 1)
        else:
            xtra = b"BC" + struct.pack("<HH", 2, len(d) + 12 + 6 + 8 - 1)
        out += bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255]) + struct.pack("<H", len(xtra)) + xtra + d + struct.pack("<II", zlib.crc32(raw[o:o + block]), len(raw[o:o + block]))
    return out + EOF_BLOCK


def run(exe, path):
    r = subprocess.run([exe, path], capture_output=True, timeout=120)
    return r.returncode, r.stdout, r.stderr.decode()


def test_every_container_gives_the_same_bytes(exe, tmp_path):
    rng = np.random.default_rng(3)
    raw = b"BAM\1" + bytes(rng.integers(0, 7, 3_000_000, dtype=np.uint8)) + bytes(rng.integers(0, 256, 200_000, dtype=np.uint8))
    forms = {
        "bgzf": (bgzf(raw), "bgzf"),
        "bgzf_small_blocks": (bgzf(raw, 1000), "bgzf"),
        "bgzf_other_subfield": (bgzf(raw, 40000, extra_first=True), "bgzf"),
        "members": (b"".join(gzip.compress(raw[o:o + 300_000], 1) for o in range(0, len(raw), 300_000)), "stream"),
        "stream": (gzip.compress(raw, 1), "stream"),
        "plain": (raw, "raw"),
    }
    for name, (data, kind) in forms.items():
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        rc, out, err = run(exe, p)
        assert rc == 0 and err.strip() == kind, (name, rc, err)
        assert out == raw, name


def test_empty_payload_and_lone_end_marker(exe, tmp_path):
    for name, data in (("only_marker", EOF_BLOCK), ("empty_gzip", gzip.compress(b""))):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        rc, out, err = run(exe, p)
        assert rc == 0 and out == b"", (name, rc, err)


def test_damage_is_an_error(exe, tmp_path):
    raw = bytes(np.random.default_rng(4).integers(0, 5, 500_000, dtype=np.uint8))
    good = bgzf(raw)
    cases = {
        "cut_in_a_block": good[:len(good) // 2],
        "flipped_payload": good[:5000] + bytes([good[5000] ^ 0x41]) + good[5001:],
        "wrong_size_in_trailer": good[:len(good) - len(EOF_BLOCK) - 4] + struct.pack("<I", 7) + EOF_BLOCK,
    }
    for name, data in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        rc, out, err = run(exe, p)
        assert rc == 3 and ("BGZF" in err), (name, rc, err)
        assert len(out) < len(raw)
