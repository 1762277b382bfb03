"""The header half of `bwa bam2bam` on the host (csrc/bam_header.hpp): which old @PG line the new one names in PP: follows the slot order
of the reference's string hash set (khash.h; bam2bam.c:212-271).  bam2bam.c cannot be compiled here (<zmq.h>), but khash.h can: the
layout restated in WordSlots is compared with the reference's own set (oracle/_ref, ref_khash_str_order) on random ID sets that
cross several table sizes, and find_pp_tag with a choice made from that order."""
import ctypes as C
import os
import random
import subprocess
import tempfile

import pytest

import nabwa_testlib as T


@pytest.fixture(scope="module")
def pp_tool():
    out = os.path.join(tempfile.mkdtemp(prefix="nabwa_pp_"), "pp_main")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", out, os.path.join(T.ROOT, "tests", "emu", "pp_main.cpp")], check=True)
    return out


def ref_order(ref, words):
    arr = (C.c_char_p * len(words))(*[w.encode() for w in words])
    order = (C.c_int * len(words))()
    ref.ref_khash_str_order.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    m = ref.ref_khash_str_order(len(words), arr, order)
    return [words[order[i]] for i in range(m)]


def words_for(rng, n):
    alphabet = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-_."
    out = []
    while len(out) < n:
        w = "".join(rng.choice(alphabet) for _ in range(rng.randint(1, 12)))
        if w not in out:
            out.append(w)
    return out


def test_slot_order_is_the_reference_hash_sets(pp_tool):
    ref = T.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    rng = random.Random(5)
    for n in [1, 2, 3, 4, 8, 9, 17, 18, 40, 41, 42, 75, 150, 400, 1200]:          # across the growth steps 3, 11, 23, 53, 97, 193, 389, 769, 1543
        for _ in range(3):
            words = words_for(rng, n)
            got = subprocess.run([pp_tool, "--slots"], input="\n".join(words), capture_output=True, text=True, check=True).stdout.split("\n")[:-1]
            assert got == ref_order(ref, words), (n, words[:5])


def test_pp_names_the_first_unlinked_pg_in_set_order(pp_tool):
    ref = T.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    rng = random.Random(11)
    for trial in range(40):
        ids = words_for(rng, rng.randint(1, 30))
        if trial % 3 == 0:
            ids[0] = "bwa"
            if trial % 6 == 0 and len(ids) > 1:
                ids[1] = "bwa-1"
        linked = [x for x in ids if rng.random() < 0.5]
        lines = ["@HD\tVN:1.0\tSO:unsorted", "@SQ\tSN:c1\tLN:100"]
        for x in ids:
            pred = rng.choice(linked) if (linked and rng.random() < 0.6) else None
            lines.append("@PG\tID:%s\tPN:x%s\tCL:ID:decoy PP:decoy" % (x, "\tPP:" + pred if pred else ""))
        lines.append("@CO\tID:notapg\tPP:neither")
        text = "\n".join(lines) + "\n"
        # what the reference computes: IDs into one set and PP values into another in file order, the first ID in slot order that no PP names
        named = [l.split("\tPP:")[1].split("\t")[0] for l in lines if l.startswith("@PG") and "\tPP:" in l]
        want = next((x for x in ref_order(ref, ids) if x not in named), "-")
        me = "bwa"
        k = 1
        while me in ids:
            me = "bwa-%d" % k
            k += 1
        got = subprocess.run([pp_tool], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
        assert got[0] == want and got[1] == me, (trial, ids, named)
