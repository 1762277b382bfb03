"""`nabwa_bam2bam` (csrc/bam2bam_main.cpp): the command line of `bwa bam2bam -t 1` -- BAM file in, BGZF BAM file out: header
(bam2bam.c:164-301), records in input order through both passes, several batches, single-end and paired records in one file,
the `name_` -> `name` rename on success (utils.c:159-173).  Record fields are pinned to the reference's samse / sampe SAM."""
import gzip
import importlib
import os
import struct
import subprocess

import numpy as np
import pytest

import bamlib as B
import nabwa_testlib as T
from test_gpu_bam import pe_records, revcomp, toy_ann

nabwa = importlib.import_module("network-aware-bwa_amd")
pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(nabwa.LIB_PATH), "nabwa_bam2bam")
OLD_HEADER = "@HD\tVN:1.0\tSO:unsorted\n@SQ\tSN:stale\tLN:5\n@RG\tID:lib1\tSM:x\n@PG\tID:first\tPN:demux\n@PG\tID:bwa\tPN:bwa\tPP:first\n@CO\tkept as it is\n"


def write_bam(path, records, bgzf):
    text = OLD_HEADER.encode()
    raw = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 6) + b"stale\0" + struct.pack("<i", 5) + b"".join(records)
    if bgzf == "blocks":                          # true BGZF: members of <= 0xff00 bytes that carry their size in a BC extra field
        import zlib
        with open(path, "wb") as f:
            for o in range(0, len(raw), 0xff00):
                c = zlib.compressobj(6, zlib.DEFLATED, -15)
                d = c.compress(raw[o:o + 0xff00]) + c.flush()
                f.write(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(raw[o:o + 0xff00]), min(0xff00, len(raw) - o)))
            f.write(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    elif bgzf == "plain":                         # not compressed at all: gzread hands such a file through as it is, and so does the reader here
        open(path, "wb").write(raw)
    elif bgzf:                                    # several gzip members without that field
        with open(path, "wb") as f:
            for o in range(0, len(raw), 40000):
                f.write(gzip.compress(raw[o:o + 40000], 1))
    else:
        with gzip.open(path, "wb", 1) as f:
            f.write(raw)


def read_bam(path):
    raw = gzip.decompress(open(path, "rb").read())
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, p)[0]; p += 4
        nm = raw[p:p + ln - 1].decode(); p += ln
        refs.append((nm, struct.unpack_from("<i", raw, p)[0])); p += 4
    off = [0]
    body = raw[p:]
    q = 0
    while q < len(body):
        q += 4 + struct.unpack_from("<I", body, q)[0]
        off.append(q)
    assert q == len(body)
    return text, refs, B.decode(np.frombuffer(body, np.uint8), np.array(off, np.int64), [r[0] for r in refs])


def run(tmp_path, records, args, env=None, bgzf=True, prefix=None):
    if not os.path.exists(EXE):
        nabwa.build()
    inp, outp = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    write_bam(inp, records, bgzf)
    e = dict(os.environ, **(env or {}))
    r = subprocess.run([EXE, "-g", prefix or T.TOY, "-f", outp + "_"] + args + [inp], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(outp) and not os.path.exists(outp + "_")
    raw = open(outp, "rb").read()
    assert raw[-28:] == bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])      # the BGZF end-of-file block
    return read_bam(outp)


def check_se(out, sam):
    for g, w in zip(out, sam):
        assert g["name"] == w["name"] and g["flag"] == w["flag"], w["name"]
        assert (g["rname"], g["pos"], g["mapq"], g["cigar"], g["seq"], g["qual"]) == (w["rname"], w["pos"], w["mapq"], w["cigar"], w["seq"], w["qual"]), w["name"]
        assert g["tags"] == w["tags"], w["name"]


def test_header_and_single_end_records(tmp_path):
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    sam = T.parse_sam(os.path.join(T.GOLDEN, "se_default.sam"))
    recs = [B.make_record(n, s, q, 4) for n, s, q in reads]
    text, refs, out = run(tmp_path, recs, [], env={"NABWA_BAM_BATCH": "100"})        # seven batches, one RNG stream
    lines = text.split("\n")
    l_pac, contigs = toy_ann()
    assert lines[0] == "@HD\tVN:1.4"
    assert lines[1].startswith("@PG\tID:bwa-1\tPP:bwa\tPN:bwa\tVN:") and "\tCL:" in lines[1] and lines[1].endswith("in.bam")     # the id "bwa" is taken; the old "bwa" entry is the one nothing links to yet
    assert lines[2:2 + len(contigs)] == ["@SQ\tSN:%s\tLN:%d" % (c, l) for c, l in refs] and [r[0] for r in refs] == contigs
    assert lines[2 + len(contigs):] == OLD_HEADER.split("\n")[2:]                      # every old line but @HD / @SQ, in order
    assert len(out) == len(sam)
    check_se(out, sam)


def test_options_reach_the_search(tmp_path):
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    sam = T.parse_sam(os.path.join(T.GOLDEN, "se_adna.sam"))
    recs = [B.make_record(n, s, q, 4) for n, s, q in reads]
    _, _, out = run(tmp_path, recs, ["-n", "0.01", "-o", "2", "-l", "16500"], bgzf=False)
    check_se(out, sam)
    sam = T.parse_sam(os.path.join(T.GOLDEN, "se_q20.sam"))
    _, _, out = run(tmp_path, recs, ["--trim-quality", "20"])
    check_se(out, sam)


def test_pairs_and_singletons_in_one_file(tmp_path):
    """pairs need the insert-size estimate of the whole file (the barrier between the passes); the output keeps the input order"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:50]
    singles = [B.make_record(n, s, q, 4) for n, s, q in reads]
    pairs, n_pairs = pe_records()
    _, _, out = run(tmp_path, singles + pairs + singles, [], env={"NABWA_BAM_BATCH": "128"})
    assert len(out) == 100 + 2 * n_pairs
    assert [o["name"] for o in out[:50]] == [r[0] for r in reads] == [o["name"] for o in out[-50:]]
    assert all(o["flag"] & 1 for o in out[50:-50]) and not any(o["flag"] & 1 for o in out[:50])
    # the first singletons share their batch with pairs: that batch waits for pass 2 with its pass-1 state packed away, and they
    # must come out of it as they come out of a file of singletons (same random stream up to there, other hits and all)
    _, _, alone = run(tmp_path, singles, [])
    assert core(out[:50]) == core(alone) and any("XA" in o["tags"] for o in alone)
    # the pairs: same answers as one library batch under bam2bam's own estimate (tests/test_gpu_bam.py pins that route);
    # here: mates are consistent with each other
    for i in range(50, 50 + 2 * n_pairs, 2):
        a, b = out[i], out[i + 1]
        assert a["name"] == b["name"] and (a["flag"] & 64) and (b["flag"] & 128)
        if not (a["flag"] & 4) and not (b["flag"] & 4):
            assert a["pnext"] == b["pos"] and b["pnext"] == a["pos"] and a["tlen"] == -b["tlen"]


def test_only_aligned_drops_logical_records_with_an_unmapped_read(tmp_path):
    """--only-aligned (pair_print_bam, bam2bam.c:910-922): a single read that stays unmapped is not written, and neither is a pair with
    an unmapped mate -- both of its records go"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:120]
    singles = [B.make_record(n, s, q, 4) for n, s, q in reads]
    pairs, n_pairs = pe_records()
    _, _, full = run(tmp_path, singles + pairs, [])
    _, _, only = run(tmp_path, singles + pairs, ["--only-aligned"])
    want = []
    i = 0
    while i < len(full):
        k = 2 if full[i]["flag"] & 1 else 1
        if not any(r["flag"] & 4 for r in full[i:i + k]):
            want += full[i:i + k]
        i += k
    assert 0 < len(want) < len(full)
    assert [(r["name"], r["flag"], r["pos"], r["cigar"]) for r in only] == [(r["name"], r["flag"], r["pos"], r["cigar"]) for r in want]


def test_bgzf_blocks_are_inflated_in_parallel_and_damage_is_noticed(tmp_path):
    """a true BGZF file (blocks with the BC field: inflated several at a time), the same bytes as members without the field and as
    one gzip stream give the same records; a block cut short or with a wrong checksum ends the run"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    recs = [B.make_record(n, s, q, 4, B.tag_z("ZZ", "x" * 300)) for n, s, q in reads] * 3          # 500 KB: several blocks
    outs = [core(run(tmp_path, recs, [], bgzf=kind, env={"NABWA_BAM_BATCH": "700"})[2]) for kind in ("blocks", True, False, "plain")]
    assert outs[0] == outs[1] == outs[2] == outs[3] and len(outs[0]) == len(recs)
    inp = str(tmp_path / "in.bam")
    write_bam(inp, recs, "blocks")
    raw = open(inp, "rb").read()
    for name, damaged in (("cut", raw[:len(raw) // 2]), ("flip", raw[:30000] + bytes([raw[30000] ^ 0x55]) + raw[30001:])):
        bad = str(tmp_path / (name + ".bam"))
        open(bad, "wb").write(damaged)
        r = subprocess.run([EXE, "-g", T.TOY, "-f", str(tmp_path / "bad_out.bam"), bad], capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and ("BGZF" in r.stderr or "truncated" in r.stderr), r.stderr[-500:]


def test_two_index_replicas_take_the_batches_in_turn(tmp_path):
    """NABWA_DEVICES names the GPUs (here GPU 0 twice: two replicas, two search threads): batches are dealt to them in turn and
    searched as they come, pass 1 and pass 2 still see them in input order on one random stream -- the records do not change"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
    singles = [B.make_record(n, s, q, 4) for n, s, q in reads]
    pairs, n_pairs = pe_records()
    recs = singles[:300] + pairs + singles[300:]
    opts = ["-n", "0.01", "-o", "2", "-l", "16500"]
    _, _, one = run(tmp_path, recs, opts, env={"NABWA_BAM_BATCH": "96"})
    _, _, two = run(tmp_path, recs, opts, env={"NABWA_BAM_BATCH": "96", "NABWA_DEVICES": "0,0"})
    assert core(two) == core(one) and len(one) == len(recs)
    sam = T.parse_sam(os.path.join(T.GOLDEN, "se_adna.sam"))
    check_se(two[:300], sam[:300])


def test_a_file_without_records_gives_a_header_and_nothing_else(tmp_path):
    for kind in ("blocks", False):
        text, refs, out = run(tmp_path, [], [], bgzf=kind)
        assert out == [] and text.startswith("@HD\tVN:1.4\n@PG\tID:bwa-1") and [r[0] for r in refs] == toy_ann()[1]
    # nothing maps: --only-aligned leaves nothing to write
    junk = [B.make_record("junk%d" % i, "ACGT" * 9 + "N" * 14, "I" * 50, 4) for i in range(5)]
    _, _, out = run(tmp_path, junk, ["--only-aligned"])
    assert out == []
    _, _, out = run(tmp_path, junk, [])
    assert len(out) == 5 and all(r["flag"] & 4 for r in out)


def test_three_hundred_thousand_reads_through_the_pipeline_in_order(tmp_path):
    """size-independent properties of the command at a size where every queue of its pipeline is in use (13 batches, two search
    threads, BGZF blocks in, BGZF blocks out): every record comes out exactly once, in input order, with its own bases; reads cut
    from the genome map where they were cut (unique places: mapping quality 37 there), and a second run gives the same bytes"""
    rng = np.random.default_rng(99)
    fa = "".join(l.strip() for l in open(os.path.join(T.GOLDEN, "toy.fa")) if not l.startswith(">")).upper()
    g = np.frombuffer(fa.encode(), np.uint8)
    code = np.full(256, 15, np.uint8); code[ord("A")] = 1; code[ord("C")] = 2; code[ord("G")] = 4; code[ord("T")] = 8
    g16 = code[g]
    N, L = 300_000, 64
    start = rng.integers(0, len(g16) - L, N)
    reads = g16[start[:, None] + np.arange(L)[None, :]]
    rec_len = 36 + 10 + L // 2 + L
    rec = np.zeros((N, rec_len), np.uint8)
    rec[:, 0:4] = np.frombuffer(struct.pack("<I", rec_len - 4), np.uint8)
    rec[:, 4:36] = np.frombuffer(struct.pack("<iiIIiiii", -1, -1, (4680 << 16) | 10, 4 << 16, L, -1, -1, 0), np.uint8)
    rec[:, 36] = ord("r")
    rec[:, 37:45] = np.frombuffer("".join(np.char.zfill(np.arange(N).astype("U8"), 8)).encode(), np.uint8).reshape(N, 8)
    rec[:, 46:46 + L // 2] = (reads[:, 0::2] << 4) | reads[:, 1::2]
    rec[:, 46 + L // 2:] = 30
    recs = [rec.tobytes()]                                        # write_bam joins the records: one blob will do
    env = {"NABWA_BAM_BATCH": "24000", "NABWA_DEVICES": "0,0"}
    _, _, out = run(tmp_path, recs, [], bgzf="blocks", env=env)
    first = open(str(tmp_path / "out.bam"), "rb").read()
    assert len(out) == N
    assert [o["name"] for o in out] == ["r%08d" % i for i in range(N)]
    fwd = np.array([not (o["flag"] & 16) for o in out])
    assert all(o["seq"] == "".join("=ACMGRSVTWYHKDBN"[c] for c in reads[i]) for i, o in enumerate(out[:2000]) if fwd[i])
    unique = [i for i, o in enumerate(out) if o["mapq"] == 37 and not (o["flag"] & 4)]
    assert len(unique) > 0.8 * N
    l_pac, names = toy_ann()
    offs = {}
    lines = open(T.TOY + ".ann").read().split("\n")
    for k in range(len(names)):
        offs[names[k]] = int(lines[2 + 2 * k].split()[0])
    wrong = [i for i in unique[:50000] if offs[out[i]["rname"]] + out[i]["pos"] - 1 != start[i]]
    assert not wrong, wrong[:5]
    _, _, again = run(tmp_path, recs, [], bgzf="blocks", env=env)
    second = open(str(tmp_path / "out.bam"), "rb").read()
    assert gzip.decompress(first).split(b"\n@SQ", 1)[1] == gzip.decompress(second).split(b"\n@SQ", 1)[1]


def reflag(rec, flag):
    """the same record with another FLAG"""
    return rec[:18] + struct.pack("<H", flag) + rec[20:]


def core(recs):
    return [(r["name"], r["flag"], r["rname"], r["pos"], r["mapq"], r["cigar"], r["rnext"], r["pnext"], r["tlen"], r["seq"], r["qual"], r["tags"], r["order"]) for r in recs]


def test_debug_bam_adds_the_search_statistic(tmp_path):
    """--debug-bam (bam2bam.c:433): YQ:i = bwt_match_gap's max_entries of the read, right after XC, left out when it is zero;
    the value is the oracle's (the reference does not write it anywhere else)"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:200]
    recs = [B.make_record(n, s, q, 4) for n, s, q in reads]
    _, _, plain = run(tmp_path, recs, [])
    _, _, dbg = run(tmp_path, recs, ["--debug-bam"])
    lib = T.load_oracle()
    oix = T.OracleIndex(lib)
    seq, rseq, off, _ = T.encode_reads(reads)
    _, maxe = T.oracle_cal_sa_reg_gap(lib, oix.h, T.default_opt(), seq, rseq, off, per_read=1, n_threads=8)
    assert (maxe > 0).any()
    for g, w, m in zip(dbg, plain, maxe):
        t = dict(g["tags"])
        assert t.pop("YQ", 0) == int(m), g["name"]
        assert t == w["tags"] and [k for k in g["order"] if k != "YQ:i"] == w["order"], g["name"]
        if m:
            assert g["order"][1 if "XC:i" in g["order"] else 0] == "YQ:i"


def test_skip_duplicates_passes_flagged_records_through(tmp_path):
    """--skip-duplicates (unique(), bam2bam.c:595-606): a logical record with a duplicate flag is not aligned, draws no random
    number, adds nothing to the insert sizes and comes out as it came in, less the tags erase_unwanted_tags removes; the other
    records come out as if the duplicates were not in the file"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:90]
    singles = [B.make_record(n, s, q, 4) for n, s, q in reads]
    pairs, n_pairs = pe_records()
    stale = B.tag_i("NM", 7) + B.tag_z("ZZ", "kept") + B.tag_a("XT", "U")
    dup_single = B.make_record("dup_single", reads[3][1], reads[3][2], 4 | 1024, stale)
    n0, s0, q0 = reads[5]
    dup_pair = [B.make_record("dup_pair", s0, q0, 1 | 4 | 8 | 64), B.make_record("dup_pair", reads[6][1], reads[6][2], 1 | 4 | 8 | 128 | 1024 | 512, stale)]
    rest = singles[:40] + pairs[:100] + singles[40:] + pairs[100:]
    mixed = singles[:40] + [dup_single] + pairs[:100] + dup_pair + singles[40:] + pairs[100:] + [dup_single]
    _, _, want = run(tmp_path, rest, [], env={"NABWA_BAM_BATCH": "64"})
    _, _, got = run(tmp_path, mixed, ["--skip-duplicates"], env={"NABWA_BAM_BATCH": "64"})
    dups = [r for r in got if r["name"].startswith("dup_")]
    assert core([r for r in got if not r["name"].startswith("dup_")]) == core(want)
    assert [r["name"] for r in got].index("dup_single") == 40 and got[-1]["name"] == "dup_single"
    assert [(r["name"], r["flag"], r["pos"], r["cigar"], r["tags"]) for r in dups] == [
        ("dup_single", 4 | 1024, 0, "*", {"ZZ": "kept"}),
        ("dup_pair", 1 | 4 | 8 | 64 | 512, 0, "*", {}),                       # the QC flag goes over both mates (bwaseqio.c:486-489)
        ("dup_pair", 1 | 4 | 8 | 128 | 1024 | 512, 0, "*", {"ZZ": "kept"}),
        ("dup_single", 4 | 1024, 0, "*", {"ZZ": "kept"})]
    assert dups[0]["seq"] == reads[3][1] and dups[0]["qual"] == reads[3][2]
    # without the option a duplicate is aligned like any other read
    _, _, aligned = run(tmp_path, mixed, [])
    a = [r for r in aligned if r["name"] == "dup_single"][0]
    assert (a["flag"], a["cigar"], a["tags"].get("ZZ")) == (want[3]["flag"] | 1024, want[3]["cigar"], "kept") and "NM" not in a["order"][:1]


def test_drop_aligned_leaves_out_records_that_are_mapped_already(tmp_path):
    """--drop-aligned (read_bam_pair, bwaseqio.c:466-474): a logical record any read of which lacks the unmapped flag is not
    read at all; the rest comes out as if it had not been in the file"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:60]
    singles = [B.make_record(n, s, q, 4) for n, s, q in reads]
    pairs, n_pairs = pe_records()
    mapped_single = B.make_record("mapped_single", reads[0][1], reads[0][2], 0)
    half = [B.make_record("half_mapped", reads[1][1], reads[1][2], 1 | 4 | 64), B.make_record("half_mapped", reads[2][1], reads[2][2], 1 | 8 | 128)]
    rest = singles[:30] + pairs[:60] + singles[30:]
    mixed = [mapped_single] + singles[:30] + half + pairs[:60] + [mapped_single] + singles[30:] + half
    _, _, want = run(tmp_path, rest, [], env={"NABWA_BAM_BATCH": "32"})
    _, _, got = run(tmp_path, mixed, ["--drop-aligned"], env={"NABWA_BAM_BATCH": "32"})
    assert core(got) == core(want)
    _, _, every = run(tmp_path, mixed, [])
    assert len(every) == len(mixed)


def test_broken_input_is_mended_the_way_the_reference_mends_it(tmp_path):
    """--broken-input (read_bam_pair_core's allow_broken, bwaseqio.c:345-410): two reads of one name with wrong read 1 / read 2 flags
    become read 1 and read 2 in file order; a paired read followed by another name is discarded, and so is one at the end of the
    file; without the option each of these ends the run with an error"""
    reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:40]
    singles = [B.make_record(n, s, q, 4) for n, s, q in reads]
    pairs, n_pairs = pe_records()
    pairs = [p for i in range(0, 80, 2) for p in (pairs[i:i + 2] if i % 14 != 4 else pairs[i:i + 2][::-1])]     # read 1 first everywhere
    both_first = [reflag(pairs[10], 1 | 4 | 8 | 64), reflag(pairs[11], 1 | 4 | 8 | 64)]
    unflagged = [reflag(pairs[12], 1 | 4 | 8), reflag(pairs[13], 4)]                    # the second is not even flagged as paired
    lone, lone_b = [B.make_record(nm, reads[0][1], reads[0][2], 1 | 4 | 8 | 64) for nm in ("lone_mate", "another_lone_mate")]
    clean = singles[:20] + pairs[:40] + singles[20:] + pairs[40:]
    broken = singles[:20] + pairs[:10] + both_first + unflagged + pairs[14:40] + [lone] + singles[20:] + [lone, lone_b] + pairs[40:] + [lone]
    _, _, want = run(tmp_path, clean, [], env={"NABWA_BAM_BATCH": "16"})
    _, _, got = run(tmp_path, broken, ["--broken-input"], env={"NABWA_BAM_BATCH": "16"})
    assert core(got) == core(want)
    inp = str(tmp_path / "strict.bam")
    for bad in (both_first, [lone] + singles[:1], singles[:1] + [lone]):
        write_bam(inp, bad, True)
        r = subprocess.run([EXE, "-g", T.TOY, "-f", str(tmp_path / "strict_out.bam"), inp], capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and ("lone mate" in r.stderr or "flags are wrong" in r.stderr or "without its mate" in r.stderr), r.stderr[-500:]


def test_unsupported_modes_are_refused(tmp_path):
    if not os.path.exists(EXE):
        nabwa.build()
    r = subprocess.run([EXE, "-g", T.TOY, "-p", "5000", "x.bam"], capture_output=True, text=True)
    assert r.returncode != 0 and "not provided" in r.stderr


def test_config1_ecoli_size_genome_10k_reads_against_the_reference_commands(tmp_path):
    """BASELINE config 1: an E. coli K-12-sized genome (4.6 Mbp, synthetic: there is no network), 10 k 100 bp single-end reads in a
    BAM file.  The reference's own commands (oracle/_ref/bwa_ref, compiled from /root/reference: `index`, `aln`, `samse` -- its CPU
    path) build the index and the expected records; nabwa_bam2bam loads that index and must write the same records."""
    refbin = os.path.join(T.ROOT, "oracle", "_ref", "bwa_ref")
    if not os.path.exists(refbin):
        pytest.skip("the compiled reference (oracle/_ref) did not travel")
    rng = np.random.default_rng(20261004)
    genome = rng.integers(0, 4, 4_641_652, dtype=np.uint8)                       # the length of E. coli K-12 MG1655
    for _ in range(40):                                                          # a few repeats (rRNA operons, IS elements)
        a, b, L = int(rng.integers(0, 4_600_000)), int(rng.integers(0, 4_600_000)), int(rng.integers(800, 5000))
        genome[b:b + L] = genome[a:a + L]
    text = "".join("ACGT"[c] for c in genome)
    fa = str(tmp_path / "eco.fa")
    with open(fa, "w") as f:
        f.write(">eco_syn\n")
        for o in range(0, len(text), 70):
            f.write(text[o:o + 70] + "\n")
    reads = []
    for i in range(10000):
        p = int(rng.integers(0, len(text) - 100))
        s = list(text[p:p + 100])
        for j in range(100):
            if rng.random() < 0.01:
                s[j] = "ACGT"[("ACGT".index(s[j]) + 1 + int(rng.integers(0, 3))) % 4]
        s = "".join(s)
        if i % 50 == 0:                                                          # some gapped reads
            s = s[:40] + s[42:] + "AC"
        if rng.random() < 0.5:
            s = revcomp(s)
        reads.append(("r%05d" % i, s, "".join(chr(33 + int(q)) for q in rng.integers(20, 41, 100))))
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as f:
        for n, sq, q in reads:
            f.write("@%s\n%s\n+\n%s\n" % (n, sq, q))
    def ref(args, out=None):
        r = subprocess.run([refbin] + args, stdout=open(out, "wb") if out else subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
    ref(["index", fa])
    sai, samf = str(tmp_path / "reads.sai"), str(tmp_path / "reads.sam")
    ref(["aln", fa, fq], sai)
    ref(["samse", fa, sai, fq], samf)
    sam = T.parse_sam(samf)
    recs = [B.make_record(n, sq, q, 4) for n, sq, q in reads]
    _, refs, out = run(tmp_path, recs, [], prefix=fa)
    assert refs == [("eco_syn", len(text))]
    assert len(out) == len(sam) == 10000
    check_se(out, sam)
    assert sum(1 for o in out if not (o["flag"] & 4)) > 9900


@pytest.mark.parametrize("switches", [[], ["--only-aligned"], ["--skip-duplicates", "--debug-bam"]])
def test_temp_dir_keeps_what_waits_for_pass_2_in_a_file(tmp_path, switches):
    """--temp-dir (bam2bam.c:1733-1758): batches that have to wait for the insert-size estimates leave memory as the reference's temporary
    file holds them (u32 length + the positioned message of msg_init_from_pair; a batch of single reads that only waits for its turn as
    finished records) and come back for pass 2.  Same input, small batches, pairs of two read groups between single reads: the output is
    the output of the run that kept everything in memory, byte for byte."""
    se = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))[:150]
    pe = [T.read_fastq(os.path.join(T.GOLDEN, "reads_pe_%d.fq" % e)) for e in (1, 2)]
    recs = []
    for i in range(260):
        if i % 2 == 0 and i // 2 < len(se):
            n, s, q = se[i // 2]
            recs.append(B.make_record(n, s, q, 4 | (0x400 if i % 14 == 0 else 0)))
        n, s1, q1 = pe[0][i]
        _, s2, q2 = pe[1][i]
        rg = B.tag_z("RG", "libA" if i % 3 else "libB")
        recs.append(B.make_record(n, s1, q1, 1 | 64 | 4 | 8, rg))
        recs.append(B.make_record(n, s2, q2, 1 | 128 | 4 | 8, rg))
    a = tmp_path / "a"; b = tmp_path / "b"; tdir = tmp_path / "scratch"
    for d in (a, b, tdir):
        d.mkdir()
    env = {"NABWA_BAM_BATCH": "64"}
    text_a, refs_a, out_a = run(a, recs, switches, env=env)
    text_b, refs_b, out_b = run(b, recs, switches + ["--temp-dir", str(tdir)], env=env)
    assert refs_a == refs_b and len(out_a) == len(out_b) > 0
    assert out_a == out_b
    assert os.listdir(str(tdir)) == []                               # the file is gone with the process (unlinked when it was made)
    raw_a = gzip.decompress(open(str(a / "out.bam"), "rb").read())
    raw_b = gzip.decompress(open(str(b / "out.bam"), "rb").read())
    assert raw_a[raw_a.index(b"\n@SQ"):] == raw_b[raw_b.index(b"\n@SQ"):]      # everything after the @PG line (it holds the command line)
    if not switches:
        assert len(out_a) == len(recs)
