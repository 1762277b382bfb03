"""The host side of `nabwa_aln` without a GPU (NABWA_ALN_PARSE_ONLY): FASTA/FASTQ parsing as kseq_read does it
(kseq.h:155-193), the filters, barcode removal, quality trimming and encoding of bwa_read_seq (bwaseqio.c:172-252),
checked read by read against an independent restatement written here from those descriptions.  (That the tool's .sai is
byte-identical to the reference's for the same files is the GPU test, tests/test_gpu_aln_cli.py.)"""
import gzip
import os
import subprocess

import numpy as np
import pytest

import nabwa_testlib as T
import test_gpu_aln_cli as CLI          # the awkward inputs are built by the same helpers

TOOL = CLI.TOOL
FNV0, FNVP, M64 = 1469598103934665603, 1099511628211, (1 << 64) - 1


def fnv(h, data):
    for b in data:
        h = ((h ^ b) * FNVP) & M64
    return h


def kseq_records(data):
    """(name, comment, seq, qual) records of a byte string, the way kseq_read cuts them; stops at a truncated quality"""
    i, n, pending, out = 0, len(data), False, []

    def getc():
        nonlocal i
        if i >= n:
            return -1
        i += 1
        return data[i - 1]

    while True:
        if not pending:
            c = getc()
            while c != -1 and c not in (62, 64):           # '>' '@'
                c = getc()
            if c == -1:
                return out
        pending = False
        name, comment, seq, qual = bytearray(), bytearray(), bytearray(), bytearray()
        c = getc()
        while c != -1 and not chr(c).isspace():
            name.append(c)
            c = getc()
        if c == -1 and not name:
            return out
        if c != 10 and c != -1:
            c = getc()
            while c != -1 and c != 10:
                comment.append(c)
                c = getc()
        c = getc()
        while c != -1 and c not in (62, 43, 64):            # '>' '+' '@'
            if 33 <= c <= 126:
                seq.append(c)
            c = getc()
        if c in (62, 64):
            pending = True
        if c != 43:
            out.append((bytes(name), bytes(comment), bytes(seq), b""))
            continue
        c = getc()
        while c != -1 and c != 10:
            c = getc()
        if c == -1:
            return out
        c = getc()
        while c != -1 and len(qual) < len(seq):
            if 33 <= c <= 127:
                qual.append(c)
            c = getc()
        if len(qual) != len(seq):
            return out
        out.append((bytes(name), bytes(comment), bytes(seq), bytes(qual)))


def expected_lines(data, trim_qual=0, barcode=0, casava=False, il13=False, comp=True):
    code = {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}
    lines, n_bases, total = [], 0, FNV0
    for name, comment, seq, qual in kseq_records(data):
        if casava and comment:
            p = comment.find(b":")
            if p >= 0 and p + 1 < len(comment) and comment[p + 1:p + 2] == b"Y":
                continue
        if len(seq) <= barcode:
            continue
        seq, qual = seq[barcode:], qual[barcode:]
        ln = len(seq)
        if qual and trim_qual >= 1:
            shift = 33 + (31 if il13 else 0)
            s, best, best_l = 0, 0, ln - 1
            for l in range(ln - 1, 33, -1):
                s += trim_qual - (qual[l] - shift)
                if s < 0:
                    break
                if s > best:
                    best, best_l = s, l
            ln = best_l + 1
        fwd = [code.get(c, 4) for c in seq[:ln]]
        rev = bytes(reversed(fwd))
        rc = bytes((3 - c if (comp and c < 4) else c) for c in rev)
        h = fnv(fnv(FNV0, rev), rc)
        lines.append("%d %016x" % (ln, h))
        total = fnv(total, h.to_bytes(8, "little"))
        n_bases += ln
    lines.append("reads %d bases %d fnv %016x" % (len(lines), n_bases, total))
    return lines


def tool_lines(args, path, buf=None, window=None, threads=None):
    env = dict(os.environ, NABWA_ALN_PARSE_ONLY="2")
    if buf:
        env["NABWA_ALN_BUF"] = str(buf)
    if window:
        env["NABWA_ALN_WINDOW"] = str(window)
    if threads:
        env["NABWA_ALN_THREADS"] = str(threads)
    r = subprocess.run([TOOL] + args + ["unused_prefix", path], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    return r.stdout.split("\n")[:-1]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    if not os.path.exists(TOOL):
        import importlib
        importlib.import_module("network-aware-bwa_amd").build()
    d = tmp_path_factory.mktemp("fx")
    rng = np.random.default_rng(77)
    recs = CLI.awkward_reads(rng)
    out = {}
    out["fq"] = str(d / "a.fq")
    CLI.write_fastq(out["fq"], recs, np.random.default_rng(1))
    out["fq64"] = str(d / "a64.fq")
    CLI.write_fastq(out["fq64"], recs, np.random.default_rng(1), base=64)
    out["gz"] = str(d / "a.fq.gz")
    CLI.write_fastq(out["gz"], recs, np.random.default_rng(1), opener=gzip.open)
    out["fa"] = str(d / "a.fa")
    with open(out["fa"], "w") as f:               # multi-line FASTA, a blank record, trailing blanks, no final newline
        for i, (n, cm, s) in enumerate(recs):
            f.write(">%s\t%s\n" % (n, cm))
            if i == 11:
                f.write("\n")
                continue
            for j in range(0, len(s), 25):
                f.write(s[j:j + 25] + ("\n" if i % 2 or j + 25 < len(s) else " \n"))
        f.write(">last\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT")
    out["odd"] = str(d / "odd.fq")
    with open(out["odd"], "wb") as f:             # what a strict parser would reject and kseq takes in its stride
        f.write(b"junk before the first record\n@r1 c1\nACGT\nACGT\n+r1\nIIII\nIIII\n"          # two-line sequence and quality
                b"@r2\nAC@GT\n+\nIIII\n"                                                       # '@' inside a sequence line ends it
                b"\n\n>f1\nacgtnACGT\n>f2 x y\n\n>f3\nAC-GT.AC\n"                               # FASTA records in a FASTQ file, an empty one
                b"@r3\nACGTACGTAC\n+\nIIIIIIIIII@r4\nACGTA\n+\nIIIII\n"                         # no newline after a quality: its '@' is eaten
                b"@r5\nACGTACGT\n+\nIII")                                                       # truncated quality: reading stops here
    return out


@pytest.mark.parametrize("args,key,kw", [
    ([], "fq", {}), ([], "gz", {}), ([], "fa", {}), ([], "odd", {}),
    (["-Y"], "fq", {"casava": True}), (["-B", "6"], "fq", {"barcode": 6}),
    (["-B", "6", "-Y", "-q", "15"], "fq", {"barcode": 6, "casava": True, "trim_qual": 15}),
    (["-I", "-q", "20"], "fq64", {"il13": True, "trim_qual": 20}), (["-q", "25"], "fq", {"trim_qual": 25}),
    (["-c"], "fq", {"comp": False}), (["-Y", "-q", "30"], "fa", {"casava": True, "trim_qual": 30}),
], ids=lambda x: "_".join(x) if isinstance(x, list) else None)
def test_parsing_filters_trimming_and_encoding(files, args, key, kw):
    path = files[key]
    raw = gzip.open(path, "rb").read() if key == "gz" else open(path, "rb").read()
    assert tool_lines(args, path) == expected_lines(raw, **kw)


@pytest.mark.parametrize("buf", [1, 2, 7, 64, 1000])
def test_every_scan_survives_the_end_of_the_buffer(files, buf):
    """the parser works on whatever the read buffer holds and appends whole runs: with buffers of a few bytes every run,
    header, line end and quality string is cut somewhere"""
    for key, args, kw in (("odd", [], {}), ("fa", ["-Y"], {"casava": True}), ("fq", ["-B", "6", "-q", "15"], {"barcode": 6, "trim_qual": 15}),
                          ("gz", [], {})):
        path = files[key]
        raw = gzip.open(path, "rb").read() if key == "gz" else open(path, "rb").read()
        assert tool_lines(args, path, buf) == expected_lines(raw, **kw), (key, buf)


@pytest.mark.parametrize("window,threads", [(200, 2), (333, 3), (1000, 8), (5000, 5), (1 << 20, 1), (1 << 20, 7)])
def test_pieces_of_a_mapped_file_give_the_sequential_parse(files, window, threads):
    """a plain file is mapped and parsed in pieces that start at GUESSED record starts (aln_main.cpp: read_everything);
    a piece counts only if the parse before it stopped exactly there.  Tiny windows put guesses everywhere: into
    multi-line records, onto quality lines that begin with '@', into FASTA records inside a FASTQ file, behind the
    truncated record that ends the input."""
    for key, args, kw in (("odd", [], {}), ("fa", [], {}), ("fa", ["-Y", "-q", "30"], {"casava": True, "trim_qual": 30}),
                          ("fq", [], {}), ("fq", ["-B", "6", "-Y", "-q", "15"], {"barcode": 6, "casava": True, "trim_qual": 15})):
        path = files[key]
        raw = open(path, "rb").read()
        assert tool_lines(args, path, window=window, threads=threads) == expected_lines(raw, **kw), (key, window, threads)


def test_quality_lines_that_look_like_headers(tmp_path):
    """four-line FASTQ whose quality strings begin with '@' and '+' and '>' -- the shapes the boundary guess has to survive"""
    rng = np.random.default_rng(3)
    path = str(tmp_path / "tricky.fq")
    with open(path, "wb") as f:
        for i in range(3000):
            L = int(rng.integers(30, 90))
            s = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), L, p=[.24, .24, .24, .24, .04]))
            q = bytearray(int(x) for x in rng.integers(35, 74, L))
            q[0] = b"@+>I"[i % 4]
            if i % 5 == 0:
                q[1] = 64                                  # "@@..."
            f.write(b"@t%d %d:N\n" % (i, i % 3) + s + b"\n+" + (b"t%d" % i if i % 2 else b"") + b"\n" + bytes(q) + b"\n")
    raw = open(path, "rb").read()
    want = expected_lines(raw, trim_qual=20)
    assert want[-1].startswith("reads 3000 ")
    for window, threads in ((400, 4), (4096, 8), (1 << 16, 3), (None, None)):
        assert tool_lines(["-q", "20"], path, window=window, threads=threads) == want, (window, threads)


def expected_bam_lines(recs, which=7, trim_qual=0):
    """bwa_read_bam (bwaseqio.c:125-168) restated on the records a BAM was written from"""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    lines, n_bases, total = [], 0, FNV0
    for name, flag, seq, qual in recs:
        paired = flag & 1
        if not ((which & 1 and paired and flag & 64) or (which & 2 and paired and flag & 128) or (which & 4 and not paired)):
            continue
        c = [code.get(ch, 4) for ch in seq]
        q = [min(x + 33, 126) for x in qual]
        if flag & 16:
            c = [3 - x if x < 4 else x for x in reversed(c)]
            q = list(reversed(q))
        ln = len(c)
        if trim_qual >= 1:
            s, best, best_l = 0, 0, ln - 1
            for l in range(ln - 1, 33, -1):
                s += trim_qual - (q[l] - 33)
                if s < 0:
                    break
                if s > best:
                    best, best_l = s, l
            ln = best_l + 1
        ln = max(ln, 0)
        rev = bytes(reversed(c[:ln]))
        rc = bytes(3 - x if x < 4 else x for x in rev)
        h = fnv(fnv(FNV0, rev), rc)
        lines.append("%d %016x" % (ln, h))
        total = fnv(total, h.to_bytes(8, "little"))
        n_bases += ln
    lines.append("reads %d bases %d fnv %016x" % (len(lines), n_bases, total))
    return lines


@pytest.mark.parametrize("args,which,trim", [(["-b"], 7, 0), (["-b", "-0"], 4, 0), (["-b", "-1", "-q", "15"], 1, 15), (["-b", "-2"], 2, 0),
                                             (["-b", "-1", "-2", "-q", "30"], 3, 30)], ids=lambda x: "_".join(x) if isinstance(x, list) else None)
def test_bam_records_selection_strand_and_trimming(tmp_path, args, which, trim):
    recs = CLI.bam_records(np.random.default_rng(8))
    for members in (1, 5):
        path = str(tmp_path / ("r%d.bam" % members))
        CLI.write_bam(path, recs, members)
        assert tool_lines(args, path) == expected_bam_lines(recs, which, trim), members


@pytest.mark.skipif(not os.path.exists(CLI.REFBIN), reason="compiled reference not built")
def test_the_bam_restatement_against_the_compiled_reference(tmp_path):
    """pins expected_bam_lines' reading of bwa_read_bam to the reference itself, on the CPU: the reference's .sai for a BAM equals
    its .sai for the FASTQ that holds the reads as the restatement says the BAM path sees them (turned back, qualities as
    capped ASCII) -- with and without trimming.  (Empty reads left out: the FASTQ path drops them, the BAM path keeps them.)"""
    recs = [r for r in CLI.bam_records(np.random.default_rng(8), 300) if len(r[2])]
    bam, fq = str(tmp_path / "r.bam"), str(tmp_path / "r.fq")
    CLI.write_bam(bam, recs, 3)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    with open(fq, "w") as f:
        for name, flag, seq, qual in recs:
            q = [min(x + 33, 126) for x in qual]
            if flag & 16:
                seq, q = "".join(comp[c] for c in reversed(seq)), list(reversed(q))
            f.write("@%s\n%s\n+\n%s\n" % (name, seq, "".join(chr(x) for x in q)))
    for extra in ([], ["-q", "15"], ["-q", "30", "-n", "2"]):
        a = CLI.run_ref(["-b"] + extra + [T.TOY, bam])
        b = CLI.run_ref(extra + [T.TOY, fq])
        assert len(a) > 64 + 4 * len(recs) and a[64:] == b[64:], extra


def test_reads_from_standard_input(files, tmp_path):
    """'-' as the file name: FASTQ (plain and gzip) and BAM arrive through a pipe"""
    env = dict(os.environ, NABWA_ALN_PARSE_ONLY="2")
    for key in ("fq", "gz"):
        raw = gzip.open(files[key], "rb").read() if key == "gz" else open(files[key], "rb").read()
        with open(files[key], "rb") as f:
            r = subprocess.run([TOOL, "-q", "15", "unused_prefix", "-"], stdin=f, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert r.stdout.split("\n")[:-1] == expected_lines(raw, trim_qual=15), key
    recs = CLI.bam_records(np.random.default_rng(8), 120)
    bam = str(tmp_path / "p.bam")
    CLI.write_bam(bam, recs, 2)
    with open(bam, "rb") as f:
        r = subprocess.run([TOOL, "-b", "-1", "unused_prefix", "-"], stdin=f, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n")[:-1] == expected_bam_lines(recs, 1, 0)


def test_golden_reads_parse_like_the_test_library():
    """and the same against tests/nabwa_testlib.py's reader, which the parity tests feed the GPU from"""
    fq = os.path.join(T.GOLDEN, "reads_se.fq")
    reads = T.read_fastq(fq)
    seq, rseq, off, _ = T.encode_reads(reads, trim_qual=20)
    want = []
    for i in range(len(reads)):
        a, b = seq[off[i]:off[i + 1]].tobytes(), rseq[off[i]:off[i + 1]].tobytes()
        want.append("%d %016x" % (len(a), fnv(fnv(FNV0, a), b)))
    assert tool_lines(["-q", "20"], fq)[:-1] == want
