"""`nabwa_aln` -- the reference's `bwa aln` command line on the GPU library (bwtaln.c:178-395).
The .sai it writes must be byte for byte the file the reference writes for the same arguments:
 * against the committed goldens (the option sets of make_golden.py -- the two slowest, `adna` and `loggap`, only through the
   library in test_gpu_parity.py: the command line adds nothing to them but a minute --, -N, both paired-end files);
 * against the compiled reference run on the spot (oracle/_ref/bwa_ref, when it travelled) for inputs the goldens do not
   hold: multi-line FASTA, gzip, barcodes, the Casava filter, Illumina-1.3 qualities with trimming, and a file long
   enough to cross the reference's 0x40000-read chunks with a different max_gapo clamp on either side."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import nabwa_testlib as T

pytestmark = pytest.mark.gpu

TOOL = os.path.join(T.ROOT, "network-aware-bwa_amd", "nabwa_aln")
REFBIN = os.path.join(T.ROOT, "oracle", "_ref", "bwa_ref")

GOLDEN_RUNS = {
    "se_default": ([], "reads_se.fq"),
    "se_n3": (["-n", "3"], "reads_se.fq"),
    "se_e3": (["-e", "3", "-o", "2"], "reads_se.fq"),
    "se_k1R5": (["-k", "1", "-R", "5", "-l", "25"], "reads_se.fq"),
    "se_i2": (["-i", "2", "-M", "2", "-O", "7", "-E", "3"], "reads_se.fq"),
    "se_q20": (["-q", "20"], "reads_se.fq"),
    "se_m64": (["-m", "64"], "reads_se.fq"),
    "se_nonstop": (["-N"], "reads_se_head.fq"),
    "pe_1": ([], "reads_pe_1.fq"),
    "pe_2": ([], "reads_pe_2.fq"),
    "pe150_1": ([], "reads_pe150_1.fq"),          # 2 x 150 bp pairs with BASELINE config 3's error load (make_golden.py pe_chain150)
    "pe150_2": ([], "reads_pe150_2.fq"),
}


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(TOOL):          # the tool is built with the library (csrc/Makefile: all)
        import importlib
        importlib.import_module("network-aware-bwa_amd").build()
    assert os.path.exists(TOOL)


def run_tool(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([TOOL] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
    return r.returncode, r.stdout, r.stderr.decode(errors="replace")


def run_ref(args):
    r = subprocess.run([REFBIN, "aln"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
    assert r.returncode == 0
    return r.stdout


@pytest.mark.parametrize("name", sorted(GOLDEN_RUNS))
def test_sai_bytes_equal_the_reference_goldens(name):
    args, fq = GOLDEN_RUNS[name]
    rc, out, err = run_tool(args + [T.TOY, os.path.join(T.GOLDEN, fq)])
    assert rc == 0, err
    with open(os.path.join(T.GOLDEN, name + ".sai"), "rb") as f:
        want = f.read()
    assert out == want


def awkward_reads(rng):
    """records that exercise the parser and the filters: FASTQ with comments, lower case, N, '-', '.'"""
    genome = "".join(s for _, s in T.read_fasta(T.TOY + ".fa"))
    recs = []
    for i in range(400):
        L = int(rng.integers(36, 120))
        p = int(rng.integers(0, len(genome) - L))
        s = list(genome[p:p + L])
        for j in range(L):
            if rng.random() < 0.01:
                s[j] = "ACGT"[int(rng.integers(4))]
        if i % 7 == 0:
            s[int(rng.integers(L))] = "N"
        if i % 31 == 0:
            s[int(rng.integers(L))] = "-"
        if i % 37 == 0:
            s[int(rng.integers(L))] = "."
        s = "".join(s)
        if i % 3 == 0:
            s = s.lower()
        recs.append(("q%03d" % i, "1:%s:0:ACGT" % ("Y" if i % 5 == 0 else "N"), s))
    return recs


def write_fastq(path, recs, rng, base=33, opener=open):
    with opener(path, "wt") as f:
        for n, cm, s in recs:
            q = [int(x) for x in rng.integers(2, 41, len(s))]
            k = int(rng.integers(0, len(s) // 2))
            for j in range(len(s) - k, len(s)):
                q[j] = int(rng.integers(2, 12))
            f.write("@%s %s\n%s\n+\n%s\n" % (n, cm, s, "".join(chr(base + x) for x in q)))


@pytest.mark.skipif(not os.path.exists(REFBIN), reason="compiled reference did not travel")
def test_awkward_inputs_equal_the_compiled_reference(tmp_path):
    rng = np.random.default_rng(77)
    recs = awkward_reads(rng)
    fq = str(tmp_path / "a.fq")
    write_fastq(fq, recs, np.random.default_rng(1))
    fq64 = str(tmp_path / "a64.fq")
    write_fastq(fq64, recs, np.random.default_rng(1), base=64)
    fqz = str(tmp_path / "a.fq.gz")
    write_fastq(fqz, recs, np.random.default_rng(1), opener=gzip.open)
    fa = str(tmp_path / "a.fa")
    with open(fa, "w") as f:                      # multi-line FASTA, a blank line, a record without sequence, no final newline
        for i, (n, cm, s) in enumerate(recs):
            f.write(">%s\t%s\n" % (n, cm))
            if i == 11:
                f.write("\n")
                continue
            for j in range(0, len(s), 25):
                f.write(s[j:j + 25] + ("\n" if i % 2 or j + 25 < len(s) else " \n"))
        f.write(">last\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT")
    cases = [
        ([], fq), ([], fqz), ([], fa),
        (["-Y"], fq), (["-B", "6"], fq), (["-B", "6", "-Y", "-q", "15"], fq),
        (["-I", "-q", "20"], fq64), (["-q", "25", "-n", "0.02"], fq),
        (["-c"], fq), (["-t", "4", "-o", "0"], fq), (["-n", "2", "-o", "3", "-e", "2"], fq),
    ]
    for args, path in cases:
        want = run_ref(args + [T.TOY, path])
        rc, got, err = run_tool(args + [T.TOY, path])
        assert rc == 0, err
        assert got == want, (args, os.path.basename(path))


@pytest.mark.skipif(not os.path.exists(REFBIN), reason="compiled reference did not travel")
def test_chunks_with_different_gap_open_clamps(tmp_path):
    """bwtaln.c:104-105: max_gapo is clamped by the max_diff of the longest read of each 0x40000-read call.  First
    chunk: 24-base reads only (max_diff 2 < -o 3); afterwards 100-base reads appear (max_diff 5, clamp stays 3).  The
    tool reads everything as one batch and has to split it at the chunk boundary."""
    rng = np.random.default_rng(5)
    genome = "".join(s for _, s in T.read_fasta(T.TOY + ".fa")).replace("N", "A")
    n_short, n_long = 0x40000, 3000
    starts = rng.integers(0, len(genome) - 120, n_short + n_long)
    fa = str(tmp_path / "two_chunks.fa")
    with open(fa, "w") as f:
        for i, p in enumerate(starts):
            L = 24 if i < n_short else 100
            s = genome[p:p + L]
            if i % 3 == 0:                        # a 1-base deletion, so that gap opens matter
                s = s[:L // 2] + s[L // 2 + 1:]
            f.write(">r%d\n%s\n" % (i, s))
    args = ["-o", "3", "-e", "2", "-i", "3", fa]
    want = run_ref(args[:-1] + [T.TOY, fa])
    rc, got, err = run_tool(args[:-1] + [T.TOY, fa])
    assert rc == 0, err
    assert got == want
    # and the same with GPU batches no larger than one reference chunk
    rc, got, err = run_tool(args[:-1] + [T.TOY, fa], env={"NABWA_ALN_BATCH": str(0x40000)})
    assert rc == 0, err
    assert got == want
    # and with two index replicas (NABWA_DEVICES: one worker per entry; here both on GPU 0) taking the batches as they come:
    # the records still leave in input order
    rc, got, err = run_tool(args[:-1] + [T.TOY, fa], env={"NABWA_ALN_BATCH": str(0x40000), "NABWA_DEVICES": "0,0"})
    assert rc == 0, err
    assert got == want


def write_bam(path, recs, members=1):
    """an unaligned BAM (one @RG-less header, no references) of (name, flag, sequence, phred list) records, as a gzip stream of
    `members` members (BGZF is a series of gzip members; the reference reads BAM through gzopen, bamlite.h:7-11)"""
    import struct
    nt16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    text = b"@HD\tVN:1.0\tSO:unsorted\n"
    body = [b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chrT\x00" + struct.pack("<i", 1000)]
    for name, flag, seq, qual in recs:
        qn = name.encode() + b"\x00"
        packed = bytearray((len(seq) + 1) // 2)
        for i, c in enumerate(seq):
            packed[i >> 1] |= nt16[c] << (4 if i % 2 == 0 else 0)
        core = struct.pack("<iiIIiiii", -1, -1, (4680 << 16) | (0 << 8) | len(qn), (flag << 16) | 0, len(seq), -1, -1, 0)
        data = qn + bytes(packed) + bytes(qual) + b"XYZ\x00" * (len(name) % 2)       # sometimes a tag-like tail
        body.append(struct.pack("<i", len(core) + len(data)) + core + data)
    blob = b"".join(body)
    with open(path, "wb") as f:
        step = (len(blob) + members - 1) // members
        for i in range(0, len(blob), step):
            f.write(gzip.compress(blob[i:i + step]))


def bam_records(rng, n=500):
    genome = "".join(s for _, s in T.read_fasta(T.TOY + ".fa")).replace("N", "A")
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    recs = []
    for i in range(n):
        L = int(rng.integers(0, 3)) * 0 + int(rng.choice([0, 36, 50, 76, 100, 101]))
        p = int(rng.integers(0, len(genome) - 120))
        s = list(genome[p:p + L])
        for j in range(L):
            if rng.random() < 0.01:
                s[j] = "ACGTN"[int(rng.integers(5))]
        flag = [4, 4 | 16, 1 | 64 | 4, 1 | 128 | 4, 1 | 64 | 16, 1 | 128][i % 6]
        if flag & 16:
            s = [comp[c] for c in reversed(s)]                       # stored the way an aligner stores a reverse-strand read
        q = [int(x) for x in rng.integers(2, 41, L)]
        for j in range(L - int(rng.integers(0, L // 2 + 1)), L):
            q[j] = int(rng.integers(2, 12))
        if i % 50 == 0:
            q = [255] * L                                            # qualities absent
        recs.append(("b%04d" % i, flag, "".join(s), q))
    return recs


@pytest.mark.skipif(not os.path.exists(REFBIN), reason="compiled reference did not travel")
def test_bam_input_equals_the_compiled_reference(tmp_path):
    """-b with the read selections -0 -1 -2 and trimming: reverse-strand records are turned back, empty reads stay in"""
    recs = bam_records(np.random.default_rng(8))
    one, many = str(tmp_path / "r.bam"), str(tmp_path / "r_members.bam")
    write_bam(one, recs)
    write_bam(many, recs, members=7)
    for args, path in ((["-b"], one), (["-b"], many), (["-b", "-0"], one), (["-b", "-1", "-q", "15"], many), (["-b", "-2"], one),
                       (["-b", "-1", "-2"], one), (["-b", "-0", "-q", "25", "-n", "3"], one)):
        want = run_ref(args + [T.TOY, path])
        rc, got, err = run_tool(args + [T.TOY, path])
        assert rc == 0, err
        assert got == want, args


def test_resume_into_an_interrupted_file_and_final_rename(tmp_path):
    fq = os.path.join(T.GOLDEN, "reads_se.fq")
    with open(os.path.join(T.GOLDEN, "se_k1R5.sai"), "rb") as f:
        want = f.read()
    part = str(tmp_path / "out.sai_")
    with open(part, "wb") as f:
        f.write(want[:len(want) // 2 + 7])          # an interrupted run: ends in the middle of a record
    # options come back from the file's header; the finished file loses its trailing underscore (utils.c:159-173)
    rc, out, err = run_tool(["-f", part, T.TOY, fq])
    assert rc == 0, err
    assert out == b"" and "attempting recovery" in err
    assert not os.path.exists(part)
    with open(str(tmp_path / "out.sai"), "rb") as f:
        assert f.read() == want
    # a fresh -f file without underscore stays where it is
    plain = str(tmp_path / "fresh.sai")
    rc, out, err = run_tool(["-k", "1", "-R", "5", "-l", "25", "-f", plain, T.TOY, fq])
    assert rc == 0, err
    with open(plain, "rb") as f:
        assert f.read() == want


def test_refusals(tmp_path):
    fq = os.path.join(T.GOLDEN, "reads_se.fq")
    rc, out, err = run_tool(["-b", T.TOY, fq])                 # -b on something that is not BAM
    assert rc == 2 and out == b"" and "not a BAM file" in err
    rc, out, err = run_tool([T.TOY])
    assert rc == 1 and "Usage" in err
    rc, out, err = run_tool([str(tmp_path / "no_such_index"), fq])
    assert rc == 2 and out == b"" and "cannot set up the index" in err
