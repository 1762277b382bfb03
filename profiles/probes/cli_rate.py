"""File to file rate of nabwa_bam2bam: N single-end 100 bp reads drawn from the toy genome (tests/golden/toy.fa) in a true BGZF
BAM file -> aligned BGZF BAM file.  Prints the wall time of the command and its own stage timing (NABWA_TIMING)."""
import os, struct, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
PAIRED = os.environ.get("CLI_RATE_PAIRED") == "1"        # reads 2k and 2k+1 are mates: same name, 300 bp apart, second one reverse-complemented
OUT = sys.argv[2] if len(sys.argv) > 2 else "/tmp/cli_rate"
os.makedirs(OUT, exist_ok=True)


def bgzf_block(chunk):
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    d = c.compress(chunk) + c.flush()
    return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", len(d) + 25) + d
            + struct.pack("<II", zlib.crc32(chunk), len(chunk)))


def write_bgzf(path, raw):
    view = memoryview(raw)
    chunks = [view[o:o + 0xff00] for o in range(0, len(raw), 0xff00)]
    with ThreadPoolExecutor(16) as ex, open(path, "wb") as f:
        for b in ex.map(lambda c: bgzf_block(bytes(c)), chunks, chunksize=64):
            f.write(b)
        f.write(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))


def main():
    fa = "".join(l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "toy.fa")) if not l.startswith(">"))
    g = np.frombuffer(fa.upper().encode(), np.uint8)
    code = np.full(256, 15, np.uint8); code[ord("A")] = 1; code[ord("C")] = 2; code[ord("G")] = 4; code[ord("T")] = 8
    g16 = code[g]
    rng = np.random.default_rng(7)
    L = 100
    start = rng.integers(0, len(g16) - L - 400, N)
    if PAIRED:
        start[1::2] = start[0::2] + 300 + rng.integers(-30, 30, N // 2)
    reads = g16[start[:, None] + np.arange(L)[None, :]]
    if PAIRED:
        comp = np.zeros(16, np.uint8); comp[[1, 2, 4, 8, 15]] = [8, 4, 2, 1, 15]
        reads[1::2] = comp[reads[1::2, ::-1]]
    sub = rng.random((N, L)) < 0.01
    reads[sub] = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, int(sub.sum()))]
    rec_len = 36 + 10 + L // 2 + L
    rec = np.zeros((N, rec_len), np.uint8)
    rec[:, 0:4] = np.frombuffer(struct.pack("<I", rec_len - 4), np.uint8)
    rec[:, 4:36] = np.frombuffer(struct.pack("<iiIIiiii", -1, -1, (4680 << 16) | 10, 4 << 16, L, -1, -1, 0), np.uint8)
    rec[:, 36] = ord("r")
    ids = np.arange(N) // 2 * 2 if PAIRED else np.arange(N)
    rec[:, 37:45] = np.frombuffer("".join(np.char.zfill(ids.astype("U8"), 8)).encode(), np.uint8).reshape(N, 8)
    if PAIRED:
        rec[0::2, 18] = 1 | 4 | 8 | 64; rec[1::2, 18] = 1 | 4 | 8 | 128
    rec[:, 46:46 + L // 2] = (reads[:, 0::2] << 4) | reads[:, 1::2]
    rec[:, 46 + L // 2:] = 40
    text = b"@HD\tVN:1.0\n"
    raw = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0) + rec.tobytes()
    inp, outp = os.path.join(OUT, "in.bam"), os.path.join(OUT, "out.bam")
    t0 = time.time()
    write_bgzf(inp, raw)
    print("input: %d reads, %.1f MB plain, %.1f MB BGZF (written in %.1f s)" % (N, len(raw) / 1e6, os.path.getsize(inp) / 1e6, time.time() - t0), flush=True)
    exe = os.path.join(ROOT, "network-aware-bwa_amd", "nabwa_bam2bam")
    env = dict(os.environ, NABWA_TIMING="1")
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([exe, "-g", os.path.join(ROOT, "tests", "golden", "toy"), "-f", outp] + sys.argv[3:] + [inp], capture_output=True, text=True, env=env)
        dt = time.time() - t0
        print("run %d: rc %d, %.2f s wall, %.2f M reads/s file to file, output %.1f MB" % (rep, r.returncode, dt, N / dt / 1e6, os.path.getsize(outp) / 1e6), flush=True)
        lines = [l for l in r.stderr.split("\n") if "[nabwa_bam2bam] timing" in l or "index_load" in l]
        print("\n".join(lines[-12:]))
        if rep == 1:
            print("\n".join(l[:260] for l in r.stderr.split("\n") if l.startswith("[nabwa] ") and "kernel D" not in l)[-4200:])
        if r.returncode:
            print(r.stderr[-2000:])
    # the records: every read back, in order
    import gzip
    body = gzip.decompress(open(outp, "rb").read())
    l_text = struct.unpack_from("<i", body, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", body, p)[0]; p += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", body, p)[0]; p += 4 + ln + 4
    cnt = 0; mapped = 0
    while p < len(body):
        bs, = struct.unpack_from("<I", body, p)
        if cnt % 100000 == 0:
            assert body[p + 36:p + 45] == b"r%08d" % (cnt // 2 * 2 if PAIRED else cnt), (cnt, body[p + 36:p + 46])
        mapped += not (struct.unpack_from("<I", body, p + 16)[0] >> 16 & 4)
        p += 4 + bs; cnt += 1
    print("output records: %d (%d mapped)" % (cnt, mapped))
    assert cnt == N


main()
