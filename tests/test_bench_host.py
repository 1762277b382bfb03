"""Host-side pieces of bench.py that need no GPU: the ancient-DNA read profile that `--adna` lays over the
synthetic reads, and the command line itself."""
import importlib.util
import os
import subprocess
import sys

import numpy as np

import nabwa_testlib as T


def load_bench():
    spec = importlib.util.spec_from_file_location("nabwa_bench", os.path.join(T.ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_adna_profile_keeps_the_encoding_and_damages_only_the_ends():
    b = load_bench()
    rng = np.random.default_rng(4)
    n, L = 20000, 76
    seq = rng.integers(0, 4, n * L).astype(np.uint8)              # bwa_seq_t.seq: every read reversed
    s, r, off = b.adna_profile(seq, n, L, 9)
    lens = np.diff(off)
    assert off[0] == 0 and len(s) == len(r) == off[-1]
    assert lens.min() >= 50 and lens.max() <= L and len(set(lens.tolist())) == 27        # U{50..76}
    assert np.array_equal(r, 3 - s)                                # rseq: complement of seq (the read's reverse complement)
    s2, r2, off2 = b.adna_profile(seq, n, L, 9)
    assert np.array_equal(s, s2) and np.array_equal(off, off2)     # seeded
    changed = at5 = at3 = 0
    for i in range(0, n, 7):
        li = int(lens[i])
        new = s[off[i]:off[i + 1]][::-1]                           # back to sequencing order
        old = seq[i * L:(i + 1) * L][::-1][:li]                    # the first len bases of the original read
        d = np.nonzero(new != old)[0]
        changed += len(d)
        for x in d:
            if x < 12 and old[x] == 1 and new[x] == 3:             # C > T within 12 bases of the 5' end
                at5 += 1
            elif li - 1 - x < 12 and old[x] == 2 and new[x] == 0:  # G > A within 12 bases of the 3' end
                at3 += 1
            else:
                raise AssertionError("base %d of read %d changed %d -> %d" % (x, i, old[x], new[x]))
    assert changed == at5 + at3 and at5 > 100 and at3 > 100


def test_bench_command_line_parses_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--adna", "--pipeline"):
        assert flag in r.stdout


def test_gpus_without_a_launcher_starts_the_ranks_as_a_child(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset (the driver's call): one torch.distributed.run child with N ranks on 127.0.0.1,
    the same arguments passed on, its exit code returned -- and no rank count taken from anywhere else"""
    b = load_bench()
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = b.parse_args()
    assert b.launch_ranks(args) == 7
    c = seen["cmd"]
    assert c[0] == sys.executable and c[1:3] == ["-m", "torch.distributed.run"]
    assert c[c.index("--nproc-per-node") + 1] == "4" and c[c.index("--master-addr") + 1] == "127.0.0.1" and "--nnodes=1" in c
    i = c.index(os.path.join(T.ROOT, "bench.py"))
    assert c[i + 1:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
