"""The N > 1 paths on the one-GPU box (two processes, gloo rendezvous, both on device 0 -- the 8-GPU run is the driver's):
 * shard.run_sharded with the GPU search as the per-batch engine: records dealt round-robin to the ranks, merged in record order,
   equal to the reference's .sai (the same helper the CPU gloo test drives with the oracle);
 * bench.py's own multi-rank path (per-rank shard seeds, barrier + max-over-ranks timing, rank 0 prints one line): n_gpus = 2,
   twice the per-rank reads in the rate, the bit-exact sample check on."""
import json
import os
import pickle
import subprocess
import sys
import tempfile

import pytest

import nabwa_testlib as T

pytestmark = pytest.mark.gpu

WORKER = r'''
import ctypes as C, importlib, os, sys, pickle
sys.path.insert(0, os.environ["NABWA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NABWA_ROOT"], "tests"))
import numpy as np, torch.distributed as dist
import nabwa_testlib as T
nabwa = importlib.import_module("network-aware-bwa_amd")
shard = importlib.import_module("network-aware-bwa_amd.shard")
dist.init_process_group("gloo")
ix = nabwa.Index.load(T.TOY, 0, True)                       # every rank: its own replica of the index (here: both on device 0)
opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
g = nabwa.GapOpt(); C.memmove(C.byref(g), C.byref(opt), 64)
reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
def compute(lo, hi):
    seq, rseq, off, _ = T.encode_reads(reads[lo:hi])
    got, _ = ix.cal_sa_reg_gap(g, seq, rseq, off, per_read=True)
    return [x.tobytes() for x in got]
res = shard.run_sharded(len(reads), 37, compute, dist)
if dist.get_rank() == 0:
    pickle.dump(res, open(os.environ["NABWA_OUT"], "wb"))
ix.close()
dist.barrier(); dist.destroy_process_group()
'''


def launch(args, env, timeout=900):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", "29741"] + args, env=e, capture_output=True, text=True, timeout=timeout, cwd=T.ROOT)


def test_two_ranks_shard_the_reads_on_the_gpu():
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "worker.py")
        open(script, "w").write(WORKER)
        out = os.path.join(td, "out.pkl")
        r = launch([script], dict(NABWA_ROOT=T.ROOT, NABWA_OUT=out))
        assert r.returncode == 0, r.stderr[-3000:]
        res = pickle.load(open(out, "rb"))
    assert len(res) == len(gold) and all(res[i] == gold[i].tobytes() for i in range(len(gold)))


def test_bench_two_rank_rehearsal():
    r = launch(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-len", "30000000", "--reads", "300000", "--cpu-seconds", "3", "--no-e2e"],
               dict(NABWA_BENCH_BACKEND="gloo", NABWA_BENCH_SINGLE_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]                  # rank 0 only
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["reads_per_gpu"] == 300000 and d["config"]["bit_exact_vs_cpu_sample"] is True
    assert abs(d["value"] - 2 * 300000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]        # whole-job rate: both ranks' reads over the slowest rank's time
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None          # the CPU leg is reported at N = 1 only; the sample check above ran all the same


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it): bench.py starts its two ranks itself, as a child
    process before anything touches the GPU, and relays ONE line with n_gpus = 2 -- the kernel-only rate and the records-in ->
    records-out leg, both over both ranks."""
    e = dict(os.environ, NABWA_BENCH_BACKEND="gloo", NABWA_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-len", "30000000", "--reads", "300000",
                        "--cpu-seconds", "3", "--e2e-reads", "50000"], env=e, capture_output=True, text=True, timeout=900, cwd=T.ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["bit_exact_vs_cpu_sample"] is True
    assert abs(d["value"] - 2 * 300000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert d["cpu_baseline"] is None                          # a CPU leg is reported at N = 1 only
    assert d["e2e"]["n_gpus"] == 2 and d["e2e"]["reads"] == 50000 and d["e2e"]["reads_per_s"] > 0
    assert d["e2e"]["bit_exact_vs_reference_sample"] is True
