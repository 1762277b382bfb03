"""The N>1 path on CPU: world_size-2 gloo processes shard the reads, results are merged in record order,
and the order-dependent hit choice runs over the merged stream.  The per-batch compute engine here is the
CPU oracle (this is a test of the sharding logic; on GPUs the same `run_sharded` wraps Index.cal_sa_reg_gap)."""
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import nabwa_testlib as T

shard = importlib.import_module("network-aware-bwa_amd.shard")

WORKER = r'''
import ctypes as C, importlib, os, sys, pickle
sys.path.insert(0, os.environ["NABWA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NABWA_ROOT"], "tests"))
import numpy as np, torch.distributed as dist
import nabwa_testlib as T
shard = importlib.import_module("network-aware-bwa_amd.shard")
dist.init_process_group("gloo")
lib = T.load_oracle(); ix = T.OracleIndex(lib)
opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
reads = T.read_fastq(os.path.join(T.GOLDEN, "reads_se.fq"))
def compute(lo, hi):
    seq, rseq, off, _ = T.encode_reads(reads[lo:hi])
    got, _ = T.oracle_cal_sa_reg_gap(lib, ix.h, opt, seq, rseq, off, per_read=1)
    return [g.tobytes() for g in got]
res = shard.run_sharded(len(reads), 37, compute, dist)
if dist.get_rank() == 0:
    pickle.dump(res, open(os.environ["NABWA_OUT"], "wb"))
dist.barrier(); dist.destroy_process_group()
'''


def test_plan_and_merge():
    plan = shard.plan_shards(10, 3, 4)
    assert plan == [(0, 0, 4), (1, 4, 8), (2, 8, 10)]
    assert shard.my_batches(shard.plan_shards(100, 2, 10), 1) == [(10, 20), (30, 40), (50, 60), (70, 80), (90, 100)]
    parts = [[(0, 2, ["a", "b"]), (4, 5, ["e"])], [(2, 4, ["c", "d"])]]
    assert shard.merge_in_order(5, parts) == ["a", "b", "c", "d", "e"]
    with pytest.raises(ValueError):
        shard.merge_in_order(5, [[(0, 2, ["a", "b"])]])                  # records missing
    with pytest.raises(ValueError):
        shard.merge_in_order(2, [[(0, 2, ["a", "b"])], [(1, 2, ["x"])]])  # produced twice
    assert shard.plan_shards(0, 2, 5) == []


def test_two_rank_gloo_run_matches_single_process():
    import pickle
    opt, gold = T.read_sai(os.path.join(T.GOLDEN, "se_default.sai"))
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "worker.py")
        open(script, "w").write(WORKER)
        out = os.path.join(td, "out.pkl")
        env = dict(os.environ, NABWA_ROOT=T.ROOT, NABWA_OUT=out, MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29731", script],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res = pickle.load(open(out, "rb"))
    # per-read results are independent of the sharding (default options: per-read == per-batch derivation)
    assert len(res) == len(gold)
    assert all(res[i] == gold[i].tobytes() for i in range(len(gold)))
