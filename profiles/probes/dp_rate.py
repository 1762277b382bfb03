"""Rates of the three alignment kernels on the shapes the finishing chains give them:
    python3 profiles/probes/dp_rate.py            -> one JSON line (cells, wall seconds per call incl. transfers)
under `rocprofv3 --kernel-trace --stats` the per-kernel averages of the same run are the kernel times (profiles/r02_dp_kernel_stats.csv).
  global : bwa_refine_gapped's call (bwase.c:212): read of 100 bases against its reference window, aln_param_bwa (band 50: the whole matrix)
  local  : bwa_sw_core's call (bwape.c:456): a 150-base mate against a window of 2 x 150 + 6 sigma = 540 bases
  extend : aln_extend_core on the same windows (not on bam2bam's path; kept for completeness)"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
nabwa = importlib.import_module("network-aware-bwa_amd")

MAQ = np.array([11, -19, -19, -19, -13, -19, 11, -19, -19, -13, -19, -19, 11, -19, -13, -19, -19, -19, 11, -13, -13, -13, -13, -13, -13], np.int32)


def pairs(n, lq, lr, rng, indel=True):
    ref = rng.integers(0, 4, (n, lr), dtype=np.uint8)
    start = (lr - lq) // 2
    qry = ref[:, start:start + lq].copy()
    mut = rng.random((n, lq)) < 0.02
    qry[mut] = (qry[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) & 3
    if indel:                                   # a one-base deletion in the read at a random place
        q = rng.integers(15, lq - 15, n)
        for i in range(n):
            qry[i, q[i]:-1] = qry[i, q[i] + 1:]
    return ref.reshape(-1), np.arange(n + 1, dtype=np.int64) * lr, qry.reshape(-1), np.arange(n + 1, dtype=np.int64) * lq


def main():
    rng = np.random.default_rng(5)
    out = {}
    n = 200_000
    ref, ro, qry, qo = pairs(n, 100, 101, rng)
    nabwa.global_align(ref[:101 * 1000], ro[:1001], qry[:100 * 1000], qo[:1001], 26, 9, 5, MAQ, 50)          # code objects, pools
    t = time.perf_counter(); sc, cg = nabwa.global_align(ref, ro, qry, qo, 26, 9, 5, MAQ, 50); dt = time.perf_counter() - t
    out["global"] = {"pairs": n, "shape": "100 x 101, band 50", "cells": n * 100 * 101, "wall_s": round(dt, 4), "gapped": int(sum(1 for c in cg if len(c) > 1))}
    n = 100_000
    ref, ro, qry, qo = pairs(n, 150, 540, rng, indel=False)
    nabwa.local_align(ref[:540 * 1000], ro[:1001], qry[:150 * 1000], qo[:1001], 26, 9, MAQ, 50)
    t = time.perf_counter(); sc, co, su, cg = nabwa.local_align(ref, ro, qry, qo, 26, 9, MAQ, 50); dt = time.perf_counter() - t
    out["local"] = {"pairs": n, "shape": "150 x 540", "cells_forward": n * 150 * 540, "wall_s": round(dt, 4), "mean_score": float(sc.mean())}
    g0 = np.full(n, 20, np.int32)
    nabwa.extend_align(ref[:540 * 1000], ro[:1001], qry[:150 * 1000], qo[:1001], 26, 9, MAQ, 50, g0[:1000])
    t = time.perf_counter(); sc, cg = nabwa.extend_align(ref, ro, qry, qo, 26, 9, MAQ, 50, g0); dt = time.perf_counter() - t
    out["extend"] = {"pairs": n, "shape": "150 x 540, band 50", "wall_s": round(dt, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
