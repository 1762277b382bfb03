"""Host-side paired-end pieces of libnabwa (no GPU needed): insert-size inference and pairing against
known answers produced by the reference's own functions (tests/golden/make_golden.py: make_pe_vectors)."""
import ctypes as C
import importlib
import os

import numpy as np

import nabwa_testlib as T

nabwa = importlib.import_module("network-aware-bwa_amd")


def test_isize_inference_golden():
    v = np.load(os.path.join(T.GOLDEN, "vectors_pe.npz"))
    for t in range(len(v["is_res"])):
        rec = v["is_hist"][v["is_hist_off"][t]:v["is_hist_off"][t + 1]]
        L = int(rec[0])
        body = rec[1:]
        k = int(np.where(body == -1)[0][0])
        h = np.zeros(100000, np.uint16)
        h[body[:k]] = body[k + 1:].astype(np.uint16)
        ap_in, avg, std, ap, low, high, hb = v["is_res"][t]
        rc, ii = nabwa.isize_infer(h, ap_in, L)
        # bit-exact doubles: same expression order, same libm
        assert (ii.avg, ii.std) == (avg, std), t
        assert (ii.low, ii.high, ii.high_bayesian) == (low, high, hb), t
        if avg >= 0 or ap != 0:
            assert ii.ap_prior == ap, t
        assert rc == (0 if avg >= 0 else -1)


def test_isize_bin():
    assert nabwa.lib().nabwa_isize_bin(2, 37, 37, 1000, 100, 1300, 100) == 400     # outer distance of the pair
    assert nabwa.lib().nabwa_isize_bin(2, 37, 37, 1300, 100, 1000, 100) == 400
    assert nabwa.lib().nabwa_isize_bin(2, 37, 19, 1000, 100, 1300, 100) == -1      # both ends need mapQ >= 20
    assert nabwa.lib().nabwa_isize_bin(1, 25, 0, 5, 76, 0, 0) == 76                # single read: its length
    assert nabwa.lib().nabwa_isize_bin(2, 37, 37, 1000, 100, 200000, 100) == -1    # >= MAX_ISIZE


def test_pairing_golden():
    v = np.load(os.path.join(T.GOLDEN, "vectors_pe.npz"))
    n = len(v["pr_cnt"])
    for t in range(n):
        na0, na1, max_isize = [int(x) for x in v["pr_misc"][t][:3]]
        iiv = v["pr_misc"][t][3:]
        ii = nabwa.IsizeInfo(iiv[0], iiv[1], iiv[2], int(iiv[3]), int(iiv[4]), int(iiv[5]))
        rows = v["pr_aln"][v["pr_aln_off"][t]:v["pr_aln_off"][t + 1]].astype(np.uint32)
        r0 = np.ascontiguousarray(rows[:4 * na0]).view(nabwa.ALN_DT)
        r1 = np.ascontiguousarray(rows[4 * na0:]).view(nabwa.ALN_DT)
        h = v["pr_hit"][v["pr_hit_off"][t]:v["pr_hit_off"][t + 1]].reshape(-1, 3)
        hits = (h[:, 0].astype(np.uint64) << np.uint64(32)) | (h[:, 1].astype(np.uint64) << np.uint64(1)) | h[:, 2].astype(np.uint64)
        ends = (nabwa.PeEnd * 2)()
        for e in range(2):
            q = v["pr_in"][t][11 * e:11 * e + 11]
            ends[e] = nabwa.PeEnd(*[int(x) for x in q])
        cnt = nabwa.pairing(ends, hits, r0, r1, max_isize, 3, ii)
        assert cnt == v["pr_cnt"][t], t
        for e in range(2):
            want = [int(x) for x in v["pr_out"][t][11 * e:11 * e + 11]]
            got = [getattr(ends[e], f) for f, _ in nabwa.PeEnd._fields_]
            assert got == want, (t, e)


def test_isize_add_pairs_bins_like_isize_bin_and_wraps():
    """improve_isize_est over a batch of positioned pairs (insert_size.c:141-165): the same bins as nabwa_isize_bin record by
    record, and the reference's uint16_t bins wrap"""
    rng = np.random.default_rng(3)
    n = 70000
    recs = (nabwa.PeRec * (2 * n))()
    want = np.zeros(100000, np.uint16)
    L = nabwa.lib()
    for i in range(n):
        a, b = recs[2 * i].se, recs[2 * i + 1].se
        a.mapQ, b.mapQ = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        a.len, b.len = 100, int(rng.integers(30, 101))
        a.pos = int(rng.integers(0, 1 << 20))
        b.pos = int(rng.integers(0, 1 << 20))
        if i % 3 == 0:                                                              # every third pair lands in ONE bin
            a.mapQ = b.mapQ = 37
            b.len, b.pos = 100, a.pos + 250
        k = L.nabwa_isize_bin(2, a.mapQ, b.mapQ, a.pos, a.len, b.pos, b.len)
        if k >= 0:
            want[k] += 4                                                            # numpy wraps uint16 the way the reference's array does
    got = np.zeros(100000, np.uint16)
    for _ in range(4):                                                              # four batches into the same histogram: 93 k in bin 350
        nabwa.isize_add_pairs(recs, n, got)
    assert (got == want).all()
    assert int(got[350]) == (4 * ((n + 2) // 3)) % 65536 and 4 * ((n + 2) // 3) > 65535
