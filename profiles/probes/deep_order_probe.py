"""Which feature known BEFORE kernel D runs predicts how long a search will take there?  (The launch ends with its longest search;
the work order decides when that one starts.)   NABWA_DEEP_DUMP=<file> python3 bench.py --adna ... ; python3 deep_order_probe.py <file>
Simulates list scheduling on W waves with the rounds a search took as its cost, for several orders."""
import heapq
import sys

import numpy as np

d = np.fromfile(sys.argv[1], np.int32).reshape(-1, 8)
rid, ln, md, c0, c1, trips, naln, rounds = d.T
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
print("searches", len(d), "rounds: sum", rounds.sum(), "max", rounds.max(), "mean", rounds.mean())
top = np.argsort(-rounds)[:15]
print("longest 15: rounds, position in the launch order, len, max_diff, cls, S trips, S hits")
for t in top:
    print("  ", rounds[t], t, ln[t], md[t], (c0[t], c1[t]), trips[t], naln[t])


def makespan(order):
    h = [0] * W
    heapq.heapify(h)
    for t in order:
        heapq.heappush(h, heapq.heappop(h) + int(rounds[t]))
    return max(h)


cur = np.arange(len(d))
key = md - np.minimum(c0, c1)
cands = {
    "as launched": cur,
    "ideal (longest first)": np.argsort(-rounds, kind="stable"),
    "random": np.random.default_rng(1).permutation(len(d)),
    "max_diff desc": np.argsort(-md, kind="stable"),
    "len desc": np.argsort(-ln, kind="stable"),
    "S trips asc (fast to fill the arena)": np.argsort(trips, kind="stable"),
    "S trips desc": np.argsort(-trips, kind="stable"),
    "key desc, then S trips asc": np.lexsort((trips, -key)),
    "key desc, then len desc": np.lexsort((-ln, -key)),
    "max_diff desc, then S trips asc": np.lexsort((trips, -md)),
    "S hits asc (no hit yet first), then max_diff desc": np.lexsort((-md, naln)),
    "no hit first, then key desc, then trips asc": np.lexsort((trips, -key, naln > 0)),
}
lb = max(int(rounds.max()), int(rounds.sum() // W))
print("lower bound (max of longest search and mean load):", lb)
for name, o in cands.items():
    print("%-55s makespan %8d  (%.2f x bound)" % (name, makespan(o), makespan(o) / lb))
for name, v in (("len", ln), ("max_diff", md), ("key", key), ("S trips", trips), ("S hits", naln), ("min cls", np.minimum(c0, c1))):
    r = np.corrcoef(np.argsort(np.argsort(v)), np.argsort(np.argsort(rounds)))[0, 1]
    print("rank correlation of rounds with %-10s %.3f" % (name, r))
