"""Multi-GPU sharding of the per-read path.

Reads are independent: the record stream is cut into contiguous batches by record number, the batches are
dealt round-robin to the ranks (one process per GPU, every rank holding a replica of the index) and the
per-batch results are put back in record order.  There is no collective on the data path; the only steps
that need the global order are the RNG-consuming hit choice (reference bwase.c:33-37) -- done by
rank 0 over the merged stream -- and, for paired-end data, the per-read-group insert-size histogram, a
host-side sum between the two passes (reference insert_size.c:141-167).
"""
from typing import Callable, List, Sequence, Tuple


def plan_shards(n_records: int, world: int, batch: int) -> List[Tuple[int, int, int]]:
    """-> [(rank, lo, hi)]: contiguous batches [lo, hi) by record number, dealt round-robin"""
    if world < 1 or batch < 1:
        raise ValueError("world and batch must be positive")
    plan = []
    for b, lo in enumerate(range(0, n_records, batch)):
        plan.append((b % world, lo, min(lo + batch, n_records)))
    return plan


def my_batches(plan: Sequence[Tuple[int, int, int]], rank: int) -> List[Tuple[int, int]]:
    return [(lo, hi) for r, lo, hi in plan if r == rank]


def merge_in_order(n_records: int, parts: Sequence[Sequence[Tuple[int, int, list]]]) -> list:
    """parts[rank] = [(lo, hi, per-record results)] -> one list in record order; checks full, disjoint coverage"""
    out = [None] * n_records
    seen = 0
    for per_rank in parts:
        for lo, hi, res in per_rank:
            if len(res) != hi - lo:
                raise ValueError("batch [%d,%d) came back with %d results" % (lo, hi, len(res)))
            for i, r in enumerate(res):
                if out[lo + i] is not None:
                    raise ValueError("record %d produced twice" % (lo + i))
                out[lo + i] = r
            seen += hi - lo
    if seen != n_records or any(r is None for r in out):
        raise ValueError("records missing after the merge")
    return out


def run_sharded(n_records: int, batch: int, compute: Callable[[int, int], list], dist=None) -> list:
    """Every rank runs `compute(lo, hi)` on its batches; rank 0 returns all results in record order
    (other ranks return None).  `dist` is torch.distributed (initialised) or None for a single process."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    plan = plan_shards(n_records, world, batch)
    mine = [(lo, hi, compute(lo, hi)) for lo, hi in my_batches(plan, rank)]
    if dist is None:
        return merge_in_order(n_records, [mine])
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    return merge_in_order(n_records, gathered) if rank == 0 else None
