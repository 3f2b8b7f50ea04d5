"""Multi-GPU plan for PE-link inference: read pairs are independent and the only shared state
is the two N x N counters (reference utils/VStrains_PE_Inference.py:139-140,174-188), so the
pairs are cut into contiguous blocks, one per rank, every rank holds its own copy of the node
index, and one sum all-reduce (RCCL over xGMI when the backend is "nccl") folds the counters.
Integer addition commutes: any partition gives bit-identical results."""
from __future__ import annotations

from typing import Tuple


def shard_range(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Pair r goes to rank floor(r * world / n_pairs): contiguous, sizes differ by at most one."""
    lo = (n_pairs * rank) // world
    hi = (n_pairs * (rank + 1)) // world
    return lo, hi


def group_size() -> int:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def all_reduce_counts(mats, stats):
    """In-place sum over ranks of the [2, N, N] counter tensor (int32 storage of uint32 cells, or
    int64 totals; may be None) and of a small int64 tensor (the 3 stats, or PeCounter's flags)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if mats is not None:
            dist.all_reduce(mats, op=dist.ReduceOp.SUM)
        if stats is not None:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)


def all_reduce_counts_async(mats, stats):
    """Same sums, enqueued behind the work already on the current stream and NOT waited for: returns
    the work handles (``w.wait()`` makes the then-current stream wait), or [] when there is no process
    group.  The caller keeps counting into a second buffer meanwhile (bench.py)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return []
    return [dist.all_reduce(mats, op=dist.ReduceOp.SUM, async_op=True),
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, async_op=True)]
