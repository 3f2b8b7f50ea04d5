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


def all_reduce_max(t):
    """In-place maximum over the ranks of a small tensor."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)


def all_reduce_counts_async(mats, stats):
    """Same sums, enqueued behind the work already on the current stream and NOT waited for: returns
    the work handles (``w.wait()`` makes the then-current stream wait), or [] when there is no process
    group.  The caller keeps counting into a second buffer meanwhile (bench.py)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return []
    return [dist.all_reduce(mats, op=dist.ReduceOp.SUM, async_op=True),
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, async_op=True)]


# ---- exchange of the occupied stretches only -----------------------------------------------------------------------
# The counters are mostly zero (a pair of 2x150 bp touches a dozen nodes that lie next to each other in the index's path
# numbering, so the non-zero cells of [2,N,N] sit in short runs along a band) and every rank's non-zero cells lie in
# about the SAME places (reads are sharded at random over the same genomes).  So the ranks first OR their occupancy maps
# of 64-cell stretches (256 bytes of uint32 counters: one small all-reduce), gather the occupied stretches of the UNION
# into a dense [U, 64] tensor, all-reduce THAT with the ordinary ring, and scatter the sums back.  What moves is
# U * 256 bytes through the ring instead of the whole buffer; the decision compact / dense follows from U, which every
# rank knows after the first all-reduce, so no rank can take another branch than its peers.
STRETCH = 64          # counter cells per stretch
COMPACT_MAX_FILL = 0.5  # above this share of occupied stretches the dense ring is no worse


def strong_share(total_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Strong scaling: the job's pairs are fixed and rank r takes pairs [lo, hi) of them (``shard_range``)."""
    return shard_range(total_pairs, rank, world)


def _occupancy(head):
    """uint8 [M]: 1 where a 64-cell stretch of ``head`` ([M, 64]) holds a non-zero cell (slabs bound the temporaries)."""
    import torch

    m = head.shape[0]
    occ = torch.empty(m, dtype=torch.uint8, device=head.device)
    slab = 1 << 22  # stretches per slab: 1 GiB of int32 cells
    for lo in range(0, m, slab):
        occ[lo:lo + slab] = (head[lo:lo + slab] != 0).any(dim=1)
    return occ


def sum_counts_compact(mats, allow_compact: bool = True) -> str:
    """In-place sum over the ranks of a counter tensor (int32 storage of uint32 cells, or int64 totals) by way of its
    occupied 64-cell stretches.  Returns "compact" or "dense" (what was done; the same on every rank).  Same integers as
    ``all_reduce(SUM)`` on the whole tensor: the stretches that are left out are zero on every rank."""
    import torch
    import torch.distributed as dist

    flat = mats.view(-1)
    m = flat.numel() // STRETCH
    if not allow_compact or m == 0:
        dist.all_reduce(mats, op=dist.ReduceOp.SUM)
        return "dense"
    head = flat[: m * STRETCH].view(m, STRETCH)
    occ = _occupancy(head)
    dist.all_reduce(occ, op=dist.ReduceOp.MAX)  # union of the ranks' occupancy maps
    idx = torch.nonzero(occ).view(-1)
    if idx.numel() > COMPACT_MAX_FILL * m:
        dist.all_reduce(mats, op=dist.ReduceOp.SUM)
        return "dense"
    if idx.numel():
        compact = head.index_select(0, idx)
        dist.all_reduce(compact, op=dist.ReduceOp.SUM)
        head.index_copy_(0, idx, compact)
    tail = flat[m * STRETCH:]
    if tail.numel():
        dist.all_reduce(tail, op=dist.ReduceOp.SUM)
    return "compact"
