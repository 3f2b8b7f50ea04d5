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


# Above this size a counter tensor is summed by exchanging its non-zero cells when that is less data
# (VS_SPARSE_ALLREDUCE_BYTES overrides; 0 = never).
SPARSE_MIN_BYTES = 2 << 30


def count_nonzero_cells(mats) -> int:
    """Non-zero cells of a counter tensor (chunks of 2^30 cells: the index kernels take 32-bit sizes)."""
    import torch

    flat = mats.view(-1)
    step = 1 << 30
    return int(sum(int(torch.count_nonzero(flat[lo:lo + step]).item()) for lo in range(0, flat.numel(), step)))


def sum_counts_sparse(mats) -> None:
    """In-place sum over the ranks of a mostly-zero counter tensor by exchanging (cell, count) lists: every rank
    gathers the others' non-zero cells and adds them into its own copy.  The [2,N,N] counters of a 54 k-node
    graph are 23.7 GB dense and hold ~1e8 non-zero cells after a rank's 25 M pairs; a dense ring
    all-reduce moves 1.75 x 23.7 GB per rank, this 7 x ~1.5 GB.  Same integers as the dense sum: additions into
    int32 storage wrap like the uint32 cells they stand for, int64 totals add as they are."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    flat = mats.view(-1)
    # (gloo gathers host tensors only: the functional two-ranks-on-one-GPU runs stage the lists through the
    # host; RCCL gathers device tensors)
    via_host = flat.is_cuda and dist.get_backend() == "gloo"
    step = 1 << 30
    parts = []
    for lo in range(0, flat.numel(), step):
        nz = torch.nonzero(flat[lo:lo + step]).view(-1)
        if nz.numel():
            parts.append(nz + lo)
    idx = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=flat.device)
    val = flat[idx]
    where = torch.device("cpu") if via_host else flat.device
    idx, val = idx.to(where), val.to(where)
    mine = torch.tensor([idx.numel()], dtype=torch.int64, device=where)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [int(x.item()) for x in sizes]
    longest = max(sizes)
    if longest == 0:
        return
    pad_idx = torch.zeros(longest, dtype=torch.int64, device=where)
    pad_val = torch.zeros(longest, dtype=flat.dtype, device=where)
    pad_idx[: idx.numel()] = idx
    pad_val[: idx.numel()] = val
    all_idx = [torch.empty_like(pad_idx) for _ in range(world)]
    all_val = [torch.empty_like(pad_val) for _ in range(world)]
    dist.all_gather(all_idx, pad_idx)
    dist.all_gather(all_val, pad_val)
    for r in range(world):
        if r != rank and sizes[r]:
            flat.index_add_(0, all_idx[r][: sizes[r]].to(flat.device), all_val[r][: sizes[r]].to(flat.device))
