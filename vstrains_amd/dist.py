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


def _occupancy_from_tiles(tile_map, n: int, m: int):
    """The same map from a counter's dirty-tile bytes (``PeCounter.tile_map``: one byte per 64 x 64 tile of node_mat, then
    of short_mat, set wherever a block added to a cell): stretch j of the flat [2, n, n] buffer is taken as occupied when
    any tile it overlaps is marked -- a superset of the non-zero stretches, found without reading the 23.7 GB of counters a
    54 k-node graph has.  A stretch is 64 consecutive cells of the flat buffer: it lies in one or two matrix rows and
    in one or two tile columns of each."""
    import torch

    T = (n + 63) // 64
    dev = tile_map.device
    first = torch.arange(m, dtype=torch.int64, device=dev) * STRETCH
    last = first + (STRETCH - 1)
    # A stretch (64 <= n cells) lies in one matrix row or in the end of one and the start of the next; each part is shorter
    # than a tile is wide, so it overlaps at most two tile columns: those of its first and of its last cell.  Four cells
    # name every tile: the stretch's first and last, the last cell of the first row, the first cell of the last row.
    row_end = torch.minimum(last, (first // n + 1) * n - 1)
    row_start = torch.maximum(first, (last // n) * n)
    occ = torch.zeros(m, dtype=torch.uint8, device=dev)
    for cell in (first, row_end, row_start, last):
        grow = cell // n  # row of the flat [2 n, n] buffer
        col = cell - grow * n
        mat = grow // n
        row = grow - mat * n
        occ |= tile_map[(mat * T + (row >> 6)) * T + (col >> 6)]
    return occ


# One bounded staging buffer per (device, dtype) for the gathered stretches: the exchange walks the occupied stretches in
# slabs of at most SLAB_STRETCHES, so its transient is 256 MB (int32) whatever the union's size -- a counter of 23.7 GB half
# occupied used to ask for 12 GB on top of the counters.  
SLAB_STRETCHES = 1 << 20
_staging = {}


def _staging_buffer(head, want_rows: int):
    """[rows, 64] buffer of ``head``'s dtype on its device, rows = min(want_rows, SLAB_STRETCHES), grown only; None when the
    allocation fails (the caller tells its peers before any of them enters the collective)."""
    import torch

    rows = max(1, min(int(want_rows), SLAB_STRETCHES))
    key = (str(head.device), head.dtype)
    buf = _staging.get(key)
    if buf is not None and buf.shape[0] >= rows:
        return buf
    try:
        buf = torch.empty((rows, STRETCH), dtype=head.dtype, device=head.device)
    except (RuntimeError, MemoryError):  # (torch.cuda.OutOfMemoryError is a RuntimeError)
        return None
    _staging[key] = buf
    return buf


def sum_counts_compact(mats, allow_compact: bool = True, tile_map=None, timing=None, occupancy_fn=None) -> str:
    """In-place sum over the ranks of a counter tensor (int32 storage of uint32 cells, or int64 totals) by way of its
    occupied 64-cell stretches.  Returns "compact" or "dense" (what was done; the same on every rank).  Same integers as
    ``all_reduce(SUM)`` on the whole tensor: the stretches that are left out are zero on every rank.

    ``occupancy_fn``: head ([m, 64] view) -> uint8 [m] on the same device (default: a torch expression).
    ``tile_map``: the counter's dirty-tile bytes (uint32 buffers of 2 GiB and more keep them); the occupancy then comes
    from that map instead of a scan of the buffer.  ``timing``: a dict that receives the seconds of every phase
    (measurement runs: each phase is followed by a device synchronisation).
    Collectives of one call, in order, on every rank alike: the occupancy map (MAX), one status word (MAX), then the
    staged slabs (SUM) -- or the whole tensor (SUM) when the union is dense or some rank could not get its staging buffer."""
    import time

    import torch
    import torch.distributed as dist

    def mark(name, t0):
        if timing is not None:
            if mats.is_cuda:
                torch.cuda.synchronize(mats.device)
            timing[name] = timing.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    flat = mats.view(-1)
    m = flat.numel() // STRETCH
    if not allow_compact or m == 0:
        t = time.perf_counter()
        dist.all_reduce(mats, op=dist.ReduceOp.SUM)
        mark("ring", t)
        return "dense"
    head = flat[: m * STRETCH].view(m, STRETCH)
    t = time.perf_counter()
    n = mats.shape[-1]
    use_tiles = tile_map is not None and mats.dim() == 3 and mats.shape[0] == 2 and mats.shape[1] == n and n >= STRETCH
    # (occupancy_fn: the library's one-pass kernel over the buffer, vs_counts_occupied -- PeCounter hands it in for device
    # tensors; the torch expression below makes three passes and a [m, 64] temporary)
    occ = _occupancy_from_tiles(tile_map, n, m) if use_tiles else occupancy_fn(head) if occupancy_fn is not None else _occupancy(head)
    t = mark("occupancy", t)
    dist.all_reduce(occ, op=dist.ReduceOp.MAX)  # union of the ranks' occupancy maps
    t = mark("occupancy_allreduce", t)
    idx = torch.nonzero(occ).view(-1)  # (the one host wait of the exchange: the union's size decides the branch)
    u = int(idx.numel())
    t = mark("nonzero", t)
    if timing is not None:
        timing["stretches"] = m
        timing["occupied_stretches_of_the_union"] = u
    # the staging buffer BEFORE any rank commits to the compact branch: a rank that cannot have it says so, and all take
    # the dense ring together (a lone rank raising here would leave its peers waiting in the collective)
    buf = _staging_buffer(head, u) if 0 < u <= COMPACT_MAX_FILL * m else None
    status = torch.tensor([1 if (0 < u <= COMPACT_MAX_FILL * m and buf is None) else 0], dtype=torch.int32, device=mats.device)
    if 0 < u <= COMPACT_MAX_FILL * m:
        dist.all_reduce(status, op=dist.ReduceOp.MAX)
    if u > COMPACT_MAX_FILL * m or int(status.item()):
        t = time.perf_counter()
        dist.all_reduce(mats, op=dist.ReduceOp.SUM)
        mark("ring", t)
        return "dense"
    for lo in range(0, u, SLAB_STRETCHES):
        part = idx[lo:lo + SLAB_STRETCHES]
        stage = buf[: part.numel()]
        t = time.perf_counter()
        torch.index_select(head, 0, part, out=stage)
        t = mark("gather", t)
        dist.all_reduce(stage, op=dist.ReduceOp.SUM)
        t = mark("ring", t)
        head.index_copy_(0, part, stage)
        mark("scatter", t)
    tail = flat[m * STRETCH:]
    if tail.numel():
        dist.all_reduce(tail, op=dist.ReduceOp.SUM)
    return "compact"
