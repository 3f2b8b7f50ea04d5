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
# slabs of at most SLAB_STRETCHES, so its transient is bounded (1 GB of int32 cells) whatever the union's size -- a counter
# of 23.7 GB half occupied used to ask for 12 GB on top of the counters.  
SLAB_STRETCHES = 1 << 22
_staging = {}


def _staging_buffer(head, want_rows: int):
    """[rows, 64] buffer of ``head``'s dtype on its device, rows = min(want_rows, SLAB_STRETCHES + 1), grown only; None when
    the allocation fails (the caller says so in the flag bytes of the first collective: no rank is left waiting)."""
    import torch

    rows = max(3, min(int(want_rows), SLAB_STRETCHES + 2))
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


# ---- the packed exchange (r6): two collectives per sum ----------------------------------------------------------------
# a step's exchange was an agreement all-reduce, the occupancy MAX, a host wait on torch.nonzero, a status
# MAX, the slab SUMs, the stats SUM and the tile-map MAX -- six or seven collectives whose latency on eight ranks nobody has
# measured.  Now:
#   C1  MAX over ONE uint8 tensor: [occupancy map of the 64-cell stretches (or the counter's dirty-tile map, from which
#       every rank derives the same occupancy) | FLAG_BYTES flag bytes].  MAX over bytes is OR for 0 / 1 flags.
#   C2  SUM over ONE staging tensor: [the occupied stretches of the union | one more row that carries the caller's small
#       int64 sums -- the three stats, the pairs in the buffers -- as 16-bit limbs in the cells' dtype].
# (a union of more than SLAB_STRETCHES stretches takes one SUM per slab; a dense union, a veto or a rank without staging
# memory send the whole tensor through the ring, and the tail row after it.)
FLAG_BYTES = 8
FLAG_NO_COMPACT, FLAG_WIDE, FLAG_NO_STAGING, FLAG_PAIRS_LOG2 = 0, 1, 2, 3
LIMBS = 4  # 16-bit limbs per int64 of the tail: exact while every addend is below 2^63 and there are fewer than 2^15 ranks


def _tail_encode(tail, row):
    """``tail`` (int64 [k], k <= 16, non-negative) into ``row`` ([64] of the cells' dtype): int64 cells take the values as
    they are, int32 cells four 16-bit limbs each (a sum over ranks of limbs stays far below 2^31)."""
    import torch

    row.zero_()
    k = tail.numel()
    if row.dtype == torch.int64:
        row[:k] = tail
    else:
        shifts = torch.arange(0, 16 * LIMBS, 16, dtype=torch.int64, device=tail.device).unsqueeze(1)
        row.view(LIMBS, 16)[:, :k] = ((tail.to(torch.int64).unsqueeze(0) >> shifts) & 0xFFFF).to(row.dtype)


def _tail_decode(row, k):
    import torch

    if row.dtype == torch.int64:
        return row[:k].clone()
    shifts = torch.arange(0, 16 * LIMBS, 16, dtype=torch.int64, device=row.device).unsqueeze(1)
    return (row.view(LIMBS, 16)[:, :k].to(torch.int64) << shifts).sum(dim=0)


class ExchangeState:
    """What a counter keeps between its exchanges: the union size of the last one (the next is staged for 1.25 times that
    without asking the device), and the unsettled predicted exchange."""

    def __init__(self):
        self.cap = None
        self.pending = None
        self.collectives = 0   # of the most recent call
        self.host_waits = 0


def sum_counts_packed(mats, tail=None, flags=None, tile_map=None, timing=None, occupancy_fn=None, state=None, predict=False, dst=None,
                      on_flags=None):
    """In-place sum over the ranks of a counter tensor (int32 storage of uint32 cells, or int64 totals), of ``tail`` (small
    int64 tensor, summed) and OR of ``flags`` (uint8 [FLAG_BYTES]; byte FLAG_PAIRS_LOG2 is MAXed as a number), in TWO
    collectives (see above).  -> (how, flags_of_all_ranks, summed_tail): how is "compact" or "dense", the same on every rank.

    ``predict`` (needs ``state`` with a union size from an earlier call): no host wait at all -- the gather is sized from
    the previous union (``torch.nonzero_static``), the device keeps the real size, and ``settle_exchange(state)`` -- called
    when the sums are consumed, steps later -- looks at it and at the flags and sums the stretches that did not fit then.
    ``dst``: the SUM collectives reduce to that rank only (the drop-in's writer); the other ranks' tensors keep their own
    counts.  ``on_flags``: called with the flag bytes of all ranks (a list) once the host has them; if it returns a tensor,
    THAT is what has to be summed instead (PeCounter folds into int64 totals when the uint32 buffers cannot hold the sum):
    the exchange starts over on it -- the rare path, two more collectives."""
    import time

    import torch
    import torch.distributed as dist

    def mark(name, t0):
        if timing is not None:
            if mats.is_cuda:
                torch.cuda.synchronize(mats.device)
            timing[name] = timing.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    if state is None:
        state = ExchangeState()
    state.collectives = state.host_waits = 0

    def c_sum(t):
        state.collectives += 1
        if dst is None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)

    mine = dst is None or dist.get_rank() == dst
    dev = mats.device
    flat = mats.view(-1)
    m = flat.numel() // STRETCH
    head = flat[: m * STRETCH].view(m, STRETCH)
    k_tail = 0 if tail is None else int(tail.numel())
    assert k_tail <= 16
    my_flags = torch.zeros(FLAG_BYTES, dtype=torch.uint8, device=dev) if flags is None else flags.to(device=dev, dtype=torch.uint8).clone()
    t = time.perf_counter()
    n = mats.shape[-1]
    use_tiles = tile_map is not None and mats.dim() == 3 and mats.shape[0] == 2 and mats.shape[1] == n and n >= STRETCH
    # the staging buffer BEFORE the first collective: a rank that cannot have it says so in its flag bytes
    want = (state.cap + 2) if (predict and state.cap is not None) else min(m // 2 + 3, 1 << 20)
    buf = _staging_buffer(head, want) if m else None
    if m and buf is None:
        my_flags[FLAG_NO_STAGING] = 1
    # C1 = [occupancy map (unless the tile map stands for it) | the counter's dirty-tile map, if it keeps one | flag bytes]
    occ_local = None if use_tiles else ((occupancy_fn(head) if occupancy_fn is not None else _occupancy(head)) if m
                                        else torch.zeros(0, dtype=torch.uint8, device=dev))
    n_occ = 0 if occ_local is None else occ_local.numel()
    n_tiles = 0 if tile_map is None else tile_map.numel()
    nb = n_occ + n_tiles
    c1 = torch.empty(nb + FLAG_BYTES, dtype=torch.uint8, device=dev)
    if n_occ:
        c1[:n_occ] = occ_local
    if n_tiles:
        c1[n_occ:nb] = tile_map
    c1[nb:] = my_flags
    t = mark("occupancy", t)
    state.collectives += 1
    dist.all_reduce(c1, op=dist.ReduceOp.MAX)
    t = mark("occupancy_allreduce", t)
    all_flags = c1[nb:]
    if n_tiles:
        tile_map.copy_(c1[n_occ:nb])  # (the sum brings the other ranks' cells: their tiles are dirty here too)
    occ = _occupancy_from_tiles(tile_map, n, m) if use_tiles else c1[:n_occ]

    def tail_only():  # (dense branch: the tail goes through on its own)
        if not k_tail:
            return None
        row = torch.empty(STRETCH, dtype=mats.dtype, device=dev)
        _tail_encode(tail, row)
        c_sum(row)
        return _tail_decode(row, k_tail)

    left = flat[m * STRETCH:]  # the cells behind the last whole stretch (fewer than 64): they ride in a row of their own
    n_left = int(left.numel())

    def extra_rows(stage, at):  # stage[at] <- the tail's limbs, stage[at + 1] <- the leftover cells
        if k_tail:
            _tail_encode(tail, stage[at])
        else:
            stage[at].zero_()
        stage[at + 1].zero_()
        if n_left:
            stage[at + 1][:n_left] = left

    def extra_back(stage, at):
        if n_left and mine:
            left.copy_(stage[at + 1][:n_left])
        return _tail_decode(stage[at], k_tail) if k_tail else None

    if predict and state.cap is not None and m and buf is not None:
        cap = min(state.cap, buf.shape[0] - 2)
        idx = torch.nonzero_static(occ, size=cap, fill_value=0).view(-1)
        stage = buf[: cap + 2]
        torch.index_select(head, 0, idx, out=stage[:cap])
        extra_rows(stage, cap)
        t = mark("gather", t)
        c_sum(stage)
        t = mark("ring", t)
        if mine:
            head.index_copy_(0, idx, stage[:cap])
        out_tail = extra_back(stage, cap)
        mark("scatter", t)
        state.pending = {"u": occ.sum(dtype=torch.int64), "cap": cap, "flags": all_flags, "occ": occ, "head": head, "dst": dst}
        return "compact", all_flags, out_tail

    # one host wait: the flags and the union's size together
    if m:
        word = torch.cat([all_flags.to(torch.int64), occ.sum(dtype=torch.int64).view(1)]).cpu()
        state.host_waits += 1
        fl, u = word[:FLAG_BYTES].tolist(), int(word[FLAG_BYTES])
    else:
        fl, u = all_flags.cpu().tolist(), 0
        state.host_waits += 1
    t = mark("nonzero", t)
    if timing is not None:
        timing["stretches"] = m
        timing["occupied_stretches_of_the_union"] = u
    if on_flags is not None:
        other = on_flags(fl)
        if other is not None:
            done = state.collectives
            clean = torch.zeros(FLAG_BYTES, dtype=torch.uint8)
            clean[FLAG_NO_COMPACT] = 1 if fl[FLAG_NO_COMPACT] else 0
            res = sum_counts_packed(other, tail, clean, tile_map=None, timing=timing, occupancy_fn=occupancy_fn, state=state, dst=dst)
            state.collectives += done
            return res
    state.cap = max(64, int(1.25 * u) + 16)
    if m == 0 or fl[FLAG_NO_COMPACT] or fl[FLAG_NO_STAGING] or u > COMPACT_MAX_FILL * m:
        t = time.perf_counter()
        c_sum(mats)
        out_tail = tail_only()
        mark("ring", t)
        return "dense", all_flags, out_tail
    idx = torch.nonzero_static(occ, size=u, fill_value=0).view(-1)
    bigger = _staging_buffer(head, u + 2)  # (grown when it can be: fewer slabs; the old one serves otherwise)
    if bigger is not None:
        buf = bigger
    rows = buf.shape[0] - 2
    out_tail = None
    lo = 0
    while True:
        part = idx[lo:lo + rows]
        k = int(part.numel())
        last = lo + k >= u
        stage = buf[: k + (2 if last else 0)]
        t = time.perf_counter()
        if k:
            torch.index_select(head, 0, part, out=stage[:k])
        if last:
            extra_rows(stage, k)
        t = mark("gather", t)
        c_sum(stage)
        t = mark("ring", t)
        if mine and k:
            head.index_copy_(0, part, stage[:k])
        if last:
            out_tail = extra_back(stage, k)
        mark("scatter", t)
        lo += k
        if last:
            break
    return "compact", all_flags, out_tail


def settle_exchange(state) -> None:
    """The deferred half of a predicted exchange: ONE host read (the union's real size and the flag bytes, both long since
    computed), then -- every rank alike, they all hold the same numbers -- the stretches beyond the predicted size summed
    now, and the prediction raised.  Raises if a flag says the predicted exchange was not allowed (a rank vetoed the compact
    form, held int64 totals or had no staging memory): the bench's steady state checks those statically."""
    import torch
    import torch.distributed as dist

    p = state.pending
    if p is None:
        return
    state.pending = None
    word = torch.cat([p["flags"].to(torch.int64), p["u"].view(1)]).cpu()
    state.host_waits += 1
    fl, u, cap = word[:FLAG_BYTES].tolist(), int(word[FLAG_BYTES]), p["cap"]
    if fl[FLAG_NO_COMPACT] or fl[FLAG_WIDE] or fl[FLAG_NO_STAGING]:
        raise RuntimeError("a predicted exchange ran although a rank's flags forbid it: %r" % (fl,))
    if u > cap:
        head, dst = p["head"], p["dst"]
        rest = torch.nonzero_static(p["occ"], size=u, fill_value=0).view(-1)[cap:]
        mine = dst is None or dist.get_rank() == dst
        for lo in range(0, int(rest.numel()), SLAB_STRETCHES):
            part = rest[lo:lo + SLAB_STRETCHES]
            stage = head.index_select(0, part)
            state.collectives += 1
            if dst is None:
                dist.all_reduce(stage, op=dist.ReduceOp.SUM)
            else:
                dist.reduce(stage, dst=dst, op=dist.ReduceOp.SUM)
            if mine:
                head.index_copy_(0, part, stage)
    state.cap = max(state.cap or 0, int(1.25 * u) + 16, 64)


def sum_counts_compact(mats, allow_compact: bool = True, tile_map=None, timing=None, occupancy_fn=None) -> str:
    """In-place sum over the ranks of a counter tensor by way of its occupied 64-cell stretches: ``sum_counts_packed``
    without a tail.  Returns "compact" or "dense" (what was done; the same on every rank).  Same integers as
    ``all_reduce(SUM)`` on the whole tensor: the stretches that are left out are zero on every rank."""
    import torch

    flags = torch.zeros(FLAG_BYTES, dtype=torch.uint8)
    if not allow_compact:
        flags[FLAG_NO_COMPACT] = 1
    return sum_counts_packed(mats, flags=flags, tile_map=tile_map, timing=timing, occupancy_fn=occupancy_fn)[0]
