"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the host-side
text handling matches the oracle (= the reference's semantics), and the multi-rank fold works
over gloo."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pe_cases
from oracle import pe_oracle


def test_library_exports_every_symbol_of_the_header():
    from vstrains_amd import _native

    L = _native.lib()  # raises if the .so is missing or lacks a symbol in SYMBOLS
    header = open(os.path.join(ROOT, "include", "vstrains_hip.h")).read()
    declared = set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", header))
    declared -= {"vs_ctx", "vs_reads"}
    from vstrains_amd.graph import native_stage

    bound = set(_native.SYMBOLS) | set(native_stage.STAGE_SYMBOLS)  # (the stage handle's prototypes live next to its blob layout)
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(L, name)
    assert L.vs_abi_version() == 10


def test_no_device_means_loud_failure_not_fallback():
    import torch
    from vstrains_amd import _native, pe as host

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_native.NativeError) as ei:
        host.Context(0)
    assert "no CPU path" in str(ei.value)


@pytest.mark.parametrize("name,d,meta", pe_cases(ok_only=False), ids=[c[0] for c in pe_cases(ok_only=False)])
def test_host_gfa_segment_reader_matches_reference_semantics(name, d, meta):
    from vstrains_amd import pe as host

    assert host.read_gfa_segments(os.path.join(d, "graph.gfa")) == pe_oracle.read_gfa_segments(os.path.join(d, "graph.gfa"))


def test_matrix_writer_matches_reference_format(tmp_path):
    from vstrains_amd import pe as host

    ids = ["0", "7&8*0", "x"]
    mat = np.arange(9, dtype=np.int64).reshape(3, 3) * 1000003
    p = tmp_path / "pe_info"
    host.write_matrix_text(str(p), ids, mat)
    assert open(p).read() == pe_oracle.matrix_text(ids, mat)
    # the writer works in blocks of whole rows (256 MB; a 50 k-node matrix is 36 GB of text): the same bytes
    # whatever the block size, also when one row is larger than a block
    rng = np.random.default_rng(3)
    ids = ["n%d%s" % (i, "*A" * (i % 3)) for i in range(57)]
    mat = rng.integers(0, 10 ** 9, size=(57, 57)).astype(np.int64) * (rng.random((57, 57)) < 0.3)
    want = pe_oracle.matrix_text(ids, mat)
    for block in ("1", "100", "2000", "100000"):
        os.environ["VS_TEXT_BLOCK"] = block
        try:
            host.write_matrix_text(str(p), ids, mat)
        finally:
            os.environ.pop("VS_TEXT_BLOCK")
        assert open(p).read() == want, block


def _rank_main(rank, world, port, case_dir, k, q):
    import torch
    import torch.distributed as dist

    from oracle import pe_oracle_c
    from vstrains_amd import dist as vdist

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ids, seqs = pe_oracle.read_gfa_segments(os.path.join(case_dir, "graph.gfa"))
    f = pe_oracle.fastq_sequences(os.path.join(case_dir, "fwd.fq"))
    r = pe_oracle.fastq_sequences(os.path.join(case_dir, "rve.fq"))
    n = min(len(f), len(r))
    lo, hi = vdist.shard_range(n, rank, world)
    # the oracle stands in for the device kernel here: what is under test is the shard + reduce logic
    node, short, stats = pe_oracle_c.Oracle(seqs, k).count_pairs(f[lo:hi], r[lo:hi])
    mats = torch.from_numpy(np.stack([node, short]).astype(np.int32))
    st = torch.from_numpy(stats.astype(np.int64))
    vdist.all_reduce_counts(mats, st)
    if rank == 0:
        q.put((mats.numpy().astype(np.int64), st.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_read_sharding_and_allreduce_gloo():
    import torch.multiprocessing as mp

    name, d, meta = [c for c in pe_cases() if c[0] == "hiv_like_k55"][0]
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctxm.Process(target=_rank_main, args=(rk, 2, port, d, meta["k"], q)) for rk in range(2)]
    for p in procs:
        p.start()
    mats, st = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids, seqs = pe_oracle.read_gfa_segments(os.path.join(d, "graph.gfa"))
    assert pe_oracle.matrix_text(ids, mats[0]) == open(os.path.join(d, "pe_info")).read()
    assert pe_oracle.matrix_text(ids, mats[1]) == open(os.path.join(d, "st_info")).read()
    assert int(st.sum()) == min(len(pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq"))),
                                 len(pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))))


def _rank_overlap(rank, world, port, q):
    """bench.py's step scheme on CPU tensors: two counter buffers, the all-reduce of step i is only
    waited for when its buffer comes up again (step i+2) or at the end."""
    import torch
    import torch.distributed as dist

    from vstrains_amd import dist as vdist

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    bufs = [(torch.zeros((2, 5, 5), dtype=torch.int32), torch.zeros(3, dtype=torch.int64)) for _ in range(2)]
    pending = [[], []]
    seen = []
    for step in range(5):
        b = step % 2
        for wk in pending[b]:
            wk.wait()
        if step >= 2:
            seen.append((step - 2, bufs[b][0].clone(), bufs[b][1].clone()))  # result of step-2, now final
        mats, st = bufs[b]
        mats.zero_()
        st.zero_()
        mats += (rank + 1) * (step + 1)
        st += rank + 10 * step
        pending[b] = vdist.all_reduce_counts_async(mats, st)
    for b in range(2):
        for wk in pending[b]:
            wk.wait()
    seen.append((3, bufs[1][0].clone(), bufs[1][1].clone()))
    seen.append((4, bufs[0][0].clone(), bufs[0][1].clone()))
    if rank == 0:
        q.put([(i, m.numpy(), s_.numpy()) for i, m, s_ in seen])
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_allreduce_with_two_buffers_gloo():
    import torch.multiprocessing as mp

    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + ((os.getpid() + 17) % 500)
    procs = [ctxm.Process(target=_rank_overlap, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    seen = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(i for i, _, _ in seen) == [0, 1, 2, 3, 4]
    for step, mats, st in seen:
        assert (mats == (1 + 2) * (step + 1)).all()           # rank 0 + rank 1 contributions of that step
        assert (st == (0 + 1) + 2 * 10 * step).all()


def _cpu_counter(n, pairs_in_buffer):
    """PeCounter on CPU tensors with the device fold (vs_counts_fold) replaced by its definition:
    what is under test is the range bookkeeping and the collective logic around it."""
    import torch

    from vstrains_amd import pe as host

    class FakeCtx:
        n_nodes = n
        device = 0

    class CpuCounter(host.PeCounter):
        def fold(self):
            if self.wide is None:
                self.wide = torch.zeros(self.mats.shape, dtype=torch.int64)
            self.wide += torch.from_numpy(self.mats.numpy().view(np.uint32).astype(np.int64))
            self.mats.zero_()
            self.pairs_in_buffer = 0

    c = CpuCounter(FakeCtx(), device="cpu")
    c.pairs_in_buffer = pairs_in_buffer
    return c


def _set_cell(counter, mat, i, j, value):
    counter.mats[mat, i, j] = int(np.array([value], dtype=np.uint32).view(np.int32)[0])


def test_counter_cells_above_2_31_come_back_positive():
    """uint32 cells in int32 storage: values in [2^31, 2^32) must widen as unsigned (the diagonal of
    short_mat takes two increments per pair, PE_Inference.py:174-184), and folded totals add on top."""
    c = _cpu_counter(3, 2 ** 31 - 1)
    _set_cell(c, 0, 1, 2, 2 ** 31 + 5)
    _set_cell(c, 1, 1, 1, 2 ** 32 - 1)
    node, short, _ = c.result()
    assert node[1, 2] == 2 ** 31 + 5 and short[1, 1] == 2 ** 32 - 1 and node.min() >= 0 and short.min() >= 0
    c.fold()
    assert int(c.mats.abs().sum()) == 0 and c.pairs_in_buffer == 0
    _set_cell(c, 1, 1, 1, 2 ** 32 - 1)
    node, short, _ = c.result()
    assert node[1, 2] == 2 ** 31 + 5 and short[1, 1] == 2 * (2 ** 32 - 1)


def _rank_range(rank, world, port, q):
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    out = []
    # (a) the uint32 buffers can hold the sum: summed as they are (int32 storage, unsigned meaning)
    a = _cpu_counter(16, 2 ** 29)
    _set_cell(a, 1, 2, 2, 2 ** 31 + 5 if rank == 0 else 2 ** 31 - 10)
    _set_cell(a, 0, 0, 3, 7 + rank)
    a.all_reduce()
    out.append(("u32", a.wide is None, a.pairs_in_buffer, a.result()))
    how = [getattr(a, "last_all_reduce", None)]
    # (b) they cannot: every rank folds, the int64 totals are summed
    b = _cpu_counter(16, 2 ** 31 - 1)
    _set_cell(b, 1, 2, 2, 2 ** 32 - 1)
    _set_cell(b, 0, 0, 3, 2 ** 31 + rank)
    b.all_reduce()
    out.append(("wide", b.wide is not None, b.pairs_in_buffer, b.result()))
    how.append(getattr(b, "last_all_reduce", None))
    # (c) one rank already holds int64 totals: all of them switch
    c = _cpu_counter(4, 10)
    _set_cell(c, 0, 1, 1, 5)
    if rank == 1:
        c.fold()
        _set_cell(c, 0, 1, 1, 3)
    c.all_reduce()
    out.append(("mixed", c.wide is not None, c.pairs_in_buffer, c.result()))
    # the overlapped form refuses what it cannot hold
    d = _cpu_counter(4, 2 ** 30)
    try:
        d.all_reduce_async()
        out.append(("async", False, 0, None))
    except OverflowError:
        out.append(("async", True, 0, None))
    # the two exchanges on random mostly-zero counters, cell for cell (values beyond 2^31 included): a banded one (few
    # occupied stretches: compact), a dense one (falls back to the whole buffer), sizes that leave a tail of < 64 cells
    import torch

    from vstrains_amd import dist as vdist

    gen = torch.Generator().manual_seed(100 + rank)
    modes = []
    for dtype, top in ((torch.int32, 2 ** 31 - 1), (torch.int64, 2 ** 40)):
        for n, fill in ((61, 0.002), (61, 0.9), (64, 0.002), (5, 0.5)):
            m = torch.randint(0, top, (2, n, n), generator=gen, dtype=torch.int64)
            m = (m * (torch.rand((2, n, n), generator=gen) < fill)).to(dtype)
            if dtype == torch.int32:
                m[0, 3, 4] = -7  # (uint32 2^32 - 7 in int32 storage)
            dense, compact = m.clone(), m.clone()
            dist.all_reduce(dense, op=dist.ReduceOp.SUM)
            modes.append(vdist.sum_counts_compact(compact))
            assert torch.equal(dense, compact)
            off = m.clone()
            assert vdist.sum_counts_compact(off, allow_compact=False) == "dense" and torch.equal(off, dense)
    assert modes[:4] == ["compact", "dense", "compact", "dense"], modes
    assert vdist.strong_share(10, 0, 2) == (0, 5) and vdist.strong_share(11, 1, 2) == (5, 11)
    if rank == 0:
        q.put((out, how))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["dense", "compact"])
def test_counter_range_across_two_ranks_gloo(exchange, monkeypatch):
    """... and the same sums when the ranks sum the occupied 64-cell stretches of their counters instead of the whole
    buffers (dist.sum_counts_compact: what PeCounter.all_reduce does by default; "dense": turned off)."""
    import torch.multiprocessing as mp

    monkeypatch.setenv("VS_COMPACT_ALLREDUCE", "1" if exchange == "compact" else "0")
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + ((os.getpid() + (41 if exchange == "dense" else 97)) % 500)
    procs = [ctxm.Process(target=_rank_range, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    out, how = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (ta, fa, pa, ra), (tb, fb, pb, rb), (tc, fc, pc, rc), (td, fd, _, _) = out
    assert fa and pa == 2 ** 30 and ra[1][2, 2] == 2 ** 32 - 5 and ra[0][0, 3] == 15
    assert fb and pb == 0 and rb[1][2, 2] == 2 * (2 ** 32 - 1) and rb[0][0, 3] == 2 ** 32 + 1
    assert fc and rc[0][1, 1] == 5 + 5 + 3
    assert fd
    assert how == [exchange, exchange]


def _rank_exchange_shapes(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from vstrains_amd import dist as vdist

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    gen = torch.Generator().manual_seed(500 + rank)
    res = []
    # (1) occupancy from the dirty-tile map instead of a scan, staged in slabs much smaller than the union
    vdist.SLAB_STRETCHES = 7
    vdist._staging.clear()
    for n in (333, 600, 1000):
        T = (n + 63) // 64
        k = 5
        mi, ri, ci = torch.randint(0, 2, (k,), generator=gen), torch.randint(0, n, (k,), generator=gen), torch.randint(0, n, (k,), generator=gen)
        m = torch.zeros((2, n, n), dtype=torch.int32)
        m[mi, ri, ci] = torch.randint(1, 2 ** 31 - 1, (k,), generator=gen, dtype=torch.int64).to(torch.int32)
        tiles = torch.zeros(2 * T * T, dtype=torch.uint8)
        tiles[(mi * T + (ri >> 6)) * T + (ci >> 6)] = 1
        dense, compact = m.clone(), m.clone()
        dist.all_reduce(dense, op=dist.ReduceOp.SUM)
        timing = {}
        how = vdist.sum_counts_compact(compact, tile_map=tiles, timing=timing)
        assert torch.equal(dense, compact), n
        res.append((n, how, timing["occupied_stretches_of_the_union"] > vdist.SLAB_STRETCHES, sorted(k for k in timing if isinstance(timing[k], float))))
    # (2) one rank cannot get its staging buffer: every rank takes the dense ring, nobody is left in a collective
    vdist._staging.clear()
    plain = vdist._staging_buffer
    if rank == 1:
        vdist._staging_buffer = lambda head, rows: None
    m = torch.zeros((2, 100, 100), dtype=torch.int32)
    m[0, rank, 3] = 5 + rank
    dense = m.clone()
    dist.all_reduce(dense, op=dist.ReduceOp.SUM)
    how = vdist.sum_counts_compact(m)
    assert torch.equal(dense, m)
    res.append(("no staging buffer on rank 1", how))
    vdist._staging_buffer = plain
    assert vdist.sum_counts_compact(m.clone()) == "compact"  # (and the next call is compact again)
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_compact_exchange_from_the_tile_map_in_slabs_and_without_a_staging_buffer():
    """Round 5: the occupancy of a large counter comes from its dirty-tile map (no scan of the
    buffer), the occupied stretches go through the ring in bounded slabs (256 MB of staging whatever the union's size), and
    a rank that cannot allocate its staging buffer tells its peers before any of them enters the collective -- all take
    the dense ring.  Two gloo ranks; every sum equals the dense all-reduce cell for cell."""
    import torch.multiprocessing as mp

    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + ((os.getpid() + 211) % 500)
    procs = [ctxm.Process(target=_rank_exchange_shapes, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for n, how, many_slabs, phases in res[:3]:
        assert how == "compact" and many_slabs, (n, how)
        assert phases == ["gather", "nonzero", "occupancy", "occupancy_allreduce", "ring", "scatter"], phases
    assert res[3] == ("no staging buffer on rank 1", "dense")


def _rank_packed_eight(rank, world, port, q):
    """Eight gloo ranks through PeCounter.all_reduce / dist.sum_counts_packed (r6: two collectives per sum)."""
    import os

    import torch
    import torch.distributed as dist

    from vstrains_amd import dist as vdist

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    res = {}
    n = 97  # (2 * 97 * 97 cells: two cells are left behind the last whole stretch and ride in the staging tensor)

    def banded(seed, cells):
        gen = torch.Generator().manual_seed(seed)
        m = torch.zeros((2, n, n), dtype=torch.int32)
        if cells:
            i = torch.randint(0, n, (cells,), generator=gen)
            j = (i + torch.randint(0, 6, (cells,), generator=gen)).clamp(max=n - 1)
            m[torch.randint(0, 2, (cells,), generator=gen), i, j] = torch.randint(1, 1000, (cells,), generator=gen).to(torch.int32)
        return m

    def dense_sum(t):
        d = t.clone()
        dist.all_reduce(d, op=dist.ReduceOp.SUM)
        return d

    # (1) uneven shards, one rank with no pairs at all: compact, TWO collectives, stats and pairs ride in the second
    c = _cpu_counter(n, 0 if rank == 5 else 1000 * (rank + 1))
    c.mats.copy_(banded(100 + rank, 0 if rank == 5 else 4 * (rank + 1)))
    c.stats += torch.tensor([rank + 1, 2 * rank, 3], dtype=torch.int64)
    want = dense_sum(c.mats)
    c.all_reduce()
    res["uneven"] = (c.last_all_reduce, c.last_collectives, bool(torch.equal(c.mats, want)), c.stats.tolist(), c.pairs_in_buffer, c.wide is None)
    # (2) one rank vetoes the compact form: every rank takes the dense ring, same sums
    c = _cpu_counter(n, 10)
    c.mats.copy_(banded(200 + rank, 5))
    want = dense_sum(c.mats)
    if rank == 3:
        os.environ["VS_COMPACT_ALLREDUCE"] = "0"
    c.all_reduce()
    os.environ.pop("VS_COMPACT_ALLREDUCE", None)
    res["veto"] = (c.last_all_reduce, bool(torch.equal(c.mats, want)), c.pairs_in_buffer)
    # (3) the uint32 buffers cannot hold the sum of eight ranks (cells above 2^31 among them): all fold, int64 totals summed
    c = _cpu_counter(n, 2 ** 28 + rank)
    _set_cell(c, 1, 7, 7, 2 ** 32 - 1 - rank)
    _set_cell(c, 0, 2, 9, 2 ** 31 + rank)
    c.all_reduce()
    node, short, _ = c.result()
    res["fold"] = (c.wide is not None, c.pairs_in_buffer, int(short[7, 7]), int(node[2, 9]), c.last_all_reduce)
    # (3b) ... and they can when the bound says so: 2^25 pairs on the largest of eight ranks
    c = _cpu_counter(n, 2 ** 25 if rank == 2 else 17)
    _set_cell(c, 0, 1, 1, 2 ** 28 + rank)
    c.all_reduce()
    res["nofold"] = (c.wide is None, c.pairs_in_buffer, int(c.result()[0][1, 1]))
    # (4) only rank 0 needs the sums (the drop-in's writer): reduce instead of all-reduce
    c = _cpu_counter(n, 5)
    c.mats.copy_(banded(300 + rank, 6))
    mine = c.mats.clone()
    want = dense_sum(c.mats)
    c.all_reduce(dst=0)
    res["dst"] = bool(torch.equal(c.mats, want)) if rank == 0 else bool(torch.equal(c.mats, mine))
    # (5) the steady state of fixed-size steps: the first exchange learns the union's size, the following ones are PREDICTED
    # (no host wait, two collectives); a step whose union outgrows the prediction is completed by settle()
    c = _cpu_counter(n, 0)
    steps = []
    for step in range(5):
        c.settle()
        c.mats.zero_()
        c.stats.zero_()
        c.pairs_in_buffer = 100
        cells = 3 if step < 3 else 60 if step == 3 else 3  # (step 3: twenty times the cells)
        c.mats.copy_(banded(1000 * step + rank, cells))
        c.stats += 1
        want = dense_sum(c.mats)
        c.all_reduce(predict=True)
        waits_before = c._xstate.host_waits
        coll = c.last_collectives
        c.settle()
        steps.append((coll, waits_before, bool(torch.equal(c.mats, want)), c.stats.tolist(), c._xstate.collectives))
    res["steady"] = steps
    if rank in (0, 5):
        q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_packed_exchange_eight_ranks_gloo():
    """Round 6: PeCounter.all_reduce is TWO collectives -- MAX over [occupancy | flag bytes],
    SUM over [occupied stretches | stats row] -- on eight ranks with uneven shards, a rank without pairs, a rank that vetoes
    the compact form, counters beyond 2^31 that force the fold, a reduce to the writer only, and the predicted steady state
    (no host wait; a union that outgrows the prediction is repaired by settle())."""
    import torch.multiprocessing as mp

    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + ((os.getpid() + 333) % 500)
    world = 8
    procs = [ctxm.Process(target=_rank_packed_eight, args=(rk, world, port, q)) for rk in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 5):
        r = got[rank]
        how, coll, same, stats, pairs, narrow = r["uneven"]
        assert how == "compact" and coll == 2 and same and narrow
        assert stats == [sum(range(1, 9)), 2 * sum(range(8)), 24] and pairs == sum(1000 * (k + 1) for k in range(8) if k != 5)
        assert r["veto"] == ("dense", True, 80)
        wide, pairs, s77, n29, _ = r["fold"]
        assert wide and pairs == 0 and s77 == sum(2 ** 32 - 1 - k for k in range(8)) and n29 == sum(2 ** 31 + k for k in range(8))
        assert r["nofold"] == (True, 2 ** 25 + 7 * 17, 8 * 2 ** 28 + 28)
        assert r["dst"]
        st = r["steady"]
        assert st[0][0] == 2 and st[0][1] == 1                    # the first exchange asks the device once
        for coll, waits, same, stats, coll_after in st[1:]:
            assert coll == 2 and waits == 0 and same and stats == [8, 8, 8]
        assert st[0][2] and st[3][4] > 2                          # step 3 outgrew the prediction: settle() summed the rest
        assert st[4][4] == 2


def test_async_allreduce_without_process_group_is_a_no_op():
    import torch

    from vstrains_amd import dist as vdist

    assert vdist.all_reduce_counts_async(torch.zeros((2, 2, 2), dtype=torch.int32), torch.zeros(3, dtype=torch.int64)) == []


def test_shard_range_covers_everything_once():
    from vstrains_amd import dist as vdist

    for n in (0, 1, 7, 100, 10 ** 7 + 3):
        for world in (1, 2, 3, 8):
            spans = [vdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]


@pytest.mark.parametrize("name,d,meta", pe_cases(ok_only=False), ids=[c[0] for c in pe_cases(ok_only=False)])
def test_native_fastq_ingest_matches_reference_semantics(name, d, meta):
    """vs_fastq_* (C++, multi-threaded, host only) against the oracle's restatement of
    readlines() + line[:-1] (PE_Inference.py:146-159): CRLF, missing final newline, unequal files."""
    from vstrains_amd import pe as host

    # (the ingest delivers one byte per CHARACTER; a character outside ASCII -- case utf8_reads_k21 -- arrives as '?')
    wf = [s.encode("ascii", "replace").decode() for s in pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq"))]
    wr = [s.encode("ascii", "replace").decode() for s in pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))]
    for threads in ("1", "3", "8"):
        os.environ["VS_HOST_THREADS"] = threads
        try:
            fq = host.FastqPair(os.path.join(d, "fwd.fq"), os.path.join(d, "rve.fq"))
        finally:
            os.environ.pop("VS_HOST_THREADS")
        n = min(len(wf), len(wr))
        assert len(fq) == n
        for p in range(n):
            assert fq.sequence(0, p) == wf[p]
            assert fq.sequence(1, p) == wr[p]
        if n:
            data, off = fq.gather(0, n)
            for p in range(n):
                assert bytes(data[int(off[2 * p]):int(off[2 * p + 1])]).decode("latin-1") == wf[p]
                assert bytes(data[int(off[2 * p + 1]):int(off[2 * p + 2])]).decode("latin-1") == wr[p]
        fq.close()


def test_native_fastq_ingest_odd_inputs(tmp_path):
    from vstrains_amd import pe as host

    cases = {
        "empty": b"",
        "one_line": b"@r",
        "lone_cr": b"@a\rACGT\r+\rIIII\r@b\rGG\r+\rII",
        "crlf_tail": b"@a\r\nACGT\r\n+\r\nIIII\r\n@b\r\nGGTT\r\n+\r\nIIII",
        "blank_lines": b"\n\n\n\n@a\n\n+\n\n",
        "three_lines": b"@a\nACGT\n+\n",
    }
    for name, raw in cases.items():
        f = tmp_path / (name + "_f.fq")
        f.write_bytes(raw)
        want = pe_oracle.fastq_sequences(str(f))
        fq = host.FastqPair(str(f), str(f))
        assert len(fq) == len(want), name
        assert [fq.sequence(0, i) for i in range(len(want))] == want, name
        fq.close()
    # gzip input: one member, several members, CRLF inside, an empty member; a cut-off stream is refused
    import gzip

    raw = b"".join(b"@r%d\nACGT%s\n+\nIIII\n" % (i, b"ACGT" * (i % 7)) for i in range(3000)) + b"@x\r\nGGA\r\n+\r\nIII"
    plain = tmp_path / "plain.fq"
    plain.write_bytes(raw)
    want = pe_oracle.fastq_sequences(str(plain))
    members = {"one": [raw], "three": [raw[:1000], raw[1000:50000], raw[50000:]], "empty_first": [b"", raw]}
    for name, parts in members.items():
        gz = tmp_path / (name + ".fq.gz")
        gz.write_bytes(b"".join(gzip.compress(x) for x in parts))
        for threads in ("1", "5"):
            os.environ["VS_HOST_THREADS"] = threads
            try:
                fq = host.FastqPair(str(gz), str(plain))
            finally:
                os.environ.pop("VS_HOST_THREADS")
            assert len(fq) == len(want), name
            assert [fq.sequence(0, i) for i in range(len(want))] == want, name
            assert [fq.sequence(1, i) for i in range(len(want))] == want, name
            fq.close()
    cut = tmp_path / "cut.fq.gz"
    cut.write_bytes(gzip.compress(raw)[:-30])
    with pytest.raises(Exception) as ei:
        host.FastqPair(str(cut), str(plain))
    assert "gzip" in str(ei.value)
    # bytes >= 0x80: the reference reads in text mode, so valid UTF-8 is fine anywhere -- in a sequence line a
    # multi-byte character is ONE character of the read (delivered as '?'); invalid UTF-8 in a sequence line is
    # ValueError (the reference's readlines() raises UnicodeDecodeError, a ValueError)
    ok = tmp_path / "utf8_header.fq"
    ok.write_bytes("@r\u00e9ad 1\nACGT\n+\nII\u00e9I\n".encode("utf-8"))
    fq = host.FastqPair(str(ok), str(ok))
    assert len(fq) == 1 and fq.sequence(0, 0) == "ACGT"
    fq.close()
    seq = tmp_path / "utf8_seq.fq"
    seq.write_bytes("@a\nAC\u00e9T\n+\nIIII\n@b\n\u20acACG\U0001F9ECT\n+\nIIIIII\n@c\nACGT\u00e9\n+\nII\u00e9".encode("utf-8"))
    fq = host.FastqPair(str(seq), str(seq))
    assert [fq.sequence(0, i) for i in range(3)] == ["AC?T", "?ACG?T", "ACGT?"]
    data, off = fq.gather(0, 3)
    assert bytes(data[: int(off[-1])]) == b"AC?TAC?T?ACG?T?ACG?TACGT?ACGT?" and [int(x) for x in off] == [0, 4, 8, 14, 20, 25, 30]
    fq.close()
    for raw in (b"@a\nAC\xc3T\n+\nIIII\n", b"@a\nAC\xa9T\n+\nIIII\n", b"@a\nAC\xed\xa0\x80T\n+\nI\n", b"@a\nAC\xc0\xafT\n+\nI\n",
                b"@a\nAC\xf4\x90\x80\x80T\n+\nI\n"):  # cut short, stray continuation, surrogate, overlong, > U+10FFFF
        bad = tmp_path / "bad.fq"
        bad.write_bytes(raw)
        with pytest.raises(UnicodeDecodeError):
            raw.decode("utf-8")
        with pytest.raises(ValueError):  # at open, as the reference fails in readlines() before it counts anything
            host.FastqPair(str(bad), str(bad))
    # ... and anywhere in the file, not only in sequence lines: a header, a quality line, the line past the last whole record;
    # a character cut in two by the worker threads' ranges (a long file of two-byte characters) is not an error
    good = tmp_path / "many_utf8.fq"
    good.write_bytes(("@h\u00e9\nACGT\n+\n\u00e9\u00e9\u00e9\u00e9\n" * 5000).encode("utf-8"))
    fq = host.FastqPair(str(good), str(good))
    assert len(fq) == 5000 and fq.sequence(1, 4999) == "ACGT"
    fq.close()
    import ctypes as C

    from vstrains_amd import _native as nat

    for raw in (b"@a\xff\nACGT\n+\nIIII\n", b"@a\nACGT\n+\nII\xc3I\n", b"@a\nACGT\n+\nIIII\n@b\xa9\n", b"@a\nACGT\n+\nIIII\n" * 3000 + b"\xed\xa0\x80"):
        bad = tmp_path / "bad2.fq"
        bad.write_bytes(raw)
        with pytest.raises(ValueError):
            host.FastqPair(str(bad), str(good))
        out = (C.c_uint64 * 3)()
        assert nat.lib().vs_fastq_count_part(str(bad).encode(), 0, 1, out) == nat.VS_E_UTF8  # (the sharded open: every rank its byte range)
        assert any(nat.lib().vs_fastq_count_part(str(bad).encode(), r, 3, out) == nat.VS_E_UTF8 for r in range(3))
    out = (C.c_uint64 * 3)()
    assert all(nat.lib().vs_fastq_count_part(str(good).encode(), r, 7, out) == nat.VS_OK for r in range(7))
    with pytest.raises(FileNotFoundError):
        host.FastqPair(str(tmp_path / "nope.fq"), str(bad))


def test_vector_host_packer_equals_the_byte_loop():
    """vs_pack_sequence: the SSE2 / AVX2 body against the byte-by-byte one and against a Python
    statement of the format (16 bases per word, LSB first, ACGT = 0123, other bytes 0 + flag)."""
    import ctypes as C

    import numpy as np

    from vstrains_amd import _native as nat

    lib = nat.lib()
    rng = np.random.default_rng(5)
    code = {65: 0, 67: 1, 71: 2, 84: 3}

    def want(seq):
        words = [0] * ((len(seq) + 15) // 16)
        flags = 0
        for i, c in enumerate(seq):
            if c in code:
                words[i // 16] |= code[c] << (2 * (i % 16))
            elif c == 78:
                flags |= 1
            elif c < 128:
                flags |= 2
            else:
                flags |= 0x80
        return words, flags

    def run(seq, plain):
        buf = np.frombuffer(bytes(seq) + b"\x00", dtype=np.uint8).copy()
        out = np.full((len(seq) + 15) // 16 + 1, 0xDEADBEEF, dtype=np.uint32)
        fl = C.c_uint32(0)
        assert lib.vs_pack_sequence(buf.ctypes.data, len(seq), out.ctypes.data, C.byref(fl), plain) == 0
        assert out[-1] == 0xDEADBEEF  # nothing written past the last word
        return out[:-1].tolist(), fl.value

    cases = [b"", b"A", b"ACGT" * 4, b"ACGT" * 8, b"T" * 33, b"ACGTN" * 13, b"acgt" * 9, b"ACGT" * 7 + b"\xc3\xa9"]
    for n in list(range(0, 70)) + [149, 150, 151, 250, 1000, 1023]:
        cases.append(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)))
    for n in (15, 16, 17, 31, 32, 33, 47, 48, 64, 150):
        for _ in range(20):
            s = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
            for _ in range(int(rng.integers(1, 4))):
                s[int(rng.integers(0, n))] = int(rng.choice(np.frombuffer(b"NnRY@\x7f\x80\xff", dtype=np.uint8)))
            cases.append(bytes(s))
    for seq in cases:
        w = want(seq)
        assert run(seq, 0) == w, seq
        assert run(seq, 1) == w, seq


def _child_reads_fastq(path, q):
    from vstrains_amd import pe as host

    fq = host.FastqPair(path, path)
    q.put([fq.sequence(0, i) for i in range(len(fq))])
    fq.close()


def test_fastq_ingest_after_fork(tmp_path):
    """The ingest keeps worker threads across calls; a forked child has none of them and must start
    its own instead of waiting for the parent's (pthread_atfork in csrc/vs_fastq.hip)."""
    import multiprocessing as mp

    from vstrains_amd import pe as host

    raw = b"".join(b"@r%d\nACGT%s\n+\nIIII\n" % (i, b"GATTACA" * (i % 5)) for i in range(4000))
    path = tmp_path / "a.fq"
    path.write_bytes(raw)
    want = pe_oracle.fastq_sequences(str(path))
    os.environ["VS_HOST_THREADS"] = "6"
    try:
        fq = host.FastqPair(str(path), str(path))  # (the pool exists from here on)
        assert [fq.sequence(0, i) for i in range(len(fq))] == want
        ctx = mp.get_context("fork")
        q = ctx.Queue()
        p = ctx.Process(target=_child_reads_fastq, args=(str(path), q))
        p.start()
        got = q.get(timeout=120)
        p.join(timeout=60)
        assert p.exitcode == 0 and got == want
        fq2 = host.FastqPair(str(path), str(path))  # (and the parent's pool still works)
        assert [fq2.sequence(1, i) for i in range(len(fq2))] == want
        fq.close()
        fq2.close()
    finally:
        os.environ.pop("VS_HOST_THREADS")


# ---- cooperative FASTQ open: one rank per GPU, nobody reads a whole file -------------------------------------------------
def _open_all_shards(host, fwd, rve, world):
    """Every rank's FastqPair.open_shard in one process: the exchange of the counts is a lookup in what all ranks counted."""
    import ctypes as C

    from vstrains_amd import _native as nat

    per_rank = []
    for rk in range(world):
        vals = []
        for path in (fwd, rve):
            out = (C.c_uint64 * 3)()
            assert nat.lib().vs_fastq_count_part(path.encode(), rk, world, out) == 0
            vals += [int(out[0]), int(out[1]), int(out[2])]
        per_rank.append(vals + [0])  # (seventh integer: this rank's failure flag)
    # (the second exchange of open_shard is one status integer per rank)
    return [host.FastqPair.open_shard(fwd, rve, None, rk, world, all_gather=lambda mine: per_rank if len(mine) > 1 else [[0]] * world)
            for rk in range(world)]


def test_a_rank_that_fails_in_the_whole_file_open_fails_every_rank(tmp_path, monkeypatch):
    """ Files with carriage returns (or gzip) are opened whole by every rank; a rank that fails alone there
    reports it through the one-integer status exchange, so that its peers raise too instead of entering the next
    collective without it."""
    from vstrains_amd import pe as host

    fwd, rve = str(tmp_path / "f.fq"), str(tmp_path / "r.fq")
    for path in (fwd, rve):
        with open(path, "wb") as fh:
            fh.write(b"@a\r\nACGT\r\n+\r\nIIII\r\n" * 40)
    calls = {0: [], 1: []}

    def gather_for(rank, peer_status):
        def all_gather(mine):
            calls[rank].append(list(mine))
            if len(mine) > 1:
                return [mine, mine]  # (both ranks count the same small files: any flags agree)
            return [mine, [peer_status]] if rank == 0 else [[peer_status], mine]
        return all_gather

    # rank 0 opens fine, its peer reports a failure: rank 0 raises and says who
    with pytest.raises(RuntimeError, match=r"rank\(s\) \[1\]"):
        host.FastqPair.open_shard(fwd, rve, None, 0, 2, all_gather=gather_for(0, 1))
    assert [len(c) for c in calls[0]] == [7, 1] and calls[0][1] == [0]
    # rank 1 fails in the open itself: it still takes part in the status exchange, then raises its own error
    real_init = host.FastqPair.__init__

    def failing_init(self, *a, **kw):
        raise MemoryError("inflating the file")

    monkeypatch.setattr(host.FastqPair, "__init__", failing_init)
    with pytest.raises(MemoryError):
        host.FastqPair.open_shard(fwd, rve, None, 1, 2, all_gather=gather_for(1, 0))
    assert [len(c) for c in calls[1]] == [7, 1] and calls[1][1] == [1]
    monkeypatch.setattr(host.FastqPair, "__init__", real_init)
    # and with nobody failing the block comes back
    fq = host.FastqPair.open_shard(fwd, rve, None, 0, 2, all_gather=gather_for(0, 0))
    assert fq.whole and fq.total_pairs == 40 and fq.n_pairs == 20


def test_a_rank_that_cannot_open_its_files_fails_every_rank(tmp_path):
    """FastqPair.open_shard: the failure of one rank (a file it cannot open) travels with the exchanged counts, so its
    peers raise instead of waiting in the next collective."""
    from vstrains_amd import pe as host

    fwd, rve = str(tmp_path / "f.fq"), str(tmp_path / "r.fq")
    for path in (fwd, rve):
        with open(path, "w") as fh:
            fh.write("@a\nACGT\n+\nIIII\n" * 40)
    seen = {}

    def gather_for(rank):
        def all_gather(mine):
            seen[rank] = list(mine)
            other = [10, 80, 0, 10, 80, 0, 1]  # rank 1 reports that it could not open a file
            return [mine, other] if rank == 0 else [other, mine]
        return all_gather

    with pytest.raises(RuntimeError, match="rank"):
        host.FastqPair.open_shard(fwd, rve, None, 0, 2, all_gather=gather_for(0))
    assert seen[0][6] == 0
    with pytest.raises(FileNotFoundError):
        host.FastqPair.open_shard(str(tmp_path / "missing.fq"), rve, None, 1, 2, all_gather=gather_for(1))
    assert seen[1][6] == 1  # (it still took part in the exchange before raising)


def test_cooperative_fastq_open_gives_every_rank_its_block_and_only_its_bytes(tmp_path):
    """``FastqPair.open_shard``: the ranks' blocks, one after the other, are the records the whole-file open gives
    (PE_Inference.py:146-159 incl. total = min(lines // 4)), and a rank goes through about 1 / world of the text."""
    from vstrains_amd import pe as host
    from vstrains_amd import synth

    rng = np.random.default_rng(12)
    st = synth.make_strains(3, 3000, 0.01, seed=5)
    f, r = synth.sample_pairs(st, 2000, 90, seed=6)
    f = [s[: int(rng.integers(1, 91))] for s in f]   # ragged: byte ranges do not line up with records
    r = [s[: int(rng.integers(0, 91))] for s in r]
    texts = {
        "plain": (synth.fastq_text(f, "f"), synth.fastq_text(r, "r")),
        "unequal": (synth.fastq_text(f, "f"), synth.fastq_text(r[:1500], "r") + "@partial\nACGT\n"),
        "open_tail": (synth.fastq_text(f, "f")[:-1], synth.fastq_text(r, "r")[:-1]),
        "tiny": (synth.fastq_text(f[:3], "f"), synth.fastq_text(r[:3], "r")),
        "empty": ("", ""),
    }
    for name, (ft, rt) in texts.items():
        fp, rp = str(tmp_path / (name + "_f.fq")), str(tmp_path / (name + "_r.fq"))
        open(fp, "w").write(ft)
        open(rp, "w").write(rt)
        whole = host.FastqPair(fp, rp)
        want = [(whole.sequence(0, i), whole.sequence(1, i)) for i in range(len(whole))]
        whole_bytes = whole.bytes_indexed
        whole.close()
        for world in (1, 2, 3, 7):
            shards = _open_all_shards(host, fp, rp, world)
            got = []
            for rk, fq in enumerate(shards):
                assert fq.total_pairs == len(want) and not fq.whole, (name, world)
                assert fq.first == len(got)
                got += [(fq.sequence(0, i), fq.sequence(1, i)) for i in range(len(fq))]
                if name == "plain" and world > 1:
                    assert fq.bytes_indexed <= whole_bytes / world * 1.15 + 1000, (world, rk, fq.bytes_indexed, whole_bytes)
                fq.close()
            assert got == want, (name, world)
    # carriage returns / gzip: every rank opens the files whole and takes its block
    import gzip

    fp, rp = str(tmp_path / "crlf_f.fq"), str(tmp_path / "gz_r.fq.gz")
    open(fp, "w", newline="").write(synth.fastq_text(f[:200], "f", "\r\n"))
    open(rp, "wb").write(gzip.compress(synth.fastq_text(r[:200], "r").encode()))
    whole = host.FastqPair(fp, rp)
    want = [(whole.sequence(0, i), whole.sequence(1, i)) for i in range(len(whole))]
    whole.close()
    got = []
    for fq in _open_all_shards(host, fp, rp, 3):
        assert fq.whole and fq.total_pairs == 200
        got += [(fq.sequence(0, fq.block_offset + i), fq.sequence(1, fq.block_offset + i)) for i in range(len(fq))]
        fq.close()
    assert got == want


def _rank_open_shard(rank, world, port, fwd, rve, q):
    import torch.distributed as dist

    from vstrains_amd import pe as host

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    fq = host.FastqPair.open_shard(fwd, rve, None, rank, world)  # (counts exchanged through torch.distributed.all_gather)
    q.put((rank, fq.first, fq.total_pairs, fq.bytes_indexed, [fq.sequence(0, i) + "|" + fq.sequence(1, i) for i in range(len(fq))]))
    dist.barrier()
    dist.destroy_process_group()


def test_cooperative_fastq_open_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp

    from vstrains_amd import pe as host
    from vstrains_amd import synth

    st = synth.make_strains(3, 3000, 0.01, seed=15)
    f, r = synth.sample_pairs(st, 3001, 100, seed=16)
    fp, rp = str(tmp_path / "f.fq"), str(tmp_path / "r.fq")
    open(fp, "w").write(synth.fastq_text(f, "f"))
    open(rp, "w").write(synth.fastq_text(r, "r"))
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctxm.Process(target=_rank_open_shard, args=(rk, 2, port, fp, rp, q)) for rk in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total_bytes = os.path.getsize(fp) + os.path.getsize(rp)
    assert [x[1] for x in res] == [0, 1500] and all(x[2] == 3001 for x in res)
    assert all(x[3] <= total_bytes * 0.51 for x in res), [x[3] for x in res]  # each rank went through about half the text
    assert res[0][4] + res[1][4] == [a + "|" + b for a, b in zip(f, r)]


def test_node_order_runs_along_paths_and_equals_its_python_statement():
    """vs_node_order_host (csrc/vs_order_host.cpp) against vstrains_amd/node_order.py:path_order: a permutation, the same
    one, on the bench graph of configs[1], on the same nodes shuffled / partly reverse-complemented / with odd members
    (empty, shorter than k, bytes outside ACGT), and along a chain cut into shuffled pieces."""
    import random
    import tempfile

    from vstrains_amd import pe as host
    from vstrains_amd.node_order import _revcomp, path_order
    from vstrains_amd.workloads import CONFIGS, workload_for

    st, pre, names, seqs, cum, logger, n_in = workload_for(1, tempfile.mkdtemp(prefix="order_"))
    k = CONFIGS[1]["k"]
    rnd = random.Random(11)
    mixed = list(seqs)
    rnd.shuffle(mixed)
    mixed = [s if rnd.random() < 0.5 else _revcomp(s) for s in mixed] + ["AC", "", "N" * 80, "acgt" * 20]
    for nodes in (seqs, mixed):
        got = host.node_order(nodes, k)
        assert sorted(got.tolist()) == list(range(len(nodes)))
        assert got.tolist() == path_order(nodes, k)
    # neighbours along a path are neighbours in the numbering: the nodes under a 150-base window of a strain
    home = {}
    K = k + 1
    for i, s in enumerate(mixed):
        for strand in (s, _revcomp(s)):
            for o in range(len(strand) - K + 1):
                home[strand[o:o + K]] = i
    rank = np.empty(len(mixed), dtype=np.int64)
    rank[host.node_order(mixed, k)] = np.arange(len(mixed))
    spans_given, spans_ordered = [], []
    for g in st.genomes[:4]:
        for p in range(0, len(g) - 150, 97):
            ns = {home.get(g[q:q + K]) for q in range(p, p + 150 - K + 1)} - {None}
            if len(ns) > 1:
                ids = np.array(sorted(ns))
                spans_given.append(ids.max() - ids.min())
                spans_ordered.append(rank[ids].max() - rank[ids].min())
    assert np.median(spans_ordered) * 4 < np.median(spans_given), (np.median(spans_ordered), np.median(spans_given))
    # a chain in shuffled pieces, mixed strands: one walk forwards from where it starts, then the rest backwards
    g = "".join(rnd.choice("ACGT") for _ in range(1500))
    cuts = sorted(rnd.sample(range(40, len(g) - 40), 30))
    pieces, a = [], 0
    for c in cuts + [len(g)]:
        pieces.append(g[a:c + 21] if c < len(g) else g[a:])
        a = c
    idx = list(range(len(pieces)))
    rnd.shuffle(idx)
    chain = [pieces[i] if rnd.random() < 0.5 else _revcomp(pieces[i]) for i in idx]
    walk = [idx[o] for o in host.node_order(chain, 21).tolist()]
    first, m = walk[0], len(pieces)
    assert walk in (list(range(first, m)) + list(range(first - 1, -1, -1)), list(range(first, -1, -1)) + list(range(first + 1, m))), walk
    assert host.node_order([], 5).tolist() == [] and host.node_order(["ACGT"], 0).tolist() == [0]


def test_counter_matrices_come_back_in_the_callers_numbering():
    """PeCounter.user_order / result with an index built in another numbering (Context.build_index renumber): node_mat is
    permuted on both axes; short_mat holds a pair of nodes at (smaller, larger) INTERNAL number and must come back at
    (smaller, larger) position of the caller's list (PE_Inference.py:174-184), the diagonal untouched; folded int64
    totals take the same way; Context.internal_cells names the same cells from the other side."""
    import torch

    from vstrains_amd import pe as host

    n = 7
    rng = np.random.default_rng(4)
    order = rng.permutation(n)            # order[internal] = caller position
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n)
    c = _cpu_counter(n, 0)
    c.ctx.node_order, c.ctx.node_rank = order, rank
    c.node_order, c.node_rank = order, rank  # (what PeCounter.__init__ takes from a context whose index is numbered so)
    c.ctx.internal_cells = host.Context.internal_cells.__get__(c.ctx)
    # what the caller's numbering should show
    want_node = rng.integers(0, 50, size=(n, n))
    want_short = np.triu(rng.integers(0, 50, size=(n, n)))
    want_node[1, 2] = 2 ** 31 + 7         # (uint32 cells above 2^31 stay positive on the way)
    for u in range(n):
        for v in range(n):
            _set_cell(c, 0, rank[u], rank[v], int(want_node[u, v]))
            if u <= v:
                a, b = sorted((rank[u], rank[v]))
                _set_cell(c, 1, a, b, int(want_short[u, v]))
    node, short, _ = c.result()
    assert np.array_equal(node, want_node) and np.array_equal(short, want_short)
    uu, vv = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    for mat, want in ((0, want_node), (1, want_short)):
        flat = c.mats[mat].reshape(-1).numpy().view(np.uint32).astype(np.int64)
        sel = uu <= vv if mat else np.ones_like(uu, dtype=bool)
        assert np.array_equal(flat[c.ctx.internal_cells(mat, uu[sel], vv[sel])], want[sel])
    c.fold()
    _set_cell(c, 1, *sorted((rank[0], rank[3])), 5)
    want_short[0, 3] += 5
    node, short, _ = c.result()
    assert np.array_equal(node, want_node) and np.array_equal(short, want_short)
