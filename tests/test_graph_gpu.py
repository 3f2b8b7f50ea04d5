"""GPU parity of the graph-stage kernels (through the C ABI) and of the whole vstrains-compatible
pipeline: device ops vs the checker in oracle/graph_ops.py on the same inputs, and the full CLI
(PE-link inference on the device + graph stages) vs the reference's golden outputs."""
import os
import random

import numpy as np
import pytest

from graph_case import Case, case_names, compare, quiet_logger
from oracle import graph_ops as chk
from vstrains_amd.graph.asm_graph import AsmGraph
from vstrains_amd.graph.formats import read_stage_gfa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    from vstrains_amd.graph.hip_ops import HipBackend

    return HipBackend(0)


def random_graph(rng, nv, ne, gray_frac=0.0, hub=0):
    g = AsmGraph()
    for i in range(nv):
        g.add_vertex(str(i), rng.uniform(0.5, 2000.0), "ACGT", rng.random() >= gray_frac)
    seen = set()
    tries = 0
    while g.num_edges() < ne and tries < 50 * ne:
        tries += 1
        s, t = rng.randrange(nv), rng.randrange(nv)
        if hub and rng.random() < 0.3:
            s = 0
        if hub and rng.random() < 0.3:
            t = 1
        if (s, t) in seen:
            continue
        seen.add((s, t))
        g.add_edge(s, t, 21, 0.0, rng.random() >= gray_frac)
    return g


def assert_same_ops(ops, g):
    ref = chk.NumpyGraphOps()
    want_scan = ref.scan(g)
    got_scan = ops.scan(g)
    assert got_scan.nontrivial == want_scan.nontrivial
    assert got_scan.fork_kind == want_scan.fork_kind
    assert got_scan.chain_next == want_scan.chain_next
    assert got_scan.chain_rank == want_scan.chain_rank
    for v in range(g.num_vertices()):
        if want_scan.chain_rank[v] >= 0:
            assert got_scan.chain_top[v] == want_scan.chain_top[v], v
    h = AsmGraph.__new__(AsmGraph)
    for slot in AsmGraph.__slots__:
        setattr(h, slot, getattr(g, slot))
    h.eflow = list(g.eflow)
    ref.edge_flows(h)
    ops.edge_flows(g)
    for e in g.edges():
        assert g.eflow[e] == h.eflow[e], (e, g.eflow[e], h.eflow[e])  # bit-exact fp64


@pytest.mark.parametrize("nv,ne,gray,hub", [(1, 0, 0.0, 0), (5, 4, 0.0, 0), (40, 60, 0.0, 0), (300, 420, 0.2, 0),
                                            (2000, 2600, 0.05, 0), (400, 1500, 0.0, 1), (3000, 2999, 0.0, 0)])
def test_refresh_kernels_match_numpy_checker(backend, nv, ne, gray, hub):
    rng = random.Random(nv * 7 + ne)
    if ne == nv - 1 and nv > 100:  # one long simple chain plus a ring: list ranking depth
        g = AsmGraph()
        for i in range(nv):
            g.add_vertex(str(i), rng.uniform(1, 100), "A", True)
        order = list(range(nv - 50))
        rng.shuffle(order)
        for a, b in zip(order, order[1:]):
            g.add_edge(a, b, 21, 0.0, True)
        ring = list(range(nv - 50, nv))
        for a, b in zip(ring, ring[1:] + ring[:1]):
            g.add_edge(a, b, 21, 0.0, True)
    else:
        g = random_graph(rng, nv, ne, gray, hub)
    assert_same_ops(backend.graph_ops, g)


def test_flow_sum_order_for_large_degrees(backend):
    """numpy's pairwise summation changes shape at 8 and 128 addends; hubs of every size class."""
    rng = random.Random(5)
    g = AsmGraph()
    for i in range(1200):
        g.add_vertex(str(i), rng.uniform(0.001, 5000.0), "A", True)
    v = 10
    for hub, deg in ((0, 7), (1, 8), (2, 9), (3, 127), (4, 128), (5, 129), (6, 300), (7, 1000)):
        for k in range(deg):
            g.add_edge(hub, 10 + (v % 1100), 21, 0.0, True)
            g.add_edge(10 + ((v * 7) % 1100), hub, 21, 0.0, True)
            v += 1
    assert_same_ops(backend.graph_ops, g)


def test_zero_neighbour_sum_raises_like_numpy_seterr(backend):
    g = AsmGraph()
    a = g.add_vertex("a", 5.0, "A", True)
    b = g.add_vertex("b", 0.0, "A", True)
    g.add_edge(a, b, 21, 0.0, True)
    with pytest.raises(FloatingPointError):
        backend.graph_ops.edge_flows(g)


@pytest.mark.parametrize("n", [1, 3, 33, 70, 257])
def test_link_table_matches_process_pe_info(backend, n):
    from vstrains_amd.graph.hip_ops import HipPeLinks

    rng = np.random.default_rng(n)
    names = [str(i) for i in range(n)]
    node = rng.integers(0, 50, size=(n, n)) * (rng.random((n, n)) < 0.3)
    short = np.triu(rng.integers(0, 90, size=(n, n)) * (rng.random((n, n)) < 0.4))
    dev = HipPeLinks.from_matrices(backend.ctx, names, node, short)
    ref = chk.DictPeLinks(names, node, short)
    p0 = dev.to_numpy()
    for i in range(n):
        for j in range(n):
            a, b = names[i], names[j]
            assert p0[i, j] == ref.table[(min(a, b), max(a, b))]
    py = random.Random(n)
    queries = []
    for _ in range(200):
        rows = [py.randrange(n) for _ in range(py.randrange(0, 6))]
        cols = [py.randrange(n) for _ in range(py.choice([0, 1, 2, 5, 70, 130]))]
        queries.append((rows, cols))
    assert dev.block_sums(queries) == ref.block_sums(queries)
    groups = [[py.randrange(n) for _ in range(py.randrange(0, 5))] for _ in range(min(n, 40))]
    assert np.array_equal(dev.group_matrix(groups), ref.group_matrix(groups))


@pytest.mark.parametrize("n", [64, 65, 200, 517])
def test_sparse_link_table_from_the_dirty_tiles_equals_the_dense_one(backend, n):
    """(r6, ABI 10) vs_links_from_counts_tracked: counters that keep a dirty-tile map give the table as CSR rows of its
    non-zero cells, built from the marked 64 x 64 tiles only (the form graphs of 32 768 nodes and more take: 0.4 GB instead
    of 23.7 GB at configs[4]).  Same cells as the dense table of the same counters and as process_pe_info's dict
    (IO.py:598-627), same block sums and group matrices; cells outside the marked tiles are never read -- a tile that is
    NOT marked may hold anything."""
    import torch

    from vstrains_amd import pe as host
    from vstrains_amd.graph.hip_ops import HipPeLinks

    ctx = backend.ctx
    rng = np.random.default_rng(1000 + n)
    names = [str(i) for i in range(n)]

    seqs = ["ACGT" * 20 + "".join("ACGT"[(i >> (2 * j)) & 3] for j in range(8)) for i in range(n)]
    ctx.build_index(seqs, 21)  # (a counter takes its size from the context's index)
    counter = host.PeCounter(ctx, track_tiles=True)
    T = (n + 63) // 64
    node = rng.integers(1, 2 ** 31, size=(n, n)) * (rng.random((n, n)) < 0.05)
    short = np.triu(rng.integers(1, 2 ** 32 - 1, size=(n, n)) * (rng.random((n, n)) < 0.05))
    # whole tiles without any cell: left unmarked, and filled with garbage the build must not look at
    tiles = np.zeros((2, T, T), dtype=np.uint8)
    for m, mat in enumerate((node, short)):
        for I in range(T):
            for J in range(T):
                if rng.random() < 0.3:
                    mat[I * 64:(I + 1) * 64, J * 64:(J + 1) * 64] = 0
                tiles[m, I, J] = 1 if mat[I * 64:(I + 1) * 64, J * 64:(J + 1) * 64].any() else 0
    dirty_node, dirty_short = node.copy(), short.copy()
    for m, mat in enumerate((dirty_node, dirty_short)):
        for I in range(T):
            for J in range(T):
                if not tiles[m, I, J]:
                    mat[I * 64:(I + 1) * 64, J * 64:(J + 1) * 64] = 12345
    order = counter.node_order
    counter.node_order = None  # (this test speaks the counters' own numbering)
    counter.node_rank = None
    counter.mats[0].copy_(torch.from_numpy(dirty_node.astype(np.uint32).view(np.int32)))
    counter.mats[1].copy_(torch.from_numpy(dirty_short.astype(np.uint32).view(np.int32)))
    counter.tile_map.copy_(torch.from_numpy(tiles.reshape(-1)))
    sparse = HipPeLinks.from_counter(ctx, counter, names, sparse_min_nodes=64)
    ref = chk.DictPeLinks(names, node, short)
    want = node + node.T + short + short.T
    want[np.arange(n), np.arange(n)] = node.diagonal() + short.diagonal()
    assert np.array_equal(sparse.to_numpy(), want)
    py = random.Random(n)
    queries = []
    for _ in range(300):
        rows = [py.randrange(n) for _ in range(py.randrange(0, 6))]
        cols = [py.randrange(n) for _ in range(py.choice([0, 1, 2, 5, 70, 130]))]
        queries.append((rows, cols))
    assert sparse.block_sums(queries) == ref.block_sums(queries)
    groups = [[py.randrange(n) for _ in range(py.randrange(0, 5))] for _ in range(20)]
    assert np.array_equal(sparse.group_matrix(groups), ref.group_matrix(groups))
    # ... and below the threshold the same call gives the dense table of the counters as they are (garbage and all)
    dense = HipPeLinks.from_counter(ctx, counter, names, sparse_min_nodes=10 ** 6)
    wd = dirty_node + dirty_node.T + dirty_short + dirty_short.T
    wd[np.arange(n), np.arange(n)] = dirty_node.diagonal() + dirty_short.diagonal()
    assert np.array_equal(dense.to_numpy(), wd)
    counter.node_order = order


def test_link_table_takes_a_reserved_buffer_of_its_size_and_only_that(backend):
    """vs_links_reserve (ABI 9) sets the table's buffer aside ahead of time: a table of that size takes it, another size is
    allocated as before, a second reservation replaces the first, n = 0 gives it back -- the table is the same either way."""
    import ctypes as C

    from vstrains_amd import _native as nat
    from vstrains_amd.graph.hip_ops import HipPeLinks

    ctx = backend.ctx
    rng = np.random.default_rng(11)

    def table(n):
        names = [str(i) for i in range(n)]
        node = rng.integers(0, 50, size=(n, n)) * (rng.random((n, n)) < 0.3)
        short = np.triu(rng.integers(0, 90, size=(n, n)) * (rng.random((n, n)) < 0.4))
        want = node + node.T + short + short.T
        want[np.arange(n), np.arange(n)] = node.diagonal() + short.diagonal()
        dev = HipPeLinks.from_matrices(ctx, names, node, short)
        assert np.array_equal(dev.to_numpy(), want)
        return dev

    for reserve, build in ((45, 45), (7, 45), (45, 7), (0, 33)):
        nat.check(ctx._h, nat.lib().vs_links_reserve(ctx._h, C.c_uint32(reserve)))
        a = table(build)
        b = table(build)  # (the reservation is gone after one table of its size: this one allocates)
        del a, b
    nat.check(ctx._h, nat.lib().vs_links_reserve(ctx._h, C.c_uint32(300)))
    nat.check(ctx._h, nat.lib().vs_links_reserve(ctx._h, C.c_uint32(20)))  # replaces
    table(20)
    nat.check(ctx._h, nat.lib().vs_links_reserve(ctx._h, C.c_uint32(0)))


def test_link_table_from_device_counters(backend):
    """vs_links_from_counts reads the uint32 counters vs_pe_count filled, in place."""
    import torch
    from vstrains_amd.graph.hip_ops import HipPeLinks

    n = 45
    rng = np.random.default_rng(2)
    node = rng.integers(0, 2 ** 32, size=(n, n), dtype=np.int64)  # the whole uint32 range
    short = np.triu(rng.integers(0, 2 ** 32, size=(n, n), dtype=np.int64))
    node[0, 1], short[2, 2], short[3, 3] = 2 ** 31 + 5, 2 ** 32 - 1, 2 ** 31

    class Counter:
        pass

    c = Counter()
    c.n = n
    c.torch = torch
    c.device = torch.device("cuda:0")
    c.mats = torch.from_numpy(np.stack([node, short]).astype(np.uint32).view(np.int32)).to(c.device)
    dev = HipPeLinks.from_counter(backend.ctx, c, [str(i) for i in range(n)])
    want = node + node.T + short + short.T
    want[np.arange(n), np.arange(n)] = node.diagonal() + short.diagonal()
    assert np.array_equal(dev.to_numpy(), want)


@pytest.mark.parametrize("name", case_names())
def test_stage_graphs_of_golden_cases(backend, name, tmp_path):
    """Every stage graph a golden case passes through: device flows/scan == numpy checker."""
    from vstrains_amd.graph import pipeline
    import test_graph_golden as T

    case = Case(name)
    inp = case.inputs(str(tmp_path))
    out = str(tmp_path / "out")
    if case.meta["returncode"] != 0:
        with pytest.raises(case.expected_exception()):
            pipeline.run(case.args(inp, out), quiet_logger(), T.CheckerBackend(case, False))
    else:
        pipeline.run(case.args(inp, out), quiet_logger(), T.CheckerBackend(case, False))
    n = 0
    for fn in sorted(os.listdir(os.path.join(out, "gfa"))):
        if fn in ("graph_L0.gfa", "graph_L0r.gfa"):
            continue
        g, _, _ = read_stage_gfa(os.path.join(out, "gfa", fn))
        assert_same_ops(backend.graph_ops, g)
        n += 1
    assert n > 5 or case.meta["returncode"] != 0


@pytest.mark.parametrize("name", case_names())
def test_full_cli_on_device_matches_reference(backend, name, tmp_path):
    """The vstrains-compatible CLI end to end on the GPU: FASTQ -> pe_info/st_info (HIP PE-link
    inference) -> graph stages with HIP ops -> strain.paths / strain.fasta, all files compared
    with what the reference wrote."""
    from vstrains_amd import cli

    case = Case(name)
    inp = case.inputs(str(tmp_path), with_reads=True)
    out = str(tmp_path / "out")
    argv = ["-a", "spades", "-g", inp["gfa"], "-p", inp["paths"], "-o", out, "-fwd", inp["fwd"], "-rve", inp["rve"]]
    argv += case.meta["cli_extra"]
    if case.meta["returncode"] != 0:
        with pytest.raises(case.expected_exception()):  # the reference (its PE subprocess, or merge_id) dies the same way
            cli.main(argv, backend=backend)
    else:
        cli.main(argv, backend=backend)
    problems, _ = compare(case, out)
    binding = case.binding(problems)
    assert not binding, binding


def test_full_size_graph_stages_properties(backend, tmp_path):
    """BASELINE configs[2] graph (4.5 k-node input GFA, ~4.1 k nodes after the coverage cut-off):
    the reference cannot run this size, so the stages are checked through properties --
    the device link table equals the symmetrised counters taken on the host, every stage
    graph's device flows / scan equal the numpy checker, every extracted strain is a walk of
    graph_L0 whose FASTA record is the overlap-aware concatenation of its segments, and a second
    run writes the same files."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import numpy as np
    from vstrains_amd import pe as host
    from vstrains_amd.graph import pipeline
    from vstrains_amd.graph.formats import gfa_records
    from vstrains_amd.graph.hip_ops import HipPeLinks

    ctx = backend.ctx
    runs = []
    for attempt in range(2):
        out = str(tmp_path / ("run%d" % attempt))
        st, pre, names, seqs, cum, logger, _ = bench.workload(out, k=55)
        ctx.build_index(seqs, 55)
        counter = host.PeCounter(ctx)
        block = ctx.synth_pairs(st.genomes, cum, 20250001, 0, 2_000_000, 150, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
        counter.add(block)
        ctx.sync()
        block.free()
        table = HipPeLinks.from_counter(ctx, counter, names)
        if attempt == 0:
            node_mat, short_mat, _ = counter.result()
            p0 = node_mat + node_mat.T + short_mat + short_mat.T
            d = np.arange(len(names))
            p0[d, d] = node_mat.diagonal() + short_mat.diagonal()
            assert np.array_equal(table.to_numpy(), p0)
        pipeline.extract_strains(pre, table, backend, logger, out)
        runs.append(out)
    out = runs[0]
    # device flows / scan on every stage graph of the run
    n_graphs = 0
    for fn in sorted(os.listdir(os.path.join(out, "gfa"))):
        if fn in ("graph_L0.gfa", "graph_L0r.gfa") or not fn.endswith(".gfa"):
            continue
        if n_graphs % 6 == 0:  # every sixth graph keeps the test under a minute; all kinds are hit
            g, _, _ = read_stage_gfa(os.path.join(out, "gfa", fn))
            assert_same_ops(backend.graph_ops, g)
        n_graphs += 1
    assert n_graphs > 50
    # strains are walks of graph_L0 and the FASTA is their concatenation
    segs, links = gfa_records(os.path.join(out, "gfa", "graph_L0.gfa"))
    seq_of = {rec[1]: rec[2] for rec in segs}
    ovl = {(rec[1], rec[3]): int(rec[5][:-1]) for rec in links}
    paths = open(os.path.join(out, "strain.paths")).read().split("\n")
    fasta = open(os.path.join(out, "strain.fasta")).read().split("\n")
    n_strains = 0
    for i in range(0, len(paths) - 1, 2):
        name, ids = paths[i], paths[i + 1].split(",")
        l0 = ["-" + x[:-1] if x.endswith("-") else x for x in ids]  # graph_L0 names flipped segments "-id"
        text = ""
        for a, b in zip(l0, l0[1:]):
            assert (a, b) in ovl, (name, a, b)
        for j, a in enumerate(l0):
            text += seq_of[a] if j == len(l0) - 1 else seq_of[a][: len(seq_of[a]) - ovl[(a, l0[j + 1])]]
        assert fasta[i + 1] == text
        assert int(name.split("_")[2]) == len(text)
        n_strains += 1
    assert n_strains > 5
    # determinism: the second run wrote the same files
    for sub in ("gfa", "tmp", ""):
        d0 = os.path.join(runs[0], sub) if sub else runs[0]
        for fn in sorted(os.listdir(d0)):
            p0f = os.path.join(d0, fn)
            if os.path.isfile(p0f) and not fn.startswith("input."):
                assert open(p0f).read() == open(os.path.join(runs[1], sub, fn) if sub else os.path.join(runs[1], fn)).read(), fn


def test_cli_without_pe_text_gives_the_same_strains(tmp_path):
    """--no-pe-text: the N^2-line hand-off files are skipped, the counters go to the graph stages
    in device memory; everything else is byte-identical."""
    from vstrains_amd import cli

    case = Case("three_strain_k21")
    inp = case.inputs(str(tmp_path), with_reads=True)
    out = str(tmp_path / "out")
    cli.main(["-a", "spades", "-g", inp["gfa"], "-p", inp["paths"], "-o", out, "-fwd", inp["fwd"], "-rve", inp["rve"],
              "--no-pe-text"])
    assert not os.path.exists(os.path.join(out, "aln", "pe_info"))
    problems, _ = compare(case, out, skip=("aln/pe_info", "aln/st_info"))
    assert not problems, problems


@pytest.mark.parametrize("name", ["hiv_like_k55", "ten_strain_k31"])
def test_reference_shaped_driver_on_the_device(backend, name, tmp_path):
    """The reference's own call sequence and argument lists (vstrains_amd/graph/reference_api.py)
    with the HIP backend behind them: process_pe_info builds the link table on the device from the
    two text files, every stage runs its kernels, and the outputs equal the reference's."""
    from graph_case import Case, compare, quiet_logger
    from test_reference_api_cpu import reference_shaped_driver
    from vstrains_amd.graph import reference_api as api

    case = Case(name)
    api.set_backend(backend)
    try:
        inp = case.inputs(str(tmp_path))
        out = str(tmp_path / "out")
        reference_shaped_driver(case.args(inp, out), quiet_logger(), case)
    finally:
        api.set_backend(None)
    problems, _ = compare(case, out)
    binding = case.binding(problems)
    assert not binding, binding


def test_native_reinit_equals_the_python_rebuild(backend, tmp_path):
    """vs_stage_reinit on the device (survivor filter by name in map order, adjacency by the container's placement rule,
    flow / scan / chain kernels, GFA text from cached lines) against formats.stage_graph_from_state + the numpy checker
    on random graphs with gray vertices / edges, unmapped names, re-inserted names and multi-edges: same maps, same
    adjacency rows in the same order, same flows, same scan, same GFA bytes."""
    from vstrains_amd.graph.formats import stage_graph_from_state
    from vstrains_amd.graph.hip_ops import HipPeLinks
    from vstrains_amd.graph.native_stage import NativeStage

    import random

    table = HipPeLinks.from_matrices(backend.ctx, ["x"], np.zeros((1, 1), dtype=np.int64), np.zeros((1, 1), dtype=np.int64))
    st = NativeStage.on_device(backend.ctx, table, table.names)
    rng = np.random.default_rng(77)
    for trial in range(8):
        nv, ne = int(rng.integers(2, 400)), int(rng.integers(1, 900))
        g = random_graph(random.Random(trial), nv, ne, gray_frac=0.15, hub=int(rng.integers(0, 3)))
        nodes = {g.vid[v]: v for v in range(g.num_vertices())}
        edges = {}
        for e in g.edges():
            edges[(g.vid[g.esrc[e]], g.vid[g.etgt[e]])] = e  # (a repeated pair keeps its last edge, as a dict would)
        # retire a few names from the map (their edges must vanish) and re-insert one (it moves to the end)
        names = list(nodes)
        for name in names[:: max(7, nv // 9)][:5]:
            nodes.pop(name)
        if len(names) > 3 and names[1] in nodes:
            v = nodes.pop(names[1])
            nodes[names[1]] = v
        a = stage_graph_from_state(g, nodes, edges, gfa_path=str(tmp_path / "a.gfa"))
        ref = chk.NumpyGraphOps()
        try:
            scan_a = ref.refresh(a[0])
            err_a = None
        except FloatingPointError:
            err_a = "FloatingPointError"
        st.load_graph(g, nodes, edges)
        try:
            st.reinit(str(tmp_path / "b.gfa"))
            err_b = None
        except FloatingPointError:
            err_b = "FloatingPointError"
        assert (tmp_path / "a.gfa").read_bytes() == (tmp_path / "b.gfa").read_bytes()
        assert err_a == err_b
        if err_a is not None:
            continue
        ga = a[0]
        gb, nb, eb = st.graph()
        assert list(a[1].items()) == list(nb.items()) and list(a[2].items()) == list(eb.items())
        assert ga.adj == gb.adj and ga.nout == gb.nout and ga.esrc == gb.esrc and ga.etgt == gb.etgt and ga.eovl == gb.eovl
        assert ga.vid == gb.vid and ga.vdp == gb.vdp and ga.vseq == gb.vseq and ga.eflow == gb.eflow  # (bit-exact fp64)
        sb = st.scan()
        assert (scan_a.nontrivial, scan_a.fork_kind, scan_a.chain_next, scan_a.chain_rank) == \
               (sb.nontrivial, sb.fork_kind, sb.chain_next, sb.chain_rank)
        for v in range(ga.num_vertices()):
            if scan_a.chain_rank[v] >= 0:
                assert scan_a.chain_top[v] == sb.chain_top[v]
    st.close()


@pytest.mark.parametrize("hub_degrees", [(), (7, 128), (129,), (8, 129, 300, 1000)])
def test_stage_graph_with_and_without_rows_of_more_than_128_neighbours(backend, tmp_path, hub_degrees):
    """The stage handle's flow / scan operation launches the kernel for neighbour sums of more than 128 addends (numpy's
    recursive pairwise split) only when the graph has such a row: graphs without one, with one at the boundary and with
    several, flows and scan against the checker."""
    import random

    from vstrains_amd.graph.formats import stage_graph_from_state
    from vstrains_amd.graph.hip_ops import HipPeLinks
    from vstrains_amd.graph.native_stage import NativeStage

    rng = random.Random(len(hub_degrees) + sum(hub_degrees))
    g = AsmGraph()
    for i in range(1300):
        g.add_vertex(str(i), rng.uniform(0.001, 5000.0), "ACGT", True)
    v = 10
    for hub, deg in enumerate(hub_degrees):
        for k in range(deg):
            g.add_edge(hub, 10 + (v % 1200), 21, 0.0, True)
            g.add_edge(10 + ((v * 7) % 1200), hub, 21, 0.0, True)
            v += 1
    for i in range(20, 1290, 3):  # a sparse remainder: chains and small forks
        if g.edge(i, i + 1) is None:
            g.add_edge(i, i + 1, 21, 0.0, True)
    nodes = {g.vid[u]: u for u in range(g.num_vertices())}
    edges = {(g.vid[g.esrc[e]], g.vid[g.etgt[e]]): e for e in g.edges()}
    a = stage_graph_from_state(g, nodes, edges)
    want = chk.NumpyGraphOps().refresh(a[0])
    table = HipPeLinks.from_matrices(backend.ctx, ["x"], np.zeros((1, 1), dtype=np.int64), np.zeros((1, 1), dtype=np.int64))
    for _ in range(2):  # (twice: the second handle call finds the device buffers of the first)
        st = NativeStage.on_device(backend.ctx, table, table.names)
        st.load_graph(g, nodes, edges)
        st.reinit(str(tmp_path / "hubs.gfa"))
        got = st.scan()
        gb, _, _ = st.graph()
        assert gb.eflow == a[0].eflow
        assert (want.nontrivial, want.fork_kind, want.chain_next, want.chain_rank) == (got.nontrivial, got.fork_kind, got.chain_next, got.chain_rank)
        st.close()


def test_chain_ranking_of_a_large_graph_on_the_device(backend, tmp_path):
    """More than 8 192 vertices take the multi-workgroup pointer jumping (rounds that find nothing left return at once):
    long simple paths, short ones, a ring of simple edges and a branching remainder, scan and flows against the checker."""
    from vstrains_amd.graph.hip_ops import HipPeLinks
    from vstrains_amd.graph.native_stage import NativeStage

    import random

    rng = random.Random(3)
    g = AsmGraph()
    def path(n, ring=False):
        vs = [g.add_vertex(str(g.num_vertices()), rng.uniform(1.0, 500.0), "ACGT", True) for _ in range(n)]
        for a, b in zip(vs, vs[1:]):
            g.add_edge(a, b, 21, 0.0, True)
        if ring:
            g.add_edge(vs[-1], vs[0], 21, 0.0, True)
        return vs
    ends = [path(n) for n in (1, 2, 3, 100, 5000, 2500, 700)]
    path(50, ring=True)
    hubs = [g.add_vertex(str(g.num_vertices()), rng.uniform(1.0, 500.0), "ACGT", True) for _ in range(600)]
    for h in hubs:  # branching part: chains end in hubs, hubs feed chains
        g.add_edge(rng.choice(ends)[-1], h, 21, 0.0, True) if rng.random() < 0.5 else None
        t = rng.choice(hubs)
        if t != h and g.edge(h, t) is None:
            g.add_edge(h, t, 21, 0.0, True)
    assert g.num_vertices() > 8192
    nodes = {g.vid[v]: v for v in range(g.num_vertices())}
    edges = {(g.vid[g.esrc[e]], g.vid[g.etgt[e]]): e for e in g.edges()}
    from vstrains_amd.graph.formats import stage_graph_from_state

    a = stage_graph_from_state(g, nodes, edges)
    want = chk.NumpyGraphOps().refresh(a[0])
    table = HipPeLinks.from_matrices(backend.ctx, ["x"], np.zeros((1, 1), dtype=np.int64), np.zeros((1, 1), dtype=np.int64))
    st = NativeStage.on_device(backend.ctx, table, table.names)
    st.load_graph(g, nodes, edges)
    st.reinit(str(tmp_path / "big.gfa"))
    got = st.scan()
    gb, _, _ = st.graph()
    assert gb.eflow == a[0].eflow
    assert (want.nontrivial, want.fork_kind, want.chain_next, want.chain_rank) == (got.nontrivial, got.fork_kind, got.chain_next, got.chain_rank)
    assert max(want.chain_rank) == 4999 and min(want.chain_rank) == -1
    for v in range(a[0].num_vertices()):
        if want.chain_rank[v] >= 0:
            assert want.chain_top[v] == got.chain_top[v]
    st.close()


@pytest.mark.parametrize("config", [0, 1, 2, 3, 5])
def test_extraction_on_the_device_equals_the_checker_at_bench_size(backend, config, tmp_path):
    """The strain-extract leg of bench configs[0..3] (216 / 853 / 5 039 / 10 084 nodes, the reference cannot run the
    larger ones) on the link table of the config's WHOLE per-GPU block -- the 10 M pairs (12.5 M at configs[3]) the bench
    extracts from, not a prefix: every file the device run writes -- 116 stage GFAs at configs[2], about a thousand at
    configs[3], contig files, strain.paths, strain.fasta (the native stage handle with the HIP kernels underneath) --
    against the Python restatement of the stages over the numpy checker (oracle/graph_stages: Python rebuild, numpy
    flows / scans, link sums off the host copy of the counters).  [5] is the walk-heavy extra workload (3 strains of 150 kb,
    5 578 nodes: the greedy walk follows a strain for 86 kb), 2 M of its pairs."""
    import copy
    import hashlib

    import profile_extract_cpu as pec  # tests/profile_extract_cpu.py: checker backend + digests
    from vstrains_amd import pe as host
    from vstrains_amd.graph import pipeline
    from vstrains_amd.graph.hip_ops import HipPeLinks
    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[config]
    st, pre, names, seqs, cum, logger, _ = workload_for(config, str(tmp_path / "work"))
    ctx = backend.ctx
    ctx.build_index(seqs, cfg["k"])
    n_pairs = cfg["total_pairs"] // cfg["gpus"] if config != 5 else 2_000_000
    reads = ctx.synth_pairs(st.genomes, cum, 20250000 + config, 0, n_pairs, cfg["read_len"], int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    counter = host.PeCounter(ctx)
    counter.add(reads)
    node_mat, short_mat, _ = counter.result()
    outs = []
    for which in ("device", "checker"):
        out = str(tmp_path / which)
        for sub in ("gfa", "tmp"):
            os.makedirs(os.path.join(out, sub), exist_ok=True)
        if which == "device":
            strains = pipeline.extract_strains(copy.deepcopy(pre), HipPeLinks.from_counter(ctx, counter, names), backend, logger, out)
        else:
            strains = pipeline.extract_strains(copy.deepcopy(pre), pec.NumpyPeLinks(names, node_mat, short_mat), pec.Backend(), logger, out)
        outs.append((pec.digests(out), len(strains)))
    (dev, n_dev), (ref, n_ref) = outs
    assert n_dev == n_ref and n_dev > 0
    assert sorted(dev) == sorted(ref) and len(dev) > 40
    assert [f for f in dev if dev[f] != ref[f]] == []


def test_extraction_at_config4_size_device_operations_equal_the_cpu_checker(backend, tmp_path):
    """configs[4] (54 465 nodes, the whole 25 M-pair block of one of its eight GPUs): the strain-extract leg with the HIP
    kernels underneath -- flows, scans, chain ranking beyond one workgroup's 8 192 vertices, link sums over the device's
    link table -- against
      (a) (r6) the digests of an INDEPENDENT run: the checker's Python statement of the stages (oracle/graph_stages: its own
          graph container, GFA reader / writer, contig bookkeeping; numpy operations) on the same link table, run once in
          the build container (tools/extract_digests_from_csr.py on the CSR rows tools/dump_links_csr.py wrote here;
          tests/golden/extract_digests_config4.json) -- every decision of the engine at the largest BASELINE graph;
      (b) the SAME engine over the CPU checker of its device operations (oracle/stage_check.cpp, the link table as CSR rows
          of its non-zero cells): what the DEVICE computes.
    All files byte for byte, the same strains."""
    import copy
    import json

    import native_check
    import profile_extract_cpu as pec
    from conftest import GOLDEN
    from vstrains_amd import pe as host
    from vstrains_amd.graph import pipeline
    from vstrains_amd.graph.hip_ops import HipPeLinks
    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[4]
    st, pre, names, seqs, cum, logger, _ = workload_for(4, str(tmp_path / "work"))
    ctx = backend.ctx
    ctx.build_index(seqs, cfg["k"])
    reads = ctx.synth_pairs(st.genomes, cum, 20250004, 0, cfg["total_pairs"] // cfg["gpus"], cfg["read_len"], int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    counter = host.PeCounter(ctx)
    counter.add(reads)
    del reads
    row_ptr, col, val, csr_sha = native_check.links_csr_of_counter(counter)
    fixture = None
    fpath = os.path.join(GOLDEN, "extract_digests_config4.json")
    if os.path.exists(fpath):
        with open(fpath) as fh:
            fixture = json.load(fh)
        assert fixture["csr_sha256"] == csr_sha, "the independent run worked on another link table than this device counted"

    class CheckerOps:
        def native_stage(self, table):
            return native_check.stage_over_checker_sparse(list(names), row_ptr, col, val)

    outs = []
    for which in ("device", "checker"):
        out = str(tmp_path / which)
        for sub in ("gfa", "tmp"):
            os.makedirs(os.path.join(out, sub), exist_ok=True)
        if which == "device":
            strains = pipeline.extract_strains(copy.deepcopy(pre), HipPeLinks.from_counter(ctx, counter, names), backend, logger, out)
        else:
            strains = pipeline.extract_strains(copy.deepcopy(pre), None, CheckerOps(), logger, out)
        outs.append((pec.digests(out), {k: (list(rec[0]), rec[1], rec[2]) for k, rec in strains.items()}))
        import shutil

        shutil.rmtree(out)  # (2.7 GB of stage graphs per run)
    (dev, s_dev), (ref, s_ref) = outs
    assert s_dev == s_ref and len(s_dev) > 100
    assert sorted(dev) == sorted(ref) and len(dev) > 500
    assert [f for f in dev if dev[f] != ref[f]] == []
    assert fixture is not None, "tests/golden/extract_digests_config4.json is missing: the independent leg did not run"
    want = fixture["files_sha256"]
    assert sorted(dev) == sorted(want), sorted(set(dev) ^ set(want))[:10]
    assert [f for f in sorted(dev) if dev[f] != want[f]] == []
    assert s_dev == {k: (list(r[0]), r[1], r[2]) for k, r in fixture["strains"].items()}


def test_randomized_extraction_campaign_short():
    """tests/fuzz_graph.py for fifteen seconds: random strain sets through the workload generator, the
    extraction leg on the device against the same host logic over the numpy checker, every written
    file compared (the full campaign of the round: 1 749 draws of 5-2 430 nodes, no mismatch)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_graph.py"), "15", "3"], cwd=root,
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "mismatches 0" in proc.stdout and "draws 0," not in proc.stdout, proc.stdout[-500:]
    import warnings

    warnings.warn(UserWarning("fuzz_graph: %s" % [l for l in proc.stdout.splitlines() if l.startswith("draws ")][-1]))
