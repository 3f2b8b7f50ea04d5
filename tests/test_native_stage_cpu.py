"""The native stage engine (vstrains_amd/csrc/vs_stage.cpp) on the CPU: the pieces of Python / numpy behaviour it restates
(float repr, round, numpy.mean / median, iteration order of a set of ints), its re-initialisation against the Python
rebuild, the blob round trip of its state, and the whole extraction leg against the Python restatement of the stages
(oracle/graph_stages) on random workloads.  The engine runs over the CPU checker of its device operations
(oracle/stage_check.cpp); the GPU suite runs the same engine over the HIP kernels (tests/test_graph_gpu.py)."""
import ctypes as C
import math
import os
import random
import struct
import subprocess
import sys

import numpy as np
import pytest

import native_check
from oracle import graph_ops as chk
from vstrains_amd.graph.asm_graph import AsmGraph
from vstrains_amd.graph.formats import stage_graph_from_state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    L = native_check.check_lib()
    L.vs_check_py_repr.restype = C.c_int
    L.vs_check_py_repr.argtypes = [C.c_double, C.c_char_p, C.c_int]
    L.vs_check_py_round2.restype = C.c_double
    L.vs_check_py_round2.argtypes = [C.c_double]
    for name in ("vs_check_np_mean", "vs_check_np_median"):
        getattr(L, name).restype = C.c_double
        getattr(L, name).argtypes = [C.c_void_p, C.c_uint64]
    L.vs_check_int_set_order.restype = C.c_uint32
    L.vs_check_int_set_order.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    return L


def test_float_text_is_pythons_repr(lib):
    rng = random.Random(5)
    values = [0.0, -0.0, 1.0, 100.0, 70.5, 0.1 + 0.2, 1e15, 1e16, 1.5e16, 123456789012345678.0, 1e-4, 1e-5, 9.999e-5, 5e-324, 1.7976931348623157e308,
              2.5, 1 / 3, 1e22, 1e23, 0.05 * 101.25, float("inf"), float("-inf")]
    values += [rng.uniform(0, 3000) for _ in range(2000)]
    values += [struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0] for _ in range(3000)]
    buf = C.create_string_buffer(64)
    for x in values:
        if math.isnan(x):
            continue
        n = lib.vs_check_py_repr(x, buf, 64)
        assert n > 0 and buf.value.decode() == repr(x), (x, buf.value)
    assert lib.vs_check_py_repr(float("nan"), buf, 64) == 3 and buf.value == b"nan"


def test_round_to_two_places_is_pythons_round(lib):
    rng = random.Random(6)
    values = [2.675, 0.285, 1.005, 0.125, 0.375, 2.5, 1e-9, 123456.785, 0.0, 70.5]
    values += [rng.uniform(0, 5000) for _ in range(5000)] + [k / 1000.0 for k in range(0, 4000, 5)]
    for x in values:
        assert lib.vs_check_py_round2(x) == round(x, 2), x


def test_mean_and_median_are_numpys(lib):
    rng = np.random.default_rng(7)
    for n in list(range(1, 20)) + [127, 128, 129, 255, 256, 257, 1000, 4099]:
        a = np.ascontiguousarray(rng.uniform(0.1, 3000.0, size=n))
        assert lib.vs_check_np_mean(a.ctypes.data, n) == float(np.mean(a.tolist())), n
        assert lib.vs_check_np_median(a.ctypes.data, n) == float(np.median(a.tolist())), n
    a = np.asarray([5.0, 1.0, 5.0, 1.0], dtype=np.float64)
    assert lib.vs_check_np_median(a.ctypes.data, 4) == float(np.median(a.tolist()))


def test_int_set_iteration_order_is_cpythons(lib):
    """The reference iterates ``set(vertex.in_neighbors())`` (Decomposition.py:561,625; hash(Vertex) = its index): the
    engine must visit the neighbours in the order THIS interpreter's set gives."""
    rng = random.Random(8)
    for trial in range(4000):
        n = rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 21, 22, 40, 86, 200])
        hi = rng.choice([8, 64, 1000, 60000, 5_000_000])
        vals = [rng.randrange(hi) for _ in range(n)]
        a = np.asarray(vals, dtype=np.uint32)
        out = np.zeros(n, dtype=np.uint32)
        k = lib.vs_check_int_set_order(a.ctypes.data, n, out.ctypes.data)
        assert out[:k].tolist() == list(set(vals)), vals


def random_graph(rng, nv, ne, gray_frac=0.0, hub=0):
    g = AsmGraph()
    for i in range(nv):
        g.add_vertex(str(i), rng.uniform(0.5, 2000.0), "ACGT" * rng.randrange(1, 4), rng.random() >= gray_frac)
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


def scrambled_state(seed):
    rng = np.random.default_rng(seed)
    nv, ne = int(rng.integers(2, 300)), int(rng.integers(1, 700))
    g = random_graph(random.Random(seed), nv, ne, gray_frac=0.15, hub=int(rng.integers(0, 3)))
    # a few removals, so that freed edge indices and shifted adjacency rows are part of the state
    live = list(g.edges())
    for e in live[:: max(5, len(live) // 6)][:4]:
        g.remove_edge(e)
    nodes = {g.vid[v]: v for v in range(g.num_vertices())}
    edges = {}
    for e in g.edges():
        edges[(g.vid[g.esrc[e]], g.vid[g.etgt[e]])] = e
    names = list(nodes)
    for name in names[:: max(7, nv // 9)][:5]:
        nodes.pop(name)
    if len(names) > 3 and names[1] in nodes:
        v = nodes.pop(names[1])
        nodes[names[1]] = v
    return g, nodes, edges


@pytest.mark.parametrize("seed", range(12))
def test_state_round_trips_through_the_handle(seed):
    g, nodes, edges = scrambled_state(seed)
    st = native_check.stage_over_checker(["x"], np.zeros((1, 1), dtype=np.int64))
    st.load_graph(g, nodes, edges)
    h, n2, e2 = st.graph()
    for slot in AsmGraph.__slots__:
        a, b = getattr(g, slot), getattr(h, slot)
        assert (list(a) == list(b)) if slot != "_n_edges" else a == b, slot
    assert list(n2.items()) == list(nodes.items()) and list(e2.items()) == list(edges.items())
    contigs = {"c1": [["1", "2"], 300, 12.5], "c2$0": [["0"], 7, float(np.float64(3.25))], "e": [[], 0, 1.0]}
    st.load_contigs(contigs)
    assert st.contigs() == contigs
    table = {"5": {("1", "2"): 7, ("3", "2"): 1}, "9": {}}
    st.load_full_link(table)
    got = st.full_link()
    assert got == table and [list(v) for v in got.values()] == [list(v) for v in table.values()]
    st.close()


@pytest.mark.parametrize("seed", range(12))
def test_native_reinit_equals_the_python_rebuild(seed, tmp_path):
    """vs_stage_reinit against formats.stage_graph_from_state + the numpy checker: same GFA bytes, maps, adjacency rows,
    flows (bit-exact) and scan; a divide-by-zero is reported as the reference's FloatingPointError by both."""
    g, nodes, edges = scrambled_state(100 + seed)
    a = stage_graph_from_state(g, nodes, edges, gfa_path=str(tmp_path / "a.gfa"))
    ref = chk.NumpyGraphOps()
    try:
        scan_a = ref.refresh(a[0])
        err_a = None
    except FloatingPointError:
        err_a = "FloatingPointError"
    st = native_check.stage_over_checker(["x"], np.zeros((1, 1), dtype=np.int64))
    st.load_graph(g, nodes, edges)
    try:
        st.reinit(str(tmp_path / "b.gfa"))
        err_b = None
    except FloatingPointError:
        err_b = "FloatingPointError"
    assert (tmp_path / "a.gfa").read_bytes() == (tmp_path / "b.gfa").read_bytes()
    assert err_a == err_b
    if err_a is None:
        ga = a[0]
        gb, nb, eb = st.graph()
        assert list(a[1].items()) == list(nb.items()) and list(a[2].items()) == list(eb.items())
        assert ga.adj == gb.adj and ga.nout == gb.nout and ga.esrc == gb.esrc and ga.etgt == gb.etgt and ga.eovl == gb.eovl
        assert ga.vid == gb.vid and ga.vdp == gb.vdp and ga.vseq == gb.vseq and ga.eflow == gb.eflow
        sb = st.scan()
        assert (scan_a.nontrivial, scan_a.fork_kind, scan_a.chain_next, scan_a.chain_top, scan_a.chain_rank) == \
               (sb.nontrivial, sb.fork_kind, sb.chain_next, sb.chain_top, sb.chain_rank)
        # an untouched stage re-initialises to the same bytes under a new name, without another scan
        before = st.counters()["graph_refresh_launches"]
        st.reinit(str(tmp_path / "c.gfa"))
        assert (tmp_path / "c.gfa").read_bytes() == (tmp_path / "a.gfa").read_bytes()
        assert st.counters()["graph_refresh_launches"] == before and st.counters()["reinit_reused"] == 1
    st.close()


def test_checker_operations_equal_the_numpy_checker():
    """oracle/stage_check.cpp (what the engine runs over in the CPU suite) against oracle/graph_ops.py on packed graphs."""
    for seed in range(8):
        rng = random.Random(seed)
        g0 = random_graph(rng, rng.randrange(2, 200), rng.randrange(1, 500), hub=seed % 3)
        nodes = {g0.vid[v]: v for v in range(g0.num_vertices())}
        edges = {(g0.vid[g0.esrc[e]], g0.vid[g0.etgt[e]]): e for e in g0.edges()}
        g, nodes, edges = stage_graph_from_state(g0, nodes, edges)
        ref = chk.NumpyGraphOps()
        try:
            want = ref.refresh(g)
        except FloatingPointError:
            continue
        st = native_check.stage_over_checker(["x"], np.zeros((1, 1), dtype=np.int64))
        st.load_graph(g, nodes, edges)
        st.refresh_scan()
        got = st.scan()
        assert (want.nontrivial, want.fork_kind, want.chain_next, want.chain_top, want.chain_rank) == \
               (got.nontrivial, got.fork_kind, got.chain_next, got.chain_top, got.chain_rank)
        st.close()


def test_errors_carry_the_reference_exception():
    st = native_check.stage_over_checker(["a", "b"], np.zeros((2, 2), dtype=np.int64))
    with pytest.raises(KeyError):
        st.link("a", "nobody")
    assert st.link("a", "b") == 0
    with pytest.raises(RuntimeError):
        st.best_matching()  # (no re-initialisation yet: no scan)
    st.close()


def test_randomized_campaign_against_the_python_stages_short():
    """tests/fuzz_native_cpu.py for twenty seconds: random workloads, the extraction leg through the native engine and
    through the Python restatement of the stages, every written file and the strain records compared."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_native_cpu.py"), "20", "5"], cwd=ROOT, capture_output=True,
                          text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "mismatches 0" in proc.stdout and "draws 0," not in proc.stdout, proc.stdout[-500:]
    import warnings

    warnings.warn(UserWarning("fuzz_native_cpu: %s" % [l for l in proc.stdout.splitlines() if l.startswith("draws ")][-1]))


def test_fork_chain_limit_is_the_one_measured_on_the_real_reference_function():
    """tests/golden/probe_merge_id_depth.py imported ``contig_dict_remapping`` from /root/reference and measured how many
    nested ``merge_id`` frames (Utilities.py:318-327) this interpreter lets it have at the CLI's call depth: 993; the next
    link raises RecursionError.  The native engine (check_fork_depth) and the Python checker restate that number; the golden
    case ``circular_runaway_k55`` shows the reference ending that way (8 549 links)."""
    import json
    import re

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "merge_id_depth.json")) as fh:
        want = json.load(fh)
    assert want["error_one_beyond"].startswith("RecursionError: maximum recursion depth exceeded")
    from oracle.graph_stages import contig_ops

    assert contig_ops.PY_MERGE_ID_FRAMES == want["max_nested_merge_id_frames"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "vstrains_amd", "csrc", "vs_stage.cpp")) as fh:
        (n,) = re.findall(r"atoi\(e\) : (\d+)u;", fh.read())  # (the default of merge_id_frames(); VS_STAGE_MERGE_ID_FRAMES overrides it)
    assert int(n) == want["max_nested_merge_id_frames"]
    # the checker on chains either side of the limit (no graph needed: nothing is a contig)
    import logging

    for frames, ok in ((want["max_nested_merge_id_frames"], True), (want["max_nested_merge_id_frames"] + 1, False)):
        ids = ["n%d" % i for i in range(frames)]
        id_mapping = {ids[i]: {ids[i + 1]: None} for i in range(frames - 1)}
        id_mapping[ids[-1]] = {}
        lg = logging.getLogger("chain")
        lg.handlers[:] = [logging.NullHandler()]
        if ok:
            closure = contig_ops.remap_contigs(None, {}, {}, {}, id_mapping, [ids[0]], lg)
            assert list(closure[ids[0]]) == [ids[-1]]
        else:
            with pytest.raises(RecursionError):
                contig_ops.remap_contigs(None, {}, {}, {}, id_mapping, [ids[0]], lg)


def test_import_refuses_sections_that_do_not_fit_and_keeps_one_copy_of_a_sequence():
    """ vs_stage_import is public C ABI: a scan of another snapshot, an edge whose end is no vertex, a free-list
    entry that is no slot are ValueErrors, not out-of-bounds reads later; and a handle that is loaded again and again (as
    reference_api does for every call) holds every distinct sequence once."""
    import copy
    import resource

    from vstrains_amd.graph import native_stage as ns

    g, nodes, edges = scrambled_state(3)
    st = native_check.stage_over_checker(["x"], np.zeros((1, 1), dtype=np.int64))
    st.load_graph(g, nodes, edges)

    class Scan:
        pass

    sc = Scan()
    nv = len(g.vid) + 1  # (a scan of a graph with one vertex more)
    sc.nontrivial, sc.fork_kind = [0] * nv, [0] * nv
    sc.chain_next, sc.chain_top, sc.chain_rank = [-1] * nv, [0] * nv, [0] * nv
    with pytest.raises(ValueError, match="scan section"):
        st._import(ns.pack_scan(sc))
    live = [e for e in range(len(g.esrc)) if e not in set(g._free)]
    bad = copy.deepcopy(g)
    bad.esrc[live[0]] = len(g.vid) + 7
    with pytest.raises(ValueError, match="edge end out of range"):
        st.load_graph(bad, nodes, edges)
    bad = copy.deepcopy(g)
    bad._free.append(len(g.esrc) + 3)
    with pytest.raises(ValueError, match="free edge slot out of range"):
        st.load_graph(bad, nodes, edges)
    # thirty loads of 20 MB of sequences: one copy stays
    big = copy.deepcopy(g)
    rng = random.Random(1)
    big.vseq = ["".join(rng.choice("ACGT") for _ in range(200)) * 500 + str(i) for i in range(len(g.vid))]
    st.load_graph(big, nodes, edges)
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for _ in range(30):
        st.load_graph(big, nodes, edges)
    h, _, _ = st.graph()
    assert list(h.vseq) == list(big.vseq)
    grown_mb = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before) / 1024.0
    assert grown_mb < 0.3 * 30 * sum(map(len, big.vseq)) / 1e6, grown_mb
    st.close()
