"""AsmGraph against the test stand-in that produced the golden fixtures (tests/golden/gt_standin):
the two must agree on every ordering rule, otherwise fixtures and product would drift apart
silently.  Also: the in-memory stage rebuild equals re-reading the stage GFA."""
import os
import random
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden", "gt_standin"))

from vstrains_amd.graph.asm_graph import AsmGraph  # noqa: E402
from vstrains_amd.graph.formats import read_stage_gfa, stage_graph_from_state, write_stage_gfa  # noqa: E402


def _standin():
    import graph_tool

    assert graph_tool.INEDGE_ROTATION, "fixtures are generated with the rotating in-entry rule"
    return graph_tool


@pytest.mark.parametrize("seed", range(8))
def test_random_edit_sequences_agree_with_the_standin(seed):
    gt = _standin()
    rng = random.Random(seed)
    ref = gt.Graph(directed=True)
    ovl = ref.new_edge_property("int", val=0)
    mine = AsmGraph()
    ref_edges = {}  # my edge index -> stand-in Edge
    nv = rng.randrange(3, 30)
    for i in range(nv):
        ref.add_vertex()
        mine.add_vertex(str(i), 1.0, "A", True)
    live = []
    for step in range(400):
        if live and rng.random() < 0.3:
            e = live.pop(rng.randrange(len(live)))
            ref.remove_edge(ref_edges.pop(e))
            mine.remove_edge(e)
        else:
            s, t = rng.randrange(nv), rng.randrange(nv)
            if mine.edge(s, t) is not None:
                continue
            re_ = ref.add_edge(ref.vertex(s), ref.vertex(t))
            ovl[re_] = step
            e = mine.add_edge(s, t, step, 0.0, True)
            assert e == re_._idx            # same edge index (first-in first-out reuse)
            ref_edges[e] = re_
            live.append(e)
        if step % 25 == 0:
            for v in range(nv):
                rv = ref.vertex(v)
                assert [int(x) for x in rv.out_neighbors()] == mine.out_neighbors(v)
                assert [int(x) for x in rv.in_neighbors()] == mine.in_neighbors(v)
                assert [x._idx for x in rv.all_edges()] == mine.all_edges(v)
                assert rv.in_degree() == mine.in_degree(v) and rv.out_degree() == mine.out_degree(v)
            assert [x._idx for x in ref.edges()] == list(mine.edges())
            assert ref.num_edges() == mine.num_edges()
    for e in live:
        assert ovl[ref_edges[e]] == mine.eovl[e]


@pytest.mark.parametrize("seed", range(5))
def test_stage_rebuild_in_memory_equals_reading_the_file(seed, tmp_path):
    rng = random.Random(100 + seed)
    g = AsmGraph()
    nodes, edges = {}, {}
    nv = rng.randrange(5, 60)
    for i in range(nv):
        name = "n%d" % i if rng.random() < 0.8 else "%d&%d*A" % (i, i + 1)
        nodes[name] = g.add_vertex(name, rng.uniform(0.001, 5000.0), "ACGT" * rng.randrange(1, 5), rng.random() < 0.85)
    names = list(nodes)
    for _ in range(3 * nv):
        a, b = rng.choice(names), rng.choice(names)
        if (a, b) in edges:
            continue
        edges[(a, b)] = g.add_edge(nodes[a], nodes[b], 21, rng.random(), rng.random() < 0.85)
    for name in rng.sample(names, k=max(1, nv // 8)):  # mapped-out vertices (popped from the map, not gray)
        nodes.pop(name)
    path = str(tmp_path / "stage.gfa")
    write_stage_gfa(g, nodes, edges, path)
    a = read_stage_gfa(path)
    path2 = str(tmp_path / "stage_one_pass.gfa")
    b = stage_graph_from_state(g, nodes, edges, gfa_path=path2)  # rebuild + file in one pass (reinit)
    assert open(path2).read() == open(path).read()
    for (ga, na, ea), (gb, nb, eb) in ((a, b),):
        assert list(na.items()) == list(nb.items())
        assert list(ea.items()) == list(eb.items())
        assert ga.vid == gb.vid and ga.vseq == gb.vseq and ga.vdp == gb.vdp and ga.vblack == gb.vblack
        assert ga.adj == gb.adj and ga.nout == gb.nout
        assert ga.esrc == gb.esrc and ga.etgt == gb.etgt and ga.eovl == gb.eovl and ga.eblack == gb.eblack
        assert ga.num_edges() == gb.num_edges()


def test_csr_arrays_are_the_adjacency_in_stored_order():
    import numpy as np

    rng = random.Random(5)
    g = AsmGraph()
    for i in range(40):
        g.add_vertex(str(i), 1.0, "ACGT", True)
    for _ in range(120):
        a, b = rng.randrange(40), rng.randrange(40)
        if g.edge(a, b) is None:
            g.add_edge(a, b, 3, 0.0, True)
    row_ptr, n_out, nbr, eidx = g.csr_arrays()
    assert row_ptr.dtype == np.uint64 and n_out.dtype == nbr.dtype == eidx.dtype == np.uint32
    assert row_ptr[0] == 0 and row_ptr[-1] == len(nbr) == len(eidx) == 2 * g.num_edges()
    for v in range(40):
        lo, hi = int(row_ptr[v]), int(row_ptr[v + 1])
        assert list(zip(nbr[lo:hi].tolist(), eidx[lo:hi].tolist())) == [tuple(x) for x in g.adj[v]]
        assert n_out[v] == g.nout[v]
    empty = AsmGraph()
    r, no, nb, ei = empty.csr_arrays()
    assert r.tolist() == [0] and len(no) == len(nb) == len(ei) == 0
