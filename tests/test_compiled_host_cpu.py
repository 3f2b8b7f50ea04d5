"""The graph-stage host modules exist twice: as .py (source of truth) and, after
`__graft_entry__.build()`, as extension modules compiled from the same files
(vstrains_amd/graph/_compile.py).  Whichever the suite runs on, the other one has to give the same
files, and a compiled module must never outlive its source."""
import json
import os
import subprocess
import sys

import pytest

from vstrains_amd import graph as graph_pkg
from vstrains_amd.graph import _compile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, **env):
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)


def test_interpreted_modules_give_the_reference_files_too():
    if graph_pkg.host_modules() == "interpreted":
        pytest.skip("this run already uses the .py modules (no compiled build present)")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_graph_golden.py", "-x", "-q", "-k", "closed_form_links"],
                       cwd=ROOT, env=dict(os.environ, VS_GRAPH_INTERPRETED="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    probe = _run("import vstrains_amd.graph as g, vstrains_amd.graph.extend as e; print(g.host_modules(), e.__file__)",
                 VS_GRAPH_INTERPRETED="1")
    assert probe.stdout.split()[0] == "interpreted" and probe.stdout.split()[1].endswith("extend.py"), probe.stdout + probe.stderr


def test_a_compiled_module_is_dropped_when_its_source_changed(tmp_path):
    if not any(_compile.compiled_path(m) for m in _compile.MODULES):
        pytest.skip("no compiled build present")
    # a stamp that names another source digest for one module: that module must come from its .py
    stamp = _compile.read_stamp()
    mod = next(m for m in _compile.MODULES if _compile.compiled_path(m))
    saved = open(_compile.STAMP).read()
    try:
        stamp[mod] = "0" * 64
        with open(_compile.STAMP, "w") as fh:
            json.dump(stamp, fh)
        probe = _run("import importlib, vstrains_amd.graph as g; m = importlib.import_module('vstrains_amd.graph.%s'); "
                     "print(g.COMPILED['%s'], m.__file__)" % (mod, mod))
        assert probe.returncode == 0, probe.stderr
        flag, path = probe.stdout.split()
        assert flag == "False" and path.endswith(mod + ".py"), probe.stdout
    finally:
        with open(_compile.STAMP, "w") as fh:
            fh.write(saved)


def test_typed_front_half_of_reinit_equals_the_python_rebuild(tmp_path):
    """_stage_fast.prepare (typed Cython, no .py twin) against formats.stage_graph_from_state on random
    stages: gray vertices and edges, names dropped from the map, a re-inserted name, duplicate edge keys,
    odd depths (ints, 1e-300, 1e22, -0.0): same kept ids / depths / sequences, same maps, same edge arrays,
    the same GFA bytes."""
    import random

    import numpy as np

    from vstrains_amd.graph import fast_module
    from vstrains_amd.graph.asm_graph import AsmGraph
    from vstrains_amd.graph.formats import stage_graph_from_state

    fast = fast_module("_stage_fast")
    if fast is None:
        pytest.skip("_stage_fast is not built")
    rng = random.Random(9)
    cache = {}
    for trial in range(40):
        nv, ne = rng.randint(0, 120), rng.randint(0, 300)
        g = AsmGraph()
        for v in range(nv):
            dp = rng.choice([rng.random() * 1000, float(rng.randint(0, 50)), 1e-300, 1e22, -0.0, 123456789.123456789, rng.randint(1, 9)])
            g.add_vertex("n%d%s" % (v, rng.choice(["", "*A", "&x", "*0*B"])), dp, "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 90))),
                         rng.random() > 0.15)
        for _ in range(ne if nv else 0):
            g.add_edge(rng.randrange(nv), rng.randrange(nv), rng.choice([21, 31, 55, 0, 127]), 0.0, rng.random() > 0.15)
        nodes = {g.vid[v]: v for v in range(nv)}
        edges = {(g.vid[g.esrc[e]], g.vid[g.etgt[e]]): e for e in g.edges()}
        names = list(nodes)
        for name in names[::7][:4]:
            nodes.pop(name)
        if len(names) > 3 and names[1] in nodes:
            nodes[names[1]] = nodes.pop(names[1])
        # what the by-index shortcut must not get wrong: an edge filed under names that are not those of
        # its own ends, key strings that are equal to the ids without being the same objects, a vertex
        # that sits in the map under another name, two kept vertices with one id
        if trial % 4 == 1 and nv > 6 and edges:
            (ka, kb), e0 = next(iter(edges.items()))
            edges[(g.vid[nv - 1], g.vid[nv - 2])] = e0
        if trial % 4 == 2 and edges:
            edges = {("".join(list(a)), "".join(list(b))): e for (a, b), e in edges.items()}
        if trial % 4 == 3 and nv > 6:
            nodes["alias"] = nodes.pop(g.vid[nv - 1], nv - 1)
            g.vid[nv - 3] = g.vid[nv - 4]
        ref_g, ref_nn, ref_ne, ref_text = stage_graph_from_state(g, nodes, edges, gfa_path=str(tmp_path / "a.gfa"), want_text=True)
        n_vid, n_vdp, n_vseq, nn, kept_keys, src, tgt, ovl, a_src, a_tgt, a_dp, text = fast.prepare(
            g.vblack, g.vid, g.vdp, g.vseq, g.eblack, g.eovl, g.esrc, g.etgt, nodes, edges, cache)
        assert text == ref_text.encode()
        assert (n_vid, n_vdp, n_vseq) == (ref_g.vid, ref_g.vdp, ref_g.vseq)
        assert list(nn.items()) == list(ref_nn.items())
        assert dict(zip(kept_keys, range(len(src)))) == ref_ne and list(dict(zip(kept_keys, range(len(src))))) == list(ref_ne)
        assert (src, tgt, ovl) == (ref_g.esrc, ref_g.etgt, ref_g.eovl)
        assert a_src.tolist() == src and a_tgt.tolist() == tgt and a_dp.tolist() == [float(x) for x in n_vdp]
        assert a_src.dtype == np.uint32 and a_dp.dtype == np.float64
    # ids that are not str take the Python path: a TypeError, nothing else
    with pytest.raises(TypeError):
        fast.prepare([True], [5], [1.0], ["ACGT"], [], [], [], [], {5: 0}, {}, cache)
