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
