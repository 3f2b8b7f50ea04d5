"""Graph stages (edge cleaning, disentanglement, path extension): Python host logic like the
reference's, device operations behind the C ABI.

The host modules listed in `_compile.MODULES` may be present as ahead-of-time compiled extension
modules (same names, built from the same sources by `_compile.build()`).  A compiled module is
used only when it was built from exactly the source that lies next to it; otherwise -- and always
with VS_GRAPH_INTERPRETED=1 -- the import goes to the .py."""
import importlib.abc
import importlib.util
import os
import sys

from . import _compile


class _SourceFinder(importlib.abc.MetaPathFinder):
    """Sends the named modules of this package to their .py (the default finder prefers the
    extension module in the same directory)."""

    def __init__(self, names):
        self.names = set(names)

    def find_spec(self, fullname, path=None, target=None):
        pkg, _, mod = fullname.rpartition(".")
        if pkg == __name__ and mod in self.names:
            return importlib.util.spec_from_file_location(fullname, os.path.join(_compile.HERE, mod + ".py"))
        return None


def _route_imports():
    interpreted = os.environ.get("VS_GRAPH_INTERPRETED", "") not in ("", "0")
    fresh = _compile.current()
    to_source = [m for m in _compile.MODULES if _compile.compiled_path(m) and (interpreted or not fresh[m])]
    if to_source:
        sys.meta_path.insert(0, _SourceFinder(to_source))
    return {m: fresh[m] and not interpreted for m in _compile.MODULES}


COMPILED = _route_imports()  # {module: runs as a compiled extension module}


def fast_module(name: str):
    """A module of `_compile.NATIVE_ONLY` (typed Cython, no .py twin), or None when it is not built from
    the present source or VS_GRAPH_INTERPRETED is set -- the caller then runs its Python statement."""
    if os.environ.get("VS_GRAPH_INTERPRETED", "") not in ("", "0") or not _compile.current().get(name):
        return None
    import importlib

    try:
        return importlib.import_module(__name__ + "." + name)
    except ImportError:
        return None


def host_modules() -> str:
    """'compiled' / 'interpreted' / 'mixed' -- reported by bench.py next to strain_extract_s."""
    vals = set(COMPILED.values())
    return "compiled" if vals == {True} else "interpreted" if vals == {False} else "mixed"
