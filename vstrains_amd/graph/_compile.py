"""Ahead-of-time compilation of the graph-stage host modules (Cython, C API of CPython).

The graph stages are Python like the reference's (`utils/VStrains_Decomposition.py`,
`VStrains_Extension.py`): per stage they walk every vertex and edge of the assembly graph a few
times, and at 5 000 nodes x 116 stages the interpreter's dispatch is most of `strain_extract_s`.
`build()` turns the listed modules, unchanged, into extension modules next to their sources (same
names, same code; the .py stays the source of truth and runs when no current build is there):

    python -c "from vstrains_amd.graph import _compile; _compile.build()"      # or __graft_entry__.build()

`compiled.json` records the SHA-256 of every source a module was built from;
`vstrains_amd/graph/__init__.py` sends a module whose source has changed since back to the .py.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MODULES = ["asm_graph", "contigs", "disentangle", "extend", "formats", "hip_ops", "ops"]
# hand-typed Cython without an interpreted twin: imported only when built from the present source
# (`vstrains_amd.graph.fast_module`); its callers carry the Python statement of the same thing
NATIVE_ONLY = ["_stage_fast"]
STAMP = os.path.join(HERE, "compiled.json")
BUILD_DIR = os.path.join(HERE, "_cbuild")


def source_path(module: str) -> str:
    return os.path.join(HERE, module + (".pyx" if module in NATIVE_ONLY else ".py"))


def source_digest(module: str) -> str:
    with open(source_path(module), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def compiled_path(module: str):
    import importlib.machinery

    for suffix in importlib.machinery.EXTENSION_SUFFIXES:
        path = os.path.join(HERE, module + suffix)
        if os.path.exists(path):
            return path
    return None


def read_stamp() -> dict:
    try:
        with open(STAMP) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def current() -> dict:
    """{module: True when a compiled module built from exactly the present source is in place}."""
    stamp = read_stamp()
    return {m: compiled_path(m) is not None and stamp.get(m) == source_digest(m) for m in MODULES + NATIVE_ONLY}


def build(force: bool = False, quiet: bool = True) -> None:
    state = current()
    todo = [m for m in MODULES + NATIVE_ONLY if force or not state[m]]
    if not todo:
        return
    from Cython.Build import cythonize
    from setuptools import Extension
    from setuptools.dist import Distribution

    root = os.path.dirname(os.path.dirname(HERE))
    cwd = os.getcwd()
    os.chdir(root)  # (module paths below are relative to the repository root)
    try:
        exts = [Extension("vstrains_amd.graph.%s" % m, [os.path.relpath(source_path(m), root)],
                          extra_compile_args=["-O2", "-g0"]) for m in todo]
        exts = cythonize(exts, language_level=3, build_dir=BUILD_DIR, quiet=quiet,
                         compiler_directives={"binding": True, "boundscheck": True, "wraparound": True})
        dist = Distribution({"name": "vstrains_amd_graph", "ext_modules": exts})
        cmd = dist.get_command_obj("build_ext")
        cmd.inplace = 1
        cmd.build_temp = os.path.join(BUILD_DIR, "tmp")
        cmd.build_lib = os.path.join(BUILD_DIR, "lib")
        cmd.parallel = min(len(todo), os.cpu_count() or 1)
        cmd.ensure_finalized()
        if quiet:
            import contextlib
            import io

            with contextlib.redirect_stdout(io.StringIO()):
                cmd.run()
        else:
            cmd.run()
    finally:
        os.chdir(cwd)
    stamp = read_stamp()
    for m in todo:
        stamp[m] = source_digest(m)
    with open(STAMP, "w") as fh:
        json.dump(stamp, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv, quiet="-v" not in sys.argv)
    print(json.dumps(current(), indent=1))
