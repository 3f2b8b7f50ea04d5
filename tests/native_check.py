"""Test helper: the native stage engine (vstrains_amd/csrc/vs_stage.cpp) over the CPU checker of its three device
operations (oracle/stage_check.cpp -> oracle/_build/libvs_stage_check.so), so that the engine's decisions can be run
against the golden cases without a GPU.  The product path creates the same handle with ``vs_stage_create`` on a HIP
context instead (``hip_ops.HipBackend.native_stage``)."""
import ctypes as C
import os

import numpy as np

from vstrains_amd.graph import native_stage as ns

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def check_lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "_build", "libvs_stage_check.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        ns.bind(L)
        L.vs_stage_check_create.restype = C.c_int
        L.vs_stage_check_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.vs_stage_check_create_sparse.restype = C.c_int
        L.vs_stage_check_create_sparse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        _lib = L
    return _lib


def dense_links(table) -> np.ndarray:
    """The symmetrised PE-link table (oracle.graph_ops.DictPeLinks or anything with names + block_sums / sym) as a
    dense int64 matrix."""
    if hasattr(table, "sym"):
        return np.ascontiguousarray(table.sym, dtype=np.int64)
    names = table.names
    n = len(names)
    p0 = np.zeros((n, n), dtype=np.int64)
    for i, a in enumerate(names):
        for j, b in enumerate(names):
            p0[i, j] = table.table[(min(a, b), max(a, b))]
    return p0


def stage_over_checker(names, p0: np.ndarray) -> ns.NativeStage:
    L = check_lib()
    p0 = np.ascontiguousarray(p0, dtype=np.int64)
    h = C.c_void_p()
    rc = L.vs_stage_check_create(p0.ctypes.data, len(names), C.byref(h))
    assert rc == 0, rc
    st = ns.NativeStage(L, h)
    st.set_link_names(list(names))
    return st


def stage_over_checker_sparse(names, row_ptr, col, val) -> ns.NativeStage:
    """The same with the symmetrised table as CSR rows of its non-zero cells (graphs whose dense table would not fit:
    54 k nodes are 23.7 GB)."""
    L = check_lib()
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    val = np.ascontiguousarray(val, dtype=np.int64)
    assert row_ptr.shape[0] == len(names) + 1 and col.shape == val.shape and int(row_ptr[-1]) == col.shape[0]
    h = C.c_void_p()
    rc = L.vs_stage_check_create_sparse(row_ptr.ctypes.data, col.ctypes.data, val.ctypes.data, len(names), C.byref(h))
    assert rc == 0, rc
    st = ns.NativeStage(L, h)
    st.set_link_names(list(names))
    return st
