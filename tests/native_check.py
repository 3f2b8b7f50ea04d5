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


def links_csr_of_counter(counter):
    """The symmetrised PE-link table (IO.py:598-627: both orders of both matrices summed, the diagonal once) of a device
    counter as CSR rows of its non-zero cells, in the GFA's node order (whatever numbering the device index uses):
    (row_ptr u64 [n + 1], col u32 ascending inside a row, val i64, sha256 of the three).  tools/dump_links_csr.py writes the
    same arrays for the build container (tests/golden/extract_digests_config4.json names their hash)."""
    import hashlib

    import torch

    n = counter.n
    order = counter.node_order
    parts_ij, parts_v = [], []
    for lo in range(0, n, 4096):  # (row slabs: torch.nonzero does not take a tensor of 3e9 cells)
        blk = counter.mats[0, lo:lo + 4096].to(torch.int64) + counter.mats[1, lo:lo + 4096]
        nz = torch.nonzero(blk)
        parts_v.append(blk[nz[:, 0], nz[:, 1]])
        nz[:, 0] += lo
        parts_ij.append(nz)
    ij, v = torch.cat(parts_ij), torch.cat(parts_v)
    del parts_ij, parts_v
    if order is not None:
        o = torch.as_tensor(np.asarray(order, dtype=np.int64), device=ij.device)
        ij = torch.stack([o[ij[:, 0]], o[ij[:, 1]]], dim=1)
    off = ij[:, 0] != ij[:, 1]
    keys = torch.cat([ij[:, 0] * n + ij[:, 1], ij[off, 1] * n + ij[off, 0]])
    vals = torch.cat([v, v[off]])
    uk, inv = torch.unique(keys, return_inverse=True)
    uv = torch.zeros(uk.shape[0], dtype=torch.int64, device=uk.device).scatter_add_(0, inv, vals)
    rows = (uk // n).cpu().numpy()
    col = (uk % n).cpu().numpy().astype(np.uint32)
    val = uv.cpu().numpy()
    row_ptr = np.zeros(n + 1, dtype=np.uint64)
    row_ptr[1:] = np.cumsum(np.bincount(rows, minlength=n))
    sha = hashlib.sha256(row_ptr.tobytes() + col.tobytes() + val.astype(np.uint32).tobytes()).hexdigest()
    del ij, v, keys, vals, uk, inv, uv
    torch.cuda.empty_cache()
    return row_ptr, col, val, sha
