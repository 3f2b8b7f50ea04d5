"""ctypes face of oracle/pe_oracle.c  --  TEST INFRASTRUCTURE ONLY (see that file's header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpe_oracle.so")
_lib = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.peo_create.restype = C.c_void_p
        L.peo_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        L.peo_destroy.argtypes = [C.c_void_p]
        L.peo_table_entries.restype = C.c_uint64
        L.peo_table_entries.argtypes = [C.c_void_p]
        L.peo_map_end.restype = C.c_uint32
        L.peo_map_end.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.peo_count_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.peo_count_pairs_keys.restype = C.c_uint64
        L.peo_count_pairs_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
        L.peo_synth_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                      C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def concat(seqs: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    # one byte per CHARACTER (the reference works on decoded text, PE_Inference.py:147-152): a character outside
    # ASCII becomes '?', which like any byte outside ACGT makes the windows over it miss
    data = np.frombuffer("".join(seqs).encode("ascii", "replace"), dtype=np.uint8).copy()
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)
    return data, off


class Oracle:
    def __init__(self, seqs: Sequence[str], ksize: int):
        self.n = len(seqs)
        data, off = concat(seqs)
        err = C.c_int(0)
        bad_node = C.c_uint32(0)
        bad_char = C.c_uint8(0)
        self._h = lib().peo_create(data.ctypes.data, off.ctypes.data, self.n, ksize,
                                   C.byref(err), C.byref(bad_node), C.byref(bad_char))
        if err.value == 1:
            raise KeyError(chr(bad_char.value))
        if not self._h:
            raise MemoryError("peo_create failed")

    def close(self):
        if self._h:
            lib().peo_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def table_entries(self) -> int:
        return int(lib().peo_table_entries(self._h))

    def map_end(self, read: str) -> List[int]:
        buf = np.frombuffer(read.encode("ascii", "replace"), dtype=np.uint8)
        out = np.zeros(max(self.n, 1), dtype=np.uint32)
        k = lib().peo_map_end(self._h, buf.ctypes.data if buf.size else None, buf.size,
                              out.ctypes.data, out.size)
        return out[:k].tolist()

    def count_pairs(self, fwd: Sequence[str], rve: Sequence[str]):
        n_pairs = min(len(fwd), len(rve))
        fd, fo = concat(fwd[:n_pairs])
        rd, ro = concat(rve[:n_pairs])
        return self.count_pairs_raw(fd, fo, rd, ro, n_pairs)

    def count_pairs_raw(self, fd, fo, rd, ro, n_pairs, node_mat=None, short_mat=None, stats=None):
        if node_mat is None:
            node_mat = np.zeros((self.n, self.n), dtype=np.int64)
            short_mat = np.zeros((self.n, self.n), dtype=np.int64)
            stats = np.zeros(3, dtype=np.uint64)
        lib().peo_count_pairs(self._h, fd.ctypes.data, fo.ctypes.data, rd.ctypes.data,
                              ro.ctypes.data, n_pairs, node_mat.ctypes.data,
                              short_mat.ctypes.data, stats.ctypes.data)
        return node_mat, short_mat, stats


def _sparse(self, fd, fo, rd, ro, n_pairs):
    """Sparse form of count_pairs_raw for graphs whose dense matrices do not fit the host:
    -> (node cells, node counts, short cells, short counts, stats); a cell is i * n + j."""
    stats = np.zeros(3, dtype=np.uint64)
    cap = max(1 << 20, 96 * int(n_pairs))
    while True:
        keys = np.empty(cap, dtype=np.uint64)
        st = np.zeros(3, dtype=np.uint64)
        m = int(lib().peo_count_pairs_keys(self._h, fd.ctypes.data, fo.ctypes.data, rd.ctypes.data, ro.ctypes.data,
                                           n_pairs, keys.ctypes.data, cap, st.ctypes.data))
        if m <= cap:
            stats = st
            break
        cap = m
    cells, counts = np.unique(keys[:m], return_counts=True)
    short = cells >= np.uint64(1 << 63)
    return (cells[~short].astype(np.int64), counts[~short].astype(np.int64),
            (cells[short] - np.uint64(1 << 63)).astype(np.int64), counts[short].astype(np.int64), stats)


Oracle.count_pairs_sparse = _sparse


def synth_pairs(genomes: Sequence[str], cum: np.ndarray, seed: int, first_pair: int, n: int,
                read_len: int, sub_thresh: int, n_thresh: int):
    """CPU twin of vs_synth_pairs -> (fwd bytes [n, L], rve bytes [n, L])."""
    gd, go = concat(genomes)
    cum = np.ascontiguousarray(cum, dtype=np.uint32)
    fwd = np.zeros((n, read_len), dtype=np.uint8)
    rve = np.zeros((n, read_len), dtype=np.uint8)
    lib().peo_synth_pairs(gd.ctypes.data, go.ctypes.data, cum.ctypes.data, len(genomes), seed,
                          first_pair, n, read_len, sub_thresh, n_thresh, fwd.ctypes.data,
                          rve.ctypes.data)
    return fwd, rve
