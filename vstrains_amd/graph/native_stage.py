"""Python face of the native stage handle (``vs_stage``, include/vstrains_hip.h "graph stages: native stage handle").

The graph, both ordered maps, the contigs, the PE-link bookkeeping and every stage decision live in the library
(``csrc/vs_stage.cpp``); this module only moves state across the C ABI and forwards the library's log lines.  One method
per reference function (``edge_cleaning``, ``store_reinit_graph``, ``iter_graph_disentanglement``, ``best_matching``,
``increment_nt_branch_coverage``, ``path_extension``: utils/VStrains_SPAdes.py:140-248).

Blob layout (little endian; what ``vs_stage_import`` reads and ``vs_stage_export`` writes): a sequence of sections, each
``u32 tag`` + payload, closed by ``u32 0``.  A string list is ``u64 byte count`` + the strings joined by ``\\n``.

* GRAPH (1): ``u32 nv``, ids[nv]; ``u32 n_seq``, sequences[n_seq], ``u32 seq_of[nv]``; ``f64 dp[nv]``, ``u8 black[nv]``;
  adjacency rows ``u32 len[nv]``, ``u32 n_out[nv]``, ``u32 nbr[sum len]``, ``u32 edge[sum len]``; edge slots ``u32 n``,
  ``u32 src[n]``, ``u32 tgt[n]``, ``i64 overlap[n]``, ``f64 flow[n]``, ``u8 black[n]``; free list ``u32 n``, ``u32[n]``;
  ``u32 live edges``; node map ``u32 n``, ids[n], ``u32 vertex[n]``; edge map ``u32 n``, source ids[n], target ids[n],
  ``u32 edge[n]``.
* CONTIGS (2) / STRAINS (8): ``u32 n``, names[n], ``i64 length[n]``, ``f64 coverage[n]``, ``u8 numpy_float[n]``,
  ``u32 count[n]``, ids[sum count].
* LINKS (4): ``u32 n``, branch ids[n], ``u32 count[n]``, in ids[sum], out ids[sum], ``i64 pe[sum]``.
* USAGES (16): ``u32 n``, ids[n], ``i64[n]``.   LOG (32): ``u32 n``, ``i32 level[n]``, lines[n].
* ASSIGNED (128): ``u32 n``, source ids[n], target ids[n], ``u8 flag[n]``.
* SCAN (64): ``u32 nv``, ``u8 nontrivial[nv]``, ``u8 fork_kind[nv]``, ``i32 chain_next / chain_top / chain_rank [nv]``.
"""
from __future__ import annotations

import ctypes as C
import struct
from operator import itemgetter
from typing import Dict, List, Optional, Tuple

import numpy as np

from .asm_graph import AsmGraph

GRAPH, CONTIGS, LINKS, STRAINS, USAGES, LOG, SCAN, ASSIGNED = 1, 2, 4, 8, 16, 32, 64, 128

STAGE_SYMBOLS = {
    "vs_stage_destroy": (None, [C.c_void_p]),
    "vs_stage_error": (C.c_char_p, [C.c_void_p, C.POINTER(C.c_char_p)]),
    "vs_stage_set_debug": (C.c_int, [C.c_void_p, C.c_int]),
    "vs_stage_set_link_names": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint64]),
    "vs_stage_import": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64]),
    "vs_stage_export": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "vs_stage_link": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "vs_stage_edge_cleaning": (C.c_int, [C.c_void_p]),
    "vs_stage_reinit": (C.c_int, [C.c_void_p, C.c_char_p]),
    "vs_stage_refresh_scan": (C.c_int, [C.c_void_p]),
    "vs_stage_disentangle": (C.c_int, [C.c_void_p, C.c_double, C.c_char_p]),
    "vs_stage_best_matching": (C.c_int, [C.c_void_p]),
    "vs_stage_increment_nt_coverage": (C.c_int, [C.c_void_p]),
    "vs_stage_write_gfa": (C.c_int, [C.c_void_p, C.c_char_p]),
    "vs_stage_write_contigs": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "vs_stage_path_extension": (C.c_int, [C.c_void_p, C.c_double, C.c_char_p]),
    "vs_stage_keep_graph": (C.c_int, [C.c_void_p]),
    "vs_stage_finish_strains": (C.c_int, [C.c_void_p, C.c_char_p]),
    "vs_stage_median_depth": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "vs_stage_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "vs_stage_sections": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64]),
}

_EXCEPTIONS = {"KeyError": KeyError, "IndexError": IndexError, "ValueError": ValueError, "FloatingPointError": FloatingPointError,
               "OSError": OSError, "MemoryError": MemoryError, "RecursionError": RecursionError}


def bind(lib) -> None:
    """Set the prototypes of the handle's entry points on a loaded library (the product library, or the test library
    that holds the same engine over the CPU checker)."""
    for name, (res, args) in STAGE_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def _strings(items) -> bytes:
    body = "\n".join(items).encode("utf-8")
    return struct.pack("<Q", len(body)) + body


def _arr(values, dtype) -> bytes:
    return np.asarray(values, dtype=dtype).tobytes()


class _Reader:
    def __init__(self, buf: bytes):
        self.buf = buf
        self.at = 0

    def u32(self) -> int:
        v = struct.unpack_from("<I", self.buf, self.at)[0]
        self.at += 4
        return v

    def arr(self, n: int, dtype):
        a = np.frombuffer(self.buf, dtype=dtype, count=n, offset=self.at)
        self.at += a.nbytes
        return a

    def strings(self, n: int) -> List[str]:
        size = struct.unpack_from("<Q", self.buf, self.at)[0]
        self.at += 8
        body = self.buf[self.at:self.at + size].decode("utf-8")
        self.at += size
        if n == 0:
            return []
        out = body.split("\n")
        assert len(out) == n, (len(out), n)
        return out


def pack_graph(g: AsmGraph, nodes: Dict[str, int], edges: Dict[Tuple[str, str], int]) -> bytes:
    from itertools import chain

    nv = len(g.vid)
    seq_index: Dict[str, int] = {}
    seq_of = [seq_index.setdefault(s, len(seq_index)) for s in g.vseq]
    lens = np.fromiter(map(len, g.adj), dtype="<u4", count=nv)
    # (neighbour, edge) pairs of all rows, flattened in C: one pass over the tuples
    flat = np.fromiter(chain.from_iterable(chain.from_iterable(g.adj)), dtype="<u4", count=2 * int(lens.sum()))
    n_slots = len(g.esrc)
    parts = [struct.pack("<II", GRAPH, nv), _strings(g.vid), struct.pack("<I", len(seq_index)), _strings(seq_index.keys()),
             _arr(seq_of, "<u4"), _arr(g.vdp, "<f8"), _arr(g.vblack, "u1"), lens.tobytes(), _arr(g.nout, "<u4"),
             np.ascontiguousarray(flat[0::2]).tobytes(), np.ascontiguousarray(flat[1::2]).tobytes(),
             struct.pack("<I", n_slots), _arr(g.esrc, "<u4"), _arr(g.etgt, "<u4"), _arr(g.eovl, "<i8"), _arr(g.eflow, "<f8"),
             _arr(g.eblack, "u1"), struct.pack("<I", len(g._free)), _arr(list(g._free), "<u4"), struct.pack("<I", g._n_edges),
             struct.pack("<I", len(nodes)), _strings(nodes.keys()), _arr(list(nodes.values()), "<u4"),
             struct.pack("<I", len(edges)), _strings(map(itemgetter(0), edges)), _strings(map(itemgetter(1), edges)),
             _arr(list(edges.values()), "<u4")]
    return b"".join(parts)


def pack_scan(scan) -> bytes:
    nv = len(scan.nontrivial)
    return b"".join([struct.pack("<II", SCAN, nv), _arr(scan.nontrivial, "u1"), _arr(scan.fork_kind, "u1"), _arr(scan.chain_next, "<i4"),
                     _arr(scan.chain_top, "<i4"), _arr(scan.chain_rank, "<i4")])


def pack_contigs(contigs: Dict[str, list], tag: int = CONTIGS) -> bytes:
    names = list(contigs.keys())
    recs = list(contigs.values())
    return b"".join([struct.pack("<II", tag, len(names)), _strings(names), _arr([r[1] for r in recs], "<i8"),
                     _arr([float(r[2]) for r in recs], "<f8"), _arr([isinstance(r[2], np.floating) for r in recs], "u1"),
                     _arr([len(r[0]) for r in recs], "<u4"), _strings([n for r in recs for n in r[0]])])


def pack_link_table(table: Dict[str, Dict[Tuple[str, str], int]]) -> bytes:
    nos = list(table.keys())
    links = [(u, w, pe) for kept in table.values() for (u, w), pe in kept.items()]
    return b"".join([struct.pack("<II", LINKS, len(nos)), _strings(nos), _arr([len(k) for k in table.values()], "<u4"),
                     _strings([l[0] for l in links]), _strings([l[1] for l in links]), _arr([l[2] for l in links], "<i8")])


def _read_graph(r: _Reader):
    nv = r.u32()
    g = AsmGraph()
    g.vid = r.strings(nv)
    n_seq = r.u32()
    seqs = r.strings(n_seq)
    g.vseq = [seqs[i] for i in r.arr(nv, "<u4").tolist()]
    g.vdp = r.arr(nv, "<f8").tolist()
    g.vblack = [bool(x) for x in r.arr(nv, "u1").tolist()]
    lens = r.arr(nv, "<u4").tolist()
    g.nout = r.arr(nv, "<u4").tolist()
    tot = sum(lens)
    nbr = r.arr(tot, "<u4").tolist()
    eidx = r.arr(tot, "<u4").tolist()
    g.adj = []
    at = 0
    for n in lens:
        g.adj.append(list(zip(nbr[at:at + n], eidx[at:at + n])))
        at += n
    n_slots = r.u32()
    g.esrc = r.arr(n_slots, "<u4").tolist()
    g.etgt = r.arr(n_slots, "<u4").tolist()
    g.eovl = r.arr(n_slots, "<i8").tolist()
    g.eflow = r.arr(n_slots, "<f8").tolist()
    g.eblack = [bool(x) for x in r.arr(n_slots, "u1").tolist()]
    n_free = r.u32()
    g._free.extend(r.arr(n_free, "<u4").tolist())
    g._n_edges = r.u32()
    n_nodes = r.u32()
    names = r.strings(n_nodes)
    nodes = dict(zip(names, r.arr(n_nodes, "<u4").tolist()))
    n_em = r.u32()
    eu = r.strings(n_em)
    ew = r.strings(n_em)
    edges = dict(zip(zip(eu, ew), r.arr(n_em, "<u4").tolist()))
    return g, nodes, edges


def _read_contigs(r: _Reader) -> Dict[str, list]:
    n = r.u32()
    names = r.strings(n)
    lens = r.arr(n, "<i8").tolist()
    covs = r.arr(n, "<f8").tolist()
    flags = r.arr(n, "u1").tolist()
    counts = r.arr(n, "<u4").tolist()
    ids = r.strings(sum(counts))
    out: Dict[str, list] = {}
    at = 0
    for i in range(n):
        # (a coverage that came out of numpy.median is a numpy.float64 in the reference, whose round() is numpy's)
        out[names[i]] = [ids[at:at + counts[i]], lens[i], np.float64(covs[i]) if flags[i] else covs[i]]
        at += counts[i]
    return out


def _read_link_table(r: _Reader):
    n = r.u32()
    nos = r.strings(n)
    counts = r.arr(n, "<u4").tolist()
    tot = sum(counts)
    us = r.strings(tot)
    ws = r.strings(tot)
    pes = r.arr(tot, "<i8").tolist()
    out = {}
    at = 0
    for i in range(n):
        out[nos[i]] = {(us[k], ws[k]): pes[k] for k in range(at, at + counts[i])}
        at += counts[i]
    return out


class NativeStage:
    """One ``vs_stage`` handle.  ``lib``: the library that holds its entry points; ``keep``: objects the handle refers
    to (context, link table) that must outlive it."""

    def __init__(self, lib, handle, keep=()):
        self._lib = lib
        self._h = handle
        self._keep = keep

    # ---- construction
    @classmethod
    def on_device(cls, ctx, links, names):
        """Product path: device operations on ``ctx`` (``pe.Context``), PE-link table ``links`` (``hip_ops.HipPeLinks``)."""
        from .. import _native as nat

        L = nat.lib()
        h = C.c_void_p()
        nat.check(ctx._h, L.vs_stage_create(ctx._h, links._h, C.byref(h)))
        st = cls(L, h, keep=(ctx, links))
        st.set_link_names(links.names)
        return st

    def close(self):
        if self._h:
            self._lib.vs_stage_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _error(self, rc: int):
        """The exception the reference raises in the situation the library reports (the library names it)."""
        kind = C.c_char_p()
        msg = self._lib.vs_stage_error(self._h, C.byref(kind))
        text = msg.decode("utf-8", "replace") if msg else "?"
        exc = _EXCEPTIONS.get(kind.value.decode() if kind.value else "", None)
        if exc is None:
            return RuntimeError("vs_stage: %s (code %d)" % (text, rc))
        return exc(text)

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise self._error(rc)

    def _call(self, fn, logger, *args) -> None:
        rc = fn(self._h, *args)
        err = self._error(rc) if rc != 0 else None  # (read before the next call on the handle clears it)
        if logger is not None:
            self.drain_log(logger)
        if err is not None:
            raise err

    # ---- state in
    def set_link_names(self, names) -> None:
        body = "\n".join(names).encode("utf-8")
        self._check(self._lib.vs_stage_set_link_names(self._h, len(names), body, len(body)))

    def set_debug(self, on: bool) -> None:
        self._check(self._lib.vs_stage_set_debug(self._h, 1 if on else 0))

    def _import(self, blob: bytes) -> None:
        blob += struct.pack("<I", 0)
        self._check(self._lib.vs_stage_import(self._h, blob, len(blob)))

    def load_graph(self, g: AsmGraph, nodes, edges, scan=None) -> None:
        self._import(pack_graph(g, nodes, edges) + (pack_scan(scan) if scan is not None else b""))

    def load_contigs(self, contigs) -> None:
        self._import(pack_contigs(contigs))

    def load_full_link(self, table) -> None:
        self._import(pack_link_table(table))

    # ---- state out
    def _export(self, what: int) -> _Reader:
        ptr = C.c_void_p()
        size = C.c_uint64()
        self._check(self._lib.vs_stage_export(self._h, what, C.byref(ptr), C.byref(size)))
        return _Reader(C.string_at(ptr.value, size.value))

    def drain_log(self, logger) -> None:
        r = self._export(LOG)
        assert r.u32() == LOG
        n = r.u32()
        levels = r.arr(n, "<i4").tolist()
        for level, line in zip(levels, r.strings(n)):
            logger.log(level, line)

    def graph(self):
        """-> (AsmGraph, node map, edge map): a copy of the handle's graph state."""
        r = self._export(GRAPH)
        assert r.u32() == GRAPH
        return _read_graph(r)

    def contigs(self) -> Dict[str, list]:
        r = self._export(CONTIGS)
        assert r.u32() == CONTIGS
        return _read_contigs(r)

    def contigs_into(self, contigs: Dict[str, list]) -> None:
        """The reference mutates ``contig_dict`` in place: the caller's dict receives the handle's records."""
        fresh = self.contigs()
        contigs.clear()
        contigs.update(fresh)

    def full_link(self):
        r = self._export(LINKS)
        assert r.u32() == LINKS
        return _read_link_table(r)

    def strains(self) -> Dict[str, list]:
        r = self._export(STRAINS)
        assert r.u32() == STRAINS
        return _read_contigs(r)

    def usages(self) -> Dict[str, int]:
        r = self._export(USAGES)
        assert r.u32() == USAGES
        n = r.u32()
        names = r.strings(n)
        return dict(zip(names, r.arr(n, "<i8").tolist()))

    def scan(self):
        from .ops import GraphScan

        r = self._export(SCAN)
        assert r.u32() == SCAN
        nv = r.u32()
        return GraphScan(r.arr(nv, "u1").astype(bool).tolist(), r.arr(nv, "u1").tolist(), r.arr(nv, "<i4").tolist(),
                         r.arr(nv, "<i4").tolist(), r.arr(nv, "<i4").tolist())

    def assigned(self) -> Dict[Tuple[str, str], bool]:
        """``edge_cleaning``'s result (Decomposition.py:822-905)."""
        r = self._export(ASSIGNED)
        assert r.u32() == ASSIGNED
        n = r.u32()
        eu, ew = r.strings(n), r.strings(n)
        return dict(zip(zip(eu, ew), [bool(x) for x in r.arr(n, "u1").tolist()]))

    def link(self, a: str, b: str) -> int:
        """``pe_info[(min(a, b), max(a, b))]`` as the reference's rewritten dict would hold it now."""
        out = C.c_int64()
        self._check(self._lib.vs_stage_link(self._h, a.encode(), b.encode(), C.byref(out)))
        return int(out.value)

    def median_depth(self) -> np.float64:
        out = C.c_double()
        self._check(self._lib.vs_stage_median_depth(self._h, C.byref(out)))
        return np.float64(out.value)

    def counters(self) -> dict:
        info = (C.c_uint64 * 8)()
        secs = (C.c_double * 4)()
        self._check(self._lib.vs_stage_counters(self._h, info, secs))
        return {"reinit_calls": int(info[0]), "reinit_reused": int(info[1]), "graph_refresh_launches": int(info[2]),
                "link_table_launches": int(info[3]), "files_written": int(info[4]), "bytes_written": int(info[5]),
                "vertices": int(info[6]), "edges": int(info[7]), "reinit_s": secs[0], "refresh_op_s": secs[1],
                "link_op_s": secs[2], "file_writer_busy_s": secs[3]}

    def sections(self) -> Dict[str, float]:
        buf = C.create_string_buffer(4096)
        self._check(self._lib.vs_stage_sections(self._h, buf, 4096))
        return {k: float(v) for k, v in (item.split("=") for item in buf.value.decode().split(";") if item)}

    # ---- the stages
    def edge_cleaning(self, logger=None) -> None:
        self._call(self._lib.vs_stage_edge_cleaning, logger)

    def reinit(self, filename: str, logger=None) -> None:
        self._call(self._lib.vs_stage_reinit, logger, filename.encode())

    def refresh_scan(self) -> None:
        self._check(self._lib.vs_stage_refresh_scan(self._h))

    def disentangle(self, threshold, temp_dir: str, logger=None) -> None:
        self._call(self._lib.vs_stage_disentangle, logger, float(threshold), temp_dir.encode())

    def best_matching(self, logger=None) -> None:
        self._call(self._lib.vs_stage_best_matching, logger)

    def increment_nt_branch_coverage(self, logger=None) -> None:
        self._call(self._lib.vs_stage_increment_nt_coverage, logger)

    def write_gfa(self, filename: str, logger=None) -> None:
        self._call(self._lib.vs_stage_write_gfa, logger, filename.encode())

    def write_contigs(self, paths_file: Optional[str], fasta_file: Optional[str]) -> None:
        self._check(self._lib.vs_stage_write_contigs(self._h, paths_file.encode() if paths_file else None,
                                                     fasta_file.encode() if fasta_file else None))

    def keep_graph(self) -> None:
        """Remember the graph as it stands (es_graph_L2): ``finish_strains`` measures the strain records on it."""
        self._check(self._lib.vs_stage_keep_graph(self._h))

    def finish_strains(self, tmp_paths_file: str, logger=None) -> None:
        """VStrains_SPAdes.py:251-262: resolve, trim on es_graph_L2, drop duplicates, ``tmp/tmp_strain.paths``."""
        self._call(self._lib.vs_stage_finish_strains, logger, tmp_paths_file.encode())

    def path_extension(self, threshold, temp_dir: str, logger=None) -> None:
        self._call(self._lib.vs_stage_path_extension, logger, float(threshold), temp_dir.encode())
