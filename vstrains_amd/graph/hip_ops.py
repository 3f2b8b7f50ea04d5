"""Device side of the graph stages: ``ops.GraphOps`` / ``ops.PeLinks`` on top of the C ABI
(``include/vstrains_hip.h``, "graph stages").  No CPU path: constructing ``HipBackend`` without a
usable HIP device raises ``NativeError``."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _native as nat
from .asm_graph import AsmGraph
from .ops import GraphOps, GraphScan, PeLinks


def _ptr(a: np.ndarray):
    return a.ctypes.data if a.size else None


class HipGraphOps(GraphOps):
    """K6 + K7 called directly: one ``vs_graph_refresh`` call computes the flows and the scan of a graph snapshot.
    (The stages reach the same kernels from inside the library, through the native stage handle; this wrapper is for
    callers that hold an ``AsmGraph`` -- the kernel tests.)"""

    def __init__(self, ctx):
        self.ctx = ctx
        self.calls = 0

    def _refresh(self, g: AsmGraph):
        nv = g.num_vertices()
        row_ptr, n_out, nbr, eidx = g.csr_arrays()
        a_row = np.asarray(row_ptr, dtype=np.uint64)
        a_no = np.asarray(n_out, dtype=np.uint32)
        a_nbr = np.asarray(nbr, dtype=np.uint32)
        a_eidx = np.asarray(eidx, dtype=np.uint32)
        a_dp = np.asarray(g.vdp, dtype=np.float64)
        a_vb = np.asarray(g.vblack, dtype=np.uint8)
        n_slots = len(g.esrc)
        a_eb = np.asarray(g.eblack, dtype=np.uint8) if n_slots else np.zeros(1, dtype=np.uint8)
        flow = np.zeros(max(n_slots, 1), dtype=np.float64)
        nt = np.zeros(max(nv, 1), dtype=np.uint8)
        fk = np.zeros(max(nv, 1), dtype=np.uint8)
        nxt = np.full(max(nv, 1), -1, dtype=np.int32)
        top = np.zeros(max(nv, 1), dtype=np.int32)
        rank = np.zeros(max(nv, 1), dtype=np.int32)
        bad = C.c_uint32(0xFFFFFFFF)
        nat.check(self.ctx._h, nat.lib().vs_graph_refresh(
            self.ctx._h, nv, n_slots, _ptr(a_row), _ptr(a_no), _ptr(a_nbr), _ptr(a_eidx), _ptr(a_dp), _ptr(a_vb),
            a_eb.ctypes.data, flow.ctypes.data, nt.ctypes.data, fk.ctypes.data, nxt.ctypes.data, top.ctypes.data,
            rank.ctypes.data, C.byref(bad)))
        self.calls += 1
        return flow, nt, fk, nxt, top, rank, bad.value

    @staticmethod
    def _apply_flows(g: AsmGraph, flow, bad) -> None:
        if bad != 0xFFFFFFFF:
            # numpy.seterr(all="raise") in the reference's main process (vstrains:25)
            raise FloatingPointError("divide by zero encountered in edge flow of edge %s -> %s"
                                     % (g.vid[g.esrc[bad]], g.vid[g.etgt[bad]]))
        vals = flow.tolist()
        for e in g.edges():
            g.eflow[e] = vals[e]

    @staticmethod
    def _as_scan(g: AsmGraph, nt, fk, nxt, top, rank) -> GraphScan:
        nv = g.num_vertices()
        return GraphScan(nt[:nv].astype(bool).tolist(), fk[:nv].tolist(), nxt[:nv].tolist(), top[:nv].tolist(),
                         rank[:nv].tolist())

    def edge_flows(self, g: AsmGraph) -> None:
        flow, _, _, _, _, _, bad = self._refresh(g)
        self._apply_flows(g, flow, bad)

    def scan(self, g: AsmGraph) -> GraphScan:
        _, nt, fk, nxt, top, rank, _ = self._refresh(g)
        return self._as_scan(g, nt, fk, nxt, top, rank)

    def refresh(self, g: AsmGraph) -> GraphScan:
        flow, nt, fk, nxt, top, rank, bad = self._refresh(g)
        self._apply_flows(g, flow, bad)
        return self._as_scan(g, nt, fk, nxt, top, rank)


class HipPeLinks(PeLinks):
    """K5: the symmetrised PE-link matrix, resident in HBM."""

    def __init__(self, ctx, handle, names: Sequence[str], caller_names: Optional[Sequence[str]] = None):
        """``names``: row i of the device table belongs to names[i]; ``caller_names``: the same names in the order the
        caller knows them (``to_numpy`` answers in that order), when the two differ."""
        self.ctx = ctx
        self._h = handle
        self.names = list(names)
        self._index = {n: i for i, n in enumerate(self.names)}
        self._caller_rows = None if caller_names is None else np.asarray([self._index[n] for n in caller_names], dtype=np.int64)
        self.calls = 0

    @classmethod
    def from_counter(cls, ctx, counter, names: Sequence[str], sparse_min_nodes: int = 0):
        """``counter``: ``pe.PeCounter`` whose [2,N,N] int32 tensor ``vs_pe_count`` filled.  ``sparse_min_nodes``: from how
        many nodes a counter with a dirty-tile map gives a CSR table (0: the library's default, 32 768; tests pass 64)."""
        n = len(names)
        assert counter.n == n
        counter.torch.cuda.synchronize(counter.device)
        h = C.c_void_p()
        if getattr(counter, "wide", None) is not None:  # int64 totals exist: everything goes there first
            counter.fold()
            counter.torch.cuda.synchronize(counter.device)
            nat.check(ctx._h, nat.lib().vs_links_from_wide(ctx._h, C.c_void_p(counter.wide[0].data_ptr()),
                                                           C.c_void_p(counter.wide[1].data_ptr()), n, C.byref(h)))
        elif getattr(counter, "tile_map", None) is not None:
            # (ABI 10) counters that keep a dirty-tile map: from 32 768 nodes on the table is built from the marked tiles
            # only and held as CSR rows of its non-zero cells (0.4 GB instead of 23.7 GB at 54 465 nodes)
            nat.check(ctx._h, nat.lib().vs_links_from_counts_tracked(ctx._h, C.c_void_p(counter.mats[0].data_ptr()),
                                                                     C.c_void_p(counter.mats[1].data_ptr()), n,
                                                                     C.c_void_p(counter.tile_map.data_ptr()), sparse_min_nodes, C.byref(h)))
        else:
            nat.check(ctx._h, nat.lib().vs_links_from_counts(ctx._h, C.c_void_p(counter.mats[0].data_ptr()),
                                                             C.c_void_p(counter.mats[1].data_ptr()), n, C.byref(h)))
        ctx.sync()
        # (the counters are in the numbering the index was built in -- pe.Context.build_index -- and so is the table
        # made from them: rows are found by name, so only the name list has to follow)
        order = getattr(counter, "node_order", None)  # (the numbering of the index it counted under)
        if order is not None:
            return cls(ctx, h, [names[i] for i in order.tolist()], caller_names=names)
        return cls(ctx, h, names)

    @classmethod
    def from_matrices(cls, ctx, names: Sequence[str], node_mat, short_mat):
        n = len(names)
        a = np.ascontiguousarray(node_mat, dtype=np.int64).reshape(n, n)
        b = np.ascontiguousarray(short_mat, dtype=np.int64).reshape(n, n)
        h = C.c_void_p()
        nat.check(ctx._h, nat.lib().vs_links_from_host(ctx._h, _ptr(a), _ptr(b), n, C.byref(h)))
        return cls(ctx, h, names)

    @classmethod
    def from_files(cls, ctx, names: Sequence[str], pe_file: str, st_file: str):
        """The reference's own hand-off (IO.py:603-623): parse the two N^2-line text files."""
        from .formats import read_pe_text

        index = {n: i for i, n in enumerate(names)}
        mats = []
        for path in (pe_file, st_file):
            m = np.zeros((len(names), len(names)), dtype=np.int64)
            for u, v, c in read_pe_text(path):
                if u in index and v in index:
                    m[index[u], index[v]] += c
            mats.append(m)
        return cls.from_matrices(ctx, names, mats[0], mats[1])

    def close(self):
        if self._h:
            nat.lib().vs_links_free(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def index_of(self, name: str) -> int:
        return self._index[name]

    def to_numpy(self) -> np.ndarray:
        n = len(self.names)
        out = np.zeros((max(n, 1), max(n, 1)), dtype=np.int64)
        if n:
            nat.check(self.ctx._h, nat.lib().vs_links_to_host(self.ctx._h, self._h, out.ctypes.data))
        out = out[:n, :n]
        if self._caller_rows is not None:
            out = out[np.ix_(self._caller_rows, self._caller_rows)]
        return out

    @staticmethod
    def _pool(lists: Sequence[Sequence[int]]):
        ids: Dict[Tuple[int, ...], int] = {}
        order: List[Tuple[int, ...]] = []
        which = []
        for l in lists:
            t = tuple(l)
            k = ids.get(t)
            if k is None:
                k = ids[t] = len(order)
                order.append(t)
            which.append(k)
        off = np.zeros(len(order) + 1, dtype=np.uint64)
        if order:
            off[1:] = np.cumsum([len(t) for t in order], dtype=np.uint64)
        flat = np.asarray([x for t in order for x in t], dtype=np.uint32)
        return off, flat, which, len(order)

    def block_sums(self, queries):
        if not queries:
            return []
        lists = []
        for rows, cols in queries:
            lists.append(rows)
            lists.append(cols)
        off, flat, which, n_lists = self._pool(lists)
        qa = np.asarray(which[0::2], dtype=np.uint32)
        qb = np.asarray(which[1::2], dtype=np.uint32)
        out = np.zeros(len(queries), dtype=np.int64)
        nat.check(self.ctx._h, nat.lib().vs_links_block_sums(self.ctx._h, self._h, off.ctypes.data, _ptr(flat), n_lists,
                                                             qa.ctypes.data, qb.ctypes.data, len(queries),
                                                             out.ctypes.data))
        self.calls += 1
        return out.tolist()

    def group_matrix(self, groups):
        n = len(groups)
        out = np.zeros((max(n, 1), max(n, 1)), dtype=np.int64)
        if n == 0:
            return out[:0, :0]
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(gp) for gp in groups], dtype=np.uint64)
        flat = np.asarray([x for gp in groups for x in gp], dtype=np.uint32)
        out = np.zeros((n, n), dtype=np.int64)
        nat.check(self.ctx._h, nat.lib().vs_links_group_matrix(self.ctx._h, self._h, off.ctypes.data, _ptr(flat), n,
                                                               out.ctypes.data))
        self.calls += 1
        return out


class HipBackend:
    """What ``pipeline.run`` needs from the device: PE-link inference + the graph kernels."""

    def __init__(self, device: int = 0, write_info_text: bool = True, ctx=None):
        from .. import pe as host

        self.ctx = ctx if ctx is not None else host.Context(device)  # raises NativeError without a HIP device
        self.graph_ops = HipGraphOps(self.ctx)
        self.write_info_text = write_info_text
        self.pe_stats = None

    def pe_links(self, gfa: str, aln_dir: str, fwd: str, rve: str, ksize: int, names: List[str]) -> HipPeLinks:
        """Boundary 1 (VStrains_SPAdes.py:119-138) without the process hop: count on the device,
        write ``pe_info`` / ``st_info`` for drop-in compatibility, and keep the counters in HBM as
        the link table instead of re-parsing N^2 text lines into a dict."""
        from .. import pe_inference

        print("----------------------Paired-End Information Alignment----------------------")
        if self.write_info_text:
            self.pe_stats = pe_inference.run(gfa, aln_dir, fwd, rve, ksize, ctx=self.ctx, stages_follow=True)
            ids, counter = pe_inference.run.last
        else:
            os.makedirs(aln_dir, exist_ok=True)
            ids, counter = pe_inference.count_links(self.ctx, gfa, fwd, rve, ksize, stages_follow=True)
        if list(ids) != list(names):
            raise RuntimeError("node order of %s differs from the stage graph" % gfa)
        return HipPeLinks.from_counter(self.ctx, counter, names)

    def links_from_files(self, names: Sequence[str], pe_file: str, st_file: str) -> HipPeLinks:
        """``process_pe_info`` (IO.py:598-627) from the two text files, table resident on the device."""
        return HipPeLinks.from_files(self.ctx, list(names), pe_file, st_file)

    def native_stage(self, table: HipPeLinks):
        """The stage graph in the library (``vs_stage``): what ``pipeline.extract_strains`` runs on."""
        from .native_stage import NativeStage

        return NativeStage.on_device(self.ctx, table, table.names)
