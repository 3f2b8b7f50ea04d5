"""Device side of the graph stages: ``ops.GraphOps`` / ``ops.PeLinks`` on top of the C ABI
(``include/vstrains_hip.h``, "graph stages").  No CPU path: constructing ``HipBackend`` without a
usable HIP device raises ``NativeError``."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _timing as _tm

from .. import _native as nat
from .asm_graph import AsmGraph
from .ops import GraphOps, GraphScan, LiveLinks, PeLinks


def _ptr(a: np.ndarray):
    return a.ctypes.data if a.size else None


class HipGraphOps(GraphOps):
    """K6 + K7: one ``vs_graph_refresh`` call computes the flows and the scan of a graph snapshot
    (the reference recomputes flows after every re-initialisation and asks for the branch /
    simple-edge facts of the same snapshot right after: ``refresh``)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.calls = 0
        self.reinit_s = 0.0     # seconds spent in reinit() since the object was made, of which in the library call:
        self.native_s = 0.0
        self.reinit_calls = 0
        self._seg_lines: Dict[str, tuple] = {}   # id -> (depth, "S ..." line)
        self._link_lines: Dict[tuple, tuple] = {}  # (id, id) -> (overlap, "L ..." line)
        from . import fast_module

        self._fast = fast_module("_stage_fast")   # typed Cython front half of reinit, or None
        self._dp_repr: Dict[float, bytes] = {}     # depth -> repr(depth) as it goes into a segment line

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

    def reinit(self, g: AsmGraph, nodes, edges, gfa_path: str):
        import time

        t0 = time.perf_counter()
        try:
            return self._reinit(g, nodes, edges, gfa_path)
        finally:
            self.reinit_s += time.perf_counter() - t0
            self.reinit_calls += 1

    def _reinit(self, g: AsmGraph, nodes, edges, gfa_path: str):
        """``formats.stage_graph_from_state`` + ``refresh`` with the per-edge work in the library
        (``vs_stage_rebuild``): this side filters the surviving vertices and edges (map order, by
        name, as the reference's ``graph_to_gfa`` does, IO.py:345-369), writes the stage GFA, and
        gets back the adjacency in the container's order together with the flows and the scan.
        -> (graph, node map, edge map, GFA text, scan)"""
        from .asm_graph import BLACK

        fast = self._fast
        if _tm.ON:
            _t0 = _tm.now()
        if fast is not None:
            try:
                (n_vid, n_vdp, n_vseq, nn, kept_keys, src, tgt, ovl, a_src, a_tgt, a_dp, text) = fast.prepare(
                    g.vblack, g.vid, g.vdp, g.vseq, g.eblack, g.eovl, g.esrc, g.etgt, nodes, edges, self._dp_repr)
            except TypeError:  # (ids that are not str, overlaps that are not int: the Python statement handles them)
                fast = None
        if fast is not None:
            if len(self._dp_repr) > 1 << 20:
                self._dp_repr.clear()
            nv, n_e = len(n_vid), len(src)
            if _tm.ON:
                _t = _tm.add("reinit.prepare", _t0)
            with open(gfa_path, "wb") as fh:
                fh.write(text)
            if _tm.ON:
                _tm.add("reinit.write_file", _t)
            return self._rebuild(n_vid, n_vdp, n_vseq, nn, kept_keys, src, tgt, ovl, a_src, a_tgt, a_dp, text)
        vblack, vid, vdp, vseq = g.vblack, g.vid, g.vdp, g.vseq
        keep = [v for v in nodes.values() if vblack[v]]
        n_vid = [vid[v] for v in keep]
        n_vdp = [vdp[v] for v in keep]
        n_vseq = [vseq[v] for v in keep]
        nv = len(keep)
        nn = dict(zip(n_vid, range(nv)))
        # GFA lines: most vertices and edges of a stage were there, unchanged, in the stage before --
        # their lines are kept (a segment line by id, checked against depth and sequence object)
        sl = self._seg_lines
        chunks = []
        add = chunks.append
        for t in zip(n_vid, n_vseq, n_vdp):
            hit = sl.get(t[0])
            if hit is None or hit[0] != t[2] or hit[2] is not t[1]:  # (same depth, the very same sequence object)
                hit = (t[2], "S\t%s\t%s\tDP:f:%r\n" % t, t[1])
                sl[t[0]] = hit
            add(hit[1])
        eblack, eovl_src = g.eblack, g.eovl
        get = nn.get
        src: List[int] = []
        tgt: List[int] = []
        ovl: List[int] = []
        kept_keys = []
        for key, e in edges.items():
            s_ = get(key[0])
            t_ = get(key[1])
            if s_ is None or t_ is None or not eblack[e]:
                continue
            src.append(s_)
            tgt.append(t_)
            ovl.append(eovl_src[e])
            kept_keys.append(key)
        n_e = len(src)
        ll = self._link_lines
        for k, o in zip(kept_keys, ovl):
            hit = ll.get(k)
            if hit is None or hit[0] != o:
                hit = (o, "L\t%s\t+\t%s\t+\t%dM\n" % (k[0], k[1], o))
                ll[k] = hit
            add(hit[1])
        if len(sl) > 8 * nv + 4096:  # (ids of retired nodes pile up over a run: start over now and then)
            sl.clear()
        if len(ll) > 8 * n_e + 4096:
            ll.clear()
        text = "".join(chunks)
        with open(gfa_path, "w") as fh:
            fh.write(text)
        a_src = np.asarray(src, dtype=np.uint32)
        a_tgt = np.asarray(tgt, dtype=np.uint32)
        a_dp = np.asarray(n_vdp, dtype=np.float64)
        return self._rebuild(n_vid, n_vdp, n_vseq, nn, kept_keys, src, tgt, ovl, a_src, a_tgt, a_dp, text)

    def _rebuild(self, n_vid, n_vdp, n_vseq, nn, kept_keys, src, tgt, ovl, a_src, a_tgt, a_dp, text):
        """Back half of ``reinit``: adjacency, flows and scan of the filtered stage (``vs_stage_rebuild``),
        unpacked into a fresh ``AsmGraph``."""
        from .asm_graph import BLACK

        nv, n_e = len(n_vid), len(src)
        if _tm.ON:
            _tr = _tm.now()
        row_ptr = np.zeros(nv + 1, dtype=np.uint64)
        n_out = np.zeros(max(nv, 1), dtype=np.uint32)
        nbr = np.zeros(max(2 * n_e, 1), dtype=np.uint32)
        eidx = np.zeros(max(2 * n_e, 1), dtype=np.uint32)
        flow = np.zeros(max(n_e, 1), dtype=np.float64)
        nt = np.zeros(max(nv, 1), dtype=np.uint8)
        fk = np.zeros(max(nv, 1), dtype=np.uint8)
        nxt = np.full(max(nv, 1), -1, dtype=np.int32)
        top = np.zeros(max(nv, 1), dtype=np.int32)
        rank = np.zeros(max(nv, 1), dtype=np.int32)
        bad = C.c_uint32(0xFFFFFFFF)
        import time

        t0 = time.perf_counter()
        nat.check(self.ctx._h, nat.lib().vs_stage_rebuild(
            self.ctx._h, nv, n_e, _ptr(a_src), _ptr(a_tgt), _ptr(a_dp), row_ptr.ctypes.data, n_out.ctypes.data,
            nbr.ctypes.data, eidx.ctypes.data, flow.ctypes.data, nt.ctypes.data, fk.ctypes.data, nxt.ctypes.data,
            top.ctypes.data, rank.ctypes.data, C.byref(bad)))
        self.native_s += time.perf_counter() - t0
        self.calls += 1
        if _tm.ON:
            _t = _tm.add("reinit.arrays_and_native", _tr)
        ng = AsmGraph()
        ng.vid, ng.vdp, ng.vseq = n_vid, n_vdp, n_vseq
        ng.vblack = [BLACK] * nv
        pairs = list(zip(nbr[: 2 * n_e].tolist(), eidx[: 2 * n_e].tolist()))
        ptr = row_ptr.tolist()
        ng.adj = [pairs[a:b] for a, b in zip(ptr[:-1], ptr[1:])]
        ng.nout = n_out[:nv].tolist()
        ng.esrc, ng.etgt, ng.eovl = src, tgt, ovl
        ng.eflow = flow[:n_e].tolist()
        ng.eblack = [BLACK] * n_e
        ng._n_edges = n_e
        if bad.value != 0xFFFFFFFF:
            raise FloatingPointError("divide by zero encountered in edge flow of edge %s -> %s"
                                     % (n_vid[src[bad.value]], n_vid[tgt[bad.value]]))
        ne = dict(zip(kept_keys, range(n_e)))
        scan = self._as_scan(ng, nt, fk, nxt, top, rank)
        if _tm.ON:
            _tm.add("reinit.unpack", _t)
        return ng, nn, ne, text, scan

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
    def from_counter(cls, ctx, counter, names: Sequence[str]):
        """``counter``: ``pe.PeCounter`` whose [2,N,N] int32 tensor ``vs_pe_count`` filled."""
        n = len(names)
        assert counter.n == n
        counter.torch.cuda.synchronize(counter.device)
        h = C.c_void_p()
        if getattr(counter, "wide", None) is not None:  # int64 totals exist: everything goes there first
            counter.fold()
            counter.torch.cuda.synchronize(counter.device)
            nat.check(ctx._h, nat.lib().vs_links_from_wide(ctx._h, C.c_void_p(counter.wide[0].data_ptr()),
                                                           C.c_void_p(counter.wide[1].data_ptr()), n, C.byref(h)))
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
            self.pe_stats = pe_inference.run(gfa, aln_dir, fwd, rve, ksize, ctx=self.ctx)
            ids, counter = pe_inference.run.last
        else:
            os.makedirs(aln_dir, exist_ok=True)
            ids, counter = pe_inference.count_links(self.ctx, gfa, fwd, rve, ksize)
        if list(ids) != list(names):
            raise RuntimeError("node order of %s differs from the stage graph" % gfa)
        return HipPeLinks.from_counter(self.ctx, counter, names)

    def links_from_files(self, names: Sequence[str], pe_file: str, st_file: str) -> HipPeLinks:
        """``process_pe_info`` (IO.py:598-627) from the two text files, table resident on the device."""
        return HipPeLinks.from_files(self.ctx, list(names), pe_file, st_file)

    def live_links(self, table: PeLinks) -> LiveLinks:
        return LiveLinks(table)

    def native_stage(self, table: HipPeLinks):
        """The stage graph in the library (``vs_stage``): what ``pipeline.extract_strains`` runs on."""
        from .native_stage import NativeStage

        return NativeStage.on_device(self.ctx, table, table.names)
