"""CHECKER for the graph-stage device operations (test infrastructure, not product code).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this.  It restates, in the reference's own terms, the three data-parallel operations that
``vstrains_amd/graph`` sends to the GPU:

* ``assign_edge_flow`` with ``numpy.sum`` / ``numpy.mean`` (``utils/VStrains_Utilities.py:14-31``);
* the vertex scan (``is_non_trivial`` :162-172, the fork tests of
  ``utils/VStrains_Decomposition.py:715,763``, the simple-edge test of ``simp_path``
  Utilities.py:398-402);
* ``pe_info``: the literal dict of ``process_pe_info`` (``utils/VStrains_IO.py:598-627``) with the
  literal rewrites the stages perform on it (Decomposition.py:492-503, :608-617, :672-684;
  Utilities.py:488-499).  ``DictLiveLinks`` replays those rewrites key by key, so that tests can
  check the product's closed form (``vstrains_amd/graph/ops.py``: sums over the original matrix)
  against the dict the reference would hold.

Pinned by ``tests/golden/graph/*`` (outputs of the real reference run behind the test stand-in).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy


class NumpyGraphOps:
    def refresh(self, g):
        self.edge_flows(g)
        return self.scan(g)

    def edge_flows(self, g) -> None:
        for e in g.edges():
            u, v = g.esrc[e], g.etgt[e]
            u_out = numpy.sum([g.vdp[n] for n in g.out_neighbors(u)])
            v_in = numpy.sum([g.vdp[n] for n in g.in_neighbors(v)])
            with numpy.errstate(all="raise"):
                g.eflow[e] = float(numpy.mean([(g.vdp[v] / u_out) * g.vdp[u], (g.vdp[u] / v_in) * g.vdp[v]]))

    def scan(self, g):
        from .graph_stages.model import GraphScan

        nv = g.num_vertices()
        nontrivial = [False] * nv
        fork = [0] * nv
        nxt = [-1] * nv
        has_simple_in = [False] * nv
        for v in range(nv):
            us = [g.esrc[e] for e in g.black_in_edges(v)]
            ws = [g.etgt[e] for e in g.black_out_edges(v)]
            both = len(set(us) & set(ws))
            nontrivial[v] = len(us) > max(both, 1) and len(ws) > max(both, 1)
            if g.vblack[v]:
                if len(us) == 1 and len(ws) > 1:
                    fork[v] = 1
                elif len(us) > 1 and len(ws) == 1:
                    fork[v] = 2
            if g.out_degree(v) == 1:
                t = g.out_neighbors(v)[0]
                if g.in_degree(t) == 1 and t != v:
                    nxt[v] = t
                    has_simple_in[t] = True
        top = list(range(nv))
        rank = [0] * nv
        for v in range(nv):
            if nxt[v] >= 0 and not has_simple_in[v]:
                cur, d = v, 0
                while nxt[cur] >= 0:
                    cur = nxt[cur]
                    d += 1
                    top[cur] = v
                    rank[cur] = d
        for v in range(nv):  # rings of simple edges have no head
            if has_simple_in[v] and top[v] == v:
                rank[v] = -1
        return GraphScan(nontrivial, fork, nxt, top, rank)


class SparsePeLinks:
    """The same table (``process_pe_info`` IO.py:598-627: key {u, v} = both orders of both matrices summed, the diagonal
    once) held as CSR rows of its non-zero cells -- graphs whose dict would have 1.5e9 keys (54 k nodes).  ``row_ptr``
    [n + 1], ``col`` ascending inside a row, ``val`` int64; rows and columns in the order of ``names``."""

    def __init__(self, names: Sequence[str], row_ptr, col, val):
        self.names = list(names)
        self._index = {n: i for i, n in enumerate(self.names)}
        self.row_ptr = numpy.asarray(row_ptr, dtype=numpy.int64)
        self.col = numpy.asarray(col, dtype=numpy.int64)
        self.val = numpy.asarray(val, dtype=numpy.int64)
        self._rows: Dict[int, Dict[int, int]] = {}

    def index_of(self, name: str) -> int:
        return self._index[name]

    def _row(self, r: int) -> Dict[int, int]:
        d = self._rows.get(r)
        if d is None:
            a, b = int(self.row_ptr[r]), int(self.row_ptr[r + 1])
            d = self._rows[r] = dict(zip(self.col[a:b].tolist(), self.val[a:b].tolist()))
        return d

    def cell(self, r: int, c: int) -> int:
        return self._row(r).get(c, 0)

    def block_sums(self, queries):
        out = []
        for rows, cols in queries:
            s = 0
            for r in rows:
                d = self._row(r)
                if d:
                    for c in cols:
                        s += d.get(c, 0)
            out.append(s)
        return out

    def group_matrix(self, groups):
        """Entry (i, j) = block sum of groups[i] x groups[j], worked out when it is asked for (the reference fills a V x V
        table, Extension.py:766-799, and reads the (in-neighbour, out-neighbour) entries of the remaining branches)."""
        table = self

        class _Lazy:
            shape = (len(groups), len(groups))

            def __getitem__(self, ij):
                i, j = ij
                return table.block_sums([(groups[i], groups[j])])[0]

        return _Lazy()


class DictPeLinks:
    """``process_pe_info``: the N(N+1)/2-key dict built from the two count matrices."""

    def __init__(self, names: Sequence[str], node_mat, short_mat):
        self.names = list(names)
        self._index = {n: i for i, n in enumerate(self.names)}
        table: Dict[Tuple[str, str], int] = {}
        for u in self.names:
            for v in self.names:
                table[(min(u, v), max(u, v))] = 0
        for mat in (node_mat, short_mat):
            for i, u in enumerate(self.names):
                row = mat[i]
                for j, v in enumerate(self.names):
                    table[(min(u, v), max(u, v))] += int(row[j])
        self.table = table

    @classmethod
    def from_files(cls, names: Sequence[str], pe_file: str, st_file: str):
        """The reference's own route: parse the two text files (IO.py:603-623)."""
        self = cls.__new__(cls)
        self.names = list(names)
        self._index = {n: i for i, n in enumerate(self.names)}
        table: Dict[Tuple[str, str], int] = {}
        for u in self.names:
            for v in self.names:
                table[(min(u, v), max(u, v))] = 0
        for path in (pe_file, st_file):
            with open(path, "r") as fh:
                for line in fh:
                    if line == "\n":
                        break
                    u, v, mark = line[:-1].split(":")[:3]
                    key = (min(u, v), max(u, v))
                    if table.get(key) is not None:
                        table[key] += int(mark)
        self.table = table
        return self

    def index_of(self, name: str) -> int:
        return self._index[name]

    def block_sums(self, queries):
        out = []
        for rows, cols in queries:
            s = 0
            for r in rows:
                for c in cols:
                    a, b = self.names[r], self.names[c]
                    s += self.table[(min(a, b), max(a, b))]
            out.append(s)
        return out

    def group_matrix(self, groups):
        n = len(groups)
        out = numpy.zeros((n, n), dtype=numpy.int64)
        for i in range(n):
            for j in range(i, n):
                s = self.block_sums([(groups[i], groups[j])])[0]
                out[i, j] = s
                out[j, i] = s
        return out


class DictLiveLinks:
    """The mutable ``pe_info`` of the disentanglement stage, rewritten exactly as the reference
    rewrites it.  Same interface as ``vstrains_amd.graph.ops.LiveLinks``."""

    def __init__(self, base: DictPeLinks):
        self.table: Dict[Tuple[str, str], Optional[int]] = dict(base.table)

    @staticmethod
    def _key(a: str, b: str):
        return (min(a, b), max(a, b))

    def is_fresh(self, name: str) -> bool:
        return self.table[(name, name)] is None

    def prefetch(self, pairs) -> None:
        pass

    def get(self, a: str, b: str):
        return self.table[self._key(a, b)]

    def note_split(self, removed: str, subs: List[str], live_ids: Iterable[str]) -> None:
        for s in subs:
            for n in live_ids:
                self.table[self._key(s, n)] = None
        for pu, pv in list(self.table.keys()):
            if pu == removed or pv == removed:
                self.table.pop((pu, pv))

    def note_fork(self, sub: str, live_ids: Iterable[str]) -> None:
        for n in live_ids:
            self.table[self._key(sub, n)] = None

    def note_drop(self, removed: str) -> None:
        for pu, pv in list(self.table.keys()):
            if pu == removed or pv == removed:
                self.table.pop((pu, pv))

    def note_merge(self, new_id: str, members: List[str], live_ids: Iterable[str]) -> None:
        for n in live_ids:
            key = self._key(new_id, n)
            self.table[key] = 0
            if n != new_id:
                for m in members:
                    self.table[key] += self.table[self._key(m, n)]
        gone = set(members)
        for pu, pv in list(self.table.keys()):
            if pu in gone or pv in gone:
                self.table.pop((pu, pv))

    def end_pass(self) -> None:
        for k in self.table.keys():
            if self.table[k] is None:
                self.table[k] = 0
