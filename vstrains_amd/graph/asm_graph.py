"""Assembly-graph container for the disentanglement / path-extraction stages.

The reference keeps its graph in a ``graph_tool.Graph`` with string/double property maps
(``utils/VStrains_IO.py:13-24``) and its results depend on that library's iteration order
(SURVEY.md 8c).  This is an index-based restatement of the ordering rules, not a wrapper:
vertices and edges are plain ints, properties are parallel Python lists, and the adjacency of a
vertex is ONE list holding its out-entries ``[0, n_out)`` followed by its in-entries, which is
also exactly the CSR the device kernels consume (``csr_arrays``).

Ordering rules (graph-tool ``adj_list`` as recalled; unverified against its source, which is not
available here -- DESIGN.md "parity pins"):

* ``add_edge(s, t)`` places the new out-entry at slot ``n_out`` of ``s``; an in-entry living
  there moves to the back.  The in-entry of ``t`` is appended.
* ``remove_edge`` erases in place; freed edge indices are reused first-in first-out and keep
  whatever property values were stored at them.
* vertex order is index order; ``edges()`` is vertex-major in out-entry order.
"""
from __future__ import annotations

from collections import deque
from itertools import chain
from typing import Deque, Dict, Iterator, List, Optional, Tuple

BLACK = True
GRAY = False


class AsmGraph:
    __slots__ = ("vid", "vseq", "vdp", "vblack", "adj", "nout", "esrc", "etgt", "eovl", "eflow",
                 "eblack", "_free", "_n_edges")

    def __init__(self) -> None:
        self.vid: List[str] = []
        self.vseq: List[str] = []
        self.vdp: List[float] = []
        self.vblack: List[bool] = []
        self.adj: List[List[Tuple[int, int]]] = []  # (neighbour vertex, edge index)
        self.nout: List[int] = []
        self.esrc: List[int] = []
        self.etgt: List[int] = []
        self.eovl: List[int] = []
        self.eflow: List[float] = []
        self.eblack: List[bool] = []
        self._free: Deque[int] = deque()
        self._n_edges = 0

    # ---- construction ------------------------------------------------------------------
    def add_vertex(self, name: str = "UD", dp: float = 0.0, seq: str = "", black: bool = BLACK) -> int:
        self.vid.append(name)
        self.vdp.append(float(dp))
        self.vseq.append(seq)
        self.vblack.append(black)
        self.adj.append([])
        self.nout.append(0)
        return len(self.adj) - 1

    def add_edge(self, s: int, t: int, overlap: Optional[int] = None, flow: Optional[float] = None,
                 black: Optional[bool] = None) -> int:
        """Properties left as ``None`` keep the value already stored at a reused index (a new
        index starts at overlap 0 / flow 0.0 / gray, the property-map defaults of IO.py:20-22)."""
        if self._free:
            e = self._free.popleft()
            self.esrc[e] = s
            self.etgt[e] = t
        else:
            e = len(self.esrc)
            self.esrc.append(s)
            self.etgt.append(t)
            self.eovl.append(0)
            self.eflow.append(0.0)
            self.eblack.append(GRAY)
        row = self.adj[s]
        slot = self.nout[s]
        if slot < len(row):
            row.append(row[slot])
            row[slot] = (t, e)
        else:
            row.append((t, e))
        self.nout[s] = slot + 1
        self.adj[t].append((s, e))
        self._n_edges += 1
        if overlap is not None:
            self.eovl[e] = int(overlap)
        if flow is not None:
            self.eflow[e] = float(flow)
        if black is not None:
            self.eblack[e] = black
        return e

    def remove_edge(self, e: int) -> None:
        s, t = self.esrc[e], self.etgt[e]
        row = self.adj[s]
        row.pop(row.index((t, e), 0, self.nout[s]))
        self.nout[s] -= 1
        row = self.adj[t]
        row.pop(row.index((s, e), self.nout[t]))
        self._free.append(e)
        self._n_edges -= 1

    # ---- queries -----------------------------------------------------------------------
    def num_vertices(self) -> int:
        return len(self.adj)

    def num_edges(self) -> int:
        return self._n_edges

    def out_degree(self, v: int) -> int:
        return self.nout[v]

    def in_degree(self, v: int) -> int:
        return len(self.adj[v]) - self.nout[v]

    def out_edges(self, v: int) -> List[int]:
        return [e for _, e in self.adj[v][: self.nout[v]]]

    def in_edges(self, v: int) -> List[int]:
        return [e for _, e in self.adj[v][self.nout[v]:]]

    def all_edges(self, v: int) -> List[int]:
        return [e for _, e in self.adj[v]]

    def out_neighbors(self, v: int) -> List[int]:
        return [n for n, _ in self.adj[v][: self.nout[v]]]

    def in_neighbors(self, v: int) -> List[int]:
        return [n for n, _ in self.adj[v][self.nout[v]:]]

    def all_neighbors(self, v: int) -> List[int]:
        return [n for n, _ in self.adj[v]]

    def black_in_edges(self, v: int) -> List[int]:
        eb = self.eblack
        return [e for _, e in self.adj[v][self.nout[v]:] if eb[e]]

    def black_out_edges(self, v: int) -> List[int]:
        eb = self.eblack
        return [e for _, e in self.adj[v][: self.nout[v]] if eb[e]]

    def edge(self, s: int, t: int) -> Optional[int]:
        for n, e in self.adj[s][: self.nout[s]]:
            if n == t:
                return e
        return None

    def edges(self) -> Iterator[int]:
        for v in range(len(self.adj)):
            for _, e in self.adj[v][: self.nout[v]]:
                yield e

    # ---- device view -------------------------------------------------------------------
    def csr_arrays(self):
        """(row_ptr[V+1] u64, n_out[V] u32, nbr[sum deg] u32, eidx[sum deg] u32) as numpy arrays:
        adjacency in the stored order, out-entries before in-entries -- the layout the graph
        kernels read."""
        import numpy as np

        adj = self.adj
        row_ptr = np.zeros(len(adj) + 1, dtype=np.uint64)
        if adj:
            np.cumsum(np.fromiter(map(len, adj), dtype=np.uint64, count=len(adj)), out=row_ptr[1:])
        flat = list(chain.from_iterable(adj))
        if flat:
            pairs = np.array(flat, dtype=np.uint32)
            nbr, eidx = np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])
        else:
            nbr = np.zeros(0, dtype=np.uint32)
            eidx = np.zeros(0, dtype=np.uint32)
        return row_ptr, np.asarray(self.nout, dtype=np.uint32), nbr, eidx


NodeMap = Dict[str, int]
EdgeMap = Dict[Tuple[str, str], int]
