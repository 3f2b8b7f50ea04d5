"""The data-parallel operations of the graph stages, as the host code sees them.

The stage logic (``disentangle.py``, ``extend.py``) is written against these two small
interfaces.  The product implementation is ``hip_ops.HipGraphOps`` / ``hip_ops.HipPeLinks``
(HIP kernels behind the C ABI, ``include/vstrains_hip.h`` "graph stage" section); there is no CPU
implementation in this package.  ``oracle/graph_ops.py`` holds the checker's restatement of the
same operations in the reference's own terms (numpy sums, the ``pe_info`` dict) and is injected
only by tests.

PE links.  The reference carries ``pe_info: {(min id, max id): count | None}`` through every
split and contraction, rewriting O(N^2) keys each time (Decomposition.py:492-503, :608-617,
Utilities.py:488-499).  The net effect of those rewrites is bilinear: a current node ``X`` stands
for the multiset ``supp(X)`` of ORIGINAL nodes it was contracted from (a split or forked copy
starts with an empty support because its rows are reset to zero; a contraction concatenates the
supports of its members), and

    pe(X, Y) = sum_{a in supp(X)} sum_{b in supp(Y)} P0[a, b]      (X != Y)
    pe(X, X) = P0[a, a] if X is an original node a, else 0

with ``P0`` the symmetrised count matrix of ``process_pe_info`` (IO.py:598-627).  So the table is
never rewritten here: ``P0`` stays resident in HBM and every lookup is a (batched) sum over two
short index lists.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

from .asm_graph import AsmGraph, NodeMap


class GraphScan:
    """Per-vertex facts of one graph snapshot (all by vertex index)."""

    __slots__ = ("nontrivial", "fork_kind", "chain_next", "chain_top", "chain_rank")

    def __init__(self, nontrivial, fork_kind, chain_next, chain_top, chain_rank):
        self.nontrivial = nontrivial  # bool: non-trivial branch (Utilities.py:162-172)
        self.fork_kind = fork_kind    # 0 none, 1: 1 in / >1 out, 2: >1 in / 1 out (black edges)
        self.chain_next = chain_next  # target of the vertex's simple out-edge, or -1 (:398-402)
        self.chain_top = chain_top    # head of the chain of simple edges the vertex lies on (itself if none)
        self.chain_rank = chain_rank  # distance from that head; -1 on a ring of simple edges


class GraphOps:
    """K6 / K7 of SURVEY.md 2.1."""

    def edge_flows(self, g: AsmGraph) -> None:
        """``assign_edge_flow`` (Utilities.py:14-31) for every edge of ``g``; fills ``g.eflow``."""
        raise NotImplementedError

    def scan(self, g: AsmGraph) -> GraphScan:
        raise NotImplementedError

    def refresh(self, g: AsmGraph) -> GraphScan:
        """Both at once for a freshly re-initialised graph (one device call in the HIP backend)."""
        self.edge_flows(g)
        return self.scan(g)


class PeLinks:
    """K5: the symmetrised PE-link matrix ``P0`` over the nodes of ``s_graph_L1`` and sums over it.

    The ``note_*`` calls tell the table what the stage just did; the device table only needs them
    to know which ids are "fresh" (the reference's ``None`` marks), the checker replays the
    reference's dict rewrites with them."""

    names: List[str]

    def index_of(self, name: str) -> int:
        raise NotImplementedError

    def block_sums(self, queries: Sequence[Tuple[Sequence[int], Sequence[int]]]) -> List[int]:
        """For each (rows, cols) query: sum of P0[r, c] over all r in rows, c in cols (with
        multiplicity)."""
        raise NotImplementedError

    def group_matrix(self, groups: Sequence[Sequence[int]]):
        """numpy int64 [n, n]: entry (i, j) = block sum of groups[i] x groups[j]."""
        raise NotImplementedError


class LiveLinks:
    """``pe_info`` as the disentanglement stage uses it, on top of a ``PeLinks``."""

    def __init__(self, table: PeLinks):
        self.table = table
        self._fresh: Dict[str, None] = {}
        self._cache: Dict[Tuple[str, str], int] = {}
        self._supp: Dict[str, List[int]] = {}
        self._derived: Dict[str, None] = {}

    def rows(self, name: str) -> List[int]:
        """Support of a live node.  Ids cannot be parsed for this (``x&y*A`` may be a fork of the
        contraction ``x&y`` or the contraction of ``x`` with a fork ``y*A``), so supports are
        recorded when the stage reports a split / fork / contraction; an id never reported is an
        original node."""
        r = self._supp.get(name)
        if r is None:
            r = [self.table.index_of(name)]
            self._supp[name] = r
        return r

    def is_fresh(self, name: str) -> bool:
        return name in self._fresh

    def prefetch(self, pairs: Iterable[Tuple[str, str]]) -> None:
        wanted: Dict[Tuple[str, str], None] = {}
        for a, b in pairs:
            key = (a, b) if a <= b else (b, a)
            if key not in self._cache:
                wanted[key] = None
        want = list(wanted)
        if not want:
            return
        sums = self.table.block_sums([self._query(a, b) for a, b in want])
        for key, s in zip(want, sums):
            self._cache[key] = int(s)

    def _query(self, a: str, b: str):
        if a == b:
            ra = self.rows(a)
            # an original node keeps its diagonal count; every derived id has 0 with itself
            return (ra, ra) if a not in self._derived else ((), ())
        return (self.rows(a), self.rows(b))

    def get(self, a: str, b: str) -> int:
        key = (a, b) if a <= b else (b, a)
        v = self._cache.get(key)
        if v is None:
            self.prefetch([key])
            v = self._cache[key]
        return v

    # ---- stage notifications
    def _born(self, name: str, support: List[int], fresh: bool) -> None:
        self._supp[name] = support
        self._derived[name] = None
        if fresh:
            self._fresh[name] = None

    def note_split(self, removed: str, subs: List[str], live_ids: Iterable[str]) -> None:
        for s in subs:
            self._born(s, [], True)

    def note_fork(self, sub: str, live_ids: Iterable[str]) -> None:
        self._born(sub, [], True)

    def note_drop(self, removed: str) -> None:
        pass

    def note_merge(self, new_id: str, members: List[str], live_ids: Iterable[str]) -> None:
        support: List[int] = []
        for m in members:
            support.extend(self.rows(m))
        self._born(new_id, support, False)

    def end_pass(self) -> None:
        self._fresh.clear()


def nontrivial_ids(scan: GraphScan, nodes: NodeMap) -> Dict[str, int]:
    """``get_non_trivial_branches`` (Utilities.py:175-180): in node-map order."""
    return {name: v for name, v in nodes.items() if scan.nontrivial[v]}
