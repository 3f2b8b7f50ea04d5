"""CHECKER (test infrastructure, not product code): ``pe_info`` as the disentanglement stages use it, on top of a
``PeLinks`` table, for the Python restatement of the stages in this package.

PE links.  The reference carries ``pe_info: {(min id, max id): count | None}`` through every
split and contraction, rewriting O(N^2) keys each time (Decomposition.py:492-503, :608-617,
Utilities.py:488-499).  The net effect of those rewrites is bilinear: a current node ``X`` stands
for the multiset ``supp(X)`` of ORIGINAL nodes it was contracted from (a split or forked copy
starts with an empty support because its rows are reset to zero; a contraction concatenates the
supports of its members), and

    pe(X, Y) = sum_{a in supp(X)} sum_{b in supp(Y)} P0[a, b]      (X != Y)
    pe(X, X) = P0[a, a] if X is an original node a, else 0

with ``P0`` the symmetrised count matrix of ``process_pe_info`` (IO.py:598-627).  The native stage handle keeps the
same bookkeeping (csrc/vs_stage.cpp); ``oracle/graph_ops.DictLiveLinks`` replays the reference's literal dict rewrites,
and tests/test_graph_golden.py runs the stages over both.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

from .model import GraphScan, NodeMap, PeLinks


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
