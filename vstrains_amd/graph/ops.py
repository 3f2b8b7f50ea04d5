"""The data-parallel operations of the graph stages, as host code sees them: ``GraphScan`` (what the vertex-scan / chain
kernels report about one graph snapshot), ``GraphOps`` (K6 / K7 of SURVEY.md 2.1) and ``PeLinks`` (K5).  The product
implementation is ``hip_ops.HipGraphOps`` / ``hip_ops.HipPeLinks`` (HIP kernels behind the C ABI,
``include/vstrains_hip.h`` "graph stage" section) and, for whole stages, the native stage handle
(``native_stage.NativeStage``), which calls the same kernels from inside the library; there is no CPU implementation in
this package.  ``oracle/graph_ops.py`` holds the checker's restatement of the same operations in the reference's own
terms (numpy sums, the ``pe_info`` dict) and is injected only by tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

from .asm_graph import AsmGraph


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
