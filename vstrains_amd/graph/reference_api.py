"""The in-process Python API of the hot path under the reference's own names and argument lists
(SURVEY.md 8b "Boundary 2"), so that a copy of ``utils/VStrains_SPAdes.py`` can swap its imports:

    from vstrains_amd.graph.reference_api import (process_pe_info, store_reinit_graph, edge_cleaning,
        iter_graph_disentanglement, best_matching, increment_nt_branch_coverage, path_extension)

| here | reference |
|---|---|
| ``process_pe_info(node_ids, pe_file, st_file)`` | ``utils/VStrains_IO.py:598`` |
| ``store_reinit_graph(graph, simp_node_dict, simp_edge_dict, logger, opt_filename)`` | ``utils/VStrains_IO.py:630`` |
| ``edge_cleaning(graph, simp_edge_dict, contig_dict, pe_info, logger)`` | ``utils/VStrains_Decomposition.py:822`` |
| ``iter_graph_disentanglement(graph, simp_node_dict, simp_edge_dict, contig_dict, pe_info, ref_file, logger, threshold, temp_dir)`` | ``utils/VStrains_Decomposition.py:908`` |
| ``best_matching(graph, simp_node_dict, simp_edge_dict, contig_dict, pe_info, logger)`` | ``utils/VStrains_Extension.py:10`` |
| ``increment_nt_branch_coverage(graph, simp_node_dict, logger)`` | ``utils/VStrains_Utilities.py:183`` |
| ``path_extension(graph, simp_node_dict, simp_edge_dict, contig_dict, full_link, pe_info, logger, threshold, temp_dir)`` | ``utils/VStrains_Extension.py:484`` |

Same argument order and meaning, same return shapes, same mutation contract (``contig_dict``, ``pe_info`` and
``full_link`` are changed in place and read by the later stages; graphs are replaced by the returned triple), same side
effects (the stage GFA files).  These are VIEWS: the work is done by the native stage handle (``native_stage.NativeStage``
-> ``vs_stage`` in the library).  Every call loads the objects it is handed into the handle, runs the one library call
that is the reference function, and copies the result back into objects of the shapes above:

* ``graph`` is an ``asm_graph.AsmGraph``; ``simp_node_dict`` / ``simp_edge_dict`` map ids / id pairs to its integer
  vertices / edges (``formats.read_stage_gfa`` makes them from a GFA file).
* ``pe_info`` is a live view of the link table the handle keeps on the device: ``process_pe_info`` returns
  ``(live view, frozen table)`` where the reference returns ``(dict, copy of the dict)``.  ``pe_info[(u, v)]`` answers
  for every id the stages create, exactly as the rewritten dict would.

(``pipeline.extract_strains`` does not pay for these copies: it loads the prepared graph into the handle once and
calls the stages back to back.)  The device side comes from a backend (``set_backend``); the default is the HIP backend
and fails loudly without a GPU.
"""
from __future__ import annotations

from typing import Optional

from .asm_graph import AsmGraph

_backend = None
_engine = None  # the handle process_pe_info made last (increment_nt_branch_coverage is handed no pe_info)


def set_backend(backend) -> None:
    """``backend``: ``hip_ops.HipBackend`` (default, created on first use) or a test double with ``links_from_files`` /
    ``native_stage``."""
    global _backend, _engine
    _backend = backend
    _engine = None


def _be():
    global _backend
    if _backend is None:
        from .hip_ops import HipBackend

        _backend = HipBackend()
    return _backend


class LivePeInfo:
    """``pe_info`` of the disentanglement stages: a view of the handle's link bookkeeping."""

    def __init__(self, engine, table):
        self.engine = engine
        self.table = table

    def __getitem__(self, key):
        return self.engine.link(key[0], key[1])

    def get(self, a: str, b: str) -> int:
        return self.engine.link(a, b)


def _copy_into(dst: AsmGraph, src: AsmGraph) -> None:
    for slot in AsmGraph.__slots__:
        setattr(dst, slot, getattr(src, slot))


def _engine_of(pe_info):
    eng = getattr(pe_info, "engine", None)
    if eng is None:
        raise TypeError("pe_info must be what process_pe_info returned")
    return eng


def process_pe_info(node_ids, pe_file: str, st_file: str):
    """IO.py:598-627 -> ``(pe_info, dcpy_pe_info)``: the symmetrised table built on the device from the two N^2-line text
    files, as a live view for the disentanglement stages and as the frozen table ``path_extension`` reads."""
    global _engine
    table = _be().links_from_files(list(node_ids), pe_file, st_file)
    _engine = _be().native_stage(table)
    table.engine = _engine
    return LivePeInfo(_engine, table), table


def store_reinit_graph(graph, simp_node_dict, simp_edge_dict, logger, opt_filename: str):
    """IO.py:630-642 -> ``(graph, simp_node_dict, simp_edge_dict)``."""
    if _engine is None:
        raise RuntimeError("store_reinit_graph: process_pe_info has not been called (it makes the stage handle)")
    eng = _engine
    eng.load_graph(graph, simp_node_dict, simp_edge_dict)
    eng.reinit(opt_filename, logger)
    return eng.graph()


def edge_cleaning(graph, simp_edge_dict, contig_dict, pe_info, logger):
    """Decomposition.py:822-905 -> the ``assigned`` map; ``graph`` and ``simp_edge_dict`` lose the removed edges."""
    eng = _engine_of(pe_info)
    nodes = {name: v for v, name in enumerate(graph.vid) if graph.vblack[v]}
    eng.load_graph(graph, nodes, simp_edge_dict)
    eng.load_contigs(contig_dict)
    eng.edge_cleaning(logger)
    g2, _, e2 = eng.graph()
    _copy_into(graph, g2)
    simp_edge_dict.clear()
    simp_edge_dict.update(e2)
    return eng.assigned()


def iter_graph_disentanglement(graph, simp_node_dict, simp_edge_dict, contig_dict, pe_info, ref_file, logger,
                               threshold, temp_dir: str):
    """Decomposition.py:908-1042 -> ``(graph, simp_node_dict, simp_edge_dict)``.  ``ref_file`` feeds
    the reference's hidden ``-r`` debug plumbing (needs minimap2) and must be None here."""
    if ref_file:
        raise NotImplementedError("the -r debug mode (minimap2) is outside the hot path")
    eng = _engine_of(pe_info)
    eng.load_graph(graph, simp_node_dict, simp_edge_dict)
    eng.refresh_scan()  # (the branch / simple-edge facts of the graph as handed over)
    eng.load_contigs(contig_dict)
    try:
        eng.disentangle(threshold, temp_dir, logger)
    finally:
        eng.contigs_into(contig_dict)
    return eng.graph()


def best_matching(graph, simp_node_dict, simp_edge_dict, contig_dict, pe_info, logger):
    """Extension.py:10-111 -> ``full_link``."""
    eng = _engine_of(pe_info)
    eng.load_graph(graph, simp_node_dict, simp_edge_dict)
    eng.refresh_scan()
    eng.load_contigs(contig_dict)
    eng.best_matching(logger)
    return eng.full_link()


def increment_nt_branch_coverage(graph, simp_node_dict, logger) -> None:
    """Utilities.py:183-208 (in place)."""
    if _engine is None:
        raise RuntimeError("increment_nt_branch_coverage: process_pe_info has not been called")
    _engine.load_graph(graph, simp_node_dict, {})
    _engine.refresh_scan()
    _engine.increment_nt_branch_coverage(logger)
    g2, _, _ = _engine.graph()
    graph.vdp = g2.vdp


def path_extension(graph, simp_node_dict, simp_edge_dict, contig_dict, full_link, pe_info, logger, threshold,
                   temp_dir: str):
    """Extension.py:484-899 -> ``(strain_dict, usages)``; ``pe_info`` is the frozen copy
    ``process_pe_info`` returned second."""
    eng = _engine_of(pe_info)
    eng.load_graph(graph, simp_node_dict, simp_edge_dict)
    eng.refresh_scan()
    eng.load_contigs(contig_dict)
    eng.load_full_link(full_link)
    try:
        eng.path_extension(threshold, temp_dir, logger)
    finally:
        eng.contigs_into(contig_dict)
        fresh = eng.full_link()
        full_link.clear()
        full_link.update(fresh)
    return eng.strains(), eng.usages()
