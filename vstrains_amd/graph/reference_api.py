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

Same argument order and meaning, same return shapes, same mutation contract (``contig_dict``,
``pe_info`` and ``full_link`` are changed in place and read by the later stages; graphs are
replaced by the returned triple), same side effects (the stage GFA files).  What differs is the
TYPE behind three of the names, because graph-tool is not what runs here:

* ``graph`` is an ``asm_graph.AsmGraph``; ``simp_node_dict`` / ``simp_edge_dict`` map ids / id pairs
  to its integer vertices / edges (``formats.read_stage_gfa`` makes them from a GFA file).
* ``pe_info`` is the device-resident link table: ``process_pe_info`` returns
  ``(live view, frozen table)`` where the reference returns ``(dict, copy of the dict)``.  The live
  view answers ``pe_info[(u, v)]`` for every id the stages create, exactly as the rewritten dict
  would (``ops.LiveLinks``; checked against the literal dict in tests/test_graph_golden.py).

The device side comes from a backend (``set_backend``); the default is the HIP backend and fails
loudly without a GPU.
"""
from __future__ import annotations

from typing import Optional

from . import disentangle as _dis
from . import extend as _ext
from .disentangle import Stage

_backend = None


def set_backend(backend) -> None:
    """``backend``: ``hip_ops.HipBackend`` (default, created on first use) or a test double with
    ``graph_ops`` / ``live_links`` / ``links_from_files``."""
    global _backend
    _backend = backend


def _be():
    global _backend
    if _backend is None:
        from .hip_ops import HipBackend

        _backend = HipBackend()
    return _backend


def _stage(graph, simp_node_dict, simp_edge_dict) -> Stage:
    # the branch / simple-edge facts of this snapshot (one vs_graph_refresh launch); flows are kept as they are
    return Stage(graph, simp_node_dict, simp_edge_dict, _be().graph_ops.scan(graph))


def process_pe_info(node_ids, pe_file: str, st_file: str):
    """IO.py:598-627 -> ``(pe_info, dcpy_pe_info)``: the symmetrised table built on the device from
    the two N^2-line text files, as a live view for the disentanglement stages and as the frozen
    table ``path_extension`` reads."""
    table = _be().links_from_files(list(node_ids), pe_file, st_file)
    return _be().live_links(table), table


def store_reinit_graph(graph, simp_node_dict, simp_edge_dict, logger, opt_filename: str):
    """IO.py:630-642 -> ``(graph, simp_node_dict, simp_edge_dict)``."""
    st = _dis.reinit(Stage(graph, simp_node_dict, simp_edge_dict), _be().graph_ops, logger, opt_filename)
    return st.triple()


def edge_cleaning(graph, simp_edge_dict, contig_dict, pe_info, logger):
    """Decomposition.py:822-905 -> the ``assigned`` map."""
    return _dis.edge_cleaning(graph, simp_edge_dict, contig_dict, pe_info, logger)


def iter_graph_disentanglement(graph, simp_node_dict, simp_edge_dict, contig_dict, pe_info, ref_file, logger,
                               threshold, temp_dir: str):
    """Decomposition.py:908-1042 -> ``(graph, simp_node_dict, simp_edge_dict)``.  ``ref_file`` feeds
    the reference's hidden ``-r`` debug plumbing (needs minimap2) and must be None here."""
    if ref_file:
        raise NotImplementedError("the -r debug mode (minimap2) is outside the hot path")
    st = _dis.iter_graph_disentanglement(_stage(graph, simp_node_dict, simp_edge_dict), contig_dict, pe_info,
                                         _be().graph_ops, logger, threshold, temp_dir)
    return st.triple()


def best_matching(graph, simp_node_dict, simp_edge_dict, contig_dict, pe_info, logger):
    """Extension.py:10-111 -> ``full_link``."""
    return _ext.best_matching(_stage(graph, simp_node_dict, simp_edge_dict), contig_dict, pe_info, logger)


def increment_nt_branch_coverage(graph, simp_node_dict, logger) -> None:
    """Utilities.py:183-208 (in place)."""
    _ext.increment_nt_branch_coverage(_stage(graph, simp_node_dict, {}), logger)


def path_extension(graph, simp_node_dict, simp_edge_dict, contig_dict, full_link, pe_info, logger, threshold,
                   temp_dir: str):
    """Extension.py:484-899 -> ``(strain_dict, usages)``; ``pe_info`` is the frozen copy
    ``process_pe_info`` returned second."""
    return _ext.path_extension(_stage(graph, simp_node_dict, simp_edge_dict), contig_dict, full_link, pe_info,
                               _be().graph_ops, logger, threshold, temp_dir)
