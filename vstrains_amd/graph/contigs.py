"""Contig bookkeeping shared by the disentanglement and extension stages.

A contig record is ``[node id list, length, coverage]`` keyed by contig name, in a dict whose
insertion order is significant (the reference's ``contig_dict``).  Restates, in order:
``contig_map_node`` Utilities.py:227-244, ``trim_contig_dict`` :147-159, ``contig_dup_removed_s``
:589-616, ``contig_dict_remapping`` :281-380, ``contig_resolve`` :211-224,
``strain_repeat_resol`` :800-836.

Where the reference iterates a Python ``set`` of strings (whose order changes with
PYTHONHASHSEED) this code iterates in first-insertion order; the golden cases record whether the
reference's outputs are invariant to the hash seed (``case.json:hashseed_invariant``).
"""
from __future__ import annotations

from functools import reduce
from typing import Dict, List, Tuple

from .asm_graph import AsmGraph, EdgeMap, NodeMap
from .formats import ContigDict, path_length


def contigs_by_node(contigs: ContigDict) -> Dict[str, List[str]]:
    """node id -> contig names that visit it (first-seen order, each once)."""
    seen: Dict[str, Dict[str, None]] = {}
    for cno, (ids, _, _) in contigs.items():
        for n in ids:
            seen.setdefault(n, {})[cno] = None
    return {n: list(d) for n, d in seen.items()}


def contig_steps(contigs: ContigDict) -> Dict[Tuple[str, str], List[str]]:
    """(id, next id) -> contig names that take that step."""
    seen: Dict[Tuple[str, str], Dict[str, None]] = {}
    for cno, (ids, _, _) in contigs.items():
        for a, b in zip(ids, ids[1:]):
            seen.setdefault((a, b), {})[cno] = None
    return {k: list(d) for k, d in seen.items()}


def trim_contigs(g: AsmGraph, nodes: NodeMap, contigs: ContigDict, logger) -> ContigDict:
    logger.info("trim contig..")
    for cno, (ids, _, cov) in list(contigs.items()):
        uniq = list(dict.fromkeys(ids))
        contigs[cno] = [uniq, path_length(g, [nodes[n] for n in uniq]), cov]
    logger.info("done")
    return contigs


def drop_duplicate_contigs(contigs: ContigDict, logger) -> ContigDict:
    """Node-SET comparison: equal sets drop the later one, a proper subset drops the smaller."""
    logger.info("drop duplicated contigs..")
    dropped: Dict[str, None] = {}
    sets = {cno: set(rec[0]) for cno, rec in contigs.items()}
    names = list(contigs.keys())
    for a in names:
        for b in names:
            if a in dropped or b in dropped or a == b:
                continue
            common = len(sets[a] & sets[b])
            la, lb = len(contigs[a][0]), len(contigs[b][0])
            if common == la and common == lb:
                dropped[b] = None
            elif common == la:
                dropped[a] = None
            elif common == lb:
                dropped[b] = None
    for cno in dropped:
        contigs.pop(cno)
    logger.debug("duplicated contigs: " + str(set(dropped)))
    logger.info("done")
    return contigs


class _Closure(dict):
    """id -> ordered set of the ids it ended up as, for the ids of the graph before the pass
    (anything else is a KeyError, as with the plain dict); filled on first use."""

    __slots__ = ("_leaves", "known")

    def __init__(self, leaves, known):
        super().__init__()
        self._leaves = leaves
        self.known = known

    def __missing__(self, name):
        if name not in self.known:
            raise KeyError(name)
        out = self[name] = self._leaves(name)
        return out


def remap_contigs(g: AsmGraph, nodes: NodeMap, edges: EdgeMap, contigs: ContigDict,
                  id_mapping: Dict[str, Dict[str, None]], prev_ids: List[str], logger):
    """Follow ``id_mapping`` (id -> ids it was forked into) transitively, then re-thread every
    contig through the forked ids along existing edges.  Returns the transitive mapping as
    ordered sets (dict keys)."""

    def leaves(name: str) -> Dict[str, None]:
        kids = id_mapping.get(name, ())
        if len(kids) == 0:
            return {name: None}
        out: Dict[str, None] = {}
        for kid in kids:
            out.update(leaves(kid))
        return out

    logger.info("contig resolution..")
    known = set(prev_ids)
    if not known <= id_mapping.keys():
        for name in prev_ids:
            id_mapping[name]  # the reference indexes it directly: unknown ids are an error
    # (the closure of an id is worked out when somebody asks for it: a pass forks a handful of the
    # thousands of ids, and the callers index by id only)
    closure = _Closure(leaves, known)

    def images(ids: List[str]) -> List[List[str]]:
        paths = [[s] for s in closure[ids[0]]]
        for nxt in ids[1:]:
            grown = []
            for p in paths:
                for cand in closure[nxt]:
                    if (p[-1], cand) in edges:
                        grown.append(p + [cand])
            paths = grown
        return paths

    for cno, (ids, _, cov) in list(contigs.items()):
        paths = images(ids)
        if len(paths) < 1:
            logger.debug("error, contig missed: " + str(cno) + str(ids))
        elif len(paths) == 1:
            if paths[0] != ids:
                contigs.pop(cno)
                contigs[cno] = [paths[0], path_length(g, [nodes[n] for n in paths[0]]), cov]
        else:
            contigs.pop(cno)
            common = reduce(lambda a, b: [i for i in a if i in b], paths)
            if len(common) > 0:
                contigs[cno] = [common, path_length(g, [nodes[n] for n in common]), cov]
    logger.info("done")
    return closure


def origin_ids(ids: List[str]) -> List[str]:
    """Un-zip contracted ids and strip split suffixes: ``a&b*0`` -> ``a, b``
    (``contig_resolve`` Utilities.py:211-224, ``reduce_id_simple`` Extension.py:458-466)."""
    out: List[str] = []
    for name in ids:
        for iid in str(name).split("&"):
            star = iid.find("*")
            out.append(iid if star == -1 else iid[:star])
    return out


def resolve_contigs(contigs: ContigDict) -> None:
    for cno in contigs.keys():
        ids, length, cov = contigs[cno]
        contigs[cno] = [origin_ids(ids), length, cov]


def restore_repeats(g: AsmGraph, nodes: NodeMap, strains: ContigDict, contig_info: Dict[str, tuple],
                    original_contigs: ContigDict, logger) -> None:
    """Put back repeat multiplicities recorded when the SPAdes paths were parsed."""
    logger.info("resolving repeat nodes..")
    for sno, (ids, _, cov) in list(strains.items()):
        sub = origin_ids(ids)
        subset = set(sub)
        times = dict.fromkeys(sub, 1)
        for cno, (cids, _, _) in original_contigs.items():
            if set(cids).issubset(subset):
                for name, count in contig_info[cno][1].items():
                    times[name] = max(times[name], count)
        expanded: List[str] = []
        for name in sub:
            expanded.extend([name] * times[name])
        strains[sno] = [expanded, path_length(g, [nodes[n] for n in expanded]), cov]
    logger.info("done")
