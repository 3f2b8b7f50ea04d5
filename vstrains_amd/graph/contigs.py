"""Contig / strain records of the final process (VStrains_SPAdes.py:251-272).

A record is ``[node id list, length, coverage]`` keyed by name, in a dict whose insertion order is
significant (the reference's ``contig_dict`` / ``strain_dict``).  Restates ``contig_map_node`` Utilities.py:227-244 (also used by the
graph preparation), ``trim_contig_dict`` :147-159, ``contig_dup_removed_s`` :589-616, ``contig_resolve`` :211-224,
``strain_repeat_resol`` :800-836.  (Inside the stages the same bookkeeping runs in the native stage handle,
csrc/vs_stage.cpp.)

Where the reference iterates a Python ``set`` of strings (whose order changes with
PYTHONHASHSEED) this code iterates in first-insertion order; the golden cases record whether the
reference's outputs are invariant to the hash seed (``case.json:hashseed_invariant``).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

from .asm_graph import AsmGraph, NodeMap
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
    # The reference tests every original contig against every strain; a contig can only be a subset of a strain that holds
    # its first id, and the maxima below do not depend on the order the contigs are met in.
    by_first: Dict[str, list] = {}
    for cno, (cids, _, _) in original_contigs.items():
        cset = frozenset(cids)
        by_first.setdefault(cids[0] if cids else None, []).append((cno, cset))
    for sno, (ids, _, cov) in list(strains.items()):
        sub = origin_ids(ids)
        subset = set(sub)
        times = dict.fromkeys(sub, 1)
        for first in [None] + list(subset):
            for cno, cset in by_first.get(first, ()):
                if cset.issubset(subset):
                    for name, count in contig_info[cno][1].items():
                        times[name] = max(times[name], count)
        expanded: List[str] = []
        for name in sub:
            expanded.extend([name] * times[name])
        strains[sno] = [expanded, path_length(g, [nodes[n] for n in expanded]), cov]
    logger.info("done")
