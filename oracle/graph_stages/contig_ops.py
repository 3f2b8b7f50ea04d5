"""CHECKER (test infrastructure, not product code): the contig bookkeeping only the stages need, for the Python
restatement of the stages in this package.  Restates ``contig_dict_remapping`` Utilities.py:281-380.

Where the reference iterates a Python ``set`` of strings (whose order changes with PYTHONHASHSEED) this code iterates
in first-insertion order; the golden cases record whether the reference's outputs are invariant to the hash seed
(``case.json:hashseed_invariant``).
"""
from __future__ import annotations

from functools import reduce
from typing import Dict, List, Tuple

from .model import OGraph as AsmGraph, ContigDict, EdgeMap, NodeMap, contig_steps, contigs_by_node, path_length  # noqa: F401  (re-exported for the stage modules)


PY_MERGE_ID_FRAMES = 993  # nested merge_id frames CPython 3.10 allows the reference's CLI (limit 1000, six frames above, len() on top)


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
        # (depth-first, forks in order, without recursion: the chain may be as long as the reference's limit allows, and
        # this checker runs at whatever depth its caller happens to be)
        out: Dict[str, None] = {}
        stack = [name]
        while stack:
            cur = stack.pop()
            kids = id_mapping.get(cur, ())
            if len(kids) == 0:
                out.setdefault(cur, None)
            else:
                stack.extend(reversed(list(kids)))
        return out

    logger.info("contig resolution..")
    known = set(prev_ids)
    if not known <= id_mapping.keys():
        for name in prev_ids:
            id_mapping[name]  # the reference indexes it directly: unknown ids are an error
    # The reference follows the forks with a recursive function, eagerly for every id of the graph before the pass
    # (merge_id, Utilities.py:318-334), six frames deep under CPython's recursion limit of 1 000: a fork chain that
    # needs a 994th nested frame ends it with RecursionError (runaway trivial splits on circular graphs).  Depth of an
    # id: 1 if it was not forked, else 1 + the deepest of its forks -- worked out here without recursion.
    if any(len(k) for k in id_mapping.values()):
        depth: Dict[str, int] = {}
        for root in prev_ids:
            stack = [root]
            while stack:
                name = stack[-1]
                if name in depth:
                    stack.pop()
                    continue
                kids = id_mapping.get(name, ())
                todo = [k for k in kids if k not in depth]
                if todo:
                    stack.extend(todo)
                    continue
                depth[name] = 1 + max((depth[k] for k in kids), default=0)
                stack.pop()
            if depth[root] > PY_MERGE_ID_FRAMES:
                raise RecursionError("maximum recursion depth exceeded while calling a Python object")
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


