"""CHECKER (test infrastructure, not product code): iterative strain-path extraction over the disentangled graph,
restated in Python over ``AsmGraph`` (see disentangle.py in this package for how it is used).

Restates ``utils/VStrains_Extension.py`` (``best_matching`` :10-111, ``contig_extension``
:115-342, ``final_extension`` :345-418, ``reduce_graph`` :429-455, ``reduce_Anode`` :469-481,
``path_extension`` :484-899) and ``increment_nt_branch_coverage``
(``utils/VStrains_Utilities.py:183-208``) over ``AsmGraph``.

Device work: every re-initialisation runs the flow and scan kernels; the O(V^2 * ids^2)
``final_link_info`` table (Extension.py:766-799) is ONE grouped contraction of the resident PE
matrix (``PeLinks.group_matrix``).  The greedy walk is a chain of data-dependent decisions and
stays on the host.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy

from .model import GRAY, NodeMap, OGraph as AsmGraph, origin_ids
from .contig_ops import contigs_by_node, remap_contigs
from .disentangle import _CHECK_UNTOUCHED, Stage, _add_edge, _add_vertex, _retire_vertex, global_trivial_split, reinit
from .model import ContigDict, GraphOps, PeLinks, path_length, path_sequence
from .links import LiveLinks, nontrivial_ids

LinkTable = Dict[str, Dict[Tuple[str, str], int]]


def best_matching(stage: Stage, contigs: ContigDict, links: LiveLinks, logger) -> LinkTable:
    """Per remaining non-trivial branch: the (in, out) pairs a contig walks through, self pairs,
    and every other pair with a positive PE count."""
    g, nodes, _ = stage.triple()
    branches = nontrivial_ids(stage.scan, nodes)
    by_node = contigs_by_node(contigs)
    pairs = []
    shape = {}
    for no, v in branches.items():
        us = [g.vid[x] for x in g.in_neighbors(v)]
        ws = [g.vid[x] for x in g.out_neighbors(v)]
        shape[no] = (us, ws)
        pairs.extend((u, w) for u in us for w in ws)
    links.prefetch(pairs)
    table: LinkTable = {}
    for no in branches:
        us, ws = shape[no]
        through = set()
        for cno in by_node.get(no, []):
            ids = contigs[cno][0]
            at = ids.index(no)
            if 0 < at < len(ids) - 1:
                through.add((ids[at - 1], ids[at + 1]))
        kept: Dict[Tuple[str, str], int] = {}
        later = []
        for u in us:
            for w in ws:
                pe = links.get(u, w)
                if (u, w) in through or u == w:
                    kept[(u, w)] = pe
                else:
                    later.append((u, w, pe))
        for u, w, pe in sorted(later, key=lambda t: t[2], reverse=True):
            if pe > 0:
                kept[(u, w)] = pe
        table[no] = kept
    return table


def increment_nt_branch_coverage(stage: Stage, logger) -> None:
    g, nodes, _ = stage.triple()
    for no, v in nontrivial_ids(stage.scan, nodes).items():
        before = g.vdp[v]
        ins = g.in_neighbors(v)
        outs = g.out_neighbors(v)
        if sum([g.out_degree(x) for x in ins]) == g.in_degree(v) and \
                sum([g.in_degree(y) for y in outs]) == g.out_degree(v):
            a = sum(g.vdp[n] for n in ins)
            b = sum(g.vdp[n] for n in outs)
        else:
            a = sum(g.eflow[e] for e in g.in_edges(v))
            b = sum(g.eflow[e] for e in g.out_edges(v))
        g.vdp[v] = float(max([before, a, b]))
        logger.debug("NT Branch:{0}, cov: {1} -> {2}".format(no, before, g.vdp[v]))


# ---- the walk ----------------------------------------------------------------------------------
class _Side:
    """Direction-specific accessors so the forward and backward walks share one body."""

    def __init__(self, g: AsmGraph, forward: bool):
        self.forward = forward
        self.ahead = g.out_neighbors if forward else g.in_neighbors
        self.behind = g.in_neighbors if forward else g.out_neighbors

    def linked(self, table: Dict[Tuple[str, str], int], prev_id: str, nodes: NodeMap) -> List[int]:
        if self.forward:
            return [nodes[w] for (u, w) in table.keys() if u == prev_id]
        return [nodes[u] for (u, w) in table.keys() if w == prev_id]


def _walk(g: AsmGraph, nodes: NodeMap, path: List[int], visited: Dict[str, bool], start: int, side: _Side,
          table: LinkTable, use_coverage: bool, ccov, threshold, logger) -> None:
    """One direction of ``contig_extension`` (``use_coverage``) / ``final_extension`` (not)."""
    dp = g.vdp
    cur: Optional[int] = start
    while cur is not None and not visited[g.vid[cur]]:
        visited[g.vid[cur]] = True
        if side.forward:
            path.append(cur)
        else:
            path.insert(0, cur)
        prev = (path[-2] if side.forward else path[1]) if len(path) > 1 else None
        here = cur
        choices = side.ahead(here)
        if len(choices) == 0:
            cur = None
            continue
        if len(choices) == 1:
            cur = choices[0]
            continue
        by_coverage = False
        if g.vid[here] in table and prev is not None:
            cands = side.linked(table[g.vid[here]], g.vid[prev], nodes)
            if len(cands) == 1:
                if use_coverage and dp[cands[0]] - ccov <= -2 * threshold:
                    cur = None
                else:
                    cur = cands[0]
            elif len(cands) > 1:
                cur = None
            else:
                if use_coverage:
                    by_coverage = True   # cur stays on the branch vertex for now
                else:
                    cur = None
        else:
            cur = None
        if not use_coverage:
            continue
        if by_coverage:
            rivals = side.behind(here)
            if prev is not None and len(rivals) > 0:
                ahead_rank = sorted(choices, key=lambda x: abs(dp[prev] - dp[x]))
                best = ahead_rank[0]
                rival_rank = sorted(rivals, key=lambda x: abs(dp[best] - dp[x]))
                if rival_rank[0] == prev:
                    guard = max(2 * abs(dp[prev] - dp[best]), threshold)
                    if side.forward:
                        clash = (len(rival_rank) > 1 and abs(dp[rival_rank[1]] - dp[best]) <= guard) or \
                                (len(ahead_rank) > 1 and abs(dp[prev] - dp[ahead_rank[1]]) <= guard)
                    else:
                        clash = (len(ahead_rank) > 1 and abs(dp[ahead_rank[1]] - dp[prev]) <= guard) or \
                                (len(rival_rank) > 1 and abs(dp[best] - dp[rival_rank[1]]) <= guard)
                    if clash:
                        # the reference ``continue``s with cur still on the visited branch vertex:
                        # the loop ends here and the last-bit rule below is NOT tried
                        continue
                    cur = best
                else:
                    cur = None
            else:
                cur = None
        if cur is None:
            top = sorted([(x, dp[x]) for x in choices], key=lambda t: t[1], reverse=True)
            if top[0][1] - ccov > -threshold and top[1][1] - ccov <= -threshold:
                cur = top[0][0]


def _extend(g: AsmGraph, nodes: NodeMap, contig: List[str], table: LinkTable, use_coverage: bool, ccov,
            threshold, logger) -> List[int]:
    visited = dict.fromkeys(nodes.keys(), False)
    for name in contig[1:-1]:
        visited[name] = True
    path = [nodes[name] for name in contig][1:-1]
    _walk(g, nodes, path, visited, nodes[contig[-1]], _Side(g, True), table, use_coverage, ccov, threshold, logger)
    first = nodes[contig[0]]
    if len(contig) == 1 and path[-1] not in g.in_neighbors(first):
        visited[contig[0]] = False
        path.pop(0)
    _walk(g, nodes, path, visited, first, _Side(g, False), table, use_coverage, ccov, threshold, logger)
    return path


def contig_extension(g, nodes, contig, ccov, table, logger, threshold) -> List[int]:
    return _extend(g, nodes, contig, table, True, ccov, threshold, logger)


def final_extension(g, nodes, contig, table, logger) -> List[int]:
    return _extend(g, nodes, contig, table, False, None, None, logger)


def _bubble_vertices(g: AsmGraph, nodes: NodeMap, ids: List[str]) -> List[int]:
    return [nodes[n] for n in ids if g.in_degree(nodes[n]) == 1 and g.out_degree(nodes[n]) == 1]


def _consume(g: AsmGraph, nodes: NodeMap, usages: Dict[str, int], table: LinkTable, path: List[int], pcov,
             threshold, logger) -> None:
    """``reduce_graph``: subtract the path coverage; vertices at or under the threshold go gray
    and leave ``usages``; links touching a gray vertex are dropped."""
    grayed = False
    for v in path:
        usages[g.vid[v]] += 1
        g.vdp[v] = float(g.vdp[v] - pcov)
        if g.vdp[v] <= threshold:
            g.vblack[v] = GRAY
            usages.pop(g.vid[v])
            grayed = True
    # (every linked vertex was black when the pass filtered the table against the re-initialised graph, a moment ago: only
    # a vertex that went gray just now can cost a link, so a path that left all its vertices above the threshold needs no
    # walk over the whole table; VS_CHECK_UNTOUCHED=1 walks anyway and insists that nothing goes)
    if not grayed and not _CHECK_UNTOUCHED:
        return
    for kept in table.values():
        for (u, w) in list(kept.keys()):
            if not g.vblack[nodes[u]] or not g.vblack[nodes[w]]:
                assert grayed, "a link to a gray vertex outlived the table filter: %s %s" % (u, w)
                kept.pop((u, w))


def expand_path_names(name: str, members: Dict[str, List[str]]) -> List[str]:
    """``reduce_Anode``: replace extracted-path ids (``A<n>``, possibly with a split suffix) by
    their member ids until none is left."""
    ids = [name]
    while any(x.startswith("A") for x in ids):
        for i in range(len(ids)):
            if ids[i].startswith("A"):
                key = ids.pop(i).split("*")[0]
                ids[i:i] = members[key]
                break
    return ids


def path_extension(stage: Stage, contigs: ContigDict, table: LinkTable, frozen: PeLinks, ops: GraphOps, logger,
                   threshold, temp_dir: str):
    """Returns ``(strain_dict, usages)``; ``contigs`` and ``table`` are consumed in place."""
    logger.debug("-------------------------PATH Extension, delta: {0}".format(threshold))
    usages: Dict[str, int] = dict.fromkeys(stage.nodes.keys(), 0)
    strains: ContigDict = {}
    members: Dict[str, List[str]] = {}
    rid = 1
    while len(contigs) > 0:
        prev_ids = list(stage.nodes.keys())
        n_forks, id_mapping = global_trivial_split(stage, logger)
        # (from the second path on, nothing lies between the re-initialisation that closed the previous
        # round and this one except the trivial split: no fork, no change)
        stage = reinit(stage, ops, logger, "{0}/gfa/graph_S{1}.gfa".format(temp_dir, rid), untouched=(n_forks == 0 and rid > 1))
        g, nodes, edges = stage.triple()
        closure = remap_contigs(g, nodes, edges, contigs, id_mapping, prev_ids, logger)
        if n_forks == 0:
            # Nothing forked: every id stands for itself and the graph is the one the pass started from, so the
            # rewrite below only drops what no longer is a link between an in- and an out-neighbour; popping
            # and re-inserting every key in turn leaves the survivors in their order -- done in place, and
            # without building the closure of every id the table mentions.  (A link to an id that is no
            # node raises KeyError here as there.)
            for no in list(table.keys()):
                if no not in nodes:
                    table.pop(no)
                    continue
                kept = table[no]
                if len(kept) == 0:
                    continue
                v = nodes[no]
                ins = g.in_neighbors(v)
                outs = g.out_neighbors(v)
                for link in list(kept.keys()):
                    a, b = nodes[link[0]], nodes[link[1]]  # (both looked up, as the closure of both is there)
                    if not (a in ins and b in outs):
                        del kept[link]
        else:
            for no in list(table.keys()):
                if no not in nodes:
                    table.pop(no)
                    continue
                kept = table.pop(no)
                v = nodes[no]
                ins = g.in_neighbors(v)
                outs = g.out_neighbors(v)
                for (u, w), pe in list(kept.items()):
                    kept.pop((u, w))
                    if len(closure[u]) == 1 or len(closure[w]) == 1:
                        for uu in closure[u]:
                            for ww in closure[w]:
                                if (uu, ww) not in kept and nodes[uu] in ins and nodes[ww] in outs:
                                    kept[(uu, ww)] = pe
                table[no] = kept
        if n_forks == 0:  # nothing forked: every id stands for itself, and popping and re-inserting
            # every key in turn leaves the dict as it was
            if not usages.keys() <= closure.known:  # (set comparison in C; the walk only to name the first offender)
                for no in usages:
                    if no not in closure.known:
                        raise KeyError(no)
        else:
            for no, used in list(usages.items()):
                usages.pop(no)
                for new_no in closure[no]:
                    usages[new_no] = used

        longest, (contig, clen, ccov) = max(contigs.items(), key=lambda kv: kv[1][1])
        contigs.pop(longest)
        if all(usages[n] > 0 for n in contig):
            continue
        if any(not g.vblack[nodes[n]] for n in contig):
            continue

        cb = _bubble_vertices(g, nodes, contig)
        bbl_cov = numpy.median([g.vdp[v] for v in cb]) if len(cb) != 0 else ccov
        path = contig_extension(g, nodes, contig, min(ccov, bbl_cov), table, logger, threshold)
        pno = "A" + str(rid)
        plen = path_length(g, path)
        path_ids = [g.vid[v] for v in path]
        members[pno] = []
        for pid in path_ids:
            if pid in members:
                members[pno].extend(members[pid])
            else:
                members[pno].append(pid)
        pb = _bubble_vertices(g, nodes, path_ids)
        bbl_pcov = numpy.median([g.vdp[v] for v in pb]) if len(pb) != 0 else ccov
        pcov = min([ccov, bbl_pcov, bbl_cov])
        logger.debug("name: {0}, plen: {1}, pcov: {2}, bubble cov: {3}".format(pno, plen, pcov, bbl_pcov))
        strains[pno] = [members[pno], plen, pcov]
        for pid in path_ids:
            if pid in strains:
                strains.pop(pid)
        has_in = len(g.in_neighbors(path[0])) != 0
        has_out = len(g.out_neighbors(path[-1])) != 0
        if not has_in and not has_out:
            _consume(g, nodes, usages, table, path, pcov, threshold, logger)
        elif len(path) > 1:
            lo = 1 if has_in else 0
            hi = len(path) - 1 if has_out else len(path)
            inner = path[lo:hi]
            _consume(g, nodes, usages, table, inner, pcov, threshold, logger)
            if len(inner) > 0:
                pv = _add_vertex(g, nodes, pno, pcov, path_sequence(g, inner))
                if has_in:
                    _add_edge(g, edges, path[0], pv, g.eovl[g.edge(path[0], path[1])], pcov)
                if has_out:
                    _add_edge(g, edges, pv, path[-1], g.eovl[g.edge(path[-2], path[-1])], pcov)
                usages[pno] = 0
        stage = reinit(stage, ops, logger, "{0}/gfa/graph_S{1}post.gfa".format(temp_dir, rid))
        for cno in list(contigs.keys()):
            if any(n not in stage.nodes for n in contigs[cno][0]):
                contigs.pop(cno)
        rid += 1

    # vertices that carry the same sequence (fork copies): keep the deepest one
    g, nodes, edges = stage.triple()
    same_seq: Dict[str, List[int]] = {}
    for v in range(g.num_vertices()):
        same_seq.setdefault(g.vseq[v], []).append(v)
    for group in same_seq.values():
        if len(group) > 1:
            for v in sorted(group, key=lambda x: g.vdp[x], reverse=True)[1:]:
                _retire_vertex(g, nodes, g.vid[v])
                usages.pop(g.vid[v])
    stage = reinit(stage, ops, logger, "{0}/gfa/graph_S_final.gfa".format(temp_dir))
    g, nodes, edges = stage.triple()

    # link strength between the surviving vertices, from the ORIGINAL PE table
    order = list(range(g.num_vertices()))
    groups = [[frozen.index_of(x) for x in origin_ids(expand_path_names(g.vid[v], members))] for v in order]
    strength = frozen.group_matrix(groups)

    final_links: LinkTable = {}
    for no, v in nontrivial_ids(stage.scan, nodes).items():
        final_links[no] = {}
        ins = g.in_neighbors(v)
        outs = g.out_neighbors(v)
        in_use = dict.fromkeys([g.vid[x] for x in ins], 0)
        out_use = dict.fromkeys([g.vid[x] for x in outs], 0)
        combos = [(g.vid[a], g.vid[b], int(strength[a, b])) for a in ins for b in outs]
        for u, w, lf in sorted(combos, key=lambda t: t[2], reverse=True):
            if lf > 0 and in_use[u] == 0 and out_use[w] == 0:
                final_links[no][(u, w)] = lf
                in_use[u] += 1
                out_use[w] += 1

    for v in sorted(range(g.num_vertices()), key=lambda x: len(g.vseq[x]), reverse=True):
        if len(g.vseq[v]) <= 600:
            break
        if usages[g.vid[v]] == 0:
            path = final_extension(g, nodes, [g.vid[v]], final_links, logger)
            pno = "N" + str(rid)
            plen = path_length(g, path)
            path_ids = [g.vid[x] for x in path]
            pids: List[str] = []
            for pid in path_ids:
                if pid in members:
                    pids.extend(members[pid])
                else:
                    pids.append(pid)
            for pid in path_ids:
                if pid in strains:
                    strains.pop(pid)
            pb = _bubble_vertices(g, nodes, path_ids)
            pcov = numpy.median([g.vdp[x] for x in pb]) if len(pb) != 0 else g.vdp[v]
            strains[pno] = [pids, plen, pcov]
            for x in path:
                usages[g.vid[x]] += 1
            rid += 1
    for sno, (_, _, scov) in list(strains.items()):
        if scov <= 2 * threshold:
            strains.pop(sno)
    for sno in strains.keys():
        ids, slen, scov = strains[sno]
        flat: List[str] = []
        for name in ids:
            flat.extend(origin_ids(expand_path_names(name, members)))
        strains[sno] = [flat, slen, scov]
    return strains, usages
