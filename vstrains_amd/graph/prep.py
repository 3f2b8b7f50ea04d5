"""Upstream of the hot path: assembler GFA -> strand-canonical graph, coverage cut-off, contigs.

SURVEY.md 8(f)-1.  Restates ``gfa_to_graph`` / ``flip_graph_bfs`` / ``reduce_graph``
(``utils/VStrains_IO.py:27-295``), ``spades_paths_parser`` (:398-515), ``reindexing`` /
``threshold_estimation`` / ``graph_simplification`` (``utils/VStrains_Preprocess.py:13-123``).
Host-only, O(N), runs once; nothing here is a kernel.
"""
from __future__ import annotations

import re
import sys
from typing import Dict, List, Tuple

import numpy

from .asm_graph import BLACK, GRAY, AsmGraph, EdgeMap, NodeMap
from .contigs import contig_steps, contigs_by_node
from .formats import ContigDict, gfa_records, path_length

_COMPLEMENT = {"A": "T", "T": "A", "C": "G", "G": "C"}


def reverse_complement(seq: str) -> str:
    """KeyError on any byte outside ACGT, like the reference (Utilities.py:1015-1016)."""
    return "".join(_COMPLEMENT[c] for c in reversed(seq))


def _fatal(logger, *lines: str) -> None:
    for line in lines:
        logger.error(line)
    sys.exit(1)


def load_assembly_graph(gfa_file: str, logger, init_ori: int = 1) -> Tuple[AsmGraph, NodeMap, EdgeMap]:
    """Two vertices per segment (forward at 2i, reverse complement at 2i+1), links between the
    named orientations, then one orientation per segment is chosen by the depth-first flip."""
    logger.info("Parsing GFA format graph")
    segs, links = gfa_records(gfa_file)
    logger.info("Parsed gfa file length: {0}, version: {1}".format(len(segs) + len(links), "gfa1"))
    g = AsmGraph()
    ori: List[int] = []
    pair_of: Dict[str, Tuple[int, int]] = {}
    depth: Dict[str, float] = {}
    for rec in segs:
        kind, name, seq = rec[:3]
        dp = 0
        ln = 0
        kc = 0
        for tag in rec[3:]:
            if tag.startswith("dp") or tag.startswith("DP"):
                dp = float(tag.split(":")[2])
                break
            if tag.startswith("ln") or tag.startswith("LN"):
                ln = int(tag.split(":")[2])
            if tag.startswith("kc") or tag.startswith("KC"):
                kc = int(tag.split(":")[2])
            if ln != 0 and kc != 0:
                break
        if kind != "S" or (dp == 0 and (ln == 0 or kc == 0)):
            _fatal(logger, "file: {0}, Illegal graph format, please double check if the graph has been contaminated".format(gfa_file))
        if dp == 0:
            dp = kc / ln
        vp = g.add_vertex(name, dp, seq, BLACK)
        vn = g.add_vertex(name, dp, reverse_complement(seq), BLACK)
        ori.extend((1, -1))
        pair_of[name] = (vp, vn)
        depth[name] = dp

    by_key: Dict[Tuple[str, int, str, int], int] = {}
    for rec in links:
        kind, left, ori_l, right, ori_r = rec[:5]
        ovl = [t for t in rec[5:] if t.endswith("m") or t.endswith("M")][0]
        assert kind == "L" and ovl[-1] == "M"
        u = pair_of[left][0] if ori_l == "+" else pair_of[left][1]
        v = pair_of[right][0] if ori_r == "+" else pair_of[right][1]
        key = (left, ori[u], right, ori[v])
        if key in by_key:
            _fatal(logger, "parallel edge found, invalid case in assembly graph, please double-check the assembly graph format",
                   "Pipeline aborted")
        if left == right:
            g.vseq[u] = g.vseq[u].lower()
            g.vseq[v] = g.vseq[v].lower()
            continue
        by_key[key] = g.add_edge(u, v, int(ovl[:-1]), None, BLACK)

    nodes, edges = _choose_orientations(g, ori, pair_of, by_key, depth, logger, init_ori)
    return _compact(g, nodes, edges)


def _choose_orientations(g: AsmGraph, ori: List[int], pair_of, by_key, depth: Dict[str, float], logger, init_ori: int):
    """``flip_graph_bfs``: start at the deepest unvisited segment, keep its ``init_ori`` copy, turn
    every link of the other copy around (reverse complement of a link), continue depth-first
    (the work list is popped from the back) over the kept copy's neighbours."""
    logger.info("flip graph orientation..")
    seen = [-1] * g.num_vertices()

    def turn(e: int) -> None:
        s0, t0 = g.esrc[e], g.etgt[e]
        by_key.pop((g.vid[s0], ori[s0], g.vid[t0], ori[t0]))
        s = pair_of[g.vid[t0]][0] if ori[t0] == -1 else pair_of[g.vid[t0]][1]
        t = pair_of[g.vid[s0]][0] if ori[s0] == -1 else pair_of[g.vid[s0]][1]
        ovl = g.eovl[e]
        g.remove_edge(e)
        ne = g.add_edge(s, t, ovl, None, None)  # colour/flow: whatever the reused index holds
        by_key[(g.vid[s], ori[s], g.vid[t], ori[t])] = ne

    pick: Dict[str, str] = {}
    while depth:
        name = max(depth, key=depth.get)
        vp, vn = pair_of[name]
        seen[vp] = seen[vn] = 0
        work = [(pair_of[name], init_ori)]
        while work:
            (vp, vn), o = work.pop()
            depth.pop(g.vid[vp])
            if o == 1:
                keep, other, pick[g.vid[vp]] = vp, vn, "+"
            else:
                keep, other, pick[g.vid[vp]] = vn, vp, "-"
            for e in set(g.all_edges(other)):
                turn(e)
            seen[vp] = seen[vn] = 1
            for adj in g.all_neighbors(keep):
                if seen[adj] == -1:
                    ap, an = pair_of[g.vid[adj]]
                    seen[ap] = seen[an] = 0
                    work.append((pair_of[g.vid[adj]], ori[adj]))

    logger.info("final verifying graph..")
    assert len(pick) == len(pair_of)
    for name, choice in list(pick.items()):
        vp, vn = pair_of[name]
        other = vn if choice == "+" else vp
        if g.in_degree(other) + g.out_degree(other) > 0:
            pick[name] = "t"
    logger.info("Graph is verified")

    nodes: NodeMap = {}
    for name, choice in pick.items():
        vp, vn = pair_of[name]
        if choice == "+":
            nodes[name] = vp
        elif choice == "-":
            nodes["-" + name] = vn
            g.vid[vn] = "-" + name
        else:
            nodes[name] = vp
            nodes["-" + name] = vn
            g.vid[vn] = "-" + name
    edges: EdgeMap = {}
    for e in by_key.values():
        edges[(g.vid[g.esrc[e]], g.vid[g.etgt[e]])] = e
    logger.info("done")
    return nodes, edges


def _compact(src: AsmGraph, nodes: NodeMap, edges: EdgeMap):
    """``reduce_graph`` IO.py:272-295."""
    g = AsmGraph()
    nn: NodeMap = {}
    ne: EdgeMap = {}
    for name, v in nodes.items():
        nn[name] = g.add_vertex(src.vid[v], src.vdp[v], src.vseq[v], BLACK)
    for (u, w), e in edges.items():
        ne[(u, w)] = g.add_edge(nn[u], nn[w], src.eovl[e], src.eflow[e], BLACK)
    return g, nn, ne


def reindexing(g: AsmGraph, nodes: NodeMap, edges: EdgeMap):
    mapping: Dict[str, str] = {}
    new_nodes: NodeMap = {}
    new_edges: EdgeMap = {}
    for name, v in nodes.items():
        if g.vblack[v]:
            idx = str(len(mapping))
            mapping[name] = idx
            g.vid[v] = idx
            new_nodes[idx] = v
    for (u, w), e in edges.items():
        if g.eblack[e] and g.vblack[g.esrc[e]] and g.vblack[g.etgt[e]]:
            new_edges[(mapping[u], mapping[w])] = e
    return g, new_nodes, new_edges, mapping


def threshold_estimation(g: AsmGraph, logger):
    """Histogram rule of Preprocess.py:37-70 (the 128x64 inch plot it also draws is not restated)."""
    dps = [g.vdp[v] for v in range(g.num_vertices())]
    if max(dps) == min(dps):
        return 0.00
    regions, _ = numpy.histogram(dps, bins=int((max(dps) - min(dps)) // (0.05 * numpy.median(dps))))
    peak, _ = max(list(enumerate(regions)), key=lambda p: p[1])
    ratio = 0.00
    if peak == 0:
        ratio = 0.05
        for i in range(0, 4):
            if i >= len(regions):
                logger.warning("histogram is not properly set, reset cutoff to default (0.05*M)")
                ratio = 0.05
                break
            if regions[i] > regions[i + 1]:
                ratio += 0.05
            else:
                break
    return ratio * numpy.median(dps)


def graph_simplification(g: AsmGraph, nodes: NodeMap, edges: EdgeMap, contigs, logger, min_cov) -> None:
    logger.info("graph simplification")
    logger.debug("Total nodes: " + str(len(nodes)) + " Total edges: " + str(len(edges)))
    keep_nodes = contigs_by_node(contigs) if contigs is not None else {}
    keep_steps = contig_steps(contigs) if contigs is not None else {}
    for name, v in list(nodes.items()):
        if g.vdp[v] <= min_cov:
            if name in keep_nodes:
                continue
            nodes.pop(name)
            g.vblack[v] = GRAY
            for e in g.all_edges(v):
                key = (g.vid[g.esrc[e]], g.vid[g.etgt[e]])
                if key in keep_steps:
                    continue
                if key in edges:
                    g.eblack[edges.pop(key)] = GRAY
    logger.debug("Remain nodes: " + str(len(nodes)) + " Total edges: " + str(len(edges)))
    logger.info("done")


# ---- contigs.paths -----------------------------------------------------------------------------
def _path_is_valid(p: List[str], mapping: Dict[str, str], nodes: NodeMap, edges: EdgeMap) -> bool:
    if len(p) == 0:
        return False
    if len(p) == 1:
        return p[0] in mapping and mapping[p[0]] in nodes
    for a, b in zip(p, p[1:]):
        if a not in mapping or b not in mapping:
            return False
        ma, mb = mapping[a], mapping[b]
        if ma not in nodes or mb not in nodes or (ma, mb) not in edges:
            return False
    return True


def spades_paths_parser(g: AsmGraph, nodes: NodeMap, edges: EdgeMap, mapping: Dict[str, str], logger,
                        path_file: str, min_len: int = 250, min_cov=0):
    def signed(tokens: List[str]) -> List[str]:
        return [str(t[:-1]) if t[-1] == "+" else "-" + str(t[:-1]) for t in tokens]

    def take_paths(fh, line: str):
        subs: List[List[str]] = []
        total = 0

        def consider(tokens: List[str]) -> None:
            nonlocal total
            sp = signed(tokens)
            if _path_is_valid(list(dict.fromkeys(sp)), mapping, nodes, edges):
                mapped = [mapping[x] for x in sp]
                subs.append(mapped)
                total += len(mapped)

        while line.endswith(";\n"):
            consider(str(line[:-2]).split(","))
            line = fh.readline()
        consider(line.rstrip().split(","))
        return subs, total

    logger.info("parsing SPAdes .paths file..")
    contigs: ContigDict = {}
    info: Dict[str, tuple] = {}
    try:
        with open(path_file, "r") as fh:
            name = fh.readline()
            line = fh.readline()
            while name != "" and line != "":
                cno, clen, ccov = re.search("NODE_(.*)_length_(.*)_cov_(.*)", name.strip()).group(1, 2, 3)
                fwd, n_fwd = take_paths(fh, line)
                name_r = fh.readline()
                line_r = fh.readline()
                cno_r, clen_r, ccov_r = re.search("NODE_(.*)_length_(.*)_cov_(.*)'", name_r.strip()).group(1, 2, 3)
                rev, n_rev = take_paths(fh, line_r)
                if not (cno == cno_r and clen == clen_r and ccov == ccov_r):
                    raise BaseException
                name = fh.readline()
                line = fh.readline()
                segments, total = max([(fwd, n_fwd), (rev, n_rev)], key=lambda t: t[1])
                if segments == []:
                    continue
                if total < 2 and (float(ccov) <= min_cov or int(clen) < min_len):
                    continue
                for i, sub in enumerate(segments):
                    repeats: Dict[str, int] = {}
                    for x in sub:
                        repeats[x] = repeats.get(x, 0) + 1
                    sub = list(dict.fromkeys(sub))
                    if len(segments) != 1:
                        key = cno + "$" + str(i)
                        contigs[key] = [sub, path_length(g, [nodes[x] for x in sub]), float(ccov)]
                    else:
                        key = cno
                        contigs[key] = [sub, int(clen), float(ccov)]
                    info[key] = (None, repeats)
    except BaseException as err:  # noqa: B902 - mirrors the reference's catch-all
        logger.error("{0}\nPlease make sure the correct SPAdes contigs .paths file is provided.".format(err))
        logger.error("Pipeline aborted")
        sys.exit(1)
    logger.debug(str(contigs))
    logger.debug(str(info))
    logger.info("done")
    return contigs, info
