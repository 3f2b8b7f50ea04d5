"""File formats on the hot path: the intermediate GFA1 subset, ``.paths``, FASTA.

Byte-compatible with what the reference writes and re-reads between stages
(``utils/VStrains_IO.py``: ``graph_to_gfa`` :337-372, ``flipped_gfa_to_graph`` :298-334,
``contig_dict_to_fasta`` :518-536, ``contig_dict_to_path`` :558-595).  gfapy is replaced by a
tab split that keeps file order (its segment / edge views are file-ordered).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

from .asm_graph import BLACK, AsmGraph, EdgeMap, NodeMap


def gfa_records(path: str) -> Tuple[List[List[str]], List[List[str]]]:
    """(segment field lists, link field lists) in file order."""
    segs: List[List[str]] = []
    links: List[List[str]] = []
    with open(path, "r") as fh:
        for raw in fh:
            raw = raw.rstrip("\n").rstrip("\r")
            if raw.startswith("S\t"):
                segs.append(raw.split("\t"))
            elif raw.startswith("L\t"):
                links.append(raw.split("\t"))
    return segs, links


def write_stage_gfa(g: AsmGraph, nodes: NodeMap, edges: EdgeMap, filename: str) -> None:
    """``graph_to_gfa``: black vertices in map order, then black edges between black, mapped
    vertices in map order; all orientations ``+``; dp printed with Python's shortest repr."""
    chunks: List[str] = []
    for v in nodes.values():
        if g.vblack[v]:
            chunks.append("S\t%s\t%s\tDP:f:%s\n" % (g.vid[v], g.vseq[v], repr(g.vdp[v])))
    for (u, w), e in edges.items():
        vu = nodes.get(u)
        vw = nodes.get(w)
        if vu is None or vw is None:
            continue
        if not (g.vblack[vu] and g.vblack[vw] and g.eblack[e]):
            continue
        chunks.append("L\t%s\t+\t%s\t+\t%dM\n" % (u, w, g.eovl[e]))
    with open(filename, "w") as fh:
        fh.write("".join(chunks))


def read_stage_gfa(filename: str) -> Tuple[AsmGraph, NodeMap, EdgeMap]:
    """``flipped_gfa_to_graph``: every vertex first (file order), then every edge (file order)."""
    segs, links = gfa_records(filename)
    g = AsmGraph()
    nodes: NodeMap = {}
    edges: EdgeMap = {}
    for rec in segs:
        _, name, seq, tag = rec  # exactly four fields, as the reference unpacks them
        nodes[name] = g.add_vertex(name, float(tag.split(":")[2]), seq, BLACK)
    for rec in links:
        _, left, ori_l, right, ori_r, ovl = rec
        assert ovl[-1] == "M" and ori_l == ori_r
        edges[(left, right)] = g.add_edge(nodes[left], nodes[right], int(ovl[:-1]), None, BLACK)
    return g, nodes, edges


def stage_graph_from_state(g: AsmGraph, nodes: NodeMap, edges: EdgeMap, gfa_path: Optional[str] = None,
                           want_text: bool = False):
    """The graph ``read_stage_gfa(write_stage_gfa(...))`` would give, without the file: same
    filtering and order; ``float(repr(x)) == x`` so dp survives exactly.  This runs once per
    re-initialisation (~100 times per run), so the rows are built in place instead of through
    ``add_vertex`` / ``add_edge`` calls -- same placement rule as ``AsmGraph.add_edge``.
    With ``gfa_path`` the stage GFA (``write_stage_gfa``'s bytes) is written in the same pass."""
    ng = AsmGraph()
    nn: NodeMap = {}
    ne: EdgeMap = {}
    vblack, vid, vdp, vseq = g.vblack, g.vid, g.vdp, g.vseq
    n_vid, n_vdp, n_vseq = ng.vid, ng.vdp, ng.vseq
    chunks: Optional[List[str]] = [] if gfa_path is not None else None
    for v in nodes.values():
        if vblack[v]:
            nn[vid[v]] = len(n_vid)
            n_vid.append(vid[v])
            n_vdp.append(vdp[v])
            n_vseq.append(vseq[v])
            if chunks is not None:
                chunks.append("S\t%s\t%s\tDP:f:%s\n" % (vid[v], vseq[v], repr(vdp[v])))
    nv = len(n_vid)
    ng.vblack = [BLACK] * nv
    adj = [[] for _ in range(nv)]
    nout = [0] * nv
    esrc, etgt, eovl = ng.esrc, ng.etgt, ng.eovl
    eblack, eovl_src = g.eblack, g.eovl
    get = nn.get
    for key, e in edges.items():
        s = get(key[0])
        t = get(key[1])
        if s is None or t is None or not eblack[e]:
            continue
        # (a vertex missing from ``nn`` is either unmapped or gray: the reference's three tests)
        ei = len(esrc)
        esrc.append(s)
        etgt.append(t)
        eovl.append(eovl_src[e])
        row = adj[s]
        slot = nout[s]
        if slot < len(row):
            row.append(row[slot])
            row[slot] = (t, ei)
        else:
            row.append((t, ei))
        nout[s] = slot + 1
        adj[t].append((s, ei))
        ne[key] = ei
        if chunks is not None:
            chunks.append("L\t%s\t+\t%s\t+\t%dM\n" % (key[0], key[1], eovl_src[e]))
    ne_count = len(esrc)
    ng.adj = adj
    ng.nout = nout
    ng.eflow = [0.0] * ne_count
    ng.eblack = [BLACK] * ne_count
    ng._n_edges = ne_count
    text = None
    if chunks is not None:
        text = "".join(chunks)
        with open(gfa_path, "w") as fh:
            fh.write(text)
    if want_text:
        return ng, nn, ne, text
    return ng, nn, ne


# ---- contig / strain records -----------------------------------------------------------------
ContigDict = Dict[str, list]  # name -> [node id list, length, coverage]


def _by_length_desc(contigs: ContigDict):
    return sorted(contigs.items(), key=lambda kv: kv[1][1], reverse=True)


def path_sequence(g: AsmGraph, path: List[int]) -> str:
    """Overlap-aware concatenation (``path_to_seq`` Utilities.py:909-921): consecutive vertices
    must be joined by an edge."""
    parts: List[str] = []
    last = len(path) - 1
    for i, u in enumerate(path):
        seq = g.vseq[u]
        if i != last:
            ovl = g.eovl[g.edge(u, path[i + 1])]
            if ovl != 0:
                seq = seq[:-ovl]
        parts.append(seq)
    return "".join(parts)


def path_ids_sequence(g: AsmGraph, ids: List[str], nodes: NodeMap) -> str:
    """``path_ids_to_seq`` Utilities.py:893-906: a missing edge counts as overlap 0."""
    parts: List[str] = []
    last = len(ids) - 1
    for i, name in enumerate(ids):
        u = nodes[name]
        seq = g.vseq[u]
        if i != last:
            e = g.edge(u, nodes[ids[i + 1]])
            ovl = g.eovl[e] if e is not None else 0
            if ovl != 0:
                seq = seq[:-ovl]
        parts.append(seq)
    return "".join(parts)


def path_length(g: AsmGraph, path: List[int]) -> int:
    """``path_len`` Utilities.py:839-850."""
    total = sum(len(g.vseq[u]) for u in path)
    for i in range(len(path) - 1):
        e = g.edge(path[i], path[i + 1])
        if e is not None:
            total -= g.eovl[e]
    return total


def write_contig_fasta(g: AsmGraph, nodes: NodeMap, contigs: ContigDict, filename: str) -> None:
    with open(filename, "w") as fh:
        for name, (ids, length, cov) in _by_length_desc(contigs):
            fh.write(">" + str(name) + "_" + str(length) + "_" + str(round(cov, 2)) + "\n")
            fh.write(path_ids_sequence(g, ids, nodes) + "\n")


def _origin(iid: str) -> str:
    star = iid.find("*")
    return iid if star == -1 else iid[:star]


def write_contig_paths(contigs: ContigDict, filename: str, id_mapping: Optional[Dict[str, str]] = None,
                       keep_original: bool = False) -> None:
    back: Dict[str, str] = {}
    if id_mapping is not None:
        for orig, mapped in id_mapping.items():
            back[mapped] = orig
    with open(filename, "w") as fh:
        for name, (ids, length, cov) in _by_length_desc(contigs):
            fh.write("NODE_" + str(name) + "_" + str(length) + "_" + str(cov) + "\n")
            out: List[str] = []
            for nid in ids:
                for iid in str(nid).split("&"):
                    if keep_original:
                        rid = back[_origin(iid)]
                        if rid[0] == "-":
                            rid = rid[1:] + "-"
                    else:
                        rid = str(_origin(iid))
                    out.append(rid)
            # the reference builds "a,b,c," and drops the last char; an empty contig would drop
            # nothing but the newline's predecessor -- keep that corner identical
            body = "".join(x + "," for x in out)
            fh.write(body[:-1] + "\n")


def read_pe_text(path: str) -> Iterable[Tuple[str, str, int]]:
    """Lines ``u:v:count`` up to the first empty line (``process_pe_info`` IO.py:603-612)."""
    with open(path, "r") as fh:
        for line in fh:
            if line == "\n":
                break
            u, v, mark = line[:-1].split(":")[:3]
            yield u, v, int(mark)
