"""Seeded synthetic inputs for tests and for ``bench.py`` (host side, numpy only).

Nothing of the reference's evaluation data ships with it (its .gitignore drops every
*.fa/*.fq/*.gfa), so the build invents its own: a set of related strain genomes, the compacted
de Bruijn graph of those genomes in the shape SPAdes hands to VStrains after strand
canonisation (segments overlap by ``k``; every (k+1)-mer of a strain lies in exactly one
segment; all links ``+ … +`` as in ``s_graph_L1.gfa``, reference ``utils/VStrains_IO.py:345-369``),
single-node and chain contigs, and paired-end reads sampled from the strains.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

ALPHABET = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def revcomp(seq: str) -> str:
    return seq.encode().translate(_COMP)[::-1].decode()


@dataclass
class StrainSet:
    genomes: List[str]
    abundance: List[float]


@dataclass
class SynthGraph:
    k: int
    ids: List[str]
    seqs: List[str]
    dp: List[float]
    links: List[Tuple[int, int]]  # (from index, to index), overlap k
    strain_paths: List[List[int]] = field(default_factory=list)  # node indices per strain

    def gfa_text(self) -> str:
        out = []
        for i, s, d in zip(self.ids, self.seqs, self.dp):
            out.append("S\t%s\t%s\tDP:f:%s\n" % (i, s, repr(float(d))))
        for u, v in self.links:
            out.append("L\t%s\t+\t%s\t+\t%dM\n" % (self.ids[u], self.ids[v], self.k))
        return "".join(out)


def make_strains(
    n_strains: int,
    genome_len: int,
    snp_rate: float,
    seed: int,
    abundance_ratio: float = 0.8,
    total_depth: float = 1000.0,
) -> StrainSet:
    """One random ancestor; every strain carries its own random substitutions at ``snp_rate``
    plus substitutions shared with a random subset of the others (tree-less but gives shared
    and private variation).  Abundances are geometric."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    n_sites = int(round(genome_len * snp_rate))
    sites = rng.choice(genome_len, size=min(n_sites, genome_len), replace=False)
    genomes = np.repeat(anc[None, :], n_strains, axis=0)
    for pos in sites:
        carriers = rng.random(n_strains) < rng.uniform(0.15, 0.6)
        if not carriers.any():
            carriers[rng.integers(0, n_strains)] = True
        if carriers.all():
            carriers[rng.integers(0, n_strains)] = False
        alt = (anc[pos] + rng.integers(1, 4)) % 4
        genomes[carriers, pos] = alt
    ab = np.array([abundance_ratio ** i for i in range(n_strains)], dtype=np.float64)
    ab = ab / ab.sum() * total_depth
    return StrainSet([ALPHABET[g].tobytes().decode() for g in genomes], [float(x) for x in ab])


def compact_dbg(strains: StrainSet, k: int) -> SynthGraph:
    """Forward-strand compacted de Bruijn graph: vertices = k-mers, edges = (k+1)-mers,
    segments = maximal non-branching edge paths; neighbouring segments share exactly k bases."""
    succ: Dict[str, set] = {}
    pred: Dict[str, set] = {}
    weight: Dict[str, float] = {}
    for g, ab in zip(strains.genomes, strains.abundance):
        for i in range(len(g) - k):
            e = g[i : i + k + 1]
            weight[e] = weight.get(e, 0.0) + ab
            u, v = e[:k], e[1:]
            succ.setdefault(u, set()).add(e)
            pred.setdefault(v, set()).add(e)
            succ.setdefault(v, set())
            pred.setdefault(u, set())

    def passthrough(vtx: str) -> bool:
        return len(pred[vtx]) == 1 and len(succ[vtx]) == 1

    seqs: List[str] = []
    dps: List[float] = []
    edge_owner: Dict[str, int] = {}
    start_vertex: List[str] = []
    end_vertex: List[str] = []
    # deterministic order: walk the strains, start a segment at every unclaimed edge whose
    # source is a junction (or a strain start)
    for g in strains.genomes:
        for i in range(len(g) - k):
            e = g[i : i + k + 1]
            if e in edge_owner:
                continue
            u = e[:k]
            if passthrough(u):
                # interior edge: will be claimed when its segment's head is met; but a strain may
                # enter mid-segment only if the segment head precedes it in some strain
                continue
            idx = len(seqs)
            seq = e
            tot = weight[e]
            cnt = 1
            edge_owner[e] = idx
            cur = e[1:]
            while passthrough(cur):
                (nxt,) = succ[cur]
                if nxt in edge_owner:
                    break
                edge_owner[nxt] = idx
                seq += nxt[-1]
                tot += weight[nxt]
                cnt += 1
                cur = nxt[1:]
            seqs.append(seq)
            dps.append(tot / cnt)
            start_vertex.append(u)
            end_vertex.append(cur)
    # edges never reached (perfect cycles) are ignored: linear genomes do not make them
    by_start: Dict[str, List[int]] = {}
    for idx, u in enumerate(start_vertex):
        by_start.setdefault(u, []).append(idx)
    links: List[Tuple[int, int]] = []
    for idx, v in enumerate(end_vertex):
        for j in by_start.get(v, []):
            links.append((idx, j))
    paths: List[List[int]] = []
    for g in strains.genomes:
        p: List[int] = []
        for i in range(len(g) - k):
            o = edge_owner.get(g[i : i + k + 1])
            if o is not None and (not p or p[-1] != o):
                p.append(o)
        paths.append(p)
    ids = [str(i + 1) for i in range(len(seqs))]
    return SynthGraph(k, ids, seqs, dps, links, paths)


def sample_pairs(
    strains: StrainSet,
    n_pairs: int,
    read_len: int,
    seed: int,
    sub_rate: float = 0.0,
    n_rate: float = 0.0,
    frag_mean: float | None = None,
    frag_sd: float | None = None,
) -> Tuple[List[str], List[str]]:
    """Illumina-like pairs: fragment ~ N(3L, 0.3L) clipped to [L, genome]; forward read = first L
    bases, reverse read = reverse complement of the last L; the whole fragment is flipped with
    p = 0.5."""
    rng = np.random.default_rng(seed)
    frag_mean = 3.0 * read_len if frag_mean is None else frag_mean
    frag_sd = 0.3 * read_len if frag_sd is None else frag_sd
    prob = np.array(strains.abundance) / np.sum(strains.abundance)
    which = rng.choice(len(strains.genomes), size=n_pairs, p=prob)
    fwd: List[str] = []
    rve: List[str] = []
    for r in range(n_pairs):
        g = strains.genomes[which[r]]
        flen = int(np.clip(rng.normal(frag_mean, frag_sd), read_len, len(g)))
        start = int(rng.integers(0, len(g) - flen + 1))
        frag = g[start : start + flen]
        if rng.random() < 0.5:
            frag = revcomp(frag)
        a = frag[:read_len]
        b = revcomp(frag[-read_len:])
        if sub_rate > 0.0:
            a = _mutate(a, sub_rate, rng)
            b = _mutate(b, sub_rate, rng)
        if n_rate > 0.0 and rng.random() < n_rate:
            pos = int(rng.integers(0, read_len))
            if rng.random() < 0.5:
                a = a[:pos] + "N" + a[pos + 1 :]
            else:
                b = b[:pos] + "N" + b[pos + 1 :]
        fwd.append(a)
        rve.append(b)
    return fwd, rve


def _mutate(s: str, rate: float, rng) -> str:
    hits = np.nonzero(rng.random(len(s)) < rate)[0]
    if hits.size == 0:
        return s
    b = bytearray(s.encode())
    for p in hits:
        cur = b"ACGT".index(b[p])
        b[p] = b"ACGT"[(cur + int(rng.integers(1, 4))) % 4]
    return b.decode()


def fastq_text(reads: Sequence[str], tag: str, newline: str = "\n") -> str:
    out = []
    for i, s in enumerate(reads):
        out.append("@%s_%d%s%s%s+%s%s%s" % (tag, i, newline, s, newline, newline, "I" * len(s), newline))
    return "".join(out)


def contigs_paths_text(graph: SynthGraph, contigs: Sequence[Sequence[int]], covs: Sequence[float]) -> str:
    """SPAdes ``contigs.paths`` text (forward + primed reverse record per contig), the shape
    ``spades_paths_parser`` (reference ``utils/VStrains_IO.py:398-515``) reads."""
    out = []
    for c, (nodes, cov) in enumerate(zip(contigs, covs)):
        length = sum(len(graph.seqs[i]) for i in nodes) - graph.k * (len(nodes) - 1)
        name = "NODE_%d_length_%d_cov_%s" % (c + 1, length, repr(float(cov)))
        out.append(name + "\n")
        out.append(",".join(graph.ids[i] + "+" for i in nodes) + "\n")
        out.append(name + "'\n")
        out.append(",".join(graph.ids[i] + "-" for i in reversed(nodes)) + "\n")
    return "".join(out)


# ---------------------------------------------------------------------------------------------
# Whole-pipeline cases (assembly graph in SPAdes orientation conventions + contigs.paths + reads)
# ---------------------------------------------------------------------------------------------
@dataclass
class PipelineCase:
    gfa_text: str
    paths_text: str
    fwd: List[str]
    rve: List[str]
    k: int
    graph: SynthGraph
    strains: StrainSet


def spades_like_gfa(graph: SynthGraph, flip: Sequence[bool], swap: Sequence[bool], depth_tags: str = "dp",
                    self_loops: Sequence[int] = ()) -> str:
    """GFA1 text the way an assembler emits it: segment ``i`` is stored reverse-complemented when
    ``flip[i]``; link ``j`` is written in its reverse-complement form when ``swap[j]``.  Input of
    ``gfa_to_graph`` (reference ``utils/VStrains_IO.py:27-134``).  ``depth_tags="kc"`` writes the
    SPAdes style ``LN:i`` / ``KC:i`` pair instead of ``DP:f`` (:56-77); ``self_loops`` adds
    ``L x + x +`` records, which the reference answers by lower-casing the segment (:117-120)."""
    out = []
    for i, (name, s, d) in enumerate(zip(graph.ids, graph.seqs, graph.dp)):
        seq = revcomp(s) if flip[i] else s
        if depth_tags == "kc":
            out.append("S\t%s\t%s\tLN:i:%d\tKC:i:%d\n" % (name, seq, len(seq), max(1, int(round(d * len(seq))))))
        else:
            out.append("S\t%s\t%s\tDP:f:%s\n" % (name, seq, repr(float(d))))
    for i in self_loops:
        out.append("L\t%s\t+\t%s\t+\t%dM\n" % (graph.ids[i], graph.ids[i], graph.k))
    for j, (u, v) in enumerate(graph.links):
        ou = "-" if flip[u] else "+"
        ov = "-" if flip[v] else "+"
        if swap[j]:
            inv = {"+": "-", "-": "+"}
            out.append("L\t%s\t%s\t%s\t%s\t%dM\n" % (graph.ids[v], inv[ov], graph.ids[u], inv[ou], graph.k))
        else:
            out.append("L\t%s\t%s\t%s\t%s\t%dM\n" % (graph.ids[u], ou, graph.ids[v], ov, graph.k))
    return "".join(out)


def spades_like_paths(graph: SynthGraph, flip: Sequence[bool], contigs: Sequence[Sequence[int]],
                      covs: Sequence[float], gaps: Dict[int, int] = None) -> str:
    """``gaps[c] = g`` writes contig ``c`` as two sub-paths split after ``g`` nodes, the first line
    ending in ``;`` (SPAdes' notation for a gap; reference ``utils/VStrains_IO.py:412-442``)."""
    out = []
    sign = lambda i, fwd: ("+" if fwd else "-") if not flip[i] else ("-" if fwd else "+")  # noqa: E731
    gaps = gaps or {}

    def lines(nodes, fwd, cut):
        toks = [graph.ids[i] + sign(i, fwd) for i in nodes]
        if cut is None or cut <= 0 or cut >= len(toks):
            return ",".join(toks) + "\n"
        return ",".join(toks[:cut]) + ";\n" + ",".join(toks[cut:]) + "\n"

    for c, (nodes, cov) in enumerate(zip(contigs, covs)):
        length = sum(len(graph.seqs[i]) for i in nodes) - graph.k * (len(nodes) - 1)
        name = "NODE_%d_length_%d_cov_%s" % (c + 1, length, repr(float(cov)))
        cut = gaps.get(c)
        out.append(name + "\n")
        out.append(lines(list(nodes), True, cut))
        out.append(name + "'\n")
        out.append(lines(list(reversed(nodes)), False, None if cut is None else len(nodes) - cut))
    return "".join(out)


def make_pipeline_case(
    n_strains: int,
    genome_len: int,
    snp_rate: float,
    k: int,
    n_pairs: int,
    read_len: int,
    seed: int,
    abundance_ratio: float = 0.8,
    total_depth: float = 1000.0,
    dp_noise: float = 0.02,
    scramble: bool = False,
    error_strain_depth: float = 0.0,
    contig_pieces: int = 3,
    sub_rate: float = 0.0,
    repeat_len: int = 0,
    depth_tags: str = "dp",
    gapped_contigs: int = 0,
    self_loops: int = 0,
    circular: bool = False,
) -> PipelineCase:
    """A seeded, self-contained stand-in for "SPAdes output + reads" of a viral quasispecies."""
    rng = np.random.default_rng(seed + 77)
    st = make_strains(n_strains, genome_len, snp_rate, seed, abundance_ratio, total_depth)
    if repeat_len > 0:
        # one exact tandem-free repeat: copy a block from the first third into the last third
        a = genome_len // 5
        b = 3 * genome_len // 5
        st = StrainSet([g[:b] + g[a : a + repeat_len] + g[b + repeat_len :] for g in st.genomes], st.abundance)
    if circular:
        # circular genomes: the graph closes on itself (k bases of the start repeated at the end),
        # reads are sampled across the junction too
        st = StrainSet([g + g[: 3 * read_len + k] for g in st.genomes], st.abundance)
    reads_from = st
    graph_from = st
    if circular:
        graph_from = StrainSet([g[: len(g) - 3 * read_len] for g in st.genomes], st.abundance)
        st_graph_only = graph_from
    if error_strain_depth > 0.0:
        g0 = st.genomes[0]
        arr = bytearray(g0.encode())
        for pos in rng.choice(len(arr), size=max(2, len(arr) // 700), replace=False):
            arr[pos] = b"ACGT"[(b"ACGT".index(arr[pos]) + 1 + int(rng.integers(0, 3))) % 4]
        graph_from = StrainSet(st.genomes + [arr.decode()], st.abundance + [error_strain_depth])
    g = compact_dbg(graph_from, k)
    if dp_noise > 0.0:
        g.dp = [float(d * (1.0 + dp_noise * rng.standard_normal())) for d in g.dp]
    n = len(g.ids)
    flip = [bool(scramble and rng.random() < 0.5) for _ in range(n)]
    swap = [bool(scramble and rng.random() < 0.5) for _ in g.links]
    contigs: List[List[int]] = [[i] for i in range(n)]
    covs: List[float] = [g.dp[i] for i in range(n)]
    for s, p in enumerate(g.strain_paths[:n_strains]):
        if len(p) < 2:
            continue
        for _ in range(contig_pieces):
            ln = int(rng.integers(2, min(len(p), 8) + 1))
            a = int(rng.integers(0, len(p) - ln + 1))
            piece = list(p[a : a + ln])
            contigs.append(piece)
            covs.append(float(min(g.dp[i] for i in piece)))
    fwd, rve = sample_pairs(reads_from, n_pairs, read_len, seed + 1, sub_rate=sub_rate)
    gaps: Dict[int, int] = {}
    multi = [c for c, nodes in enumerate(contigs) if len(nodes) >= 4]
    for c in multi[:gapped_contigs]:
        gaps[c] = len(contigs[c]) // 2
    loops = [int(x) for x in rng.choice(n, size=min(self_loops, n), replace=False)] if self_loops else []
    return PipelineCase(spades_like_gfa(g, flip, swap, depth_tags, loops),
                        spades_like_paths(g, flip, contigs, covs, gaps), fwd, rve, k, g, st)
