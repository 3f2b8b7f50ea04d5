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
