"""The graph-following mapping (csrc/vs_walk.h) on the CPU: the certification of node sets and the per-end walk itself
-- `vs_walk_map_ends_host` runs the very function k_pe_walk runs per lane, compiled for the host -- against the oracle's
single_end_read_mapping (PE_Inference.py:16-48).  Integer work: the accepted node lists must be identical."""
import os

import numpy as np
import pytest

from conftest import pe_cases
from oracle import pe_oracle, pe_oracle_c
from vstrains_amd import pe as host
from vstrains_amd import synth


def _check(seqs, k, reads, must_certify=True, min_mapped=1):
    got = host.walk_map_ends_host(seqs, k, reads, cap=16)
    if got is None:
        assert not must_certify, host.certify_walk_host(seqs, k)
        return 0
    orc = pe_oracle_c.Oracle(seqs, k)
    n = 0
    for read, g in zip(reads, got):
        want = sorted(orc.map_end(read)) if len(read) >= k + 1 else []
        if g is None:  # the kernel would hand this pair to the overflow kernel: allowed, but it must be rare
            continue
        n += 1
        assert g == want, (read, g, want)
    assert n >= min_mapped
    return n


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_host_walk_equals_oracle_on_golden_reads(name, d, meta):
    ids, seqs = pe_oracle.read_gfa_segments(os.path.join(d, "graph.gfa"))
    reads = [s.encode("ascii", "replace").decode() for s in
             pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq")) + pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))]
    cert = host.certify_walk_host(seqs, meta["k"])
    if name in ("palindrome_k5", "odd_split_k6", "repeats_k9", "tiny_k1", "tiny_k2"):
        assert not cert["certified"] and "more than once" in cert["why_not"]  # repeats / palindromes: the seed kernels map these
        assert host.walk_map_ends_host(seqs, meta["k"], reads[:4]) is None
        return
    assert cert["certified"], cert
    if reads:
        _check(seqs, meta["k"], reads, min_mapped=0 if name in ("empty_reads_k21", "all_short_k128") else min(10, len(reads)))


def test_certification_refuses_what_breaks_the_walk():
    rng = np.random.default_rng(3)

    def rand(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))

    k = 11
    a, b = rand(60), rand(60)
    assert host.certify_walk_host([a, b], k)["certified"]
    # the same (k+1)-mer in two nodes; in one node twice; on both strands (reverse complement of another node's stretch)
    assert not host.certify_walk_host([a, b[:20] + a[10:30] + b[40:]], k)["certified"]
    assert not host.certify_walk_host([a[:40] + a[5:30]], k)["certified"]
    assert not host.certify_walk_host([a, b[:20] + synth.revcomp(a[10:30]) + b[40:]], k)["certified"]
    # a palindromic (k+1)-mer (k + 1 even): the reference holds its entry twice
    pal = "ACGTAC" + "GTACGT"
    assert pal == synth.revcomp(pal)
    assert not host.certify_walk_host([rand(20) + pal + rand(20)], k)["certified"]
    # overlaps that are not node ends: node c continues out of the MIDDLE of a (its first k bases lie inside a): (C2)
    c = a[25:25 + k] + rand(30)
    r = host.certify_walk_host([a, c], k)
    assert not r["certified"] and "inside a node" in r["why_not"], r
    # a proper chain and a proper bubble certify, with the successor links they imply (both strands)
    chain = [a, a[-k:] + rand(30)]
    assert host.certify_walk_host(chain, k)["successor_links"] == 2
    stem, tail = rand(50), rand(50)
    mid1 = stem[-k:] + "A" + tail[:k]
    mid2 = stem[-k:] + "C" + tail[:k]
    r = host.certify_walk_host([stem, mid1, mid2, tail], k)
    assert r["certified"] and r["successor_links"] == 8, r
    # nodes shorter than k + 1 hold no (k+1)-mer: any bytes, never met
    assert host.certify_walk_host([a, "acgtnn"], k)["certified"]
    # (k+1)-mers beyond 160 bases stay with the seed kernels
    assert not host.certify_walk_host([rand(400)], 200)["certified"]


@pytest.mark.parametrize("k,read_len,seed", [(21, 100, 1), (55, 150, 2), (31, 125, 3), (127, 250, 4), (63, 150, 5), (64, 151, 6), (5, 30, 7),
                                              (95, 200, 8), (96, 200, 9), (32, 90, 10), (33, 90, 11), (128, 251, 12), (140, 300, 13)])
def test_host_walk_random_graphs_with_errors_and_dirty_bytes(k, read_len, seed):
    """Compacted de Bruijn graphs of random strains; reads with substitutions (runs break and resume), with N and other
    bytes outside ACGT (clean stretches), of ragged lengths, from both strands; k around the 64-bit word boundaries of
    the (k+1)-mer (32/33, 63/64, 95/96, 127/128)."""
    rng = np.random.default_rng(seed)
    st = synth.make_strains(4, max(8 * read_len, 1200), 0.03 if k > 20 else 0.01, seed=100 + seed)
    g = synth.compact_dbg(st, k)
    cert = host.certify_walk_host(g.seqs, k)
    if not cert["certified"]:
        pytest.skip("random strains with a repeated (k+1)-mer: " + cert["why_not"])
    f, r = synth.sample_pairs(st, 400, read_len, seed=200 + seed, sub_rate=0.01)
    reads = f + r
    dirty = list("NnRYacgt*-")
    for i in rng.choice(len(reads), size=120, replace=False):
        s = reads[int(i)]
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(0, len(s)))
            s = s[:p] + dirty[int(rng.integers(0, len(dirty)))] + s[p + 1:]
        reads[int(i)] = s
    for i in rng.choice(len(reads), size=60, replace=False):
        reads[int(i)] = reads[int(i)][: int(rng.integers(1, read_len + 1))]
    reads.append(reads[0][:k] )          # shorter than k + 1
    reads.append(reads[1][:k + 1])       # one window
    reads.append("A" * read_len)         # nothing to find (or a homopolymer node)
    reads.append("".join("ACGT"[i] for i in rng.integers(0, 4, size=read_len)))  # off the graph entirely
    n = _check(g.seqs, k, reads, min_mapped=600)
    assert n >= len(reads) - 40  # hardly any end needs the general path


def test_host_walk_after_graph_simplification_and_with_chimeric_reads():
    """The node set PE inference really runs on (s_graph_L1: strands canonised, low-coverage nodes removed) still
    certifies, reads that cross a removed node resume behind it, and reads glued from two loci find both."""
    import tempfile

    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[0]
    with tempfile.TemporaryDirectory() as tmp:
        st, pre, names, seqs, cum, logger, n_in = workload_for(0, tmp)
    seqs = list(seqs)
    assert host.certify_walk_host(seqs, cfg["k"])["certified"]
    f, r = synth.sample_pairs(st, 500, cfg["read_len"], seed=9, sub_rate=0.005)
    reads = f + r
    rng = np.random.default_rng(4)
    for _ in range(200):  # chimeras
        a, b = reads[int(rng.integers(0, len(reads)))], reads[int(rng.integers(0, len(reads)))]
        cut = int(rng.integers(20, 130))
        reads.append(a[:cut] + b[cut:])
    # and the graph with every fifth node dropped (what a coverage cut-off does): still certified, still exact
    kept = [s for i, s in enumerate(seqs) if i % 5]
    assert host.certify_walk_host(kept, cfg["k"])["certified"]
    _check(seqs, cfg["k"], reads, min_mapped=1000)
    _check(kept, cfg["k"], reads, min_mapped=1000)
