"""The seed-and-extend formulation used on the device == the reference's window-by-window
lookup, checked per read end on every golden case (CPU only)."""
import os

import pytest

from conftest import pe_cases
from oracle import pe_oracle
import seed_extend_model as model


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_model_equals_oracle_per_end(name, d, meta):
    K = meta["k"] + 1
    ids, seqs = pe_oracle.read_gfa_segments(os.path.join(d, "graph.gfa"))
    reads = pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq")) + pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))
    tab = pe_oracle.build_table(seqs, K)
    lens = [len(s) for s in seqs]
    mtab, w, s = model.build(seqs, K)
    rcs = [model.rc(x) if len(x) >= K else "" for x in seqs]
    n = 0
    for r in reads:
        if len(r) < K:
            continue
        assert model.map_end(r, seqs, rcs, mtab, w, s, K) == pe_oracle.map_read_end(r, tab, lens, K), r
        n += 1
    assert n > 0 or name in ("empty_reads_k21",)



def test_every_probe_phase_is_exact_and_the_end_phase_is_the_shortest_grid():
    """Round 5: the probe grid starts at vs_seed_phase instead of 0.  Any phase reproduces the oracle (a match of K bases
    holds s consecutive seed starts); the one the kernels use needs floor((len - w + 1) / s) probes, never more than
    phase 0, and fewer for every length with (len - w) mod s < s - 1.  Reads of every length from K to K + 2 s + 3
    against a graph with repeats, substitutions and an N."""
    import random

    rng = random.Random(5)
    for K in (22, 32, 56):
        genome = "".join(rng.choice("ACGT") for _ in range(900))
        var = list(genome)
        for p in range(40, 900, 97):
            var[p] = {"A": "C", "C": "G", "G": "T", "T": "A"}[var[p]]
        var = "".join(var)
        seqs = [genome[i:i + K + 12] for i in range(0, 900 - K - 12, 9)] + [var[i:i + K + 30] for i in range(5, 900 - K - 30, 31)]
        seqs.append(model.rc(genome[300:300 + K + 5]))
        tab = pe_oracle.build_table(seqs, K)
        lens = [len(x) for x in seqs]
        mtab, w, s = model.build(seqs, K)
        rcs = [model.rc(x) for x in seqs]
        for rlen in range(K, K + 2 * s + 4):
            for rep in range(3):
                a = rng.randrange(0, 900 - rlen)
                read = list((genome if rep != 1 else var)[a:a + rlen])
                if rep == 2:
                    read[rng.randrange(rlen)] = rng.choice("ACGTN")
                read = "".join(read)
                if rng.random() < 0.5:
                    read = "".join({"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}[c] for c in reversed(read))
                want = pe_oracle.map_read_end(read, tab, lens, K)
                probes = []
                assert model.map_end(read, seqs, rcs, mtab, w, s, K, probes=probes) == want, (K, rlen, read)
                assert len(probes) == (rlen - w + 1) // s <= (rlen - w) // s + 1
                assert probes[0] < s and probes[-1] + w + s > rlen  # (no stride of the read without a probe)
                for first in range(s):
                    assert model.map_end(read, seqs, rcs, mtab, w, s, K, first=first) == want, (K, rlen, first, read)
                # the adaptive grids of the compile-time-shape kernels: every step point t
                n = max(1, (rlen - w + 1) // s)
                for t in range(n + 1):
                    grid = model.step_grid(rlen, w, s, t)
                    assert len(grid) == n and grid[0] < s and grid[-1] + w + s > rlen and all(b - a <= s for a, b in zip(grid, grid[1:]))
                    assert model.map_end(read, seqs, rcs, mtab, w, s, K, grid=grid) == want, (K, rlen, t, read)
