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


def _check_grouped(seqs, reads, K, stats=None):
    tab = pe_oracle.build_table(seqs, K)
    lens = [len(s) for s in seqs]
    groups, w, s = model.build_groups(seqs, K)
    rcs = [model.rc(x) if len(x) >= K else "" for x in seqs]
    n = 0
    for r in reads:
        if len(r) < K:
            continue
        assert model.map_end_grouped(r, seqs, rcs, groups, w, s, K, stats) == pe_oracle.map_read_end(r, tab, lens, K), r
        n += 1
    return n


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_grouped_model_equals_oracle_per_end(name, d, meta):
    """Round 2's formulation (one comparison per seed and side against the group's reference
    posting, the other postings decided from their stored agreement with it) == the reference's
    window-by-window lookup, per read end, on every golden case."""
    K = meta["k"] + 1
    ids, seqs = pe_oracle.read_gfa_segments(os.path.join(d, "graph.gfa"))
    reads = pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq")) + pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))
    n = _check_grouped(seqs, reads, K)
    assert n > 0 or name in ("empty_reads_k21",)


@pytest.mark.parametrize("k,read_len,snp,cap", [(55, 150, 0.06, 255), (21, 90, 0.08, 255), (31, 100, 0.05, 3), (11, 60, 0.2, 255), (55, 150, 0.03, 7)])
def test_grouped_model_on_dense_graphs_with_dirty_reads(k, read_len, snp, cap, monkeypatch):
    """Dense variation (many postings per seed), substitutions, reads with bytes outside ACGT, and
    tiny LCP field widths (cap) so that the 'agreement longer than the field can say' branch runs."""
    import numpy as np

    from vstrains_amd import synth

    monkeypatch.setattr(model, "LCP_CAP", cap)
    st = synth.make_strains(8, 1500, snp, seed=1000 + k)
    g = synth.compact_dbg(st, k)
    f, r = synth.sample_pairs(st, 150, read_len, seed=7 + k, sub_rate=0.01)
    reads = f + r
    rng = np.random.default_rng(k)
    for i in rng.choice(len(reads), size=60, replace=False):
        s_ = reads[int(i)]
        for _ in range(int(rng.integers(1, 4))):
            p_ = int(rng.integers(0, len(s_)))
            s_ = s_[:p_] + str(rng.choice(list("nRYacgt*"))) + s_[p_ + 1:]
        reads[int(i)] = s_
    stats = {}
    n = _check_grouped(g.seqs, reads, k + 1, stats)
    assert n == len(reads)
    # the point of the exercise: few postings need a comparison of their own
    if cap == 255:
        assert stats["own"] < 0.5 * stats["members"], stats
