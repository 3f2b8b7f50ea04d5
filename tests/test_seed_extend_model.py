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

