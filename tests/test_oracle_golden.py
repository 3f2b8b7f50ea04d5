"""The Python oracle vs. the outputs of the real reference script (tests/golden/pe)."""
import json
import os

import pytest

from conftest import pe_cases
from oracle import pe_oracle


def _read(path):
    with open(path, "r", newline="") as fh:
        return fh.read()


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_python_oracle_matches_reference_files(name, d, meta):
    pe, st, stats = pe_oracle.run_files(
        os.path.join(d, "graph.gfa"), os.path.join(d, "fwd.fq"), os.path.join(d, "rve.fq"), meta["k"])
    assert pe == _read(os.path.join(d, "pe_info"))
    assert st == _read(os.path.join(d, "st_info"))


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_integer_acceptance_test_equals_float_form(name, d, meta):
    ids, seqs = pe_oracle.read_gfa_segments(os.path.join(d, "graph.gfa"))
    f = pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq"))
    r = pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))
    a = pe_oracle.pe_matrices(seqs, f, r, meta["k"], mapper=pe_oracle.map_read_end)
    b = pe_oracle.pe_matrices(seqs, f, r, meta["k"], mapper=pe_oracle.map_read_end_int)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2] == b[2]


def test_lowercase_node_raises_like_reference():
    (case,) = [c for c in pe_cases(ok_only=False) if c[2]["returncode"] != 0]
    name, d, meta = case
    with pytest.raises(KeyError) as ei:
        pe_oracle.run_files(os.path.join(d, "graph.gfa"), os.path.join(d, "fwd.fq"), os.path.join(d, "rve.fq"), meta["k"])
    assert "KeyError: %r" % ei.value.args[0] == meta["stderr_last"]


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_c_oracle_matches_reference_files(name, d, meta):
    from oracle import pe_oracle_c

    ids, seqs = pe_oracle.read_gfa_segments(os.path.join(d, "graph.gfa"))
    f = pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq"))
    r = pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))
    orc = pe_oracle_c.Oracle(seqs, meta["k"])
    node_mat, short_mat, stats = orc.count_pairs(f, r)
    assert pe_oracle.matrix_text(ids, node_mat) == _read(os.path.join(d, "pe_info"))
    assert pe_oracle.matrix_text(ids, short_mat) == _read(os.path.join(d, "st_info"))
    ref = pe_oracle.pe_matrices(seqs, f, r, meta["k"])
    assert tuple(int(x) for x in stats) == ref[2]
    # per-end lists too
    tab = pe_oracle.build_table(seqs, meta["k"] + 1)
    lens = [len(s) for s in seqs]
    for s in (f + r)[:200]:
        if len(s) >= meta["k"] + 1:
            assert orc.map_end(s) == pe_oracle.map_read_end(s, tab, lens, meta["k"] + 1)


def test_c_oracle_lowercase_node_keyerror():
    from oracle import pe_oracle_c

    (case,) = [c for c in pe_cases(ok_only=False) if c[2]["returncode"] != 0]
    name, d, meta = case
    ids, seqs = pe_oracle.read_gfa_segments(os.path.join(d, "graph.gfa"))
    with pytest.raises(KeyError) as ei:
        pe_oracle_c.Oracle(seqs, meta["k"])
    assert "KeyError: %r" % ei.value.args[0] == meta["stderr_last"]


def test_c_port_equals_the_real_reference_function_at_config4_size(tmp_path):
    """The C restatement against lists the REAL ``single_end_read_mapping`` returned at the graph size of BASELINE configs[4]
    (54 465 nodes; tests/golden/make_pe_end_lists.py imported the function from /root/reference): the port is the checker of
    the configs[4] GPU tests, so it is pinned at that size too, not only on the small golden graphs."""
    import hashlib
    import json

    from conftest import GOLDEN
    from oracle import pe_oracle_c
    from vstrains_amd.workloads import CONFIGS, workload_for

    with open(os.path.join(GOLDEN, "pe_end_lists_config4.json")) as fh:
        want = json.load(fh)
    cfg = CONFIGS[4]
    st, pre, names, seqs, cum, logger, n_in = workload_for(4, str(tmp_path))
    with open(os.path.join(str(tmp_path), "gfa", "s_graph_L1.gfa"), "rb") as fh:
        assert hashlib.sha256(fh.read()).hexdigest() == want["s_graph_L1_gfa_sha256"]
    orc = pe_oracle_c.Oracle(list(seqs), cfg["k"])
    fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, want["stream_seed"], 0, want["pairs"], cfg["read_len"], want["sub_thresh"], want["n_thresh"])
    for p in range(want["pairs"]):
        for side, arr in ((0, fw), (1, rv)):
            w = want["lists"][2 * p + side]
            if w is not None:
                assert sorted(orc.map_end(arr[p].tobytes().decode())) == sorted(w), (p, side)
