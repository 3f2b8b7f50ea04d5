"""Every BASELINE.json config on the device.  configs[0] / configs[1]: ALL pairs of the config (100 k / 1 M) against
the C oracle; configs[0..2]: the HIP path against digests of the pe_info / st_info files the REAL reference script
wrote for the bench stream (tests/golden/reference_digests.json, made by tools/time_reference.py);
configs[3] and configs[4] at their real graph sizes (the generators bench.py
--config uses): the code paths only these sizes take -- the counters by row owners (k_list_owners .. k_rows_sum) beyond
46 340 nodes, the multi-pass locus sort above 36 k nodes, 2 x 12 GB counters with dirty-tile tracking, the generic-loop mapping
kernel for k = 127 / 2 x 250 -- against the C oracle on a prefix of the read stream, plus the
size-independent partition property on the whole block.  Integer work: bit-exact."""
import os

import numpy as np
import pytest

from oracle import pe_oracle_c

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    from vstrains_amd import pe as host

    return host


@pytest.fixture(scope="module")
def xctx(host):
    """A context in experiment mode: the tuning switches of vs_pe_count are live on it (conftest.experiment_context)."""
    from conftest import experiment_context

    c = experiment_context(host)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx(host):
    c = host.Context(0)
    yield c
    c.close()


def _count(host, ctx, st, cum, seed, L, pieces, sub, nth):
    counter = host.PeCounter(ctx)
    for first, n in pieces:
        block = ctx.synth_pairs(st.genomes, cum, seed, first, n, L, sub, nth)
        counter.add(block)
        ctx.sync()
        block.free()
    return counter


def _assert_equals_oracle(counter, orc, st, cum, seed, L, M, sub, nth):
    import torch

    fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, seed, 0, M, L, sub, nth)
    off = np.arange(M + 1, dtype=np.uint64) * np.uint64(L)
    node_cells, node_counts, short_cells, short_counts, ref_stats = orc.count_pairs_sparse(fw.reshape(-1), off, rv.reshape(-1), off, M)
    assert tuple(int(x) for x in counter.stats.cpu().tolist()) == tuple(int(x) for x in ref_stats)
    for mat, cells, counts in ((0, node_cells, node_counts), (1, short_cells, short_counts)):
        flat = counter.mats[mat].reshape(-1)
        n = counter.n
        cells = counter.ctx.internal_cells(mat, cells.astype(np.int64) // n, cells.astype(np.int64) % n)  # (the index's numbering)
        got = flat[torch.from_numpy(cells).to(flat.device)].cpu().numpy().view(np.uint32).astype(np.int64)
        assert np.array_equal(got, counts), "mat %d: %d cells differ" % (mat, int((got != counts).sum()))
        # no count anywhere else: the totals agree
        assert int(flat.sum(dtype=torch.int64).item()) == int(counts.sum())
    assert int(node_counts.sum()) > M  # the sample says something


def test_config4_50k_nodes_row_owner_counters(host, ctx, tmp_path):
    import torch

    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[4]
    st, pre, names, seqs, cum, logger, n_in = workload_for(4, str(tmp_path))
    n = len(seqs)
    assert n >= 50000, n  # > 46340: 2*N*N no longer fits the pair-major kernel's cell keys -> row owners; > 36862: several passes of the locus sort
    ctx.build_index(seqs, cfg["k"])
    L, seed = cfg["read_len"], 4242
    sub, nth = int(0.005 * 2 ** 32), int(0.001 * 2 ** 32)
    R = 3_000_000
    whole = _count(host, ctx, st, cum, seed, L, [(0, R)], sub, nth)
    t = ctx.last_timing()
    assert ctx.last_kernel.startswith("k_pe_tiles<1")  # the straight-line instantiation serves this config
    assert ctx.last_launched & ctx.RAN_ROW_OWNERS, "a graph beyond 46 340 nodes counts node_mat by row owners (k_node_rows)"
    parts = _count(host, ctx, st, cum, seed, L, [(0, 1), (1, 4999), (5000, 1_000_001), (1_005_001, R - 1_005_001)], sub, nth)
    assert torch.equal(whole.mats, parts.mats) and torch.equal(whole.stats, parts.stats)
    stats = whole.stats.cpu().tolist()
    assert sum(stats) == R and stats[2] > 0.99 * R
    del parts
    torch.cuda.empty_cache()
    # short_mat is upper triangular (PE_Inference.py:174-184): nothing below the diagonal
    low = 0
    rows = torch.arange(n, device=whole.mats.device)
    for lo in range(0, n, 2048):
        hi = min(n, lo + 2048)
        blk = whole.mats[1, lo:hi, :hi]
        below = rows[lo:hi, None] > rows[None, :hi]
        low += int((blk * below).sum(dtype=torch.int64).item())
    assert low == 0
    del whole
    torch.cuda.empty_cache()
    M = 120_000
    prefix = _count(host, ctx, st, cum, seed, L, [(0, M)], sub, nth)
    orc = pe_oracle_c.Oracle(seqs, cfg["k"])
    _assert_equals_oracle(prefix, orc, st, cum, seed, L, M, sub, nth)
    assert t["slow_pairs"] >= 0


def test_row_owner_counting_equals_plain_atomics_at_config4_size(host, xctx, tmp_path, monkeypatch):
    """Both counters by row owners (k_list_owners .. k_rows_sum; the default beyond 46 340 nodes) against one global atomic per
    increment (VS_ACC_ROWS=0: at this size no table shape of the pair-major kernel exists) on 2 M pairs of configs[4]'s
    graph: default strips, strips of 1 and 64 rows with a cell table that spills constantly, input order, no list table, a
    crowded list table, several transpositions with several histogram passes each.  Every non-zero cell lies in a marked
    tile."""
    import torch

    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[4]
    ctx = xctx
    st, pre, names, seqs, cum, logger, n_in = workload_for(4, str(tmp_path))
    ctx.build_index(seqs, cfg["k"])
    L, seed, R = cfg["read_len"], 4545, 2_000_000
    sub, nth = int(0.005 * 2 ** 32), int(0.001 * 2 ** 32)
    monkeypatch.setenv("VS_ACC_ROWS", "0")
    old = _count(host, ctx, st, cum, seed, L, [(0, R)], sub, nth)
    assert not ctx.last_launched & ctx.RAN_ROW_OWNERS
    for env in ({}, {"VS_ROWS_PER_STRIP": "1"}, {"VS_ROWS_PER_STRIP": "64", "VS_ACC_FILL": "1"}, {"VS_NO_SORT": "1"}, {"VS_LTAB_BITS": "0"},
                {"VS_LTAB_BITS": "12"}, {"VS_ROWS_SUB": "300000", "VS_ROWS_KEYS": "20000"}):
        monkeypatch.setenv("VS_ACC_ROWS", "1")
        for k2, v in env.items():
            monkeypatch.setenv(k2, v)
        new = _count(host, ctx, st, cum, seed, L, [(0, R)], sub, nth)
        assert ctx.last_launched & ctx.RAN_ROW_OWNERS
        assert torch.equal(old.mats, new.mats) and torch.equal(old.stats, new.stats), env
        if new.tile_map is not None:  # every non-zero cell lies in a marked tile
            n = new.n
            T = (n + 63) // 64
            tm = new.tile_map.view(2, T, T).bool()
            for mat in (0, 1):
                for lo in range(0, n, 4096):
                    blk = new.mats[mat, lo:lo + 4096] != 0
                    pad = torch.zeros((blk.shape[0] + 63) // 64 * 64, T * 64, dtype=torch.bool, device=blk.device)
                    pad[:blk.shape[0], :n] = blk
                    touched = pad.view(-1, 64, T, 64).any(dim=3).any(dim=1)
                    assert not (touched & ~tm[mat, lo // 64:lo // 64 + touched.shape[0]]).any(), (env, mat, lo)
        del new
        torch.cuda.empty_cache()
        for k2 in env:
            monkeypatch.delenv(k2)


def test_row_owner_counting_with_more_rows_per_chunk_than_lds_cursors(host, xctx, tmp_path, monkeypatch):
    """VS_ACC_ROWS=1 on configs[2]'s 5 039-node graph in INPUT order (VS_NO_SORT=1): a chunk of 16 384 pairs then holds
    nearly every node, more than the 4 096 rows k_rows_fill keeps an LDS cursor for -- the rest place their pairs through a
    global cursor per entry.  Against the C oracle."""
    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[2]
    ctx = xctx
    st, pre, names, seqs, cum, logger, n_in = workload_for(2, str(tmp_path))
    ctx.build_index(seqs, cfg["k"])
    L, seed, M = cfg["read_len"], 4646, 60_000
    sub, nth = int(0.005 * 2 ** 32), int(0.001 * 2 ** 32)
    monkeypatch.setenv("VS_ACC_ROWS", "1")
    monkeypatch.setenv("VS_NO_SORT", "1")
    got = _count(host, ctx, st, cum, seed, L, [(0, M)], sub, nth)
    assert ctx.last_launched & ctx.RAN_ROW_OWNERS
    orc = pe_oracle_c.Oracle(seqs, cfg["k"])
    _assert_equals_oracle(got, orc, st, cum, seed, L, M, sub, nth)


def test_config3_10k_nodes_k127_reads_of_250(host, ctx, tmp_path):  # (also: k = 127 reads of 250 bases select the long-window shape)
    import torch

    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[3]
    st, pre, names, seqs, cum, logger, n_in = workload_for(3, str(tmp_path))
    assert len(seqs) >= 10000, len(seqs)
    ctx.build_index(seqs, cfg["k"])
    L, seed = cfg["read_len"], 4343
    sub, nth = int(0.005 * 2 ** 32), int(0.001 * 2 ** 32)
    R = 1_200_000
    whole = _count(host, ctx, st, cum, seed, L, [(0, R)], sub, nth)
    # straight-line comparison with the long windows, and its COMPILE-TIME shape (16 words, two probes, 60 ends per tile):
    # the host's LDS loop has to land on exactly 60 ends for it -- a layout change must not drop the shape silently
    assert ctx.last_kernel.startswith("k_pe_tiles<2, 16u, 2u"), ctx.last_kernel
    parts = _count(host, ctx, st, cum, seed, L, [(0, 300_001), (300_001, 7), (300_008, R - 300_008)], sub, nth)
    assert torch.equal(whole.mats, parts.mats) and torch.equal(whole.stats, parts.stats)
    stats = whole.stats.cpu().tolist()
    assert sum(stats) == R and stats[2] > 0.99 * R
    del parts, whole
    torch.cuda.empty_cache()
    M = 200_000
    prefix = _count(host, ctx, st, cum, seed, L, [(0, M)], sub, nth)
    orc = pe_oracle_c.Oracle(seqs, cfg["k"])
    _assert_equals_oracle(prefix, orc, st, cum, seed, L, M, sub, nth)


@pytest.mark.parametrize("config", [0, 1])
def test_config_all_pairs_equal_the_oracle(host, ctx, tmp_path, config):
    """configs[0] (6-strain HIV-like, ~200 nodes, 100 k pairs) and configs[1] (5-strain HCV-like, ~1 k nodes, 1 M
    pairs): EVERY pair of the config's bench stream counted on the device, every cell compared with the C
    oracle's (the port does 2e5 pairs/s: seconds), plus the partition property."""
    import torch

    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[config]
    st, pre, names, seqs, cum, logger, n_in = workload_for(config, str(tmp_path))
    assert (150 <= len(seqs) <= 300) if config == 0 else (800 <= len(seqs) <= 1200), len(seqs)
    ctx.build_index(seqs, cfg["k"])
    L, seed, R = cfg["read_len"], 20250000 + config, cfg["total_pairs"]
    sub, nth = int(0.005 * 2 ** 32), int(0.001 * 2 ** 32)
    whole = _count(host, ctx, st, cum, seed, L, [(0, R)], sub, nth)
    parts = _count(host, ctx, st, cum, seed, L, [(0, R // 3), (R // 3, 1), (R // 3 + 1, R - R // 3 - 1)], sub, nth)
    assert torch.equal(whole.mats, parts.mats) and torch.equal(whole.stats, parts.stats)
    orc = pe_oracle_c.Oracle(seqs, cfg["k"])
    _assert_equals_oracle(whole, orc, st, cum, seed, L, R, sub, nth)


@pytest.mark.parametrize("config", [0, 1, 2, 3])
def test_hip_path_reproduces_the_files_of_the_real_reference_script(host, ctx, tmp_path, config):
    """The expected value here did not come out of the port: tools/time_reference.py ran the REAL
    utils/VStrains_PE_Inference.py (its text dump is PE_Inference.py:190-207) in the build container on the bench
    stream of configs[0] (all 100 k pairs), on the 200 k-pair prefixes of configs[1] / configs[2] and (round 5) on the
    20 k-pair prefix of configs[3] (k = 127, 2 x 250 bases, 10 084 nodes: two files of 1.2 GB), and committed
    the SHA-256 of its pe_info / st_info (and of the s_graph_L1.gfa it read).  Here the same pairs are regenerated
    with vs_synth_pairs, counted by the HIP path, written with vs_write_matrix_text, and hashed."""
    import hashlib
    import json
    import os

    from conftest import GOLDEN
    from vstrains_amd.workloads import CONFIGS, workload_for

    with open(os.path.join(GOLDEN, "reference_digests.json")) as fh:
        want = json.load(fh)["configs[%d]" % config]
    cfg = CONFIGS[config]
    st, pre, names, seqs, cum, logger, n_in = workload_for(config, str(tmp_path))

    def sha(path):
        h = hashlib.sha256()
        with open(path, "rb") as fh:
            for chunk in iter(lambda: fh.read(1 << 24), b""):
                h.update(chunk)
        return h.hexdigest()

    assert len(seqs) == want["nodes"] and cfg["k"] == want["k"] and cfg["read_len"] == want["read_len"]
    assert sha(os.path.join(str(tmp_path), "gfa", "s_graph_L1.gfa")) == want["s_graph_L1_gfa_sha256"]  # the same graph, byte for byte
    ids, gseqs = host.read_gfa_segments(os.path.join(str(tmp_path), "gfa", "s_graph_L1.gfa"))
    assert gseqs == list(seqs)
    ctx.build_index(gseqs, cfg["k"])
    counter = _count(host, ctx, st, cum, want["stream_seed"], cfg["read_len"], [(0, want["pairs"])], want["sub_thresh"], want["n_thresh"])
    node_mat, short_mat, stats = counter.result()
    assert sum(stats) == want["pairs"]
    for name, mat in (("pe_info", node_mat), ("st_info", short_mat)):
        path = os.path.join(str(tmp_path), name)
        host.write_matrix_text(path, ids, mat)
        assert os.path.getsize(path) == want[name + "_bytes"]
        assert sha(path) == want[name + "_sha256"], name


def test_cli_reproduces_the_real_reference_command_at_configs0(tmp_path):
    """BASELINE configs[0] through the whole `vstrains`-compatible command on the device -- FASTQ ingest, PE-link inference,
    the native stage handle, the final files -- against what the REAL reference command wrote on the same inputs
    (tools/time_reference.py, build container: /root/reference/vstrains behind the graph-tool stand-in under both in-edge
    models and hash seeds 0-3).  The 29 files all eight reference runs agree on, strain.paths / strain.fasta /
    split_graph_final.gfa among them, must come out the same."""
    from graph_case import reference_command_inputs, reference_command_problems
    from vstrains_amd import cli

    inp, want = reference_command_inputs(str(tmp_path / "work"))
    out = str(tmp_path / "out")
    cli.main(["-a", "spades", "-g", inp["gfa"], "-p", inp["paths"], "-o", out, "-fwd", inp["fwd"], "-rve", inp["rve"]])
    assert len(want["files_sha256"]) >= 25 and "strain.paths" in want["files_sha256"]
    problems = reference_command_problems(out, want)
    assert not problems, problems


@pytest.mark.parametrize("config", [1, 2])
def test_cli_reproduces_the_real_reference_command(config, tmp_path):
    """(r6) The same at the graph sizes BASELINE's metric is quoted on: configs[1] (853 nodes, all 1 M pairs of the bench
    stream) and configs[2] (5 039 nodes, the 200 k-pair prefix the real PE script's files are pinned on).  The REAL reference
    command ran on these inputs in the build container (tools/time_reference_stages.py: /root/reference/vstrains behind the
    stand-in, PE files from the real script, the stages timed as one interval -- profiles/r6/reference_stages_config<i>.json);
    every file all its runs agree on must come out of the device path byte for byte: the strain-extraction leg's
    real-reference pin above 216 nodes."""
    import json

    from conftest import GOLDEN
    from graph_case import reference_command_inputs, reference_command_problems
    from vstrains_amd import cli

    with open(os.path.join(GOLDEN, "reference_digests.json")) as fh:
        if "configs[%d]_whole_command" % config not in json.load(fh):
            pytest.skip("the real reference has not finished this config in the build container (profiles/r6/reference_stages_config%d.json)" % config)
    inp, want = reference_command_inputs(str(tmp_path / "work"), config)
    out = str(tmp_path / "out")
    cli.main(["-a", "spades", "-g", inp["gfa"], "-p", inp["paths"], "-o", out, "-fwd", inp["fwd"], "-rve", inp["rve"]])
    assert len(want["files_sha256"]) >= 25 and "strain.paths" in want["files_sha256"]
    problems = reference_command_problems(out, want)
    assert not problems, problems


def test_per_end_lists_equal_the_real_reference_function_at_config4_size(host, ctx, tmp_path):
    """configs[4] (54 465 nodes): the whole reference script cannot run at that size (two text files of 3e9 lines), but its
    ``single_end_read_mapping`` (PE_Inference.py:16-48) can, end by end -- tests/golden/make_pe_end_lists.py imported it from
    /root/reference, ran it on the first 1 024 pairs of the bench stream against the table of this graph and committed the
    lists it returned (mean 8.6 nodes per end, up to 19: ends beyond a list row's 16 are among them, so the overflow kernels
    are held to the reference too).  The same pairs are regenerated on the device and ``vs_pe_map_ends`` must give the same
    list for every end."""
    import hashlib
    import json
    import os

    from conftest import GOLDEN
    from vstrains_amd.workloads import CONFIGS, workload_for

    with open(os.path.join(GOLDEN, "pe_end_lists_config4.json")) as fh:
        want = json.load(fh)
    cfg = CONFIGS[4]
    st, pre, names, seqs, cum, logger, n_in = workload_for(4, str(tmp_path))
    with open(os.path.join(str(tmp_path), "gfa", "s_graph_L1.gfa"), "rb") as fh:
        assert hashlib.sha256(fh.read()).hexdigest() == want["s_graph_L1_gfa_sha256"]  # the same graph, byte for byte
    assert len(seqs) == want["nodes"] and cfg["k"] == want["k"] and cfg["read_len"] == want["read_len"]
    ctx.build_index(list(seqs), cfg["k"])
    block = ctx.synth_pairs(st.genomes, cum, want["stream_seed"], 0, want["pairs"], cfg["read_len"], want["sub_thresh"], want["n_thresh"])
    got = ctx.map_ends(block, cap=64)
    assert len(got) == len(want["lists"]) == 2 * want["pairs"]
    n_long = 0
    for e, (g, w) in enumerate(zip(got, want["lists"])):
        assert g == (sorted(w) if w is not None else []), (e, g, w)  # (None: the pair loop drops the pair, :160-165)
        n_long += w is not None and len(w) > 16
    assert n_long > 0  # ends past the tile kernel's list rows were part of the comparison
