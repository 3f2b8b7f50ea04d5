"""Parity of the HIP path (through the C ABI) with the oracle and with the reference's own
outputs.  Integer work: the bar is bit-exact."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, pe_cases
from oracle import pe_oracle, pe_oracle_c

pytestmark = pytest.mark.gpu


def _read(path):
    with open(path, "r", newline="") as fh:
        return fh.read()


@pytest.fixture(scope="module")
def host():
    from vstrains_amd import pe as host

    return host


@pytest.fixture(scope="module")
def ctx(host):
    c = host.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def xctx(host):
    """A context in experiment mode: the tuning switches of vs_pe_count are live on it (conftest.experiment_context)."""
    from conftest import experiment_context

    c = experiment_context(host)
    yield c
    c.close()


def _gpu_matrices(host, ctx, seqs, fwd, rve, k):
    ctx.build_index(seqs, k)
    counter = host.PeCounter(ctx)
    block = ctx.pack_pairs(fwd, rve)
    counter.add(block)
    res = counter.result()
    return res, block


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_golden_files_bit_exact(host, ctx, name, d, meta):
    ids, seqs = host.read_gfa_segments(os.path.join(d, "graph.gfa"))
    fq = host.FastqPair(os.path.join(d, "fwd.fq"), os.path.join(d, "rve.fq"), ctx)  # native ingest
    ctx.build_index(seqs, meta["k"])
    counter = host.PeCounter(ctx)
    block = fq.block(0, len(fq))
    counter.add(block)
    node_mat, short_mat, stats = counter.result()
    assert pe_oracle.matrix_text(ids, node_mat) == _read(os.path.join(d, "pe_info"))
    assert pe_oracle.matrix_text(ids, short_mat) == _read(os.path.join(d, "st_info"))
    f = pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq"))
    r = pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))
    assert stats == pe_oracle.pe_matrices(seqs, f, r, meta["k"])[2]


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_golden_files_bit_exact_by_row_owners(host, xctx, name, d, meta, monkeypatch):
    """The same files of the real reference script with the counters summed by row owners (VS_ACC_ROWS=1: what graphs beyond
    46 340 nodes take by themselves), one-row strips and a list table of eight slots."""
    monkeypatch.setenv("VS_ACC_ROWS", "1")
    monkeypatch.setenv("VS_ROWS_PER_STRIP", "1")
    monkeypatch.setenv("VS_LTAB_BITS", "3")
    ctx = xctx
    ids, seqs = host.read_gfa_segments(os.path.join(d, "graph.gfa"))
    fq = host.FastqPair(os.path.join(d, "fwd.fq"), os.path.join(d, "rve.fq"), ctx)
    ctx.build_index(seqs, meta["k"])
    counter = host.PeCounter(ctx)
    counter.add(fq.block(0, len(fq)))
    node_mat, short_mat, stats = counter.result()
    if len(fq) and len(seqs):
        assert ctx.last_launched & ctx.RAN_ROW_OWNERS
    assert pe_oracle.matrix_text(ids, node_mat) == _read(os.path.join(d, "pe_info"))
    assert pe_oracle.matrix_text(ids, short_mat) == _read(os.path.join(d, "st_info"))


@pytest.mark.parametrize("name,d,meta", pe_cases(), ids=[c[0] for c in pe_cases()])
def test_per_end_lists_match_oracle(host, ctx, name, d, meta):
    K = meta["k"] + 1
    ids, seqs = host.read_gfa_segments(os.path.join(d, "graph.gfa"))
    f = pe_oracle.fastq_sequences(os.path.join(d, "fwd.fq"))
    r = pe_oracle.fastq_sequences(os.path.join(d, "rve.fq"))
    n = min(len(f), len(r))
    if n == 0:
        pytest.skip("no pairs")
    ctx.build_index(seqs, meta["k"])
    block = ctx.pack_pairs(f[:n], r[:n])
    lists = ctx.map_ends(block, cap=max(len(seqs), 1))
    tab = pe_oracle.build_table(seqs, K)
    lens = [len(s) for s in seqs]
    for p in range(n):
        used = not (f[p].count("N") or r[p].count("N")) and len(f[p]) >= K and len(r[p]) >= K
        want_f = pe_oracle.map_read_end(f[p], tab, lens, K) if used else []
        want_r = pe_oracle.map_read_end(r[p], tab, lens, K) if used else []
        assert lists[2 * p] == want_f, (p, f[p])
        assert lists[2 * p + 1] == want_r, (p, r[p])


def test_lowercase_node_raises_keyerror(host, ctx):
    (case,) = [c for c in pe_cases(ok_only=False) if c[2]["returncode"] != 0]
    name, d, meta = case
    ids, seqs = host.read_gfa_segments(os.path.join(d, "graph.gfa"))
    with pytest.raises(KeyError) as ei:
        ctx.build_index(seqs, meta["k"])
    assert "KeyError: %r" % ei.value.args[0] == meta["stderr_last"]


def test_a_node_of_more_than_two_to_the_23_bases_takes_the_generic_kernel(host, ctx):
    """(r6) The straight-line kernels pack a node's length above the read offset in 24 bits of one table word; a graph with a
    node of 2^23 bases or more must not take them (the guard was missing for MODE 1 until round 6).  One node of 2^23 + 500
    bases and two ordinary ones, 400 read pairs of 2 x 150 off the long node's far end: the generic kernel runs and the
    counters equal the C oracle's."""
    rng = np.random.default_rng(77)
    big = "".join("ACGT"[i] for i in rng.integers(0, 4, size=(1 << 23) + 500))
    small = ["".join("ACGT"[i] for i in rng.integers(0, 4, size=300)) for _ in range(2)]
    seqs = [small[0], big, small[1]]
    comp = str.maketrans("ACGT", "TGCA")
    fwd, rve = [], []
    for _ in range(400):
        a = int(rng.integers(len(big) - 3000, len(big) - 700))
        ins = int(rng.integers(300, 600))
        frag = big[a:a + ins]
        f, r = frag[:150], frag[-150:][::-1].translate(comp)
        if rng.random() < 0.5:
            f, r = r, f
        fwd.append(f)
        rve.append(r)
    ctx.build_index(seqs, 55)
    counter = host.PeCounter(ctx)
    counter.add(ctx.pack_pairs(fwd, rve))
    assert ctx.last_kernel.startswith("k_pe_tiles<0"), ctx.last_kernel
    node_mat, short_mat, stats = counter.result()
    ref = pe_oracle_c.Oracle(seqs, 55).count_pairs(fwd, rve)
    assert np.array_equal(node_mat, ref[0]) and np.array_equal(short_mat, ref[1])
    assert stats == tuple(int(x) for x in ref[2]) and int(node_mat[1, 1]) > 300


def test_cli_drop_in_writes_identical_files(tmp_path):
    name, d, meta = [c for c in pe_cases() if c[0] == "errors_k21"][0]
    out = tmp_path / "aln"
    proc = subprocess.run(
        [sys.executable, "-m", "vstrains_amd.pe_inference", "-g", os.path.join(d, "graph.gfa"), "-o", str(out) + "/",
         "-f", os.path.join(d, "fwd.fq"), "-r", os.path.join(d, "rve.fq"), "-k", str(meta["k"])],
        cwd=ROOT, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    assert _read(out / "pe_info") == _read(os.path.join(d, "pe_info"))
    assert _read(out / "st_info") == _read(os.path.join(d, "st_info"))
    lines = proc.stdout.splitlines()
    assert lines[0] == "----------------------Paired-End Information Alignment----------------------"
    assert lines[1] == "Start aligning reads to gfa nodes"
    assert [l for l in lines if l.startswith("Number of processed reads")] == meta["progress_lines"]
    assert lines[-2].startswith("Global time elapsed:  ")
    assert lines[-1] == "result stored in:  %s/pe_info" % out


def test_cli_drop_in_reads_gzip_fastq(tmp_path):
    """gzip input (beyond the reference, which opens text only): the files a plain-text run writes."""
    import gzip

    name, d, meta = [c for c in pe_cases() if c[0] == "errors_k21"][0]
    for which in ("fwd", "rve"):
        with open(os.path.join(d, which + ".fq"), "rb") as fh:
            raw = fh.read()
        with open(tmp_path / (which + ".fq.gz"), "wb") as fh:  # (two members, as bgzip and `cat a.gz b.gz` make them)
            fh.write(gzip.compress(raw[: len(raw) // 3]))
            fh.write(gzip.compress(raw[len(raw) // 3:]))
    out = tmp_path / "aln"
    proc = subprocess.run(
        [sys.executable, "-m", "vstrains_amd.pe_inference", "-g", os.path.join(d, "graph.gfa"), "-o", str(out) + "/",
         "-f", str(tmp_path / "fwd.fq.gz"), "-r", str(tmp_path / "rve.fq.gz"), "-k", str(meta["k"])],
        cwd=ROOT, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    assert _read(out / "pe_info") == _read(os.path.join(d, "pe_info"))
    assert _read(out / "st_info") == _read(os.path.join(d, "st_info"))


def test_cli_drop_in_fails_like_reference_on_bad_node(tmp_path):
    (case,) = [c for c in pe_cases(ok_only=False) if c[2]["returncode"] != 0]
    name, d, meta = case
    proc = subprocess.run(
        [sys.executable, "-m", "vstrains_amd.pe_inference", "-g", os.path.join(d, "graph.gfa"), "-o", str(tmp_path / "aln"),
         "-f", os.path.join(d, "fwd.fq"), "-r", os.path.join(d, "rve.fq"), "-k", str(meta["k"])],
        cwd=ROOT, capture_output=True, text=True)
    assert proc.returncode != 0
    assert proc.stderr.strip().splitlines()[-1] == meta["stderr_last"]


def _dense_case(k, n_pairs, read_len, seed, snp, n_strains=6, glen=1500, sub=0.01, nrate=0.02):
    from vstrains_amd import synth

    st = synth.make_strains(n_strains, glen, snp, seed=seed)
    g = synth.compact_dbg(st, k)
    f, r = synth.sample_pairs(st, n_pairs, read_len, seed=seed + 1, sub_rate=sub, n_rate=nrate)
    return g, f, r


@pytest.mark.parametrize("no_mid", ["0", "1"])
def test_overflow_pairs_take_slow_path_and_stay_exact(host, xctx, no_mid, monkeypatch):
    ctx = xctx  # (the switches below exist only on a context made in experiment mode)
    # dense variation at k=11: a 100-base read is accepted by more than 16 short nodes in ~30 % of
    # the ends, which overflows the per-end list kept in LDS: those pairs go to k_pe_mid (one wavefront per
    # pair, state in LDS), or with VS_NO_MID=1 straight to the general kernel k_pe_slow
    monkeypatch.setenv("VS_NO_MID", no_mid)
    g, f, r = _dense_case(11, 1500, 100, seed=301, snp=0.2)
    (node_mat, short_mat, stats), block = _gpu_matrices(host, ctx, g.seqs, f, r, 11)
    t = ctx.last_timing()
    assert t["slow_pairs"] > 0, "case does not exercise the overflow path"
    orc = pe_oracle_c.Oracle(g.seqs, 11)
    ref_node, ref_short, ref_stats = orc.count_pairs(f, r)
    assert np.array_equal(node_mat, ref_node)
    assert np.array_equal(short_mat, ref_short)
    assert stats == tuple(int(x) for x in ref_stats)
    lists = ctx.map_ends(block, cap=len(g.seqs))
    for p in range(0, len(f), 37):
        if "N" in f[p] or "N" in r[p]:
            continue
        assert lists[2 * p] == orc.map_end(f[p])
        assert lists[2 * p + 1] == orc.map_end(r[p])


@pytest.mark.parametrize("case", ["dense_k11", "chain_k20", "strains_k55"])
def test_tracked_counting_marks_every_touched_tile_and_zeroes_only_those(host, ctx, case):
    """vs_pe_count_tracked / vs_counts_zero_tracked (PeCounter(track_tiles=True)): after a block every non-zero cell of both
    matrices lies in a marked 64 x 64 tile -- the main path marks from the list rows, the overflow kernels (k_pe_mid,
    k_pe_slow: the first two cases) where they add --, a reset leaves all-zero counters and an empty map, and a second
    block counted after it equals the oracle: nothing of the first one is left."""
    import torch

    from vstrains_amd import synth

    if case == "dense_k11":
        g, f, r = _dense_case(11, 1500, 100, seed=301, snp=0.2)
        seqs, k = g.seqs, 11
    elif case == "chain_k20":
        rng = np.random.default_rng(17)
        k = 20
        genome = "".join("ACGT"[i] for i in rng.integers(0, 4, size=700))
        seqs = [genome[i: i + k + 1] for i in range(len(genome) - k)]
        f, r = [], []
        for i in range(400):
            L = 120 if i % 3 else 50
            a = int(rng.integers(0, len(genome) - 300))
            f.append(genome[a: a + L])
            r.append(synth.revcomp(genome[a + 150: a + 150 + L]))
    else:
        st = synth.make_strains(6, 5000, 0.05, seed=91)
        g = synth.compact_dbg(st, 55)
        f, r = synth.sample_pairs(st, 20000, 150, seed=92, sub_rate=0.01, n_rate=0.01)
        seqs, k = g.seqs, 55
    n = len(seqs)
    assert n > 128  # (several tiles per side)
    ctx.build_index(seqs, k)
    orc = pe_oracle_c.Oracle(seqs, k)
    half = len(f) // 2
    counter = host.PeCounter(ctx, track_tiles=True)
    counter.add(ctx.pack_pairs(f[:half], r[:half]))
    torch.cuda.synchronize()
    if case != "strains_k55":
        assert ctx.last_timing()["slow_pairs"] > 0, "case does not exercise the overflow kernels"
    T = (n + 63) // 64
    marked = counter.tile_map.view(2, T, T).bool()
    nz = counter.mats[:, :n, :n] != 0
    tiles_nz = torch.zeros((2, T, T), dtype=torch.bool, device=nz.device)
    idx = nz.nonzero()
    tiles_nz[idx[:, 0], idx[:, 1] // 64, idx[:, 2] // 64] = True
    assert bool((tiles_nz & ~marked).sum() == 0), "a non-zero cell lies in an unmarked tile"
    assert int(marked.sum()) < 2 * T * T or n < 400  # (a banded matrix leaves tiles untouched)
    first = counter.result()
    want = orc.count_pairs(f[:half], r[:half])
    assert np.array_equal(first[0], want[0]) and np.array_equal(first[1], want[1])
    counter.reset()
    torch.cuda.synchronize()
    assert int(counter.mats.abs().sum()) == 0 and int(counter.tile_map.sum()) == 0
    counter.add(ctx.pack_pairs(f[half:], r[half:]))
    second = counter.result()
    want = orc.count_pairs(f[half:], r[half:])
    assert np.array_equal(second[0], want[0]) and np.array_equal(second[1], want[1])
    assert second[2] == tuple(int(x) for x in want[2])


@pytest.mark.parametrize("case", ["dense_k11", "chain_k20"])
def test_row_owners_next_to_the_overflow_kernels(host, xctx, case, monkeypatch):
    """Row owners (VS_ACC_ROWS=1) on blocks where a third of the ends are accepted by more than 16 nodes: those pairs have
    empty list rows and are counted by k_pe_mid / k_pe_slow, the rest by the row owners, into the same counters.  Against the
    C oracle."""
    from vstrains_amd import synth

    ctx = xctx
    if case == "dense_k11":
        g, f, r = _dense_case(11, 1500, 100, seed=301, snp=0.2)
        seqs, k = g.seqs, 11
    else:
        rng = np.random.default_rng(17)
        k = 20
        genome = "".join("ACGT"[i] for i in rng.integers(0, 4, size=700))
        seqs = [genome[i: i + k + 1] for i in range(len(genome) - k)]
        f, r = [], []
        for i in range(900):
            L = (120, 50, 62, 37, 30)[i % 5]  # 100 nodes (beyond k_pe_mid), 30, 42, 17 and 10 (a list row holds 16)
            a = int(rng.integers(0, len(genome) - 300))
            f.append(genome[a: a + L])
            r.append(synth.revcomp(genome[a + 150: a + 150 + (L if i % 3 else 25)]))
    monkeypatch.setenv("VS_ACC_ROWS", "1")
    want = pe_oracle_c.Oracle(seqs, k).count_pairs(f, r)
    (node_mat, short_mat, stats), block = _gpu_matrices(host, ctx, seqs, f, r, k)
    assert ctx.last_launched & ctx.RAN_ROW_OWNERS and ctx.last_launched & ctx.RAN_PE_MID
    assert ctx.last_timing()["slow_pairs"] > 100
    assert np.array_equal(node_mat, want[0]) and np.array_equal(short_mat, want[1])
    assert stats == tuple(int(x) for x in want[2])


def test_ends_with_a_hundred_nodes_pass_through_both_overflow_kernels(host, ctx):
    """A chain of nodes that advance ONE base each (every node k + 1 bases long): a 120-base read is accepted by 100 nodes
    -- beyond a list row (16) and beyond k_pe_mid's per-end list (64), so the pairs end in the general kernel; shorter reads
    of the same block (50 bases: 30 nodes) stop at k_pe_mid; reads with many dirty bytes take the mask path in either."""
    rng = np.random.default_rng(17)
    k = 20
    genome = "".join("ACGT"[i] for i in rng.integers(0, 4, size=700))
    seqs = [genome[i: i + k + 1] for i in range(len(genome) - k)]
    from vstrains_amd import synth

    fwd, rve = [], []
    for i in range(600):
        L = 120 if i % 3 else 50
        a = int(rng.integers(0, len(genome) - 300))
        f_, r_ = genome[a: a + L], synth.revcomp(genome[a + 150: a + 150 + L])
        if i % 7 == 0:  # six bytes outside ACGT: more than inv4 holds
            for p_ in rng.choice(L, size=6, replace=False):
                f_ = f_[: int(p_)] + "n" + f_[int(p_) + 1:]
        fwd.append(f_)
        rve.append(r_)
    orc = pe_oracle_c.Oracle(seqs, k)
    want = orc.count_pairs(fwd, rve)
    (node_mat, short_mat, stats), block = _gpu_matrices(host, ctx, seqs, fwd, rve, k)
    assert ctx.last_timing()["slow_pairs"] > 500
    assert np.array_equal(node_mat, want[0]) and np.array_equal(short_mat, want[1])
    assert stats == tuple(int(x) for x in want[2])
    lists = ctx.map_ends(block, cap=len(seqs))
    for p in range(0, len(fwd), 11):
        assert lists[2 * p] == orc.map_end(fwd[p]) and lists[2 * p + 1] == orc.map_end(rve[p])
    assert max(len(l) for l in lists) >= 100


@pytest.mark.parametrize("k,read_len,n_pairs,snp", [(55, 150, 20000, 0.03), (127, 250, 6000, 0.02), (31, 100, 8000, 0.05), (32, 100, 8000, 0.05)])
def test_random_graphs_against_c_oracle(host, ctx, k, read_len, n_pairs, snp):
    g, f, r = _dense_case(k, n_pairs, read_len, seed=400 + k, snp=snp, glen=4000)
    (node_mat, short_mat, stats), _ = _gpu_matrices(host, ctx, g.seqs, f, r, k)
    orc = pe_oracle_c.Oracle(g.seqs, k)
    ref_node, ref_short, ref_stats = orc.count_pairs(f, r)
    assert np.array_equal(node_mat, ref_node)
    assert np.array_equal(short_mat, ref_short)
    assert stats == tuple(int(x) for x in ref_stats)
    assert int(ref_node.sum()) > 0


@pytest.mark.parametrize("read_len", [100, 101, 125, 126, 150, 151, 159, 160])
def test_read_lengths_around_the_compile_time_shapes(host, xctx, read_len, monkeypatch):
    """k = 55 with the read lengths the shape-specialised instantiations of k_pe_tiles are picked
    for (2x100/101, 2x125/126, 2x150/151; 159 = the longest that takes one, 160 = generic layout),
    no N in the block so that the straight-line kernels apply; each also with VS_NO_STD=1."""
    ctx = xctx  # (the switches below exist only on a context made in experiment mode)
    g, f, r = _dense_case(55, 9000, read_len, seed=900 + read_len, snp=0.03, glen=5000, nrate=0.0)
    orc = pe_oracle_c.Oracle(g.seqs, 55)
    ref_node, ref_short, ref_stats = orc.count_pairs(f, r)
    for no_std in ("0", "1"):
        monkeypatch.setenv("VS_NO_STD", no_std)
        (node_mat, short_mat, stats), _ = _gpu_matrices(host, ctx, g.seqs, f, r, 55)
        assert np.array_equal(node_mat, ref_node) and np.array_equal(short_mat, ref_short)
        assert stats == tuple(int(x) for x in ref_stats)
    assert int(ref_node.sum()) > 0


def test_variable_lengths_and_empty_block(host, ctx):
    g, f, r = _dense_case(21, 3000, 90, seed=501, snp=0.03)
    rng = np.random.default_rng(5)
    f = [s[: int(rng.integers(0, 91))] for s in f]
    r = [s[: int(rng.integers(15, 91))] for s in r]
    (node_mat, short_mat, stats), _ = _gpu_matrices(host, ctx, g.seqs, f, r, 21)
    orc = pe_oracle_c.Oracle(g.seqs, 21)
    ref_node, ref_short, ref_stats = orc.count_pairs(f, r)
    assert np.array_equal(node_mat, ref_node) and np.array_equal(short_mat, ref_short)
    assert stats == tuple(int(x) for x in ref_stats)
    # an empty block adds nothing
    counter = host.PeCounter(ctx)
    counter.add(ctx.pack_pairs([], []))
    a, b, s = counter.result()
    assert a.sum() == 0 and b.sum() == 0 and s == (0, 0, 0)


@pytest.mark.parametrize("dirty", [False, True], ids=["straight_line_kernel", "generic_kernel_masked_bytes"])
def test_ragged_block_through_sort_and_double_buffered_tiles(host, ctx, dirty):
    """Many tiles of reads of every length 0..191 (sorted by locus, next tile's words arriving by
    LDS-direct loads while a tile is worked on); with ``dirty`` some reads hold N / other bytes."""
    g, f, r = _dense_case(55, 30000, 191, seed=777, snp=0.03, glen=6000, nrate=0.0)  # no N: the block qualifies for k_pe_tiles<true>
    rng = np.random.default_rng(11)
    f = [s[: int(rng.integers(0, 192))] for s in f]
    r = [s[: int(rng.integers(40, 192))] for s in r]
    if dirty:
        for lst in (f, r):
            for i in rng.choice(len(lst), size=600, replace=False):
                s_ = lst[int(i)]
                if len(s_) > 3:
                    p_ = int(rng.integers(0, len(s_)))
                    lst[int(i)] = s_[:p_] + ("N" if rng.random() < 0.3 else rng.choice(list("nRYacgt*"))) + s_[p_ + 1:]
    (node_mat, short_mat, stats), _ = _gpu_matrices(host, ctx, g.seqs, f, r, 55)
    orc = pe_oracle_c.Oracle(g.seqs, 55)
    ref_node, ref_short, ref_stats = orc.count_pairs(f, r)
    assert np.array_equal(node_mat, ref_node) and np.array_equal(short_mat, ref_short)
    assert stats == tuple(int(x) for x in ref_stats)
    assert int(ref_node.sum()) > 0


@pytest.mark.parametrize("k,max_len", [(127, 256), (99, 250), (95, 256), (94, 287), (86, 200), (127, 300), (127, 317), (127, 318), (140, 330)])
def test_long_stride_kernel_with_ragged_dirty_reads(host, xctx, k, max_len, monkeypatch):
    """k > 85 (probe stride > 32) / reads beyond 191 bases take the long-window straight-line kernel
    (k_pe_tiles<2>): reads of every length up to its limit, some with N or other bytes outside
    ACGT; the generic kernel (VS_NO_FAST=1) must agree with it and with the oracle.  k >= 95 (k + 1 >= 96):
    63-base seeds with mixed keys -- the comparison then starts at the seed's first base, so the limit is 256
    bases instead of 31 + 256 (k = 94 / 95: either side of that switch)."""
    ctx = xctx  # (the switches below exist only on a context made in experiment mode)
    g, f, r = _dense_case(k, 9000, max_len, seed=1200 + k, snp=0.02, glen=6000, nrate=0.0)
    rng = np.random.default_rng(k)
    f = [s[: int(rng.integers(0, max_len + 1))] for s in f]
    r = [s[: int(rng.integers(k - 5, max_len + 1))] for s in r]
    for lst in (f, r):
        for i in rng.choice(len(lst), size=500, replace=False):
            s_ = lst[int(i)]
            for _ in range(int(rng.integers(1, 7))):
                if len(s_) > 3:
                    p_ = int(rng.integers(0, len(s_)))
                    s_ = s_[:p_] + ("N" if rng.random() < 0.1 else str(rng.choice(list("nRYacgt*")))) + s_[p_ + 1:]
            lst[int(i)] = s_
    orc = pe_oracle_c.Oracle(g.seqs, k)
    ref_node, ref_short, ref_stats = orc.count_pairs(f, r)
    assert int(ref_node.sum()) > 0
    for no_fast in ("0", "1"):
        monkeypatch.setenv("VS_NO_FAST", no_fast)
        (node_mat, short_mat, stats), _ = _gpu_matrices(host, ctx, g.seqs, f, r, k)
        # (reads beyond the long-window kernel's reach take the generic loops either way: its eight right windows compare 256
        # bases from where the comparison starts -- behind a verified 31-base seed, at the first base of a 63-base one -- and
        # what has to fit is the read's part behind its first probe, vs_seed_phase: up to 317 bases at k = 127)
        K, w = k + 1, (63 if k >= 95 else 31)
        s_ = K - w + 1
        in_reach = all(n - ((n - w) % s_ + s_) // 2 - (w if w <= 31 else 0) <= 256 for n in range(K, max_len + 1))
        assert ctx.last_kernel.startswith("k_pe_tiles<2" if no_fast == "0" and in_reach else "k_pe_tiles<0")
        assert np.array_equal(node_mat, ref_node) and np.array_equal(short_mat, ref_short)
        assert stats == tuple(int(x) for x in ref_stats)


def test_blocks_add_up_and_swapping_ends_transposes(host, ctx):
    g, f, r = _dense_case(55, 12000, 150, seed=601, snp=0.03, glen=5000)
    ctx.build_index(g.seqs, 55)
    whole = host.PeCounter(ctx)
    whole.add(ctx.pack_pairs(f, r))
    parts = host.PeCounter(ctx)
    for lo in range(0, len(f), 5000):
        parts.add(ctx.pack_pairs(f[lo:lo + 5000], r[lo:lo + 5000]))
    a, b = whole.result(), parts.result()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    swapped = host.PeCounter(ctx)
    swapped.add(ctx.pack_pairs(r, f))
    c = swapped.result()
    assert np.array_equal(c[0], a[0].T) and np.array_equal(c[1], a[1])


def test_results_do_not_depend_on_the_internal_numbering(host, ctx):
    """The index numbers the nodes along the graph's paths (Context.build_index, csrc/vs_order_host.cpp) and every
    result is handed out in the caller's numbering: matrices, per-end lists and the PE-link table of a graph whose
    nodes were shuffled and partly reverse-complemented equal the oracle's, with the renumbering on and off."""
    import random

    from vstrains_amd.graph.hip_ops import HipPeLinks
    from vstrains_amd.node_order import _revcomp

    g, f, r = _dense_case(55, 9000, 150, seed=733, snp=0.04, glen=5000)
    rnd = random.Random(5)
    seqs = [s if rnd.random() < 0.6 else _revcomp(s) for s in g.seqs]
    rnd.shuffle(seqs)
    names = ["n%d" % i for i in range(len(seqs))]
    orc = pe_oracle_c.Oracle(seqs, 55)
    ref_node, ref_short, ref_stats = orc.count_pairs(f, r)
    want_links = ref_node + ref_short
    want_links = want_links + want_links.T - np.diag(np.diag(want_links))  # (IO.py:598-627: both orders, the diagonal once)
    got = {}
    for renumber in (True, False):
        ctx.build_index(seqs, 55, renumber=renumber)
        assert (ctx.node_order is not None) == renumber
        counter = host.PeCounter(ctx)
        block = ctx.pack_pairs(f, r)
        counter.add(block)
        node_mat, short_mat, stats = counter.result()
        assert np.array_equal(node_mat, ref_node) and np.array_equal(short_mat, ref_short)
        assert stats == tuple(int(x) for x in ref_stats)
        lists = ctx.map_ends(block)
        table = HipPeLinks.from_counter(ctx, counter, names)
        assert np.array_equal(table.to_numpy(), want_links)
        u, v = 3, len(seqs) // 2
        assert table.block_sums([([table.index_of(names[u])], [table.index_of(names[v])])])[0] == int(want_links[u, v])
        table.close()
        got[renumber] = lists
    assert got[True] == got[False]
    ctx.build_index(seqs, 55)  # default: on
    assert ctx.node_order is not None and sorted(ctx.node_order.tolist()) == list(range(len(seqs)))


@pytest.mark.parametrize("k,read_len", [(55, 150), (127, 250), (21, 90)])
def test_repeated_read_ends_are_counted_every_time(host, ctx, k, read_len):
    """Deep coverage: most pairs of a locus repeat other pairs letter for letter (what the counter kernel's table and any
    shortcut for repeated ends must get right).  A block made of a few hundred distinct pairs repeated forty times in random
    order -- with copies that differ only in one end, only in lower-case / IUPAC bytes (same text once packed, other
    positions outside ACGT), in an N, in their length -- against the oracle, matrices and per-end lists."""
    import random

    g, f, r = _dense_case(k, 300, read_len, seed=900 + k, snp=0.05, n_strains=8, glen=2500, sub=0.004, nrate=0.01)
    rnd = random.Random(k)
    fw, rv = [], []
    for a, b in zip(f, r):
        for _ in range(40):
            fw.append(a)
            rv.append(b)
    for i in range(0, len(fw), 7):      # same forward read, another reverse read
        rv[i] = r[rnd.randrange(len(r))]
    for i in range(3, len(fw), 11):     # a byte outside ACGT: lower case here, IUPAC there, at this position or that
        x = list(fw[i])
        pos = rnd.randrange(len(x))
        x[pos] = x[pos].lower() if rnd.random() < 0.5 else "R"
        fw[i] = "".join(x)
    for i in range(5, len(fw), 13):     # the same two positions in several copies: equal also in what is outside ACGT
        x = list(rv[i])
        x[10] = x[10].lower()
        x[40] = "Y"
        rv[i] = "".join(x)
    for i in range(1, len(fw), 29):
        fw[i] = fw[i][:-3]              # shorter copy
    for i in range(2, len(fw), 97):
        rv[i] = rv[i][:20] + "N" + rv[i][21:]
    order = list(range(len(fw)))
    rnd.shuffle(order)
    fw = [fw[i] for i in order]
    rv = [rv[i] for i in order]
    (node_mat, short_mat, stats), block = _gpu_matrices(host, ctx, g.seqs, fw, rv, k)
    orc = pe_oracle_c.Oracle(g.seqs, k)
    ref_node, ref_short, ref_stats = orc.count_pairs(fw, rv)
    assert np.array_equal(node_mat, ref_node) and np.array_equal(short_mat, ref_short)
    assert stats == tuple(int(x) for x in ref_stats)
    lists = ctx.map_ends(block, cap=len(g.seqs))
    K = k + 1
    tab = pe_oracle.build_table(g.seqs, K)
    lens = [len(x) for x in g.seqs]
    for p in rnd.sample(range(len(fw)), 400):
        used = not (fw[p].count("N") or rv[p].count("N")) and len(fw[p]) >= K and len(rv[p]) >= K
        assert lists[2 * p] == (pe_oracle.map_read_end(fw[p], tab, lens, K) if used else []), p
        assert lists[2 * p + 1] == (pe_oracle.map_read_end(rv[p], tab, lens, K) if used else []), p


def test_device_read_generator_equals_cpu_twin(host, ctx):
    from vstrains_amd import synth

    st = synth.make_strains(5, 2000, 0.02, seed=71)
    ab = np.array(st.abundance)
    cum = np.minimum(np.floor(np.cumsum(ab) / ab.sum() * 2 ** 32), 2 ** 32 - 1).astype(np.uint32)
    cum[-1] = 0xFFFFFFFF
    sub = int(0.01 * 2 ** 32)
    nth = int(0.02 * 2 ** 32)
    block = ctx.synth_pairs(st.genomes, cum, seed=99, first_pair=1000, n_pairs=3000, read_len=150, sub_thresh=sub, n_thresh=nth)
    text, lens, flags = block.unpack()
    fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, 99, 1000, 3000, 150, sub, nth)
    assert (lens == 150).all()
    got = text.reshape(3000, 2, 150)
    has_n_f = (fw == ord("N")).any(axis=1)
    has_n_r = (rv == ord("N")).any(axis=1)
    assert np.array_equal((flags[0::2] & 1).astype(bool), has_n_f)
    assert np.array_equal((flags[1::2] & 1).astype(bool), has_n_r)
    ok_f = ~has_n_f
    ok_r = ~has_n_r
    assert np.array_equal(got[ok_f, 0, :], fw[ok_f])
    assert np.array_equal(got[ok_r, 1, :], rv[ok_r])
    assert has_n_f.sum() + has_n_r.sum() > 10
    # and the counts over that stream agree with the oracle run on the twin's text
    g = synth.compact_dbg(st, 55)
    ctx.build_index(g.seqs, 55)
    counter = host.PeCounter(ctx)
    counter.add(block)
    node_mat, short_mat, stats = counter.result()
    orc = pe_oracle_c.Oracle(g.seqs, 55)
    f = [bytes(x).decode() for x in fw]
    r = [bytes(x).decode() for x in rv]
    ref_node, ref_short, ref_stats = orc.count_pairs(f, r)
    assert np.array_equal(node_mat, ref_node) and np.array_equal(short_mat, ref_short)
    assert stats == tuple(int(x) for x in ref_stats)


@pytest.mark.parametrize("env", [
    {"VS_NO_SORT": "1"}, {"VS_NO_AGG": "1"}, {"VS_LOCUS_GLOBAL": "1"}, {"VS_EPT": "32"}, {"VS_EPT": "128"},
    {"VS_EPT": "6", "VS_GRID_PER_CU": "1"}, {"VS_ACC_FILL": "1"}, {"VS_ACC_FILL": "100"}, {"VS_NO_FAST": "1"},
    {"VS_ACC_QUEUE": "0"}, {"VS_ACC_GRID_PER_CU": "1", "VS_ACC_QUEUE": "0"}, {"VS_NO_XCD_MAP": "1"}, {"VS_GRID_PER_CU": "8"}, {"VS_NO_STD": "1"}, {"VS_SHORTCUT": "1"}, {"VS_SHORTCUT": "0"},
    {"VS_PHASE0": "1"}, {"VS_PHASE0": "1", "VS_NO_FAST": "1"}, {"VS_PHASE0": "1", "VS_SHORTCUT": "1"},  # (r5) the probe grid from offset 0, as before vs_seed_phase
    {"VS_ADAPT_GRID": "1"}, {"VS_ADAPT_GRID": "0"}, {"VS_ADAPT_GRID": "1", "VS_SHORTCUT": "1"}, {"VS_ADAPT_GRID": "1", "VS_NO_SORT": "1"},  # (r5) the adaptive step grid forced on / off
    {"VS_ACC_ROUND": "128"},
    {"VS_EPT": "32", "VS_ACC_ROWS": "1"}, {"VS_NO_MID": "1"},
    {"VS_ACC_ROWS": "1"}, {"VS_ACC_ROWS": "1", "VS_ROWS_PER_STRIP": "1"}, {"VS_ACC_ROWS": "1", "VS_ROWS_PER_STRIP": "64", "VS_ACC_FILL": "1"},
    {"VS_ACC_ROWS": "1", "VS_NO_SORT": "1"}, {"VS_ACC_ROWS": "1", "VS_ROWS_KEYS": "100"}, {"VS_ACC_ROWS": "1", "VS_ROWS_SUB": "2048", "VS_ROWS_KEYS": "7"},
    {"VS_ACC_ROWS": "1", "VS_LTAB_BITS": "0"}, {"VS_ACC_ROWS": "1", "VS_LTAB_BITS": "3"}, {"VS_ACC_ROWS": "1", "VS_LTAB_BITS": "7", "VS_ROWS_SUB": "1024"},
], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_every_kernel_variant_gives_the_same_counters(host, xctx, env, monkeypatch):
    """The tuning switches select other code paths (input order instead of locus order, global
    atomics instead of the LDS cell table, the global-atomic sort, other tile sizes, table
    flushed constantly / never): all of them must produce the oracle's counters."""
    ctx = xctx  # (the switches below exist only on a context made in experiment mode)
    from vstrains_amd import synth

    # 2 x 150 bases at k = 55: the shape the compile-time-layout instantiation (k_pe_tiles<true, true>)
    # is built for, so that VS_NO_STD / VS_NO_FAST / VS_EPT really switch between three kernels
    st = synth.make_strains(5, 4000, 0.04, seed=31)
    g = synth.compact_dbg(st, 55)
    fwd, rve = synth.sample_pairs(st, 12000, 150, seed=32, sub_rate=0.01, n_rate=0.01)
    orc = pe_oracle_c.Oracle(g.seqs, 55)
    want = orc.count_pairs(fwd, rve)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    (node_mat, short_mat, stats), _ = _gpu_matrices(host, ctx, g.seqs, fwd, rve, 55)
    assert np.array_equal(node_mat, want[0])
    assert np.array_equal(short_mat, want[1])
    assert stats == tuple(int(x) for x in want[2])
    assert ctx.last_kernel.startswith("k_pe_tiles"), ctx.last_kernel
    assert bool(ctx.last_launched & ctx.RAN_ROW_OWNERS) == (env.get("VS_ACC_ROWS") == "1")
    if "VS_ADAPT_GRID" in env and "VS_NO_STD" not in env:
        assert ctx.last_kernel.endswith(", true>") == (env["VS_ADAPT_GRID"] == "1"), ctx.last_kernel


def test_switches_do_not_exist_outside_experiment_mode(tmp_path):
    """A production process (no VS_EXPERIMENT) never reads the tuning environment: with the timing-only
    switches set -- VS_DEBUG_STOP stops k_pe_tiles after its first phase, VS_ACC_ABLATE=2 skips the counting,
    both give WRONG counters in experiment mode -- the counters still equal the oracle's and the default
    kernel runs.  With VS_EXPERIMENT=1 they stay unreachable too (that needs VS_EXPERIMENT=timing)."""
    script = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import pe_oracle_c
from vstrains_amd import pe as host, synth
st = synth.make_strains(5, 4000, 0.04, seed=31)
g = synth.compact_dbg(st, 55)
fwd, rve = synth.sample_pairs(st, 12000, 150, seed=32, sub_rate=0.01, n_rate=0.01)
want = pe_oracle_c.Oracle(g.seqs, 55).count_pairs(fwd, rve)
ctx = host.Context(0)
ctx.build_index(g.seqs, 55)
c = host.PeCounter(ctx)
c.add(ctx.pack_pairs(fwd, rve))
node_mat, short_mat, stats = c.result()
assert np.array_equal(node_mat, want[0]) and np.array_equal(short_mat, want[1]), "counters differ"
print("KERNEL", ctx.last_kernel)
""" % ROOT
    kernels = []
    for mode in (None, "1"):
        env = dict(os.environ, VS_DEBUG_STOP="1", VS_ACC_ABLATE="2", VS_NO_STD="1")
        env.pop("VS_EXPERIMENT", None)
        if mode:
            env["VS_EXPERIMENT"] = mode
        proc = subprocess.run([sys.executable, "-c", script], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
        kernels.append([l.split(" ", 1)[1] for l in proc.stdout.splitlines() if l.startswith("KERNEL")][0])
    # production ignores VS_NO_STD as well; experiment mode honours it (a parity-safe switch)
    assert kernels[0] != kernels[1], kernels


def test_full_size_block_properties(host, ctx, tmp_path):
    """BASELINE configs[2] itself -- `workload_for(2)`: the 5 039-node s_graph_L1 and the 10 M pairs of 2x150 bp of
    bench.py's default run (same stream seed) -- through size-independent properties: the counters of one 10 M
    block equal the sum over two 5 M halves and over four unequal pieces (any partition gives the same integer
    sums -- what read-block sharding relies on), the three pair classes add up to 10 M, short_mat is upper
    triangular with a positive diagonal wherever a row has any count, and a 200 k-pair prefix equals the CPU
    oracle bit for bit."""
    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[2]
    st, pre, names, seqs, cum, logger, _ = workload_for(2, str(tmp_path))
    assert len(seqs) >= 5000, len(seqs)

    class g:  # (the node set PE inference runs on)
        pass

    g.seqs = list(seqs)
    sub, nth, seed, L, R = int(0.005 * 2 ** 32), int(0.001 * 2 ** 32), 20250000 + 2, cfg["read_len"], cfg["total_pairs"]
    ctx.build_index(g.seqs, cfg["k"])

    def count(pieces):
        counter = host.PeCounter(ctx)
        for first, n in pieces:
            block = ctx.synth_pairs(st.genomes, cum, seed, first, n, L, sub, nth)
            counter.add(block)
            ctx.sync()
            block.free()
        return counter.result()

    whole = count([(0, R)])
    halves = count([(0, R // 2), (R // 2, R - R // 2)])
    ragged = count([(0, 1), (1, 4095), (4096, 3_000_001), (3_004_097, R - 3_004_097)])
    for other in (halves, ragged):
        assert np.array_equal(whole[0], other[0]) and np.array_equal(whole[1], other[1]) and whole[2] == other[2]
    node_mat, short_mat, stats = whole
    assert sum(stats) == R and stats[2] > 0.99 * R
    assert np.array_equal(short_mat, np.triu(short_mat))
    rows_with_counts = short_mat.sum(axis=1) > 0
    assert (short_mat.diagonal()[rows_with_counts] > 0).all()
    assert node_mat.sum() > 0 and short_mat.sum() > node_mat.sum()
    # prefix against the oracle
    M = 200_000
    fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, seed, 0, M, L, sub, nth)
    off = np.arange(M + 1, dtype=np.uint64) * np.uint64(L)
    orc = pe_oracle_c.Oracle(g.seqs, 55)
    ref = orc.count_pairs_raw(fw.reshape(-1), off, rv.reshape(-1), off, M)
    got = count([(0, M)])
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert got[2] == tuple(int(x) for x in ref[2])


def test_many_nodes_sort_and_counters(host, ctx):
    """20 k nodes: the LDS histogram of the locus sort needs more than 64 KB (dynamic LDS attribute),
    and the counter matrices are 2 x 1.6 GB; counts against the oracle on a chain graph."""
    rng = np.random.default_rng(5)
    k, n_nodes, step = 55, 20000, 40
    genome = "".join("ACGT"[i] for i in rng.integers(0, 4, size=n_nodes * step + k + 200))
    seqs = [genome[i * step: i * step + step + k] for i in range(n_nodes)]  # consecutive nodes overlap by k
    n = 30000
    starts = rng.integers(0, len(genome) - 600, size=n)
    from vstrains_amd import synth

    fwd = [genome[a: a + 150] for a in starts]
    rve = [synth.revcomp(genome[a + 250: a + 400]) for a in starts]
    orc = pe_oracle_c.Oracle(seqs, k)
    want = orc.count_pairs(fwd, rve)
    (node_mat, short_mat, stats), _ = _gpu_matrices(host, ctx, seqs, fwd, rve, k)
    assert np.array_equal(node_mat, want[0]) and np.array_equal(short_mat, want[1])
    assert stats == tuple(int(x) for x in want[2])
    assert node_mat.sum() > n


def test_counter_range_fold_and_auto_fold(host, ctx):
    """Cells in [2^31, 2^32) widen as unsigned; vs_counts_fold moves a buffer into int64 totals and
    empties it; add() folds by itself before 2 * pairs can reach 2^32 (the reference counts in
    int64, PE_Inference.py:139-140)."""
    import torch

    g, f, r = _dense_case(21, 400, 80, seed=61, snp=0.03)
    ctx.build_index(g.seqs, 21)
    base = host.PeCounter(ctx)
    base.add(ctx.pack_pairs(f, r))
    node0, short0, stats0 = base.result()
    n = len(g.seqs)
    c = host.PeCounter(ctx)
    big = np.zeros((2, n, n), dtype=np.uint32)
    big[0, 1, 2] = 2 ** 31 + 5
    big[1, 1, 1] = 2 ** 32 - 1
    big[1, 0, 0] = 2 ** 32 - 2
    c.mats.copy_(torch.from_numpy(big.view(np.int32)))
    c.pairs_in_buffer = 2 ** 31 - 1 - 100   # the next block of 400 pairs does not fit: add() must fold first
    c.add(ctx.pack_pairs(f, r))
    assert c.wide is not None and c.pairs_in_buffer == len(f)
    node, short, stats = c.result()
    assert np.array_equal(node, node0 + big[0].astype(np.int64))
    assert np.array_equal(short, short0 + big[1].astype(np.int64))
    assert stats == stats0
    # the graph stages' link table from the same counter (int64 totals + live buffer)
    from vstrains_amd.graph.hip_ops import HipPeLinks

    links = HipPeLinks.from_counter(ctx, c, [str(i) for i in range(n)])
    want = node + node.T + short + short.T
    want[np.arange(n), np.arange(n)] = node.diagonal() + short.diagonal()
    assert np.array_equal(links.to_numpy(), want)


def test_rccl_allreduce_through_the_c_abi_one_rank(host, ctx):
    """vs_comm_* / vs_pe_allreduce (ncclAllReduce on the ctx stream) with a one-rank communicator:
    the sum over one rank is the identity, for the uint32 buffers and for int64 totals."""
    import ctypes as C

    import torch

    from vstrains_amd import _native as nat

    L = nat.lib()
    uid = (C.c_uint8 * 128)()
    nat.check(ctx._h, L.vs_comm_unique_id(ctx._h, uid))
    comm = C.c_void_p()
    nat.check(ctx._h, L.vs_comm_init_rank(ctx._h, 1, uid, 0, C.byref(comm)))
    n = 37
    rng = np.random.default_rng(3)
    a = rng.integers(0, 2 ** 32, size=(2, n, n), dtype=np.uint64).astype(np.uint32)
    mats = torch.from_numpy(a.view(np.int32)).cuda()
    stats = torch.tensor([5, 6, 2 ** 40], dtype=torch.int64).cuda()
    torch.cuda.synchronize()
    nat.check(ctx._h, L.vs_pe_allreduce(ctx._h, comm, C.c_void_p(mats[0].data_ptr()), C.c_void_p(mats[1].data_ptr()),
                                        C.c_void_p(stats.data_ptr()), n, 0))
    ctx.sync()
    assert np.array_equal(mats.cpu().numpy().view(np.uint32), a)
    assert stats.cpu().tolist() == [5, 6, 2 ** 40]
    wide = torch.from_numpy(rng.integers(0, 2 ** 62, size=(2, n, n), dtype=np.int64)).cuda()
    keep = wide.clone()
    torch.cuda.synchronize()
    # two separate allocations take the two-call branch
    w0, w1 = wide[0].clone(), wide[1].clone()
    nat.check(ctx._h, L.vs_pe_allreduce(ctx._h, comm, C.c_void_p(w0.data_ptr()), C.c_void_p(w1.data_ptr()), None, n, 1))
    nat.check(ctx._h, L.vs_pe_allreduce(ctx._h, comm, C.c_void_p(wide[0].data_ptr()), C.c_void_p(wide[1].data_ptr()), None, n, 1))
    ctx.sync()
    assert torch.equal(wide, keep) and torch.equal(w0, keep[0]) and torch.equal(w1, keep[1])
    nat.check(ctx._h, L.vs_comm_destroy(ctx._h, comm))


def test_fastq_open_refuses_invalid_utf8_and_blocks_take_dirty_reads(host, ctx, tmp_path):
    """Bytes that are not valid UTF-8 are refused when the files are opened (ValueError, as the reference's text-mode
    read raises UnicodeDecodeError before it counts anything; VALID multi-byte characters are read as the reference reads
    them -- golden case utf8_reads_k21); vs_fastq_block packs on the host cores, a block with lower-case / IUPAC bytes
    falls back to the device packer (mask + position lists) and counts like the oracle."""
    bad = tmp_path / "bad.fq"
    bad.write_bytes("@r\u00e9ad\nACGTACGTAC\n+\nIIIIIIIIII\n@b\nAC".encode("utf-8") + b"\xc3TACGTAC\n+\nIIIIIIIIII\n")
    with pytest.raises(ValueError):
        host.FastqPair(str(bad), str(bad), ctx)
    g, f, r = _dense_case(21, 700, 90, seed=99, snp=0.03)
    rng = np.random.default_rng(1)
    for lst in (f, r):
        for i in rng.choice(len(lst), size=80, replace=False):
            s_ = lst[int(i)]
            p_ = int(rng.integers(0, len(s_)))
            lst[int(i)] = s_[:p_] + str(rng.choice(list("nRYacgt*N"))) + s_[p_ + 1:]
    from vstrains_amd import synth

    (tmp_path / "f.fq").write_text(synth.fastq_text(f, "f"))
    (tmp_path / "r.fq").write_text(synth.fastq_text(r, "r"))
    ctx.build_index(g.seqs, 21)
    counter = host.PeCounter(ctx)
    fq = host.FastqPair(str(tmp_path / "f.fq"), str(tmp_path / "r.fq"), ctx)
    from vstrains_amd import pe_inference

    pe_inference.count_fastq(ctx, fq, counter, 0, len(fq), batch=256)  # several blocks, staging sets alternate
    fq.close()
    node_mat, short_mat, stats = counter.result()
    ref = pe_oracle_c.Oracle(g.seqs, 21).count_pairs(f, r)
    assert np.array_equal(node_mat, ref[0]) and np.array_equal(short_mat, ref[1])
    assert stats == tuple(int(x) for x in ref[2])


@pytest.mark.parametrize("exchange,ranks", [("dense", 2), ("compact", 2), ("compact", 8)])
def test_sharded_drop_in_two_ranks_on_one_gpu(tmp_path, exchange, ranks):
    """The PE drop-in under torchrun with two ranks: each rank counts its contiguous block of the
    pairs with the real kernels, the counters are summed, rank 0 alone touches the output
    directory and writes the files -- byte-identical to the reference's.  One GPU here, so both
    ranks use device 0 and the sum goes through gloo (RCCL needs a device per rank; the RCCL call
    itself is exercised by the one-rank tests and by bench.py --gpus N).  (r6) Also with EIGHT ranks on the one device: eight
    cooperative FASTQ shards, the packed two-collective sum reduced to rank 0 (the only rank that writes)."""
    import socket

    name, d, meta = [c for c in pe_cases() if c[0] == "errors_k21"][0]
    out = tmp_path / "aln"
    out.mkdir()
    (out / "stale_file").write_text("x")  # rank 0 wipes the directory (PE_Inference.py:93-96); nobody else may
    # ("compact": the ranks sum the occupied 64-cell stretches of their counters instead of the whole buffers,
    # dist.sum_counts_compact -- the default; "dense": turned off for every rank)
    env = dict(os.environ, VS_DIST_BACKEND="gloo", VS_DIST_DEVICE="0", VS_COMPACT_ALLREDUCE="1" if exchange == "compact" else "0")
    for attempt in range(3):  # (a port that was free when asked for may be taken a moment later: ask again)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        proc = subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
             "--master-port", str(port), "-m", "vstrains_amd.pe_inference", "-g", os.path.join(d, "graph.gfa"), "-o", str(out),
             "-f", os.path.join(d, "fwd.fq"), "-r", os.path.join(d, "rve.fq"), "-k", str(meta["k"])],
            cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
        if proc.returncode == 0 or "address already in use" not in proc.stderr:
            break
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert _read(out / "pe_info") == _read(os.path.join(d, "pe_info"))
    assert _read(out / "st_info") == _read(os.path.join(d, "st_info"))
    assert sorted(os.listdir(out)) == ["pe_info", "st_info"]
    assert proc.stdout.count("result stored in:") == 1  # rank 0 only


def test_bench_step_with_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` launches its own ranks.  bench.py's multi-rank step (two counter buffers, the all-reduce of step i waited for when its
    buffer comes up again, max over ranks, one JSON line from rank 0) with two ranks sharing the
    one GPU and gloo as the collective: the line must account for both ranks' pairs.  (A functional
    run; the rate of two processes sharing a GPU says nothing.)"""
    import json

    # `python bench.py --gpus 2` exactly as the driver calls it: bench.py itself starts the two ranks
    # (a child torchrun, before anything touches the GPU) and relays rank 0's line and the exit status
    env = dict(os.environ, VS_DIST_BACKEND="gloo", VS_DIST_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run(
        [sys.executable, "bench.py", "--gpus", "2", "--config", "1", "--pairs", "200000", "--steps", "3", "--warmup", "1",
         "--cpu-seconds", "0", "--no-extract"],
        cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["collective_backend"] == "gloo" and out["config"]["rccl_ranks"] == 0  # (nccl -> rccl_ranks = 2)
    assert 0 < out["roofline"]["frac_step"] <= out["roofline"]["frac"]
    assert abs(out["value"] - 2 * 200000 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    st = out["pe_stats"]
    assert st["n_reads"] + st["short_reads"] + st["used_reads"] == 2 * 200000  # both ranks' pairs, summed
    weak_sums = (st["node_mat_sum"], st["short_mat_sum"])
    # strong scaling: the two ranks split ONE block of 400 000 pairs -- the same pairs as the weak run's two blocks, so the
    # same counters; the line counts the block once
    proc = subprocess.run(
        [sys.executable, "bench.py", "--gpus", "2", "--config", "1", "--pairs", "400000", "--steps", "2", "--warmup", "1",
         "--cpu-seconds", "0", "--no-extract", "--scaling", "strong"],
        cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["pairs_per_gpu"] == 200000
    assert abs(out["value"] - 400000 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    st = out["pe_stats"]
    assert st["n_reads"] + st["short_reads"] + st["used_reads"] == 400000
    assert (st["node_mat_sum"], st["short_mat_sum"]) == weak_sums
    assert "compact" in out["config"]["parallelism"] or "dense" in out["config"]["parallelism"]
    # (r5) the same with counters that keep a dirty-tile map (VS_TRACK_TILES=1: what counters of 2 GiB and more do by
    # themselves): resets zero the marked tiles only, the exchange takes its occupancy from the map, the ranks OR their maps
    env_t = dict(env, VS_TRACK_TILES="1", VS_DIST_TIMING="1")
    proc = subprocess.run(
        [sys.executable, "bench.py", "--gpus", "2", "--config", "1", "--pairs", "200000", "--steps", "3", "--warmup", "1",
         "--cpu-seconds", "0", "--no-extract"],
        cwd=ROOT, capture_output=True, text=True, env=env_t, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][0])
    st = out["pe_stats"]
    assert (st["node_mat_sum"], st["short_mat_sum"]) == weak_sums
    assert "compact" in out["config"]["parallelism"]
    # (tile granularity: more stretches than the scan would find, still a fraction of the buffer)
    assert 0 < out["exchange_timing"]["occupied_stretches_of_the_union"] < 0.5 * out["exchange_timing"]["stretches"]


def test_two_gpu_rccl_step_and_sharded_drop_in(tmp_path):
    """On a box with two or more GPUs: `bench.py --gpus 2` and the sharded PE drop-in with one rank per GPU and RCCL as the
    collective (backend "nccl" IS RCCL on ROCm) -- the path the 1-GPU boxes of this pool can only run over gloo.  Skips
    where there is one GPU."""
    import json
    import socket

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU here: RCCL with two ranks needs two devices")
    env = dict(os.environ)
    for key in ("VS_DIST_BACKEND", "VS_DIST_DEVICE", "WORLD_SIZE"):
        env.pop(key, None)
    proc = subprocess.run(
        [sys.executable, "bench.py", "--gpus", "2", "--config", "1", "--pairs", "200000", "--steps", "3", "--warmup", "1",
         "--cpu-seconds", "0", "--no-extract"],
        cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["collective_backend"] == "nccl" and out["config"]["rccl_ranks"] == 2
    st = out["pe_stats"]
    assert st["n_reads"] + st["short_reads"] + st["used_reads"] == 2 * 200000
    name, d, meta = [c for c in pe_cases() if c[0] == "errors_k21"][0]
    outdir = tmp_path / "aln"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    proc = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), "-m", "vstrains_amd.pe_inference", "-g", os.path.join(d, "graph.gfa"), "-o", str(outdir),
         "-f", os.path.join(d, "fwd.fq"), "-r", os.path.join(d, "rve.fq"), "-k", str(meta["k"])],
        cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert _read(outdir / "pe_info") == _read(os.path.join(d, "pe_info"))
    assert _read(outdir / "st_info") == _read(os.path.join(d, "st_info"))


@pytest.mark.parametrize("mode", ["boundaries", "tile_shapes"])
def test_randomized_campaign_short(mode):
    """tests/fuzz_pe.py for ten seconds per mode: random graph / read shapes around the points where
    vs_pe_count switches kernels, every draw against the C oracle (the full campaigns of the round:
    4 510 draws, no mismatch -- DESIGN.md 8)."""
    env = dict(os.environ)
    if mode == "tile_shapes":
        env["FUZZ_STD"] = "1"
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_pe.py"), "10", "7"], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "mismatches 0" in proc.stdout and "draws 0," not in proc.stdout, proc.stdout[-500:]
    # the number of draws this run really covered goes into the pytest summary (warnings section of the log)
    import warnings

    warnings.warn(UserWarning("fuzz_pe %s: %s" % (mode, [l for l in proc.stdout.splitlines() if l.startswith("draws ")][-1])))


def test_occupied_stretches_kernel_equals_the_torch_expression(host, ctx):
    """vs_counts_occupied (round 5: the one pass over a counter buffer that finds its non-zero 64-cell stretches for the
    multi-GPU exchange, dist.sum_counts_compact) against the torch expression it replaces, on uint32 counters and int64
    totals, banded, dense, empty, and sizes that are no multiple of anything."""
    import torch

    from vstrains_amd import dist as vdist

    c = host.PeCounter.__new__(host.PeCounter)
    c.torch, c.ctx, c.device = torch, ctx, torch.device("cuda:%d" % ctx.device)
    gen = torch.Generator().manual_seed(9)
    for dtype in (torch.int32, torch.int64):
        for m, fill in ((1, 0.5), (7, 0.0), (1000, 0.001), (4097, 0.3), (100003, 0.00002)):
            cells = (torch.rand((m, 64), generator=gen) < fill).to(dtype) * (-5 if dtype == torch.int32 else 2 ** 40)
            if m > 6:
                cells[5, 63] = 1
                cells[m - 1, 0] = -1
            head = cells.to(c.device)
            assert torch.equal(c._occupied(head), vdist._occupancy(head)), (dtype, m, fill)
