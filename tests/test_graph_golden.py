"""The graph stages against the reference's outputs on the CPU: the NATIVE stage engine (vstrains_amd/csrc/vs_stage.cpp,
the code the product runs, here over the CPU checker of its three device operations, oracle/stage_check.cpp) and the
Python restatement of the stages it is compared with (oracle/graph_stages, over oracle/graph_ops.py: closed-form links
and the literal pe_info dict).  Every intermediate GFA, contig file and the final strain.paths / strain.fasta must match
the golden run byte for byte (sequences via digest)."""
import os

import pytest

from graph_case import Case, case_names, compare, file_logger, quiet_logger
import native_check
from oracle import graph_ops as chk
from oracle import pe_oracle
from oracle.graph_stages.links import LiveLinks
from oracle.graph_stages.run import PythonStages
from vstrains_amd.graph import pipeline


class FixtureLinks:
    """PE files come from the fixture (the reference's own hand-off: two N^2-line text files)."""

    def __init__(self, case):
        self.case = case

    def pe_links(self, gfa, aln_dir, fwd, rve, ksize, names):
        assert ksize == self.case.meta["k"]
        # the reference's PE script dies with KeyError on a node base outside ACGT (self-loop
        # segments are lower-cased upstream); the oracle restates that, so does the device index
        _, seqs = pe_oracle.read_gfa_segments(gfa)
        pe_oracle.build_table(seqs, ksize + 1)
        self.case.write_info_files(names, aln_dir)
        return chk.DictPeLinks.from_files(names, os.path.join(aln_dir, "pe_info"), os.path.join(aln_dir, "st_info"))



class CheckerBackend(FixtureLinks, PythonStages):
    """The Python restatement of the stages over the numpy / dict checker."""

    def __init__(self, case, literal_dict):
        FixtureLinks.__init__(self, case)
        self.literal = literal_dict
        self.graph_ops = chk.NumpyGraphOps()

    def live_links(self, table):
        return chk.DictLiveLinks(table) if self.literal else LiveLinks(table)


class NativeBackend(FixtureLinks):
    """The native stage engine over the C++ checker of its device operations."""

    def native_stage(self, table):
        return native_check.stage_over_checker(table.names, native_check.dense_links(table))


class NativeSparseBackend(FixtureLinks):
    """The native stage engine over the C++ checker with the link table as CSR rows of its non-zero cells (the form the
    configs[4]-size GPU test hands the checker: a dense table of 54 k nodes would be 23.7 GB)."""

    def native_stage(self, table):
        import numpy as np

        p0 = native_check.dense_links(table)
        rows, cols = np.nonzero(p0)
        row_ptr = np.zeros(p0.shape[0] + 1, dtype=np.uint64)
        row_ptr[1:] = np.cumsum(np.bincount(rows, minlength=p0.shape[0]))
        return native_check.stage_over_checker_sparse(table.names, row_ptr, cols.astype(np.uint32), p0[rows, cols])


def make_backend(case, engine):
    if engine == "native_engine_sparse_links":
        return NativeSparseBackend(case)
    return NativeBackend(case) if engine == "native_engine" else CheckerBackend(case, engine == "literal_dict_links")


@pytest.mark.parametrize("name", ["ten_strain_k31", "hiv_like_k55", "six_strain_k21", "circular_k21"])
def test_sparse_link_table_of_the_checker_gives_the_same_files(name, tmp_path):
    case = Case(name)
    inp = case.inputs(str(tmp_path))
    out = str(tmp_path / "out")
    pipeline.run(case.args(inp, out), file_logger(out), make_backend(case, "native_engine_sparse_links"))
    problems, _ = compare(case, out)
    assert not case.binding(problems), problems


def test_the_engines_check_of_its_restricted_table_passes_is_live(tmp_path):
    """path_extension looks only at the link-table entries a re-initialisation noted (csrc/vs_stage.cpp: table_filtered).
    Under VS_CHECK_UNTOUCHED=1 (tests/conftest.py) the engine runs the whole pass afterwards and raises if anything was left.
    That the check would catch a pass that skipped too much is shown by taking the notes away (VS_STAGE_DROP_NOTES=1, read
    when the library loads, hence the child process): the same golden case that passes above must then end with the
    engine's state error."""
    import subprocess
    import sys

    script = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from graph_case import Case, file_logger\n"
        "from vstrains_amd.graph import pipeline\n"
        "import test_graph_golden as T\n"
        "case = Case('three_strain_k21'); inp = case.inputs(%r); out = %r\n"
        "try:\n"
        "    pipeline.run(case.args(inp, out), file_logger(out), T.make_backend(case, 'native_engine'))\n"
        "except RuntimeError as e:\n"
        "    print('RAISED', e)\n"
        "else:\n"
        "    print('PASSED')\n"
    )
    for drop, want in (("1", "RAISED"), (None, "PASSED")):
        work = tmp_path / ("notes_dropped" if drop else "as_shipped")
        work.mkdir()
        child = script % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
                          str(work), str(work / "out"))
        env = dict(os.environ, VS_CHECK_UNTOUCHED="1")
        env.pop("VS_STAGE_DROP_NOTES", None)
        if drop:
            env["VS_STAGE_DROP_NOTES"] = drop
        proc = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env, timeout=600)
        last = [l for l in proc.stdout.splitlines() if l.startswith(("RAISED", "PASSED"))]
        assert last and last[-1].startswith(want), (proc.stdout[-2000:], proc.stderr[-2000:])
        if drop:
            assert "passed over" in last[-1] or "not looked at" in last[-1], last[-1]


@pytest.mark.parametrize("engine", ["native_engine", "closed_form_links", "literal_dict_links"])
@pytest.mark.parametrize("name", case_names())
def test_pipeline_matches_reference_outputs(name, engine, tmp_path):
    case = Case(name)
    inp = case.inputs(str(tmp_path))
    out = str(tmp_path / "out")
    args = case.args(inp, out)
    logger = file_logger(out)  # OUT/vstrains.log: its INFO lines are compared with the reference's too
    if case.meta["returncode"] != 0:
        # the reference exits non-zero (its PE subprocess raises KeyError); so must this build, after
        # writing the same files up to that point
        with pytest.raises(case.expected_exception()):
            pipeline.run(args, logger, make_backend(case, engine))
    else:
        pipeline.run(args, logger, make_backend(case, engine))
    for h in list(logger.handlers):
        h.flush()
    # files the reference itself does not produce deterministically (they change with PYTHONHASHSEED, see case.json)
    # and files that change with the stand-in's model of graph-tool's adjacency container are compared too, but
    # only the others are binding (graph_case.Case.non_binding)
    problems, _ = compare(case, out)
    binding = case.binding(problems)
    assert not binding, binding


# Files of the committed (rotate in-edge model, PYTHONHASHSEED 0) reference run that this build does NOT reproduce -- all of
# them in one case, all of them files the reference itself writes differently under another hash seed (it iterates sets of
# contig names, Decomposition.py:188,444; this build iterates in insertion order).  Everything else is held to the
# committed run file by file, so that the cases whose binding set is small (hiv_like_k55 binds 4 of 59 files) still fail
# when something regresses.
KNOWN_SEED_DEPENDENT_DIFFERENCES = {"inv5_k31_mc_s219": 27}


@pytest.mark.parametrize("name", case_names())
def test_every_file_equals_the_committed_rotate_model_run(name, tmp_path):
    case = Case(name)
    inp = case.inputs(str(tmp_path))
    out = str(tmp_path / "out")
    args = case.args(inp, out)
    logger = file_logger(out)
    try:
        pipeline.run(args, logger, NativeBackend(case))
    except (KeyError, RecursionError) as err:
        assert case.meta["returncode"] != 0 and isinstance(err, case.expected_exception())
    for h in list(logger.handlers):
        h.flush()
    problems, _ = compare(case, out)
    allowed = KNOWN_SEED_DEPENDENT_DIFFERENCES.get(name, 0)
    assert len(problems) == allowed, problems
    if allowed:
        seed_dependent = set(case.meta["differs_under_other_hashseeds"])
        assert all(p.split(" ", 1)[1] in seed_dependent for p in problems), problems


@pytest.mark.parametrize("config", [0, 1, 2])
def test_native_engine_reproduces_the_real_reference_command(config, tmp_path):
    """The same pin as tests/test_configs_gpu.py::test_cli_reproduces_the_real_reference_command* on a box without a GPU: the
    C oracle counts the pairs (100 000 at configs[0], 1 M at configs[1], the 200 k-pair prefix at configs[2]), the native stage
    engine runs over the CPU checker of its device operations, and the files every run of the REAL reference command agrees on
    (58 at configs[0]; 83 at configs[1]: 853 nodes, 20 strains; 124 at configs[2]: 5 039 nodes, 30 strains -- the reference
    took 73 minutes for that leg, profiles/r6/reference_stages_config2.json) must come out the same."""
    import numpy as np

    from graph_case import reference_command_inputs, reference_command_problems
    from oracle import pe_oracle_c
    from vstrains_amd.graph.hip_ops import HipPeLinks  # noqa: F401  (only to show what the product would use)

    inp, want = reference_command_inputs(str(tmp_path / "work"), config)

    class OracleCountsNativeStages:
        def pe_links(self, gfa, aln_dir, fwd, rve, ksize, names):
            ids, seqs = pe_oracle.read_gfa_segments(gfa)
            assert list(ids) == list(names)
            f, r = pe_oracle.fastq_sequences(fwd), pe_oracle.fastq_sequences(rve)
            node_mat, short_mat, _ = pe_oracle_c.Oracle(seqs, ksize).count_pairs(f, r)
            os.makedirs(aln_dir, exist_ok=True)
            if config >= 2:  # (25 M lines of text per file and a 12.7 M-key dict: the table as a symmetric matrix, no text)
                import profile_extract_cpu as pec

                return pec.NumpyPeLinks(names, node_mat, short_mat)
            for fname, mat in (("pe_info", node_mat), ("st_info", short_mat)):
                with open(os.path.join(aln_dir, fname), "w") as fh:
                    fh.write(pe_oracle.matrix_text(ids, mat))
            return chk.DictPeLinks(names, node_mat, short_mat)

        def native_stage(self, table):
            return native_check.stage_over_checker(table.names, native_check.dense_links(table))

    import argparse

    out = str(tmp_path / "out")
    for sub in ("gfa", "tmp", "paf", "aln"):
        os.makedirs(os.path.join(out, sub))
    args = argparse.Namespace(gfa_file=inp["gfa"], path_file=inp["paths"], fwd=inp["fwd"], rve=inp["rve"], output_dir=out,
                              min_cov=None, min_len=250, ref_file=None, dev=False)
    pipeline.run(args, quiet_logger(), OracleCountsNativeStages())
    problems = reference_command_problems(out, want)
    assert not problems, problems


def test_report_which_golden_files_do_not_depend_on_the_in_edge_model(capsys):
    """The graph fixtures come from the real reference CLI behind a model of graph-tool's adjacency
    order (unpinned).  Every case was generated under both in-edge-order models;
    this prints, per case, which reference outputs are the same under both -- those pin the
    stages independently of the recalled order -- and insists that the suite holds cases that are
    invariant in EVERY file while their reference log shows link splits, coverage matching and
    trivial splits (tests/golden/make_graph_golden.py --search-invariant)."""
    rows = []
    fully = []
    for name in case_names():
        meta = Case(name).meta
        if meta["returncode"] != 0:
            continue
        dep = Case(name).model_dependent()
        indep = [f for f in meta["files"] if f not in dep]
        rows.append("%-32s %3d of %3d files model-independent%s%s" % (
            name, len(indep), len(meta["files"]), "; strain.paths too" if "strain.paths" in indep else "",
            ("; exercised: %s" % meta["operations_in_reference_log"]) if "operations_in_reference_log" in meta else ""))
        if not dep:
            fully.append((name, meta.get("operations_in_reference_log", {})))
    with capsys.disabled():
        print("\n" + "\n".join(rows))
    exercised = [n for n, ops in fully if ops and all(ops.values())]
    # >= 10 cases identical in EVERY file under the four small-doubt models (rotate / plain / lifo / swappop) while the
    # reference's log shows link splits, coverage matching and trivial splits; among them k = 55 and a case with -mc
    assert len(exercised) >= 10, fully
    metas = {n: Case(n).meta for n in exercised}
    assert any(m["k"] == 55 for m in metas.values()), "no fully invariant k = 55 case"
    assert any("-mc" in m["cli_extra"] for m in metas.values()), "no fully invariant case with -mc"


@pytest.mark.parametrize("name", ["circular_tiny_bound_k21_s480348", "circular_tiny_bound_k21_s810092", "circular_runaway_k55"])
def test_runaway_trivial_split_ends_like_the_reference_and_quickly(name, tmp_path):
    """Round 5 (fuzz_reference draw 236 of campaign 778, unresolved in round 4): on some circular genomes the reference's
    ``global_trivial_split`` forks "X*B" into "X*B*B" for ever and only its N^2 bound stops it (Decomposition.py:699-705,
    "Strange topology detected").  On 9- and 12-node graphs the reference then carries on and finishes; on the 250-node
    k = 55 draw the fork chain (8 549 links) is longer than its recursive merge_id (Utilities.py:318-327) can follow and it
    exits with RecursionError after writing graph_S6.gfa.  This build must end the same way on each -- the files are
    compared by the tests above -- and do so in seconds: before the recursion limit was restated the engine ran out of
    memory on the k = 55 draw (65 GB) where the reference needs 90 s."""
    import time

    case = Case(name)
    inp = case.inputs(str(tmp_path))
    out = str(tmp_path / "out")
    args = case.args(inp, out)
    logger = file_logger(out, "runaway")
    t0 = time.time()
    if case.meta["returncode"] != 0:
        assert case.expected_exception() is RecursionError
        with pytest.raises(RecursionError, match="maximum recursion depth exceeded"):
            pipeline.run(args, logger, NativeBackend(case))
    else:
        pipeline.run(args, logger, NativeBackend(case))
    took = time.time() - t0
    for h in list(logger.handlers):
        h.flush()
    with open(os.path.join(out, "vstrains.log")) as fh:
        assert "Strange topology detected, exit trivial split immediately" in fh.read()
    assert took < 30.0, took


def test_a_runaway_that_outgrows_memory_ends_with_memoryerror(tmp_path, monkeypatch):
    """Fuzz draw 997 of campaign 782 (round 5): after a stage that already ran away the N^2 bound of ``global_trivial_split``
    is millions of forks with ids that grow by two characters each -- the reference was still inside that loop after a
    quarter of an hour (its campaign run: reference_timeout), the engine had filled 40 GB after seven minutes.  Past 2 GiB of
    id text the engine now ends the call with MemoryError (draw 997: after 34 s, 55 911 forks).  Here the limit is set to
    1 MB on the k = 55 runaway golden, whose ids reach 73 MB: the MemoryError comes before the reference's RecursionError."""
    monkeypatch.setenv("VS_STAGE_NAME_LIMIT_MB", "1")
    case = Case("circular_runaway_k55")
    inp = case.inputs(str(tmp_path))
    out = str(tmp_path / "out")
    with pytest.raises(MemoryError, match="ran away"):
        pipeline.run(case.args(inp, out), file_logger(out, "valve"), NativeBackend(case))
