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


def make_backend(case, engine):
    return NativeBackend(case) if engine == "native_engine" else CheckerBackend(case, engine == "literal_dict_links")


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
        with pytest.raises(KeyError):
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
