"""The vstrains-compatible command line on CPU (device side replaced by the checker): argument
handling and output-directory rules of the reference (``vstrains:136-191``), log file, outputs."""
import os

import pytest

from graph_case import Case, compare
from test_graph_golden import CheckerBackend
from vstrains_amd import cli


def _argv(inp, out, extra=()):
    return ["-a", "spades", "-g", inp["gfa"], "-p", inp["paths"], "-o", out, "-fwd", inp["fwd"], "-rve", inp["rve"]] + list(extra)


def test_cli_end_to_end_with_checker_backend(tmp_path, capsys):
    case = Case("three_strain_scrambled_k21")
    inp = case.inputs(str(tmp_path))
    out = str(tmp_path / "out") + "/"          # trailing slash is stripped (vstrains:173-174)
    timings = cli.main(_argv(inp, out), backend=CheckerBackend(case, False))
    problems, _ = compare(case, out[:-1])
    assert not problems, problems
    log = open(os.path.join(out, "vstrains.log")).read()
    for needle in ("Welcome to VStrains!", "VStrains-SPAdes started", ">>>STAGE: parsing graph and contigs",
                   ">>>STAGE: preprocess", "graph kmer size: 21", ">>>STAGE: contig path extension",
                   ">>>STAGE: final process", ">>>STAGE: generate result", "VStrains-SPAdes finished",
                   "Thanks for using VStrains", "Elapsed time: "):
        assert needle in log, needle
    assert set(timings) == {"pe_inference_s", "strain_extract_s", "total_s"}
    for sub in ("gfa", "tmp", "paf", "aln"):
        assert os.path.isdir(os.path.join(out, sub))


def test_cli_refuses_a_used_output_directory(tmp_path, capsys):
    case = Case("two_strain_bubbles_k21")
    inp = case.inputs(str(tmp_path))
    out = str(tmp_path / "out")
    os.makedirs(os.path.join(out, "gfa"))
    with pytest.raises(SystemExit) as ei:
        cli.main(_argv(inp, out), backend=CheckerBackend(case, False))
    assert ei.value.code == 1
    assert "Current output directory is not empty" in capsys.readouterr().out


@pytest.mark.parametrize("broken,message", [
    ("gfa", "Path to the assembly graph is required"),
    ("paths", "Path to Contig file from SPAdes"),
])
def test_cli_checks_input_paths(tmp_path, capsys, broken, message):
    case = Case("two_strain_bubbles_k21")
    inp = dict(case.inputs(str(tmp_path)))
    inp[broken] = str(tmp_path / "missing")
    with pytest.raises(SystemExit) as ei:
        cli.main(_argv(inp, str(tmp_path / "out")), backend=CheckerBackend(case, False))
    assert ei.value.code == 1
    assert message in capsys.readouterr().out


def test_cli_rejects_negative_thresholds_and_reference_mode(tmp_path, capsys):
    case = Case("two_strain_bubbles_k21")
    inp = case.inputs(str(tmp_path))
    for extra in (["-ml", "-5"], ["-mc", "-1"], ["-r", inp["gfa"]]):
        with pytest.raises(SystemExit) as ei:
            cli.main(_argv(inp, str(tmp_path / "o"), extra), backend=CheckerBackend(case, False))
        assert ei.value.code == 1


def test_graph_without_edges_exits_like_the_reference(tmp_path, capsys):
    """VStrains_SPAdes.py:112-116: no edge -> "invalid kmer-size" -> exit 1."""
    gfa = tmp_path / "g.gfa"
    gfa.write_text("S\t1\t" + "ACGT" * 80 + "\tDP:f:50.0\nS\t2\t" + "TTGCA" * 70 + "\tDP:f:40.0\n")
    paths = tmp_path / "c.paths"
    paths.write_text("NODE_1_length_320_cov_50.0\n1+\nNODE_1_length_320_cov_50.0'\n1-\n")
    case = Case("two_strain_bubbles_k21")
    with pytest.raises(SystemExit) as ei:
        cli.main(["-a", "spades", "-g", str(gfa), "-p", str(paths), "-o", str(tmp_path / "out"), "-fwd", "x", "-rve", "y"],
                 backend=CheckerBackend(case, False))
    assert ei.value.code == 1
    assert "invalid kmer-size" in open(tmp_path / "out" / "vstrains.log").read()
