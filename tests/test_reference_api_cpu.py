"""Boundary 2: the stages under the reference's own names and argument lists
(vstrains_amd/graph/reference_api.py).  The driver below is shaped like the reference's
``utils/VStrains_SPAdes.py:134-272`` -- the same calls in the same order with the same
positional arguments -- and must reproduce the golden outputs of the real reference run."""
import os

import numpy
import pytest

from graph_case import Case, compare, quiet_logger
from oracle import graph_ops as chk
from vstrains_amd.graph import pipeline
from vstrains_amd.graph import reference_api as api
from vstrains_amd.graph.contigs import drop_duplicate_contigs, resolve_contigs, restore_repeats, trim_contigs
from vstrains_amd.graph.formats import read_stage_gfa, write_contig_fasta, write_contig_paths, write_stage_gfa


class CheckerBackend:
    """The link table from the fixture's text files; the stage handle over the CPU checker of its device operations."""

    def links_from_files(self, names, pe_file, st_file):
        return chk.DictPeLinks.from_files(list(names), pe_file, st_file)

    def native_stage(self, table):
        import native_check

        return native_check.stage_over_checker(table.names, native_check.dense_links(table))


def reference_shaped_driver(args, logger, case):
    """utils/VStrains_SPAdes.py:run from line 134 on, with the imports swapped for reference_api."""
    TEMP_DIR = args.output_dir
    pre = pipeline.prepare(args, logger)  # lines 30-116 (upstream of the hot path)
    graph1, simp_node_dict1, simp_edge_dict1, contig_dict = pre.g1, pre.nodes1, pre.edges1, pre.contigs
    case.write_info_files(list(simp_node_dict1.keys()), "{0}/aln".format(TEMP_DIR))  # (the PE subprocess, :119-132)
    pe_info_file = "{0}/aln/pe_info".format(TEMP_DIR)
    st_info_file = "{0}/aln/st_info".format(TEMP_DIR)
    pe_info, dcpy_pe_info = api.process_pe_info(simp_node_dict1.keys(), pe_info_file, st_info_file)
    api.edge_cleaning(graph1, simp_edge_dict1, contig_dict, pe_info, logger)
    graph2, simp_node_dict2, simp_edge_dict2 = api.store_reinit_graph(
        graph1, simp_node_dict1, simp_edge_dict1, logger, "{0}/gfa/es_graph_L2.gfa".format(TEMP_DIR))
    write_contig_paths(contig_dict, "{0}/tmp/pre_contigs.paths".format(TEMP_DIR))
    write_contig_fasta(graph2, simp_node_dict2, contig_dict, "{0}/tmp/pre_contigs.fasta".format(TEMP_DIR))
    graph5, simp_node_dict5, simp_edge_dict5 = api.iter_graph_disentanglement(
        graph2, simp_node_dict2, simp_edge_dict2, contig_dict, pe_info, args.ref_file, logger,
        0.05 * numpy.median([graph2.vdp[v] for v in range(graph2.num_vertices())]), TEMP_DIR)
    write_contig_paths(contig_dict, "{0}/tmp/post_contigs.paths".format(TEMP_DIR))
    write_contig_fasta(graph5, simp_node_dict5, contig_dict, "{0}/tmp/post_contigs.fasta".format(TEMP_DIR))
    full_link = api.best_matching(graph5, simp_node_dict5, simp_edge_dict5, contig_dict, pe_info, logger)
    api.increment_nt_branch_coverage(graph5, simp_node_dict5, logger)
    write_stage_gfa(graph5, simp_node_dict5, simp_edge_dict5, "{0}/gfa/split_graph_final.gfa".format(TEMP_DIR))
    p_delta = 0.05 * numpy.median([graph5.vdp[v] for v in range(graph5.num_vertices())])
    strain_dict, usages = api.path_extension(graph5, simp_node_dict5, simp_edge_dict5, contig_dict, full_link,
                                             dcpy_pe_info, logger, p_delta, TEMP_DIR)
    assert isinstance(usages, dict)
    assert pe_info[(list(simp_node_dict1)[0], list(simp_node_dict1)[0])] >= 0  # (the live view answers like the dict)
    resolve_contigs(strain_dict)
    graphl2, simp_node_dictl2, _ = read_stage_gfa("{0}/gfa/es_graph_L2.gfa".format(TEMP_DIR))
    trim_contigs(graphl2, simp_node_dictl2, strain_dict, logger)
    drop_duplicate_contigs(strain_dict, logger)
    write_contig_paths(strain_dict, "{0}/tmp/tmp_strain.paths".format(TEMP_DIR), None, False)
    restore_repeats(pre.g0, pre.nodes0, strain_dict, pre.contig_info, pre.original_contigs, logger)
    write_contig_fasta(pre.g0, pre.nodes0, strain_dict, "{0}/strain.fasta".format(TEMP_DIR))
    write_contig_paths(strain_dict, "{0}/strain.paths".format(TEMP_DIR), pre.idx_mapping, True)


@pytest.mark.parametrize("name", ["two_strain_bubbles_k21", "hiv_like_k55", "ten_strain_k31", "three_strain_scrambled_k21"])
def test_reference_shaped_driver_reproduces_the_golden_run(name, tmp_path):
    case = Case(name)
    api.set_backend(CheckerBackend())
    try:
        inp = case.inputs(str(tmp_path))
        out = str(tmp_path / "out")
        reference_shaped_driver(case.args(inp, out), quiet_logger(), case)
    finally:
        api.set_backend(None)
    problems, _ = compare(case, out)
    binding = case.binding(problems)
    assert not binding, binding


def test_ref_file_debug_mode_is_refused():
    api.set_backend(CheckerBackend())
    try:
        with pytest.raises(NotImplementedError):
            api.iter_graph_disentanglement(None, {}, {}, {}, None, "ref.fa", quiet_logger(), 1.0, "/tmp")
    finally:
        api.set_backend(None)
