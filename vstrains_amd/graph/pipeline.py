"""The stage sequence of ``utils/VStrains_SPAdes.py:25-280`` (``run``): same order, same
intermediate files under ``OUT/{gfa,tmp,aln}``, same final ``strain.fasta`` / ``strain.paths``.

``backend`` supplies the device side: PE-link inference (boundary 1 of SURVEY.md 8b, here an
in-process call instead of a ``python VStrains_PE_Inference.py`` subprocess) and the native stage
handle the graph stages run on (``native_stage.NativeStage``: the library's C++ engine over the HIP
kernels).  The default backend is the HIP one and fails loudly without a GPU; tests inject the
checker from ``oracle/``.
"""
from __future__ import annotations

import gc
import sys
import time
from typing import Dict

import numpy

from . import prep
from .contigs import restore_repeats
from .formats import read_stage_gfa, write_contig_fasta, write_contig_paths, write_stage_gfa  # noqa: F401


def _stored(logger, filename: str) -> None:
    logger.info(filename + " is stored..")


class Prepared:
    """Everything the stages after PE-link inference need (state at VStrains_SPAdes.py:117)."""

    def __init__(self, g0, nodes0, idx_mapping, contigs, contig_info, original_contigs, g1, nodes1, edges1, ksize):
        self.g0, self.nodes0, self.idx_mapping = g0, nodes0, idx_mapping
        self.contigs, self.contig_info, self.original_contigs = contigs, contig_info, original_contigs
        self.g1, self.nodes1, self.edges1, self.ksize = g1, nodes1, edges1, ksize


def prepare(args, logger) -> Prepared:
    """VStrains_SPAdes.py:30-116: parse, canonise strands, reindex, coverage cut-off, contigs,
    simplification; leaves ``gfa/s_graph_L1.gfa`` for PE-link inference."""
    out = args.output_dir
    logger.info(">>>STAGE: parsing graph and contigs")
    g, nodes, edges = prep.load_assembly_graph(args.gfa_file, logger)
    write_stage_gfa(g, nodes, edges, "{0}/gfa/graph_L0.gfa".format(out))
    _stored(logger, "{0}/gfa/graph_L0.gfa".format(out))
    g0, nodes0, edges0 = read_stage_gfa("{0}/gfa/graph_L0.gfa".format(out))
    g0, nodes0, edges0, idx_mapping = prep.reindexing(g0, nodes0, edges0)
    write_stage_gfa(g0, nodes0, edges0, "{0}/gfa/graph_L0r.gfa".format(out))
    _stored(logger, "{0}/gfa/graph_L0r.gfa".format(out))

    if args.min_cov is not None:
        threshold = args.min_cov
        logger.info("user-defined node minimum coverage: {0}".format(threshold))
    else:
        threshold = prep.threshold_estimation(g0, logger)
        logger.info("computed node minimum coverage: {0}".format(threshold))

    contigs, contig_info = prep.spades_paths_parser(g0, nodes0, edges0, idx_mapping, logger, args.path_file,
                                                    args.min_len, threshold)
    original_contigs = {cno: [list(ids), clen, ccov] for cno, (ids, clen, ccov) in contigs.items()}
    write_contig_paths(contigs, "{0}/tmp/init_contigs.paths".format(out))
    write_contig_fasta(g0, nodes0, contigs, "{0}/tmp/init_contigs.fasta".format(out))

    logger.info(">>>STAGE: preprocess")
    prep.graph_simplification(g0, nodes0, edges0, None, logger, threshold)
    write_stage_gfa(g0, nodes0, edges0, "{0}/gfa/s_graph_L1.gfa".format(out))
    _stored(logger, "{0}/gfa/s_graph_L1.gfa".format(out))
    g1, nodes1, edges1 = read_stage_gfa("{0}/gfa/s_graph_L1.gfa".format(out))
    for cno, (ids, _, _) in list(contigs.items()):
        if any([c not in nodes1 for c in ids]):
            contigs.pop(cno)
            logger.debug("unreliable contig with low coverage: {0}".format(cno))

    ksize = g1.eovl[next(iter(g1.edges()))] if g1.num_edges() > 0 else 0
    logger.info("graph kmer size: {0}".format(ksize))
    if ksize <= 0:
        logger.error("invalid kmer-size, the graph does not contain any edges, exit..")
        sys.exit(1)
    return Prepared(g0, nodes0, idx_mapping, contigs, contig_info, original_contigs, g1, nodes1, edges1, ksize)


def extract_strains(pre: Prepared, table, backend, logger, out: str):
    """VStrains_SPAdes.py:140-272: edge cleaning, disentanglement, path extraction, final files.
    ``table``: the PE-link table (``ops.PeLinks``) over the nodes of ``s_graph_L1``.  This is the
    "end-to-end strain-extract" leg of BASELINE.json's metric.

    The interpreter's cycle collector is paused for the duration (the prepared graph and the records that cross the
    boundary are cycle-free containers by the ten thousand) and put back as it was afterwards."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        return _extract_strains(pre, table, backend, logger, out)
    finally:
        if was_enabled:
            gc.enable()


def _extract_native(pre: Prepared, table, backend, logger, out: str):
    """The stages of VStrains_SPAdes.py:140-248 on the native stage handle (``native_stage.NativeStage``): the graph, the
    contigs and the link bookkeeping are loaded into the library once, every stage is one call, the strain records come
    back at the end.  Same files, same log lines, same order as the calls below state."""
    import logging

    st = backend.native_stage(table)
    st.set_debug(logger.isEnabledFor(logging.DEBUG))
    marks = [("start", time.perf_counter())]
    st.load_graph(pre.g1, pre.nodes1, pre.edges1)
    st.load_contigs(pre.contigs)
    marks.append(("load_s", time.perf_counter()))
    st.edge_cleaning(logger)
    st.reinit("{0}/gfa/es_graph_L2.gfa".format(out), logger)
    st.keep_graph()  # (the final strain records are measured on es_graph_L2, VStrains_SPAdes.py:253-258)
    st.write_contigs("{0}/tmp/pre_contigs.paths".format(out), "{0}/tmp/pre_contigs.fasta".format(out))
    marks.append(("edge_cleaning_s", time.perf_counter()))

    delta = 0.05 * st.median_depth()
    st.disentangle(delta, out, logger)
    st.write_contigs("{0}/tmp/post_contigs.paths".format(out), "{0}/tmp/post_contigs.fasta".format(out))
    marks.append(("disentanglement_s", time.perf_counter()))

    logger.info(">>>STAGE: contig path extension")
    st.best_matching(logger)
    st.increment_nt_branch_coverage(logger)
    st.write_gfa("{0}/gfa/split_graph_final.gfa".format(out), logger)
    marks.append(("best_matching_s", time.perf_counter()))
    p_delta = 0.05 * st.median_depth()
    st.path_extension(p_delta, out, logger)
    marks.append(("path_extension_s", time.perf_counter()))
    logger.info(">>>STAGE: final process")
    st.finish_strains("{0}/tmp/tmp_strain.paths".format(out), logger)
    marks.append(("finish_strains_s", time.perf_counter()))
    strains = st.strains()
    st.contigs_into(pre.contigs)  # (consumed in place, as the reference's contig_dict is)
    marks.append(("export_s", time.perf_counter()))
    extract_strains.last_stages = {name: t - marks[i][1] for i, (name, t) in enumerate(marks[1:])}
    extract_strains.last_stages.update(st.counters())
    extract_strains.last_stages["engine"] = "native stage handle (vs_stage)"
    extract_strains.last_stages["sections"] = {k: round(v, 4) for k, v in st.sections().items()}
    st.close()
    return strains


def _extract_strains(pre: Prepared, table, backend, logger, out: str):
    # Both ways end with the strain records of VStrains_SPAdes.py:262 (resolved, trimmed on es_graph_L2, duplicates
    # dropped, tmp/tmp_strain.paths written).  A test backend may bring its own statement of the stages
    # (oracle/graph_stages, the checker the native engine is compared with) through ``extract_stages``; the product
    # backend offers only the native stage handle.
    if hasattr(backend, "extract_stages"):
        strains = backend.extract_stages(pre, table, logger, out)
    else:
        strains = _extract_native(pre, table, backend, logger, out)
    t0 = time.perf_counter()
    restore_repeats(pre.g0, pre.nodes0, strains, pre.contig_info, pre.original_contigs, logger)
    logger.info(">>>STAGE: generate result")
    write_contig_fasta(pre.g0, pre.nodes0, strains, "{0}/strain.fasta".format(out))
    write_contig_paths(strains, "{0}/strain.paths".format(out), pre.idx_mapping, True)
    if isinstance(getattr(extract_strains, "last_stages", None), dict):
        extract_strains.last_stages["final_files_s"] = time.perf_counter() - t0
    return strains


def run(args, logger, backend=None):
    """``VStrains_SPAdes.run``: preparation, PE inference, strain extraction.  (The cycle collector is
    paused for the whole command, see ``extract_strains``: the preparation builds the same kind of
    cycle-free containers.)"""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        return _run(args, logger, backend)
    finally:
        if was_enabled:
            gc.enable()


def _run(args, logger, backend=None):
    if backend is None:
        from .hip_ops import HipBackend

        backend = HipBackend(getattr(args, "device", 0))
    out = args.output_dir
    timings: Dict[str, float] = {}
    t_all = time.time()
    logger.info("VStrains-SPAdes started")
    pre = prepare(args, logger)

    t0 = time.time()
    table = backend.pe_links("{0}/gfa/s_graph_L1.gfa".format(out), "{0}/aln".format(out), args.fwd, args.rve,
                             pre.ksize, list(pre.nodes1.keys()))
    timings["pe_inference_s"] = time.time() - t0
    logger.info("paired end information stored")

    t0 = time.time()
    extract_strains(pre, table, backend, logger, out)
    timings["strain_extract_s"] = time.time() - t0
    logger.info("VStrains-SPAdes finished")
    timings["total_s"] = time.time() - t_all
    return timings
