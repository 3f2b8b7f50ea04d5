"""CHECKER (test infrastructure): the stage sequence of utils/VStrains_SPAdes.py:140-248 over the Python restatement
of the stages, with the device operations supplied by ``ops`` (tests pass ``oracle.graph_ops.NumpyGraphOps``; the GPU
suite also passes ``hip_ops.HipGraphOps`` to check the kernels under the same decisions).  A test backend hands this to
``pipeline.run`` through its ``extract_stages`` hook."""
from __future__ import annotations

import numpy

from .model import (adopt, drop_duplicate_contigs, read_stage_gfa, resolve_contigs, trim_contigs, write_contig_fasta, write_contig_paths,
                    write_stage_gfa)

from . import disentangle as dis
from . import extend as ext


def extract_stages(pre, table, ops, links, logger, out: str):
    """-> strain_dict as of VStrains_SPAdes.py:262 (resolved, trimmed on es_graph_L2, duplicates dropped).  ``links``: a ``LiveLinks`` / ``DictLiveLinks`` over ``table``."""
    contigs = pre.contigs
    # the prepared graph is taken over as DATA into the checker's own container (model.adopt): adjacency order, GFA text and
    # contig bookkeeping from here on are this package's statement of the reference, not the product's
    g1, nodes1, edges1 = adopt(pre.g1, pre.nodes1, pre.edges1)
    stage1 = dis.Stage(g1, nodes1, edges1)
    dis.edge_cleaning(g1, edges1, contigs, links, logger)
    stage2 = dis.reinit(stage1, ops, logger, "{0}/gfa/es_graph_L2.gfa".format(out))
    write_contig_paths(contigs, "{0}/tmp/pre_contigs.paths".format(out))
    write_contig_fasta(stage2.g, stage2.nodes, contigs, "{0}/tmp/pre_contigs.fasta".format(out))

    delta = 0.05 * numpy.median([stage2.g.vdp[v] for v in range(stage2.g.num_vertices())])
    stagef = dis.iter_graph_disentanglement(stage2, contigs, links, ops, logger, delta, out)
    write_contig_paths(contigs, "{0}/tmp/post_contigs.paths".format(out))
    write_contig_fasta(stagef.g, stagef.nodes, contigs, "{0}/tmp/post_contigs.fasta".format(out))

    logger.info(">>>STAGE: contig path extension")
    full_link = ext.best_matching(stagef, contigs, links, logger)
    ext.increment_nt_branch_coverage(stagef, logger)
    write_stage_gfa(stagef.g, stagef.nodes, stagef.edges, "{0}/gfa/split_graph_final.gfa".format(out))
    logger.info("{0}/gfa/split_graph_final.gfa is stored..".format(out))
    p_delta = 0.05 * numpy.median([stagef.g.vdp[v] for v in range(stagef.g.num_vertices())])
    strains, _ = ext.path_extension(stagef, contigs, full_link, table, ops, logger, p_delta, out)

    logger.info(">>>STAGE: final process")  # VStrains_SPAdes.py:251-262
    resolve_contigs(strains)
    gl, nodesl, _ = read_stage_gfa("{0}/gfa/es_graph_L2.gfa".format(out))
    trim_contigs(gl, nodesl, strains, logger)
    drop_duplicate_contigs(strains, logger)
    write_contig_paths(strains, "{0}/tmp/tmp_strain.paths".format(out), None, False)
    return strains


class PythonStages:
    """Mix-in for test backends: ``pipeline.extract_strains`` runs the Python restatement of the stages over
    ``self.graph_ops`` instead of the native stage handle."""

    def live_links(self, table):
        from .links import LiveLinks

        return LiveLinks(table)

    def extract_stages(self, pre, table, logger, out: str):
        return extract_stages(pre, table, self.graph_ops, self.live_links(table), logger, out)
