"""Seeded bench / test workloads for the configurations BASELINE.json names: synthetic strains ->
assembler-style GFA + contigs.paths -> the pipeline's own preparation -> the node set PE inference
runs on (``s_graph_L1``).  Used by ``bench.py --config i`` and by the full-size GPU tests."""
import os

import numpy as np

# Generator parameters per BASELINE.json configs[i]: strains S, genome length G,
# SNP-site density, abundance decay, read length L, k, pairs of the whole job and the GPUs it is
# meant for (pairs per GPU = total / gpus).  Site density and decay are set so that ``s_graph_L1``
# has the node count BASELINE names; the 10.8 kb genome of SURVEY's config 3 tops out at 4.5 k
# nodes with 15 strains, so configs[2] uses 13.2 kb to reach the "5k-node GFA" of the metric.
CONFIGS = {
    0: dict(tag="configs[0]: 6-strain HIV-like quasispecies, 200-node GFA (the reference's own CPU-runnable case)", n_strains=6,
            genome_len=9700, snp_rate=0.007, abundance_ratio=0.8, read_len=150, k=55, total_pairs=100_000, gpus=1, seed=1001,
            extract=True),
    1: dict(tag="configs[1]: 5-strain HCV-like mix, 1k-node GFA", n_strains=5, genome_len=9600, snp_rate=0.055,
            abundance_ratio=0.8, read_len=150, k=55, total_pairs=1_000_000, gpus=1, seed=1002, extract=True),
    2: dict(tag="configs[2]: 15-strain ZIKV-like synthetic, 5k-node GFA", n_strains=15, genome_len=13200, snp_rate=0.085,
            abundance_ratio=0.8, read_len=150, k=55, total_pairs=10_000_000, gpus=1, seed=1003, extract=True),
    3: dict(tag="configs[3]: SARS-CoV-2-like lineage mix, 10k-node GFA, shard 1 of 4", n_strains=30, genome_len=29900,
            snp_rate=0.04, abundance_ratio=0.9, read_len=250, k=127, total_pairs=50_000_000, gpus=4, seed=1004, extract=False),
    4: dict(tag="configs[4]: 100-strain 30 kb synthetic, 50k-node GFA, shard 1 of 8", n_strains=100, genome_len=30000,
            snp_rate=0.076, abundance_ratio=0.97, read_len=150, k=55, total_pairs=200_000_000, gpus=8, seed=1005, extract=False),
    # not a BASELINE config: a bench-sized graph on which the reference's algorithm can follow strains for tens of kilobases
    # (three well-separated abundances, SNP sites mostly more than k apart), so that the extract leg's greedy walk
    # (contig_extension) is timed doing work -- at configs[2..4] the longest strain is 3 % .. 30 % of a genome
    5: dict(tag="extra (not in BASELINE.json): 3-strain 150 kb synthetic, 5.6k-node GFA, walk-heavy extraction", n_strains=3,
            genome_len=150000, snp_rate=0.02, abundance_ratio=0.45, read_len=150, k=55, total_pairs=10_000_000, gpus=1, seed=1010,
            extract=True),
}


def workload(out_dir, k=55, n_strains=15, genome_len=13200, snp_rate=0.085, seed=1003, read_len=150, abundance_ratio=0.8):
    """configs[2] inputs: synthetic strains -> assembler-style GFA + contigs.paths, taken through
    the pipeline's own preparation (strand canonisation, reindexing, coverage cut-off) so that PE
    inference runs on the real ``s_graph_L1`` and the graph stages can follow on the same state."""
    import argparse as ap
    import logging

    from vstrains_amd import synth
    from vstrains_amd.graph import pipeline

    pc = synth.make_pipeline_case(n_strains=n_strains, genome_len=genome_len, snp_rate=snp_rate, k=k, n_pairs=0,
                                  read_len=read_len, seed=seed, abundance_ratio=abundance_ratio)
    st = pc.strains
    for sub in ("gfa", "tmp", "paf", "aln"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    with open(os.path.join(out_dir, "input.gfa"), "w") as fh:
        fh.write(pc.gfa_text)
    with open(os.path.join(out_dir, "input.paths"), "w") as fh:
        fh.write(pc.paths_text)
    logger = logging.getLogger("vstrains-bench")
    logger.handlers[:] = [logging.NullHandler()]
    logger.propagate = False
    args = ap.Namespace(gfa_file=os.path.join(out_dir, "input.gfa"), path_file=os.path.join(out_dir, "input.paths"),
                        output_dir=out_dir, min_cov=None, min_len=250)
    pre = pipeline.prepare(args, logger)
    names = list(pre.nodes1.keys())
    seqs = [pre.g1.vseq[pre.nodes1[n]] for n in names]
    ab = np.array(st.abundance)
    cum = np.minimum(np.floor(np.cumsum(ab) / ab.sum() * 2 ** 32), 2 ** 32 - 1).astype(np.uint32)
    cum[-1] = 0xFFFFFFFF
    return st, pre, names, seqs, cum, logger, len(pc.graph.ids)




def workload_for(config: int, out_dir: str):
    """``workload`` with the generator parameters of BASELINE.json configs[config]."""
    c = CONFIGS[config]
    return workload(out_dir, k=c["k"], n_strains=c["n_strains"], genome_len=c["genome_len"], snp_rate=c["snp_rate"],
                    seed=c["seed"], read_len=c["read_len"], abundance_ratio=c["abundance_ratio"])
