#!/usr/bin/env python3
"""``vstrains``-compatible command line (reference ``vstrains:32-277``): same flags, same output
directory layout, same log lines; the work runs on MI355X through libvstrains_hip.so.

    python -m vstrains_amd.cli -a spades -g graph.gfa -p contigs.paths -o OUT -fwd f.fq -rve r.fq

Not restated: ``-r/--reference_fa`` (hidden debug mode that shells out to minimap2) is accepted
and refused with a message; the coverage histogram PNG is not drawn.
"""
import argparse
import logging
import os
import platform
import sys
import time
from datetime import date

import numpy

__version__ = "1.1.0"


def build_parser():
    p = argparse.ArgumentParser(
        prog="VStrains",
        description="""Construct full-length viral strains under de novo approach
        from contigs and assembly graph, currently supports SPAdes""")
    p.add_argument("-a", "--assembler", dest="assembler", type=str, required=True, choices=["spades"],
                   help="name of the assembler used. [spades]")
    p.add_argument("-g", "--graph", dest="gfa_file", type=str, required=True,
                   help="path to the assembly graph, (.gfa format)")
    p.add_argument("-p", "--path", dest="path_file", type=str, required=False,
                   help="contig file from SPAdes (.paths format), only required for SPAdes. e.g., contigs.paths")
    p.add_argument("-mc", "--minimum_coverage", dest="min_cov", default=None, type=int, help=argparse.SUPPRESS)
    p.add_argument("-ml", "--minimum_contig_length", dest="min_len", default=None, type=int, help=argparse.SUPPRESS)
    p.add_argument("-r", "--reference_fa", dest="ref_file", default=None, type=str, help=argparse.SUPPRESS)
    p.add_argument("-o", "--output_dir", dest="output_dir", default="acc/", type=str,
                   help="path to the output directory [default: acc/]")
    p.add_argument("-d", "--dev_mode", dest="dev", action="store_true", default=False, help=argparse.SUPPRESS)
    p.add_argument("-fwd", "--fwd_file", dest="fwd", required=True, default=None, type=str,
                   help="paired-end sequencing reads, forward strand (.fastq format)")
    p.add_argument("-rve", "--rve_file", dest="rve", required=True, default=None, type=str,
                   help="paired-end sequencing reads, reverse strand (.fastq format)")
    p.add_argument("--device", dest="device", default=0, type=int, help="HIP device ordinal (extension)")
    p.add_argument("--no-pe-text", dest="no_pe_text", action="store_true", default=False,
                   help="extension: do not write the N^2-line aln/pe_info and aln/st_info (the graph stages read "
                        "the counters from device memory either way)")
    return p


def _bail(*lines):
    for line in lines:
        print(line)
    print("\nExiting...\n")
    sys.exit(1)


def main(argv=None, backend=None):
    args = build_parser().parse_args(argv)
    if (not args.gfa_file) or (not os.path.exists(args.gfa_file)):
        _bail("\nPath to the assembly graph is required, (.gfa format)", "Please ensure the path is correct")
    args.assembler = args.assembler.lower()
    if (not args.path_file) or (not os.path.exists(args.path_file)):
        _bail("\nPath to Contig file from SPAdes (.paths format) is required for SPAdes assmbler option. e.g., contigs.paths")
    if args.min_len is not None:
        if args.min_len < 0:
            _bail("\nPlease make sure to provide the correct option (invalid value for min_len or min_cov).")
    else:
        args.min_len = 250
    if args.min_cov is not None and args.min_cov < 0:
        _bail("\nPlease make sure to provide the correct option (invalid value for min_len or min_cov).")
    if args.ref_file:
        _bail("\n-r/--reference_fa is the reference's minimap2 debug mode and is not part of this build")
    if args.output_dir[-1] == "/":
        args.output_dir = args.output_dir[:-1]

    os.makedirs(args.output_dir, exist_ok=True)
    try:
        for sub in ("/gfa/", "/tmp/", "/paf/", "/aln/"):
            os.makedirs(args.output_dir + sub)
    except OSError:
        print("\nCurrent output directory is not empty")
        print("Please empty/re-create the output directory: " + str(args.output_dir))
        _bail()
    if os.path.exists(args.output_dir + "/vstrains.log"):
        os.remove(args.output_dir + "/vstrains.log")

    logger = logging.getLogger("VStrains %s" % __version__)
    logger.setLevel(logging.DEBUG if args.dev else logging.INFO)
    console = logging.StreamHandler()
    console.setLevel(logging.INFO)
    console.setFormatter(logging.Formatter("%(message)s"))
    logger.addHandler(console)
    to_file = logging.FileHandler(args.output_dir + "/vstrains.log")
    to_file.setLevel(logging.DEBUG if args.dev else logging.INFO)
    to_file.setFormatter(logging.Formatter("%(message)s"))
    logger.addHandler(to_file)

    logger.info("Welcome to VStrains!")
    logger.info("VStrains is a strain-aware assembly tools, which constructs full-length ")
    logger.info("virus strain with aid from de Bruijn assembly graph and contigs.")
    logger.info("")
    logger.info("System information:")
    try:
        logger.info("  VStrains version: " + str(__version__).strip())
        logger.info("  Python version: " + ".".join(map(str, sys.version_info[0:3])))
        logger.info("  OS: " + platform.platform())
    except Exception:
        logger.info("  Problem occurred when getting system information")
    logger.info("")
    start = time.time()
    logger.info("Input arguments:")
    logger.info("Assembly type: " + args.assembler)
    logger.info("Assembly graph file: " + args.gfa_file)
    logger.info("Forward read file: " + args.fwd)
    logger.info("Reverse read file: " + args.rve)
    logger.info("Contig paths file: " + args.path_file)
    logger.info("Output directory: " + os.path.abspath(args.output_dir))
    if args.dev:
        logger.info("*DEBUG MODE is turned ON")
    logger.info("\n\n")
    logger.info("======= VStrains pipeline started. Log can be found here: " + os.path.abspath(args.output_dir)
                + "/vstrains.log\n")
    stamped = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    console.setFormatter(stamped)
    to_file.setFormatter(stamped)

    from .graph import pipeline

    if backend is None and args.no_pe_text:
        from .graph.hip_ops import HipBackend

        backend = HipBackend(args.device, write_info_text=False)

    old_err = numpy.seterr(all="raise")  # vstrains:25
    try:
        timings = pipeline.run(args, logger, backend)
    except BaseException:
        logger.removeHandler(to_file)
        logger.removeHandler(console)
        to_file.close()
        raise
    finally:
        numpy.seterr(**old_err)

    elapsed = time.time() - start
    console.setFormatter(logging.Formatter("%(message)s"))
    to_file.setFormatter(logging.Formatter("%(message)s"))
    logger.info("")
    logger.info("Thanks for using VStrains")
    logger.info("Result is stored in {0}/strain.fasta".format(os.path.abspath(args.output_dir)))
    logger.info("You can visualise the path stored in {0}/strain.paths via {0}/gfa/graph_L0.gfa".format(
        os.path.abspath(args.output_dir)))
    logger.info("Finished: {0}".format(date.today().strftime("%B %d, %Y")))
    logger.info("Elapsed time: {0}".format(elapsed))
    logger.info("Exiting...")
    logger.removeHandler(to_file)
    logger.removeHandler(console)
    to_file.close()
    return timings


if __name__ == "__main__":
    main()
    sys.exit(0)
