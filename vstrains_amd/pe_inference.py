#!/usr/bin/env python3
"""Drop-in for the reference's ``utils/VStrains_PE_Inference.py`` (same argv, same files, same
stdout lines; reference lines cited inline), running on MI355X through libvstrains_hip.so.

    python -m vstrains_amd.pe_inference -g s_graph_L1.gfa -o OUT/aln -f fwd.fq -r rve.fq -k 55
"""
import argparse
import os
import shutil
import sys
import time

from . import pe as host

BATCH_PAIRS = 1 << 20


def count_links(ctx, gfa: str, fwd: str, rve: str, kmer_size: int):
    """Index the nodes of ``gfa`` and count every pair of the two FASTQ files; the counters stay
    on the device.  Returns ``(node ids in file order, PeCounter)``."""
    ids, seqs = host.read_gfa_segments(gfa)  # :100-112
    ctx.build_index(seqs, kmer_size)  # :114-135  (KeyError on a bad node base, as :13)
    counter = host.PeCounter(ctx)

    print("Start aligning reads to gfa nodes")  # :146
    fq = host.FastqPair(fwd, rve, ctx)  # :146-154, native multi-threaded ingest
    total = len(fq)
    for lo in range(0, total, BATCH_PAIRS):
        hi = min(total, lo + BATCH_PAIRS)
        for mark in range(((lo + 99999) // 100000) * 100000, hi, 100000):
            print("Number of processed reads: ", mark)  # :156-157
        block = fq.block(lo, hi - lo)
        counter.add(block)
        ctx.sync()
        block.free()
    fq.close()
    return ids, counter


def write_info_files(out_dir: str, ids, counter):
    """pe_info / st_info text, PE_Inference.py:190-207.  Returns the host copies and stats."""
    node_mat, short_mat, stats = counter.result()
    out_file = "{0}/pe_info".format(out_dir)
    out_file2 = "{0}/st_info".format(out_dir)
    host.write_matrix_text(out_file, ids, node_mat)
    host.write_matrix_text(out_file2, ids, short_mat)
    return out_file, stats


def run(gfa: str, out_dir: str, fwd: str, rve: str, kmer_size: int, device: int = 0, ctx=None):
    # PE_Inference.py:93-96: the output directory is wiped and recreated
    if out_dir[-1] == "/":
        out_dir = out_dir[:-1]
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir, exist_ok=True)

    glb_start = time.time()
    if ctx is None:
        ctx = host.Context(device)
    ids, counter = count_links(ctx, gfa, fwd, rve, kmer_size)
    out_file, stats = write_info_files(out_dir, ids, counter)
    glb_elapsed = time.time() - glb_start
    print("Global time elapsed: ", glb_elapsed)  # :209-211
    print("result stored in: ", out_file)
    run.last = (ids, counter)
    return stats


def main(argv=None):
    print("----------------------Paired-End Information Alignment----------------------")  # :52-54
    parser = argparse.ArgumentParser(
        prog="pe_info", description="""Align Paired-End reads to nodes in graph to obtain strong links""")
    parser.add_argument("-g", "--gfa,", dest="gfa", type=str, required=True, help="graph, .gfa format")
    parser.add_argument("-o", "--output_dir", dest="dir", type=str, required=True, help="output directory")
    parser.add_argument("-f", "--forward", dest="fwd", required=True, help="forward read, .fastq")
    parser.add_argument("-r", "--reverse", dest="rve", required=True, help="reverse read, .fastq")
    parser.add_argument("-k", "--kmer_size", dest="kmer_size", type=int, default=128, help="unique kmer size")
    parser.add_argument("--device", dest="device", type=int, default=0, help="HIP device ordinal (extension)")
    args = parser.parse_args(argv)
    run(args.gfa, args.dir, args.fwd, args.rve, args.kmer_size, args.device)


if __name__ == "__main__":
    main()
    sys.exit(0)
