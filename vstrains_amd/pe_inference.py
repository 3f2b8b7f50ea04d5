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


def _rank_world():
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def count_links(ctx, gfa: str, fwd: str, rve: str, kmer_size: int, stages_follow: bool = False):
    """Index the nodes of ``gfa`` and count every pair of the two FASTQ files; the counters stay
    on the device.  Returns ``(node ids in file order, PeCounter)``.  ``stages_follow``: the graph stages will build their
    PE-link table from these counters in this process (the pipeline; not the stand-alone script, which writes the two text
    files) -- the table's device buffer is then set aside together with the counters."""
    ids, seqs = host.read_gfa_segments(gfa)  # :100-112
    ctx.build_index(seqs, kmer_size)  # :114-135  (KeyError on a bad node base, as :13)
    counter = host.PeCounter(ctx)

    print("Start aligning reads to gfa nodes")  # :146
    # one process per GPU (torchrun): this rank counts its contiguous block of the pairs and the
    # counters are summed over ranks afterwards (RCCL all-reduce); a single process takes everything
    rank, world = _rank_world()
    if rank == 0 and stages_follow:
        counter.reserve_link_table()  # (the graph stages run on this rank: their table's buffer is taken now)
    if world > 1:
        # nobody reads a whole file: every rank counts the lines of its byte range, the ranks exchange the counts and
        # each indexes only the bytes of its own records (:154's total follows from the counts)
        fq = host.FastqPair.open_shard(fwd, rve, ctx, rank, world)
        count_fastq(ctx, fq, counter, fq.block_offset, fq.block_offset + len(fq), progress=(rank == 0))
    else:
        fq = host.FastqPair(fwd, rve, ctx)  # :146-154, native multi-threaded ingest
        count_fastq(ctx, fq, counter, 0, len(fq), progress=True)
    fq.close()
    if world > 1:
        counter.all_reduce(dst=0)  # (only rank 0 writes the files and runs the stages: the sums are reduced to it)
    return ids, counter


def count_fastq(ctx, fq, counter, first: int, last: int, batch: int = BATCH_PAIRS, progress: bool = False):
    """Pairs [first, last) of an indexed FASTQ pair through the counters, one block of ``batch``
    pairs at a time.  The host cores pack block i+1 (``vs_fastq_block``: 2 bits per base into
    pinned staging, upload enqueued) while the device still counts block i."""
    spans = [(lo, min(last, lo + batch)) for lo in range(first, last, batch)]
    # the uploads of a block are enqueued on the context's stream: make that the stream the counting runs on
    # BEFORE the first block is made (PeCounter.add sets the same stream again for every block)
    import torch

    with torch.cuda.device(counter.device):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    nxt = fq.block(spans[0][0], spans[0][1] - spans[0][0]) if spans else None
    for i, (lo, hi) in enumerate(spans):
        if progress:
            for mark in range(((lo + 99999) // 100000) * 100000, hi, 100000):
                print("Number of processed reads: ", mark)  # :156-157 (rank 0's own block under torchrun)
        block = nxt
        counter.add(block)  # enqueued behind the block's uploads
        nxt = fq.block(spans[i + 1][0], spans[i + 1][1] - spans[i + 1][0]) if i + 1 < len(spans) else None
        ctx.sync()
        block.free()


def write_info_files(out_dir: str, ids, counter):
    """pe_info / st_info text, PE_Inference.py:190-207.  Returns the host copies and stats."""
    node_mat, short_mat, stats = counter.result()
    out_file = "{0}/pe_info".format(out_dir)
    out_file2 = "{0}/st_info".format(out_dir)
    host.write_matrix_text(out_file, ids, node_mat)
    host.write_matrix_text(out_file2, ids, short_mat)
    return out_file, stats


def run(gfa: str, out_dir: str, fwd: str, rve: str, kmer_size: int, device: int = 0, ctx=None, stages_follow: bool = False):
    # PE_Inference.py:93-96: the output directory is wiped and recreated
    if out_dir[-1] == "/":
        out_dir = out_dir[:-1]
    rank, world = _rank_world()
    if rank == 0:  # under torchrun only the writer touches the directory
        shutil.rmtree(out_dir, ignore_errors=True)
        os.makedirs(out_dir, exist_ok=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()  # nobody counts (or could fail half-way) before the directory is in its final state

    glb_start = time.time()
    if ctx is None:
        ctx = host.Context(device)
    ids, counter = count_links(ctx, gfa, fwd, rve, kmer_size, stages_follow=stages_follow)
    run.last = (ids, counter)
    if rank != 0:
        return None  # the counters were reduced to rank 0, which writes the files
    out_file, stats = write_info_files(out_dir, ids, counter)
    glb_elapsed = time.time() - glb_start
    print("Global time elapsed: ", glb_elapsed)  # :209-211
    print("result stored in: ", out_file)
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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:  # launched by torchrun: one rank per GPU, counters all-reduced over RCCL
        import torch
        import torch.distributed as dist

        local = int(os.environ.get("LOCAL_RANK", "0"))
        # (VS_DIST_BACKEND=gloo VS_DIST_DEVICE=0: several ranks on ONE GPU, counters summed through gloo --
        # how the sharded drop-in is exercised end to end on a one-GPU box; RCCL wants one device per rank)
        backend = os.environ.get("VS_DIST_BACKEND", "nccl")
        local = int(os.environ.get("VS_DIST_DEVICE", str(local)))
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        args.device = local
        # every rank indexes both FASTQ files on the host: share the cores between the local ranks
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        os.environ.setdefault("VS_HOST_THREADS", str(max(1, (os.cpu_count() or 1) // max(local_world, 1))))
    run(args.gfa, args.dir, args.fwd, args.rve, args.kmer_size, args.device)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
    sys.exit(0)
