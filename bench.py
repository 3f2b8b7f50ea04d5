#!/usr/bin/env python3
"""bench.py -- PE-link inference throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3,4}] [--pairs R]

``--config i`` = BASELINE.json configs[i] (default 2, the configuration the metric is quoted on):
synthetic strains of one ancestor genome, the compacted de Bruijn graph of those strains taken
through the pipeline's own preparation (``s_graph_L1``), and R read pairs per GPU sampled ON THE
DEVICE from the strains (0.5 % substitutions, 0.1 % N-pairs).  configs[3] / configs[4] are the
multi-GPU configurations: one GPU runs its shard of the stream (R / 4, R / 8 pairs).
One step = one pass of the hot path over that block with everything resident in HBM: zero the
counters, vs_pe_count over all pairs (locus sort, seed probe, extension, acceptance test,
node_mat/short_mat counters) and, for N > 1, the RCCL all-reduce of the counters.
Weak scaling: every rank works on its own R pairs (disjoint slices of one seeded stream).

After the timed PE steps rank 0 runs the graph stages once on the counters of the last step
(edge cleaning, disentanglement, path extraction: `strain_extract_s`, the second half of
BASELINE.json's metric; replicas only, no collective) -- configs 1 and 2 by default.

Prints ONE JSON line (rank 0).  `roofline` uses the algorithmic bytes per pair of SURVEY.md 8(d)
(2*ceil(L/4) + 2*(L-k)*8 + 16 = 1612 B at L=150,k=55; 2110 B at L=250,k=127) over the main
kernel's HIP-event time; `cpu_baseline` times the C restatement of the reference algorithm
(oracle/, 1 thread) on a prefix of the same read stream and checks that the GPU gives the same
counters on that prefix.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

from vstrains_amd.workloads import CONFIGS, workload  # noqa: E402  (generator parameters per BASELINE.json configs[i])


def pmc_traffic(config, pairs, kernel_name):
    """HBM bytes of one k_pe_tiles launch from the committed rocprofv3 PMC passes of THIS round
    (profiles/r6/pmc_summary_config<i>.json, collected with tools/profile.sh on the default workload
    of that config; counters cannot be read inside this process).  2 x FETCH_SIZE (the guide's gfx950
    rule for wide reads) + WRITE_SIZE, both in KB.  None unless the profile is of the same workload
    size and of the kernel instantiation this run launched."""
    path = os.path.join(ROOT, "profiles", "r6", "pmc_summary_config%d.json" % config)
    if not os.path.exists(path):
        return {"traffic": None}
    try:
        with open(path) as fh:
            summary = json.load(fh)
        if summary.get("_pairs_per_gpu") != pairs:
            return {"traffic": None}
        names = [name for name in summary if name.startswith("k_pe_tiles")]
        norm = lambda n: n.replace(" ", "").replace(",false>", ">")  # (rocprofv3 prints the defaulted template argument)
        if len(names) != 1 or (kernel_name and norm(names[0]) != norm(kernel_name)):
            return {"traffic": None, "traffic_note": "committed profile is of %s, this run launched %s" % (names, kernel_name)}
        k = summary[names[0]]
        fetch = k["FETCH_SIZE"]["per_dispatch_mean"] * 1024.0
        write = k["WRITE_SIZE"]["per_dispatch_mean"] * 1024.0
        return {"traffic": 2.0 * fetch + write, "traffic_unit": "B per k_pe_tiles launch",
                "traffic_source": "profiles/r6/pmc_summary_config%d.json (2*FETCH_SIZE + WRITE_SIZE of %s; separate rocprofv3 --pmc passes, not live)"
                                  % (config, names[0])}
    except Exception:
        return {"traffic": None}


def strain_recovery(genomes, fasta_path):
    """How much of the true strains the extracted ones are (outside the timed leg; VERDICT r4 "Next" 6): an extracted strain
    counts as exact when it is a substring of a true genome on either strand, a true genome as recovered when some
    extracted strain IS it."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    with open(fasta_path) as fh:
        outs = ["".join(rec.split("\n")[1:]) for rec in fh.read().split(">")[1:]]
    both = [(g, g.encode().translate(comp)[::-1].decode()) for g in genomes]
    exact = sum(1 for o in outs if any(o in g or o in r for g, r in both))
    whole = sum(1 for g, r in both if any(o == g or o == r for o in outs))
    longest = max([len(o) for o in outs] or [0])
    glen = max(len(g) for g in genomes)
    return {"true_strains": len(genomes), "genome_len": glen, "extracted": len(outs), "extracted_exact_substrings_of_a_true_strain": exact,
            "true_strains_recovered_whole": whole, "longest_extracted_bp": longest, "longest_over_genome": round(longest / glen, 4)}


def reference_stage_time(config):
    """The REAL reference's time for the same leg (VStrains_SPAdes.py:133-272: pe_info on disk -> strain_dict returned),
    measured in the build container as ONE interval by tools/time_reference_stages.py (the reference cannot travel to the
    GPU box): /root/reference/vstrains behind the graph-tool stand-in on this config's graph, 1 thread.  A committed
    record, not a measurement of this run."""
    path = os.path.join(ROOT, "profiles", "r6", "reference_stages_config%d.json" % config)
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except OSError:
        return None
    done = {k: v for k, v in rec["runs"].items() if v.get("returncode") == 0 and v.get("after_pe_to_run_return_s")}
    out = {"kind": "reference", "nodes": rec["nodes"], "pairs": rec["pairs"], "cores": 1, "where": "build container (8 vCPU), "
           "graph-tool / gfapy replaced by tests/golden/gt_standin (pure Python: slower than the C++ library)",
           "source": "profiles/r6/reference_stages_config%d.json" % config}
    if done:
        base = done.get("rotate/0") or done[sorted(done)[0]]
        out["reference_s"] = base["after_pe_to_run_return_s"]
        out["strains"] = base.get("strains")
        out["all_runs_s"] = {k: round(v["after_pe_to_run_return_s"], 2) for k, v in sorted(done.items())}
    else:
        out["reference_s"] = None
        out["progress"] = rec.get("progress")  # how far the reference got, and at which rate
    return out


def strain_extract(ctx, counter, pre, names, logger, out_dir, genomes=None):
    """The second half of the metric: pe counters (resident in HBM) -> strain.paths, i.e.
    VStrains_SPAdes.py:134-272 with the graph kernels on the device."""
    from vstrains_amd.graph import pipeline
    from vstrains_amd.graph.hip_ops import HipBackend, HipPeLinks

    backend = HipBackend(ctx=ctx)
    prof = None
    if os.environ.get("VS_PROFILE_EXTRACT"):
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    table = HipPeLinks.from_counter(ctx, counter, names)
    links_s = time.perf_counter() - t0
    strains = pipeline.extract_strains(pre, table, backend, logger, out_dir)
    secs = time.perf_counter() - t0
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(30)
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(40)
    n_stage_graphs = len([f for f in os.listdir(os.path.join(out_dir, "gfa")) if f.endswith(".gfa")])

    stages = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in getattr(pipeline.extract_strains, "last_stages", {}).items()}
    # share of the leg spent inside the library (the stage calls: host decisions in C++ + device operations) as opposed
    # to the Python around it (loading the prepared graph into the handle, the final strain records and files)
    in_library = sum(stages.get(k, 0.0) for k in ("edge_cleaning_s", "disentanglement_s", "best_matching_s", "path_extension_s"))
    res = {"seconds": secs, "strains": len(strains), "stage_graphs_written": n_stage_graphs,
           "link_table_from_counters_s": round(links_s, 4),
           "stages": stages, "library_share": round((in_library + links_s) / secs, 3) if secs > 0 else None,
           "longest_strain_bp": max([rec[1] for rec in strains.values()] or [0])}
    if genomes is not None:
        res["recovery"] = strain_recovery(genomes, os.path.join(out_dir, "strain.fasta"))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[i]")
    ap.add_argument("--pairs", type=int, default=0, help="read pairs per GPU (default: the config's total / its GPU count)")
    ap.add_argument("--read-len", type=int, default=0, help="read length instead of the config's (kernel corner measurements; not a bench line)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank counts the config's per-GPU block (the driver's contract); strong: the ranks split ONE block "
                         "(the N = 1 workload of the config) between them -- the question north_star asks at configs[2]")
    ap.add_argument("--dirty", type=float, default=0.0,
                    help="fraction of read BASES replaced by lower-case / IUPAC bytes (real FASTQ holds such bytes; 0 = none)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample time (0 = skip)")
    ap.add_argument("--no-extract", action="store_true", help="skip the strain-extract leg (kernel experiments)")
    ap.add_argument("--extract", action="store_true", help="run the strain-extract leg also for configs 3 / 4")
    ap.add_argument("--ingest-pairs", type=int, default=4_000_000,
                    help="pairs written as FASTQ text and timed through the native ingest (0 = skip)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver calls it: nothing has touched the GPU yet, so the N
        # ranks are started as a CHILD torchrun (never exec'd over this process) and rank 0's JSON line
        # and the return code are relayed.
        sys.exit(launch_ranks(args.gpus))
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")) and os.environ.get("VS_BENCH_FORCE_DIST") != "1":
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (VS_BENCH_FORCE_DIST=1 runs the multi-rank code path with a one-rank process group: the only
    # way to exercise it on a one-GPU box)
    use_dist = world > 1 or os.environ.get("VS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # (VS_DIST_BACKEND=gloo VS_DIST_DEVICE=0: all ranks on one GPU, sums through gloo -- a functional run of
        # the multi-rank step on a one-GPU box, tests only; its numbers mean nothing)
        backend = os.environ.get("VS_DIST_BACKEND", "nccl")
        local_rank = int(os.environ.get("VS_DIST_DEVICE", str(local_rank)))
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vstrains_amd import pe as host

    L, k = (args.read_len or cfg["read_len"]), cfg["k"]
    block = args.pairs if args.pairs > 0 else cfg["total_pairs"] // cfg["gpus"]  # what one GPU counts per step at N = 1
    from vstrains_amd import dist as vdist

    if args.scaling == "strong":
        lo, hi = vdist.strong_share(block, rank, world)  # contiguous slices of the one block
        R, first_pair = hi - lo, lo
    else:
        R, first_pair = block, rank * block
    seed = 20250000 + args.config
    sub_thresh = int(0.005 * 2 ** 32)
    n_thresh = int(0.001 * 2 ** 32)
    import tempfile

    work_dir = tempfile.mkdtemp(prefix="vstrains_bench_")
    t0 = time.time()
    st, pre, names, seqs, cum, logger, n_input_nodes = workload(
        work_dir, k=k, n_strains=cfg["n_strains"], genome_len=cfg["genome_len"], snp_rate=cfg["snp_rate"],
        seed=cfg["seed"], read_len=L, abundance_ratio=cfg["abundance_ratio"])
    workload_s = time.time() - t0

    class _G:  # the node set PE inference runs on (= s_graph_L1)
        pass

    g = _G()
    g.seqs = seqs
    ctx = host.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.time()
    ctx.build_index(g.seqs, k)
    ctx.sync()
    index_s = time.time() - t0
    reads = ctx.synth_pairs(st.genomes, cum, seed, first_pair, R, L, sub_thresh, n_thresh)
    if args.dirty > 0.0:
        reads = dirty_block(ctx, reads, args.dirty, seed)
    # One step = zero the counters, count the block, sum the counters over the ranks.  With more than one rank there are
    # two counter buffers and a side stream: the sum of step i (PeCounter.all_reduce: the ranks' occupied 64-cell
    # stretches through the RCCL ring, or the whole buffer when it is dense -- dist.sum_counts_compact) is issued on the
    # side stream right after the count of step i+1 was enqueued on the main stream, so the exchange -- and the host
    # waits it needs to learn sizes -- run while the next block is counted.  A buffer is counted into again only after its
    # sum has finished, and everything outstanding is drained inside the timed region.
    n_nodes = len(g.seqs)
    dense_bytes = 2 * n_nodes * n_nodes * 4
    counters = [host.PeCounter(ctx) for _ in range(2 if use_dist else 1)]
    if rank == 0 and (cfg["extract"] or args.extract) and not args.no_extract:
        counters[0].reserve_link_table()  # (as the pipeline does where it makes its counters: pe_inference.count_links)
    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream() if use_dist else None
    counted = [None for _ in counters]
    summed = [None for _ in counters]
    waiting = [None]  # the buffer whose block is counted and not yet summed
    step_no = [0]
    exchanges = []

    collectives = []

    def exchange(b):
        # (r6) two collectives, no host wait: the first exchange of the run learns the size of the ranks' union, the later
        # ones are staged for 1.25 times that (PeCounter.all_reduce(predict=True)); the flags and the real size are looked at
        # when the buffer comes up again (settle: one read of numbers the side stream finished long before)
        with torch.cuda.stream(side_stream):
            side_stream.wait_event(counted[b])
            for other in counters:  # (the two buffers share what is known about the union)
                if counters[b]._xstate_cap() is None and other._xstate_cap() is not None:
                    counters[b]._xstate_cap(other._xstate_cap())
            counters[b].all_reduce(predict=True)
            exchanges.append(counters[b].last_all_reduce)
            collectives.append(counters[b].last_collectives)
            summed[b] = side_stream.record_event()

    def settle(b):
        if use_dist and summed[b] is not None:
            with torch.cuda.stream(side_stream):
                counters[b].settle()
                summed[b] = side_stream.record_event()

    def step():
        b = step_no[0] % len(counters)
        step_no[0] += 1
        c = counters[b]
        settle(b)
        if summed[b] is not None:
            main_stream.wait_event(summed[b])
        c.reset()
        c.add(reads)
        if use_dist:
            counted[b] = main_stream.record_event()
            if waiting[0] is not None:
                exchange(waiting[0])
            waiting[0] = b

    def drain():
        if use_dist and waiting[0] is not None:
            exchange(waiting[0])
            waiting[0] = None
        for b in range(len(counters)):
            settle(b)
        for ev in summed:
            if ev is not None:
                main_stream.wait_event(ev)

    def barrier():
        if use_dist:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    kernel_ms = []
    slow_ms = []
    sort_ms = []
    acc_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events recorded by the library around the kernels on the stream they ran on; reading
        # them synchronises that stream, which the next step would do anyway (it reuses the buffers)
        t = ctx.last_timing()
        kernel_ms.append(t["main_ms"])
        slow_ms.append(t["slow_ms"])
        sort_ms.append(t["sort_ms"])
        acc_ms.append(t["accumulate_ms"])
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    counter = counters[(step_no[0] - 1) % len(counters)]  # the buffer of the last step
    if use_dist:
        import torch.distributed as dist

        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # (sums on the device: the dense matrices of configs[4] are 2 x 10 GB)
    stats = tuple(int(x) for x in counter.stats.cpu().tolist())
    node_sum = int(counter.mats[0].sum(dtype=torch.int64).item())
    short_sum = int(counter.mats[1].sum(dtype=torch.int64).item())

    rccl_ranks, coll_backend = 0, None
    if use_dist:
        import torch.distributed as dist

        coll_backend = dist.get_backend()
        rccl_ranks = dist.get_world_size() if coll_backend == "nccl" else 0
    extract_failed = False
    if rank == 0:
        b_alg = 2 * ((L + 3) // 4) + 2 * (L - k) * 8 + 16
        ms_step = elapsed / args.steps * 1e3
        job_pairs = block if args.scaling == "strong" else world * R  # pairs all ranks together count per step
        value = job_pairs * args.steps / elapsed
        avg_kernel_ms = float(np.mean(kernel_ms))
        achieved = R * b_alg / (avg_kernel_ms * 1e-3) / 1e9
        kernel_name = ctx.last_kernel
        out = {
            "metric": "PE read pairs/sec through PE-link inference (GFA index + packed reads in HBM -> node_mat/short_mat)",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d strains of a %.1f kb genome, %d x 2x%d bp pairs per GPU, %d-node GFA (s_graph_L1), k=%d%s"
                            % (cfg["tag"], cfg["n_strains"], cfg["genome_len"] / 1e3, R, L, len(g.seqs), k,
                               (", %.3g %% of read bases lower-case/IUPAC" % (100 * args.dirty)) if args.dirty else ""),
                "baseline_config": args.config,
                "pairs_per_gpu": R, "read_len": L, "k": k, "nodes": len(g.seqs),
                "node_bases": int(sum(len(s) for s in g.seqs)),
                "parallelism": ("read-block sharding x%d (%s: %d pairs per rank per step) + per-step sum of the [2,N,N] counters over the ranks, "
                                "overlapped with the next block: occupied 64-cell stretches of the union through the ring or the whole %.2f GB "
                                "buffer when it is dense (steps of this run: %s; collectives per exchange: %s)"
                                % (world, args.scaling, R, dense_bytes / 1e9, ", ".join("%s x%d" % (m, exchanges.count(m)) for m in sorted(set(exchanges))) or "none",
                                   ", ".join("%d x%d" % (m, collectives.count(m)) for m in sorted(set(collectives))) or "none"))
                               if use_dist else "one GPU, no exchange",
                "rccl_ranks": rccl_ranks, "collective_backend": coll_backend,
                "index": ctx.index_info, "index_build_s": index_s, "workload_build_s": workload_s,
                "node_numbering": "along the graph's paths (vs_node_order_host), results mapped back" if ctx.node_order is not None else "as given",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the whole step (sort + mapping + counters + overflow [+ exchange]) against the same roof
                "frac_step": R * b_alg / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None,
                "kernel": kernel_name, "kernel_ms_avg": avg_kernel_ms, "slow_kernel_ms_avg": float(np.mean(slow_ms)),
                "locus_sort_ms_avg": float(np.mean(sort_ms)),
                "accumulate_ms_avg": float(np.mean(acc_ms)),
                "algorithmic_bytes_per_pair": b_alg,
            },
            "pe_stats": {"n_reads": stats[0], "short_reads": stats[1], "used_reads": stats[2],
                         "node_mat_sum": node_sum, "short_mat_sum": short_sum,
                         "slow_pairs_per_step": ctx.last_timing()["slow_pairs"]},
        }
        out["config"]["input_gfa_nodes"] = n_input_nodes
        if use_dist and os.environ.get("VS_DIST_TIMING"):
            # measurement runs (tools/scaling_model.py): seconds per phase of dist.sum_counts_compact, mean over the steps'
            # exchanges of this rank; every phase is followed by a device synchronisation there, so the run is NOT a bench line
            tims = [t for c in counters for t in getattr(c, "exchange_timing", [])]
            keys = sorted({k for t in tims for k in t})
            out["exchange_timing"] = {k: float(np.mean([t[k] for t in tims if k in t])) for k in keys}
            out["exchange_timing"]["exchanges"] = len(tims)
        out["roofline"].update(pmc_traffic(args.config, R, kernel_name) if not args.dirty else {"traffic": None})
        out["roofline"].update(stream_copy(dev, achieved))
        want_extract = (cfg["extract"] or args.extract) and not args.no_extract
        if not want_extract:
            out["strain_extract_s"] = None
            out["strain_extract"] = {"skipped": "--no-extract" if args.no_extract else "configs 3/4 run it with --extract"}
        else:
            try:
                ex = strain_extract(ctx, counter, pre, names, logger, work_dir, genomes=st.genomes)
                out["strain_extract_s"] = ex.pop("seconds")
                out["strain_extract"] = ex
                ex["cpu_baseline"] = reference_stage_time(args.config)
            except Exception as err:  # the PE line is still printed, but the run FAILS (exit status 1)
                import traceback

                traceback.print_exc()
                out["strain_extract_s"] = None
                out["strain_extract"] = {"error": repr(err)}
                extract_failed = True
        if world == 1 and args.ingest_pairs > 0 and not args.no_extract:
            out["fastq_ingest"] = fastq_ingest(ctx, host, st, cum, seed, L, k, sub_thresh, n_thresh, min(args.ingest_pairs, R), work_dir)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(ctx, host, st, g, cum, seed, L, k, sub_thresh, n_thresh, R, args.cpu_seconds, args.config)
        print(json.dumps(out))
    if use_dist:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    if extract_failed:
        sys.exit(1)


def launch_ranks(n):
    """One process per GPU through torch.distributed.run, as a child of this (GPU-free) process."""
    import socket
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for attempt in range(3):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
        # (the port was free when asked for and may be taken a moment later: the rendezvous then fails before any rank
        # has started -- ask for another one)
        if proc.returncode != 0 and not lines and "address already in use" in proc.stderr and attempt < 2:
            continue
        break
    sys.stderr.write(proc.stderr)
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if proc.returncode == 0 and not lines:
        return 1
    return proc.returncode


def host_thread_budget():
    """Host threads this process may really use: the affinity mask cut by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as qf:
            a, b = qf.read().split()
            quota = None if a == "max" else float(a) / float(b)
    except Exception:
        pass
    budget = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return budget, n, quota


def stream_copy(dev, achieved_gbs):
    """what a plain device-to-device copy reaches on this box (1 GiB read + 1 GiB written per
    pass, HIP events on torch's stream), next to the nominal HBM peak the fraction is quoted against."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.int32, device=dev).fill_(1)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbs = reps * 2.0 * 4.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    return {"measured_stream_copy_GBs": gbs, "frac_of_measured_stream_copy": achieved_gbs / gbs}


def dirty_block(ctx, reads, frac, seed):
    """The same block with a fraction of its BASES replaced by bytes outside ACGTN (lower case,
    IUPAC codes): what real FASTQ files hold.  Such a byte makes every (k+1)-window over it miss
    (a dict lookup in the reference, PE_Inference.py:25-26) and leaves the pair in use."""
    text, lens, flags = reads.unpack()
    n = int(lens.size)
    rng = np.random.default_rng(seed)
    hits = rng.random(text.size) < frac
    alphabet = np.frombuffer(b"acgtnRYKMSW", dtype=np.uint8)
    text = text.copy()
    text[hits] = alphabet[rng.integers(0, alphabet.size, size=int(hits.sum()))]
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens, dtype=np.uint64)
    # the N-pairs of the stream stay N-pairs
    has_n = np.nonzero(flags & 1)[0]
    text[off[has_n].astype(np.int64)] = ord("N")
    reads.free()
    return ctx.pack(text, off)


def fastq_ingest(ctx, host, st, cum, seed, L, k, sub_thresh, n_thresh, M, work_dir):
    """PCIe-inclusive leg (never `value`): FASTQ text on disk -> native multi-threaded ingest
    (vs_fastq_open / vs_fastq_block) -> packed on the device -> counters.  The text is the device
    generator's own stream, unpacked (vs_synth_pairs + vs_reads_unpack); an N-pair gets its N back."""
    block = ctx.synth_pairs(st.genomes, cum, seed, 0, M, L, sub_thresh, n_thresh)
    text, lens, flags = block.unpack()
    block.free()
    text = text.reshape(M, 2, L).copy()
    text[(flags.reshape(M, 2) & 1).astype(bool), 0] = ord("N")
    paths = []
    digits = (np.arange(M, dtype=np.int64)[:, None] // 10 ** np.arange(8, -1, -1, dtype=np.int64)[None, :]) % 10
    for w, tag in enumerate(("f", "r")):
        # fixed-width records "@f000000123\n<seq>\n+\n<qual>\n", assembled as one byte matrix
        rec = np.empty((M, 2 + 9 + 1 + L + 3 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@")
        rec[:, 1] = ord(tag)
        rec[:, 2:11] = digits + ord("0")
        rec[:, 11] = ord("\n")
        rec[:, 12:12 + L] = text[:, w]
        rec[:, 12 + L:15 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 15 + L:15 + 2 * L] = ord("I")
        rec[:, 15 + 2 * L] = ord("\n")
        path = os.path.join(work_dir, "ingest_%s.fq" % tag)
        rec.tofile(path)
        paths.append(path)
    size = sum(os.path.getsize(p) for p in paths)
    from vstrains_amd import pe_inference

    counter = host.PeCounter(ctx)
    warm = host.FastqPair(paths[0], paths[1], ctx)  # (first use: pinned staging and device buffers get allocated)
    pe_inference.count_fastq(ctx, warm, counter, 0, min(len(warm), 2 * pe_inference.BATCH_PAIRS))
    warm.close()
    passes = []
    for _ in range(3):  # (host threads under a cgroup quota: single passes scatter by 2x; the median is reported)
        counter.reset()
        t0 = time.perf_counter()
        fq = host.FastqPair(paths[0], paths[1], ctx)
        t1 = time.perf_counter()
        pe_inference.count_fastq(ctx, fq, counter, 0, len(fq))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n = len(fq)
        fq.close()
        passes.append((t2 - t0, t1 - t0, t2 - t1))
    passes.sort()
    total_s, open_s, count_s = passes[1]
    budget, visible, quota = host_thread_budget()
    return {"pairs": n, "fastq_bytes": size, "host_threads": min(budget, 64) if not os.environ.get("VS_HOST_THREADS") else int(os.environ["VS_HOST_THREADS"]),
            "host_cpus_in_affinity_mask": visible, "cgroup_cpu_quota_cores": quota,
            "open_index_s": open_s, "pack_upload_count_s": count_s,
            "pairs_per_s": n / total_s, "pairs_per_s_after_open": n / count_s,
            "pairs_per_s_passes": [n / p[0] for p in passes],
            "note": "files in page cache; blocks of %d pairs, host packing of block i+1 overlapped with the device counting block i; "
                    "PCIe-inclusive; never reported as value" % pe_inference.BATCH_PAIRS}


def cpu_baseline(ctx, host, st, g, cum, seed, L, k, sub_thresh, n_thresh, R, target_s, config=2):
    """oracle/pe_oracle.c (1 thread) on the first M pairs of rank 0's stream; M sized from a
    short calibration so the whole leg stays near target_s.  Also the checker: the GPU counters
    for the same M pairs must be identical (compared cell by cell through the oracle's sparse
    increment list plus the matrix totals, so that 50k-node matrices never have to exist on the host)."""
    from oracle import pe_oracle_c

    t0 = time.perf_counter()
    orc = pe_oracle_c.Oracle(g.seqs, k)
    build_s = time.perf_counter() - t0

    def run(first, n):
        fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, seed, first, n, L, sub_thresh, n_thresh)
        off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
        t = time.perf_counter()
        res = orc.count_pairs_sparse(fw.reshape(-1), off, rv.reshape(-1), off, n)
        return res, time.perf_counter() - t

    _, cal = run(0, 20000)
    rate = 20000 / max(cal, 1e-6)
    M = int(min(R, max(20000, rate * target_s)))
    (node_cells, node_counts, short_cells, short_counts, ref_stats), secs = run(0, M)
    block = ctx.synth_pairs(st.genomes, cum, seed, 0, M, L, sub_thresh, n_thresh)
    chk = host.PeCounter(ctx)
    chk.add(block)
    ctx.sync()
    same = tuple(int(x) for x in chk.stats.cpu().tolist()) == tuple(int(x) for x in ref_stats)
    for mat, cells, counts in ((0, node_cells, node_counts), (1, short_cells, short_counts)):
        flat = chk.mats[mat].reshape(-1)
        # (the oracle numbers the nodes as the GFA does, the device as the index was built: Context.build_index)
        cells = ctx.internal_cells(mat, cells.astype(np.int64) // chk.n, cells.astype(np.int64) % chk.n)
        got = flat[torch.from_numpy(cells).to(flat.device)].cpu().numpy().view(np.uint32).astype(np.int64)
        same = same and bool(np.array_equal(got, counts)) and int(flat.sum(dtype=torch.int64).item()) == int(counts.sum())
    out = {
        "value": M / secs, "unit": "pairs/s", "cores": 1, "kind": "port",
        "sample": "first %d pairs of the same seeded stream (%.1f s; table build %.2f s not included)" % (M, secs, build_s),
        "gpu_matches_on_sample": bool(same),
    }
    # every host core (information only; the reference itself is single-threaded): a child process forks
    # one worker per core, each with its own table and its own slice of the stream
    try:
        import pickle
        import subprocess
        import tempfile

        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        with tempfile.NamedTemporaryFile(suffix=".pkl", delete=False) as fh:
            pickle.dump(dict(seqs=list(g.seqs), k=k, genomes=list(st.genomes), cum=np.asarray(cum), seed=seed, first_pair=R,
                             L=L, sub_thresh=sub_thresh, n_thresh=n_thresh), fh)
        proc = subprocess.run([sys.executable, "-m", "oracle.cpu_all_cores", fh.name, str(cores), str(min(target_s, 5.0))],
                              cwd=ROOT, capture_output=True, text=True, timeout=300)
        os.unlink(fh.name)
        allc = json.loads(proc.stdout.strip().splitlines()[-1])
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as qf:
                a, b = qf.read().split()
                quota = None if a == "max" else float(a) / float(b)
        except Exception:
            pass
        out["all_cores"] = {"value": allc["pairs_per_s"], "unit": "pairs/s", "cores": allc["workers"],
                            "cgroup_cpu_quota_cores": quota,
                            "sample": "%d pairs, %.1f s of counting per worker, %.1f s wall" % (allc["pairs"], allc["seconds"], allc["wall_s"])}
    except Exception as err:
        out["all_cores"] = {"error": repr(err)}
    # how this box's port relates to the reference script itself (which cannot travel here): measured
    # in the build container by tools/time_reference.py on the same workload
    try:
        with open(os.path.join(ROOT, "profiles", "r4", "cpu_reference.json")) as fh:
            ref = json.load(fh)["configs"].get("configs[%d]" % config)
        if ref:
            out["reference_ratio"] = ref["port_over_reference"]
            out["reference"] = {"pairs_per_s_build_container": ref["reference"]["pairs_per_s"],
                                "port_pairs_per_s_build_container": ref["port"]["pairs_per_s"],
                                "estimated_pairs_per_s_on_this_host": out["value"] / ref["port_over_reference"],
                                "source": "profiles/r4/cpu_reference.json: real VStrains_PE_Inference.py vs the port on the same "
                                          "%d pairs, same box, 1 core each" % ref["pairs"]}
    except Exception:
        pass
    return out


if __name__ == "__main__":
    main()
