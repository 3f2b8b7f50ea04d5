"""GPU side: count one config's block, save the two PE matrices in sparse form and the digests of
everything the strain-extract leg writes (gpurun_out/links_c<i>.npz, extract_digest_c<i>.json), so
that the host logic of the graph stages can be profiled and checked on a box without a GPU
(tests/profile_extract_cpu.py)."""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digests(out_dir):
    res = {}
    for base, _, files in os.walk(out_dir):
        for f in files:
            p = os.path.join(base, f)
            rel = os.path.relpath(p, out_dir)
            if rel.endswith(".log"):
                continue
            with open(p, "rb") as fh:
                res[rel] = hashlib.sha256(fh.read()).hexdigest()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    args = ap.parse_args()
    import torch

    from vstrains_amd import pe as host
    from vstrains_amd.graph import pipeline
    from vstrains_amd.graph.hip_ops import HipBackend, HipPeLinks
    from vstrains_amd.workloads import CONFIGS, workload

    cfg = CONFIGS[args.config]
    L, k = cfg["read_len"], cfg["k"]
    R = cfg["total_pairs"] // cfg["gpus"]
    work_dir = tempfile.mkdtemp(prefix="vstrains_dump_")
    st, pre, names, seqs, cum, logger, _ = workload(
        work_dir, k=k, n_strains=cfg["n_strains"], genome_len=cfg["genome_len"], snp_rate=cfg["snp_rate"],
        seed=cfg["seed"], read_len=L, abundance_ratio=cfg["abundance_ratio"])
    torch.cuda.set_device(0)
    ctx = host.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.build_index(seqs, k)
    reads = ctx.synth_pairs(st.genomes, cum, 20250000 + args.config, 0, R, L, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    counter = host.PeCounter(ctx)
    counter.add(reads)
    node_mat, short_mat, stats = counter.result()
    os.makedirs(args.out, exist_ok=True)
    ni, nj = np.nonzero(node_mat)
    si, sj = np.nonzero(short_mat)
    np.savez_compressed(os.path.join(args.out, "links_c%d.npz" % args.config), n=len(names),
                        ni=ni.astype(np.int32), nj=nj.astype(np.int32), nv=node_mat[ni, nj].astype(np.int64),
                        si=si.astype(np.int32), sj=sj.astype(np.int32), sv=short_mat[si, sj].astype(np.int64))
    backend = HipBackend(ctx=ctx)
    times = []
    for rep in range(5):
        out_dir = os.path.join(work_dir, "out%d" % rep)
        os.makedirs(os.path.join(out_dir, "gfa"), exist_ok=True)
        os.makedirs(os.path.join(out_dir, "tmp"), exist_ok=True)
        import copy

        pre_i = copy.deepcopy(pre)
        t0 = time.perf_counter()
        table = HipPeLinks.from_counter(ctx, counter, names)
        strains = pipeline.extract_strains(pre_i, table, backend, logger, out_dir)
        times.append(time.perf_counter() - t0)
    from vstrains_amd import graph as graph_pkg

    with open(os.path.join(args.out, "extract_digest_c%d.json" % args.config), "w") as fh:
        json.dump({"files": digests(out_dir), "strains": len(strains), "seconds": times,
                   "host_modules": graph_pkg.host_modules()}, fh, indent=1, sort_keys=True)
    print("strains", len(strains), "seconds", times, graph_pkg.host_modules())


if __name__ == "__main__":
    main()
