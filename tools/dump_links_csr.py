#!/usr/bin/env python3
"""GPU box: the symmetrised PE-link table (IO.py:598-627) of a BASELINE config's whole per-GPU read block as CSR rows of
its non-zero cells, in the GFA's node order, written under gpurun_out/ in parts of at most --part-mb megabytes (gpurun
brings back 64 MiB per call: name the parts a call should write with --parts a,b,...).  The build container then runs the
checker's Python statement of the stages (oracle/graph_stages) on it -- tools/extract_digests_from_csr.py -- and commits the
digests the `-m gpu` suite holds the device run to (tests/golden/extract_digests_config4.json).

    python tools/dump_links_csr.py --config 4 [--parts 0,1] [--part-mb 28]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--parts", default=None)
    ap.add_argument("--part-mb", type=float, default=28.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    args = ap.parse_args()
    import tempfile

    import torch

    from vstrains_amd import pe as host
    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[args.config]
    work = tempfile.mkdtemp(prefix="vs_dump_")
    st, pre, names, seqs, cum, logger, _ = workload_for(args.config, work)
    ctx = host.Context(0)
    ctx.build_index(seqs, cfg["k"])
    n_pairs = cfg["total_pairs"] // cfg["gpus"]
    reads = ctx.synth_pairs(st.genomes, cum, 20250000 + args.config, 0, n_pairs, cfg["read_len"], int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    counter = host.PeCounter(ctx)
    counter.add(reads)
    del reads
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import native_check

    n = counter.n
    row_ptr, col, val, csr_sha = native_check.links_csr_of_counter(counter)
    assert val.max() < 2 ** 32
    nnz = int(col.shape[0])
    # parts: whole rows, about part_mb of raw (col u32 + val u32) each
    per = int(args.part_mb * 1e6 / 8)
    cuts = [0]
    while cuts[-1] < n:
        target = int(row_ptr[cuts[-1]]) + per
        nxt = int(np.searchsorted(row_ptr, target, side="right")) - 1
        cuts.append(min(n, max(nxt, cuts[-1] + 1)))
    n_parts = len(cuts) - 1
    want = range(n_parts) if args.parts is None else [int(x) for x in args.parts.split(",") if int(x) < n_parts]
    os.makedirs(args.out, exist_ok=True)
    import hashlib

    meta = {"config": args.config, "nodes": int(n), "pairs": int(n_pairs), "nnz": nnz, "parts": n_parts, "row_cuts": cuts,
            "names_sha256": hashlib.sha256("\n".join(names).encode()).hexdigest(),
            "csr_sha256": csr_sha,
            "stats": [int(x) for x in counter.stats] if hasattr(counter, "stats") else None}
    with open(os.path.join(args.out, "links_c%d_meta.json" % args.config), "w") as fh:
        json.dump(meta, fh)
    for p in want:
        lo, hi = cuts[p], cuts[p + 1]
        a, b = int(row_ptr[lo]), int(row_ptr[hi])
        path = os.path.join(args.out, "links_c%d_part%d.npz" % (args.config, p))
        np.savez_compressed(path, row_lo=lo, row_hi=hi, row_ptr=(row_ptr[lo:hi + 1] - row_ptr[lo]).astype(np.uint32), col=col[a:b], val=val[a:b].astype(np.uint32))
        print("part", p, "rows", lo, hi, "cells", b - a, "bytes", os.path.getsize(path), flush=True)
    print(json.dumps({k: meta[k] for k in ("nodes", "pairs", "nnz", "parts", "csr_sha256")}))


if __name__ == "__main__":
    main()
