#!/usr/bin/env python3
"""GPU-side experiment: how many DISTINCT accepted-node lists do the read ends of one chunk of
locus-ordered pairs have?  (Would a counter kernel that expands each distinct list once, weighted,
save work?)   python tools/dup_probe.py [pairs]"""
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import bench  # noqa: E402
from vstrains_amd import _native as nat, pe as host  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    st, pre, names, seqs, cum, logger, _ = bench.workload(tempfile.mkdtemp())
    ctx = host.Context(0)
    ctx.build_index(seqs, 55)
    reads = ctx.synth_pairs(st.genomes, cum, 20250001, 0, M, 150, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    cap = 16
    n = 2 * M
    lists = np.zeros((n, cap), dtype=np.uint32)
    counts = np.zeros(n, dtype=np.uint32)
    nat.check(ctx._h, nat.lib().vs_pe_map_ends(ctx._h, reads._h, cap, lists.ctypes.data, counts.ctypes.data))
    counts = np.minimum(counts, cap)
    col = np.arange(cap)[None, :]
    lists = np.where(col < counts[:, None], lists, 0xFFFFFFFF).astype(np.uint32)
    lists.sort(axis=1)  # canonical order, padding last
    # locus order ~ first accepted node of the forward end (the device sorts by the first seed hit)
    fkey = lists[0::2, 0].astype(np.int64)
    order = np.argsort(fkey, kind="stable")
    # a 64-bit fingerprint per end list (only for counting distinct lists here)
    w = (np.arange(1, cap + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))[None, :]
    fp = ((lists.astype(np.uint64) + np.uint64(1)) * w).sum(axis=1, dtype=np.uint64)
    nn = counts.astype(np.int64)
    short_inc = nn * (nn + 1) // 2
    used = (counts[0::2] > 0) & (counts[1::2] > 0)
    for chunk in (64, 1024, 2048, 8192):
        tot_short = tot_short_d = 0
        tot_node = int((nn[0::2] * nn[1::2])[used].sum())
        ends = ends_d = 0
        for c0 in range(0, M, chunk):
            pr = order[c0:c0 + chunk]
            pr = pr[used[pr]]
            for side in (0, 1):
                e = 2 * pr + side
                f = fp[e]
                u, idx = np.unique(f, return_index=True)
                tot_short += int(short_inc[e].sum())
                tot_short_d += int(short_inc[e[idx]].sum())
                ends += len(e)
                ends_d += len(u)
        print("chunk %5d pairs: ends %d distinct lists %d (%.3f); short_mat increments %d -> %d; node_mat increments %d"
              % (chunk, ends, ends_d, ends_d / max(ends, 1), tot_short, tot_short_d, tot_node))


if __name__ == "__main__":
    main()
