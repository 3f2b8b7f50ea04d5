#!/usr/bin/env python3
"""GPU-side experiment for the counter kernel: in locus order, how even is the work of the 64 pairs a
wavefront would take (one lane per pair), and how often does an end repeat the list of the previous
pair's end?   python tools/acc_probe.py [pairs] [config]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from vstrains_amd import _native as nat, pe as host  # noqa: E402
from vstrains_amd.workloads import CONFIGS, workload_for  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    cfg_i = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cfg = CONFIGS[cfg_i]
    st, pre, names, seqs, cum, logger, _ = workload_for(cfg_i, tempfile.mkdtemp())
    ctx = host.Context(0)
    ctx.build_index(seqs, cfg["k"])
    reads = ctx.synth_pairs(st.genomes, cum, 20250000 + cfg_i, 0, M, cfg["read_len"], int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    cap = 64
    n = 2 * M
    lists = np.zeros((n, cap), dtype=np.uint32)
    counts = np.zeros(n, dtype=np.uint32)
    nat.check(ctx._h, nat.lib().vs_pe_map_ends(ctx._h, reads._h, cap, lists.ctypes.data, counts.ctypes.data))
    print("list length: mean %.2f, quantiles 50/90/99/max %s, > 16: %.4f, > 32: %.5f" % (
        counts.mean(), np.percentile(counts, [50, 90, 99, 100]).tolist(), (counts > 16).mean(), (counts > 32).mean()))
    counts = np.minimum(counts, cap)
    col = np.arange(cap)[None, :]
    lists = np.where(col < counts[:, None], lists, 0xFFFFFFFF).astype(np.uint32)
    lists.sort(axis=1)
    fkey = lists[0::2, 0].astype(np.int64)
    order = np.argsort(fkey, kind="stable")
    nl = counts[0::2].astype(np.int64)[order]
    nr = counts[1::2].astype(np.int64)[order]
    inc = nl * nr + nl * (nl + 1) // 2 + nr * (nr + 1) // 2
    for width in (64, 16):
        m = (M // width) * width
        blk = inc[:m].reshape(-1, width)
        print("lane per pair, groups of %d: sum of max*width %d vs increments %d -> efficiency %.3f" % (
            width, int(blk.max(axis=1).sum()) * width, int(inc[:m].sum()), inc[:m].sum() / (blk.max(axis=1).sum() * width)))
    L = lists[0::2][order]
    Rr = lists[1::2][order]
    same_l = (L[1:] == L[:-1]).all(axis=1)
    same_r = (Rr[1:] == Rr[:-1]).all(axis=1)
    print("end repeats the previous pair's list: left %.3f right %.3f both %.3f" % (same_l.mean(), same_r.mean(), (same_l & same_r).mean()))
    # distinct end lists of the whole block (a global table of lists would turn short_mat into one weighted
    # expansion per distinct list)
    allv = np.ascontiguousarray(lists).view([("", lists.dtype)] * lists.shape[1]).ravel()
    uniq, cnt = np.unique(allv, return_counts=True)
    un = np.array([int((np.frombuffer(u.tobytes(), dtype=np.uint32) != 0xFFFFFFFF).sum()) for u in uniq[:: max(1, len(uniq) // 20000)]])
    print("ends %d, distinct lists %d (%.4f); mean length of a distinct list %.2f -> ~%.3g weighted short_mat increments instead of %.3g" % (
        n, len(uniq), len(uniq) / n, un.mean(), len(uniq) * (un * (un + 1) / 2).mean(), float((counts.astype(np.int64) * (counts + 1) // 2).sum())))
    top = np.sort(cnt)[::-1]
    print("most frequent lists hold %s ends; lists seen once: %d" % (top[:5].tolist(), int((cnt == 1).sum())))
    print("increments per pair %.1f (node_mat %.1f, short_mat %.1f)" % (inc.mean(), (nl * nr).mean(), (inc - nl * nr).mean()))
    # distinct (left list, right list) combinations: one weighted node_mat expansion each
    _, l_id = np.unique(np.ascontiguousarray(lists[0::2]).view([("", lists.dtype)] * cap).ravel(), return_inverse=True)
    _, r_id = np.unique(np.ascontiguousarray(lists[1::2]).view([("", lists.dtype)] * cap).ravel(), return_inverse=True)
    combo = l_id.astype(np.int64) * (int(r_id.max()) + 1) + r_id
    ucombo, first = np.unique(combo, return_index=True)
    cl = counts[0::2].astype(np.int64)[first]
    cr = counts[1::2].astype(np.int64)[first]
    print("pairs %d, distinct (left, right) combinations %d (%.4f) -> %.3g weighted node_mat increments instead of %.3g" % (
        M, len(ucombo), len(ucombo) / M, float((cl * cr).sum()), float((counts[0::2].astype(np.int64) * counts[1::2]).sum())))
    # the floor for global atomics: distinct node_mat cells of the block (from the distinct combinations)
    if (cl * cr).sum() < 4e8:
        Ls = lists[0::2][first]
        Rs = lists[1::2][first]
        N = len(seqs)
        cells = []
        for lo in range(0, len(first), 200000):
            a = Ls[lo:lo + 200000].astype(np.int64)[:, :, None]
            b = Rs[lo:lo + 200000].astype(np.int64)[:, None, :]
            ok = (a != 0xFFFFFFFF) & (b != 0xFFFFFFFF)
            cells.append(np.unique((a * N + b)[ok]))
        cells = np.unique(np.concatenate(cells))
        print("distinct node_mat cells touched by the block: %d" % len(cells))


if __name__ == "__main__":
    main()
