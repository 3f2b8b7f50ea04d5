#!/usr/bin/env python3
"""Counting gate for an adaptive seed phase in k_pe_tiles (VERDICT r4 "Next" 2), on the CPU.

k_pe_tiles probes the read offsets 0, s, 2s, ... of every end (s = K - w + 1) and expands every posting of every seed that
hits.  Any phase phi in [0, s) of that grid is exact (every match of K bases and more holds one seed start of every residue
mod s).  This counts, on a prefix of a bench stream, how many postings each phase would expand:

    total(phi)            = sum over ends of postings(end, phi)
    best_of(m)            = sum over ends of min over phi in {0, s/m, 2s/m, ...} of postings(end, phi)
    best_of(all)          = sum over ends of min over every phi

postings(end, phi) = sum over offsets j = phi, phi + s, ... (j + w <= len) of the number of (node, position, strand)
occurrences of the read's w-mer at j among the node texts -- what the seed table's posting lists hold.

    python tools/phase_gate.py --config 2 --pairs 100000 [--out profiles/r5/phase_gate_config2.json]
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CODE = np.full(256, 255, dtype=np.uint8)
for i, c in enumerate(b"ACGT"):
    CODE[c] = i


def kmers(codes, w):
    """codes: uint8 [n] in 0..3 (255 = not ACGT) -> (forward keys uint64 [n - w + 1], rc keys, valid)"""
    n = codes.shape[-1] - w + 1
    bad = (codes > 3)
    c = np.where(bad, 0, codes).astype(np.uint64)
    f = np.zeros(codes.shape[:-1] + (n,), dtype=np.uint64)
    r = np.zeros_like(f)
    nb = np.zeros(codes.shape[:-1] + (n,), dtype=np.int32)
    for i in range(w):
        f = (f << np.uint64(2)) | c[..., i:i + n]
        r = r | ((np.uint64(3) - c[..., i:i + n]) << np.uint64(2 * i))
        nb += bad[..., i:i + n]
    return f, r, nb == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=100000)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from oracle import pe_oracle_c
    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[args.config]
    L, k = cfg["read_len"], cfg["k"]
    K = k + 1
    w = min(31, K)
    if w % 2 == 0:
        w -= 1
    if k >= 95:
        w = 63
    s = K - w + 1
    assert w <= 31, "this counter packs a seed into 62 bits (k < 95)"
    work = tempfile.mkdtemp(prefix="phase_gate_")
    st, pre, names, seqs, cum, logger, _ = workload_for(args.config, work)
    # the seed table's content: canonical w-mer -> number of occurrences over all nodes (a palindromic w-mer is found once
    # per strand by the kernel: two postings' worth; counted twice here too)
    keys = []
    for sq in seqs:
        if len(sq) < K:
            continue
        c = CODE[np.frombuffer(sq.encode(), dtype=np.uint8)]
        f, r, ok = kmers(c, w)
        keys.append(np.minimum(f, r)[ok])
        pal = (f == r) & ok
        if pal.any():
            keys.append(f[pal])
    keys = np.concatenate(keys)
    uniq, cnt = np.unique(keys, return_counts=True)
    seed = 20250000 + args.config
    fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, seed, 0, args.pairs, L, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    ends = np.concatenate([fw, rv])  # [2 * pairs, L]
    # pairs the reference drops (an N in either end) are not mapped at all
    has_n = ((fw == ord("N")).any(axis=1) | (rv == ord("N")).any(axis=1))
    keep = np.concatenate([~has_n, ~has_n])
    ends = ends[keep]
    f, r, ok = kmers(CODE[ends], w)
    can = np.minimum(f, r)
    at = np.searchsorted(uniq, can)
    at[at >= uniq.size] = 0
    hit = (uniq[at] == can) & ok
    per_off = np.where(hit, cnt[at], 0).astype(np.int64)  # [ends, L - w + 1]
    noff = per_off.shape[1]
    per_phase = np.zeros((per_off.shape[0], s), dtype=np.int64)
    probes = np.zeros(s, dtype=np.int64)
    for phi in range(s):
        per_phase[:, phi] = per_off[:, phi::s].sum(axis=1)
        probes[phi] = len(range(phi, noff, s))
    n_ends = per_off.shape[0]
    total = per_phase.sum(axis=0)
    res = {
        "config": args.config, "pairs_sampled": args.pairs, "pairs_mapped": int(n_ends // 2), "nodes": len(seqs), "K": K, "w": w, "stride": s,
        "seed_positions_in_table": int(keys.size), "distinct_seeds": int(uniq.size),
        "postings_per_end_by_phase": [round(float(t) / n_ends, 3) for t in total],
        "probes_per_end_by_phase": [int(p) for p in probes],
        "phase0_postings_per_end": round(float(total[0]) / n_ends, 3),
        "best_fixed_phase": int(np.argmin(total)), "best_fixed_phase_postings_per_end": round(float(total.min()) / n_ends, 3),
    }
    for m in (2, 4, 8, s):
        phis = sorted(set(int(round(i * s / m)) % s for i in range(m)))
        best = per_phase[:, phis].min(axis=1).sum()
        res["best_of_%s" % ("all" if m == s else m)] = {"phases": phis, "postings_per_end": round(float(best) / n_ends, 3),
                                                        "vs_phase0": round(float(best) / float(total[0]), 4)}
    # lower bound of any seed-and-extend scheme with this table: every accepted node costs one posting
    res["note"] = ("gate of VERDICT r4 #2: build the adaptive phase only if best_of_4 is >= 25 % below phase 0 "
                   "(vs_phase0 <= 0.75); probing m phases also costs m times the probes")
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(res, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    main()
