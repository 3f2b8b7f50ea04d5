#!/usr/bin/env python3
"""GPU-side: randomized parity campaign for PE-link inference.  Draws graph / read shapes around the
boundaries where vs_pe_count switches kernels (seed geometry, straight-line vs long-window vs generic
comparison, compile-time tile shapes, dirty-byte lists vs mask, list overflow), counts on the device and
compares node_mat / short_mat / stats with the C oracle.  Prints the parameters of every failing draw.

    python tests/fuzz_pe.py [seconds=300] [seed=1]        (test infrastructure: the oracle is the checker)
"""
import os

os.environ.setdefault("VS_EXPERIMENT", "1")  # the draws flip tuning switches on a live context
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pe_oracle_c  # noqa: E402  (the checker)
from vstrains_amd import pe as host, synth  # noqa: E402

K_EDGE = [3, 5, 15, 21, 29, 30, 31, 32, 33, 54, 55, 56, 61, 62, 63, 84, 85, 86, 87, 94, 95, 96, 100, 125, 126, 127, 128, 150]
DIRTY = np.frombuffer(b"acgtnRYKMSWBDHV.-*", dtype=np.uint8)


def draw(rng):
    if os.environ.get("FUZZ_STD"):  # the compile-time tile shapes: k = 55, 2 x 97..159 bases; many strains -> long lists, overflow
        if rng.random() < 0.25:  # ... and the one of the long-window kernel: k = 127, 2 x 241..256 (k_pe_tiles<2, 16, 3>)
            return dict(k=127, L=int(rng.integers(238, 259)), n_strains=int(rng.integers(2, 12)), glen=int(rng.integers(900, 5000)),
                        snp=float(rng.choice([0.01, 0.04, 0.1])), pairs=int(rng.integers(2000, 30000)),
                        sub=float(rng.choice([0.0, 0.005, 0.02])), nrate=float(rng.choice([0.0, 0.01])),
                        dirty=float(rng.choice([0.0, 0.0, 0.002, 0.03])), ragged=bool(rng.random() < 0.2), seed=int(rng.integers(0, 2 ** 31)))
        return dict(k=55, L=int(rng.integers(97, 161)), n_strains=int(rng.integers(2, 25)), glen=int(rng.integers(600, 5000)),
                    snp=float(rng.choice([0.01, 0.05, 0.12, 0.2])), pairs=int(rng.integers(3000, 60000)),
                    sub=float(rng.choice([0.0, 0.005, 0.02])), nrate=float(rng.choice([0.0, 0.01])),
                    dirty=float(rng.choice([0.0, 0.0, 0.002, 0.03])), ragged=bool(rng.random() < 0.2), seed=int(rng.integers(0, 2 ** 31)))
    k = int(rng.choice(K_EDGE)) if rng.random() < 0.7 else int(rng.integers(3, 128))
    w = min(31, k + 1)
    w -= 1 - (w & 1)
    edges = [k + 1, k + 2, w + 127, w + 128, w + 129, w + 159, w + 160, w + 161, w + 255, w + 256, w + 257, 97, 112, 113, 128, 129, 145, 150, 159, 160, 250]
    L = int(rng.choice(edges)) if rng.random() < 0.7 else int(rng.integers(max(k - 3, 4), k + 320))
    L = max(4, min(L, 600))
    return dict(k=k, L=L, n_strains=int(rng.integers(1, 7)), glen=int(rng.integers(max(2 * L, 3 * k, 200), 3500)),
                snp=float(rng.choice([0.0, 0.01, 0.04, 0.1])), pairs=int(rng.integers(200, 6000)),
                sub=float(rng.choice([0.0, 0.005, 0.02, 0.06])), nrate=float(rng.choice([0.0, 0.01, 0.1])),
                dirty=float(rng.choice([0.0, 0.0, 0.002, 0.02, 0.2])), ragged=bool(rng.random() < 0.4),
                seed=int(rng.integers(0, 2 ** 31)))


def make_reads(p, st):
    fwd, rve = synth.sample_pairs(st, p["pairs"], p["L"], seed=p["seed"] + 1, sub_rate=p["sub"], n_rate=p["nrate"])
    rng = np.random.default_rng(p["seed"] + 2)
    out = []
    for reads in (fwd, rve):
        res = []
        for s in reads:
            b = bytearray(s.encode())
            if p["ragged"] and len(b) > 2 and rng.random() < 0.5:
                b = b[: int(rng.integers(0, len(b) + 1))]
            if p["dirty"] > 0.0 and len(b):
                hits = np.nonzero(rng.random(len(b)) < p["dirty"])[0]
                for h in hits:
                    b[int(h)] = int(DIRTY[int(rng.integers(0, len(DIRTY)))])
            res.append(b.decode())
        out.append(res)
    # deep coverage: ends that repeat other ends letter for letter
    n = len(out[0])
    if n > 1 and rng.random() < 0.35:
        for _ in range(n // 2 + 1):
            i, j, which = int(rng.integers(0, n)), int(rng.integers(0, n)), int(rng.integers(0, 3))
            if which != 2:
                out[0][j] = out[0][i]
            if which != 1:
                out[1][j] = out[1][i]
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = host.Context(0)
    t0 = time.time()
    n = bad = skipped = 0
    kernels = {}
    while time.time() - t0 < budget:
        p = draw(rng)
        try:
            st = synth.make_strains(p["n_strains"], p["glen"], p["snp"], seed=p["seed"])
            g = synth.compact_dbg(st, p["k"])
        except Exception as err:  # (a draw the generator cannot make, e.g. genome shorter than k)
            skipped += 1
            continue
        if len(g.seqs) == 0:
            skipped += 1
            continue
        fwd, rve = make_reads(p, st)
        want = pe_oracle_c.Oracle(g.seqs, p["k"]).count_pairs(fwd, rve)
        env = {}
        if rng.random() < 0.3:
            env = dict([[("VS_NO_STD", "1")], [("VS_NO_FAST", "1")], [("VS_EPT", "6")], [("VS_ACC_ROWS", "1"), ("VS_ROWS_KEYS", "64")], [("VS_ACC_ROWS", "1"), ("VS_LTAB_BITS", "0")], [("VS_ACC_ROWS", "1"), ("VS_ROWS_SUB", "1024")], [("VS_ACC_ROWS", "1"), ("VS_ACC_FILL", "1")], [("VS_ACC_FILL", "100")],
                        [("VS_NO_SORT", "1")], [("VS_LOCUS_GLOBAL", "1")], [("VS_ACC_ROWS", "1")], [("VS_ACC_ROWS", "1"), ("VS_ROWS_PER_STRIP", "3"), ("VS_LTAB_BITS", "4")]][int(rng.integers(0, 12))])
        if os.environ.get("FUZZ_ROWS") == "1":  # a campaign on the row-owner counters alone: every draw takes them, under varying switches
            env = {"VS_ACC_ROWS": "1"}
            for key, choices in (("VS_ROWS_PER_STRIP", ("1", "2", "5", "64")), ("VS_ROWS_KEYS", ("7", "64", "1000")), ("VS_ROWS_SUB", ("1024", "4096")),
                                 ("VS_LTAB_BITS", ("0", "2", "6", "12")), ("VS_ACC_FILL", ("1", "30", "100")), ("VS_NO_SORT", ("1",))):
                if rng.random() < 0.3:
                    env[key] = str(choices[int(rng.integers(0, len(choices)))])
        if rng.random() < 0.15:
            env["VS_NO_MID"] = "1"  # overflow pairs straight to the general kernel
        # (r5) the probe grid: the adaptive step grid of the compile-time-shape kernels forced on / off (it is otherwise chosen by
        # the index's postings per seed), and now and then the grid from offset 0 of the rounds before
        u = rng.random()
        if u < 0.4:
            env["VS_ADAPT_GRID"] = "1"
        elif u < 0.55:
            env["VS_ADAPT_GRID"] = "0"
        elif u < 0.62 and "VS_NO_STD" not in env:
            env["VS_PHASE0"] = "1"
        os.environ.update(env)
        try:
            ctx.build_index(g.seqs, p["k"])
            counter = host.PeCounter(ctx)
            block = ctx.pack_pairs(fwd, rve)
            counter.add(block)
            node_mat, short_mat, stats = counter.result()
            kern = ctx.last_kernel + (" slow>0" if ctx.last_timing()["slow_pairs"] else "")
        finally:
            for key in env:
                os.environ.pop(key, None)
        kernels[kern] = kernels.get(kern, 0) + 1
        ok = np.array_equal(node_mat, want[0]) and np.array_equal(short_mat, want[1]) and stats == tuple(int(x) for x in want[2])
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", dict(p, env=env, nodes=len(g.seqs), kernel=kern,
                                   node_diff=int((node_mat != want[0]).sum()), short_diff=int((short_mat != want[1]).sum()),
                                   stats=(stats, tuple(int(x) for x in want[2]))), flush=True)
    print("draws %d, mismatches %d, skipped %d, %.0f s; kernels %s" % (n, bad, skipped, time.time() - t0, kernels))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
