"""Gate for a one-lane-per-window mapping kernel (VERDICT r3 #2a): how long do the 1.9e9 scattered 16-byte slot loads of
configs[2] take on their own?  Runs k_probe_gate (csrc/vs_walk.hip, experiment build) on the bench workload's real index
and reads and prints one JSON line; the decision recorded in profiles/EXPERIMENTS.md comes from this.

    VS_EXPERIMENT=1 VS_WALK=1 python tools/probe_gate.py [--config 2] [--pairs N]
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile

os.environ.setdefault("VS_EXPERIMENT", "1")
os.environ.setdefault("VS_WALK", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402,F401

from vstrains_amd import _native as nat  # noqa: E402
from vstrains_amd import pe as host  # noqa: E402
from vstrains_amd.workloads import CONFIGS, workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    R = args.pairs or cfg["total_pairs"] // cfg["gpus"]
    work = tempfile.mkdtemp(prefix="vs_gate_")
    st, pre, names, seqs, cum, logger, _ = workload(work, k=cfg["k"], n_strains=cfg["n_strains"], genome_len=cfg["genome_len"],
                                                    snp_rate=cfg["snp_rate"], seed=cfg["seed"], read_len=cfg["read_len"],
                                                    abundance_ratio=cfg["abundance_ratio"])
    ctx = host.Context(0)
    ctx.build_index(seqs, cfg["k"])
    reads = ctx.synth_pairs(st.genomes, cum, 20250000 + args.config, 0, R, cfg["read_len"], int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    counter = host.PeCounter(ctx)
    counter.add(reads)  # (leaves the locus order of these pairs in the context: use_perm = 1 walks the reads in that order)
    ctx.sync()
    base = ctx.last_timing()
    fn = nat.lib().vs_exp_probe_gate
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    out = {"config": args.config, "pairs": R, "walk_info": ctx.walk_info, "k_pe_tiles_ms": base["main_ms"], "kernel": ctx.last_kernel, "runs": []}
    for mode in (0, 1):
        for use_perm in (0, 1):
            ms, probes = C.c_double(0), C.c_uint64(0)
            nat.check(ctx._h, fn(ctx._h, reads._h, mode, use_perm, args.reps, C.byref(ms), C.byref(probes)))
            out["runs"].append({"mode": "mixed slot per lane" if mode == 0 else "real windows: LDS read tile, hash, probe chain",
                                "locus_order": bool(use_perm), "ms": ms.value, "probes": probes.value,
                                "probes_per_s": probes.value / (ms.value * 1e-3)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
