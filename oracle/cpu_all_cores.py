"""All-host-cores run of the C restatement  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's
cpu_baseline leg starts it as a child process, so that the forked workers never see a GPU runtime).

    python -m oracle.cpu_all_cores <pickle> <workers> <seconds>

pickle: dict(seqs, k, genomes, cum, seed, first_pair, L, sub_thresh, n_thresh, rate_hint).  Every
worker builds its own table, takes its own slice of the seeded read stream, and counts it in chunks
(increment keys, the sparse form: 256 dense 200 MB matrices would not fit).  Prints one JSON line:
pairs, the slowest worker's seconds, pairs/s of the whole box."""
import json
import multiprocessing as mp
import pickle
import sys
import time

import numpy as np


def _work(args):
    from oracle import pe_oracle_c

    job, w, n = args
    orc = pe_oracle_c.Oracle(job["seqs"], job["k"])
    L = job["L"]
    chunk = 20000
    fw, rv = pe_oracle_c.synth_pairs(job["genomes"], job["cum"], job["seed"], job["first_pair"] + w * n, n, L,
                                     job["sub_thresh"], job["n_thresh"])
    keys = np.empty(chunk * 400, dtype=np.uint64)
    stats = np.zeros(3, dtype=np.uint64)
    off = np.arange(chunk + 1, dtype=np.uint64) * np.uint64(L)
    L_ = pe_oracle_c.lib()
    t0 = time.perf_counter()
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        f = np.ascontiguousarray(fw[lo:lo + m]).reshape(-1)
        r = np.ascontiguousarray(rv[lo:lo + m]).reshape(-1)
        L_.peo_count_pairs_keys(orc._h, f.ctypes.data, off.ctypes.data, r.ctypes.data, off.ctypes.data, m,
                                keys.ctypes.data, keys.size, stats.ctypes.data)
    return n, time.perf_counter() - t0


def main():
    with open(sys.argv[1], "rb") as fh:
        job = pickle.load(fh)
    workers, seconds = int(sys.argv[2]), float(sys.argv[3])
    n = max(20000, int(job["rate_hint"] * seconds))
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_work, [(job, w, n) for w in range(workers)])
    total = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)
    print(json.dumps({"pairs": total, "seconds": slowest, "pairs_per_s": total / slowest, "workers": workers}))


if __name__ == "__main__":
    main()
