"""All-host-cores run of the C restatement  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's
cpu_baseline leg starts it as a child process, so that the forked workers never see a GPU runtime).

    python -m oracle.cpu_all_cores <pickle> <workers> <seconds>

pickle: dict(seqs, k, genomes, cum, seed, first_pair, L, sub_thresh, n_thresh).  Every worker builds
its own table and counts chunks of its own slice of the seeded read stream (increment keys, the
sparse form: 256 dense 200 MB matrices would not fit) until <seconds> of counting time are used.
Prints one JSON line: pairs, seconds, pairs/s of the whole box (sum of the workers' rates)."""
import json
import multiprocessing as mp
import pickle
import sys
import time

import numpy as np


def _work(args):
    from oracle import pe_oracle_c

    job, w, seconds = args
    orc = pe_oracle_c.Oracle(job["seqs"], job["k"])
    L = job["L"]
    chunk = 5000
    keys = np.empty(chunk * 400, dtype=np.uint64)
    stats = np.zeros(3, dtype=np.uint64)
    off = np.arange(chunk + 1, dtype=np.uint64) * np.uint64(L)
    L_ = pe_oracle_c.lib()
    done, spent, i = 0, 0.0, 0
    # chunk after chunk of this worker's own slice of the stream until `seconds` of counting time are used
    # (time-boxed: how fast a worker runs next to all the others is what is being measured)
    while spent < seconds:
        first = job["first_pair"] + (w * 100000 + i) * chunk
        fw, rv = pe_oracle_c.synth_pairs(job["genomes"], job["cum"], job["seed"], first, chunk, L, job["sub_thresh"], job["n_thresh"])
        f, r = fw.reshape(-1), rv.reshape(-1)
        t0 = time.perf_counter()
        L_.peo_count_pairs_keys(orc._h, f.ctypes.data, off.ctypes.data, r.ctypes.data, off.ctypes.data, chunk,
                                keys.ctypes.data, keys.size, stats.ctypes.data)
        spent += time.perf_counter() - t0
        done += chunk
        i += 1
    return done, spent


def main():
    with open(sys.argv[1], "rb") as fh:
        job = pickle.load(fh)
    workers, seconds = int(sys.argv[2]), float(sys.argv[3])
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_work, [(job, w, seconds) for w in range(workers)])
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    print(json.dumps({"pairs": total, "seconds": max(r[1] for r in res), "wall_s": wall,
                      "pairs_per_s": sum(r[0] / r[1] for r in res), "workers": workers}))


if __name__ == "__main__":
    main()
