"""Not a test: the strain-extract leg of one bench config on a box without a GPU, for profiling the
host logic of the graph stages and for checking it, file by file, against what the device run
wrote (tools/extract_dump.py leaves gpurun_out/links_c<i>.npz + extract_digest_c<i>.json).
Device operations are replaced by the numpy checker (oracle/graph_ops.py), so absolute times of the
ops differ; the Python around them is the same code.

    python tests/profile_extract_cpu.py --config 2 [--profile] [--reps 3]
"""
import argparse
import copy
import hashlib
import json
import os

os.environ.setdefault("VS_CHECK_UNTOUCHED", "1")  # (hints of the graph stages verified)
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import graph_ops as chk  # noqa: E402
from vstrains_amd.graph import pipeline  # noqa: E402
from oracle.graph_stages.run import PythonStages  # noqa: E402
from vstrains_amd.workloads import CONFIGS, workload  # noqa: E402


class Backend(PythonStages):
    """The Python restatement of the stages (oracle/graph_stages) over the numpy checker of the device operations."""

    def __init__(self):
        self.graph_ops = chk.NumpyGraphOps()


class NumpyPeLinks(chk.DictPeLinks):
    """The same table as DictPeLinks (key {u, v}: both orders of both matrices summed, the diagonal
    once) held as a symmetric matrix, so that a config of 5 000 nodes is checked in seconds."""

    def __init__(self, names, node_mat, short_mat):
        self.names = list(names)
        self._index = {n: i for i, n in enumerate(self.names)}
        m = node_mat + short_mat
        sym = m + m.T
        idx = np.arange(len(names))
        sym[idx, idx] = m[idx, idx]
        self.sym = sym

    def block_sums(self, queries):
        return [int(self.sym[np.ix_(list(rows), list(cols))].sum()) if len(rows) and len(cols) else 0 for rows, cols in queries]

    def group_matrix(self, groups):
        rows = np.stack([self.sym[list(g)].sum(axis=0) for g in groups]) if len(groups) else np.zeros((0, 0), dtype=np.int64)
        return np.stack([rows[:, list(g)].sum(axis=1) for g in groups], axis=1) if len(groups) else rows


class NativeBackend:
    def native_stage(self, table):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import native_check

        return native_check.stage_over_checker(table.names, native_check.dense_links(table))


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
    ap.add_argument("--dir", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--native", action="store_true", help="the native stage engine over the C++ checker instead of the Python stages")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    work_dir = tempfile.mkdtemp(prefix="vstrains_prof_")
    st, pre, names, seqs, cum, logger, _ = workload(
        work_dir, k=cfg["k"], n_strains=cfg["n_strains"], genome_len=cfg["genome_len"], snp_rate=cfg["snp_rate"],
        seed=cfg["seed"], read_len=cfg["read_len"], abundance_ratio=cfg["abundance_ratio"])
    z = np.load(os.path.join(args.dir, "links_c%d.npz" % args.config))
    n = int(z["n"])
    assert n == len(names)
    node_mat = np.zeros((n, n), dtype=np.int64)
    short_mat = np.zeros((n, n), dtype=np.int64)
    node_mat[z["ni"], z["nj"]] = z["nv"]
    short_mat[z["si"], z["sj"]] = z["sv"]
    table = NumpyPeLinks(names, node_mat, short_mat)
    want = json.load(open(os.path.join(args.dir, "extract_digest_c%d.json" % args.config)))
    times = []
    for rep in range(args.reps):
        out_dir = os.path.join(work_dir, "out%d" % rep)
        for sub in ("gfa", "tmp"):
            os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
        pre_i = copy.deepcopy(pre)
        prof = None
        if args.profile and rep == args.reps - 1:
            import cProfile

            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        strains = pipeline.extract_strains(pre_i, table, NativeBackend() if args.native else Backend(), logger, out_dir)
        times.append(time.perf_counter() - t0)
        if args.native:
            print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in pipeline.extract_strains.last_stages.items()})
        if prof is not None:
            import pstats

            prof.disable()
            pstats.Stats(prof).sort_stats("tottime").print_stats(45)
    got = digests(out_dir)
    bad = sorted(f for f in set(got) | set(want["files"]) if got.get(f) != want["files"].get(f))
    print("%s; seconds %s (device run that left the digests: %s)" % (
        "native engine over the C++ checker" if args.native else "Python stages over the numpy checker", ["%.3f" % t for t in times],
        ["%.3f" % t for t in want["seconds"]]))
    print("strains %d (device run %d); %d files, %d differ %s" % (len(strains), want["strains"], len(got), len(bad), bad[:8]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
