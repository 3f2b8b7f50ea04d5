#!/usr/bin/env python3
"""CPU-side randomized campaign for the native stage engine (vstrains_amd/csrc/vs_stage.cpp).  Random strain sets go
through the bench workload generator and the pipeline's own preparation, the C oracle counts a block of synthetic read
pairs, and the extraction leg then runs twice over the same CPU checker of the device operations: the Python statement
of the stages (vstrains_amd/graph/disentangle.py, extend.py: pinned to the reference by the golden cases and the
reference campaigns) and the native engine.  Every file the two runs write, the strain records and the error of a failed
run must be identical.  Test infrastructure (uses the oracle).

    python tests/fuzz_native_cpu.py [seconds=120] [seed=1]
"""
import copy
import os

os.environ.setdefault("VS_CHECK_UNTOUCHED", "1")
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import native_check  # noqa: E402
import profile_extract_cpu as pec  # noqa: E402  (checker backend, digests)
from oracle import pe_oracle_c  # noqa: E402
from vstrains_amd import synth  # noqa: E402
from vstrains_amd.graph import pipeline  # noqa: E402
from vstrains_amd.workloads import workload  # noqa: E402


class PythonStages(pec.Backend):
    """The Python statement of the stages over the numpy checker."""


class NativeStages:
    """The native engine over the C++ checker of its device operations."""

    def native_stage(self, table):
        return native_check.stage_over_checker(table.names, native_check.dense_links(table))


def one_draw(p, work):
    st, pre, names, seqs, cum, logger, _ = workload(os.path.join(work, "w"), k=p["k"], n_strains=p["n_strains"], genome_len=p["genome_len"],
                                                    snp_rate=p["snp"], seed=p["seed"], read_len=p["read_len"], abundance_ratio=p["ratio"])
    if len(names) == 0:
        return None
    fwd, rve = synth.sample_pairs(st, p["pairs"], p["read_len"], seed=p["seed"] ^ 0x5A5A, sub_rate=0.005, n_rate=0.001)
    node_mat, short_mat, _ = pe_oracle_c.Oracle(seqs, p["k"]).count_pairs(fwd, rve)
    res = []
    for which, backend in (("python", PythonStages()), ("native", NativeStages())):
        out = os.path.join(work, which)
        for sub in ("gfa", "tmp"):
            os.makedirs(os.path.join(out, sub), exist_ok=True)
        try:
            strains = pipeline.extract_strains(copy.deepcopy(pre), pec.NumpyPeLinks(names, node_mat, short_mat), backend, logger, out)
            res.append((pec.digests(out), {k: (list(v[0]), int(v[1]), repr(v[2]), type(v[2]).__name__) for k, v in strains.items()}, None))
        except Exception as err:  # both runs must fail alike (e.g. the reference's divide-by-zero on an isolated branch)
            res.append((pec.digests(out), None, type(err).__name__))
    (a, sa, ea), (b, sb, eb) = res
    diff = sorted(f for f in set(a) | set(b) if a.get(f) != b.get(f))
    return len(names), diff, sa == sb, (ea, eb)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    n = bad = failed = errors = 0
    sizes = []
    while time.time() - t0 < budget:
        p = dict(k=int(rng.choice([21, 31, 55])), n_strains=int(rng.integers(2, 11)), genome_len=int(rng.integers(900, 6000)),
                 snp=float(rng.choice([0.01, 0.03, 0.06, 0.1])), ratio=float(rng.choice([0.6, 0.8, 0.95])),
                 read_len=int(rng.choice([100, 150, 250])), pairs=int(rng.integers(1000, 20000)), seed=int(rng.integers(0, 2 ** 31)))
        work = tempfile.mkdtemp(prefix="vstrains_fuzzn_")
        try:
            try:
                got = one_draw(p, work)
            except (Exception, SystemExit) as err:  # (a draw the generator / the preparation refuses)
                failed += 1
                continue
            if got is None:
                failed += 1
                continue
            nodes, diff, same_strains, errs = got
            n += 1
            sizes.append(nodes)
            errors += errs[0] is not None
            if diff or not same_strains or errs[0] != errs[1]:
                bad += 1
                print("MISMATCH", dict(p, nodes=nodes, errors=errs, same_strains=same_strains, files=diff[:6]), flush=True)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    print("draws %d, mismatches %d, draws on which both runs raise the same error %d, refused by the generator %d, %.0f s; nodes per draw min / median / max %s"
          % (n, bad, errors, failed, time.time() - t0, (min(sizes), int(np.median(sizes)), max(sizes)) if sizes else None), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
