#!/usr/bin/env python3
"""GPU-side: randomized campaign for the strain-extract leg.  Random strain sets go through the
bench workload generator (assembler-style GFA + contig paths, the pipeline's own preparation), the
device counts a block of synthetic read pairs, and the extraction then runs twice: on the device
(HipBackend: vs_stage_rebuild, flow / scan kernels, link sums in HBM) and over the numpy checker
(oracle/graph_ops.py, Python rebuild, link sums off the host copy of the counters).  Every file the
two runs write must be identical.  Test infrastructure (uses the oracle).

    python tests/fuzz_graph.py [seconds=300] [seed=1]
"""
import copy
import os

os.environ.setdefault("VS_CHECK_UNTOUCHED", "1")  # (hints of the graph stages verified)
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import profile_extract_cpu as pec  # noqa: E402  (checker backend, digests)
from vstrains_amd import pe as host  # noqa: E402
from vstrains_amd.graph import pipeline  # noqa: E402
from vstrains_amd.graph.hip_ops import HipBackend, HipPeLinks  # noqa: E402
from vstrains_amd.workloads import workload  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = host.Context(0)
    backend = HipBackend(ctx=ctx)
    t0 = time.time()
    n = bad = failed = 0
    sizes = []
    while time.time() - t0 < budget:
        p = dict(k=int(rng.choice([21, 31, 55])), n_strains=int(rng.integers(2, 11)), genome_len=int(rng.integers(900, 6000)),
                 snp=float(rng.choice([0.01, 0.03, 0.06, 0.1])), ratio=float(rng.choice([0.6, 0.8, 0.95])),
                 read_len=int(rng.choice([100, 150, 250])), pairs=int(rng.integers(3000, 120000)), seed=int(rng.integers(0, 2 ** 31)))
        work = tempfile.mkdtemp(prefix="vstrains_fuzz_")
        try:
            try:
                st, pre, names, seqs, cum, logger, _ = workload(os.path.join(work, "w"), k=p["k"], n_strains=p["n_strains"],
                                                                genome_len=p["genome_len"], snp_rate=p["snp"], seed=p["seed"],
                                                                read_len=p["read_len"], abundance_ratio=p["ratio"])
            except (Exception, SystemExit) as err:  # (a draw the generator / the preparation refuses, e.g. nothing above the coverage cut-off)
                failed += 1
                continue
            if len(names) == 0:
                failed += 1
                continue
            ctx.build_index(seqs, p["k"])
            reads = ctx.synth_pairs(st.genomes, cum, p["seed"] ^ 0x5A5A, 0, p["pairs"], p["read_len"], int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
            counter = host.PeCounter(ctx)
            counter.add(reads)
            node_mat, short_mat, _ = counter.result()
            res = []
            for which in ("device", "checker"):
                out = os.path.join(work, which)
                for sub in ("gfa", "tmp"):
                    os.makedirs(os.path.join(out, sub), exist_ok=True)
                try:
                    if which == "device":
                        strains = pipeline.extract_strains(copy.deepcopy(pre), HipPeLinks.from_counter(ctx, counter, names), backend, logger, out)
                    else:
                        strains = pipeline.extract_strains(copy.deepcopy(pre), pec.NumpyPeLinks(names, node_mat, short_mat), pec.Backend(), logger, out)
                    res.append((pec.digests(out), len(strains), None))
                except Exception as err:  # both runs must fail alike (e.g. the reference's divide-by-zero on an isolated branch)
                    res.append((pec.digests(out), -1, "%s: %s" % (type(err).__name__, err)))
            (dev, n_dev, e_dev), (ref, n_ref, e_ref) = res
            diff = sorted(f for f in set(dev) | set(ref) if dev.get(f) != ref.get(f))
            n += 1
            sizes.append(len(names))
            if diff or n_dev != n_ref or e_dev != e_ref:
                bad += 1
                print("MISMATCH", dict(p, nodes=len(names), strains=(n_dev, n_ref), errors=(e_dev, e_ref), files=diff[:6]), flush=True)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    print("draws %d, mismatches %d, refused by the generator %d, %.0f s; nodes per draw min / median / max %s" % (
        n, bad, failed, time.time() - t0, (min(sizes), int(np.median(sizes)), max(sizes)) if sizes else None), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
