#!/usr/bin/env python3
"""Build container only: randomized campaign of the whole pipeline against the REAL reference CLI.

Every draw is a seeded synthetic case in the parameter ranges of the committed fixtures
(make_graph_golden.CASES).  The reference (``/root/reference/vstrains`` behind
``tests/golden/gt_standin``, exactly as make_graph_golden.py runs it: both in-edge-order models,
hash seeds 0-3) writes its outputs into a scratch golden directory; this build's pipeline then runs
on the same inputs with the native stage engine over the CPU checker (as tests/test_graph_golden.py does) and every file
that the reference itself produces deterministically must be identical -- stage GFAs, contig
files, pe_info / st_info, strain.paths, strain.fasta, the INFO log lines.  Nothing is committed
from here except the tally.

    python tests/golden/fuzz_reference.py [draws=40] [seed=1] [workers=4]
"""
import json
import multiprocessing as mp
import os

os.environ.setdefault("VS_CHECK_UNTOUCHED", "1")  # (hints of the graph stages verified)
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT)
sys.path.insert(0, TESTS)
sys.path.insert(0, HERE)


REFERENCE_LIMIT_S = 900


def one(job):
    idx, kwargs, extra, scratch = job
    import contextlib
    import io

    import numpy as np  # noqa: F401

    import graph_case
    import make_graph_golden as gold
    from test_graph_golden import CheckerBackend, NativeBackend
    from vstrains_amd.graph import pipeline

    name = "fuzz_%04d" % idx
    gold.OUT = scratch
    graph_case.GOLDEN = scratch
    gold.CASES[name] = (kwargs, extra)
    import subprocess

    plain_run = subprocess.run

    def run_with_limit(*a, **kw):  # (the reference's trivial-split fixpoint runs up to N^2 rounds on some circular graphs)
        kw.setdefault("timeout", REFERENCE_LIMIT_S)
        return plain_run(*a, **kw)

    gold.subprocess.run = run_with_limit
    with contextlib.redirect_stdout(io.StringIO()) as line:
        try:
            gold.emit(name)
        except subprocess.TimeoutExpired:
            return dict(idx=idx, status="reference_timeout", kwargs=kwargs)
        except Exception as err:  # (a draw the generator cannot make)
            return dict(idx=idx, status="generator", detail=repr(err))
    case = graph_case.Case(name)
    res = dict(idx=idx, kwargs=kwargs, extra=extra, rc=case.meta["returncode"], files=len(case.meta["files"]),
               inedge_invariant=case.meta["inedge_invariant"], hashseed_invariant=case.meta["hashseed_invariant"],
               ops=case.meta.get("operations_in_reference_log"))
    with tempfile.TemporaryDirectory() as tmp:
        inp = case.inputs(tmp)
        out = os.path.join(tmp, "out")
        args = case.args(inp, out)
        logger = graph_case.file_logger(out, "fuzz-%d" % idx)
        err = None
        try:
            # FUZZ_ENGINE=python: the Python statement of the stages (oracle/graph_stages); default: the native stage engine
            # (vs_stage) over the CPU checker of its device operations -- the product's decisions against the real reference
            pipeline.run(args, logger, CheckerBackend(case, False) if os.environ.get("FUZZ_ENGINE") == "python" else NativeBackend(case))
        except BaseException as e:  # noqa: B036 (KeyError / SystemExit where the reference exits non-zero too)
            err = "%s: %s" % (type(e).__name__, e)
        for h in list(logger.handlers):
            h.flush()
        problems, _ = graph_case.compare(case, out)
        binding = [p for p in problems if p.split(" ", 1)[1] not in case.meta["differs_under_other_hashseeds"]]
        # The reference iterates sets of contig names: its outputs -- down to how many paths it extracts, hence
        # which files exist -- depend on PYTHONHASHSEED, and four seeds can agree by chance.  Before anything
        # counts as a mismatch the reference is run under further seeds: a seed under which EVERY file it
        # writes equals this build's settles it.
        if binding:
            ours = gold.collect(out)
            pc = graph_case.synth.make_pipeline_case(**kwargs)
            full = dict(inp)
            for key, text in (("fwd", graph_case.synth.fastq_text(pc.fwd, "f")), ("rve", graph_case.synth.fastq_text(pc.rve, "r"))):
                full[key] = os.path.join(tmp, key + "_full.fq")
                with open(full[key], "w") as fh:
                    fh.write(text)
            for hs in range(1, 16):
                rc_hs, files_hs, _ = gold.run_reference(full, extra, "rotate", hs)
                if all(ours.get(f) == files_hs[f] for f in files_hs) and set(ours) <= set(files_hs) | {"vstrains.log.info"}:
                    res["matches_reference_under_hashseed"] = hs
                    res["reference_rc_under_that_hashseed"] = rc_hs
                    binding = []
                    break
    # (the reference's EXIT STATUS depends on the hash seed too: a run settled under another seed is held to that seed's status)
    ref_rc = res.get("reference_rc_under_that_hashseed", case.meta["returncode"])
    if (err is None) != (ref_rc == 0):
        binding.append("exit: ours %r, reference rc %d" % (err, ref_rc))
    res["status"] = "MISMATCH" if binding else "ok_other_hashseed" if "matches_reference_under_hashseed" in res else "ok"
    res["binding_problems"] = binding
    res["non_binding"] = len(problems) - len([p for p in binding if not p.startswith("exit")])
    shutil.rmtree(os.path.join(scratch, name), ignore_errors=True)
    return res


def main():
    import numpy as np

    draws = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    scratch = tempfile.mkdtemp(prefix="vstrains_fuzz_ref_")
    jobs = []
    big = os.environ.get("FUZZ_BIG") == "1"  # (round 5: larger cases -- up to 10 strains of up to 13 kb, a thousand nodes and more)
    for i in range(draws):
        k = int(rng.choice([21, 21, 31, 55]))
        L = int(rng.choice([100, 120, 150])) if k < 55 else 150
        kwargs = dict(n_strains=int(rng.integers(2, 7)) if not big else int(rng.integers(5, 11)),
                      genome_len=(int(rng.integers(1400, 4500)) if k < 55 else int(rng.integers(3000, 6500))) * (2 if big else 1),
                      snp_rate=float(rng.choice([0.004, 0.008, 0.01, 0.012, 0.015, 0.02])), k=k,
                      n_pairs=int(rng.integers(1500, 9000)) * (3 if big else 1), read_len=L, seed=int(rng.integers(1000, 10 ** 6)),
                      abundance_ratio=float(rng.choice([0.45, 0.55, 0.6, 0.7, 0.8, 0.95])))
        if rng.random() < 0.5:
            kwargs["scramble"] = True
        if rng.random() < 0.2:
            kwargs["error_strain_depth"] = float(rng.choice([3.0, 6.0]))
        if rng.random() < 0.15:
            kwargs["repeat_len"] = int(rng.choice([80, 120]))
        if rng.random() < 0.15:
            kwargs["circular"] = True
        if rng.random() < 0.2:
            kwargs["sub_rate"] = float(rng.choice([0.002, 0.004]))
        if rng.random() < 0.1:
            kwargs["gapped_contigs"] = int(rng.integers(1, 5))
        if rng.random() < 0.1:
            kwargs["depth_tags"] = "kc"
        extra = []
        if rng.random() < 0.15:
            extra = ["-mc", str(int(rng.choice([10, 20, 60, 150])))]
        elif rng.random() < 0.1:
            extra = ["-ml", "100"]
        jobs.append((i, kwargs, extra, scratch))
    if os.environ.get("FUZZ_ONLY"):  # (one draw of a campaign again: FUZZ_ONLY=<idx> with the campaign's draws / seed)
        jobs = [j for j in jobs if j[0] == int(os.environ["FUZZ_ONLY"])]
    tally = {"ok": 0, "ok_other_hashseed": 0, "MISMATCH": 0, "generator": 0, "reference_timeout": 0}
    stats = dict(rc_nonzero=0, inedge_invariant=0, hashseed_invariant=0, files=0, with_link_split=0, with_cov_match=0, with_trivial=0)
    with mp.get_context("spawn").Pool(workers) as pool:
        for res in pool.imap_unordered(one, jobs):
            tally[res["status"]] += 1
            if res["status"] == "MISMATCH":
                print("MISMATCH", json.dumps(res), flush=True)
            if res["status"] not in ("generator", "reference_timeout"):
                stats["rc_nonzero"] += int(res["rc"] != 0)
                stats["inedge_invariant"] += int(res["inedge_invariant"])
                stats["hashseed_invariant"] += int(res["hashseed_invariant"])
                stats["files"] += res["files"]
                ops = res.get("ops") or {}
                stats["with_link_split"] += int(ops.get("link_split", 0) > 0)
                stats["with_cov_match"] += int(ops.get("coverage_match", 0) > 0)
                stats["with_trivial"] += int(ops.get("trivial_split", 0) > 0)
            print("draw %d: %s" % (res["idx"], res["status"]), flush=True)
    shutil.rmtree(scratch, ignore_errors=True)
    print("draws %d: %s; %s" % (draws, tally, stats))
    return 1 if tally["MISMATCH"] else 0


if __name__ == "__main__":
    sys.exit(main())
