#!/usr/bin/env python3
"""Build container only: one draw of a ``tests/golden/fuzz_reference.py`` campaign, both sides WITHOUT a time limit and
into directories that stay, so that the stage files either side has written so far can be compared while they run.

    python tools/draw_investigate.py inputs  <draws> <seed> <idx> <work>     # work/in/{graph.gfa,contigs.paths,fwd.fq,rve.fq}, work/draw.json
    python tools/draw_investigate.py reference <work> [variant=rotate] [hashseed=0]   # the REAL reference CLI -> work/ref_<variant>_<hs>/
    python tools/draw_investigate.py ours <work>                              # this build (C oracle counts + native engine over the CPU checker) -> work/ours/
    python tools/draw_investigate.py compare <work> [ref dir name]            # digest-form comparison of every file both sides hold

(VERDICT r4 "Next" 1a: draw 236 of ``fuzz_reference.py 600 778``.)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def campaign_job(draws, seed, idx):
    """The (kwargs, extra) ``fuzz_reference.main`` draws for job ``idx`` (same generator calls, same order)."""
    import numpy as np

    rng = np.random.default_rng(seed)
    big = os.environ.get("FUZZ_BIG") == "1"  # (as tests/golden/fuzz_reference.py: the same draws under the same switch)
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
        if i == idx:
            return kwargs, extra
    raise SystemExit("no such draw")


def cmd_inputs(draws, seed, idx, work):
    from vstrains_amd import synth

    kwargs, extra = campaign_job(draws, seed, idx)
    pc = synth.make_pipeline_case(**kwargs)
    os.makedirs(os.path.join(work, "in"), exist_ok=True)
    for name, text in (("graph.gfa", pc.gfa_text), ("contigs.paths", pc.paths_text), ("fwd.fq", synth.fastq_text(pc.fwd, "f")),
                       ("rve.fq", synth.fastq_text(pc.rve, "r"))):
        with open(os.path.join(work, "in", name), "w") as fh:
            fh.write(text)
    with open(os.path.join(work, "draw.json"), "w") as fh:
        json.dump(dict(draws=draws, seed=seed, idx=idx, kwargs=kwargs, extra=extra, nodes=len(pc.graph.ids), k=pc.k), fh, indent=1)
    print(json.dumps(dict(kwargs=kwargs, extra=extra, nodes=len(pc.graph.ids))))


def _inp(work):
    d = os.path.join(work, "in")
    return {"gfa": os.path.join(d, "graph.gfa"), "paths": os.path.join(d, "contigs.paths"), "fwd": os.path.join(d, "fwd.fq"),
            "rve": os.path.join(d, "rve.fq")}


def cmd_reference(work, variant="rotate", hashseed="0"):
    import make_graph_golden as gold

    with open(os.path.join(work, "draw.json")) as fh:
        extra = json.load(fh)["extra"]
    inp = _inp(work)
    out = os.path.join(work, "ref_%s_%s" % (variant, hashseed))
    env = dict(os.environ)
    env["PYTHONPATH"] = gold.STANDIN + os.pathsep + env.get("PYTHONPATH", "")
    env["GT_STANDIN_INEDGE"] = variant
    env["PYTHONHASHSEED"] = str(hashseed)
    cmd = [sys.executable, gold.REF_CLI, "-a", "spades", "-g", inp["gfa"], "-p", inp["paths"], "-o", out, "-fwd", inp["fwd"], "-rve",
           inp["rve"], "-d"] + extra
    t0 = time.time()
    proc = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=work)
    with open(out + ".done.json", "w") as fh:
        json.dump(dict(rc=proc.returncode, seconds=time.time() - t0, stderr_tail=proc.stderr[-3000:]), fh, indent=1)
    print("reference rc %d after %.1f s" % (proc.returncode, time.time() - t0))


def cmd_ours(work):
    import argparse
    import resource

    import graph_case
    import native_check
    from oracle import graph_ops as chk
    from oracle import pe_oracle, pe_oracle_c
    from vstrains_amd.graph import pipeline

    with open(os.path.join(work, "draw.json")) as fh:
        extra = json.load(fh)["extra"]
    inp = _inp(work)
    out = os.path.join(work, "ours")
    for sub in ("gfa", "tmp", "paf", "aln"):
        os.makedirs(os.path.join(out, sub), exist_ok=True)

    class OracleCountsNativeStages:
        def pe_links(self, gfa, aln_dir, fwd, rve, ksize, names):
            ids, seqs = pe_oracle.read_gfa_segments(gfa)
            assert list(ids) == list(names)
            f, r = pe_oracle.fastq_sequences(fwd), pe_oracle.fastq_sequences(rve)
            node_mat, short_mat, _ = pe_oracle_c.Oracle(seqs, ksize).count_pairs(f, r)
            os.makedirs(aln_dir, exist_ok=True)
            for fname, mat in (("pe_info", node_mat), ("st_info", short_mat)):
                with open(os.path.join(aln_dir, fname), "w") as fh:
                    fh.write(pe_oracle.matrix_text(ids, mat))
            return chk.DictPeLinks(names, node_mat, short_mat)

        def native_stage(self, table):
            return native_check.stage_over_checker(table.names, native_check.dense_links(table))

    min_cov = int(extra[extra.index("-mc") + 1]) if "-mc" in extra else None
    min_len = int(extra[extra.index("-ml") + 1]) if "-ml" in extra else 250
    args = argparse.Namespace(gfa_file=inp["gfa"], path_file=inp["paths"], fwd=inp["fwd"], rve=inp["rve"], output_dir=out,
                              min_cov=min_cov, min_len=min_len, ref_file=None, dev=False)
    logger = graph_case.file_logger(out, "draw")
    t0 = time.time()
    err = None
    try:
        pipeline.run(args, logger, OracleCountsNativeStages())
    except BaseException as e:  # noqa: B036
        err = "%s: %s" % (type(e).__name__, e)
    ru = resource.getrusage(resource.RUSAGE_SELF)
    with open(out + ".done.json", "w") as fh:
        json.dump(dict(error=err, seconds=time.time() - t0, user=ru.ru_utime, system=ru.ru_stime), fh, indent=1)
    print("ours: %r after %.1f s (user %.1f, system %.1f)" % (err, time.time() - t0, ru.ru_utime, ru.ru_stime))


def cmd_compare(work, ref="ref_rotate_0"):
    import make_graph_golden as gold

    a = gold.collect(os.path.join(work, ref))
    b = gold.collect(os.path.join(work, "ours"))
    both = sorted(set(a) & set(b))
    same = [f for f in both if a[f] == b[f]]
    diff = [f for f in both if a[f] != b[f]]
    print("reference files %d, ours %d, in both %d: identical %d, different %d" % (len(a), len(b), len(both), len(same), len(diff)))
    for f in diff:
        print("  differs:", f)
    print("  only reference:", sorted(set(a) - set(b)))
    print("  only ours:", sorted(set(b) - set(a)))
    return 1 if diff else 0


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "inputs":
        cmd_inputs(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    elif what == "reference":
        cmd_reference(*sys.argv[2:])
    elif what == "ours":
        cmd_ours(sys.argv[2])
    elif what == "compare":
        sys.exit(cmd_compare(*sys.argv[2:]))
