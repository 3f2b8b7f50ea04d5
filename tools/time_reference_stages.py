#!/usr/bin/env python3
"""Pin AND time the strain-extraction leg of the REAL reference at bench graph sizes (VERDICT r5 "Next" 1).
Build container only; measurement tooling like tools/time_reference.py.

    python tools/time_reference_stages.py --config 2 --pairs 200000 --work /tmp/vs_refstage_c2 \
        [--models rotate,plain] [--seeds 0,1,2,3] [--jobs 4] [--out profiles/r6/reference_stages_config2.json]

Per config i (BASELINE.json configs[i], vstrains_amd.workloads):
  1. the config's assembler-style GFA + contig paths and the first `pairs` pairs of its bench stream (the CPU twin of
     the device generator) are written as input.gfa / input.paths / fwd.fq / rve.fq;
  2. the real utils/VStrains_PE_Inference.py runs ONCE on the config's s_graph_L1.gfa (its output depends on neither the
     stand-in's in-edge model nor the hash seed) -> <work>/pe_cache;
  3. for every (in-edge model, PYTHONHASHSEED) the real /root/reference/vstrains command runs behind
     tests/golden/gt_standin through tools/ref_stage_harness.py, which hands the cached PE files over where the command
     would launch the script (after checking the s_graph_L1.gfa the command itself wrote is byte-identical) and times
     everything after that point as one interval (VStrains_SPAdes.py:133-272);
  4. the digest-form files (tests/golden/make_graph_golden.collect) all runs agree on go to
     tests/golden/reference_digests.json "configs[i]_whole_command"; the per-run wall times to --out.
A run that does not finish leaves <work>/timing_<model>_<seed>.json.progress: every INFO line of the reference with the
seconds since start.  `--collect-only` gathers whatever runs have finished.
"""
import argparse
import concurrent.futures as cf
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference"


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def prepare(ci, pairs, work):
    from oracle import pe_oracle_c
    from tools.time_reference import write_fastq
    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[ci]
    meta_path = os.path.join(work, "inputs.json")
    if os.path.exists(meta_path):
        with open(meta_path) as fh:
            return json.load(fh)
    os.makedirs(work, exist_ok=True)
    st, pre, names, seqs, cum, logger, n_in = workload_for(ci, work)
    L, k = cfg["read_len"], cfg["k"]
    seed = 20250000 + ci
    fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, seed, 0, pairs, L, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
    write_fastq(os.path.join(work, "fwd.fq"), fw, b"f")
    write_fastq(os.path.join(work, "rve.fq"), rv, b"r")
    gfa = os.path.join(work, "gfa", "s_graph_L1.gfa")
    cache = os.path.join(work, "pe_cache")
    t0 = time.perf_counter()
    proc = subprocess.run([sys.executable, os.path.join(REF, "utils", "VStrains_PE_Inference.py"), "-g", gfa, "-o", cache,
                           "-f", os.path.join(work, "fwd.fq"), "-r", os.path.join(work, "rve.fq"), "-k", str(k)],
                          capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    pe_wall = time.perf_counter() - t0
    with open(os.path.join(cache, "s_graph_L1.sha256"), "w") as fh:
        fh.write(sha256_file(gfa) + "\n")
    meta = {"config": ci, "workload": cfg["tag"], "nodes": len(seqs), "input_segments": int(n_in), "pairs": int(pairs), "read_len": L, "k": k,
            "stream_seed": seed, "sub_thresh": int(0.005 * 2 ** 32), "n_thresh": int(0.001 * 2 ** 32),
            "s_graph_L1_gfa_sha256": sha256_file(gfa), "pe_info_sha256": sha256_file(os.path.join(cache, "pe_info")),
            "st_info_sha256": sha256_file(os.path.join(cache, "st_info")), "pe_script_wall_s": pe_wall}
    with open(meta_path, "w") as fh:
        json.dump(meta, fh, indent=1)
    return meta


def one_run(work, model, hashseed):
    out_v = os.path.join(work, "out_%s_%d" % (model, hashseed))
    timing = os.path.join(work, "timing_%s_%d.json" % (model, hashseed))
    if os.path.exists(timing):
        return model, hashseed, None
    if os.path.exists(out_v):
        import shutil
        shutil.rmtree(out_v)
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(ROOT, "tests", "golden", "gt_standin") + os.pathsep + env.get("PYTHONPATH", "")
    env["PYTHONHASHSEED"] = str(hashseed)
    env["GT_STANDIN_INEDGE"] = model
    env["MPLBACKEND"] = "Agg"
    t0 = time.perf_counter()
    with open(os.path.join(work, "stdout_%s_%d.log" % (model, hashseed)), "w") as log:
        pv = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_stage_harness.py"), "--timing", timing,
                             "--pe-cache", os.path.join(work, "pe_cache"), "--",
                             "-a", "spades", "-g", os.path.join(work, "input.gfa"), "-p", os.path.join(work, "input.paths"),
                             "-o", out_v, "-fwd", os.path.join(work, "fwd.fq"), "-rve", os.path.join(work, "rve.fq")],
                            stdout=log, stderr=subprocess.STDOUT, env=env, cwd=work)
    wall = time.perf_counter() - t0
    if os.path.exists(timing):
        with open(timing) as fh:
            t = json.load(fh)
        t["wall_s"], t["returncode"] = wall, pv.returncode
        with open(timing, "w") as fh:
            json.dump(t, fh, indent=1)
    return model, hashseed, pv.returncode


def strain_set_sha256(fasta_path):
    """SHA-256 of the extracted strain SEQUENCES as a set, each on its lexicographically smaller strand: what is left of
    strain.fasta when names, order and orientation -- everything that follows the stand-in's node numbering -- are dropped."""
    comp = str.maketrans("ACGT", "TGCA")
    seqs = []
    with open(fasta_path) as fh:
        for line in fh:
            line = line.strip()
            if line and not line.startswith(">"):
                seqs.append(min(line, line[::-1].translate(comp)))
    return hashlib.sha256("\n".join(sorted(seqs)).encode()).hexdigest()


def collect(work, meta, models, seeds, out_path, digests_path):
    import make_graph_golden as gold

    runs, timings = {}, {}
    for model in models:
        for hs in seeds:
            timing = os.path.join(work, "timing_%s_%d.json" % (model, hs))
            out_v = os.path.join(work, "out_%s_%d" % (model, hs))
            if not os.path.exists(timing):
                continue
            with open(timing) as fh:
                t = json.load(fh)
            if t.get("returncode") != 0:
                timings["%s/%d" % (model, hs)] = t
                continue
            files = {rel: hashlib.sha256(text.encode()).hexdigest() for rel, text in gold.collect(out_v).items()
                     if rel not in ("aln/pe_info", "aln/st_info", "vstrains.log.info")}
            runs[(model, hs)] = files
            t["strains"] = open(os.path.join(out_v, "strain.paths")).read().count("\n") // 2
            t["strain_set_sha256"] = strain_set_sha256(os.path.join(out_v, "strain.fasta"))
            t["files_written"] = len(files)
            timings["%s/%d" % (model, hs)] = t
    # runs that have not finished (or were stopped): how far the reference got and at which rate -- a measured "cannot"
    progress = {}
    for model in models:
        for hs in seeds:
            prog = os.path.join(work, "timing_%s_%d.json.progress" % (model, hs))
            if os.path.exists(prog) and not os.path.exists(prog[:-len(".progress")]):
                with open(prog) as fh:
                    lines = [l.rstrip("\n").split("\t", 1) for l in fh if "\t" in l]
                if not lines:
                    continue
                stored = [(float(t), m) for t, m in lines if m.endswith(" is stored..")]
                after_pe = next((float(t) for t, m in lines if m == "paired end information stored"), None)
                out_v = os.path.join(work, "out_%s_%d" % (model, hs))
                gfa_dir = os.path.join(out_v, "gfa")
                progress["%s/%d" % (model, hs)] = {
                    "state": "not finished when this record was written",
                    "seconds_since_start_at_last_log_line": float(lines[-1][0]),
                    "pe_files_handed_over_at_s": after_pe,
                    "stage_graphs_written": len(stored),
                    "last_stage_graph": os.path.basename(stored[-1][1].split(" ")[0]) if stored else None,
                    "seconds_per_stage_graph_after_pe": ((stored[-1][0] - after_pe) / max(len([1 for t, _ in stored if t > after_pe]), 1)) if (stored and after_pe) else None,
                    "last_lines": [[float(t), m[:120]] for t, m in lines[-6:]],
                    "gfa_files_on_disk": sorted(os.listdir(gfa_dir)) if os.path.isdir(gfa_dir) else []}
    result = dict(meta)
    result["runs"] = timings
    if progress:
        result["progress"] = progress
    result["host"] = {"cpus": os.cpu_count(), "python": sys.version.split()[0]}
    result["label"] = ("the real reference command behind tests/golden/gt_standin (a pure-Python model of graph-tool / gfapy: slower than the "
                       "C++ library it stands for); after_pe_to_run_return_s = VStrains_SPAdes.py:133-272 as one measured interval, 1 thread; "
                       "runs of different (model, seed) shared the container's 8 cores")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(result, fh, indent=1, sort_keys=True)
    print("written", out_path)
    if not runs:
        return
    base_key = ("rotate", 0) if ("rotate", 0) in runs else sorted(runs)[0]
    base = runs[base_key]
    names_all = sorted(set().union(*[set(r) for r in runs.values()]))
    # what the tests bind: the files every run under the BASE run's in-edge model agrees on (the product follows that model,
    # DESIGN.md 8); what is independent of the model is listed beside it -- at configs[1] the other model already writes another
    # graph_L0.gfa (the flip walk of IO.py:191-229 follows the adjacency order) and ends with 21 strains instead of 20
    same_model = {k: r for k, r in runs.items() if k[0] == base_key[0]}
    stable = [f for f in names_all if all(r.get(f) == base.get(f) for r in same_model.values())]
    stable_all = [f for f in names_all if all(r.get(f) == base.get(f) for r in runs.values())]
    ran_models = sorted({m for m, _ in runs})
    ran_seeds = sorted({h for _, h in runs})
    entry = {
        "workload": meta["workload"], "pairs": meta["pairs"], "stream_seed": meta["stream_seed"], "nodes": meta["nodes"],
        "s_graph_L1_gfa_sha256": meta["s_graph_L1_gfa_sha256"],
        "files_sha256": {f: base[f] for f in stable},
        "files_sha256_scope": "files every run under the in-edge model of base_run agrees on (all hash seeds run)",
        "files_identical_under_both_in_edge_models": stable_all,
        "strains_per_run": {"%s/%d" % k: timings["%s/%d" % k].get("strains") for k in sorted(runs)},
        "strain_set_sha256_per_run": {"%s/%d" % k: timings["%s/%d" % k].get("strain_set_sha256") for k in sorted(runs)},
        "strain_set_sha256": timings["%s/%d" % base_key].get("strain_set_sha256"),
        "strain_set_identical_in_every_run": len({timings["%s/%d" % k].get("strain_set_sha256") for k in runs}) == 1,
        "strain_set_form": "the sequences of strain.fasta, each on its lexicographically smaller strand, sorted, joined by newlines",
        "base_run_files_sha256": dict(base), "base_run": "%s/%d" % base_key,
        "runs": sorted("%s/%d" % k for k in runs),
        "digest_form": "tests/golden/make_graph_golden.collect (GFA / FASTA sequences as digests), then SHA-256 of that text",
        "differs_between_in_edge_models": sorted(f for f in names_all if any(
            (("plain", h) in runs and ("rotate", h) in runs and runs[("plain", h)].get(f) != runs[("rotate", h)].get(f)) for h in ran_seeds)),
        "differs_between_hash_seeds": sorted(f for f in names_all if any(
            ((m, h) in runs and (m, ran_seeds[0]) in runs and runs[(m, h)].get(f) != runs[(m, ran_seeds[0])].get(f)) for m in ran_models for h in ran_seeds)),
        "reference_stage_seconds": {"%s/%d" % k: timings["%s/%d" % k].get("after_pe_to_run_return_s") for k in sorted(runs)},
        "produced_by": "the real reference command (/root/reference/vstrains) behind tests/golden/gt_standin, run by tools/time_reference_stages.py in "
                       "the build container; files_sha256 holds the files all listed runs agree on, base_run_files_sha256 every file of the base run"}
    old = {}
    if os.path.exists(digests_path):
        with open(digests_path) as fh:
            old = json.load(fh)
    old["configs[%d]_whole_command" % meta["config"]] = entry
    with open(digests_path, "w") as fh:
        json.dump(old, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("written", digests_path, "stable under the base model", len(stable), "under both", len(stable_all), "of", len(names_all))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True)
    ap.add_argument("--pairs", type=int, default=200_000)
    ap.add_argument("--work", required=True)
    ap.add_argument("--models", default="rotate,plain")
    ap.add_argument("--seeds", default="0,1,2,3")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--out", default=None)
    ap.add_argument("--digests", default=os.path.join(ROOT, "tests", "golden", "reference_digests.json"))
    ap.add_argument("--collect-only", action="store_true")
    args = ap.parse_args()
    out = args.out or os.path.join(ROOT, "profiles", "r6", "reference_stages_config%d.json" % args.config)
    models = args.models.split(",")
    seeds = [int(x) for x in args.seeds.split(",")]
    meta = prepare(args.config, args.pairs, args.work)
    print(json.dumps(meta), flush=True)
    if not args.collect_only:
        with cf.ThreadPoolExecutor(args.jobs) as pool:
            futs = [pool.submit(one_run, args.work, m, h) for h in seeds for m in models]
            for f in cf.as_completed(futs):
                print("finished", f.result(), flush=True)
    collect(args.work, meta, ["rotate", "plain"], [0, 1, 2, 3], out, args.digests)  # (whatever runs the work directory holds)


if __name__ == "__main__":
    main()
