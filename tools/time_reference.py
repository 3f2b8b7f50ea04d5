#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path in the build container (it cannot travel to the GPU box), and
the C restatement on the same inputs on the same box, so that the two boxes can be related
(SURVEY.md 8d "CPU baseline").  Measurement tooling like bench.py's cpu_baseline leg: it may call
oracle/; nothing of the product imports it.

    python tools/time_reference.py [--configs 0,1,2] [--prefix 200000] [--out profiles/r4/cpu_reference.json]
                                   [--digests tests/golden/reference_digests.json]

Also writes the SHA-256 of the pe_info / st_info files the REAL script produced (and of the s_graph_L1.gfa it
read) as a small fixture: a `-m gpu` test regenerates the same pairs on the device, counts them with the HIP
path, formats the text with vs_write_matrix_text and must arrive at the same digests.

Per config: the bench workload of that BASELINE.json config (vstrains_amd.workloads) is written as
s_graph_L1.gfa + two FASTQ files (reads from the CPU twin of the device generator, so they are the
stream bench.py counts), then
  * the real /root/reference/utils/VStrains_PE_Inference.py runs on them as the reference's driver
    launches it (a subprocess, utils/VStrains_SPAdes.py:119-132), single-threaded as it is -- on
    all pairs of configs[0], on a prefix for the larger ones (its per-pair cost does not depend on
    the number of pairs; "Global time elapsed" minus the table build is what is extrapolated);
  * oracle/pe_oracle.c (1 thread) counts the same pairs; pe_info / st_info must be identical;
  * configs[0] only: the reference graph stages (the whole `vstrains` command minus the PE script)
    behind tests/golden/gt_standin (a model of graph-tool / gfapy, slower than the C++ library --
    labelled as such).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np  # noqa: E402


def write_fastq(path, arr, tag):
    L = arr.shape[1]
    qual = b"I" * L
    with open(path, "wb") as fh:
        for lo in range(0, arr.shape[0], 100000):
            fh.write(b"".join(b"@%s%d\n%s\n+\n%s\n" % (tag, i, arr[i].tobytes(), qual) for i in range(lo, min(arr.shape[0], lo + 100000))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="0,1,2")
    ap.add_argument("--prefix", type=int, default=200_000, help="pairs given to the reference for configs > 0")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r4", "cpu_reference.json"))
    ap.add_argument("--digests", default=os.path.join(ROOT, "tests", "golden", "reference_digests.json"))
    args = ap.parse_args()
    import hashlib

    def sha256_file(path):
        h = hashlib.sha256()
        with open(path, "rb") as fh:
            for chunk in iter(lambda: fh.read(1 << 24), b""):
                h.update(chunk)
        return h.hexdigest()

    digests = {}
    from oracle import pe_oracle, pe_oracle_c
    from vstrains_amd.workloads import CONFIGS, workload_for

    result = {"host": {"cpus": os.cpu_count(), "python": sys.version.split()[0], "numpy": np.__version__},
              "note": "measured in the build container; the reference is single-process, single-thread", "configs": {}}
    for ci in [int(x) for x in args.configs.split(",")]:
        cfg = CONFIGS[ci]
        work = tempfile.mkdtemp(prefix="vs_timeref_c%d_" % ci)
        st, pre, names, seqs, cum, logger, n_in = workload_for(ci, work)
        gfa = os.path.join(work, "gfa", "s_graph_L1.gfa")
        n_pairs = cfg["total_pairs"] if ci == 0 else min(args.prefix, cfg["total_pairs"])
        L, k = cfg["read_len"], cfg["k"]
        seed = 20250000 + ci
        fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, seed, 0, n_pairs, L, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
        fq1, fq2 = os.path.join(work, "fwd.fq"), os.path.join(work, "rve.fq")
        write_fastq(fq1, fw, b"f")
        write_fastq(fq2, rv, b"r")
        entry = {"nodes": len(seqs), "pairs": int(n_pairs), "read_len": L, "k": k, "workload": cfg["tag"]}
        # ---- the reference script, launched the way its driver launches it
        out_dir = os.path.join(work, "aln_ref")
        t0 = time.perf_counter()
        proc = subprocess.run([sys.executable, os.path.join(REF, "utils", "VStrains_PE_Inference.py"), "-g", gfa, "-o", out_dir,
                               "-f", fq1, "-r", fq2, "-k", str(k)], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        assert proc.returncode == 0, proc.stderr[-2000:]
        glob_s = [float(l.split()[-1]) for l in proc.stdout.splitlines() if l.startswith("Global time elapsed")][0]
        digests["configs[%d]" % ci] = {
            "workload": cfg["tag"], "nodes": len(seqs), "pairs": int(n_pairs), "read_len": L, "k": k, "stream_seed": seed,
            "sub_thresh": int(0.005 * 2 ** 32), "n_thresh": int(0.001 * 2 ** 32),
            "s_graph_L1_gfa_sha256": sha256_file(gfa),
            "pe_info_sha256": sha256_file(os.path.join(out_dir, "pe_info")), "st_info_sha256": sha256_file(os.path.join(out_dir, "st_info")),
            "pe_info_bytes": os.path.getsize(os.path.join(out_dir, "pe_info")), "st_info_bytes": os.path.getsize(os.path.join(out_dir, "st_info")),
            "produced_by": "utils/VStrains_PE_Inference.py (the real reference script, PE_Inference.py:190-207) run by tools/time_reference.py "
                           "in the build container on the first `pairs` pairs of the bench stream of this config"}
        # its table build: time an empty-reads run of the same script on the same graph
        empty = os.path.join(work, "empty.fq")
        open(empty, "w").close()
        proc0 = subprocess.run([sys.executable, os.path.join(REF, "utils", "VStrains_PE_Inference.py"), "-g", gfa, "-o",
                                os.path.join(work, "aln_ref0"), "-f", empty, "-r", empty, "-k", str(k)], capture_output=True, text=True)
        assert proc0.returncode == 0, proc0.stderr[-2000:]
        base_s = [float(l.split()[-1]) for l in proc0.stdout.splitlines() if l.startswith("Global time elapsed")][0]
        entry["reference"] = {"wall_s": wall, "global_time_elapsed_s": glob_s, "table_build_and_output_s": base_s,
                              "pair_loop_s": glob_s - base_s, "pairs_per_s": n_pairs / max(glob_s - base_s, 1e-9),
                              "full_config_estimate_s": base_s + (glob_s - base_s) * cfg["total_pairs"] / n_pairs}
        # ---- the C restatement on the same pairs, same box, 1 thread
        t0 = time.perf_counter()
        orc = pe_oracle_c.Oracle(seqs, k)
        build_s = time.perf_counter() - t0
        off = np.arange(n_pairs + 1, dtype=np.uint64) * np.uint64(L)
        t0 = time.perf_counter()
        node, short, stats = orc.count_pairs_raw(fw.reshape(-1), off, rv.reshape(-1), off, n_pairs)
        port_s = time.perf_counter() - t0
        ids, _ = pe_oracle.read_gfa_segments(gfa)
        same = (pe_oracle.matrix_text(ids, node) == open(os.path.join(out_dir, "pe_info")).read()
                and pe_oracle.matrix_text(ids, short) == open(os.path.join(out_dir, "st_info")).read())
        entry["port"] = {"table_build_s": build_s, "pair_loop_s": port_s, "pairs_per_s": n_pairs / port_s,
                         "identical_pe_info_st_info": bool(same)}
        entry["port_over_reference"] = entry["port"]["pairs_per_s"] / entry["reference"]["pairs_per_s"]
        # ---- configs[0]: the reference's WHOLE command behind the stand-in: timing, and the digests of what it wrote under
        # both in-edge models of the stand-in and hash seeds 0-3 (the reference iterates sets of contig names) -- the files
        # that come out the same every time are a fixture the `-m gpu` suite holds the CLI drop-in against
        if ci == 0:
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
            import make_graph_golden as gold

            runs = {}
            for model in ("rotate", "plain"):
                for hashseed in (0, 1, 2, 3):
                    env = dict(os.environ)
                    env["PYTHONPATH"] = os.path.join(ROOT, "tests", "golden", "gt_standin") + os.pathsep + env.get("PYTHONPATH", "")
                    env["PYTHONHASHSEED"] = str(hashseed)
                    env["GT_STANDIN_INEDGE"] = model
                    env["MPLBACKEND"] = "Agg"
                    out_v = os.path.join(work, "vstrains_out_%s_%d" % (model, hashseed))
                    t0 = time.perf_counter()
                    pv = subprocess.run([sys.executable, os.path.join(REF, "vstrains"), "-a", "spades", "-g", os.path.join(work, "input.gfa"),
                                         "-p", os.path.join(work, "input.paths"), "-o", out_v, "-fwd", fq1, "-rve", fq2],
                                        capture_output=True, text=True, env=env, cwd=work)
                    wall_v = time.perf_counter() - t0
                    if pv.returncode != 0:
                        entry["reference_whole_command"] = {"error": pv.stderr[-800:]}
                        break
                    files = {rel: hashlib.sha256(text.encode()).hexdigest() for rel, text in gold.collect(out_v).items()
                             if rel not in ("aln/pe_info", "aln/st_info", "vstrains.log.info")}
                    runs[(model, hashseed)] = files
                    if model == "rotate" and hashseed == 0:
                        strains = open(os.path.join(out_v, "strain.paths")).read().count("\n") // 2
                        entry["reference_whole_command"] = {
                            "wall_s": wall_v, "pe_subprocess_s": glob_s, "graph_stages_and_rest_s": wall_v - glob_s, "strains": strains,
                            "label": "graph stages behind tests/golden/gt_standin (pure-Python model of graph-tool/gfapy; slower than the C++ library)"}
            if len(runs) == 8:
                base = runs[("rotate", 0)]
                names_all = sorted(set().union(*[set(r) for r in runs.values()]))
                stable = [f for f in names_all if all(r.get(f) == base.get(f) for r in runs.values())]
                digests["configs[0]_whole_command"] = {
                    "workload": cfg["tag"], "pairs": int(n_pairs), "stream_seed": seed,
                    "files_sha256": {f: base[f] for f in stable},
                    "digest_form": "tests/golden/make_graph_golden.collect (GFA / FASTA sequences as digests), then SHA-256 of that text",
                    "differs_between_in_edge_models": sorted(f for f in names_all if any(runs[("plain", h)].get(f) != runs[("rotate", h)].get(f) for h in (0, 1, 2, 3))),
                    "differs_between_hash_seeds": sorted(f for f in names_all if any(runs[(m, h)].get(f) != runs[(m, 0)].get(f) for m in ("rotate", "plain") for h in (1, 2, 3))),
                    "produced_by": "the real reference command (/root/reference/vstrains) behind tests/golden/gt_standin, in-edge models rotate and plain, "
                                   "PYTHONHASHSEED 0-3, run by tools/time_reference.py in the build container; files_sha256 holds the files all eight runs agree on"}
        result["configs"]["configs[%d]" % ci] = entry
        print(json.dumps({("configs[%d]" % ci): entry}, indent=1), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(result, fh, indent=1, sort_keys=True)
    print("written", args.out)
    if digests:
        old = {}
        if os.path.exists(args.digests):
            with open(args.digests) as fh:
                old = json.load(fh)
        old.update(digests)
        with open(args.digests, "w") as fh:
            json.dump(old, fh, indent=1, sort_keys=True)
            fh.write("\n")
        print("written", args.digests)


if __name__ == "__main__":
    main()
