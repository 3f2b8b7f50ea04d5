#!/usr/bin/env python3
"""Build container only: per-end node lists of the REAL reference function at the graph size of BASELINE configs[4].

The reference's ``single_end_read_mapping`` (utils/VStrains_PE_Inference.py:16-48) is IMPORTED from /root/reference and
called, end by end, on the first ``--pairs`` pairs of the bench stream of a config (the CPU twin of the device generator:
the same pairs ``bench.py --config i`` counts) against the (k+1)-mer table of that config's ``s_graph_L1`` (54 465 nodes
at configs[4]), filled the way the script's ``main`` fills it (:116-135: every window under its own text and under its
reverse complement, forward offset both times).  The whole script cannot run there (its two N x N text files would hold
3e9 lines each); the function per end can.

Committed: ``tests/golden/pe_end_lists_config<i>.json`` -- stream parameters, the digest of the graph file, and the list the
reference returned for every end (node indices in GFA order).  ``tests/test_configs_gpu.py`` regenerates the pairs on the
device and holds ``vs_pe_map_ends`` to these lists.  (VERDICT r4 "Next" 1b.)

    python tests/golden/make_pe_end_lists.py [--config 4] [--pairs 1024]
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SCRIPT = "/root/reference/utils/VStrains_PE_Inference.py"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--pairs", type=int, default=1024)
    args = ap.parse_args()
    spec = importlib.util.spec_from_file_location("ref_pe_inference", REF_SCRIPT)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)  # (defines functions only; its main() runs under __main__)

    from oracle import pe_oracle_c  # the CPU twin of the read generator (and nothing else of the oracle)
    from vstrains_amd.workloads import CONFIGS, workload_for

    cfg = CONFIGS[args.config]
    work = tempfile.mkdtemp(prefix="vs_endlists_c%d_" % args.config)
    st, pre, names, seqs, cum, logger, _ = workload_for(args.config, work)
    gfa = os.path.join(work, "gfa", "s_graph_L1.gfa")
    # the node table exactly as the script reads it (:100-112) ...
    index2id, index2seq, index2seqlen = [], [], []
    with open(gfa) as fh:
        for line in fh:
            sp = line[:-1].split("\t")
            if sp[0] == "S":
                index2id.append(sp[1])
                index2seq.append(sp[2])
                index2seqlen.append(len(sp[2]))
    assert index2seq == list(seqs)
    # ... and its (k+1)-mer table (:116-135), with the reference's own reverse_seq
    split_len = cfg["k"] + 1
    t0 = time.time()
    table = {}
    for i, seq in enumerate(index2seq):
        for p in range(index2seqlen[i] - split_len + 1):
            kmer = seq[p:p + split_len]
            table.setdefault(kmer, []).append((i, p))
            table.setdefault(ref.reverse_seq(kmer), []).append((i, p))
    build_s = time.time() - t0
    seed = 20250000 + args.config
    sub, nth = int(0.005 * 2 ** 32), int(0.001 * 2 ** 32)
    fw, rv = pe_oracle_c.synth_pairs(st.genomes, cum, seed, 0, args.pairs, cfg["read_len"], sub, nth)
    lists = []
    t0 = time.time()
    for p in range(args.pairs):
        f, r = fw[p].tobytes().decode(), rv[p].tobytes().decode()
        # the pair loop's filters (:160-165): an N in either end, or an end shorter than k + 1, drops the pair
        if f.count("N") or r.count("N") or len(f) < split_len or len(r) < split_len:
            lists.append(None)
            lists.append(None)
            continue
        for s in (f, r):
            lists.append([int(x) for x in ref.single_end_read_mapping(s, table, index2seqlen, split_len, len(index2id))])
    map_s = time.time() - t0
    with open(gfa, "rb") as fh:
        gfa_sha = hashlib.sha256(fh.read()).hexdigest()
    out = {
        "config": args.config, "workload": cfg["tag"], "nodes": len(index2id), "k": cfg["k"], "read_len": cfg["read_len"],
        "stream_seed": seed, "sub_thresh": sub, "n_thresh": nth, "pairs": args.pairs, "s_graph_L1_gfa_sha256": gfa_sha,
        "lists": lists,
        "produced_by": "single_end_read_mapping imported from /root/reference/utils/VStrains_PE_Inference.py (:16-48), table filled as "
                       ":116-135, by tests/golden/make_pe_end_lists.py in the build container; null = pair dropped by :160-165",
        "seconds": {"table": round(build_s, 1), "mapping": round(map_s, 1)},
    }
    path = os.path.join(HERE, "pe_end_lists_config%d.json" % args.config)
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
        fh.write("\n")
    n_lists = [len(x) for x in lists if x is not None]
    print("written %s: %d ends mapped, %d dropped; nodes per end mean %.2f max %d; table %.0f s, mapping %.0f s" % (
        path, len(n_lists), len(lists) - len(n_lists), sum(n_lists) / max(len(n_lists), 1), max(n_lists), build_s, map_s))


if __name__ == "__main__":
    main()
