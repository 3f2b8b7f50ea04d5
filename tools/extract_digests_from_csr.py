#!/usr/bin/env python3
"""Build container: the checker's Python statement of the graph stages (oracle/graph_stages -- its own graph container,
GFA reader / writer and contig bookkeeping; numpy for the three data-parallel operations) on a BASELINE config's graph
with the PE-link table of the config's whole per-GPU read block, taken as the CSR rows tools/dump_links_csr.py wrote on
the GPU box.  Commits the SHA-256 of every file the stages write and the strain records as the fixture the `-m gpu`
suite holds the DEVICE run to (tests/golden/extract_digests_config<i>.json): at 54 465 nodes the engine's decisions are
then compared with an independent statement of the reference, not with the engine itself (VERDICT r5 missing 3).

    python tools/extract_digests_from_csr.py --config 4 [--parts-dir gpurun_out] [--out tests/golden/extract_digests_config4.json]
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("VS_CHECK_UNTOUCHED", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def load_csr(parts_dir, config):
    with open(os.path.join(parts_dir, "links_c%d_meta.json" % config)) as fh:
        meta = json.load(fh)
    n = meta["nodes"]
    row_ptr = np.zeros(n + 1, dtype=np.uint64)
    cols, vals = [], []
    at = 0
    for p in range(meta["parts"]):
        z = np.load(os.path.join(parts_dir, "links_c%d_part%d.npz" % (config, p)))
        lo, hi = int(z["row_lo"]), int(z["row_hi"])
        assert lo == meta["row_cuts"][p] and hi == meta["row_cuts"][p + 1]
        row_ptr[lo:hi + 1] = z["row_ptr"].astype(np.uint64) + np.uint64(at)
        cols.append(z["col"])
        vals.append(z["val"])
        at += int(z["col"].shape[0])
    col = np.concatenate(cols).astype(np.uint32)
    val = np.concatenate(vals).astype(np.uint32)
    assert at == meta["nnz"] == int(row_ptr[-1])
    sha = hashlib.sha256(row_ptr.tobytes() + col.tobytes() + val.tobytes()).hexdigest()
    assert sha == meta["csr_sha256"], "the parts do not add up to the table the GPU box hashed"
    return meta, row_ptr, col, val.astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--parts-dir", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--work", default=None)
    args = ap.parse_args()
    import copy
    import tempfile

    import profile_extract_cpu as pec
    from oracle import graph_ops as chk
    from oracle.graph_stages.run import PythonStages
    from vstrains_amd.graph import pipeline
    from vstrains_amd.workloads import workload_for

    meta, row_ptr, col, val = load_csr(args.parts_dir, args.config)
    work = args.work or tempfile.mkdtemp(prefix="vs_csr_extract_")
    st, pre, names, seqs, cum, logger, _ = workload_for(args.config, os.path.join(work, "work"))
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == meta["names_sha256"], "another graph than the GPU box counted on"

    class Backend(PythonStages):
        def __init__(self):
            self.graph_ops = chk.NumpyGraphOps()

    table = chk.SparsePeLinks(names, row_ptr, col, val)
    out = os.path.join(work, "out")
    for sub in ("gfa", "tmp"):
        os.makedirs(os.path.join(out, sub), exist_ok=True)
    t0 = time.perf_counter()
    strains = pipeline.extract_strains(copy.deepcopy(pre), table, Backend(), logger, out)
    secs = time.perf_counter() - t0
    files = pec.digests(out)
    rec = {"config": args.config, "nodes": meta["nodes"], "pairs": meta["pairs"], "nnz": meta["nnz"], "csr_sha256": meta["csr_sha256"],
           "names_sha256": meta["names_sha256"], "files_sha256": files,
           "strains": {k: [list(r[0]), r[1], r[2]] for k, r in strains.items()},
           "python_stages_s": secs,
           "produced_by": "tools/extract_digests_from_csr.py in the build container: oracle/graph_stages (the checker's own graph container, GFA "
                          "reader / writer, contig bookkeeping; numpy device operations; oracle.graph_ops.SparsePeLinks over the CSR rows "
                          "tools/dump_links_csr.py wrote on the GPU box: the symmetrised link table of the config's whole per-GPU read block)"}
    outp = args.out or os.path.join(ROOT, "tests", "golden", "extract_digests_config%d.json" % args.config)
    with open(outp, "w") as fh:
        json.dump(rec, fh, indent=0, sort_keys=True)
        fh.write("\n")
    print("written", outp, len(files), "files", len(strains), "strains", "%.1f s" % secs)


if __name__ == "__main__":
    main()
