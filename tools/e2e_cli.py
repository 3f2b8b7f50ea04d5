#!/usr/bin/env python3
"""GPU-side: wall time of the whole vstrains-compatible command on a configs[2]-shaped input
(15-strain synthetic, ~4.5 k-node GFA) with M read pairs written as FASTQ text.

    python tools/e2e_cli.py [M [n_strains genome_len]]        (default 2,000,000 15 10800)
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from vstrains_amd import cli, pe as host, synth  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    n_strains = int(sys.argv[2]) if len(sys.argv) > 3 else 15
    genome_len = int(sys.argv[3]) if len(sys.argv) > 3 else 10800
    L = 150
    tmp = tempfile.mkdtemp(prefix="vstrains_e2e_")
    t0 = time.time()
    pc = synth.make_pipeline_case(n_strains=n_strains, genome_len=genome_len, snp_rate=0.09, k=55, n_pairs=0, read_len=L, seed=1003)
    st = pc.strains
    ab = np.array(st.abundance)
    cum = np.minimum(np.floor(np.cumsum(ab) / ab.sum() * 2 ** 32), 2 ** 32 - 1).astype(np.uint32)
    cum[-1] = 0xFFFFFFFF
    paths = {"gfa": os.path.join(tmp, "graph.gfa"), "paths": os.path.join(tmp, "contigs.paths")}
    open(paths["gfa"], "w").write(pc.gfa_text)
    open(paths["paths"], "w").write(pc.paths_text)
    qual = b"I" * L
    for tag in ("f", "r"):
        paths[tag] = os.path.join(tmp, "reads_%s.fq" % tag)
    CH = 500_000
    ctx = host.Context(0)  # the device generator writes the reads (vs_synth_pairs + vs_reads_unpack), as in bench.py
    with open(paths["f"], "wb") as ff, open(paths["r"], "wb") as fr:
        for first in range(0, M, CH):
            n = min(CH, M - first)
            block = ctx.synth_pairs(st.genomes, cum, 20250001, first, n, L, int(0.005 * 2 ** 32), int(0.001 * 2 ** 32))
            text, lens, flags = block.unpack()
            block.free()
            text = text.reshape(n, 2, L).copy()
            text[(flags.reshape(n, 2) & 1).astype(bool), 0] = ord("N")
            fw, rv = text[:, 0], text[:, 1]
            ff.write(b"".join(b"@f%d\n%s\n+\n%s\n" % (first + i, fw[i].tobytes(), qual) for i in range(n)))
            fr.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (first + i, rv[i].tobytes(), qual) for i in range(n)))
    del ctx
    prep_s = time.time() - t0
    out = os.path.join(tmp, "out")
    t1 = time.time()
    timings = cli.main(["-a", "spades", "-g", paths["gfa"], "-p", paths["paths"], "-o", out, "-fwd", paths["f"], "-rve", paths["r"]])
    wall = time.time() - t1
    n_out = open(os.path.join(out, "strain.paths")).read().count("NODE_")
    print(json.dumps({"pairs": M, "fastq_bytes": os.path.getsize(paths["f"]) + os.path.getsize(paths["r"]),
                      "input_generation_s": prep_s, "cli_wall_s": wall, "stages": timings, "strains": n_out,
                      "gfa_nodes": len(pc.graph.ids)}))


if __name__ == "__main__":
    main()
