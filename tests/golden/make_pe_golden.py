#!/usr/bin/env python3
"""Generate the PE-inference golden vectors under tests/golden/pe/.

Runs ONLY in the build container: it invents small inputs (seeded) and executes the real
reference script ``/root/reference/utils/VStrains_PE_Inference.py`` on them as a subprocess,
exactly as the reference's driver does (``utils/VStrains_SPAdes.py:119-132``).  What is
committed is data: inputs (graph.gfa, fwd.fq, rve.fq), the reference's outputs (pe_info,
st_info) and meta.json (k, exit code, last stderr line).  No reference source is copied.

    python tests/golden/make_pe_golden.py            # regenerate everything
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from vstrains_amd import synth  # noqa: E402

REF_SCRIPT = "/root/reference/utils/VStrains_PE_Inference.py"
OUT = os.path.join(HERE, "pe")


def emit(case, gfa_text, fwd_text, rve_text, k, note):
    d = os.path.join(OUT, case)
    if os.path.isdir(d):
        shutil.rmtree(d)
    os.makedirs(d)
    for name, text in (("graph.gfa", gfa_text), ("fwd.fq", fwd_text), ("rve.fq", rve_text)):
        with open(os.path.join(d, name), "w", newline="", encoding="utf-8") as fh:
            fh.write(text)
    with tempfile.TemporaryDirectory() as tmp:
        aln = os.path.join(tmp, "aln")
        proc = subprocess.run(
            [sys.executable, REF_SCRIPT, "-g", os.path.join(d, "graph.gfa"), "-o", aln,
             "-f", os.path.join(d, "fwd.fq"), "-r", os.path.join(d, "rve.fq"), "-k", str(k)],
            capture_output=True, text=True)
        meta = {"k": k, "returncode": proc.returncode, "note": note}
        if proc.returncode == 0:
            shutil.copy(os.path.join(aln, "pe_info"), os.path.join(d, "pe_info"))
            shutil.copy(os.path.join(aln, "st_info"), os.path.join(d, "st_info"))
            progress = [l for l in proc.stdout.splitlines() if l.startswith("Number of processed reads")]
            meta["progress_lines"] = progress
        else:
            err = [l for l in proc.stderr.strip().splitlines() if l.strip()]
            meta["stderr_last"] = err[-1] if err else ""
    with open(os.path.join(d, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("%-28s rc=%d" % (case, meta["returncode"]))


def simple_graph(seqs, k, ids=None, dp=None):
    ids = ids or [str(i) for i in range(len(seqs))]
    dp = dp or [10.0 + i for i in range(len(seqs))]
    lines = ["S\t%s\t%s\tDP:f:%s\n" % (i, s, repr(d)) for i, s, d in zip(ids, seqs, dp)]
    return "".join(lines)


def rand_seq(rng, n):
    return synth.ALPHABET[rng.integers(0, 4, size=n)].tobytes().decode()


def main():
    os.makedirs(OUT, exist_ok=True)

    # 1. two-strain bubble graph, both strands, error free
    st = synth.make_strains(3, 700, 0.02, seed=11)
    g = synth.compact_dbg(st, 21)
    f, r = synth.sample_pairs(st, 300, 80, seed=12)
    emit("bubbles_k21", g.gfa_text(), synth.fastq_text(f, "f"), synth.fastq_text(r, "r"), 21,
         "3 strains, compacted DBG k=21, 300 error-free 2x80 pairs, random strand")

    # 2. sequencing errors, N reads, short reads, lower case / IUPAC bytes in reads
    f, r = synth.sample_pairs(st, 300, 80, seed=13, sub_rate=0.02, n_rate=0.05)
    f[5] = f[5][:15]                      # shorter than k+1 -> pair dropped as short
    r[9] = r[9][:21]                      # exactly k   -> short
    r[10] = r[10][:22]                    # exactly k+1 -> used (one window)
    f[20] = f[20][:30] + f[20][30:40].lower() + f[20][40:]   # lower case run: windows miss
    r[21] = r[21][:50] + "R" + r[21][51:]                    # IUPAC byte: windows miss
    f[22] = f[22][:10] + "n" + f[22][11:]                    # lower-case n is NOT an N read
    f[23] = "N" + f[23][1:]
    r[24] = r[24][:-1] + "N"
    emit("errors_k21", g.gfa_text(), synth.fastq_text(f, "f"), synth.fastq_text(r, "r"), 21,
         "2% substitutions, 5% N pairs, short reads, lower-case and IUPAC bytes")

    # 3. palindromic (k+1)-mers: split_len even (k=5 -> 6).  ACGCGT, AATT.. are own revcomps
    rng = np.random.default_rng(31)
    n0 = "TTGACGCGTCAAGG" + rand_seq(rng, 20)          # ACGCGT palindrome inside
    n1 = rand_seq(rng, 12) + "GAATTC" + rand_seq(rng, 9) + "GAATTC"  # palindrome twice
    n2 = rand_seq(rng, 30)
    n3 = "ACGT"                                         # shorter than split_len: no entries
    seqs = [n0, n1, n2, n3]
    reads_f, reads_r = [], []
    for i in range(120):
        src = seqs[int(rng.integers(0, 3))]
        L = int(rng.integers(8, 18))
        a = int(rng.integers(0, max(1, len(src) - L)))
        s = src[a:a + L]
        src2 = seqs[int(rng.integers(0, 3))]
        b = int(rng.integers(0, max(1, len(src2) - L)))
        t = src2[b:b + L]
        if rng.random() < 0.5:
            s = synth.revcomp(s)
        if rng.random() < 0.5:
            t = synth.revcomp(t)
        reads_f.append(s)
        reads_r.append(t)
    reads_f.append("ACGCGT")          # the palindrome alone
    reads_r.append("GAATTCGAATTC")    # palindrome, partly tandem
    emit("palindrome_k5", simple_graph(seqs, 5), synth.fastq_text(reads_f, "f"),
         synth.fastq_text(reads_r, "r"), 5, "even split_len=6 with palindromic windows (double entries)")
    emit("odd_split_k6", simple_graph(seqs, 6), synth.fastq_text(reads_f, "f"),
         synth.fastq_text(reads_r, "r"), 6, "same inputs, split_len=7")

    # 4. repeats: tandem repeats inside a node, the same stretch in several nodes, homopolymers
    rng = np.random.default_rng(41)
    core = rand_seq(rng, 25)
    seqs = [
        rand_seq(rng, 10) + core + rand_seq(rng, 10),
        core + rand_seq(rng, 15) + core,
        "A" * 40,
        "ACACACACACACACACACACACACACACAC",
        rand_seq(rng, 18) + "T" * 22,
        synth.revcomp(core) + rand_seq(rng, 12),
    ]
    reads_f, reads_r = [], []
    for i in range(200):
        L = int(rng.integers(12, 45))
        src = seqs[int(rng.integers(0, len(seqs)))]
        a = int(rng.integers(0, max(1, len(src) - L + 1)))
        s = src[a:a + L]
        src2 = seqs[int(rng.integers(0, len(seqs)))]
        b = int(rng.integers(0, max(1, len(src2) - L + 1)))
        t = src2[b:b + L]
        if rng.random() < 0.5:
            s = synth.revcomp(s)
        if rng.random() < 0.5:
            t = synth.revcomp(t)
        reads_f.append(s)
        reads_r.append(t)
    reads_f += ["A" * 30, "T" * 35, "ACACACACACACACAC", core, core + core]
    reads_r += ["T" * 30, "A" * 12, "GTGTGTGTGTGTGTGTGT", synth.revcomp(core), core[5:] + core[:5]]
    emit("repeats_k9", simple_graph(seqs, 9), synth.fastq_text(reads_f, "f"),
         synth.fastq_text(reads_r, "r"), 9, "tandem repeats, shared stretches, homopolymers")

    # 5. short nodes (also lower-case ones below split_len: legal) and non-integer ids
    rng = np.random.default_rng(51)
    seqs = [rand_seq(rng, 60), "acgtacgt", rand_seq(rng, 21), rand_seq(rng, 22), rand_seq(rng, 45), "NNNN"]
    ids = ["7&8*0", "x", "12*A", "A3", "3", "n"]
    reads_f, reads_r = [], []
    for i in range(150):
        L = int(rng.integers(22, 50))
        src = seqs[(0, 3, 4)[int(rng.integers(0, 3))]]
        a = int(rng.integers(0, max(1, len(src) - L + 1)))
        s = src[a:a + L]
        src2 = seqs[(0, 3, 4)[int(rng.integers(0, 3))]]
        b = int(rng.integers(0, max(1, len(src2) - L + 1)))
        t = synth.revcomp(src2[b:b + L])
        reads_f.append(s)
        reads_r.append(t)
    emit("short_nodes_ids_k21", simple_graph(seqs, 21, ids=ids), synth.fastq_text(reads_f, "f"),
         synth.fastq_text(reads_r, "r"), 21, "nodes below split_len, lower-case short node, string ids")

    # 6. unequal FASTQ lengths + trailing partial record + last line without newline
    f, r = synth.sample_pairs(st, 60, 80, seed=61)
    ftxt = synth.fastq_text(f, "f") + "@partial\nACGT\n"
    rtxt = synth.fastq_text(r[:50], "r")
    rtxt = rtxt[:-1]  # final quality line loses its newline (harmless: only line 2 is used)
    emit("unequal_fastq_k21", g.gfa_text(), ftxt, rtxt, 21, "60 vs 50 records, partial record, no final newline")

    # 6b. the very last *sequence* line has no newline -> reference chops a real base
    f1, r1 = f[:3], r[:3]
    ftxt = synth.fastq_text(f1, "f")
    rtxt = "".join("@r_%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(r1[:2]))
    rtxt += "@r_2\n%s\n+\n%s" % (r1[2], "I" * len(r1[2]))
    emit("no_final_newline_k21", g.gfa_text(), ftxt, rtxt, 21, "final line of rve.fq has no newline")

    # 7. CRLF FASTQ and CRLF GFA (text mode translates: same answer as LF)
    emit("crlf_k21", g.gfa_text().replace("\n", "\r\n"), synth.fastq_text(f, "f", "\r\n"),
         synth.fastq_text(r, "r", "\r\n"), 21, "CRLF line ends everywhere")

    # 8. node >= split_len with lower-case bases -> KeyError in the reference
    seqs = [rand_seq(np.random.default_rng(81), 40), "ACGTACGTACGTacgtACGTACGTACGTAC", rand_seq(np.random.default_rng(82), 40)]
    emit("lowercase_node_error_k21", simple_graph(seqs, 21), synth.fastq_text(f[:4], "f"),
         synth.fastq_text(r[:4], "r"), 21, "node with lower-case bases, len >= split_len: reference exits non-zero")

    # 9. empty FASTQs
    emit("empty_reads_k21", g.gfa_text(), "", "", 21, "no reads at all")

    # 10. realistic k=55 / 2x150
    st55 = synth.make_strains(4, 1500, 0.015, seed=101)
    g55 = synth.compact_dbg(st55, 55)
    f, r = synth.sample_pairs(st55, 250, 150, seed=102, sub_rate=0.005, n_rate=0.01)
    emit("hiv_like_k55", g55.gfa_text(), synth.fastq_text(f, "f"), synth.fastq_text(r, "r"), 55,
         "4 strains 1.5 kb, k=55, 250 pairs 2x150, 0.5% errors")

    # 11. k=127 / 2x250 and variable read lengths
    st127 = synth.make_strains(3, 2200, 0.01, seed=111)
    g127 = synth.compact_dbg(st127, 127)
    f, r = synth.sample_pairs(st127, 120, 250, seed=112, sub_rate=0.003)
    rng = np.random.default_rng(113)
    f = [s[: int(rng.integers(100, 251))] for s in f]
    r = [s[: int(rng.integers(100, 251))] for s in r]
    emit("sars_like_k127", g127.gfa_text(), synth.fastq_text(f, "f"), synth.fastq_text(r, "r"), 127,
         "3 strains 2.2 kb, k=127, trimmed reads of 100..250 bp")

    # 12. default -k (128) path with k large vs reads: everything short
    emit("all_short_k128", g127.gfa_text(), synth.fastq_text(f[:10], "f"), synth.fastq_text([s[:100] for s in r[:10]], "r"), 128,
         "k=128: rve reads are all shorter than split_len")

    # 13. k = 1, 2 (degenerate seeds)
    rng = np.random.default_rng(131)
    seqs = [rand_seq(rng, 12), rand_seq(rng, 9), "AT"]
    reads_f = [rand_seq(rng, int(rng.integers(2, 10))) for _ in range(40)]
    reads_r = [rand_seq(rng, int(rng.integers(2, 10))) for _ in range(40)]
    emit("tiny_k1", simple_graph(seqs, 1), synth.fastq_text(reads_f, "f"), synth.fastq_text(reads_r, "r"), 1, "split_len=2")
    emit("tiny_k2", simple_graph(seqs, 2), synth.fastq_text(reads_f, "f"), synth.fastq_text(reads_r, "r"), 2, "split_len=3")

    # 14. bytes >= 0x80 in sequence lines: the reference reads the files in text mode (PE_Inference.py:147-152), so a
    # valid UTF-8 sequence is ONE character of the read (counts once toward its length, every window over it misses,
    # the pair stays in use); an upper-case N next to it still drops the pair
    f, r = synth.sample_pairs(st, 40, 80, seed=141)
    f[0] = f[0][:30] + "\u00e9" + f[0][31:]                 # two bytes, one character, read length unchanged
    r[1] = "\u20ac" + r[1][1:]                              # three bytes at the start
    f[2] = f[2][:40] + "\U0001F9EC" + f[2][41:]             # four bytes
    r[3] = r[3][:25] + "\u00e9\u00e9" + r[3][27:]          # two characters in a row
    f[4] = f[4][:10] + "\u00e9" + f[4][11:50] + "N" + f[4][51:]   # with an N: pair dropped
    r[5] = r[5][:22] + "\u00df"                             # 23 characters: one window, missed
    # (a sequence line is never the last line of a file it is taken from: records are whole groups of four lines)
    ftxt = synth.fastq_text(f, "f").replace("@f_7\n", "@f_7 \u00e9tiquette\n")   # headers may hold anything that decodes
    rtxt = synth.fastq_text(r, "r")
    emit("utf8_reads_k21", g.gfa_text(), ftxt, rtxt, 21, "valid UTF-8 multi-byte characters inside sequence lines")


if __name__ == "__main__":
    main()
