"""CPU restatement of VStrains' PE-link inference  --  TEST INFRASTRUCTURE ONLY.

This file is the *checker* for the HIP path.  It is never imported by the product
package (``vstrains_amd``); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it.

It restates, in plain Python, what ``utils/VStrains_PE_Inference.py`` of the reference
computes (citations are ``file:line`` into /root/reference):

* GFA ``S``-line scan                         -> PE_Inference.py:100-112
* (k+1)-mer multimap over both strands        -> PE_Inference.py:116-135
* per-read-end lookup + saturation test       -> PE_Inference.py:16-48
* FASTQ record slicing / pair filters         -> PE_Inference.py:146-165
* node_mat / short_mat updates                -> PE_Inference.py:174-188
* pe_info / st_info text                      -> PE_Inference.py:190-207
* process_pe_info fold                        -> VStrains_IO.py:598-627

Parity pin: ``tests/golden/pe/*`` hold inputs + outputs produced by running the real
reference script in the build container (generator: ``tests/golden/make_pe_golden.py``);
``tests/test_oracle_golden.py`` checks this restatement against every one of them.
"""
from __future__ import annotations

import io
import sys
from typing import Dict, List, Sequence, Tuple

import numpy as np

_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(kmer: str) -> str:
    """Reverse complement; raises KeyError(char) on anything outside ACGT exactly like
    the reference's dict lookup does (PE_Inference.py:9-13), scanning from the right."""
    out = []
    for ch in reversed(kmer):
        out.append(_COMPLEMENT[ch])
    return "".join(out)


def read_gfa_segments(gfa_path: str) -> Tuple[List[str], List[str]]:
    """S-lines in file order -> (ids, seqs).  Each line loses its last character before the
    tab split (PE_Inference.py:107), universal-newline text mode as in ``open(.., 'r')``."""
    ids: List[str] = []
    seqs: List[str] = []
    with open(gfa_path, "r") as fh:
        for line in fh:
            cols = line[:-1].split("\t")
            if cols[0] == "S":
                ids.append(cols[1])
                seqs.append(cols[2])
    return ids, seqs


def build_table(seqs: Sequence[str], split_len: int) -> Dict[str, List[Tuple[int, int]]]:
    """(k+1)-mer -> list of (node index, forward offset); each window is entered under its own
    text and under its reverse complement, so a palindromic window holds the pair twice
    (PE_Inference.py:117-135)."""
    table: Dict[str, List[Tuple[int, int]]] = {}
    for idx, seq in enumerate(seqs):
        for off in range(len(seq) - split_len + 1):
            word = seq[off : off + split_len]
            rc = revcomp(word)
            table.setdefault(word, []).append((idx, off))
            table.setdefault(rc, []).append((idx, off))
    return table


def map_read_end(
    read: str,
    table: Dict[str, List[Tuple[int, int]]],
    seqlens: Sequence[int],
    split_len: int,
) -> List[int]:
    """Node indices (ascending) that the read end "saturates" (PE_Inference.py:16-48).

    Sparse bookkeeping instead of the reference's three length-N arrays; the arithmetic per
    touched node is the reference's, including the float64 ``expected`` term."""
    rlen = len(read)
    hits: Dict[int, List[int]] = {}  # node -> [count, min node offset, min read offset]
    for i in range(rlen - split_len + 1):
        bucket = table.get(read[i : i + split_len])
        if bucket is None:
            continue
        for node, off in bucket:
            rec = hits.get(node)
            if rec is None:
                hits[node] = [1, off, i]
            else:
                rec[0] += 1
                if off < rec[1]:
                    rec[1] = off
                if i < rec[2]:
                    rec[2] = i
    keep: List[int] = []
    for node in sorted(hits):
        count, coord, kidx = hits[node]
        nlen = seqlens[node]
        left = max(coord, coord - kidx)
        right = min(coord + nlen - 1, coord - kidx + rlen - 1)
        saturate = right - left - (split_len - 1) + 1
        expected = (min(rlen, nlen) - split_len + 1) * (rlen - split_len) / rlen
        if count >= max(min(saturate, expected), 1):
            keep.append(node)
    return keep


def map_read_end_int(
    read: str,
    table: Dict[str, List[Tuple[int, int]]],
    seqlens: Sequence[int],
    split_len: int,
) -> List[int]:
    """Same as :func:`map_read_end` but with the all-integer form of the acceptance test that
    the C oracle and the HIP kernel use:  keep  <=>  count >= saturate  or
    count*rlen >= (min(rlen,nlen)-split_len+1)*(rlen-split_len).   (count >= 1 always.)"""
    rlen = len(read)
    hits: Dict[int, List[int]] = {}
    for i in range(rlen - split_len + 1):
        bucket = table.get(read[i : i + split_len])
        if bucket is None:
            continue
        for node, off in bucket:
            rec = hits.setdefault(node, [0, off, i])
            rec[0] += 1
            rec[1] = min(rec[1], off)
            rec[2] = min(rec[2], i)
    keep: List[int] = []
    for node in sorted(hits):
        count, coord, kidx = hits[node]
        nlen = seqlens[node]
        right = min(coord + nlen - 1, coord - kidx + rlen - 1)
        saturate = right - coord - split_len + 2
        if count >= saturate or count * rlen >= (min(rlen, nlen) - split_len + 1) * (rlen - split_len):
            keep.append(node)
    return keep


def fastq_sequences(path: str) -> List[str]:
    """Sequence line of every complete 4-line record, each minus its last character
    (PE_Inference.py:147-159: ``readlines()`` then ``s[:-1]``)."""
    with open(path, "r") as fh:
        lines = fh.readlines()
    return [lines[4 * r + 1][:-1] for r in range(len(lines) // 4)]


def pe_matrices(
    seqs: Sequence[str],
    fwd_reads: Sequence[str],
    rve_reads: Sequence[str],
    ksize: int,
    table: Dict[str, List[Tuple[int, int]]] | None = None,
    mapper=map_read_end,
):
    """-> (node_mat, short_mat, (n_reads, short_reads, used_reads)); PE_Inference.py:137-188."""
    split_len = ksize + 1
    if table is None:
        table = build_table(seqs, split_len)
    seqlens = [len(s) for s in seqs]
    n = len(seqs)
    node_mat = np.zeros((n, n), dtype=np.int64)
    short_mat = np.zeros((n, n), dtype=np.int64)
    n_reads = short_reads = used_reads = 0
    for r in range(min(len(fwd_reads), len(rve_reads))):
        fseq, rseq = fwd_reads[r], rve_reads[r]
        if fseq.count("N") or rseq.count("N"):
            n_reads += 1
            continue
        if len(fseq) < split_len or len(rseq) < split_len:
            short_reads += 1
            continue
        used_reads += 1
        lefts = mapper(fseq, table, seqlens, split_len)
        rights = mapper(rseq, table, seqlens, split_len)
        for side in (lefts, rights):
            for a in range(len(side)):
                for b in range(a, len(side)):
                    short_mat[side[a], side[b]] += 1
        for i in lefts:
            for j in rights:
                node_mat[i, j] += 1
    return node_mat, short_mat, (n_reads, short_reads, used_reads)


def matrix_text(ids: Sequence[str], mat: np.ndarray) -> str:
    """Dense row-major ``id_i:id_j:count`` lines, zeros included (PE_Inference.py:196-205)."""
    buf = io.StringIO()
    n = len(ids)
    for i in range(n):
        row = mat[i]
        for j in range(n):
            buf.write("%s:%s:%d\n" % (ids[i], ids[j], int(row[j])))
    return buf.getvalue()


def run_files(gfa: str, fwd: str, rve: str, ksize: int):
    """File-level restatement: -> (pe_info text, st_info text, stats)."""
    ids, seqs = read_gfa_segments(gfa)
    node_mat, short_mat, stats = pe_matrices(seqs, fastq_sequences(fwd), fastq_sequences(rve), ksize)
    return matrix_text(ids, node_mat), matrix_text(ids, short_mat), stats


def fold_pe_info(node_ids: Sequence[str], pe_text: str, st_text: str) -> Dict[Tuple[str, str], int]:
    """VStrains_IO.py:598-627: unordered-pair table keyed by (min_id, max_id) under *string*
    ordering; both files are added into it; a blank line stops a file."""
    table: Dict[Tuple[str, str], int] = {}
    for u in node_ids:
        for v in node_ids:
            table[(min(u, v), max(u, v))] = 0
    for text in (pe_text, st_text):
        for line in text.splitlines(keepends=True):
            if line == "\n":
                break
            u, v, mark = line[:-1].split(":")[:3]
            key = (min(u, v), max(u, v))
            if key in table:
                table[key] += int(mark)
    return table


if __name__ == "__main__":  # tiny manual driver: gfa fwd rve k
    pe, st, stats = run_files(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]))
    sys.stdout.write(pe)
    sys.stderr.write("stats n/short/used = %s\n" % (stats,))
