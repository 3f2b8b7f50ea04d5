"""String-level model of the device algorithm (seeds at stride s from the end's phase, canonical w-mer table, exact
extension, first-seed-owns-the-match rule).  It exists to pin the *algorithm* of
vstrains_amd/csrc/vs_pe.hip against the oracle on the CPU, where no GPU is available; the HIP
kernels themselves are checked against the oracle in the -m gpu tests."""
from typing import Dict, List, Sequence, Tuple

_C = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rc(s: str) -> str:
    return "".join(_C[c] for c in reversed(s))


def geometry(K: int) -> Tuple[int, int]:
    w = min(K, 31)
    if w % 2 == 0:
        w -= 1
    return w, K - w + 1


def build(seqs: Sequence[str], K: int):
    w, s = geometry(K)
    table: Dict[str, List[Tuple[int, int, int]]] = {}
    for i, seq in enumerate(seqs):
        if len(seq) < K:
            continue
        for p in range(len(seq) - w + 1):
            f = seq[p : p + w]
            r = rc(f)
            key, strand = (f, 0) if f < r else (r, 1)
            table.setdefault(key, []).append((i, p, strand))
    return table, w, s


def phase(rlen: int, w: int, s: int) -> int:
    """First probe offset of an end (vs_seed_phase in csrc/vs_internal.h): any phase of the grid is exact; this one gives
    the fewest probes, floor((rlen - w + 1) / s)."""
    return ((rlen - w) % s + s) // 2 if rlen >= w else 0


def valid(ch: str) -> bool:
    return ch in _C


def step_grid(rlen: int, w: int, s: int, t: int) -> List[int]:
    """The adaptive grid of the compile-time-shape kernels (round 5): grid positions s-1, 2s-1, ... with the positions from
    the t-th on moved D = s - 2 - (rlen - w) mod s bases towards the read's start (t = 0: all of them, the lowest exact phase;
    t = n: none, the highest).  Every such grid starts within the first s offsets, ends within the last s, and has no gap
    wider than s -- so every match of K bases holds one of its points; which t an end gets (the one with the fewest
    postings) changes the work, not the result."""
    m = rlen - w
    n = max(1, (m + 1) // s)
    D = max(0, s - 2 - m % s)
    return [s - 1 + i * s - (D if i >= t else 0) for i in range(n)]


def map_end(read: str, seqs: Sequence[str], rcs: Sequence[str], table, w: int, s: int, K: int, first=None, probes=None, grid=None) -> List[int]:
    rlen = len(read)
    agg: Dict[int, List[int]] = {}
    if grid is not None:  # explicit probe offsets: a match is credited by the first of them inside it (left extension < gap)
        todo = [(j, (j - grid[i - 1]) if i else s) for i, j in enumerate(grid) if j + w <= rlen]
    else:
        j = phase(rlen, w, s) if first is None else first
        todo = []
        while j + w <= rlen:
            todo.append((j, s))
            j += s
    for j, gap in todo:
        if probes is not None:
            probes.append(j)
        f = read[j : j + w]
        if all(valid(c) for c in f):
            r = rc(f)
            key, sr = (f, 0) if f < r else (r, 1)
            for node, p, sn in table.get(key, ()):
                opp = sn ^ sr
                text = rcs[node] if opp else seqs[node]
                tlen = len(text)
                q = tlen - p - w if opp else p
                c = min(gap, j, q)
                left = 0
                while left < c and read[j - 1 - left] == text[q - 1 - left]:
                    left += 1
                if left >= gap:
                    continue
                ext = 0
                while j + w + ext < rlen and q + w + ext < tlen and read[j + w + ext] == text[q + w + ext]:
                    ext += 1
                ln = left + w + ext
                if ln < K:
                    continue
                a = j - left
                qa = q - left
                minp = tlen - qa - ln if opp else qa
                rec = agg.setdefault(node, [0, minp, a])
                rec[0] += ln - K + 1
                rec[1] = min(rec[1], minp)
                rec[2] = min(rec[2], a)
    keep = []
    for node in sorted(agg):
        v, c, ki = agg[node]
        nlen = len(seqs[node])
        right = min(c + nlen - 1, c - ki + rlen - 1)
        saturate = right - c - K + 2
        if v >= saturate or v * rlen >= (min(rlen, nlen) - K + 1) * (rlen - K):
            keep.append(node)
    return keep

