"""A locality-preserving order of the graph's nodes, used only for the device's INTERNAL numbering.

``node_mat`` / ``short_mat`` (``VStrains_PE_Inference.py:139-140``) are indexed by the position of a node in the GFA,
which an assembler assigns arbitrarily.  The cells one read pair increments are (nodes under the forward read) x (nodes
under the reverse read): neighbours along the genome.  If neighbours along the genome are neighbours in the numbering,
those cells share cache lines and the counter kernel's atomics stay in L2; with the GFA's numbering every increment
is its own 32-byte read-modify-write in HBM (configs[4]: 1.2e9 of them, 45 ms).  Counts are order-independent
sums, so the numbering never changes a result -- the host maps back (``vstrains_amd.pe``).

The order: nodes are laid out on a line by following k-base overlaps (suffix of one oriented node = prefix of the
next, either strand, as the assembler's L lines would say -- derived from the text because PE inference is handed
the S lines only, ``PE_Inference.py:106-114``); a node that follows another starts ``len - k`` bases further.
Breadth-first from the first unplaced node, first visit wins; components one after the other.
"""
from collections import deque
from typing import List, Sequence

_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def _revcomp(s: str) -> str:
    return s.translate(_COMP)[::-1]


def locality_order(seqs: Sequence[str], k: int) -> List[int]:
    """Permutation ``order`` with ``order[internal] = position in seqs``."""
    n = len(seqs)
    rc = [_revcomp(s) for s in seqs]
    heads = {}
    for i, s in enumerate(seqs):
        if len(s) < k or k <= 0:
            continue
        heads.setdefault(s[:k], []).append(2 * i)
        heads.setdefault(rc[i][:k], []).append(2 * i + 1)
    coord = [0] * n
    flip = [0] * n
    comp = [-1] * n
    for start in range(n):
        if comp[start] >= 0:
            continue
        comp[start] = start
        queue = deque([start])
        while queue:
            i = queue.popleft()
            x = coord[i]
            fw, bw = (seqs[i], rc[i]) if not flip[i] else (rc[i], seqs[i])
            if len(fw) < k or k <= 0:
                continue
            for t in heads.get(fw[-k:], ()):  # what follows this node as laid out
                j = t >> 1
                if comp[j] < 0:
                    comp[j], coord[j], flip[j] = start, x + len(fw) - k, t & 1
                    queue.append(j)
            for t in heads.get(bw[-k:], ()):  # what follows its reverse complement = what precedes it
                j = t >> 1
                if comp[j] < 0:
                    comp[j], coord[j], flip[j] = start, x - (len(seqs[j]) - k), (t & 1) ^ 1
                    queue.append(j)
    return sorted(range(n), key=lambda i: (comp[i], coord[i], i))


def path_order(seqs: Sequence[str], k: int) -> List[int]:
    """Depth-first variant: a node is followed by one of its successors, so the numbering runs along paths
    (the first one a whole walk through the component, the later ones the branches it left out, each next to
    where it rejoins).  Pairs sorted by the number of the node their read starts in then meet the pairs of the
    NEXT node of the same path, whose cells are mostly the same."""
    n = len(seqs)
    rc = [_revcomp(s) for s in seqs]
    heads = {}
    for i, s in enumerate(seqs):
        if len(s) < k or k <= 0:
            continue
        heads.setdefault(s[:k], []).append(2 * i)
        heads.setdefault(rc[i][:k], []).append(2 * i + 1)
    seen = [False] * n
    order: List[int] = []
    for start in range(n):
        if seen[start]:
            continue
        stack = [2 * start]
        while stack:
            t = stack.pop()
            i = t >> 1
            if seen[i]:
                continue
            seen[i] = True
            order.append(i)
            fw, bw = (seqs[i], rc[i]) if not (t & 1) else (rc[i], seqs[i])
            if len(fw) < k or k <= 0:
                continue
            # what precedes goes under what follows: the walk continues forwards first
            for u in reversed(heads.get(bw[-k:], ())):
                if not seen[u >> 1]:
                    stack.append(u ^ 1)
            for u in reversed(heads.get(fw[-k:], ())):
                if not seen[u >> 1]:
                    stack.append(u)
    return order
