"""String-level model of the device algorithm (seed at stride s, canonical w-mer table, exact
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


def valid(ch: str) -> bool:
    return ch in _C


def map_end(read: str, seqs: Sequence[str], rcs: Sequence[str], table, w: int, s: int, K: int) -> List[int]:
    rlen = len(read)
    agg: Dict[int, List[int]] = {}
    j = 0
    while j + w <= rlen:
        f = read[j : j + w]
        if all(valid(c) for c in f):
            r = rc(f)
            key, sr = (f, 0) if f < r else (r, 1)
            for node, p, sn in table.get(key, ()):
                opp = sn ^ sr
                text = rcs[node] if opp else seqs[node]
                tlen = len(text)
                q = tlen - p - w if opp else p
                c = min(s, j, q)
                left = 0
                while left < c and read[j - 1 - left] == text[q - 1 - left]:
                    left += 1
                if left >= s:
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
        j += s
    keep = []
    for node in sorted(agg):
        v, c, ki = agg[node]
        nlen = len(seqs[node])
        right = min(c + nlen - 1, c - ki + rlen - 1)
        saturate = right - c - K + 2
        if v >= saturate or v * rlen >= (min(rlen, nlen) - K + 1) * (rlen - K):
            keep.append(node)
    return keep


# ---- round 2: shared-context groups ------------------------------------------------------------
# The postings of one seed mostly describe the same stretch of sequence (the k-overlap a node shares
# with its neighbours), so the device compares the read with ONE text per side (the group's
# reference posting on that side) and decides the other postings from what the index knows about
# them: how far each agrees with the reference (lcp) and which base it holds where it stops
# agreeing.  Only a posting that keeps agreeing with the read beyond that point needs a comparison
# of its own.  Below: the same decisions on strings.  LCP_CAP models the width of the stored fields.
LCP_CAP = 255


def _canon_text(seq: str, p: int, strand: int, w: int):
    """text and seed offset of a posting in the orientation of the canonical seed"""
    return (rc(seq), len(seq) - p - w) if strand else (seq, p)


def build_groups(seqs: Sequence[str], K: int):
    """seed -> (ref_a, ref_b, members); member = (node, pos, strand, lcp_a, lcp_b, nb_a, nb_b);
    side a = towards lower offsets of the canonical-orientation text.  nb = the member's base just
    beyond its agreement with the reference ('' when the member ends there)."""
    table, w, s = build(seqs, K)
    groups = {}
    for key, posts in table.items():
        ctx = [_canon_text(seqs[n], p, sn, w) for n, p, sn in posts]
        da = [q for _, q in ctx]
        db = [len(t) - q - w for t, q in ctx]
        ra = max(range(len(posts)), key=lambda i: (da[i], -i))
        rb = max(range(len(posts)), key=lambda i: (db[i], -i))
        members = []
        for i, (n, p, sn) in enumerate(posts):
            t, q = ctx[i]
            ta, qa_ = ctx[ra]
            la = 0
            while la < min(da[i], da[ra]) and t[q - 1 - la] == ta[qa_ - 1 - la]:
                la += 1
            tb, qb_ = ctx[rb]
            lb = 0
            while lb < min(db[i], db[rb]) and t[q + w + lb] == tb[qb_ + w + lb]:
                lb += 1
            nba = t[q - 1 - la] if la < da[i] else ""
            nbb = t[q + w + lb] if lb < db[i] else ""
            members.append((n, p, sn, min(la, LCP_CAP), min(lb, LCP_CAP), nba, nbb))
        groups[key] = (ra, rb, members)
    return groups, w, s


def _limits(read: str, j: int, w: int):
    """bytes outside ACGT cut a read into segments: seed validity and the segment around it"""
    lo = 0
    for p in range(j - 1, -1, -1):
        if not valid(read[p]):
            lo = p + 1
            break
    hi = len(read)
    for p in range(j + w, len(read)):
        if not valid(read[p]):
            hi = p
            break
    return all(valid(c) for c in read[j:j + w]), lo, hi


def map_end_grouped(read: str, seqs: Sequence[str], rcs: Sequence[str], groups, w: int, s: int, K: int, stats=None) -> List[int]:
    rlen = len(read)
    agg: Dict[int, List[int]] = {}

    def credit(node, opp, tlen, q, j, left, ext):
        ln = left + w + ext
        if left >= s or ln < K:
            return
        a = j - left
        qa = q - left
        minp = tlen - qa - ln if opp else qa
        rec = agg.setdefault(node, [0, minp, a])
        rec[0] += ln - K + 1
        rec[1] = min(rec[1], minp)
        rec[2] = min(rec[2], a)

    def oriented(node, p, sn, sr):
        opp = sn ^ sr
        text = rcs[node] if opp else seqs[node]
        return opp, text, (len(text) - p - w if opp else p)

    def agree_left(text, q, j, lim):
        n = 0
        while n < lim and read[j - 1 - n] == text[q - 1 - n]:
            n += 1
        return n

    def agree_right(text, q, j, lim):
        n = 0
        while n < lim and read[j + w + n] == text[q + w + n]:
            n += 1
        return n

    j = 0
    while j + w <= rlen:
        ok, lo, hi = _limits(read, j, w)
        f = read[j:j + w]
        j0 = j
        j += s
        if not ok:
            continue
        j = j0
        r = rc(f)
        key, sr = (f, 0) if f < r else (r, 1)
        grp = groups.get(key)
        if grp is not None:
            ra, rb, members = grp
            cap_l, cap_r = min(s, j - lo), hi - j - w   # what the read allows on either side
            if len(members) == 1:
                n, p, sn = members[0][:3]
                opp, text, q = oriented(n, p, sn, sr)
                left = agree_left(text, q, j, min(cap_l, q))
                ext = agree_right(text, q, j, min(cap_r, len(text) - q - w))
                credit(n, opp, len(text), q, j, left, ext)
            else:
                # stage R: the read against the reference text of either side
                il, ir = (rb, ra) if sr else (ra, rb)
                _, tl, ql = oriented(*members[il][:3], sr)
                _, tr, qr = oriented(*members[ir][:3], sr)
                r_l = agree_left(tl, ql, j, min(cap_l, ql))
                r_r = agree_right(tr, qr, j, min(cap_r, len(tr) - qr - w))
                for (n, p, sn, la, lb, nba, nbb) in members:
                    opp, text, q = oriented(n, p, sn, sr)
                    dl, dr = q, len(text) - q - w
                    l_l, l_r = (lb, la) if sr else (la, lb)
                    nb_l = (_C[nbb] if nbb else "") if sr else nba
                    nb_r = (_C[nba] if nba else "") if sr else nbb
                    lim_l, lim_r = min(cap_l, dl), min(cap_r, dr)
                    # stage M
                    if r_l < l_l:
                        left, lk = r_l, True
                    elif r_l > l_l:
                        left, lk = l_l, l_l < LCP_CAP
                    else:
                        left, lk = l_l, True
                        if l_l >= LCP_CAP or (l_l < lim_l and read[j - 1 - l_l] == nb_l):
                            lk = False
                    if r_r < l_r:
                        ext, rk = r_r, True
                    elif r_r > l_r:
                        ext, rk = l_r, l_r < LCP_CAP
                    else:
                        ext, rk = l_r, True
                        if l_r >= LCP_CAP or (l_r < lim_r and read[j + w + l_r] == nb_r):
                            rk = False
                    if stats is not None:
                        stats["members"] = stats.get("members", 0) + 1
                    if lk and left >= s:
                        continue
                    if lk and rk:
                        assert left == agree_left(text, q, j, lim_l) and ext == agree_right(text, q, j, lim_r), (read, j, n, p)
                        credit(n, opp, len(text), q, j, left, ext)
                        continue
                    ub = (left if lk else lim_l) + w + (ext if rk else lim_r)
                    if ub < K:
                        continue
                    # stage X: this posting's own text
                    if stats is not None:
                        stats["own"] = stats.get("own", 0) + 1
                    credit(n, opp, len(text), q, j, agree_left(text, q, j, lim_l), agree_right(text, q, j, lim_r))
        j += s
    keep = []
    for node in sorted(agg):
        v, c, ki = agg[node]
        nlen = len(seqs[node])
        right = min(c + nlen - 1, c - ki + rlen - 1)
        saturate = right - c - K + 2
        if v >= saturate or v * rlen >= (min(rlen, nlen) - K + 1) * (rlen - K):
            keep.append(node)
    return keep
