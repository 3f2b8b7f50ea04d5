// Walk index: what lets the mapping kernel FOLLOW a read through the graph instead of probing every
// stride-th offset against every node that shares a seed (vs_walk.hip), for node sets the build can
// certify as a proper overlap graph of (k+1)-mers:
//
//   (C1) every (k+1)-mer occurs once in the index, counting both strands of every node -- then a read
//        window has at most one home, and the reference's table (PE_Inference.py:116-135) holds exactly
//        one entry for it;
//   (C2) a k-mer that starts a (k+1)-mer somewhere INSIDE a node strand (position > 0) starts no other
//        (k+1)-mer of the index and ends no node strand -- then the window after a window that sits at
//        (node, q) can only sit at (node, q + 1);
//   (C3) where a node strand ends, the (k+1)-mers that start with its last k bases are first windows of
//        node strands (position 0): its successors, told apart by their base k (C1 makes them differ).
//
// Compacted de Bruijn graphs (SPAdes output, with or without nodes removed afterwards) have these
// properties; any other node set fails the certification and is mapped by the seed kernels.
// Under (C1)-(C3) the coincidences of PE_Inference.py:23-31 of one read form runs of consecutive
// windows; a run is found by ONE exact (k+1)-mer lookup and followed base by base: inside a node by
// comparing text, across a node end through succ[].  Results are the reference's, bit for bit.
//
// Shared by the host builder (vs_walk_host.cpp, plain C++) and the kernels.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VS_HD __host__ __device__ inline
#else
#define VS_HD inline
#endif

#define VS_WALK_MAX_NW 5u          // (k+1)-mers of up to 160 bases (five 64-bit words)
#define VS_WALK_EMPTY 0xFFFFFFFFu  // VsKSlot.ns of an empty slot
#define VS_PSET_EMPTY 0xFFFFFFFFFFFFFFFFull

// One node strand (ns = 2 * node + strand; strand 1 = reverse complement), 32 B = two dwordx4 loads.
struct VsWalkRec {
    uint32_t len;               // bases of the node
    uint32_t woff;              // first word of THIS strand's text in the index's text array
    uint32_t tail_lo, tail_hi;  // bases K .. K+31 of this strand (what a read is compared with right after entering at 0)
    uint32_t succ[4];           // by base code b: 1 + ns of the strand that starts with this strand's last k bases + b; 0 = none
};

// One slot of the (k+1)-mer table (both strands of every node are entered, so a read window is looked up
// as it is): 16 B.  Open addressing, slot = hash & mask, linear probing.
struct VsKSlot {
    uint32_t tag;   // hash >> 32
    uint32_t ns;    // node strand, VS_WALK_EMPTY = empty slot
    uint32_t pos;   // position of the (k+1)-mer in that strand
    uint32_t woff;  // first word of that strand's text (saves the record load before the verification)
};

struct VsWalkDev {
    const VsWalkRec *rec;   // [2 * n_nodes]
    const VsKSlot *ktab;    // [1 << k_bits]
    const uint64_t *pset;   // [1 << p_bits] canonical wp-mers that occur in some node of length >= K (presence only)
    uint32_t k_bits, p_bits;
    uint32_t wp;            // length of the presence mers: min(31, ceil((K - 1) / 2) + 1)
    uint32_t nw;            // 64-bit words of a (k+1)-mer
};

// hash of a (k+1)-mer given as nw 64-bit words (32 bases each, LSB first, the last one masked)
VS_HD uint64_t vs_kmer_hash_init(uint32_t K) { return 0x9E3779B97F4A7C15ull * (uint64_t)(K + 1u); }
VS_HD uint64_t vs_kmer_hash_step(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 32);
}
VS_HD uint64_t vs_kmer_hash_done(uint64_t h) {
    h *= 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 29);
}
VS_HD uint32_t vs_pset_slot(uint64_t key, uint32_t bits) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64u - bits)); }
VS_HD uint32_t vs_walk_wp(uint32_t K) {
    uint32_t wp = (K - 1u + 1u) / 2u + 1u;  // ceil((K - 1) / 2) + 1: one flank of any base inside a K-window holds that many bases
    if (wp > 31u) wp = 31u;
    if (wp > K) wp = K;
    return wp;
}


// ---- the walk of one read end, as steps (shared by k_pe_walk and by the host twin the CPU tests run) -------------------
// An end is always in one of three classes: L (window j needs a lookup), W (window j sits at (ns, q): follow the node) or
// P (a presence probe decides how far to skip).  The host twin takes the steps of one end one after the other; the
// kernel keeps the ends of a tile in LDS queues per class and gives every lane one step of SOME end per pass, so that
// lanes stay busy while the ends of a tile differ in how far they are (wavefront ballot + prefix sum compaction).
#if defined(__HIP_DEVICE_COMPILE__)
VS_HD uint64_t vsw_win(const uint32_t *w, uint32_t base) {  // 32 bases from base offset `base`: two funnel shifts
    const uint32_t i = base >> 4, sh = (base & 15u) * 2u;
    const uint32_t w0 = w[i], w1 = w[i + 1], w2 = w[i + 2];
    return (uint64_t)__builtin_amdgcn_alignbit(w1, w0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32);
}
#else
VS_HD uint64_t vsw_win(const uint32_t *w, uint32_t base) {
    const uint32_t i = base >> 4, sh = (base & 15u) * 2u;
    const uint64_t lo = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
    return sh ? (lo >> sh) | ((uint64_t)w[i + 2] << (64u - sh)) : lo;
}
#endif
VS_HD uint64_t vsw_lowmask(uint32_t bits) { return bits >= 64u ? ~0ull : ((1ull << bits) - 1ull); }
VS_HD uint32_t vsw_first_diff(uint64_t x) { return (uint32_t)__builtin_ctzll(x) >> 1; }  // x != 0: first differing base
VS_HD uint64_t vsw_rc(uint64_t x, uint32_t w) {  // reverse complement of a w-mer in the low 2w bits
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = __builtin_bswap64(x);
    return (~x) >> (64u - 2u * w);
}
VS_HD bool vsw_accept(uint32_t v, uint32_t coord, uint32_t kidx, uint32_t nlen, uint32_t rlen, uint32_t K) {
    // PE_Inference.py:36-47 in integers (oracle/pe_oracle.py: map_read_end_int); 32-bit arithmetic where nothing can overflow
    if (rlen < 32768u && nlen < (1u << 30)) {
        const int c = (int)coord, ki = (int)kidx, nl = (int)nlen, rl = (int)rlen, k = (int)K;
        int right = c + nl - 1;
        const int alt = c - ki + rl - 1;
        if (alt < right) right = alt;
        const int saturate = right - c - k + 2;
        const int span = (rl < nl ? rl : nl) - k + 1;
        return ((int)v >= saturate) || ((int)v * rl >= span * (rl - k));
    }
    const long long c = coord, ki = kidx, nl = nlen, rl = rlen, k = K;
    long long right = c + nl - 1;
    const long long alt = c - ki + rl - 1;
    if (alt < right) right = alt;
    const long long saturate = right - c - k + 2;
    const long long span = (rl < nl ? rl : nl) - k + 1;
    return ((long long)v >= saturate) || ((long long)v * rl >= span * (rl - k));
}
// smallest position >= from that holds a byte outside ACGT (inv4: up to four positions, 0xFF = none), else rlen
VS_HD uint32_t vsw_next_inv(uint32_t inv4, uint32_t from, uint32_t rlen) {
    uint32_t best = rlen;
    for (uint32_t i = 0; i < 4u; i++) {
        const uint32_t p = (inv4 >> (8u * i)) & 0xFFu;
        if (p != 0xFFu && p >= from && p < best) best = p;
    }
    return best;
}

enum { VSW_DONE = 0, VSW_L = 1, VSW_W = 2, VSW_P = 3 };
// probe modes of class P: the two flanks of a broken run's base p, or the skip search behind a missed lookup
enum { VSW_P_FLANKS = 0, VSW_P_SKIP = 1 };

struct VsWalkEnd {
    uint32_t j;                 // next unresolved window (L, P), or the window that sits at (ns, q) (W)
    uint32_t lo, hi;            // the clean stretch [lo, hi) of the read that j lies in
    uint32_t ns, q;             // W: node strand and position of window j
    uint32_t cur_node, cur_v, cur_coord, cur_kidx, cur_nlen;  // the visit being merged (cur_node = ~0: none)
    uint32_t nt;                // nodes written to the row so far (may pass the row's capacity: overflow)
    uint32_t pm;                // P: mode | d << 4 (skip search) or mode | p << 4 (flanks)
    uint32_t over;              // the end needs the general path
};

VS_HD void vsw_end_init(VsWalkEnd &e, uint32_t rlen, uint32_t inv4, bool dirty) {
    e.j = 0u; e.lo = 0u; e.hi = dirty ? vsw_next_inv(inv4, 0u, rlen) : rlen;
    e.ns = 0u; e.q = 0u;
    e.cur_node = 0xFFFFFFFFu; e.cur_v = e.cur_coord = e.cur_kidx = e.cur_nlen = 0u;
    e.nt = 0u; e.pm = 0u; e.over = 0u;
}
// the node of the current visit differs from the one held (or the end is finished): settle the one held
VS_HD void vsw_flush(VsWalkEnd &e, uint32_t *row, uint32_t cap, uint32_t rlen, uint32_t K) {
    if (e.cur_node == 0xFFFFFFFFu) return;
    for (uint32_t i = 0; i < e.nt && i < cap; i++)
        if ((row[i] & 0x7FFFFFFFu) == e.cur_node) e.over = 1u;  // met before, not just now: general path
    const uint32_t acc = vsw_accept(e.cur_v, e.cur_coord, e.cur_kidx, e.cur_nlen, rlen, K) ? 0x80000000u : 0u;
    if (e.nt < cap) row[e.nt] = e.cur_node | acc; else e.over = 1u;
    e.nt++;
    e.cur_node = 0xFFFFFFFFu;
}
// is the wp-mer at read offset x absent from every node?
VS_HD bool vsw_absent(const VsWalkDev &wk, const uint32_t *rw, uint32_t rbase, uint32_t x) {
    const uint64_t f = vsw_win(rw, rbase + x) & vsw_lowmask(2u * wk.wp);
    const uint64_t r = vsw_rc(f, wk.wp);
    const uint64_t key = r < f ? r : f;
    const uint32_t pmask = (1u << wk.p_bits) - 1u;
    uint32_t s = vs_pset_slot(key, wk.p_bits);
    for (;;) {
        const uint64_t got = wk.pset[s];
        if (got == key) return false;
        if (got == VS_PSET_EMPTY) return true;
        s = (s + 1u) & pmask;
    }
}
// Class L, also the start of every end.  Moves to the next clean stretch when this one holds no window any more; one exact
// lookup of window j: hit -> W at (ns, q), miss -> P (skip search).  Returns the next class.
template <uint32_t NW>
VS_HD uint32_t vsw_step_lookup(const VsWalkDev &wk, const uint32_t *text, uint32_t K, const uint32_t *rw, uint32_t rbase, uint32_t rlen,
                               uint32_t inv4, VsWalkEnd &e) {
    if (e.over) return VSW_DONE;
    while (e.j + K > e.hi) {  // no window left in this clean stretch
        if (e.hi >= rlen) return VSW_DONE;
        e.lo = e.hi + 1u;
        e.hi = vsw_next_inv(inv4, e.lo, rlen);
        e.j = e.lo;
    }
    uint64_t wv[NW];
    uint64_t h = vs_kmer_hash_init(K);
    for (uint32_t i = 0; i < NW; i++) {
        wv[i] = vsw_win(rw, rbase + e.j + 32u * i);
        if (i == NW - 1u) wv[i] &= vsw_lowmask(2u * K - 64u * (NW - 1u));
        h = vs_kmer_hash_step(h, wv[i]);
    }
    h = vs_kmer_hash_done(h);
    const uint32_t tag = (uint32_t)(h >> 32), kmask = (1u << wk.k_bits) - 1u;
    uint32_t s = (uint32_t)h & kmask;
    for (;;) {
        const VsKSlot sl = wk.ktab[s];
        if (sl.ns == VS_WALK_EMPTY) break;
        if (sl.tag == tag) {
            const uint32_t tb = sl.woff * 16u + sl.pos;
            bool same = true;
            for (uint32_t i = 0; i < NW; i++) {
                uint64_t tw = vsw_win(text, tb + 32u * i);
                if (i == NW - 1u) tw &= vsw_lowmask(2u * K - 64u * (NW - 1u));
                same = same && tw == wv[i];
            }
            if (same) {
                e.ns = sl.ns;
                e.q = sl.pos;
                return VSW_W;
            }
        }
        s = (s + 1u) & kmask;
    }
    e.pm = VSW_P_SKIP | ((K - wk.wp) << 4);  // window j coincides with nothing: how far can the scan skip?
    return VSW_P;
}
// Class W: window j sits at (ns, q).  Compares up to 32 further bases of the read with the node, credits the windows that
// coincide to the visit, and says where the end goes next: on in this node or into the successor the next base selects (W),
// run broken at base p = j + K (P: flanks), stretch used up (L: next stretch, or done).
VS_HD uint32_t vsw_step_walk(const VsWalkDev &wk, const uint32_t *text, uint32_t K, const uint32_t *rw, uint32_t rbase, uint32_t rlen,
                             VsWalkEnd &e, uint32_t *row, uint32_t cap) {
    const VsWalkRec r = wk.rec[e.ns];
    const uint32_t nlen = r.len, u = e.j, qu = e.q;
    const uint32_t in_node = nlen - K - e.q, in_read = e.hi - K - e.j;
    uint32_t m = in_node < in_read ? in_node : in_read;  // windows that may follow here
    const bool more = m > 32u;                           // (a long node: the rest in the next step)
    if (more) m = 32u;
    uint32_t adv = m;
    if (m) {
        const uint64_t tw = e.q == 0u ? ((uint64_t)r.tail_lo | ((uint64_t)r.tail_hi << 32)) : vsw_win(text, r.woff * 16u + e.q + K);
        const uint64_t x = (tw ^ vsw_win(rw, rbase + e.j + K)) & vsw_lowmask(2u * m);
        if (x) adv = vsw_first_diff(x);
    }
    const bool broke = adv < m;
    e.j += adv;
    e.q += adv;
    {   // windows u .. j sit in this node
        const uint32_t node = e.ns >> 1, add = e.j - u + 1u;
        const uint32_t coord = (e.ns & 1u) ? nlen - K - e.q : qu;
        if (node == e.cur_node) {
            // (a step that continues a visit counts its first window with the previous step: see `more` below)
            e.cur_v += add;
            e.cur_coord = coord < e.cur_coord ? coord : e.cur_coord;
        } else {
            vsw_flush(e, row, cap, rlen, K);
            e.cur_node = node; e.cur_v = add; e.cur_coord = coord; e.cur_kidx = u; e.cur_nlen = nlen;
        }
    }
    if (e.over) return VSW_DONE;
    if (!broke && more) {  // the same node goes on: window j is counted, the next step starts at j + 1
        // next step re-counts nothing: it starts from window j (already credited), so take its own first window off
        e.cur_v -= 1u;
        return VSW_W;
    }
    uint32_t nxt = 0u;
    if (!broke && e.j + K < e.hi) {  // the strand ends here and the read goes on: its next base picks the successor
        const uint32_t pb = rbase + e.j + K;
        const uint32_t b = (rw[pb >> 4] >> (2u * (pb & 15u))) & 3u;
        nxt = b == 0u ? r.succ[0] : b == 1u ? r.succ[1] : b == 2u ? r.succ[2] : r.succ[3];
    }
    if (nxt) {
        e.ns = nxt - 1u;
        e.q = 0u;
        e.j++;
        return VSW_W;
    }
    if (broke || e.j + K < e.hi) {
        // the run breaks at read base p = j + K: window j + 1 coincides with nothing (certified); whether any window over
        // p can coincide is what the flank probes decide
        e.pm = VSW_P_FLANKS | ((e.j + K) << 4);
        return VSW_P;
    }
    e.j++;  // the stretch is used up (j + K == hi): the lookup step moves on to the next stretch or ends
    return VSW_L;
}
// Class P: presence probes.  Flanks: every window over the broken run's base p holds the wp-mer that ends at p or the one
// that starts at p; if neither occurs in a node, none of them coincides and the scan resumes at p + 1, else at the window
// after the one the certification rules out.  Skip search (window j missed): every window that holds the wp-mer at x lies
// in [x - (K - wp), x]; with x = j + d absent, all of [j, j + d] are empty.
VS_HD uint32_t vsw_step_probe(const VsWalkDev &wk, uint32_t K, const uint32_t *rw, uint32_t rbase, VsWalkEnd &e) {
    const uint32_t wp = wk.wp;
    if ((e.pm & 15u) == VSW_P_FLANKS) {
        const uint32_t p = e.pm >> 4;  // = j + K
        const bool l_out = p + 1u < e.lo + wp, r_out = p + wp > e.hi;  // (a flank that leaves the stretch holds no window)
        const bool none = (l_out || vsw_absent(wk, rw, rbase, p + 1u - wp)) && (r_out || vsw_absent(wk, rw, rbase, p));
        e.j = none ? p + 1u : e.j + 2u;
        return VSW_L;
    }
    const uint32_t d = e.pm >> 4;
    if (vsw_absent(wk, rw, rbase, e.j + d)) {
        e.j += d + 1u;
        return VSW_L;
    }
    if (d == 0u) {
        e.j += 1u;
        return VSW_L;
    }
    e.pm = VSW_P_SKIP | ((d >> 1) << 4);
    return VSW_P;
}

// One read end from start to finish (the host twin; the kernel schedules the same steps per class): row[cap] receives
// node | accepted << 31 for every node the end touches; returns how many it touched, *over_out says whether the end needs
// the general path (a node met again later, more than cap nodes).
template <uint32_t NW>
VS_HD uint32_t vs_walk_end(const VsWalkDev &wk, const uint32_t *text, uint32_t K, const uint32_t *rw, uint32_t rbase, uint32_t rlen,
                           uint32_t inv4, bool dirty, uint32_t *row, uint32_t cap, bool *over_out) {
    VsWalkEnd e;
    vsw_end_init(e, rlen, inv4, dirty);
    uint32_t cls = VSW_L;
    while (cls != VSW_DONE) {
        if (cls == VSW_L) cls = vsw_step_lookup<NW>(wk, text, K, rw, rbase, rlen, inv4, e);
        else if (cls == VSW_W) cls = vsw_step_walk(wk, text, K, rw, rbase, rlen, e, row, cap);
        else cls = vsw_step_probe(wk, K, rw, rbase, e);
    }
    vsw_flush(e, row, cap, rlen, K);
    *over_out = e.over != 0u;
    return e.nt;
}

#if !defined(__HIPCC__) || defined(VS_WALK_HOST_DECL)
#include <string>
#include <vector>
struct VsWalkHost {
    bool certified = false;
    std::string why;  // first reason the certification failed
    uint32_t K = 0, wp = 0, nw = 0, k_bits = 0, p_bits = 0;
    uint64_t n_kmers = 0, n_pmers = 0, n_succ = 0;
    std::vector<VsWalkRec> rec;
    std::vector<VsKSlot> ktab;
    std::vector<uint64_t> pset;
};
// node_ascii / node_off: the N node sequences; woff[N + 1]: word offset of every node in the packed text (the index's own
// layout: ceil(len / 16) words per node); rc_delta: words between a node's forward text and its reverse complement.
void vs_walk_build_host(const uint8_t *node_ascii, const uint64_t *node_off, uint32_t n_nodes, uint32_t K, const uint32_t *woff,
                        uint32_t rc_delta, VsWalkHost &out, std::vector<uint32_t> *text_out = nullptr);
#endif
