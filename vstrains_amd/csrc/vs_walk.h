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


// ---- the walk of one read end (shared by k_pe_walk and by the host twin the CPU tests run) ------------------------------
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

// One read end: rw = packed read words (the read starts at base offset rbase; words behind it are readable), rlen its
// length, inv4 / dirty its bytes outside ACGT.  row[cap] receives node | accepted << 31 for every visit the end pays to a
// node (a node met twice in a row -- an error inside it -- is one visit); returns the number of visits (more than cap:
// overflow).  *over_out: the end needs the general path (a node visited again later: its counts would have to be merged).
// Structure: rounds of { one exact lookup, a tight loop that follows the run node by node, the probes that decide where
// the scan resumes }; all lanes of a wavefront go through the rounds together, so the loop that runs most often is the
// shortest one.
template <uint32_t NW>
VS_HD uint32_t vs_walk_end(const VsWalkDev &wk, const uint32_t *text, uint32_t K, const uint32_t *rw, uint32_t rbase, uint32_t rlen,
                           uint32_t inv4, bool dirty, uint32_t *row, uint32_t cap, bool *over_out) {
    const uint32_t wp = wk.wp, kmask = (1u << wk.k_bits) - 1u;
    uint32_t lo = 0u, hi = dirty ? vsw_next_inv(inv4, 0u, rlen) : rlen;
    uint32_t j = 0u, nt = 0u;
    uint32_t cur_node = 0xFFFFFFFFu, cur_v = 0u, cur_coord = 0u, cur_kidx = 0u, cur_nlen = 0u;
    // the node of the current visit differs from the one held (or the end is finished): settle the one held
    auto flush = [&]() {
        if (cur_node == 0xFFFFFFFFu) return;
        const uint32_t acc = vsw_accept(cur_v, cur_coord, cur_kidx, cur_nlen, rlen, K) ? 0x80000000u : 0u;
        if (nt < cap) row[nt] = cur_node | acc;
        nt++;
    };
    for (;;) {
        if (j + K > hi) {  // no window left in this clean stretch
            if (hi >= rlen) break;
            lo = hi + 1u;
            hi = vsw_next_inv(inv4, lo, rlen);
            j = lo;
            continue;
        }
        // ---- one exact lookup: where does window j sit?
        uint32_t ns = 0xFFFFFFFFu, q = 0u;
        {
            uint64_t wv[NW];
            uint64_t h = vs_kmer_hash_init(K);
            for (uint32_t i = 0; i < NW; i++) {
                wv[i] = vsw_win(rw, rbase + j + 32u * i);
                if (i == NW - 1u) wv[i] &= vsw_lowmask(2u * K - 64u * (NW - 1u));
                h = vs_kmer_hash_step(h, wv[i]);
            }
            h = vs_kmer_hash_done(h);
            const uint32_t tag = (uint32_t)(h >> 32);
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
                    if (same) { ns = sl.ns; q = sl.pos; break; }
                }
                s = (s + 1u) & kmask;
            }
        }
        if (ns == 0xFFFFFFFFu) {
            // window j coincides with nothing.  Skip as far as an absent wp-mer proves: every window that holds the
            // wp-mer at x lies in [x - (K - wp), x], so with x = j + d all of [j, j + d] are empty.
            uint32_t nj = j + 1u;
            for (uint32_t d = K - wp;; d >>= 1) {
                if (vsw_absent(wk, rw, rbase, j + d)) { nj = j + d + 1u; break; }
                if (d == 0u) break;
            }
            j = nj;
            continue;
        }
        // ---- follow the run: one turn per node (per 32 bases of a long one)
        uint32_t p;  // read base at which the run breaks, or ~0 when the stretch is used up
        for (;;) {
            const VsWalkRec r = wk.rec[ns];
            const uint32_t nlen = r.len;
            const uint32_t in_node = nlen - K - q, in_read = hi - K - j;
            const uint32_t m = in_node < in_read ? in_node : in_read;  // windows that may follow here
            uint32_t adv = 0u;
            if (m) {
                uint64_t tw = q == 0u ? ((uint64_t)r.tail_lo | ((uint64_t)r.tail_hi << 32)) : vsw_win(text, r.woff * 16u + q + K);
                for (;;) {
                    const uint32_t n = m - adv < 32u ? m - adv : 32u;
                    const uint64_t x = (tw ^ vsw_win(rw, rbase + j + K + adv)) & (~0ull >> (64u - 2u * n));
                    if (x) { adv += vsw_first_diff(x); break; }
                    adv += n;
                    if (adv >= m) break;
                    tw = vsw_win(text, r.woff * 16u + q + K + adv);
                }
            }
            {   // windows j .. j + adv sit in this node
                const uint32_t node = ns >> 1;
                const uint32_t coord = (ns & 1u) ? nlen - K - q - adv : q;
                if (node == cur_node) {
                    cur_v += adv + 1u;
                    cur_coord = coord < cur_coord ? coord : cur_coord;
                } else {
                    flush();
                    cur_node = node; cur_v = adv + 1u; cur_coord = coord; cur_kidx = j; cur_nlen = nlen;
                }
            }
            j += adv;
            if (adv < m) { p = j + K; break; }            // mismatch inside the node
            if (j + K >= hi) { p = 0xFFFFFFFFu; break; }   // the stretch is used up
            // the strand ends here and the read goes on: its next base picks the successor
            const uint32_t pb = rbase + j + K;
            const uint32_t b = (rw[pb >> 4] >> (2u * (pb & 15u))) & 3u;
            const uint32_t nxt = b == 0u ? r.succ[0] : b == 1u ? r.succ[1] : b == 2u ? r.succ[2] : r.succ[3];
            if (!nxt) { p = j + K; break; }
            ns = nxt - 1u;
            q = 0u;
            j++;
        }
        if (p == 0xFFFFFFFFu) {
            j++;  // (j + K == hi: the loop head moves on to the next stretch or ends)
            continue;
        }
        // ---- the run broke at read base p: window j + 1 coincides with nothing (certified), and if neither wp-mer that
        // touches p occurs in a node, no window over p does
        const bool l_out = p + 1u < lo + wp, r_out = p + wp > hi;  // (a flank that leaves the stretch holds no window)
        const bool none = (l_out || vsw_absent(wk, rw, rbase, p + 1u - wp)) && (r_out || vsw_absent(wk, rw, rbase, p));
        j = none ? p + 1u : j + 2u;
    }
    flush();
    // a node visited twice (not in a row): its two counts belong together -- the general path does that
    bool over = nt > cap;
    for (uint32_t a = 1; a < nt && a < cap; a++)
        for (uint32_t b = 0; b < a; b++)
            if (((row[a] ^ row[b]) & 0x7FFFFFFFu) == 0u) over = true;
    *over_out = over;
    return nt;
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
