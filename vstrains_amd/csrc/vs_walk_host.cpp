// Host side of the walk index (vs_walk.h): packs both strands of every node the way the device index lays them out,
// certifies (C1)-(C3), and builds the node-strand records, the (k+1)-mer table and the presence set.  One pass over the
// node text with two sorts; one-off per graph (0.1 s at 54 k nodes).  Plain C++ (no HIP), so that the certification --
// the one piece the exactness of the walk kernel rests on -- is also exercised by the CPU tests.
#include <algorithm>
#include <string.h>

#include "vs_walk.h"

namespace {

inline uint64_t win64(const uint32_t *w, uint64_t base) {  // 32 bases from base offset `base` (as vs_win64 on the device)
    const uint64_t i = base >> 4;
    const uint32_t sh = (uint32_t)(base & 15u) * 2u;
    const uint64_t lo = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
    const uint64_t hi = (uint64_t)w[i + 2];
    return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}
inline uint64_t lowmask(uint32_t bits) { return bits >= 64u ? ~0ull : ((1ull << bits) - 1ull); }

// hash of the `n` bases at base offset `at` of `text` (n >= 1), the kernel's vs_kmer_hash_* over 64-bit words
inline uint64_t mer_hash(const uint32_t *text, uint64_t at, uint32_t n, uint32_t seed_len) {
    uint64_t h = vs_kmer_hash_init(seed_len);
    for (uint32_t done = 0; done < n; done += 32u) {
        const uint32_t m = n - done < 32u ? n - done : 32u;
        h = vs_kmer_hash_step(h, win64(text, at + done) & lowmask(2u * m));
    }
    return vs_kmer_hash_done(h);
}
inline bool mer_equal(const uint32_t *text, uint64_t a, uint64_t b, uint32_t n) {
    for (uint32_t done = 0; done < n; done += 32u) {
        const uint32_t m = n - done < 32u ? n - done : 32u;
        if ((win64(text, a + done) ^ win64(text, b + done)) & lowmask(2u * m)) return false;
    }
    return true;
}
inline uint32_t base_at(const uint32_t *text, uint64_t at) { return (text[at >> 4] >> (2u * (uint32_t)(at & 15u))) & 3u; }

inline uint64_t rc_mer(uint64_t x, uint32_t w) {  // reverse complement of a w-mer in the low 2w bits
    uint64_t r = 0;
    for (uint32_t i = 0; i < w; i++) {
        r = (r << 2) | (3u - (x & 3u));
        x >>= 2;
    }
    return r;
}

struct Occ {
    uint64_t hash;
    uint32_t ns, pos;
    uint32_t kind;  // k-mer pass: 0 = starts a (k+1)-mer, 1 = the last k bases of a strand
};

}  // namespace

void vs_walk_build_host(const uint8_t *node_ascii, const uint64_t *node_off, uint32_t n_nodes, uint32_t K, const uint32_t *woff,
                        uint32_t rc_delta, VsWalkHost &out, std::vector<uint32_t> *text_out) {
    out = VsWalkHost();
    out.K = K;
    out.wp = vs_walk_wp(K);
    out.nw = (2u * K + 63u) / 64u;
    auto fail = [&](const char *why) { out.certified = false; out.why = why; };
    if (K < 2u) return fail("k + 1 < 2");
    if (out.nw > VS_WALK_MAX_NW) return fail("(k+1)-mers longer than 160 bases");
    if (n_nodes == 0u || n_nodes >= (1u << 25)) return fail("no nodes, or too many");
    const uint32_t k = K - 1u;

    // ---- both strands, packed (2 bits per base, every node on a word boundary, the reverse complement rc_delta words on)
    std::vector<uint32_t> text(2ull * rc_delta + 8u, 0u);
    std::vector<uint32_t> len(n_nodes);
    uint64_t n_strands_long = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        const uint8_t *t = node_ascii + node_off[i];
        const uint64_t l = node_off[i + 1] - node_off[i];
        if (l >= (1ull << 24)) return fail("a node of 2^24 bases or more");
        len[i] = (uint32_t)l;
        if (l < K) continue;  // no (k+1)-mer: never met by the walk (may hold any bytes, PE_Inference.py:118)
        n_strands_long += 2;
        const uint64_t f = (uint64_t)woff[i] * 16u, r = ((uint64_t)woff[i] + rc_delta) * 16u;
        for (uint64_t p = 0; p < l; p++) {
            const uint8_t c = t[p];
            const uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
            if (code > 3u) return fail("a node with a byte outside ACGT");  // (vs_index_build refuses such a graph anyway)
            text[(f + p) >> 4] |= code << (2u * (uint32_t)((f + p) & 15u));
            const uint64_t q = r + (l - 1u - p);
            text[q >> 4] |= (3u - code) << (2u * (uint32_t)(q & 15u));
        }
    }
    auto strand_base = [&](uint32_t ns) { return ((uint64_t)woff[ns >> 1] + ((ns & 1u) ? rc_delta : 0u)) * 16u; };

    // ---- (C1): all (k+1)-mers of both strands distinct
    std::vector<Occ> occ;
    {
        uint64_t total = 0;
        for (uint32_t i = 0; i < n_nodes; i++)
            if (len[i] >= K) total += 2ull * (len[i] - K + 1u);
        if (total >= (1ull << 31)) return fail("too many (k+1)-mers");
        occ.reserve(total);
        for (uint32_t ns = 0; ns < 2u * n_nodes; ns++) {
            const uint32_t l = len[ns >> 1];
            if (l < K) continue;
            const uint64_t b = strand_base(ns);
            for (uint32_t p = 0; p + K <= l; p++) occ.push_back(Occ{mer_hash(text.data(), b + p, K, K), ns, p, 0u});
        }
        out.n_kmers = occ.size();
    }
    {
        std::vector<uint32_t> order(occ.size());
        for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return occ[a].hash < occ[b].hash; });
        for (size_t i = 0; i < order.size();) {
            size_t j = i + 1;
            while (j < order.size() && occ[order[j]].hash == occ[order[i]].hash) j++;
            for (size_t a = i; a < j; a++)
                for (size_t b2 = a + 1; b2 < j; b2++) {
                    const Occ &x = occ[order[a]], &y = occ[order[b2]];
                    if (mer_equal(text.data(), strand_base(x.ns) + x.pos, strand_base(y.ns) + y.pos, K))
                        return fail("a (k+1)-mer occurs more than once (repeat, palindrome, or a node set that is no compacted de Bruijn graph)");
                }
            i = j;
        }
    }

    // ---- records (successors filled below)
    out.rec.assign(2ull * n_nodes, VsWalkRec{0u, 0u, 0u, 0u, {0u, 0u, 0u, 0u}});
    for (uint32_t ns = 0; ns < 2u * n_nodes; ns++) {
        VsWalkRec &r = out.rec[ns];
        r.len = len[ns >> 1];
        r.woff = woff[ns >> 1] + ((ns & 1u) ? rc_delta : 0u);
        if (r.len > K) {
            const uint32_t m = r.len - K < 32u ? r.len - K : 32u;
            const uint64_t t = win64(text.data(), strand_base(ns) + K) & lowmask(2u * m);
            r.tail_lo = (uint32_t)t;
            r.tail_hi = (uint32_t)(t >> 32);
        }
    }

    // ---- (C2), (C3): the k-mers that start a (k+1)-mer, and the last k bases of every strand
    {
        std::vector<Occ> kocc;
        kocc.reserve(occ.size() + n_strands_long);
        for (uint32_t ns = 0; ns < 2u * n_nodes; ns++) {
            const uint32_t l = len[ns >> 1];
            if (l < K) continue;
            const uint64_t b = strand_base(ns);
            for (uint32_t p = 0; p + K <= l; p++) kocc.push_back(Occ{mer_hash(text.data(), b + p, k, k), ns, p, 0u});
            kocc.push_back(Occ{mer_hash(text.data(), b + (l - k), k, k), ns, l - k, 1u});
        }
        std::vector<uint32_t> order(kocc.size());
        for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return kocc[a].hash < kocc[b].hash; });
        std::vector<uint32_t> group;  // members of one k-mer (equal text), indices into kocc
        std::vector<char> taken;
        for (size_t i = 0; i < order.size();) {
            size_t j = i + 1;
            while (j < order.size() && kocc[order[j]].hash == kocc[order[i]].hash) j++;
            // equal hashes: split by text (a run holds one k-mer unless two k-mers collide in 64 bits)
            taken.assign(j - i, 0);
            for (size_t a = i; a < j; a++) {
                if (taken[a - i]) continue;
                group.clear();
                group.push_back(order[a]);
                const Occ &x = kocc[order[a]];
                for (size_t b2 = a + 1; b2 < j; b2++) {
                    if (taken[b2 - i]) continue;
                    const Occ &y = kocc[order[b2]];
                    if (mer_equal(text.data(), strand_base(x.ns) + x.pos, strand_base(y.ns) + y.pos, k)) {
                        taken[b2 - i] = 1;
                        group.push_back(order[b2]);
                    }
                }
                uint32_t starts_at_0 = 0, starts_inside = 0, ends = 0;
                for (uint32_t g : group) {
                    const Occ &o = kocc[g];
                    if (o.kind == 1u) ends++;
                    else if (o.pos == 0u) starts_at_0++;
                    else starts_inside++;
                }
                if (starts_inside > 1u || (starts_inside == 1u && (starts_at_0 || ends)))
                    return fail("a k-mer inside a node also starts or ends another stretch (overlaps that are not node ends)");
                if (ends && starts_at_0) {
                    for (uint32_t ge : group) {
                        const Occ &e = kocc[ge];
                        if (e.kind != 1u) continue;
                        for (uint32_t gs : group) {
                            const Occ &s = kocc[gs];
                            if (s.kind != 0u) continue;  // (pos == 0: starts_inside is 0 here)
                            const uint32_t b = base_at(text.data(), strand_base(s.ns) + k);
                            uint32_t &slot = out.rec[e.ns].succ[b];
                            if (slot) return fail("two successors with the same next base");  // (cannot happen after (C1))
                            slot = s.ns + 1u;
                            out.n_succ++;
                        }
                    }
                }
            }
            i = j;
        }
    }

    // ---- (k+1)-mer table
    {
        uint32_t bits = 4;
        while ((1ull << bits) < 2ull * occ.size() + 2u) bits++;
        out.k_bits = bits;
        out.ktab.assign(1ull << bits, VsKSlot{0u, VS_WALK_EMPTY, 0u, 0u});
        const uint64_t mask = (1ull << bits) - 1ull;
        for (const Occ &o : occ) {
            uint64_t s = o.hash & mask;
            while (out.ktab[s].ns != VS_WALK_EMPTY) s = (s + 1u) & mask;
            out.ktab[s] = VsKSlot{(uint32_t)(o.hash >> 32), o.ns, o.pos, out.rec[o.ns].woff};
        }
    }
    // ---- presence set: canonical wp-mers of every node that holds a (k+1)-mer
    {
        const uint32_t wp = out.wp;
        uint64_t total = 0;
        for (uint32_t i = 0; i < n_nodes; i++)
            if (len[i] >= K) total += len[i] - wp + 1u;
        uint32_t bits = 4;
        while ((1ull << bits) < 2ull * total + 2u) bits++;
        out.p_bits = bits;
        out.pset.assign(1ull << bits, VS_PSET_EMPTY);
        const uint64_t mask = (1ull << bits) - 1ull;
        for (uint32_t i = 0; i < n_nodes; i++) {
            if (len[i] < K) continue;
            const uint64_t b = strand_base(2u * i);
            for (uint32_t p = 0; p + wp <= len[i]; p++) {
                const uint64_t f = win64(text.data(), b + p) & lowmask(2u * wp);
                const uint64_t r = rc_mer(f, wp);
                const uint64_t key = r < f ? r : f;
                uint64_t s = vs_pset_slot(key, bits);
                while (out.pset[s] != VS_PSET_EMPTY && out.pset[s] != key) s = (s + 1u) & mask;
                if (out.pset[s] == VS_PSET_EMPTY) {
                    out.pset[s] = key;
                    out.n_pmers++;
                }
            }
        }
    }
    out.certified = true;
    if (text_out) text_out->swap(text);
}

// ---- host twin of k_pe_walk's per-end work (tests; no device): the same vs_walk_end the kernel runs -------------------------
// lists[n_ends * cap] / counts[n_ends]: accepted node indices per end; counts[e] = 0xFFFFFFFF where the kernel would hand
// the pair to the general path (node met twice, more than cap nodes touched, more than four bytes outside ACGT).
// Returns 0, 1 when the node set does not certify (nothing is mapped then), < 0 on bad arguments.
extern "C" int vs_walk_map_ends_host(const uint8_t *node_ascii, const uint64_t *node_off, uint32_t n_nodes, uint32_t ksize,
                                     const uint8_t *read_ascii, const uint64_t *read_off, uint64_t n_ends, uint32_t cap,
                                     uint32_t *lists, uint32_t *counts) {
    if (!node_off || !read_off || !lists || !counts || !cap || ksize < 1u) return -1;
    std::vector<uint32_t> woff(n_nodes + 1);
    uint64_t words = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        woff[i] = (uint32_t)words;
        words += (node_off[i + 1] - node_off[i] + 15) / 16;
    }
    woff[n_nodes] = (uint32_t)words;
    VsWalkHost wh;
    std::vector<uint32_t> text;
    const uint32_t K = ksize + 1u;
    vs_walk_build_host(node_ascii, node_off, n_nodes, K, woff.data(), (uint32_t)(words + 16u), wh, &text);
    if (!wh.certified) return 1;
    VsWalkDev wk;
    wk.rec = wh.rec.data(); wk.ktab = wh.ktab.data(); wk.pset = wh.pset.data();
    wk.k_bits = wh.k_bits; wk.p_bits = wh.p_bits; wk.wp = wh.wp; wk.nw = wh.nw;
    std::vector<uint32_t> rw, row(cap);
    for (uint64_t e = 0; e < n_ends; e++) {
        const uint8_t *t = read_ascii + read_off[e];
        const uint32_t rlen = (uint32_t)(read_off[e + 1] - read_off[e]);
        counts[e] = 0u;
        if (rlen < K) continue;
        rw.assign((rlen + 15u) / 16u + 4u, 0u);
        uint32_t inv4 = 0xFFFFFFFFu, n_inv = 0;
        bool many = false;
        for (uint32_t p = 0; p < rlen; p++) {
            const uint8_t c = t[p];
            const uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
            if (code <= 3u) { rw[p >> 4] |= code << (2u * (p & 15u)); continue; }
            if (n_inv < 4u && p <= 254u) inv4 = (inv4 & ~(0xFFu << (8u * n_inv))) | (p << (8u * n_inv));
            else many = true;
            n_inv++;
        }
        if (many) { counts[e] = 0xFFFFFFFFu; continue; }
        bool over = false;
        uint32_t nt = 0;
        switch (wh.nw) {
            case 1: nt = vs_walk_end<1>(wk, text.data(), K, rw.data(), 0u, rlen, inv4, n_inv != 0u, row.data(), cap, &over); break;
            case 2: nt = vs_walk_end<2>(wk, text.data(), K, rw.data(), 0u, rlen, inv4, n_inv != 0u, row.data(), cap, &over); break;
            case 3: nt = vs_walk_end<3>(wk, text.data(), K, rw.data(), 0u, rlen, inv4, n_inv != 0u, row.data(), cap, &over); break;
            case 4: nt = vs_walk_end<4>(wk, text.data(), K, rw.data(), 0u, rlen, inv4, n_inv != 0u, row.data(), cap, &over); break;
            default: nt = vs_walk_end<5>(wk, text.data(), K, rw.data(), 0u, rlen, inv4, n_inv != 0u, row.data(), cap, &over); break;
        }
        if (over) { counts[e] = 0xFFFFFFFFu; continue; }
        uint32_t c = 0;
        for (uint32_t i = 0; i < nt; i++)
            if (row[i] >> 31) lists[e * (uint64_t)cap + c++] = row[i] & 0x7FFFFFFFu;
        counts[e] = c;
    }
    return 0;
}
