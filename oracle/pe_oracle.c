/*
 * CPU restatement of VStrains' PE-link inference in plain C  --  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker and the `cpu_baseline` ("port", 1 thread) for the HIP path.  Nothing in
 * the product package links or loads it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Algorithm = the reference's (file:line into /root/reference/utils/VStrains_PE_Inference.py):
 *   table of every (k+1)-mer window of every node, forward and reverse-complemented, each
 *   holding (node, forward offset)                                     :116-135
 *   per read end: one table lookup per window position; per posting count++, min offset,
 *   min read position                                                  :23-31
 *   acceptance ("saturation") test per touched node                    :36-47
 *   pair filters (upper-case N, shorter than k+1)                      :160-163
 *   short_mat upper-triangle-with-diagonal and node_mat updates        :174-188
 * The only liberties are data-structure ones: windows are hashed with a rolling polynomial and
 * compared byte-wise; per-read node state is kept for touched nodes only (the reference
 * allocates three length-N arrays per read end, :19-21); the float `expected` term is
 * replaced by the exact integer inequality (see oracle/pe_oracle.py:map_read_end_int, tested
 * equal to the float form on every golden case).
 *
 * Parity pin: tests/test_oracle_golden.py runs this library on every tests/golden/pe case
 * (outputs of the real reference script).
 *
 * Also here: the CPU twin of the on-device synthetic read generator (vs_synth_pairs in the HIP
 * library), so that the CPU baseline and the GPU run the very same read stream.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t hash;
    const uint8_t *text; /* representative window (split_len bytes) */
    uint32_t head;       /* first posting, chained through post_next */
    uint32_t used;
} slot_t;

typedef struct peo {
    uint32_t n_nodes, split_len;
    uint32_t *node_len;
    uint8_t *fwd_text, *rc_text; /* concatenated node texts */
    uint64_t *text_off;          /* n_nodes + 1 */
    slot_t *slots;
    uint64_t slot_mask;
    uint32_t *post_node, *post_pos, *post_next;
    uint64_t n_post;
    uint64_t pow_top; /* B^(split_len-1) */
    /* per-read scratch */
    uint32_t *cnt, *minp, *mini, *touched;
    uint32_t n_touched;
} peo;

#define HASH_B 0x100000001B3ull
#define NONE 0xFFFFFFFFu

static uint64_t window_hash(const uint8_t *p, uint32_t len) {
    uint64_t h = 0;
    for (uint32_t i = 0; i < len; i++) h = h * HASH_B + (uint64_t)p[i] + 1;
    return h;
}
static inline uint64_t scramble(uint64_t h) {
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}

static void table_add(peo *o, const uint8_t *win, uint32_t node, uint32_t pos) {
    uint64_t h = window_hash(win, o->split_len);
    uint64_t s = scramble(h) & o->slot_mask;
    for (;;) {
        slot_t *sl = &o->slots[s];
        if (!sl->used) {
            sl->used = 1;
            sl->hash = h;
            sl->text = win;
            sl->head = NONE;
        }
        if (sl->hash == h && memcmp(sl->text, win, o->split_len) == 0) {
            uint64_t id = o->n_post++;
            o->post_node[id] = node;
            o->post_pos[id] = pos;
            o->post_next[id] = sl->head;
            sl->head = (uint32_t)id;
            return;
        }
        s = (s + 1) & o->slot_mask;
    }
}

void peo_destroy(peo *o) {
    if (!o) return;
    free(o->node_len); free(o->fwd_text); free(o->rc_text); free(o->text_off); free(o->slots);
    free(o->post_node); free(o->post_pos); free(o->post_next);
    free(o->cnt); free(o->minp); free(o->mini); free(o->touched);
    free(o);
}

/* err: 0 ok; 1 = a node of length >= split_len holds a byte outside ACGT (the reference dies
 * with KeyError there); bad_node/bad_char report the node and the byte the reference would
 * name: the LAST offending byte of the FIRST window that holds one. */
peo *peo_create(const uint8_t *node_ascii, const uint64_t *node_off, uint32_t n_nodes,
                uint32_t ksize, int *err, uint32_t *bad_node, uint8_t *bad_char) {
    *err = 0;
    peo *o = (peo *)calloc(1, sizeof(peo));
    o->n_nodes = n_nodes;
    o->split_len = ksize + 1;
    uint32_t K = o->split_len;
    uint64_t total = node_off[n_nodes];
    o->node_len = (uint32_t *)malloc(sizeof(uint32_t) * (n_nodes + 1));
    o->text_off = (uint64_t *)malloc(sizeof(uint64_t) * (n_nodes + 1));
    o->fwd_text = (uint8_t *)malloc(total + 1);
    o->rc_text = (uint8_t *)malloc(total + 1);
    memcpy(o->fwd_text, node_ascii, total);
    memcpy(o->text_off, node_off, sizeof(uint64_t) * (n_nodes + 1));
    uint64_t n_win = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        uint64_t a = node_off[i], b = node_off[i + 1];
        uint32_t len = (uint32_t)(b - a);
        o->node_len[i] = len;
        if (len < K) continue;
        n_win += len - K + 1;
        for (uint64_t p = a; p < b; p++) {
            uint8_t c = node_ascii[p];
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
                /* first window holding a bad byte starts at max(0, p-a-K+1); report the last bad
                 * byte inside that window */
                uint64_t ws = (p - a + 1 > K) ? (p - K + 1) : a;
                uint64_t we = ws + K;
                uint8_t last = c;
                for (uint64_t q = ws; q < we; q++) {
                    uint8_t d = node_ascii[q];
                    if (d != 'A' && d != 'C' && d != 'G' && d != 'T') last = d;
                }
                *err = 1; *bad_node = i; *bad_char = last;
                peo_destroy(o);
                return NULL;
            }
        }
        for (uint32_t p = 0; p < len; p++) {
            uint8_t c = node_ascii[a + p], r;
            r = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
            o->rc_text[a + len - 1 - p] = r;
        }
    }
    uint64_t cap = 16;
    while (cap < 4 * n_win + 16) cap <<= 1;
    o->slot_mask = cap - 1;
    o->slots = (slot_t *)calloc(cap, sizeof(slot_t));
    o->post_node = (uint32_t *)malloc(sizeof(uint32_t) * (2 * n_win + 1));
    o->post_pos = (uint32_t *)malloc(sizeof(uint32_t) * (2 * n_win + 1));
    o->post_next = (uint32_t *)malloc(sizeof(uint32_t) * (2 * n_win + 1));
    for (uint32_t i = 0; i < n_nodes; i++) {
        uint32_t len = o->node_len[i];
        if (len < K) continue;
        uint64_t a = node_off[i];
        for (uint32_t p = 0; p + K <= len; p++) {
            table_add(o, o->fwd_text + a + p, i, p);
            /* revcomp(node[p:p+K]) == rc_text[len-p-K : len-p], payload keeps forward p */
            table_add(o, o->rc_text + a + (len - p - K), i, p);
        }
    }
    o->pow_top = 1;
    for (uint32_t i = 1; i < K; i++) o->pow_top *= HASH_B;
    o->cnt = (uint32_t *)calloc(n_nodes + 1, sizeof(uint32_t));
    o->minp = (uint32_t *)malloc(sizeof(uint32_t) * (n_nodes + 1));
    o->mini = (uint32_t *)malloc(sizeof(uint32_t) * (n_nodes + 1));
    o->touched = (uint32_t *)malloc(sizeof(uint32_t) * (n_nodes + 1));
    return o;
}

uint64_t peo_table_entries(const peo *o) { return o->n_post; }

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* One read end -> ascending node indices in out (at most cap written); returns how many pass. */
uint32_t peo_map_end(peo *o, const uint8_t *read, uint32_t rlen, uint32_t *out, uint32_t cap) {
    uint32_t K = o->split_len;
    if (rlen < K) return 0;
    o->n_touched = 0;
    uint64_t h = window_hash(read, K);
    for (uint32_t i = 0;; i++) {
        uint64_t s = scramble(h) & o->slot_mask;
        for (;;) {
            const slot_t *sl = &o->slots[s];
            if (!sl->used) break;
            if (sl->hash == h && memcmp(sl->text, read + i, K) == 0) {
                for (uint32_t id = sl->head; id != NONE; id = o->post_next[id]) {
                    uint32_t nd = o->post_node[id], pp = o->post_pos[id];
                    if (o->cnt[nd] == 0) {
                        o->touched[o->n_touched++] = nd;
                        o->minp[nd] = pp;
                        o->mini[nd] = i;
                    } else {
                        if (pp < o->minp[nd]) o->minp[nd] = pp;
                        if (i < o->mini[nd]) o->mini[nd] = i;
                    }
                    o->cnt[nd]++;
                }
                break;
            }
            s = (s + 1) & o->slot_mask;
        }
        if (i + K >= rlen) break;
        h = (h - ((uint64_t)read[i] + 1) * o->pow_top) * HASH_B + (uint64_t)read[i + K] + 1;
    }
    qsort(o->touched, o->n_touched, sizeof(uint32_t), cmp_u32);
    uint32_t kept = 0;
    for (uint32_t t = 0; t < o->n_touched; t++) {
        uint32_t nd = o->touched[t];
        int64_t v = o->cnt[nd], c = o->minp[nd], ki = o->mini[nd], nlen = o->node_len[nd];
        o->cnt[nd] = 0;
        int64_t right = c + nlen - 1;
        int64_t alt = c - ki + (int64_t)rlen - 1;
        if (alt < right) right = alt;
        int64_t saturate = right - c - (int64_t)K + 2;
        int64_t span = ((int64_t)rlen < nlen ? (int64_t)rlen : nlen) - (int64_t)K + 1;
        int pass = (v >= saturate) || (v * (int64_t)rlen >= span * ((int64_t)rlen - (int64_t)K));
        if (pass) {
            if (kept < cap) out[kept] = nd;
            kept++;
        }
    }
    return kept;
}

static int has_N(const uint8_t *s, uint64_t n) { return memchr(s, 'N', n) != NULL; }

/* Pairs given as concatenated ASCII with offsets (n_pairs+1 each).  Adds into node_mat /
 * short_mat (row-major n*n int64) and stats {n_reads, short_reads, used_reads}. */
void peo_count_pairs(peo *o, const uint8_t *fwd, const uint64_t *foff, const uint8_t *rve,
                     const uint64_t *roff, uint64_t n_pairs, int64_t *node_mat,
                     int64_t *short_mat, uint64_t *stats) {
    uint32_t n = o->n_nodes, K = o->split_len;
    uint32_t *lefts = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t *rights = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    for (uint64_t r = 0; r < n_pairs; r++) {
        const uint8_t *fs = fwd + foff[r], *rs = rve + roff[r];
        uint64_t fl = foff[r + 1] - foff[r], rl = roff[r + 1] - roff[r];
        if (has_N(fs, fl) || has_N(rs, rl)) { stats[0]++; continue; }
        if (fl < K || rl < K) { stats[1]++; continue; }
        stats[2]++;
        uint32_t nl = peo_map_end(o, fs, (uint32_t)fl, lefts, n);
        uint32_t nr = peo_map_end(o, rs, (uint32_t)rl, rights, n);
        for (uint32_t a = 0; a < nl; a++)
            for (uint32_t b = a; b < nl; b++) short_mat[(uint64_t)lefts[a] * n + lefts[b]]++;
        for (uint32_t a = 0; a < nr; a++)
            for (uint32_t b = a; b < nr; b++) short_mat[(uint64_t)rights[a] * n + rights[b]]++;
        for (uint32_t a = 0; a < nl; a++)
            for (uint32_t b = 0; b < nr; b++) node_mat[(uint64_t)lefts[a] * n + rights[b]]++;
    }
    free(lefts); free(rights);
}

/* The same pair loop for graphs whose dense matrices do not fit the host (5e4 nodes = 2 x 20 GB of
 * int64): every increment is emitted as a key instead -- bit 63 = matrix (0 node_mat, 1 short_mat),
 * low bits = i * n + j -- and the caller sums equal keys.  At most `cap` keys are written; the
 * number of increments is returned (call again with a larger buffer if it exceeds cap). */
uint64_t peo_count_pairs_keys(peo *o, const uint8_t *fwd, const uint64_t *foff, const uint8_t *rve,
                              const uint64_t *roff, uint64_t n_pairs, uint64_t *keys, uint64_t cap,
                              uint64_t *stats) {
    uint32_t n = o->n_nodes, K = o->split_len;
    uint32_t *lefts = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t *rights = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint64_t m = 0;
    const uint64_t SHORT = 1ull << 63;
#define EMIT(k) do { if (m < cap) keys[m] = (k); m++; } while (0)
    for (uint64_t r = 0; r < n_pairs; r++) {
        const uint8_t *fs = fwd + foff[r], *rs = rve + roff[r];
        uint64_t fl = foff[r + 1] - foff[r], rl = roff[r + 1] - roff[r];
        if (has_N(fs, fl) || has_N(rs, rl)) { stats[0]++; continue; }
        if (fl < K || rl < K) { stats[1]++; continue; }
        stats[2]++;
        uint32_t nl = peo_map_end(o, fs, (uint32_t)fl, lefts, n);
        uint32_t nr = peo_map_end(o, rs, (uint32_t)rl, rights, n);
        for (uint32_t a = 0; a < nl; a++)
            for (uint32_t b = a; b < nl; b++) EMIT(SHORT | ((uint64_t)lefts[a] * n + lefts[b]));
        for (uint32_t a = 0; a < nr; a++)
            for (uint32_t b = a; b < nr; b++) EMIT(SHORT | ((uint64_t)rights[a] * n + rights[b]));
        for (uint32_t a = 0; a < nl; a++)
            for (uint32_t b = 0; b < nr; b++) EMIT((uint64_t)lefts[a] * n + rights[b]);
    }
#undef EMIT
    free(lefts); free(rights);
    return m;
}

/* ------------------------------------------------------------------------------------------
 * CPU twin of the device read generator (vstrains_amd/csrc/vs_synth.hip).  Same integer
 * recipe, ASCII output.  See include/vstrains_hip.h: vs_synth_pairs for the parameter meaning.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint8_t comp_ascii(uint8_t c) {
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
}

/* genomes: concatenated ACGT ASCII, goff[n_strains+1]; cum[s] = upper u32 threshold (inclusive)
 * for strain s (last must be 0xFFFFFFFF).  Writes pair r (first_pair <= r < first_pair+n) as
 * read_len bytes each into fwd/rve (dense, stride read_len). */
void peo_synth_pairs(const uint8_t *genomes, const uint64_t *goff, const uint32_t *cum,
                     uint32_t n_strains, uint64_t seed, uint64_t first_pair, uint64_t n,
                     uint32_t read_len, uint32_t sub_thresh, uint32_t n_thresh, uint8_t *fwd,
                     uint8_t *rve) {
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    int64_t L = read_len;
    for (uint64_t q = 0; q < n; q++) {
        uint64_t r = first_pair + q;
        uint64_t base = mix64(seed * 0xD1342543DE82EF95ull + r);
        uint64_t u0 = mix64(base + 1), u1 = mix64(base + 2), u2 = mix64(base + 3), u3 = mix64(base + 4);
        uint32_t pick = (uint32_t)(u0 >> 32), s = 0;
        while (s + 1 < n_strains && pick > cum[s]) s++;
        const uint8_t *g = genomes + goff[s];
        int64_t glen = (int64_t)(goff[s + 1] - goff[s]);
        int64_t sum = (int64_t)(u1 & 0xFFFF) + (int64_t)((u1 >> 16) & 0xFFFF) +
                      (int64_t)((u1 >> 32) & 0xFFFF) + (int64_t)((u1 >> 48) & 0xFFFF);
        int64_t flen = 3 * L + (sum - 131070) * (3 * L) / 378372;
        if (flen < L) flen = L;
        if (flen > glen) flen = glen;
        int64_t start = (int64_t)((u2 >> 11) % (uint64_t)(glen - flen + 1));
        int flip = (int)(u2 & 1);
        uint8_t *a = fwd + q * read_len, *b = rve + q * read_len;
        if (flip) { uint8_t *t = a; a = b; b = t; }
        for (int64_t i = 0; i < L; i++) {
            a[i] = g[start + i];
            b[i] = comp_ascii(g[start + flen - 1 - i]);
        }
        /* substitutions: end 0 = fwd, end 1 = rve (after the flip) */
        for (int e = 0; e < 2; e++) {
            uint8_t *t = e ? rve + q * read_len : fwd + q * read_len;
            if (sub_thresh) {
                for (int64_t i = 0; i < L; i++) {
                    uint64_t h = mix64(base + 16 + (uint64_t)e * 4096 + (uint64_t)i);
                    if ((uint32_t)h < sub_thresh) {
                        uint8_t c = t[i];
                        uint32_t code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
                        code = (code + 1 + (uint32_t)((h >> 32) % 3)) & 3;
                        t[i] = (uint8_t)ACGT[code];
                    }
                }
            }
        }
        if ((uint32_t)u3 < n_thresh) {
            uint32_t end = (uint32_t)(u3 >> 62) & 1u;
            uint32_t pos = (uint32_t)((u3 >> 32) & 0x3FFFFFFFu) % read_len;
            (end ? rve : fwd)[q * read_len + pos] = 'N';
        }
    }
}
