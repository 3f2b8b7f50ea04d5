// K2-K4 fused: seed probe -> load-balanced hit expansion + exact extension -> per-end
// aggregation in LDS -> acceptance test -> PE-link counters.
//
// What it computes is single_end_read_mapping + the pair loop of the reference
// (utils/VStrains_PE_Inference.py:16-48 and :155-188), bit for bit:
//   for a read end and a node, the reference's three numbers are
//     v        = number of (read window, table entry) coincidences for that node   (:29)
//     coords   = smallest forward node offset among them                           (:30)
//     kindices = smallest read offset among them                                   (:31)
//   Windows that coincide lie on diagonals; on one diagonal they form maximal runs, and a run
//   is exactly a maximal exact match (MEM) of length l >= K between the read and one strand of
//   the node: it contributes l-K+1 to v, its first node offset to coords (mapped back to the
//   forward strand for reverse matches: the reference stores the forward offset under the
//   reverse-complemented window, :132) and its first read offset to kindices.  Palindromic
//   windows are held twice by the reference (:125,:132) and are found here once per strand.
//   Every MEM holds a seed at a read offset divisible by s; the MEM is credited by the first
//   such seed only (left extension < s), so nothing is counted twice.
//
// Work split per 256-thread workgroup and tile of `ept` read ends (ept/2 pairs, taken in locus
// order, see k_pe_locus):
//   P0  tile header + packed reads -> LDS (one wpe-word slot per end)
//   P1  one thread per (end, probe): seed, canonical form, open-address table probe (one 16-B
//       load per slot visited); posting count per probe -> LDS
//   P2  workgroup inclusive scan of the posting counts
//   P3  one thread per posting (binary search of the scan = CSR-style frontier expansion, so a
//       repeat seed's many postings spread over lanes): left/right extension on 64-bit windows
//       (XOR + clz/ctz), LDS atomics into an 8-slot per-end table keyed by node
//   P4  acceptance test per occupied slot (integer form, see oracle/pe_oracle.py)
//   P5  one thread per node_mat / short_mat increment (global atomics; tiles are locus-sorted
//       and taken in contiguous runs, so the same cells are hit again while still in L2)
// Ends that touch more than 8 nodes overflow the LDS table; their pairs go to a list that a
// second, fully general kernel (dense per-workgroup node state in HBM) works through.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vs_internal.h"

#define TPB 256
// The hand-off between the mapping kernel and the counter kernels (r6): the accepted nodes of a tile's ends are PACKED --
// end after end, every list padded to whole quads (16 bytes), no fixed row per end.  Found by a wavefront prefix sum over
// the ends' list lengths (k_pe_tiles P4), in LDS first, then written out as one coalesced stretch.
//   LC    list words a tile's region of the hand-off buffer holds per end ON AVERAGE (region = ept * LC words)
//   LCAP  accepted nodes an end may have on the main path (more -> the overflow kernels); rows of 16 sent 2.3 % of the
//         pairs of configs[4] (ends of 17 .. 19 nodes) through k_pe_mid a second time
//   pair-major counters (k_pe_accumulate): counts[end] = n | quad offset inside the tile's region << 8
//   row owners (graphs beyond 46 340 nodes): their entries name lists by END INDEX, so the lists keep a fixed stride
//         there -- a row of LC words per end, of which only the quads that hold nodes are written, and ONE more quad per
//         end, in an array of its own behind the rows, for nodes 17 .. LCAP; counts[end] = n
#define LC 16u
#define LCAP 20u
// (rows of LCAP words, 80 bytes apart, straddle a 64-byte stretch every other time -- the row owners took 13.0 ms instead of
// 10.9 at configs[4] -- and rows 128 bytes apart are fetched as whole 128-byte lines: k_list_owners 2.9 -> 4.1 ms)
__device__ __forceinline__ const uint32_t *vs_row_quad(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ hi, uint64_t row, uint32_t q) {
    return q < LC / 4u ? lists + row * LC + 4u * q : hi + row * 4u;
}
#define EMPTY_NODE 0xFFFFFFFFu
#ifndef PPT
#define PPT 2u               // postings per thread and expansion chunk
#endif
#ifndef TILES_WAVES
#define TILES_WAVES 5        // k_pe_tiles is compiled for 5 waves per SIMD (<= 96 VGPRs); its LDS tile fits 5 times too
#endif
#ifndef TTPB
#define TTPB 256             // threads per k_pe_tiles workgroup (a tile holds TTPB / 4 read ends)
#endif
#ifndef TILES_WAVES_LONG
#define TILES_WAVES_LONG 5   // the long-window instantiation (k = 127) too: 21.9 ms at configs[3] against 23.3 at 4 waves (118 VGPRs)
#endif
#ifndef VS_ADAPT
#define VS_ADAPT 1           // compile-time-shape instantiations pick, per end, the cheaper of two candidate offsets per grid position (step grid)
#endif
#ifndef PPT_LONG
#define PPT_LONG 1u          // the same for the long-window instantiations (MODE 2, k = 127)
#endif
#define CHUNK (TTPB * PPT)   // (the owner array of a tile is sized for the larger of the two)

struct PeParams {
    VsIndexDev idx;
    VsReadsDev rd;
    uint32_t *node_mat, *short_mat;
    unsigned long long *stats;
    uint32_t ept, pmax, words_cap, pool, pool_bits, debug_stop;
    uint64_t n_tiles;
    uint32_t *slow_list, *slow_count;
    uint32_t *dbg_lists, *dbg_counts;
    uint32_t dbg_cap;
    uint32_t accumulate;
    const uint32_t *perm;  // pair order of the tiles (locus-sorted) or NULL = input order
    uint32_t wpe;          // LDS words reserved per read end
    uint32_t tiles_per_wg; // contiguous run of tiles per workgroup
    uint32_t magic_pmax, magic_wpe;  // vs_fastdiv constants
    uint32_t count_postings;         // VS_DEBUG_POSTINGS: sum the postings expanded (diagnostics)
    uint32_t *out_lists;             // accepted node ids, tile order: packed regions of ept * LC words / rows of LCAP words (out_rows)
    uint32_t *out_counts;            // [n_tiles * ept] list lengths (0 for ends that add nothing), packed form: | quad offset << 8
    uint32_t out_rows;               // 1: fixed rows of LC words per end + out_lists_hi (the row owners follow), 0: packed
    uint32_t *out_lists_hi;          // [n_tiles * ept * 4] nodes 17 .. LCAP of every end (row layout)
    uint64_t n_pairs;
    uint32_t no_xcd_map;             // VS_NO_XCD_MAP=1: workgroup b takes run b (experiments)
    uint32_t shortcut;               // overlapping-seed ownership shortcut for single postings (P3 stage A)
    uint8_t *tile_map;               // vs_pe_count_tracked: one byte per 64 x 64 tile of node_mat, then of short_mat; NULL = none
    uint32_t tile_T;                 // tiles per matrix side = ceil(N / 64)
    uint32_t phase0;                 // VS_PHASE0: probe grid 0, s, 2s, ... (generic instantiations only; see vs_seed_phase)
    uint32_t mid_fast;               // k_pe_mid: clean ends are compared straight-line (vs_agree_fast; the block has the shape of MODE 1)
};

// A counter cell is about to be added to: its tile is marked (a plain store of 1; racing stores write the same value).
__device__ __forceinline__ void vs_mark_tile(uint8_t *map, uint32_t T, uint32_t mat, uint32_t x, uint32_t y) {
    if (!map) return;
    const uint64_t t = ((uint64_t)mat * T + (x >> 6)) * T + (y >> 6);
    if (!map[t]) map[t] = 1;
}

// The same without looking first: a plain store that nothing waits for (k_pe_mid marks from inside loops of fire-and-forget
// atomics; the load of the test above was a round trip per loop turn -- r5).
__device__ __forceinline__ void vs_mark_tile_store(uint8_t *map, uint32_t T, uint32_t mat, uint32_t x, uint32_t y) {
    if (map) map[((uint64_t)mat * T + (x >> 6)) * T + (y >> 6)] = 1;
}

// Inclusive scans over the 64 lanes of a wavefront by DPP moves -- row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then the
// last lane of a row broadcast to the rows after it (row_bcast 15 / 31) -- six VALU instructions where the __shfl_up form
// takes six ds_bpermute round trips through the LDS crossbar (r6; -DVS_DPP_SCAN=0: the __shfl_up form).  A lane whose source
// lies outside its row keeps the `old` operand: the identity of the operation.
#ifndef VS_DPP_SCAN
#define VS_DPP_SCAN 1
#endif
// (r6) SIX workgroups per CU for the compile-time shapes of graphs whose ends touch few nodes (the non-adaptive instantiations):
// a tile table of 768 slots instead of 1 024 (12 per end, placed by a multiply-high range reduction instead of a power-of-two
// mask) and a list region of 960 words make 26.8 KB of LDS per workgroup, and with the thread index laundered at the top of the
// tile loop as well (nothing P0 .. P2 derive from it is hoisted across P3) the kernel fits 80 vector registers without
// scratch: six wavefronts per SIMD instead of five.  configs[2]: 4.70 -> 4.45 ms; either half alone gains nothing
// (VS_LAUNDER_TOP alone: noise; the smaller table at five workgroups: +2 %).  -DVS_POOL12=0 -DVS_LAUNDER_TOP=0: round 5's shape.
#ifndef VS_LAUNDER_TOP
#define VS_LAUNDER_TOP 1
#endif
#ifndef VS_POOL12
#define VS_POOL12 1
#endif
#define VS_DPP_STEP(op, ctrl, rowmask) { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xf, false); v = op; }
__device__ __forceinline__ uint32_t vs_wave_scan_add(uint32_t v) {
#if VS_DPP_SCAN
    VS_DPP_STEP(v + t_, 0x111, 0xf) VS_DPP_STEP(v + t_, 0x112, 0xf) VS_DPP_STEP(v + t_, 0x114, 0xf) VS_DPP_STEP(v + t_, 0x118, 0xf)
    VS_DPP_STEP(v + t_, 0x142, 0xa) VS_DPP_STEP(v + t_, 0x143, 0xc)
#else
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t2 = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t2;
    }
#endif
    return v;
}
__device__ __forceinline__ uint32_t vs_wave_scan_max(uint32_t v) {
#if VS_DPP_SCAN
    VS_DPP_STEP(v > t_ ? v : t_, 0x111, 0xf) VS_DPP_STEP(v > t_ ? v : t_, 0x112, 0xf) VS_DPP_STEP(v > t_ ? v : t_, 0x114, 0xf)
    VS_DPP_STEP(v > t_ ? v : t_, 0x118, 0xf) VS_DPP_STEP(v > t_ ? v : t_, 0x142, 0xa) VS_DPP_STEP(v > t_ ? v : t_, 0x143, 0xc)
#else
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t2 = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d && t2 > v) v = t2;
    }
#endif
    return v;
}

struct Mem {  // one credited maximal exact match
    uint32_t cnt, minp, minj;
};

// Straight-line agreement of a read with node text on either side of a seed, for the common block
// shape (probe stride <= 32, reads <= 191 bases with w = 31): one left window, four / five right
// windows, everything loaded up front, the answer out of selects -- no data-dependent branch.
//   left : bases over which the read (going down from j) equals the text going down from tl, at
//          most cl (<= 32)
//   ext  : bases over which the read (going up from j + w) equals the text going up from tr, at most rem
// tl / tr are base indices into the one array of node texts (forward, and rc_delta words on the
// reverse complements).
struct __attribute__((packed, aligned(4))) VsQuad { uint32_t x, y, z, w; };  // 16-byte load at dword alignment

template <bool W4>  // W4: four right windows are enough (reads <= 159 bases with w = 31)
__device__ __forceinline__ void vs_agree_fast(const uint32_t *rw, uint32_t rbase, const uint32_t *tw, uint32_t tl, uint32_t cl,
                                              uint32_t tr, uint32_t rem, uint32_t j, uint32_t w, uint32_t *left_out,
                                              uint32_t *ext_out) {
    const uint32_t n0 = cl;
    const uint32_t rj = j + w;
    // Node text to the right of the seed: eleven words cover the five windows; they come as 16-byte
    // loads, and the second / third only when the match can reach that far (rem) -- on a graph of
    // short nodes one load settles most postings.  Words not loaded read as zero: whatever they make
    // of the comparison lies beyond `rem` and is clipped.
    const uint32_t ti = tr >> 4, sh = (tr & 15u) * 2u;
    const VsQuad q0 = *(const VsQuad *)(tw + ti);
    VsQuad q1 = {0u, 0u, 0u, 0u}, q2 = {0u, 0u, 0u, 0u};
    if (rem > 48u) q1 = *(const VsQuad *)(tw + ti + 4u);
    if (!W4 && rem > 112u) q2 = *(const VsQuad *)(tw + ti + 8u);  // (the fourth window needs word 8 only)
    if (W4 && rem > 112u) q2.x = tw[ti + 8u];
    auto tw64 = [&](uint32_t w0, uint32_t w1, uint32_t w2) {
        return (uint64_t)__builtin_amdgcn_alignbit(w1, w0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32);
    };
    // the read's side of the same five windows: eleven consecutive LDS words, one shift
    const uint32_t rr = rbase + rj, rsh = (rr & 15u) * 2u;
    const uint32_t *rp = rw + (rr >> 4);
    const uint32_t r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3], r4 = rp[4];
    auto rw64 = [&](uint32_t w0, uint32_t w1, uint32_t w2) {
        return (uint64_t)__builtin_amdgcn_alignbit(w1, w0, rsh) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, rsh) << 32);
    };
    const uint64_t xl = (vs_win(rw, rbase + j - n0) ^ vs_win(tw, tl - n0)) & vs_lowmask(2u * n0);
    const uint64_t x0 = rw64(r0, r1, r2) ^ tw64(q0.x, q0.y, q0.z);
    const uint64_t x1 = rw64(r2, r3, r4) ^ tw64(q0.z, q0.w, q1.x);
    *left_out = xl ? n0 - 1u - (uint32_t)((63 - __clzll((long long)xl)) >> 1) : n0;
    // Windows 2..4 only where the match can reach them: on a graph of nodes hardly longer than k+1 most
    // wavefronts have no such posting and skip the block (a branch on "any lane", not on data of one lane).
    uint32_t far = 160u;  // (nothing differs within reach: clipped by rem below)
    if (rem > 64u) {
        const uint32_t r5 = rp[5], r6 = rp[6], r7 = rp[7], r8 = rp[8], r9 = W4 ? 0u : rp[9], r10 = W4 ? 0u : rp[10];
        const uint64_t x2 = rw64(r4, r5, r6) ^ tw64(q1.x, q1.y, q1.z);
        const uint64_t x3 = rw64(r6, r7, r8) ^ tw64(q1.z, q1.w, q2.x);
        const uint64_t x4 = W4 ? 0ull : rw64(r8, r9, r10) ^ tw64(q2.x, q2.y, q2.z);
        uint64_t xs = x4;  // (W4: zero -- no difference found inside 128 bases means ext = 160, clipped by rem <= 128)
        uint32_t xb = 128u;
        if (x3) { xs = x3; xb = 96u; }
        if (x2) { xs = x2; xb = 64u; }
        if (xs) far = xb + ((uint32_t)(__ffsll((long long)xs) - 1) >> 1);
    }
    // first window that differs (selects), then one find-first-set
    uint64_t xn = x1;
    uint32_t xb = 32u;
    if (x0) { xn = x0; xb = 0u; }
    const uint32_t ext = xn ? xb + ((uint32_t)(__ffsll((long long)xn) - 1) >> 1) : far;
    *ext_out = ext < rem ? ext : rem;
}

// The same for longer strides and reads (k up to 158: stride <= 128; reads <= w + 256 bases): four
// left windows, eight right windows, still without a data-dependent branch.  Text words come as
// 16-byte loads that are skipped where the match cannot reach (cl / rem).
__device__ __forceinline__ void vs_agree_long(const uint32_t *rw, uint32_t rbase, const uint32_t *tw, uint32_t tl, uint32_t cl,
                                              uint32_t tr, uint32_t rem, uint32_t j, uint32_t w, uint32_t *left_out,
                                              uint32_t *ext_out) {
    // ---- left: window i holds the n_i = clamp(cl - 32 i, 0, 32) bases just below the previous one,
    // lowest base first, so a difference nearest the seed is the highest set bit
    uint32_t left = cl;
    {
        uint64_t xs = 0ull;
        uint32_t at = 0u, nn = 0u;
#pragma unroll
        for (int i = 3; i >= 0; i--) {  // (from the far window to the near one: the nearest difference wins)
            const uint32_t lo = 32u * (uint32_t)i;
            const uint32_t n = cl > lo ? (cl - lo < 32u ? cl - lo : 32u) : 0u;
            uint64_t x = 0ull;
            if (n) x = (vs_win(rw, rbase + j - lo - n) ^ vs_win(tw, tl - lo - n)) & vs_lowmask(2u * n);
            if (x) { xs = x; at = lo; nn = n; }
        }
        if (xs) left = at + nn - 1u - (uint32_t)((63 - __clzll((long long)xs)) >> 1);
    }
    // ---- right: seventeen words either side cover eight windows
    const uint32_t rj = j + w;
    const uint32_t ti = tr >> 4, sh = (tr & 15u) * 2u;
    const VsQuad q0 = *(const VsQuad *)(tw + ti);
    VsQuad q1 = {0u, 0u, 0u, 0u}, q2 = q1, q3 = q1;
    uint32_t t16 = 0u;
    if (rem > 48u) q1 = *(const VsQuad *)(tw + ti + 4u);
    if (rem > 112u) q2 = *(const VsQuad *)(tw + ti + 8u);
    if (rem > 176u) q3 = *(const VsQuad *)(tw + ti + 12u);
    if (rem > 240u) t16 = tw[ti + 16u];
    auto tw64 = [&](uint32_t w0, uint32_t w1, uint32_t w2) {
        return (uint64_t)__builtin_amdgcn_alignbit(w1, w0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32);
    };
    const uint32_t rr = rbase + rj, rsh = (rr & 15u) * 2u;
    const uint32_t *rp = rw + (rr >> 4);
    uint32_t r[17];
#pragma unroll
    for (int i = 0; i < 17; i++) r[i] = rp[i];
    auto rw64 = [&](uint32_t w0, uint32_t w1, uint32_t w2) {
        return (uint64_t)__builtin_amdgcn_alignbit(w1, w0, rsh) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, rsh) << 32);
    };
    const uint64_t x0 = rw64(r[0], r[1], r[2]) ^ tw64(q0.x, q0.y, q0.z);
    const uint64_t x1 = rw64(r[2], r[3], r[4]) ^ tw64(q0.z, q0.w, q1.x);
    const uint64_t x2 = rw64(r[4], r[5], r[6]) ^ tw64(q1.x, q1.y, q1.z);
    const uint64_t x3 = rw64(r[6], r[7], r[8]) ^ tw64(q1.z, q1.w, q2.x);
    const uint64_t x4 = rw64(r[8], r[9], r[10]) ^ tw64(q2.x, q2.y, q2.z);
    const uint64_t x5 = rw64(r[10], r[11], r[12]) ^ tw64(q2.z, q2.w, q3.x);
    const uint64_t x6 = rw64(r[12], r[13], r[14]) ^ tw64(q3.x, q3.y, q3.z);
    const uint64_t x7 = rw64(r[14], r[15], r[16]) ^ tw64(q3.z, q3.w, t16);
    uint64_t xs = x7;
    uint32_t xb = 224u;
    if (x6) { xs = x6; xb = 192u; }
    if (x5) { xs = x5; xb = 160u; }
    if (x4) { xs = x4; xb = 128u; }
    if (x3) { xs = x3; xb = 96u; }
    if (x2) { xs = x2; xb = 64u; }
    if (x1) { xs = x1; xb = 32u; }
    if (x0) { xs = x0; xb = 0u; }
    const uint32_t ext = xs ? xb + ((uint32_t)(__ffsll((long long)xs) - 1) >> 1) : 256u;
    *left_out = left;
    *ext_out = ext < rem ? ext : rem;
}

// Bytes outside ACGT cut a read into segments no match can cross (a dict lookup of a window over
// such a byte misses, PE_Inference.py:25-26).  `inv4` lists up to four such positions of an end
// (one byte each, 0xFF = none; ends with more go to the overflow path): is the seed at j clean, and
// which stretch [lo, hi) of the read around it is.
__device__ __forceinline__ bool vs_seed_limits(uint32_t inv4, uint32_t j, uint32_t w, uint32_t rlen, uint32_t *lo, uint32_t *hi) {
    uint32_t l = 0u, h = rlen;
    bool ok = true;
#pragma unroll
    for (uint32_t i = 0; i < 4u; i++) {
        const uint32_t p = (inv4 >> (8u * i)) & 0xFFu;
        if (p != 0xFFu) {
            if (p < j) l = l > p + 1u ? l : p + 1u;
            else if (p >= j + w) h = h < p ? h : p;
            else ok = false;
        }
    }
    *lo = l;
    *hi = h;
    return ok;
}

// Extension of a seed hit.  rw/rbase: packed read (LDS or global) and its first base; tw/tbase:
// packed node strand.  mk/mbase: validity mask of the read or NULL.  Returns false when the
// match is owned by an earlier probe or is shorter than K.
// The first left window and the first two right windows are loaded before anything is decided
// (independent loads, one memory round trip); most hits are settled by them.  Windows may reach
// past the end of a read / node: every packed buffer carries VS_PAD_WORDS of padding and the
// bits beyond the valid range are never used.
template <typename RB>
__device__ __forceinline__ bool vs_extend(const uint32_t *rw, RB rbase, uint32_t rlen, const uint32_t *tw,
                                          uint32_t tbase, uint32_t tlen, uint32_t j, uint32_t q, uint32_t w,
                                          uint32_t s, uint32_t K, const uint32_t *mk, uint64_t mbase,
                                          uint32_t *a_out, uint32_t *qa_out, uint32_t *len_out) {
    uint32_t c = s < j ? s : j;
    c = c < q ? c : q;
    const uint32_t n0 = c < 32u ? c : 32u;
    const uint32_t rj = j + w, rq = q + w;
    uint32_t rem = rlen - rj;
    {
        const uint32_t rem2 = tlen - rq;
        rem = rem < rem2 ? rem : rem2;
    }
    uint64_t xl = vs_win(rw, rbase + j - n0) ^ vs_win(tw, tbase + q - n0);
    uint64_t xr0 = vs_win(rw, rbase + rj) ^ vs_win(tw, tbase + rq);
    uint64_t xr1 = vs_win(rw, rbase + rj + 32u) ^ vs_win(tw, tbase + rq + 32u);
    if (mk) {
        xl |= vs_win64(mk, mbase + j - n0);
        xr0 |= vs_win64(mk, mbase + rj);
        xr1 |= vs_win64(mk, mbase + rj + 32u);
    }
    xl &= vs_lowmask(2u * n0);
    uint32_t left = 0;
    if (xl) {
        left = n0 - 1u - (uint32_t)((63 - __clzll((long long)xl)) >> 1);
    } else {
        left = n0;
        while (left < c) {  // s > 32 only (k > 85): further windows backwards
            uint32_t n = c - left < 32u ? c - left : 32u;
            uint64_t x = vs_win(rw, rbase + j - left - n) ^ vs_win(tw, tbase + q - left - n);
            if (mk) x |= vs_win64(mk, mbase + j - left - n);
            x &= vs_lowmask(2u * n);
            if (x) {
                left += n - 1u - (uint32_t)((63 - __clzll((long long)x)) >> 1);
                break;
            }
            left += n;
        }
    }
    if (left >= s) return false;  // an earlier probe lies inside this match and owns it
    uint32_t ext = 0;
    if (rem) {
        if (xr0) {
            uint32_t m = (uint32_t)(__ffsll((long long)xr0) - 1) >> 1;
            ext = m < rem ? m : rem;
            rem = 0;
        } else {
            uint32_t adv = rem < 32u ? rem : 32u;
            ext = adv;
            rem -= adv;
            if (rem) {
                if (xr1) {
                    uint32_t m = (uint32_t)(__ffsll((long long)xr1) - 1) >> 1;
                    ext += m < rem ? m : rem;
                    rem = 0;
                } else {
                    adv = rem < 32u ? rem : 32u;
                    ext += adv;
                    rem -= adv;
                }
            }
        }
    }
    while (rem) {
        uint64_t x = vs_win(rw, rbase + rj + ext) ^ vs_win(tw, tbase + rq + ext);
        if (mk) x |= vs_win64(mk, mbase + rj + ext);
        if (x) {
            uint32_t m = (uint32_t)(__ffsll((long long)x) - 1) >> 1;
            ext += m < rem ? m : rem;
            break;
        }
        uint32_t adv = rem < 32u ? rem : 32u;
        ext += adv;
        rem -= adv;
    }
    uint32_t len = left + w + ext;
    if (len < K) return false;
    *a_out = j - left;
    *qa_out = q - left;
    *len_out = len;
    return true;
}

__device__ __forceinline__ bool vs_accept(uint32_t v, uint32_t coord, uint32_t kidx, uint32_t nlen, uint32_t rlen, uint32_t K) {
    long long c = coord, ki = kidx, nl = nlen, rl = rlen, k = K;
    long long right = c + nl - 1;
    long long alt = c - ki + rl - 1;
    if (alt < right) right = alt;
    long long saturate = right - c - k + 2;
    long long span = (rl < nl ? rl : nl) - k + 1;
    return ((long long)v >= saturate) || ((long long)v * rl >= span * (rl - k));
}

// The same test in 32-bit arithmetic for the straight-line kernels (reads <= 191 bases: v, the read
// length and k+1 stay below 2^8, node offsets and lengths below 2^25, so nothing overflows int32).
__device__ __forceinline__ bool vs_accept32(uint32_t v, uint32_t coord, uint32_t kidx, uint32_t nlen, uint32_t rlen, uint32_t K) {
    const int c = (int)coord, ki = (int)kidx, nl = (int)nlen, rl = (int)rlen, k = (int)K;
    int right = c + nl - 1;
    const int alt = c - ki + rl - 1;
    if (alt < right) right = alt;
    const int saturate = right - c - k + 2;
    const int span = (rl < nl ? rl : nl) - k + 1;
    return ((int)v >= saturate) || ((int)v * rl >= span * (rl - k));
}

// Probe the table for the seed at read offset j.  Returns posting count (0 = miss) and payload.
// -DVS_PROBE_PAIR=1 requests two neighbouring slots together (at an eighth fill one lane in ten needs the second slot,
// so nearly every wavefront does, as a second dependent round trip): measured equal within noise at configs[2..4]
// (5.97 / 30.2 / 11.7 ms against 5.99 / 30.6 / 11.5), like tables of 1/16 .. 1/64 fill -- the probes are not where
// k_pe_tiles waits.  Fuller tables lose: 1/4 fill +3 %, 1/2 fill +13 % (tools/r3_table.sh).
#ifndef VS_PROBE_PAIR
#define VS_PROBE_PAIR 0
#endif
__device__ __forceinline__ uint32_t vs_probe(const VsIndexDev &idx, uint64_t key, uint32_t sr, uint32_t *pa, uint32_t *pb) {
    uint32_t mask = (1u << idx.table_bits) - 1u;
    uint32_t sl = vs_slot_of(key, idx.table_bits);
    const uint4 *tab = (const uint4 *)idx.table;
    for (;;) {
        const uint4 r0 = tab[sl];
#if VS_PROBE_PAIR
        const uint4 r1 = tab[(sl + 1u) & mask];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint4 raw = h ? r1 : r0;
#else
        {
            const uint4 raw = r0;
#endif
            uint64_t k = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
            if (k == VS_EMPTY_KEY) return 0u;
            if ((k & ~VS_MULTI_BIT) == key) {
                if (k & VS_MULTI_BIT) {
                    *pa = raw.z;
                    *pb = raw.w | (sr << 31);
                    return raw.w;
                }
                *pa = raw.z;
                *pb = raw.w ^ (sr << 31);
                return 1u;
            }
        }
        sl = (sl + (VS_PROBE_PAIR ? 2u : 1u)) & mask;
    }
}

// The same with the first slot of the chain already loaded (`raw`, from slot `sl`): two probes of one thread have their
// first -- nearly always only -- slot loads in flight together before either is looked at.
__device__ __forceinline__ uint32_t vs_probe_from(const VsIndexDev &idx, uint64_t key, uint32_t sr, uint32_t sl, uint4 raw, uint32_t *pa, uint32_t *pb) {
    const uint32_t mask = (1u << idx.table_bits) - 1u;
    const uint4 *tab = (const uint4 *)idx.table;
    for (;;) {
        const uint64_t k = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
        if (k == VS_EMPTY_KEY) return 0u;
        if ((k & ~VS_MULTI_BIT) == key) {
            *pa = raw.z;
            if (k & VS_MULTI_BIT) {
                *pb = raw.w | (sr << 31);
                return raw.w;
            }
            *pb = raw.w ^ (sr << 31);
            return 1u;
        }
        sl = (sl + 1u) & mask;
        raw = tab[sl];
    }
}

extern __shared__ __attribute__((aligned(16))) uint32_t vs_lds[];

// LDS carve shared by the kernel and the host-side size computation
struct TileLayout {
    uint32_t woff, gend, meta, inv, words, pcnt, pa, pb, hkey, hcnt, hminp, hminj, ns, state, list, owner, misc, total;
};
__host__ __device__ inline TileLayout tile_layout(uint32_t ept, uint32_t pmax, uint32_t words_cap, uint32_t pool, uint32_t list_trim = 0u) {
    TileLayout t;
    uint32_t o = 0;
    const uint32_t NI = ept * pmax;
    // headers and packed words exist twice: the next tile's are brought in while this one is worked on
    t.woff = o;  o += 2u * ((ept + 2u) & ~1u);  // global word offset of every end of the tile
    t.gend = o;  o += 2u * ept;                 // global end index
    t.meta = o;  o += 2u * ept;
    t.inv = o;   o += ept;                      // positions of bytes outside ACGT (straight-line kernels; current tile only)
    t.words = o; o += 2u * (words_cap + 8u);
    t.pcnt = o;  o += NI + 1u;  // [0] stays zero: the scan is read as s_pcnt[it - 1] .. s_pcnt[it] without a test for it == 0
    t.pa = o;    o += NI;
    t.pb = o;    o += NI;
    t.hkey = o;  o += pool;
    t.hcnt = o;  o += pool;
    t.hminp = o; o += pool;
    t.hminj = o; o += pool;
    t.ns = o;    o += ept;
    t.state = o; o += ept;
    // the posting owners of P3 and the accepted lists of P4/P5 are never live together
    t.list = o;
    t.owner = o; o += (ept * LC - list_trim > CHUNK ? ept * LC - list_trim : CHUNK);
    t.misc = o;  o += 16u;
    t.total = o;
    return t;
}

// SW, SP != 0: the tile shape is a compile-time one for k = 55 -- 64 ends per tile, 1024-slot table,
// SW packed words and SP probes per end (vs_seed_probes): (10, 4) = 2 x 145..159 bases, (8, 3) = 2 x 113..128, (7, 3) = 2 x 108..112,
// (7, 2) = 2 x 97..107.  Every LDS array then sits at a constant offset (folded into the LDS
// instructions) instead of costing a scalar register and an add, the divisions by pmax / wpe and
// k+1 / seed length / stride become constants.  The host picks one when the block has that shape.
#ifndef STD_EPT
#define STD_EPT (TTPB / 4u)
#endif
#define STD_POOL_BITS (STD_EPT > 32u ? 10u : STD_EPT > 16u ? 9u : 8u)  // pool_for(STD_EPT)
#define STD_K 56u   // k + 1
#define STD_W 31u   // seed length and probe stride that follow from it (seed_geometry)
#define STD_S 26u
#define STD2_EPT 60u  // the k = 127 shape (MODE 2): ends per tile, pool_for(60) = 1024 slots, k + 1, seed length, stride
#define STD2_POOL_BITS 10u
#define STD2_K 128u
#define STD2_W 63u
#define STD2_S 66u
// MODE 0: generic loops (masked reads through the validity mask, any stride / read length);
//      1: straight-line comparison, stride <= 32, reads <= w + 160; 2: the same for stride <= 128, reads <= w + 256.
// tile and pair indices inside k_pe_tiles: a block holds fewer than 2^32 ends, so 32 bits do (-DVS_TILE32=0: 64, as before --
// six more scalar registers spilled and two more vector registers in the k = 55 shape)
#ifndef VS_TILE32
#define VS_TILE32 1
#endif
#if VS_TILE32
typedef uint32_t tidx_t;
#else
typedef uint64_t tidx_t;
#endif
template <int MODE, uint32_t SW, uint32_t SP, bool AD = false>
__global__ void __launch_bounds__(TTPB)
__attribute__((amdgpu_waves_per_eu((VS_POOL12 && SW != 0u && !AD) ? 6 : MODE == 2 ? TILES_WAVES_LONG : TILES_WAVES, (VS_POOL12 && SW != 0u && !AD) ? 6 : MODE == 2 ? TILES_WAVES_LONG : TILES_WAVES)))
k_pe_tiles(PeParams P) {
    constexpr bool FAST = MODE != 0;
    constexpr uint32_t AB = MODE == 2 ? 9u : 8u;  // bits of the read offset packed under the node length (credit / P4)
    constexpr bool STD = SW != 0u;
    // postings per thread and expansion chunk: two for the k = 55 shapes; ONE for the long-window comparison (MODE 2), whose
    // second posting's state went to scratch (72 B per lane at 96 VGPRs): 8.72 -> 7.29 ms at configs[3] (r5)
    constexpr uint32_t KPPT = MODE == 2 ? PPT_LONG : PPT, KCHUNK = TTPB * KPPT;
    // (r5) ADAPT: the probe grid of an end is chosen among the exact "step grids" (tests/seed_extend_model.step_grid): grid
    // positions s-1, 2s-1, ... of which those from the t-th on are moved D = s - 2 - (len - w) mod s bases towards the
    // read's start.  P1 probes both candidates of every position (8 slot loads per 150-base end instead of 4), picks the t
    // with the fewest postings, and only that grid is expanded: 57 -> 43 postings per end at configs[4], 21.9 -> 17.1
    // at configs[2] (tools/phase_gate.py).  Any grid of the family gives the same lists; the choice only changes the work.
    constexpr bool ADAPT = STD && AD && VS_ADAPT != 0;
    static_assert(!ADAPT || (MODE == 2 ? STD2_EPT : STD_EPT) * SP <= TTPB, "the adaptive probe phase takes one probe position per thread");
    constexpr uint32_t STD_WPE = SW, STD_PMAX = SP;
    const uint32_t tid = threadIdx.x;
    // (the compile-time-shape instantiation is also the one without diagnostics: the host only picks
    // it for plain counting runs)
    const uint32_t debug_stop = STD ? 0u : P.debug_stop;
    const bool count_postings = !STD && P.count_postings, want_dbg = !STD && P.dbg_counts != nullptr;
    const bool accumulate = STD || P.accumulate;
    // (r3) MODE 2 has one compile-time shape too: k = 127 with 2 x 241..256 bases -- 63-base seeds, stride 66, 16 words and
    // two probes per end, 60 ends per tile (what the host's LDS budget gives that shape)
    constexpr uint32_t C_EPT = MODE == 2 ? STD2_EPT : STD_EPT, C_K = MODE == 2 ? STD2_K : STD_K;
    constexpr uint32_t C_W = MODE == 2 ? STD2_W : STD_W, C_S = MODE == 2 ? STD2_S : STD_S;
    constexpr uint32_t C_POOL_BITS = MODE == 2 ? STD2_POOL_BITS : STD_POOL_BITS;
    const uint32_t ept = STD ? C_EPT : P.ept, pmax = STD ? STD_PMAX : P.pmax;
    const uint32_t NI = ept * pmax;
    const uint32_t w = STD ? C_W : P.idx.w, s = STD ? C_S : P.idx.s, K = STD ? C_K : P.idx.K;
    const uint32_t wv = VS_SEED_VERIFIED(w);  // seed bases the comparison skips (0: seeds with mixed keys, vs_seed_key)
    // (the 768-slot table is for graphs whose ends touch few nodes: the adaptive instantiations -- many postings per seed, long
    // lists -- keep 1 024 slots and five wavefronts per SIMD)
    constexpr bool P12 = VS_POOL12 != 0 && STD && !AD;
    constexpr uint32_t C_POOL = P12 ? 768u : (1u << C_POOL_BITS), C_TRIM = P12 ? (MODE == 2 ? 192u : 64u) : 0u;  // (k = 127: 16 words per end of read text leave less)
    const uint32_t pool = STD ? C_POOL : P.pool, pool_shift = 32u - (STD ? C_POOL_BITS : P.pool_bits);
    const uint32_t words_cap = STD ? C_EPT * STD_WPE : P.words_cap;
    const TileLayout T = tile_layout(ept, pmax, words_cap, pool, C_TRIM);
    // (these five point at the current tile's copy; see the top of the tile loop)
    uint32_t *s_gwoff = vs_lds + T.woff;   // global word offsets (mask reads, slow path)
    uint32_t *s_gend = vs_lds + T.gend;
    uint32_t *s_meta = vs_lds + T.meta;
    uint32_t *s_inv = vs_lds + T.inv;      // up to four positions of bytes outside ACGT per end (vs_seed_limits)
    uint32_t *s_words = vs_lds + T.words;  // end e occupies words [e*wpe, (e+1)*wpe)
    const uint32_t hw = (ept + 2u) & ~1u, wcap = words_cap + 8u;
    uint32_t *s_pcnt = vs_lds + T.pcnt + 1u;  // posting counts per probe, then their inclusive scan; s_pcnt[-1] == 0
    uint32_t *s_pa = vs_lds + T.pa;
    uint32_t *s_pb = vs_lds + T.pb;
    uint32_t *s_hkey = vs_lds + T.hkey;    // tile-wide (end, node) table: key = end << 25 | node
    uint32_t *s_hcnt = vs_lds + T.hcnt;
    uint32_t *s_hminp = vs_lds + T.hminp;
    uint32_t *s_hminj = vs_lds + T.hminj;
    uint32_t *s_ns = vs_lds + T.ns;        // accepted nodes per end
    uint32_t *s_state = vs_lds + T.state;  // bit0: end belongs to a used pair, bit1: overflow; bits 8..: first probe offset (vs_seed_phase)
    const bool phase0 = !STD && P.phase0;
    uint32_t *s_list = vs_lds + T.list;    // accepted node ids, LC per end
    uint32_t *s_misc = vs_lds + T.misc;
    uint32_t *s_owner = vs_lds + T.owner;
    const uint32_t wpe = STD ? STD_WPE : P.wpe;
    const uint32_t ppt = ept / 2u;
    const bool has_inv = FAST && P.rd.inv4 != nullptr;  // (the generic kernel reads the mask instead)

    if (tid < 3) s_misc[12 + tid] = 0;  // workgroup-local stats
    if (tid == 3) vs_lds[T.pcnt] = 0;
    // a workgroup takes a contiguous run of the locus-sorted tiles (node text stays in L1/L2)
    // Workgroups go to the 8 XCDs round-robin (blockIdx % 8).  The runs are handed out so that XCD x
    // works through the x-th eighth of the locus order: what a locus touches (table slots, postings,
    // node text) is then cached in one L2 instead of eight.
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0u && !P.no_xcd_map) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // (pairs and tiles of a block fit 32 bits: vs_reads holds fewer than 2^32 ends)
    const tidx_t n_pairs32 = (tidx_t)P.n_pairs, n_tiles32 = (tidx_t)P.n_tiles;
    const tidx_t tile_lo = (tidx_t)wg * P.tiles_per_wg;
    const tidx_t tile_hi = tile_lo + P.tiles_per_wg < n_tiles32 ? tile_lo + P.tiles_per_wg : n_tiles32;
    // The headers of a tile (pair order -> end index -> word offset, length) are two dependent
    // global loads; they run ahead in registers, one link per tile (pair order two tiles ahead, word
    // offset and length one tile ahead; one end per thread, ept <= TTPB), so that neither waits for
    // the other and both are covered by the previous tiles' work.
    uint32_t pf_gend = 0, pf_gwoff = 0, pf_meta = 0, pf_inv = 0xFFFFFFFFu, pf_pair = 0xFFFFFFFFu;
    uint32_t cur_inv = 0xFFFFFFFFu;  // this thread's end of the CURRENT tile (goes to LDS at the top of the tile)
    auto prefetch_pair = [&](tidx_t t) {  // pair (in input order) of this thread's end in tile t
        pf_pair = 0xFFFFFFFFu;
        if (t < tile_hi && tid < ept) {
            const tidx_t p = t * ppt + (tid >> 1);
            if (p < n_pairs32) pf_pair = P.perm ? P.perm[p] : (uint32_t)p;
        }
    };
    auto prefetch_headers = [&]() {  // of the tile whose pair order sits in pf_pair
        if (pf_pair != 0xFFFFFFFFu) {
            pf_gend = 2u * pf_pair + (tid & 1u);
            pf_gwoff = P.rd.woff[pf_gend];
            pf_meta = P.rd.meta[pf_gend];
            if (has_inv) pf_inv = P.rd.inv4[pf_gend];
        }
    };
    // One credited maximal exact match into the tile's (end, node) table.
    auto credit = [&](uint32_t e, uint32_t node, uint32_t nlen, uint32_t add, uint32_t minp, uint32_t a) {
        const uint32_t key = (e << 25) | node;
        uint32_t at = P12 ? __umulhi(key * 0x9E3779B1u, C_POOL) : (key * 0x9E3779B1u) >> pool_shift;
        bool placed = false;
        for (uint32_t pr = 0; pr < 64u; pr++) {
            const uint32_t old = atomicCAS(&s_hkey[at], EMPTY_NODE, key);
            if (old == EMPTY_NODE || old == key) {
                atomicAdd(&s_hcnt[at], add);
                atomicMin(&s_hminp[at], minp);
                // (straight-line instantiations: reads <= 191 bases, so the node length rides in the
                // upper 24 bits -- equal for every update of the slot -- and P4 needs no header)
                atomicMin(&s_hminj[at], FAST ? (nlen << AB) | a : a);
                placed = true;
                break;
            }
            at = P12 ? (at + 1u == C_POOL ? 0u : at + 1u) : (at + 1u) & (pool - 1u);
        }
        if (!placed) atomicOr(&s_state[e], 2u);
    };
    // The first tile's headers and words are fetched the plain way; from then on the headers of tile
    // t+1 sit in registers during tile t, are put into the other LDS copy at its start, and the words
    // of tile t+1 follow them there as LDS-direct loads (global_load_lds: no registers, nothing waits
    // for them until just before P4) -- P0 of the next tile finds everything in place.
    const uint32_t lane0 = tid & 63u, wv0 = tid >> 6;
    prefetch_pair(tile_lo);
    prefetch_headers();
    {
        const tidx_t np0 = n_pairs32 - tile_lo * ppt;
        const uint32_t ne0 = 2u * (uint32_t)(np0 < ppt ? np0 : ppt);
        if (tid < ne0) {
            s_gend[tid] = pf_gend;
            s_gwoff[tid] = pf_gwoff;
            s_meta[tid] = pf_meta;
        }
        cur_inv = pf_inv;
        if (tid < 8u) { s_words[words_cap + tid] = 0u; s_words[wcap + words_cap + tid] = 0u; }  // pads of both copies
        __syncthreads();
        for (uint32_t i = tid; i < ne0 * wpe; i += TTPB) {
            const uint32_t e = STD ? i / STD_WPE : vs_fastdiv(i, P.magic_wpe), k = i - e * wpe;
            const uint32_t nw = ((s_meta[e] & VS_LEN_MASK) + 15u) >> 4;
            s_words[i] = k < nw ? P.rd.words[s_gwoff[e] + k] : 0u;
        }
    }
    prefetch_pair(tile_lo + 1u);
    prefetch_headers();
    prefetch_pair(tile_lo + 2u);
    for (tidx_t tile = tile_lo; tile < tile_hi; tile++) {
#if VS_LAUNDER_TOP
        uint32_t ltop = tid;  // (nothing P0 .. P2 derive from the thread index is hoisted out of the tile loop: see VS_POOL12)
        asm volatile("" : "+v"(ltop));
#else
        const uint32_t ltop = tid;
#endif
        const tidx_t p0 = tile * ppt;
        const uint32_t npair = (uint32_t)((n_pairs32 - p0) < ppt ? (n_pairs32 - p0) : ppt);
        const uint32_t ne = 2u * npair;
        const uint32_t cur = (uint32_t)(tile - tile_lo) & 1u, nxt = cur ^ 1u;
        s_gwoff = vs_lds + T.woff + cur * hw;
        s_gend = vs_lds + T.gend + cur * ept;
        s_meta = vs_lds + T.meta + cur * ept;
        s_words = vs_lds + T.words + cur * wcap;
        uint32_t *n_gwoff = vs_lds + T.woff + nxt * hw, *n_gend = vs_lds + T.gend + nxt * ept;
        uint32_t *n_meta = vs_lds + T.meta + nxt * ept, *n_words = vs_lds + T.words + nxt * wcap;
        uint32_t ne1 = 0;  // ends of the next tile of this run
        if (tile + 1u < tile_hi) {
            const tidx_t np1 = n_pairs32 - (tile + 1u) * ppt;
            ne1 = 2u * (uint32_t)(np1 < ppt ? np1 : ppt);
        }
        __syncthreads();  // previous tile fully consumed; this tile's words have landed (see before P4)
        // ---- P0: the next tile's headers go to the other copy, the (end, node) table is emptied
        if (ltop < ne1) {
            n_gend[ltop] = pf_gend;
            n_gwoff[ltop] = pf_gwoff;
            n_meta[ltop] = pf_meta;
        }
        if (has_inv && ltop < ept) {  // (one copy is enough: nothing reads it before the barrier below)
            s_inv[ltop] = cur_inv;
            cur_inv = pf_inv;
        }
        for (uint32_t i = ltop; i < pool; i += TTPB) {
            s_hkey[i] = EMPTY_NODE;
            s_hcnt[i] = 0;
            s_hminp[i] = 0xFFFFFFFFu;
            s_hminj[i] = 0xFFFFFFFFu;
        }
        __syncthreads();
        prefetch_headers();            // tile + 2 (its pair order arrived during the previous tile)
        prefetch_pair(tile + 3u);
        // packed reads of the next tile: one end per wpe-word slot; a wavefront's load instruction
        // fills 64 consecutive LDS words, every lane from its own global address
        for (uint32_t b64 = wv0 * 64u; b64 < ne1 * wpe; b64 += TTPB) {
            const uint32_t i = b64 + lane0;
            if (i < ne1 * wpe) {
                const uint32_t e = STD ? i / STD_WPE : vs_fastdiv(i, P.magic_wpe), k = i - e * wpe;
                const uint32_t nw = ((n_meta[e] & VS_LEN_MASK) + 15u) >> 4;
                // (words behind the read's last one are never looked at: they get a copy of the last)
                if (nw) __builtin_amdgcn_global_load_lds(P.rd.words + n_gwoff[e] + (k < nw ? k : nw - 1u), n_words + b64, 4, 0, 0);
            }
        }
        // pair classification (PE_Inference.py:160-165): one thread per pair
        if (ltop < npair) {
            uint32_t mf = s_meta[2 * ltop], mr = s_meta[2 * ltop + 1];
            uint32_t cls;
            if (((mf | mr) >> 24) & VS_FLAG_N) cls = 0;
            else if ((mf & VS_LEN_MASK) < K || (mr & VS_LEN_MASK) < K) cls = 1;
            else cls = 2;
            atomicAdd(&s_misc[12 + cls], 1u);
            uint32_t st = (cls == 2) ? 1u : 0u;
            // (an end with more bytes outside ACGT than vs_seed_limits can hold: the pair takes the overflow path)
            if (FAST && st && (((mf | mr) >> 24) & VS_FLAG_MANY)) st = 3u;
            // the probe grid of either end.  Plain: phi, phi + s, ... (vs_seed_phase: the shortest exact grid for the end's
            // length), phi in bits 8..31.  ADAPT: base s - 1 in bits 8..15, the step point t in bits 16..19 (set by P1), the
            // shift D in bits 20..27 (0: the length leaves one phase only)
            auto grid_bits = [&](uint32_t len) {
                if (!ADAPT || !(st & 1u)) return vs_seed_phase(len, w, s, phase0) << 8;
                const uint32_t r = (len - w) % s;
                return ((s - 1u) << 8) | ((r + 2u <= s ? s - 2u - r : 0u) << 20);
            };
            s_state[2 * ltop] = st | grid_bits(mf & VS_LEN_MASK);
            s_state[2 * ltop + 1] = st | grid_bits(mr & VS_LEN_MASK);
            s_ns[2 * ltop] = s_ns[2 * ltop + 1] = 0;
        }
        __syncthreads();
        if (debug_stop == 1u) continue;
        // ---- P1: probes
        if (ADAPT) {
            // one grid position per thread; both candidate offsets probed with their slot loads in flight together
            const uint32_t it = ltop;
            const uint32_t e = it / (STD ? STD_PMAX : 1u), pi = it - e * pmax;
            uint32_t c0 = 0, pa0 = 0, pb0 = 0, c1 = 0, pa1 = 0, pb1 = 0, D = 0;
            const uint32_t est = (it < NI && e < ne) ? s_state[e] : 0u;
            const bool used = (est & 3u) == 1u;
            if (used) {
                const uint32_t meta = s_meta[e], rlen = meta & VS_LEN_MASK;
                D = est >> 20;
                const uint32_t j0 = ((est >> 8) & 0xFFu) + pi * s, j1 = j0 - D;
                if (j0 + w <= rlen) {
                    bool ok0 = true, ok1 = D != 0u;
                    if ((meta >> 24) & VS_FLAG_INVALID) {
                        uint32_t lo, hi;
                        ok0 = vs_seed_limits(s_inv[e], j0, w, rlen, &lo, &hi);
                        if (D) ok1 = vs_seed_limits(s_inv[e], j1, w, rlen, &lo, &hi);
                    }
                    uint32_t sr0, sr1;
                    const uint64_t key0 = vs_seed_key(s_words, e * wpe * 16u + j0, w, &sr0);
                    const uint64_t key1 = vs_seed_key(s_words, e * wpe * 16u + j1, w, &sr1);
                    const uint4 *tab = (const uint4 *)P.idx.table;
                    const uint32_t sl0 = vs_slot_of(key0, P.idx.table_bits), sl1 = vs_slot_of(key1, P.idx.table_bits);
                    uint4 r0 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u), r1 = r0;  // (VS_EMPTY_KEY: a seed that is not probed misses)
                    if (ok0) r0 = tab[sl0];
                    if (ok1) r1 = tab[sl1];
                    c0 = vs_probe_from(P.idx, key0, sr0, sl0, r0, &pa0, &pb0);
                    c1 = vs_probe_from(P.idx, key1, sr1, sl1, r1, &pa1, &pb1);
                    if (!D) { c1 = c0; pa1 = pa0; pb1 = pb0; }
                }
            }
            if (it < NI) { s_pa[it] = c0; s_pb[it] = c1; }  // (staged: every thread of an end needs the end's counts)
            __syncthreads();
            uint32_t cnt = c0, pa = pa0, pb = pb0;
            if (used && D) {
                // postings of the grid with step point t: positions below t unmoved, the others moved; the first minimum
                const uint32_t b = e * pmax;
                uint32_t cur = 0;
                for (uint32_t i = 0; i < pmax; i++) cur += s_pb[b + i];
                uint32_t best = cur, bt = 0;
                for (uint32_t t = 1; t <= pmax; t++) {
                    cur += s_pa[b + t - 1u] - s_pb[b + t - 1u];
                    if (cur < best) { best = cur; bt = t; }
                }
                if (pi >= bt) { cnt = c1; pa = pa1; pb = pb1; }
                if (pi == 0u) s_state[e] = est | (bt << 16);
            }
            __syncthreads();
            if (it < NI) {
                s_pcnt[it] = cnt;
                s_pa[it] = pa;
                s_pb[it] = pb;
            }
        } else
        for (uint32_t it = ltop; it < NI; it += TTPB) {
            uint32_t e = STD ? it / STD_PMAX : vs_fastdiv(it, P.magic_pmax), pi = it - e * pmax;
            uint32_t cnt = 0, pa = 0, pb = 0;
            const uint32_t est = e < ne ? s_state[e] : 0u;
            if ((est & 3u) == 1u) {
                uint32_t meta = s_meta[e];
                uint32_t rlen = meta & VS_LEN_MASK;
                uint32_t j = (est >> 8) + pi * s;
                if (j + w <= rlen) {
                    uint32_t sr;
                    const uint64_t key = vs_seed_key(s_words, e * wpe * 16u + j, w, &sr);
                    bool ok = true;
                    if ((meta >> 24) & VS_FLAG_INVALID) {
                        if (FAST) {
                            uint32_t lo, hi;
                            ok = vs_seed_limits(s_inv[e], j, w, rlen, &lo, &hi);
                        } else {
                            ok = !vs_seed_dirty(P.rd.mask, (uint64_t)s_gwoff[e] * 16u + j, w);
                        }
                    }
                    if (ok) cnt = vs_probe(P.idx, key, sr, &pa, &pb);
                }
            }
            s_pcnt[it] = cnt;
            s_pa[it] = pa;
            s_pb[it] = pb;
        }
        __syncthreads();
        if (debug_stop == 2u) continue;
        // ---- P2: inclusive scan of the postings per probe, s_pcnt[0..NI)
        {
            const uint32_t chunk = (NI + TTPB - 1u) / TTPB;
            const uint32_t b = ltop * chunk;
            uint32_t local = 0;
            for (uint32_t i = 0; i < chunk; i++)
                if (b + i < NI) local += s_pcnt[b + i];
            const uint32_t incl = vs_wave_scan_add(local);
            const uint32_t lane = ltop & 63u;
            if (lane == 63u) s_misc[ltop >> 6] = incl;
            __syncthreads();
            uint32_t off = incl - local;
            for (uint32_t wv = 0; wv < (ltop >> 6); wv++) off += s_misc[wv];
            for (uint32_t i = 0; i < chunk; i++)
                if (b + i < NI) {
                    off += s_pcnt[b + i];
                    s_pcnt[b + i] = off;
                }
        }
        __syncthreads();
        if (debug_stop == 3u) continue;
        // ---- P3: one thread per posting.  Expansion of the per-probe posting counts (CSR-style
        // frontier expansion) in chunks of KCHUNK postings: every probe marks the first position it
        // owns in the chunk, a workgroup-wide running maximum fills the gaps, and each thread ends
        // up with the owners of its KPPT consecutive postings in registers.
        {
        const uint32_t total = s_pcnt[NI - 1u];
        if (count_postings && tid == 0) atomicAdd((unsigned long long *)(P.slow_count + 2), (unsigned long long)total);
        for (uint32_t c0 = 0; c0 < total; c0 += KCHUNK) {
            for (uint32_t i = tid; i < KCHUNK; i += TTPB) s_owner[i] = 0;
            __syncthreads();
            for (uint32_t it = tid; it < NI; it += TTPB) {
                const uint32_t incl = s_pcnt[it], excl = s_pcnt[(int)it - 1];
                if (incl > excl) {
                    const uint32_t lo = excl > c0 ? excl : c0;
                    const uint32_t hi = incl < c0 + KCHUNK ? incl : c0 + KCHUNK;
                    if (lo < hi) s_owner[lo - c0] = it + 1u;
                }
            }
            __syncthreads();
            uint32_t own[KPPT];
            uint32_t run = 0;
#pragma unroll
            for (uint32_t k2 = 0; k2 < KPPT; k2++) {
                const uint32_t v = s_owner[tid * KPPT + k2];
                run = v > run ? v : run;
                own[k2] = run;
            }
            const uint32_t incl = vs_wave_scan_max(run);
            const uint32_t lane = tid & 63u;
            if (lane == 63u) s_misc[4u + (tid >> 6)] = incl;
            uint32_t carry = __shfl_up(incl, 1, 64);
            if (lane == 0u) carry = 0;
            __syncthreads();
            {   // the wavefronts before this one (TTPB = 256: at most three)
                const uint32_t wv = tid >> 6;
#pragma unroll
                for (uint32_t pw = 0; pw + 1u < TTPB / 64u; pw++) {
                    const uint32_t mw = s_misc[4u + pw];
                    if (wv > pw) carry = mw > carry ? mw : carry;
                }
            }
            // The thread's KPPT postings go through stages with every stage done for all of them before
            // the next one starts: A) which posting (LDS) and its record (one global load for
            // multi-posting seeds; the node header for single ones), B) text windows + decision.
            bool live[KPPT];
            uint32_t p_e[KPPT], p_j[KPPT], p_node[KPPT], p_pos[KPPT], p_opp[KPPT];
            VsNodeMeta p_nm[KPPT];
#pragma unroll
            for (uint32_t k2 = 0; k2 < KPPT; k2++) {
                const uint32_t t = c0 + tid * KPPT + k2;
                live[k2] = t < total;
                const uint32_t it = live[k2] ? (own[k2] > carry ? own[k2] : carry) - 1u : 0u;
                const uint32_t excl = s_pcnt[(int)it - 1];
                const uint32_t cnt = s_pcnt[it] - excl;
                const uint32_t pa = s_pa[it], pb = s_pb[it];
                const uint32_t e = STD ? it / STD_PMAX : vs_fastdiv(it, P.magic_pmax), pi = it - e * pmax;
                p_e[k2] = e;
                // the probe's read offset, and above it (bits 16..) its distance from the probe before it: a match that reaches
                // that far down holds the earlier probe, which credits it
                uint32_t gap = s;
                {
                    const uint32_t est = s_state[e];
                    if (ADAPT) {
                        const uint32_t D = est >> 20, t = (est >> 16) & 15u;
                        p_j[k2] = ((est >> 8) & 0xFFu) + pi * s - (pi >= t ? D : 0u);
                        if (pi && pi == t) gap = s - D;
                    } else {
                        p_j[k2] = (est >> 8) + pi * s;
                    }
                }
                uint32_t node = pa, pos = pb & 0x7FFFFFFFu, opp = pb >> 31;
                p_nm[k2].woff = 0; p_nm[k2].len = 0;
                if (live[k2] && cnt != 1u) {  // (the record carries the node header: no second round trip)
                    const VsPosting po = vs_posting_unpack(P.idx.postings[pa + (t - excl)]);
                    node = po.node; pos = po.pos; opp = po.strand ^ (pb >> 31);
                    p_nm[k2].woff = po.woff; p_nm[k2].len = po.len;
                } else if (live[k2]) {
                    if (P.shortcut && s <= w && pi) {
                        // Overlapping seeds (s <= w): if the previous probe of this end holds the single
                        // posting one stride back on the same diagonal, the bases in between match too,
                        // so that probe (or an earlier one) owns this match -- no memory traffic needed.
                        const uint32_t excl2 = s_pcnt[(int)it - 2];  // (pi != 0, so it >= 1)
                        if (excl - excl2 == 1u && s_pa[it - 1u] == node) {
                            const uint32_t pbp = s_pb[it - 1u];
                            const uint32_t want = opp ? pos + gap : pos - gap;
                            if ((pbp >> 31) == opp && (pbp & 0x7FFFFFFFu) == want && (opp || pos >= gap)) live[k2] = false;
                        }
                    }
                    if (live[k2]) p_nm[k2] = P.idx.meta[node];
                }
                p_node[k2] = node; p_pos[k2] = pos; p_opp[k2] = opp;
                if (ADAPT) p_j[k2] |= gap << 16;
            }
#pragma unroll
            for (uint32_t k2 = 0; k2 < KPPT; k2++) {
                if (!live[k2]) continue;
                const uint32_t e = p_e[k2], j = ADAPT ? p_j[k2] & 0xFFFFu : p_j[k2], node = p_node[k2], opp = p_opp[k2];
                const uint32_t gap = ADAPT ? p_j[k2] >> 16 : s;
                VsNodeMeta nm = p_nm[k2];
                const uint32_t meta = s_meta[e];
                const uint32_t rlen = meta & VS_LEN_MASK;
                // either strand off the one (wave-uniform) text base: the reverse complements lie rc_delta words on
                const uint32_t *tw = P.idx.fwd_words;
                const uint32_t q = opp ? nm.len - p_pos[k2] - w : p_pos[k2];
                uint32_t a, qa, len;
                if (FAST) {
                    // the clean stretch [lo, hi) of the read around the seed bounds the match
                    uint32_t lo = 0u, hi = rlen;
                    if ((meta >> 24) & VS_FLAG_INVALID) vs_seed_limits(s_inv[e], j, w, rlen, &lo, &hi);
                    uint32_t cl = gap < j - lo ? gap : j - lo;
                    cl = cl < q ? cl : q;
                    uint32_t rem = hi - j - wv;
                    const uint32_t dr = nm.len - q - wv;
                    rem = rem < dr ? rem : dr;
                    uint32_t left = 0u, ext = 0u;
                    const uint32_t tb = (nm.woff + (opp ? P.idx.rc_delta : 0u)) * 16u;
                    if (MODE == 2) vs_agree_long(s_words, e * wpe * 16u, tw, tb + q, cl, tb + q + wv, rem, j, wv, &left, &ext);
                    else vs_agree_fast<STD>(s_words, e * wpe * 16u, tw, tb + q, cl, tb + q + wv, rem, j, wv, &left, &ext);
                    len = left + wv + ext;
                    if (left >= gap || len < K) continue;  // an earlier probe lies inside this match and owns it / too short
                    a = j - left;
                    qa = q - left;
                } else {
                    const uint32_t tb = (nm.woff + (opp ? P.idx.rc_delta : 0u)) * 16u;
                    const uint32_t *mk = ((meta >> 24) & VS_FLAG_INVALID) ? P.rd.mask : nullptr;
                    if (!vs_extend(s_words, e * wpe * 16u, rlen, tw, tb, nm.len, j, q, wv, s, K, mk, (uint64_t)s_gwoff[e] * 16u, &a, &qa, &len))
                        continue;
                }
                credit(e, node, nm.len, len - K + 1u, opp ? nm.len - qa - len : qa, a);
            }
            __syncthreads();  // the owner array is reused by the next chunk
        }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wavefront's LDS-direct loads of the next tile's words are in
        __syncthreads();
        if (debug_stop == 4u) continue;
        // ---- P4: acceptance test per table slot; an accepted node takes the next position of its end's list
        // (r6) The lists are PACKED: first every slot is judged and the ends' list lengths counted (the slot keeps its
        // position, its count is not needed any more), then ONE wavefront turns the lengths -- rounded up to whole quads -- into
        // offsets by a prefix sum over the ends, then the nodes go to their places.  An end with more than LCAP nodes, or one
        // that does not fit what is left of the tile's region (ept * LC words), is an overflow end.
        // (the thread index is laundered through an empty asm: what this phase derives from it -- slot and end indices,
        // LDS and global addresses -- then cannot be hoisted out of the tile loop, where it would sit in vector registers
        // across P3, the phase the kernel's 96-register budget is made for)
        uint32_t lt = tid;
        asm volatile("" : "+v"(lt));
        // (compile-time shapes: a thread's four slots -- key and position -- stay in registers across the prefix sum)
        constexpr uint32_t SPT = STD ? C_POOL / TTPB : 1u;
        uint32_t r_key[SPT], r_at[SPT];
        auto judge = [&](uint32_t i, uint32_t key) -> uint32_t {  // the position of slot i's node in its end's list, or none
            const uint32_t e = key >> 25, node = key & 0x01FFFFFFu;
            const uint32_t rlen = s_meta[e] & VS_LEN_MASK;
            const uint32_t hj = s_hminj[i];
            const uint32_t nlen = FAST ? hj >> AB : P.idx.meta[node].len;
            const uint32_t minj = FAST ? hj & ((1u << AB) - 1u) : hj;
            if (FAST ? vs_accept32(s_hcnt[i], s_hminp[i], minj, nlen, rlen, K) : vs_accept(s_hcnt[i], s_hminp[i], minj, nlen, rlen, K))
                return atomicAdd(&s_ns[e], 1u);
            return 0xFFFFFFFFu;
        };
        if (STD) {
#pragma unroll
            for (uint32_t j = 0; j < SPT; j++) {
                const uint32_t i = lt + j * TTPB;
                r_key[j] = s_hkey[i];
                r_at[j] = r_key[j] != EMPTY_NODE ? judge(i, r_key[j]) : 0xFFFFFFFFu;
            }
        } else {
            for (uint32_t i = lt; i < pool; i += TTPB) {
                const uint32_t key = s_hkey[i];
                if (key != EMPTY_NODE) s_hcnt[i] = judge(i, key);
            }
        }
        __syncthreads();
        if (lt < 64u) {  // two ends per lane (ept <= TTPB / 2 = 128)
            const uint32_t e0 = 2u * lt, e1 = e0 + 1u;
            const uint32_t st0 = e0 < ne ? s_state[e0] : 0u, st1 = e1 < ne ? s_state[e1] : 0u;
            const uint32_t n0 = (st0 & 3u) == 1u ? s_ns[e0] : 0u, n1 = (st1 & 3u) == 1u ? s_ns[e1] : 0u;
            bool over0 = n0 > LCAP, over1 = n1 > LCAP;
            const uint32_t q0 = over0 ? 0u : (n0 + 3u) >> 2, q1 = over1 ? 0u : (n1 + 3u) >> 2;
            const uint32_t incl = vs_wave_scan_add(q0 + q1);
            const uint32_t capq = (ept * LC - C_TRIM) >> 2;
            const uint32_t o0 = incl - q0 - q1, o1 = o0 + q0;
            over0 |= o0 + q0 > capq;
            over1 |= o1 + q1 > capq;
            // (an end whose nodes are not placed -- overflow, or not an end of a used pair -- says so in its offset)
            if (e0 < ne) { s_pcnt[e0] = over0 || (st0 & 3u) != 1u ? 0xFFFFFFFFu : 4u * o0; if (over0) s_state[e0] = st0 | 2u; }
            if (e1 < ne) { s_pcnt[e1] = over1 || (st1 & 3u) != 1u ? 0xFFFFFFFFu : 4u * o1; if (over1) s_state[e1] = st1 | 2u; }
            if (lt == 63u) s_misc[8] = 4u * (incl < capq ? incl : capq);  // words of the tile's packed lists
        }
        __syncthreads();
        uint32_t *s_off = s_pcnt;  // (the scan of P2 is done with)
        if (STD) {
#pragma unroll
            for (uint32_t j = 0; j < SPT; j++)
                if (r_at[j] != 0xFFFFFFFFu) {
                    const uint32_t o = s_off[r_key[j] >> 25];
                    if (o != 0xFFFFFFFFu) s_list[o + r_at[j]] = r_key[j] & 0x01FFFFFFu;
                }
        } else {
            for (uint32_t i = lt; i < pool; i += TTPB) {
                const uint32_t key = s_hkey[i];
                if (key != EMPTY_NODE) {
                    const uint32_t at = s_hcnt[i], o = s_off[key >> 25];
                    if (at != 0xFFFFFFFFu && o != 0xFFFFFFFFu) s_list[o + at] = key & 0x01FFFFFFu;
                }
            }
        }
        __syncthreads();
        // pairs with an overflowed end go to the slow list
        if (lt < npair) {
            uint32_t st = s_state[2 * lt] | s_state[2 * lt + 1];
            if ((st & 1u) && (st & 2u)) {
                uint32_t at = atomicAdd(P.slow_count, 1u);
                P.slow_list[at] = s_gend[2 * lt] >> 1;
                s_state[2 * lt] |= 2u;
                s_state[2 * lt + 1] |= 2u;
            }
        }
        __syncthreads();
        if (debug_stop == 5u) continue;
        // ---- P5: hand the accepted lists to the counter kernels, tile order; length 0 for ends of dropped pairs and of
        // pairs the slow path takes
        if (accumulate) {
            if (!P.out_rows) {  // the tile's packed lists as they lie in LDS: one coalesced stretch
                uint32_t *ol = P.out_lists + (uint64_t)tile * ept * LC;
                const uint32_t total = s_misc[8];
                for (uint32_t i = lt; i < total; i += TTPB) ol[i] = s_list[i];
                for (uint32_t i = lt; i < ne; i += TTPB) {
                    const uint32_t n = (s_state[i] & 3u) == 1u ? s_ns[i] : 0u;
                    P.out_counts[(uint64_t)tile * ept + i] = n ? n | (s_off[i] >> 2) << 8 : 0u;
                }
            } else {  // a row per end (+ its quad of the second array); a lane per (end, quad), only the quads that hold nodes
                uint32_t *ol = P.out_lists + (uint64_t)tile * ept * LC, *oh = P.out_lists_hi + (uint64_t)tile * ept * 4u;
                for (uint32_t i = lt; i < ne * (LCAP / 4u); i += TTPB) {
                    const uint32_t e = i / (LCAP / 4u), q4 = 4u * (i - e * (LCAP / 4u));
                    const uint32_t n = (s_state[e] & 3u) == 1u ? s_ns[e] : 0u;
                    if (q4 < n) {
                        const uint32_t *src = s_list + s_off[e] + q4;
                        *(VsQuad *)(q4 < LC ? ol + e * LC + q4 : oh + e * 4u) = VsQuad{src[0], src[1], src[2], src[3]};
                    }
                }
                for (uint32_t i = lt; i < ne; i += TTPB)
                    P.out_counts[(uint64_t)tile * ept + i] = (s_state[i] & 3u) == 1u ? s_ns[i] : 0u;
            }
        }
        if (want_dbg) {
            for (uint32_t i = tid; i < ne; i += TTPB) {
                if (s_state[i] & 2u) continue;  // the slow kernel reports these
                uint32_t n = s_ns[i];
                const uint64_t ge = s_gend[i];
                P.dbg_counts[ge] = n;
                for (uint32_t k2 = 0; k2 < n && k2 < P.dbg_cap; k2++) P.dbg_lists[ge * P.dbg_cap + k2] = s_list[s_off[i] + k2];
            }
        }
    }
    __syncthreads();
    if (tid < 3 && P.stats && s_misc[12 + tid]) atomicAdd(&P.stats[tid], (unsigned long long)s_misc[12 + tid]);
}

// ---- K4: counters ------------------------------------------------------------------------------------
// PE_Inference.py:174-188 from the per-end lists k_pe_tiles wrote (tile order = locus order).  A pair
// with lists l (nl nodes) and r (nr nodes) makes nl*nr node_mat increments and, per end, n(n+1)/2
// short_mat increments: positions a <= b give the cell (min, max) of the two node ids -- the
// reference's "i <= i2 over ascending indices" (:174-184).
// Scattered global atomics run at ~2e10/s chip-wide and would bound the whole step (5.4e8 increments
// at configs[2]), so increments are summed per cell in LDS before they reach memory.  Two loop orders:
//   * k_pe_accumulate (graphs of at most 46 340 nodes): pair-major.  A workgroup (1024 threads, one per CU, a
//     contiguous run of pairs = a few loci) sums in a 16k-slot cell table and issues ONE global atomic per cell when the
//     table is written out.  One wavefront expands 64 pairs at a time, one lane per run of at most four increments.
//   * the row owners below (larger graphs): output-major.  There a round of locus-ordered pairs brings more distinct
//     cells than the table holds, and the same cell returns from loci hundreds of rounds apart.
#ifndef ACC_TPB
#define ACC_TPB 1024
#endif
#ifndef ACC_BITS
#define ACC_BITS 14
#endif
#define ACC_SLOTS (1u << ACC_BITS)
#define ACC_LDS_BYTES ((2u * ACC_SLOTS + (ACC_TPB / 64) * 66u + (LCAP + 1u) + (LCAP + 1u) * ACC_GMAX + 4u) * 4u)
// Work units: a list row is cut into runs of at most ACC_RUN partners, one lane per run, so that
// every lane of a wavefront has about the same (small, fully unrolled) amount of work:
//   node_mat : left node a against right positions [4c, 4c+4)           -> nl * ceil(nr/4) runs
//   short_mat: list position a against positions [a+4c, a+4c+4) (b >= a) -> g(n) runs per list,
//              g(n) = sum_{m=1..n} ceil(m/4)
#ifndef ACC_RUN
#define ACC_RUN 4u   // 4 or 8 (8: two 16-byte partner loads per run; measured r5, see profiles/EXPERIMENTS.md)
#endif
#define ACC_GMAX 60u  // g(LCAP = 20) for runs of 4 (runs of 8 need 36)
// The cell table: 16 k slots, 32-bit keys (k_pe_accumulate: mat * N*N + x * N + y, while 2*N*N fits 32 bits, N <= 46340;
// the row owners: cell index relative to the strip's first row).  The 16 cells of one 64-byte stretch of a matrix row
// sit in 16 NEIGHBOURING slots (the hash picks a group of 16 slots from the key >> 4, the low four bits pick the slot
// inside it; a taken slot sends the probe to the same position of the next group).  A write-out walks the slots in
// order, a lane per slot, so the lanes of a wavefront that hold cells of one stretch issue their atomics side by side
// -- and integer atomics, like the float ones of MI355X_MICROARCH.md, leave the L2 as one memory-side request per
// 64-byte stretch a wave instruction touches.
template <uint32_t BITS>
struct CellTable {  // 1 << BITS slots
    static constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    __device__ static uint32_t key(uint32_t mat, uint32_t x, uint32_t y, uint32_t N) { return (mat * N + x) * N + y; }
    __device__ static uint32_t slot(uint32_t k) { return ((((k >> 4) * 0x9E3779B1u) >> (36u - BITS)) << 4) | (k & 15u); }
    __device__ static uint32_t next(uint32_t at) { return (at + 16u) & ((1u << BITS) - 1u); }
};
typedef CellTable<ACC_BITS> Acc32;  // k_pe_accumulate's

// The slot a key hashes to is empty or holds another cell: claim / probe on.  Kept out of line of the common case (the
// cell is already in the table), which stays a short straight-line sequence.  false: no place within eight probes.
template <typename TB>
__device__ __forceinline__ bool vs_cell_claim(uint32_t *s_key, uint32_t *s_cnt, uint32_t *s_used, uint32_t key, uint32_t at, uint32_t wgt) {
    for (uint32_t pr = 0; pr < 8u; pr++) {
        uint32_t kx = s_key[at];
        if (kx == TB::EMPTY) {
            kx = atomicCAS(&s_key[at], TB::EMPTY, key);
            if (kx == TB::EMPTY) { atomicAdd(s_used, 1u); kx = key; }
        }
        if (kx == key) {
            atomicAdd(&s_cnt[at], wgt);
            return true;
        }
        at = TB::next(at);
    }
    return false;
}

__global__ void __launch_bounds__(ACC_TPB)
k_pe_accumulate(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ counts, uint64_t n_slots_pairs,
                uint32_t pairs_per_wg, uint32_t N, uint32_t use_table, uint32_t fill_limit,
                uint32_t *__restrict__ node_mat, uint32_t *__restrict__ short_mat, uint32_t *__restrict__ queue,
                uint32_t *__restrict__ dbg, uint32_t ppw, uint32_t ept) {  // VS_DEBUG_ACC: [0] increments past the table, [1] write-outs, [2] cells written, [3] rounds
    uint32_t *s_key = vs_lds;                      // [ACC_SLOTS]
    uint32_t *s_cnt = vs_lds + ACC_SLOTS;          // [ACC_SLOTS]
    uint32_t(*s_pref)[66] = (uint32_t(*)[66])(vs_lds + 2u * ACC_SLOTS);  // [ACC_TPB / 64][66]
    uint32_t *s_g = vs_lds + 2u * ACC_SLOTS + (ACC_TPB / 64) * 66u;      // [LCAP + 1]: g(n)
    uint32_t *s_ua = s_g + (LCAP + 1u);                                   // [LCAP + 1][ACC_GMAX]: run -> position a
    uint32_t &s_used = s_ua[(LCAP + 1u) * ACC_GMAX];
    uint32_t &s_lost = s_ua[(LCAP + 1u) * ACC_GMAX + 1u];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t NN = N * N;  // (use_table: 2 * N * N fits 32 bits)
    for (uint32_t i = tid; i < ACC_SLOTS; i += ACC_TPB) { s_key[i] = Acc32::EMPTY; s_cnt[i] = 0; }
    if (tid <= LCAP) {
        uint32_t gsum = 0;
        for (uint32_t m = 1; m <= tid; m++) gsum += (m + ACC_RUN - 1u) / ACC_RUN;
        s_g[tid] = gsum;
        // runs of an n-list in order: position a = 0 first (n - a partners), then a = 1, ...
        uint32_t r = 0;
        for (uint32_t a2 = 0; a2 < tid; a2++)
            for (uint32_t c = 0; c < (tid - a2 + ACC_RUN - 1u) / ACC_RUN; c++) s_ua[tid * ACC_GMAX + r++] = a2;
    }
    if (tid == 0) { s_used = 0; s_lost = 0; }
    __syncthreads();
    // chunks of pairs_per_wg pairs: chunk blockIdx.x, or (queue) whichever chunk is next when this
    // workgroup is free -- the table then lives across chunks and is written out on fill only
    uint32_t &s_chunk = s_ua[(LCAP + 1u) * ACC_GMAX + 2u];
    const uint32_t ppt = ept >> 1, region_q = (ept * LC) >> 2;  // pairs per tile of the mapping kernel; quads of a tile's region
    // every cell of the table to its counter (one global atomic per cell), the table emptied
    auto write_out = [&]() {
        for (uint32_t i = tid; i < ACC_SLOTS; i += ACC_TPB) {
            const uint32_t key = s_key[i];
            if (key != Acc32::EMPTY) {
                if (use_table != 3u)  // (3: timing experiment without the write-outs)
                    atomicAdd(key >= NN ? short_mat + (key - NN) : node_mat + key, s_cnt[i]);
                s_key[i] = Acc32::EMPTY;
                s_cnt[i] = 0;
            }
        }
        if (tid == 0) { s_used = 0; s_lost = 0; }
        __syncthreads();
    };
    for (;;) {
    uint64_t chunk = blockIdx.x;
    if (queue) {
        __syncthreads();
        if (tid == 0) s_chunk = atomicAdd(queue, 1u);
        __syncthreads();
        chunk = s_chunk;
    }
    const uint64_t lo = chunk * pairs_per_wg;
    if (lo >= n_slots_pairs) break;
    const uint64_t hi = lo + pairs_per_wg < n_slots_pairs ? lo + pairs_per_wg : n_slots_pairs;
    // a round = ppw pairs per wavefront (64: one per lane; VS_ACC_ROUND for fewer)
    for (uint64_t base = lo; base < hi; base += (ACC_TPB / 64u) * ppw) {
        const uint64_t wbase = base + wv * ppw;             // wave-uniform
        const uint32_t *wcounts = counts + 2u * wbase;
        // the lists are packed per tile of the mapping kernel (k_pe_tiles P4 / P5): where a pair's two lists start, in quads
        // from the region of the wavefront's first tile -- at most 64 pairs further on, so 11 bits each, next to the lengths
        const uint64_t tile0 = wbase / ppt;                 // wave-uniform
        const uint32_t rem0 = (uint32_t)(wbase - tile0 * ppt);
        const uint32_t *wlists = lists + tile0 * ept * LC;
        uint32_t nl = 0, nr = 0, packed = 0;
        if (lane < ppw && wbase + lane < hi) {
            const uint2 c = *(const uint2 *)(wcounts + 2u * lane);
            nl = c.x & 0xFFu; nr = c.y & 0xFFu;
            const uint32_t tq = (rem0 + lane) / ppt * region_q;
            packed = nl | nr << 5 | (tq + (c.x >> 8)) << 10 | (tq + (c.y >> 8)) << 21;
        }
        const uint32_t u = nl * ((nr + ACC_RUN - 1u) / ACC_RUN) + s_g[nl] + s_g[nr];
        const uint32_t incl = vs_wave_scan_add(u);
        s_pref[wv][lane + 1u] = incl;
        if (lane == 0) s_pref[wv][0] = 0;
        if (lane == 63u) s_pref[wv][65] = 0xFFFFFFFFu;  // sentinel: the look-ahead below never runs off
        const uint32_t U = __shfl(incl, 63, 64);
        // (the same wavefront wrote and reads s_pref: no workgroup barrier, only the LDS wait)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint32_t cur = 0;  // wave-uniform: first pair whose runs reach into the current window
        // One window of 64 runs: which pair and which of its runs a lane takes, the run's first node and
        // its (up to) four partners.  The two list loads of window i+1 are issued before window i is
        // counted, so their round trip (L2) is covered by the cell-table work instead of preceding it.
        struct Run {
            uint32_t x, mat, bi, be;
            VsQuad yq;
#if ACC_RUN == 8u
            VsQuad yq2;
#endif
            bool ok;
        };
        auto fetch = [&](uint32_t t0) {
            Run R;
            while (s_pref[wv][cur + 1u] <= t0) cur++;
            const uint32_t t_raw = t0 + lane;
            const uint32_t t = t_raw < U ? t_raw : U - 1u;  // (lanes past the end ride along, their result is dropped)
            uint32_t a0 = cur;  // last pair of this wavefront with pref <= t
            a0 += s_pref[wv][a0 + 1u] <= t ? 1u : 0u;
            a0 += s_pref[wv][a0 + 1u] <= t ? 1u : 0u;
            a0 += s_pref[wv][a0 + 1u] <= t ? 1u : 0u;
            while (s_pref[wv][a0 + 1u] <= t) a0++;
            uint32_t r = t - s_pref[wv][a0];
            // list lengths and places of pair a0: lane a0 holds them (ONE cross-lane read, no memory round trip)
            const uint32_t pk = __shfl(packed, (int)a0, 64);
            const uint32_t ql = pk & 31u, qr = (pk >> 5) & 31u, rowl = ((pk >> 10) & 0x7FFu) << 2, rowr = (pk >> 21) << 2;
            R.ok = t_raw < U;
            const uint32_t cq = (qr + ACC_RUN - 1u) / ACC_RUN;
            uint32_t off;
            if (r < ql * cq) {  // node_mat: left node a, right positions of run c
                const uint32_t a = cq == 1u ? r : cq == 2u ? r >> 1 : cq == 4u ? r >> 2 : cq == 3u ? (r * 43691u) >> 17 : (r * 52429u) >> 18;
                R.x = wlists[rowl + a]; R.mat = 0u; off = rowr; R.bi = ACC_RUN * (r - a * cq); R.be = qr;
            } else {
                r -= ql * cq;
                uint32_t n = ql;
                off = rowl;
                const uint32_t gl = s_g[ql];
                if (r >= gl) { r -= gl; n = qr; off = rowr; }
                const uint32_t a = s_ua[n * ACC_GMAX + r];
                const uint32_t crun = r - (s_g[n] - s_g[n - a]);  // runs of positions before a: g(n) - g(n-a)
                R.x = wlists[off + a]; R.mat = 1u; R.bi = a + ACC_RUN * crun; R.be = n;
            }
            // up to ACC_RUN partners, loaded together (reading a few words past `be` -- the list's padding, the next
            // list -- stays inside the lists buffer, which carries padding at its end, and is ignored)
            R.yq = *(const VsQuad *)(wlists + off + R.bi);  // one 16-byte load
#if ACC_RUN == 8u
            R.yq2 = *(const VsQuad *)(wlists + off + R.bi + 4u);
#endif
            return R;
        };
        // (r6: the list loads of two and three windows ahead instead of one were measured -- 2.23 / 2.25 / 2.31 ms at configs[2]: the
        // kernel does not wait on these loads; profiles/EXPERIMENTS.md)
        Run nxt;
        nxt.ok = false;
        if (U) nxt = fetch(0u);
        for (uint32_t t0 = 0; t0 < U; t0 += 64u) {
            const Run c = nxt;
            if (t0 + 64u < U) nxt = fetch(t0 + 64u);
            if (!c.ok) continue;
            const uint32_t x = c.x, mat = c.mat, bi = c.bi, be = c.be;
#if ACC_RUN == 8u
            const uint32_t ys[ACC_RUN] = {c.yq.x, c.yq.y, c.yq.z, c.yq.w, c.yq2.x, c.yq2.y, c.yq2.z, c.yq2.w};
#else
            const uint32_t ys[ACC_RUN] = {c.yq.x, c.yq.y, c.yq.z, c.yq.w};
#endif
            if (use_table == 2u) {
                if ((ys[0] ^ ys[1] ^ ys[2] ^ ys[3] ^ x) == 0xDEADBEEFu) atomicAdd(&s_lost, 1u);  // (timing experiment: decode only)
            } else if (use_table) {
                // the four cells' slots are read together (independent LDS loads), then counted; a
                // slot that does not hold the cell yet goes the slow way (claim / probe / global)
                uint32_t key[ACC_RUN], seen[ACC_RUN], at[ACC_RUN];
#pragma unroll
                for (uint32_t j = 0; j < ACC_RUN; j++) {
                    const uint32_t yv = ys[j];
                    const uint32_t cx = (mat && yv < x) ? yv : x, cy = (mat && yv < x) ? x : yv;
                    key[j] = Acc32::key(mat, cx, cy, N);
                    at[j] = Acc32::slot(key[j]);
                    seen[j] = s_key[at[j]];
                }
#pragma unroll
                for (uint32_t j = 0; j < ACC_RUN; j++) {
                    if (bi + j >= be) continue;  // (j = 0 always counts)
                    if (seen[j] == key[j]) {
                        atomicAdd(&s_cnt[at[j]], 1u);
                    } else if (!vs_cell_claim<Acc32>(s_key, s_cnt, &s_used, key[j], at[j], 1u)) {
                        atomicAdd(&s_lost, 1u);
                        atomicAdd(key[j] >= NN ? short_mat + (key[j] - NN) : node_mat + key[j], 1u);
                    }
                }
            } else {
                // (VS_NO_AGG=1: every increment a global atomic)
#pragma unroll
                for (uint32_t j = 0; j < ACC_RUN; j++) {
                    if (bi + j >= be) continue;
                    const uint32_t yv = ys[j];
                    const uint32_t cx = (mat && yv < x) ? yv : x, cy = (mat && yv < x) ? x : yv;
                    atomicAdd((mat ? short_mat : node_mat) + (uint64_t)cx * N + cy, 1u);
                }
            }
        }
        __syncthreads();
        const bool spill = s_used > fill_limit || s_lost > 4096u;
        if (dbg && tid == 0) {
            atomicAdd(dbg + 3, 1u);
            if (spill) { atomicAdd(dbg + 0, s_lost); atomicAdd(dbg + 1, 1u); atomicAdd(dbg + 2, s_used); }
        }
        __syncthreads();
        // (everything goes, also the cells of the locus still being worked on: keeping those across
        // write-outs was measured -- 3.1 -> 3.9 ms -- the table is only fast while nearly empty, when a
        // cell sits in the first slot its key hashes to)
        if (spill) write_out();
    }
    if (!queue) break;
    }
    __syncthreads();
    if (dbg && tid == 0) atomicAdd(dbg + 0, s_lost);
    write_out();
}

// ---- which tiles of the counters a block touches (vs_pe_count_tracked) -------------------------------------------------
// A pair adds to node_mat[l][r] for l in its left list, r in its right list, and to short_mat[min][max] for the pairs of
// nodes of either list (PE_Inference.py:174-188): one lane per pair marks the 64 x 64 tiles those cells lie in, from the
// same list rows k_pe_accumulate counts (the overflow kernels and the row owners mark theirs where they add).  A caller that zeroes its
// counters before every block then zeroes these tiles only (k_zero_tiles) -- a few per cent of a 50 k-node matrix.
__global__ void __launch_bounds__(256) k_mark_tiles(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ counts,
                                                   uint64_t n_slots_pairs, uint8_t *__restrict__ map, uint32_t T, uint32_t ept) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_slots_pairs) return;
    const uint32_t cl = counts[2u * p], cr = counts[2u * p + 1u], nl = cl & 0xFFu, nr = cr & 0xFFu;
    if (!nl && !nr) return;
    // (packed lists: the region of the pair's tile, the quad offsets of its two ends)
    const uint32_t *region = lists + 2u * p / ept * ept * LC, *rl = region + 4u * (cl >> 8), *rr = region + 4u * (cr >> 8);
    uint8_t *smap = map + (uint64_t)T * T;
    // The distinct tile coordinates (node >> 6) of a list, up to eight of them, in registers (every index below is a
    // compile-time constant: no scratch memory).  The rows come as 16-byte loads, only as many as the list is long.  A list
    // with more than eight distinct coordinates takes the plain way.
    uint32_t tl[8], tr[8], kl = 0, kr = 0;
    bool over = false;
    auto put = [&](uint32_t (&tt)[8], uint32_t &k, uint32_t t) {
        bool seen = false;
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) seen |= j < k && tt[j] == t;
        if (seen) return;
        if (k >= 8u) { over = true; return; }
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++)
            if (j == k) tt[j] = t;
        k++;
    };
#pragma unroll
    for (uint32_t j = 0; j < 8u; j++) { tl[j] = 0u; tr[j] = 0u; }
#pragma unroll
    for (uint32_t q = 0; q < LCAP / 4u; q++) {
        if (4u * q < nl) {
            const VsQuad v = *(const VsQuad *)(rl + 4u * q);
            const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++)
                if (4u * q + i < nl) put(tl, kl, e[i] >> 6);
        }
        if (4u * q < nr) {
            const VsQuad v = *(const VsQuad *)(rr + 4u * q);
            const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++)
                if (4u * q + i < nr) put(tr, kr, e[i] >> 6);
        }
    }
    if (over) {
        for (uint32_t i = 0; i < nl; i++)
            for (uint32_t j = 0; j < nr; j++) vs_mark_tile(map, T, 0u, rl[i], rr[j]);
        for (uint32_t side = 0; side < 2u; side++) {
            const uint32_t *row = side ? rr : rl;
            const uint32_t n = side ? nr : nl;
            for (uint32_t i = 0; i < n; i++)
                for (uint32_t j = 0; j < n; j++) {
                    const uint32_t x = row[i], y = row[j];
                    if (x <= y) vs_mark_tile(map, T, 1u, x, y);
                }
        }
        return;
    }
#pragma unroll
    for (uint32_t a = 0; a < 8u; a++)
#pragma unroll
        for (uint32_t b = 0; b < 8u; b++)
            if (a < kl && b < kr) {
                const uint64_t t = (uint64_t)tl[a] * T + tr[b];
                if (!map[t]) map[t] = 1;
            }
    // short_mat: a cell sits at (smaller node, larger node), its tile at (smaller, larger) tile coordinates
#pragma unroll
    for (uint32_t a = 0; a < 8u; a++)
#pragma unroll
        for (uint32_t b = 0; b < 8u; b++) {
            if (a < kl && b < kl && tl[a] <= tl[b]) {
                const uint64_t t = (uint64_t)tl[a] * T + tl[b];
                if (!smap[t]) smap[t] = 1;
            }
            if (a < kr && b < kr && tr[a] <= tr[b]) {
                const uint64_t t = (uint64_t)tr[a] * T + tr[b];
                if (!smap[t]) smap[t] = 1;
            }
        }
}

// every marked tile of both matrices to zero, the map cleared.  (r6) A wavefront looks at 64 map bytes at once (a lane each) and
// zeroes the marked ones among them, a lane per column -- one workgroup per tile meant 1.45 M launches of which 2 % had work
// (0.33 ms at configs[4]).
__global__ void __launch_bounds__(64) k_zero_tiles(uint32_t *__restrict__ node_mat, uint32_t *__restrict__ short_mat, uint32_t N, uint32_t T,
                                                  uint8_t *__restrict__ map, uint64_t n_tiles) {
    const uint64_t t0 = (uint64_t)blockIdx.x * 64u;
    const uint64_t mine = t0 + threadIdx.x;
    const bool marked = mine < n_tiles && map[mine] != 0;
    unsigned long long todo = __ballot(marked);
    if (marked) map[mine] = 0;
    while (todo) {
        const uint32_t k = (uint32_t)__ffsll((long long)todo) - 1u;
        todo &= todo - 1ull;
        const uint64_t t = t0 + k;
        const uint32_t mat = (uint32_t)(t / ((uint64_t)T * T));
        const uint64_t r = t - (uint64_t)mat * T * T;
        const uint32_t x0 = (uint32_t)(r / T) << 6, y = (((uint32_t)(r % T)) << 6) + threadIdx.x;
        uint32_t *m = mat ? short_mat : node_mat;
        if (y < N)
            for (uint32_t x = x0; x < x0 + 64u && x < N; x++) m[(uint64_t)x * N + y] = 0u;
    }
}

// ---- both matrices by ROW OWNERS (graphs beyond the one-table shape of the cell table) ---------------------------------
// In locus order a round of 1 024 pairs of a 54 k-node graph brings ~30 k distinct cells, more than an LDS table holds,
// and the same cell comes back from loci hundreds of rounds apart: 4.4e9 increments left the pair-major kernel as 7.9e8
// memory-side atomics for 2.9e7 distinct cells (configs[4]; tools/cell_probe.py).  Turned round -- output-major -- the
// sums fit: ONE matrix row has a few hundred distinct cells however many pairs add to it.  So the lists are transposed
// (row x -> the items whose list holds x: a counting sort of one word per listed node), a workgroup owns a strip of rows
// at a time, adds the partner lists of the strip's items into its LDS cell table, and writes every cell ONCE.
// PE_Inference.py:174-188 is the loop nest being reordered; integer sums do not care.
//   node_mat  (mode 0): item = pair, row x in its LEFT list, partners = its RIGHT list             (:185-188)
//   short_mat (mode 1): item = a DISTINCT end list of the block, row x in it, partners = its nodes y >= x, added as often
//                       as the block holds that list (7 % of the ends at configs[4] bring a list of their own)   (:174-184)
//   k_list_owners   every end against the block's list table: who stands for a list, who repeats one (gown, mult)
//   k_owners_mult   the multiplicities from the table to the owning ends; k_owners_collect: the owning ends, in end order
//   k_rows_count    histogram of the items' lists per chunk (LDS, 16-bit counts), added to row_count
//   (scan)          row_ptr = exclusive sums
//   k_rows_fill     the same histogram again; a chunk reserves its stretch of every row it holds with one global atomic,
//                   then places its items through LDS cursors.  An entry names the ROW OF THE LIST to add -- the owner of
//                   an equal list, so that a matrix row reads a few hundred cached rows, not one per pair -- and its length
//   k_rows_sum      strips of rows off a queue; a lane per (entry of the row, quad of the list it names)
#define ROWS_CHUNK 16384u  // pairs per chunk: a node is in a list at most once, so a 16-bit count (<= 32 768 ends) cannot wrap
#define ROWS_CHUNK1 8192u  // owning ends per chunk (mode 1: few items, spread over more workgroups)
#define ROWS_TPB 1024u
#define ROWS_CAP 4096u     // distinct rows of a chunk that get an LDS cursor (the rest: a global atomic per entry)
#define ROWS_KEYS 65536u   // rows per histogram pass (larger graphs take several passes over the lists)
#define ROWS_SUB ((1u << 26) - 1024u)  // pairs per transposition (an entry names a read end in 27 bits, its list's length in five; row offsets are 32-bit)
static inline size_t rows_lds_bytes(uint32_t n_keys) { return sizeof(uint32_t) * (((size_t)n_keys + 2u) / 2u + ROWS_CAP + ROWS_CAP / 2u + 4u); }

// Which read ends stand for a list of their own, and for how many ends: short_mat takes one weighted pass per DISTINCT
// list of the block, and node_mat's entries name distinct lists (cached rows) instead of one row per pair.  Every end
// looks its list up in a table of the whole block (device memory, one 64-bit word per slot: tag << 32 | owner end + 1,
// claimed by compare-and-swap; a separate word per slot counts the ends).  A lookup reads with plain loads -- a word never
// changes once claimed, a stale zero only sends the lane into a CAS that returns the real word.  Fingerprints are
// order-independent (the lists arrive in no particular order); a tag match is confirmed node by node against the owner's
// row, which is input data.  mult[end] = 0 for an end that is merged into another one; an end that claimed a slot gets the
// slot's count from k_owners_mult; an end that finds no place within LTAB_PROBES slots stands for itself alone.
// gown[end] = the end whose row holds this end's list.  (Merging the ends of a round in LDS first was measured: forward
// reads of one locus repeat each other nine times in ten, their mates rarely; with it the kernel took 5.6 ms at
// configs[4], without 4.1.)
#define LTAB_PROBES 16u
__device__ __forceinline__ void vs_load_list(const uint32_t *__restrict__ row, const uint32_t *__restrict__ hi, uint32_t n, uint32_t (&v)[LCAP]) {
    const VsQuad z{0, 0, 0, 0};
    const VsQuad a = *(const VsQuad *)row, b = n > 4u ? *(const VsQuad *)(row + 4) : z, c = n > 8u ? *(const VsQuad *)(row + 8) : z;
    const VsQuad d = n > 12u ? *(const VsQuad *)(row + 12) : z, f = n > 16u ? *(const VsQuad *)hi : z;
    const uint32_t w[LCAP] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w, f.x, f.y, f.z, f.w};
#pragma unroll
    for (uint32_t i = 0; i < LCAP; i++) v[i] = i < n ? w[i] : 0xFFFFFFFFu;
}
// the same set of n nodes?  (both padded with 0xFFFFFFFF)
__device__ __forceinline__ bool vs_same_list(const uint32_t (&mine)[LCAP], const uint32_t (&other)[LCAP], uint32_t n) {
    bool same = true;
    if (n <= LC) {  // (nearly every list: the first LC positions of either side hold everything)
#pragma unroll
        for (uint32_t k2 = 0; k2 < LC; k2++) {
            bool found = false;
#pragma unroll
            for (uint32_t i = 0; i < LC; i++) found |= mine[i] == other[k2];
            same &= found || k2 >= n;
        }
        return same;
    }
#pragma unroll
    for (uint32_t k2 = 0; k2 < LCAP; k2++) {
        bool found = false;
#pragma unroll
        for (uint32_t i = 0; i < LCAP; i++) found |= mine[i] == other[k2];
        same &= found || k2 >= n;
    }
    return same;
}

__global__ void __launch_bounds__(256)
k_list_owners(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ lists_hi, const uint32_t *__restrict__ counts, uint64_t n_ends, uint32_t *__restrict__ mult,
              uint32_t *__restrict__ gown, unsigned long long *__restrict__ ltab, uint32_t *__restrict__ lmult, uint32_t ltab_bits) {
    const uint64_t e64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e64 >= n_ends) return;
    const uint32_t e = (uint32_t)e64, n = counts[e];
    uint32_t m = n ? 1u : 0u, own = e;
    if (n && ltab) {
        uint32_t mine[LCAP];
        vs_load_list(lists + e64 * LC, lists_hi + e64 * 4u, n, mine);
        unsigned long long f2 = n;
#pragma unroll
        for (uint32_t i = 0; i < LCAP; i++)
            if (i < n) {
                unsigned long long g = (mine[i] + 1ull) * 0x9E3779B97F4A7C15ull;
                g ^= g >> 29;
                f2 += g * 0xBF58476D1CE4E5B9ull;  // (a sum: the order of the nodes does not matter)
            }
        const uint32_t tag = ((uint32_t)(f2 >> 32) & ~31u) | (n - 1u);  // (the list length rides in the tag: no load for the owner's)
        const unsigned long long word = ((unsigned long long)tag << 32) | (e + 1u);
        // (r6: sub-tables per stretch of consecutive ends, so that a sub-table stays in the Infinity Cache, were measured and LOSE --
        // lists recur across the whole block and get an owner per sub-table; profiles/EXPERIMENTS.md)
        uint32_t h = (uint32_t)((f2 * 0xD6E8FEB86659FD93ull) >> (64u - ltab_bits));
        for (uint32_t pr = 0; pr < LTAB_PROBES; pr++) {
            unsigned long long cur = ltab[h];
            if (cur == 0ull) {
                cur = atomicCAS(&ltab[h], 0ull, word);
                if (cur == 0ull) {  // claimed: k_owners_mult hands this end the slot's count
                    atomicAdd(&lmult[h], 1u);
                    m = 0u;
                    break;
                }
            }
            if ((uint32_t)(cur >> 32) == tag) {
                uint32_t other[LCAP];
                vs_load_list(lists + (uint64_t)((uint32_t)cur - 1u) * LC, lists_hi + (uint64_t)((uint32_t)cur - 1u) * 4u, n, other);
                if (vs_same_list(mine, other, n)) {
                    atomicAdd(&lmult[h], 1u);
                    m = 0u;
                    own = (uint32_t)cur - 1u;
                    break;
                }
            }
            h = (h + 1u) & ((1u << ltab_bits) - 1u);
        }
        // (no place within LTAB_PROBES slots -- a crowded table: m = 1, the end stands for itself)
    }
    mult[e] = m;
    gown[e] = own;
}

// every claimed slot of the block's list table: its owner gets the slot's multiplicity
__global__ void __launch_bounds__(256)
k_owners_mult(const unsigned long long *__restrict__ ltab, const uint32_t *__restrict__ lmult, uint64_t n_slots, uint32_t *__restrict__ mult) {
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_slots) return;
    const unsigned long long w = ltab[h];
    if (w != 0ull) mult[(uint32_t)w - 1u] = lmult[h];
}

// The owning ends (mult > 0) in END order -- locus order, so that a chunk of them touches few matrix rows, as a chunk of
// pairs does.  A workgroup takes a stretch of ends, counts the owners, reserves their places with ONE atomic (the counter is
// a single word: an atomic per wavefront would queue up behind each other), then reads the stretch again and writes.
#define COLLECT_TPB 256u
__global__ void __launch_bounds__(COLLECT_TPB)
k_owners_collect(const uint32_t *__restrict__ mult, uint64_t n_ends, uint64_t per_wg, uint32_t *__restrict__ owners, uint32_t *__restrict__ n_owners) {
    __shared__ uint32_t s_wave[COLLECT_TPB / 64u], s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * per_wg, hi = lo + per_wg < n_ends ? lo + per_wg : n_ends;
    uint32_t mine = 0;
    for (uint64_t e = lo + tid; e < hi; e += COLLECT_TPB) mine += mult[e] != 0u ? 1u : 0u;
#pragma unroll
    for (int d = 32; d; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane == 0) s_wave[wv] = mine;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (uint32_t i = 0; i < COLLECT_TPB / 64u; i++) tot += s_wave[i];
        s_base = tot ? atomicAdd(n_owners, tot) : 0u;
        s_wave[0] = 0u;  // (now the running offset inside the reservation)
    }
    __syncthreads();
    for (uint64_t e0 = lo; e0 < hi; e0 += COLLECT_TPB) {  // (uniform trip count: the ballot sees every lane)
        const uint64_t e = e0 + tid;
        const bool own = e < hi && mult[e] != 0u;
        const unsigned long long live = __ballot(own);
        uint32_t wbase = 0;
        if (lane == 0 && live) wbase = atomicAdd(&s_wave[0], (uint32_t)__popcll(live));  // (LDS)
        wbase = __shfl(wbase, 0, 64);
        if (own) owners[s_base + wbase + (uint32_t)__popcll(live & ((1ull << lane) - 1ull))] = (uint32_t)e;
    }
}

// The list of item i that is transposed, and its length (0: not an item).  Mode 0: pair i, its left list -- only if the
// right list holds anything; mode 1: the i-th owning read end.
template <int MODE>
__device__ __forceinline__ uint32_t vs_rows_item(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ owners, uint64_t i, uint64_t &row) {
    if (MODE == 0) {
        const uint2 c = *(const uint2 *)(counts + 2u * i);
        row = 2u * i;
        return c.y ? c.x : 0u;
    }
    row = owners[i];
    return counts[row];
}

// one lane per (item, quad of its list): the 16-bit bin of every listed node in [key_lo, key_lo + n_keys) + 1
template <int MODE>
__device__ __forceinline__ void vs_rows_histogram(uint32_t *h32, const uint32_t *__restrict__ lists, const uint32_t *__restrict__ lists_hi, const uint32_t *__restrict__ counts,
                                                  const uint32_t *__restrict__ owners, uint64_t lo, uint64_t hi, uint32_t key_lo, uint32_t n_keys) {
    for (uint64_t i = lo * 4u + threadIdx.x; i < hi * 4u; i += ROWS_TPB) {
        uint64_t row;
        const uint32_t q0 = (uint32_t)i & 3u, n = vs_rows_item<MODE>(counts, owners, i >> 2, row);
        for (uint32_t q = q0; 4u * q < n; q += 4u) {  // (a list of 17 .. LCAP nodes: its fifth quad is the first lane's too)
            const VsQuad v = *(const VsQuad *)vs_row_quad(lists, lists_hi, row, q);
            const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t x = e[j] - key_lo;
                if (4u * q + j < n && x < n_keys) atomicAdd(&h32[x >> 1], 1u << ((x & 1u) * 16u));
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(ROWS_TPB)
k_rows_count(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ lists_hi, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ owners, uint64_t n_items,
             const uint32_t *__restrict__ n_owners, uint32_t key_lo, uint32_t n_keys, uint32_t *__restrict__ row_count) {
    uint32_t *h32 = vs_lds;
    const uint32_t words = (n_keys + 1u) >> 1, tid = threadIdx.x;
    if (MODE) n_items = *n_owners;  // (mode 1: the launch covers every end, the items are the owners among them)
    const uint64_t per = MODE ? ROWS_CHUNK1 : ROWS_CHUNK;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n_items ? lo + per : n_items;
    if (lo >= n_items) return;
    for (uint32_t i = tid; i < words; i += ROWS_TPB) h32[i] = 0u;
    __syncthreads();
    vs_rows_histogram<MODE>(h32, lists, lists_hi, counts, owners, lo, hi, key_lo, n_keys);
    __syncthreads();
    for (uint32_t i = tid; i < words; i += ROWS_TPB) {
        const uint32_t v = h32[i];
        if (v & 0xFFFFu) atomicAdd(&row_count[key_lo + 2u * i], v & 0xFFFFu);
        if (v >> 16) atomicAdd(&row_count[key_lo + 2u * i + 1u], v >> 16);
    }
}

template <int MODE>
__global__ void __launch_bounds__(ROWS_TPB)
k_rows_fill(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ lists_hi, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ owners, uint64_t n_items,
            const uint32_t *__restrict__ n_owners, const uint32_t *__restrict__ gown, uint32_t key_lo, uint32_t n_keys,
            const uint32_t *__restrict__ row_ptr, uint32_t *__restrict__ row_cursor, uint32_t *__restrict__ entries) {
    uint32_t *h32 = vs_lds;
    const uint32_t words = (n_keys + 1u) >> 1, tid = threadIdx.x;
    if (MODE) n_items = *n_owners;
    if ((uint64_t)blockIdx.x * (MODE ? ROWS_CHUNK1 : ROWS_CHUNK) >= n_items) return;
    uint32_t *s_base = h32 + words;          // [ROWS_CAP] where this chunk's stretch of the row starts
    uint32_t *s_fill = s_base + ROWS_CAP;    // [ROWS_CAP / 2] 16-bit cursors inside the stretch
    uint32_t &s_nc = s_fill[ROWS_CAP / 2u];
    for (uint32_t i = tid; i < words; i += ROWS_TPB) h32[i] = 0u;
    for (uint32_t i = tid; i < ROWS_CAP / 2u; i += ROWS_TPB) s_fill[i] = 0u;
    if (tid == 0) s_nc = 0u;
    __syncthreads();
    const uint64_t per = MODE ? ROWS_CHUNK1 : ROWS_CHUNK;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n_items ? lo + per : n_items;
    vs_rows_histogram<MODE>(h32, lists, lists_hi, counts, owners, lo, hi, key_lo, n_keys);
    __syncthreads();
    // a bin that is not empty: reserve the chunk's stretch of that row, and turn the bin into the number of its cursor
    for (uint32_t i = tid; i < words; i += ROWS_TPB) {
        const uint32_t v = h32[i];
        if (!v) continue;
        uint32_t out = v;
#pragma unroll
        for (uint32_t half = 0; half < 2u; half++) {
            const uint32_t c = (v >> (16u * half)) & 0xFFFFu;
            if (!c) continue;
            const uint32_t x = key_lo + 2u * i + half;
            const uint32_t ci = atomicAdd(&s_nc, 1u);
            uint32_t code = 0xFFFFu;
            if (ci < ROWS_CAP) {
                s_base[ci] = row_ptr[x] + atomicAdd(&row_cursor[x], c);
                code = ci;
            }
            out = (out & ~(0xFFFFu << (16u * half))) | (code << (16u * half));
        }
        h32[i] = out;
    }
    __syncthreads();
    for (uint64_t i = lo * 4u + tid; i < hi * 4u; i += ROWS_TPB) {
        uint64_t row;
        const uint32_t q0 = (uint32_t)i & 3u, n = vs_rows_item<MODE>(counts, owners, i >> 2, row);
        if (4u * q0 >= n) continue;
        // what an entry says: whose list row k_rows_sum reads, and that list's length - 1 in the top five bits -- the block's
        // owner of the pair's RIGHT list (a few hundred distinct rows per matrix row, cached, instead of one row per pair) /
        // the owning end itself
        uint32_t payload;
        if (MODE) payload = (uint32_t)row | ((n - 1u) << 27);
        else {
            const uint64_t re = 2u * (i >> 2) + 1u;
            payload = gown[re] | ((counts[re] - 1u) << 27);
        }
        for (uint32_t q = q0; 4u * q < n; q += 4u) {
            const VsQuad v = *(const VsQuad *)vs_row_quad(lists, lists_hi, row, q);
            const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t x = e[j] - key_lo;
                if (4u * q + j >= n || x >= n_keys) continue;
                const uint32_t code = (h32[x >> 1] >> ((x & 1u) * 16u)) & 0xFFFFu;
                uint32_t pos;
                if (code != 0xFFFFu) {
                    const uint32_t sh = (code & 1u) * 16u;
                    const uint32_t old = atomicAdd(&s_fill[code >> 1], 1u << sh);
                    pos = s_base[code] + ((old >> sh) & 0xFFFFu);
                } else {
                    pos = row_ptr[e[j]] + atomicAdd(&row_cursor[e[j]], 1u);
                }
                entries[pos] = payload;
            }
        }
    }
}

// The strips' cell table is smaller than the pair-major kernel's (8 k slots, 64 KB) and the strips narrower for it: two
// workgroups share a CU, and one waits in its LDS queue while the other computes (configs[4]: 14.1 -> 11.2 ms)
#ifndef RS_BITS
#define RS_BITS 13
#endif
#ifndef RS_TPB
#define RS_TPB 1024
#endif
#define RS_SLOTS (1u << RS_BITS)
typedef CellTable<RS_BITS> RsTable;
#define RSUM_LDS_BYTES ((2u * RS_SLOTS + 64u + 8u) * 4u)
#define RSUM_MAX_ROWS 64u
template <int MODE>
__global__ void __launch_bounds__(RS_TPB)
k_rows_sum(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ lists_hi, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ mult, uint32_t N,
           const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ entries, uint32_t R, uint32_t n_strips, uint32_t fill_limit,
           uint32_t *__restrict__ mat, uint32_t off, uint8_t *__restrict__ tile_map, uint32_t T, uint32_t *__restrict__ queue,
           uint32_t *__restrict__ dbg) {
    uint32_t *s_key = vs_lds, *s_cnt = vs_lds + RS_SLOTS, *s_rowend = vs_lds + 2u * RS_SLOTS;  // [RSUM_MAX_ROWS]
    uint32_t &s_used = s_rowend[64], &s_lost = s_rowend[65], &s_strip = s_rowend[66];
    const uint32_t tid = threadIdx.x, q = tid & 3u;
    constexpr uint32_t STEP = RS_TPB / 4u;  // entries per pass of the workgroup
    for (uint32_t i = tid; i < RS_SLOTS; i += RS_TPB) { s_key[i] = RsTable::EMPTY; s_cnt[i] = 0u; }
    if (tid == 0) { s_used = 0u; s_lost = 0u; }
    for (;;) {
        __syncthreads();
        if (tid == 0) s_strip = atomicAdd(queue, 1u);
        __syncthreads();
        const uint32_t strip = s_strip;
        if (strip >= n_strips) break;
        const uint32_t first = strip * R, nrows = first + R <= N ? R : N - first;
        if (tid < nrows) s_rowend[tid] = row_ptr[first + tid + 1u];
        const uint32_t e0 = row_ptr[first], e1 = row_ptr[first + nrows];
        // keys are cell indices relative to the strip's first row, shifted so that key >> 4 is one 64-byte stretch of the matrix
        const uint32_t align = (uint32_t)(((uint64_t)first * N + off) & 15u);
        const uint64_t cell0 = (uint64_t)first * N;
        __syncthreads();
        auto write_out = [&]() {
            for (uint32_t i = tid; i < RS_SLOTS; i += RS_TPB) {
                const uint32_t key = s_key[i];
                if (key != RsTable::EMPTY) {
                    const uint32_t k2 = key - align;
                    atomicAdd(mat + cell0 + k2, s_cnt[i]);
                    if (tile_map) {
                        const uint32_t xl = k2 / N;
                        vs_mark_tile(tile_map, T, (uint32_t)MODE, first + xl, k2 - xl * N);
                    }
                    s_key[i] = RsTable::EMPTY;
                    s_cnt[i] = 0u;
                }
            }
            if (dbg && tid == 0) { atomicAdd(dbg + 0, s_lost); atomicAdd(dbg + 1, 1u); atomicAdd(dbg + 2, s_used); }
            __syncthreads();
            if (tid == 0) { s_used = 0u; s_lost = 0u; }
            __syncthreads();
        };
        // the item of entry i is loaded two passes ahead, its partner list one pass ahead (two dependent loads off the
        // critical path)
        auto load_item = [&](uint32_t base) -> uint32_t {
            const uint32_t i = base + (tid >> 2);
            return i < e1 ? entries[i] : 0xFFFFFFFFu;
        };
        auto load_list = [&](uint32_t it, uint32_t &n, uint32_t &wgt, VsQuad &yq) {
            n = 0u;
            wgt = 1u;
            yq = VsQuad{0u, 0u, 0u, 0u};
            if (it != 0xFFFFFFFFu) {
                const uint64_t row = it & 0x07FFFFFFu;  // (a read end of the transposition)
                n = (it >> 27) + 1u;
                if (MODE) wgt = mult[row];
                yq = *(const VsQuad *)(lists + row * LC + 4u * q);
            }
        };
        uint32_t p1 = e0 < e1 ? load_item(e0) : 0xFFFFFFFFu;
        uint32_t p2 = e1 - e0 > STEP ? load_item(e0 + STEP) : 0xFFFFFFFFu;
        uint32_t n1, w1;
        VsQuad y1;
        load_list(p1, n1, w1, y1);
        uint32_t pass = 0;
        for (uint32_t base = e0; base < e1; base += STEP, pass++) {
            const uint32_t n = n1, wgt = w1, p0 = p1;
            const VsQuad yq = y1;
            p1 = p2;
            load_list(p1, n1, w1, y1);
            p2 = e1 - base > 2u * STEP ? load_item(base + 2u * STEP) : 0xFFFFFFFFu;
            if (4u * q < n) {
                const uint32_t i = base + (tid >> 2);
                uint32_t xl = 0;  // rows of the strip that end at or before entry i
#pragma unroll
                for (uint32_t step = RSUM_MAX_ROWS / 2u; step; step >>= 1) {
                    const uint32_t t = xl + step;
                    if (t < nrows && s_rowend[t - 1u] <= i) xl = t;
                }
                const uint32_t kbase = xl * N + align, x = first + xl;
                auto add_quad = [&](const VsQuad &yv, uint32_t q4) {
                    const uint32_t ys[4] = {yv.x, yv.y, yv.z, yv.w};
                    uint32_t key[4], seen[4], at[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; j++) {
                        key[j] = kbase + ys[j];
                        at[j] = RsTable::slot(key[j]);
                        seen[j] = s_key[at[j]];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 4u; j++) {
                        if (q4 + j >= n || (MODE && ys[j] < x)) continue;  // (short_mat: the cell (x, y) belongs to the smaller node's row)
                        if (seen[j] == key[j]) {
                            atomicAdd(&s_cnt[at[j]], wgt);
                        } else if (!vs_cell_claim<RsTable>(s_key, s_cnt, &s_used, key[j], at[j], wgt)) {
                            atomicAdd(&s_lost, 1u);
                            vs_mark_tile(tile_map, T, (uint32_t)MODE, x, ys[j]);
                            atomicAdd(mat + cell0 + (key[j] - align), wgt);
                        }
                    }
                };
                add_quad(yq, 4u * q);
                // (a list of 17 .. LCAP nodes: its fifth quad is the first lane's too -- one entry in forty at configs[4])
                if (n > 16u && q == 0u) add_quad(*(const VsQuad *)(lists_hi + (uint64_t)(p0 & 0x07FFFFFFu) * 4u), 16u);
            }
            if ((pass & 3u) == 3u) {  // (a strip that outruns the limit between two looks finds the table crowded and adds the rest to memory itself: slower, the same sums)
                __syncthreads();
                const bool spill = s_used > fill_limit || s_lost > 4096u;
                __syncthreads();
                if (spill) write_out();
            }
        }
        __syncthreads();
        write_out();
    }
}

// ---- locus order -----------------------------------------------------------------------------------
// Pairs are handed to k_pe_tiles sorted by the first node their forward read's seeds hit, so that
// a tile holds pairs from one locus: they touch the same few node_mat / short_mat cells (summed in
// LDS before any global atomic) and the same node text (L1/L2 hits).  Any order gives the same
// counters (integer addition commutes); this one only changes how many global atomics it takes.
// key: node index, N = no seed of the forward read hits, N+1 = pair dropped by the N / length
// filters (still counted in the stats by k_pe_tiles).
__device__ __forceinline__ uint32_t vs_locus_key(const VsIndexDev &idx, const VsReadsDev &rd, uint64_t p) {
    const uint32_t mf = rd.meta[2 * p], mr = rd.meta[2 * p + 1];
    const uint32_t N = idx.n_nodes, w = idx.w, s = idx.s, K = idx.K;
    if ((((mf | mr) >> 24) & VS_FLAG_N) || (mf & VS_LEN_MASK) < K || (mr & VS_LEN_MASK) < K) return N + 1u;
    const uint32_t rlen = mf & VS_LEN_MASK;
    const uint64_t base = (uint64_t)rd.woff[2 * p] * 16u;
    const bool inv = (mf >> 24) & VS_FLAG_INVALID;
    for (uint32_t j = vs_seed_phase(rlen, w, s); j + w <= rlen; j += s) {
        if (inv && vs_seed_dirty(rd.mask, base + j, w)) continue;
        uint32_t pa, pb, sr;
        const uint64_t key = vs_seed_key(rd.words, base + j, w, &sr);
        const uint32_t c = vs_probe(idx, key, sr, &pa, &pb);
        if (c) return c == 1u ? pa : (idx.postings[pa].x & 0x01FFFFFFu);
    }
    return N;
}

// The first node a seed of read end e hits (N = none): the locus of that end.
__device__ __forceinline__ uint32_t vs_locus_key_end(const VsIndexDev &idx, const VsReadsDev &rd, uint64_t e) {
    const uint32_t m = rd.meta[e];
    const uint32_t N = idx.n_nodes, w = idx.w, s = idx.s, rlen = m & VS_LEN_MASK;
    const uint64_t base = (uint64_t)rd.woff[e] * 16u;
    const bool inv = (m >> 24) & VS_FLAG_INVALID;
    for (uint32_t j = vs_seed_phase(rlen, w, s); j + w <= rlen; j += s) {
        if (inv && vs_seed_dirty(rd.mask, base + j, w)) continue;
        uint32_t pa, pb, sr;
        const uint64_t key = vs_seed_key(rd.words, base + j, w, &sr);
        const uint32_t c = vs_probe(idx, key, sr, &pa, &pb);
        if (c) return c == 1u ? pa : (idx.postings[pa].x & 0x01FFFFFFu);
    }
    return N;
}

// Counting sort without global atomics (used while the N+2 keys fit an LDS histogram):
//   k_locus_count   : workgroup g takes pairs [g*chunk, (g+1)*chunk): keys -> keys[], LDS histogram
//                     -> column g of cnt[key][g]  (key-major so that the scan below is contiguous)
//   vs_scan_u32     : exclusive scan of cnt[] in (key, workgroup) order = first slot of every
//                     (key, workgroup) run in the sorted order -- stable, deterministic
//   k_locus_scatter : workgroup g loads its column as LDS cursors and places its pairs
#define LOCUS_LDS_KEYS 36864u  // 144 KB of LDS counters
#define LOCUS_LDS_MAX_PASSES 4u  // ... per pass; graphs of up to 147 k nodes are sorted through LDS histograms
#ifndef LOCUS_TPB
#define LOCUS_TPB 1024     // threads per workgroup of the two LDS-histogram sort kernels
#endif
#ifndef LOCUS_WGS
#define LOCUS_WGS 256u       // one per CU (measured: 1024 x 256 threads 0.70 ms, 256 x 1024 threads 0.57 ms)
#endif
// (key_lo, nk): the keys this pass counts / places -- a graph with more keys than one LDS histogram holds (54 k nodes at
// configs[4]) takes two or three passes over the stored keys instead of the global-atomic sort; `compute`: the first
// pass derives the keys (the expensive part: a read's seeds are probed) and stores them, the others read them back.
__global__ void __launch_bounds__(LOCUS_TPB)
k_locus_count(VsIndexDev idx, VsReadsDev rd, uint64_t n_pairs, uint32_t chunk, uint32_t n_wg, uint32_t *__restrict__ keys,
              uint32_t *__restrict__ cnt, uint32_t key_lo, uint32_t nk, uint32_t compute) {
    for (uint32_t i = threadIdx.x; i < nk; i += LOCUS_TPB) vs_lds[i] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    const uint64_t hi = lo + chunk < n_pairs ? lo + chunk : n_pairs;
    for (uint64_t p = lo + threadIdx.x; p < hi; p += LOCUS_TPB) {
        uint32_t key;
        if (compute) {
            key = vs_locus_key(idx, rd, p);
            keys[p] = key;
        } else {
            key = keys[p];
        }
        if (key - key_lo < nk) atomicAdd(&vs_lds[key - key_lo], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nk; i += LOCUS_TPB) cnt[(uint64_t)(key_lo + i) * n_wg + blockIdx.x] = vs_lds[i];
}

__global__ void __launch_bounds__(LOCUS_TPB)
k_locus_scatter(uint32_t key_lo, uint32_t nk, uint64_t n_pairs, uint32_t chunk, uint32_t n_wg, const uint32_t *__restrict__ keys,
                const uint32_t *__restrict__ first, uint32_t *__restrict__ perm) {
    for (uint32_t i = threadIdx.x; i < nk; i += LOCUS_TPB) vs_lds[i] = first[(uint64_t)(key_lo + i) * n_wg + blockIdx.x];
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    const uint64_t hi = lo + chunk < n_pairs ? lo + chunk : n_pairs;
    for (uint64_t p = lo + threadIdx.x; p < hi; p += LOCUS_TPB) {
        const uint32_t key = keys[p];
        if (key - key_lo < nk) perm[atomicAdd(&vs_lds[key - key_lo], 1u)] = (uint32_t)p;
    }
}

// Fallback for graphs with more nodes than the LDS histogram holds: global atomics.
__global__ void __launch_bounds__(TPB)
k_pe_locus(VsIndexDev idx, VsReadsDev rd, uint64_t n_pairs, uint32_t *__restrict__ keys, uint32_t *__restrict__ hist) {
    const uint64_t p = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t key = vs_locus_key(idx, rd, p);
    keys[p] = key;
    atomicAdd(&hist[key], 1u);
}

__global__ void __launch_bounds__(TPB)
k_pe_permute(uint64_t n_pairs, const uint32_t *__restrict__ keys, uint32_t *__restrict__ cursor, uint32_t *__restrict__ perm) {
    const uint64_t p = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (p >= n_pairs) return;
    perm[atomicAdd(&cursor[keys[p]], 1u)] = (uint32_t)p;
}

// ---- overflow path, first stop: one WAVEFRONT per listed pair ------------------------------------------------------------
// Pairs the tile kernels hand over -- an end with more accepted nodes than a list row holds (3 % of the pairs at configs[4]),
// a full tile table, an end with more bytes outside ACGT than inv4 holds -- are mapped here the general way (every probe,
// every posting, vs_extend with the validity mask), but with the per-end (node -> count, min offset, min window) state in an
// LDS hash table of the wavefront instead of the dense per-workgroup node state of k_pe_slow, four pairs per workgroup at a
// time and no workgroup barrier: 8 192 pairs in flight on the chip.  Its own limits (MID_SLOTS touched nodes, MID_LIST
// accepted nodes per end, 64 probes per end) send what is left to k_pe_slow, which has none.
#define MID_SLOTS 256u
#define MID_LIST 64u
#define MID_WORDS (4u * MID_SLOTS + 3u * 64u + 65u + 2u * MID_LIST + 4u)
__device__ __forceinline__ void vs_wave_sync() {  // LDS written by this wavefront is visible to all of its lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__global__ void __launch_bounds__(TPB)
k_pe_mid(PeParams P, const uint32_t *__restrict__ in_list, const uint32_t *__restrict__ in_count, uint32_t in_cap,
         uint32_t *__restrict__ out_list, uint32_t *__restrict__ out_count) {
    __shared__ uint32_t s_all[(TPB / 64u) * MID_WORDS];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t *s_key = s_all + wv * MID_WORDS, *s_cnt = s_key + MID_SLOTS, *s_minp = s_cnt + MID_SLOTS, *s_minj = s_minp + MID_SLOTS;
    uint32_t *s_pa = s_minj + MID_SLOTS, *s_pb = s_pa + 64u, *s_pc = s_pb + 64u, *s_pref = s_pc + 64u;  // s_pref[65]
    uint32_t *s_lst = s_pref + 65u;  // [2][MID_LIST] accepted nodes of the two ends
    uint32_t *s_flag = s_lst + 2u * MID_LIST;  // [0] overflow, [1..2] list lengths
    const uint32_t N = P.idx.n_nodes, w = P.idx.w, s = P.idx.s, K = P.idx.K, wv_seed = VS_SEED_VERIFIED(P.idx.w);
    uint32_t n_in = *in_count;
    if (n_in > in_cap) n_in = in_cap;
    const uint32_t n_waves = gridDim.x * (TPB / 64u);
    for (uint32_t li = blockIdx.x * (TPB / 64u) + wv; li < n_in; li += n_waves) {
        const uint32_t pair = in_list[li];
        if (lane < 4u) s_flag[lane] = 0u;
        vs_wave_sync();
        for (uint32_t side = 0; side < 2u; side++) {
            const uint64_t e = 2ull * pair + side;
            const uint32_t meta = P.rd.meta[e];
            const uint32_t rlen = meta & VS_LEN_MASK;
            const uint64_t rbase = (uint64_t)P.rd.woff[e] * 16u;
            const uint32_t *mk = ((meta >> 24) & VS_FLAG_INVALID) ? P.rd.mask : nullptr;
            const uint32_t nprobe = vs_seed_probes(rlen, w, s, P.phase0), phase = vs_seed_phase(rlen, w, s, P.phase0);
            for (uint32_t i = lane; i < MID_SLOTS; i += 64u) { s_key[i] = EMPTY_NODE; s_cnt[i] = 0u; s_minp[i] = 0xFFFFFFFFu; s_minj[i] = 0xFFFFFFFFu; }
            if (nprobe > 64u) { if (lane == 0u) s_flag[0] = 1u; }
            // probes: one per lane
            uint32_t c = 0u, pa = 0u, pb = 0u;
            if (lane < nprobe && nprobe <= 64u) {
                const uint32_t j = phase + lane * s;
                if (!(mk && vs_seed_dirty(mk, rbase + j, w))) {
                    uint32_t sr;
                    const uint64_t key = vs_seed_key(P.rd.words, rbase + j, w, &sr);
                    c = vs_probe(P.idx, key, sr, &pa, &pb);
                }
            }
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t2 = __shfl_up(incl, d, 64);
                if (lane >= (uint32_t)d) incl += t2;
            }
            s_pa[lane] = pa; s_pb[lane] = pb; s_pc[lane] = c;
            s_pref[lane + 1u] = incl;
            if (lane == 0u) s_pref[0] = 0u;
            vs_wave_sync();
            const uint32_t total = s_pref[64];
            for (uint32_t t = lane; t < total; t += 64u) {
                const uint32_t pr = vs_upper_idx(s_pref, 65u, t);
                const uint32_t k2 = t - s_pref[pr], j = phase + pr * s;
                uint32_t node, pos, opp;
                VsNodeMeta nm;
                if (s_pc[pr] == 1u) {
                    node = s_pa[pr]; pos = s_pb[pr] & 0x7FFFFFFFu; opp = s_pb[pr] >> 31;
                    nm = P.idx.meta[node];
                } else {
                    const VsPosting po = vs_posting_unpack(P.idx.postings[s_pa[pr] + k2]);
                    node = po.node; pos = po.pos; opp = po.strand ^ (s_pb[pr] >> 31);
                    nm.woff = po.woff; nm.len = po.len;
                }
                const uint32_t q = opp ? nm.len - pos - w : pos;
                uint32_t a, qa, len;
                if (P.mid_fast && !mk) {
                    // (r5) an end without bytes outside ACGT in a block of the straight-line shape (stride <= 32, reads <= w + 160):
                    // the comparison of k_pe_tiles<1> -- one left window, five right ones, no data-dependent loop -- off the read's
                    // words in global memory (the 64 lanes of the wavefront share them)
                    uint32_t cl = s < j ? s : j;
                    cl = cl < q ? cl : q;
                    uint32_t rem = rlen - j - wv_seed;
                    const uint32_t dr = nm.len - q - wv_seed;
                    rem = rem < dr ? rem : dr;
                    const uint32_t tb = (nm.woff + (opp ? P.idx.rc_delta : 0u)) * 16u;
                    uint32_t left, ext;
                    vs_agree_fast<false>(P.rd.words + (rbase >> 4), 0u, P.idx.fwd_words, tb + q, cl, tb + q + wv_seed, rem, j, wv_seed, &left, &ext);
                    len = left + wv_seed + ext;
                    if (left >= s || len < K) continue;
                    a = j - left;
                    qa = q - left;
                } else {
                    const uint32_t *tw = opp ? P.idx.rc_words : P.idx.fwd_words;
                    if (!vs_extend(P.rd.words, rbase, rlen, tw, nm.woff * 16u, nm.len, j, q, wv_seed, s, K, mk, rbase, &a, &qa, &len)) continue;
                }
                uint32_t at = (node * 0x9E3779B1u) >> 24;  // MID_SLOTS = 256
                bool placed = false;
                for (uint32_t tr = 0; tr < MID_SLOTS; tr++) {
                    const uint32_t old = atomicCAS(&s_key[at], EMPTY_NODE, node);
                    if (old == EMPTY_NODE || old == node) {
                        atomicAdd(&s_cnt[at], len - K + 1u);
                        atomicMin(&s_minp[at], opp ? nm.len - qa - len : qa);
                        // (mid_fast: reads of at most 191 bases and nodes below 2^23 bases -- the node's length rides above the
                        // read offset, equal for every update of the slot, and the acceptance sweep needs no header load)
                        atomicMin(&s_minj[at], P.mid_fast ? (nm.len << 8) | a : a);
                        placed = true;
                        break;
                    }
                    at = (at + 1u) & (MID_SLOTS - 1u);
                }
                if (!placed) s_flag[0] = 1u;
            }
            vs_wave_sync();
            // acceptance test per occupied slot; accepted nodes to the end's list (ballot + prefix count)
            uint32_t base = 0u;
            for (uint32_t i0 = 0; i0 < MID_SLOTS; i0 += 64u) {
                const uint32_t i = i0 + lane;
                const uint32_t node = s_key[i];
                bool acc = false;
                if (node != EMPTY_NODE)
                    acc = P.mid_fast ? vs_accept32(s_cnt[i], s_minp[i], s_minj[i] & 0xFFu, s_minj[i] >> 8, rlen, K)
                                     : vs_accept(s_cnt[i], s_minp[i], s_minj[i], P.idx.meta[node].len, rlen, K);
                const unsigned long long mask = __ballot(acc);
                const uint32_t at = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                if (acc) {
                    if (at < MID_LIST) s_lst[side * MID_LIST + at] = node; else s_flag[0] = 1u;
                }
                base += (uint32_t)__popcll(mask);
            }
            if (lane == 0u) s_flag[1u + side] = base;
            vs_wave_sync();
        }
        if (s_flag[0]) {  // beyond this kernel's tables: the general kernel
            if (lane == 0u) out_list[atomicAdd(out_count, 1u)] = pair;
            vs_wave_sync();
            continue;
        }
        const uint32_t nl = s_flag[1], nr = s_flag[2];
        const uint32_t *L = s_lst, *R = s_lst + MID_LIST;
        if (P.accumulate) {  // PE_Inference.py:174-188
            // Both lists ascending first (one entry per lane, bitonic network over the wavefront): neighbouring lanes
            // below then add to neighbouring cells of one matrix row, and the atomics of a wave instruction leave the L2
            // as one memory-side request per 64-byte stretch they touch (r3: 7.5 -> 5.8 ms with the path numbering).
            for (uint32_t side = 0; side < 2u; side++) {
                uint32_t *sv = s_lst + side * MID_LIST;
                const uint32_t n = side ? nr : nl;
                uint32_t v = lane < n ? sv[lane] : 0xFFFFFFFFu;
#pragma unroll
                for (uint32_t k2 = 2u; k2 <= 64u; k2 <<= 1)
#pragma unroll
                    for (uint32_t j2 = k2 >> 1; j2 > 0u; j2 >>= 1) {
                        const uint32_t o = (uint32_t)__shfl_xor((int)v, (int)j2, 64);
                        const bool keep_min = ((lane & j2) == 0u) == ((lane & k2) == 0u);
                        v = keep_min ? (v < o ? v : o) : (v > o ? v : o);
                    }
                if (lane < n) sv[lane] = v;
            }
            vs_wave_sync();
            for (uint32_t i = lane; i < nl * nr; i += 64u) {
                vs_mark_tile_store(P.tile_map, P.tile_T, 0u, L[i / nr], R[i % nr]);
                atomicAdd(&P.node_mat[(uint64_t)L[i / nr] * N + R[i % nr]], 1u);
            }
            for (uint32_t side = 0; side < 2u; side++) {
                const uint32_t *sv = side ? R : L;
                const uint32_t n = side ? nr : nl;
                for (uint32_t i = lane; i < n * n; i += 64u) {
                    const uint32_t a = i / n, b = i % n, x = sv[a], y = sv[b];
                    if (x < y || a == b) {
                        vs_mark_tile_store(P.tile_map, P.tile_T, 1u, x, y);
                        atomicAdd(&P.short_mat[(uint64_t)x * N + y], 1u);
                    }
                }
            }
        }
        if (P.dbg_counts) {
            for (uint32_t side = 0; side < 2u; side++) {
                const uint32_t *sv = side ? R : L;
                const uint32_t n = side ? nr : nl;
                const uint64_t e = 2ull * pair + side;
                if (lane == 0u) P.dbg_counts[e] = n;
                for (uint32_t i = lane; i < n && i < P.dbg_cap; i += 64u) P.dbg_lists[e * P.dbg_cap + i] = sv[i];
            }
        }
        vs_wave_sync();
    }
}

// ---- overflow path: any number of nodes per end, any read ------------------------------------------
// One workgroup per listed pair (ends with more accepted nodes than an LDS row holds, ends with many
// bytes outside ACGT, seeds with a huge posting list).  Node state is dense in HBM per workgroup --
// the reference's own layout (PE_Inference.py:19-21) -- but only the nodes an end touches are ever
// looked at: the first credit of a node appends it to a list, and the acceptance sweep walks that
// list and restores the state.  Probes go through the workgroup 256 at a time; their postings are
// spread over the threads (prefix sum of the counts in LDS, one posting per thread and round).
// dense layout per workgroup: cnt[N] minp[N] minj[N] touched[N] surv0[N] surv1[N].
#define SLOW_WORDS_PER_NODE 6u
__global__ void __launch_bounds__(TPB)
k_pe_slow(PeParams P, uint32_t *dense, uint32_t n_slow_cap, const uint32_t *__restrict__ list, const uint32_t *__restrict__ count) {
    __shared__ uint32_t s_n[2], s_nt, s_cnt[TPB + 1], s_pa[TPB], s_pb[TPB];
    const uint32_t tid = threadIdx.x;
    const uint32_t N = P.idx.n_nodes, w = P.idx.w, s = P.idx.s, K = P.idx.K;
    uint32_t n_slow = *count;
    if (n_slow > n_slow_cap) n_slow = n_slow_cap;
    uint32_t *cnt = dense + (uint64_t)blockIdx.x * SLOW_WORDS_PER_NODE * N;
    uint32_t *minp = cnt + N, *minj = minp + N, *touched = minj + N, *surv0 = touched + N, *surv1 = surv0 + N;
    for (uint32_t li = blockIdx.x; li < n_slow; li += gridDim.x) {
        const uint32_t pair = list[li];
        if (tid < 2) s_n[tid] = 0;
        for (uint32_t side = 0; side < 2; side++) {
            const uint64_t e = 2ull * pair + side;
            const uint32_t meta = P.rd.meta[e];
            const uint32_t rlen = meta & VS_LEN_MASK;
            const uint64_t rbase = (uint64_t)P.rd.woff[e] * 16u;
            const uint32_t *mk = ((meta >> 24) & VS_FLAG_INVALID) ? P.rd.mask : nullptr;
            uint32_t *surv = side ? surv1 : surv0;
            const uint32_t nprobe = vs_seed_probes(rlen, w, s, P.phase0), phase = vs_seed_phase(rlen, w, s, P.phase0);
            if (tid == 0) s_nt = 0;
            __syncthreads();
            for (uint32_t p0 = 0; p0 < nprobe; p0 += TPB) {
                // probes p0 .. p0 + 255: one per thread
                uint32_t c = 0, pa = 0, pb = 0;
                const uint32_t pi = p0 + tid;
                if (pi < nprobe) {
                    const uint32_t j = phase + pi * s;
                    if (!(mk && vs_seed_dirty(mk, rbase + j, w))) {
                        uint32_t sr;
                        const uint64_t key = vs_seed_key(P.rd.words, rbase + j, w, &sr);
                        c = vs_probe(P.idx, key, sr, &pa, &pb);
                    }
                }
                s_pa[tid] = pa;
                s_pb[tid] = pb;
                // inclusive scan of the posting counts over the workgroup -> s_cnt[1 .. 256], s_cnt[0] = 0
                uint32_t incl = c;
                const uint32_t lane = tid & 63u;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t2 = __shfl_up(incl, d, 64);
                    if (lane >= (uint32_t)d) incl += t2;
                }
                s_cnt[tid + 1u] = incl;
                if (tid == 0) s_cnt[0] = 0;
                __syncthreads();
                uint32_t off = 0;
                for (uint32_t wv = 0; wv < (tid >> 6); wv++) off += s_cnt[wv * 64u + 64u];
                __syncthreads();
                s_cnt[tid + 1u] = incl + off;
                __syncthreads();
                const uint32_t total = s_cnt[TPB];
                for (uint32_t t = tid; t < total; t += TPB) {
                    // the probe that owns posting t: last index with s_cnt[idx] <= t
                    const uint32_t pr = vs_upper_idx(s_cnt, TPB + 1u, t);
                    const uint32_t k2 = t - s_cnt[pr], j = phase + (p0 + pr) * s;
                    const uint32_t qa0 = s_pa[pr], qb0 = s_pb[pr];
                    uint32_t node, pos, opp;
                    VsNodeMeta nm;
                    if (s_cnt[pr + 1u] - s_cnt[pr] == 1u) {
                        node = qa0; pos = qb0 & 0x7FFFFFFFu; opp = qb0 >> 31;
                        nm = P.idx.meta[node];
                    } else {
                        const VsPosting po = vs_posting_unpack(P.idx.postings[qa0 + k2]);
                        node = po.node; pos = po.pos; opp = po.strand ^ (qb0 >> 31);
                        nm.woff = po.woff; nm.len = po.len;
                    }
                    const uint32_t *tw = opp ? P.idx.rc_words : P.idx.fwd_words;
                    const uint32_t q = opp ? nm.len - pos - w : pos;
                    uint32_t a, qa, len;
                    if (!vs_extend(P.rd.words, rbase, rlen, tw, nm.woff * 16u, nm.len, j, q, VS_SEED_VERIFIED(w), s, K, mk, rbase, &a, &qa, &len))
                        continue;
                    if (atomicAdd(&cnt[node], len - K + 1u) == 0u) touched[atomicAdd(&s_nt, 1u)] = node;
                    atomicMin(&minp[node], opp ? nm.len - qa - len : qa);
                    atomicMin(&minj[node], a);
                }
                __syncthreads();  // s_cnt / s_pa / s_pb are rewritten by the next batch of probes
            }
            // (all of it is this workgroup's own traffic, the loads below bypass L1: a workgroup barrier
            // orders it; a device-wide fence would write back the whole L2 of the XCD each time)
            __syncthreads();
            const uint32_t nt = s_nt;
            for (uint32_t i = tid; i < nt; i += TPB) {
                const uint32_t nd = __hip_atomic_load(&touched[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t v = __hip_atomic_load(&cnt[nd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t c = __hip_atomic_load(&minp[nd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t ki = __hip_atomic_load(&minj[nd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (vs_accept(v, c, ki, P.idx.meta[nd].len, rlen, K)) surv[atomicAdd(&s_n[side], 1u)] = nd;
                __hip_atomic_store(&cnt[nd], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&minp[nd], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&minj[nd], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        const uint32_t nl = s_n[0], nr = s_n[1];
        if (P.accumulate) {
            for (uint64_t i = tid; i < (uint64_t)nl * nr; i += TPB) {
                uint32_t a = (uint32_t)(i / nr), b = (uint32_t)(i - (uint64_t)a * nr);
                const uint32_t x = __hip_atomic_load(&surv0[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t y = __hip_atomic_load(&surv1[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vs_mark_tile(P.tile_map, P.tile_T, 0u, x, y);
                atomicAdd(&P.node_mat[(uint64_t)x * N + y], 1u);
            }
            for (uint32_t side = 0; side < 2; side++) {
                const uint32_t *sv = side ? surv1 : surv0;
                const uint32_t n = side ? nr : nl;
                for (uint64_t i = tid; i < (uint64_t)n * n; i += TPB) {
                    uint32_t a = (uint32_t)(i / n), b = (uint32_t)(i - (uint64_t)a * n);
                    uint32_t x = __hip_atomic_load(&sv[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    uint32_t y = __hip_atomic_load(&sv[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (x < y || a == b) {
                        vs_mark_tile(P.tile_map, P.tile_T, 1u, x, y);
                        atomicAdd(&P.short_mat[(uint64_t)x * N + y], 1u);
                    }
                }
            }
        }
        if (P.dbg_counts) {
            for (uint32_t side = 0; side < 2; side++) {
                const uint32_t *sv = side ? surv1 : surv0;
                const uint32_t n = side ? nr : nl;
                const uint64_t e = 2ull * pair + side;
                if (tid == 0) P.dbg_counts[e] = n;
                for (uint32_t i = tid; i < n && i < P.dbg_cap; i += TPB)
                    P.dbg_lists[e * P.dbg_cap + i] = __hip_atomic_load(&sv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(TPB) k_dense_zero_cnt(uint32_t *dense, uint64_t N, uint64_t groups) {
    uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < N * groups) dense[(i / N) * (uint64_t)SLOW_WORDS_PER_NODE * N + (i % N)] = 0u;
}

// ---- host side ---------------------------------------------------------------------------------
#define LDS_BUDGET_BYTES (64u * 1024u)
#define NI_CAP 4096u
// workgroups of the overflow kernel: as many as 4 GB of dense per-workgroup node state allow, 64 .. 2048
static uint32_t slow_grid_for(uint32_t n_nodes) {
    const uint64_t per_wg = sizeof(uint32_t) * (uint64_t)SLOW_WORDS_PER_NODE * (n_nodes ? n_nodes : 1u);
    uint64_t g = (4ull << 30) / per_wg;
    return (uint32_t)(g < 64u ? 64u : g > 2048u ? 2048u : g);
}

static uint32_t pool_for(uint32_t ept, uint32_t *bits) {
    uint32_t b = 6;
    while ((1u << b) < 16u * ept) b++;
    *bits = b;
    return 1u << b;
}

static size_t lds_bytes(uint32_t ept, uint32_t pmax, uint32_t words_cap) {
    uint32_t bits;
    uint32_t pool = pool_for(ept, &bits);
    return (size_t)tile_layout(ept, pmax, words_cap, pool).total * sizeof(uint32_t);
}

// The row-owner path: both counters of one block from the per-end lists (see "both matrices by ROW OWNERS").
static int pe_count_by_rows(vs_ctx *ctx, uint64_t slots_pairs, uint32_t *d_node_mat, uint32_t *d_short_mat, uint8_t *d_tile_map, uint32_t T) {
    hipStream_t st = ctx->stream;
    const VsTuning &tn = ctx->tune;
    const uint32_t N = ctx->idx.n_nodes;
    const uint64_t sub_max = tn.rows_sub ? (uint64_t)tn.rows_sub : (uint64_t)ROWS_SUB;
    const uint64_t sub_pairs = slots_pairs < sub_max ? slots_pairs : sub_max;  // pairs per transposition
    if (ctx->rows_cap < (uint64_t)N + 2u) {
        if (ctx->d_rows) VS_HIP(ctx, hipFree(ctx->d_rows));
        ctx->d_rows = nullptr;
        ctx->rows_cap = 0;
        // per mode: counts, cursors, offsets; then the block sums of the scan (2 048 values per block, 64 bits each)
        VS_HIP(ctx, hipMalloc(&ctx->d_rows, sizeof(uint32_t) * 6u * ((uint64_t)N + 2u) + sizeof(uint64_t) * ((uint64_t)N / 2048u + 8u)));
        ctx->rows_cap = (uint64_t)N + 2u;
    }
    // the list table of a transposition: a power of two of slots, at least one per read end (VS_LTAB_BITS: tests crowd it)
    uint32_t ltab_bits = 10;
    while ((1ull << ltab_bits) < 2u * sub_pairs && ltab_bits < 31u) ltab_bits++;
    if (tn.ltab_bits >= 0) ltab_bits = (uint32_t)tn.ltab_bits;
    const bool use_ltab = ltab_bits > 0;
    const uint64_t ltab_slots = use_ltab ? 1ull << ltab_bits : 0;
    // entries: one word per listed node -- the left lists (node_mat) and the lists of the owning ends (short_mat); the
    // multiplicity of every end, the owning ends, the table (a 64-bit word and a 32-bit sum per slot)
    if (ctx->row_entries_cap < sub_pairs || ctx->ltab_cap < ltab_slots) {
        for (void **q : {&ctx->d_row_entries, &ctx->d_mult, &ctx->d_ltab})
            if (*q) { VS_HIP(ctx, hipFree(*q)); *q = nullptr; }
        ctx->row_entries_cap = ctx->ltab_cap = 0;
        VS_HIP(ctx, hipMalloc(&ctx->d_row_entries, sizeof(uint32_t) * (3u * sub_pairs * LCAP + 16u)));
        VS_HIP(ctx, hipMalloc(&ctx->d_mult, sizeof(uint32_t) * (6u * sub_pairs + 4u)));  // mult[2 np], owners[2 np], gown[2 np]
        VS_HIP(ctx, hipMalloc(&ctx->d_ltab, (sizeof(uint64_t) + sizeof(uint32_t)) * ltab_slots + 16u));
        ctx->row_entries_cap = sub_pairs;
        ctx->ltab_cap = ltab_slots;
    }
    const uint64_t cap = ctx->rows_cap;
    uint32_t *rows = (uint32_t *)ctx->d_rows;
    uint64_t *scan_tmp = (uint64_t *)(rows + ((6u * cap + 1u) & ~1ull));
    const uint32_t keys_max = tn.rows_keys ? tn.rows_keys : ROWS_KEYS;
    const uint32_t n_keys = N < keys_max ? N : keys_max;
    const size_t rl = rows_lds_bytes(n_keys);
    for (const void *fn : {(const void *)k_rows_count<0>, (const void *)k_rows_count<1>, (const void *)k_rows_fill<0>, (const void *)k_rows_fill<1>})
        VS_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
    for (const void *fn : {(const void *)k_rows_sum<0>, (const void *)k_rows_sum<1>})
        VS_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RSUM_LDS_BYTES));
    // the table is written out once this share of its slots is taken (a strip that holds more cells than that is written
    // in pieces: still one atomic per cell and piece); VS_ACC_FILL: percent
    uint32_t fill = RS_SLOTS / 2u;
    if (tn.acc_fill_pct >= 0) fill = (uint32_t)((uint64_t)RS_SLOTS * (uint32_t)tn.acc_fill_pct / 100u);
    if (fill > RS_SLOTS - 1024u) fill = RS_SLOTS - 1024u;
    // rows per strip: a strip's distinct cells should fill the table less than half.  configs[4]: 4 rows of node_mat hold
    // 1.7 k cells at the median and 6 k at most, 32 rows of short_mat 2.1 k and 6.6 k.  VS_ROWS_PER_STRIP overrides both.
    const uint32_t R[2] = {tn.rows_per_strip ? tn.rows_per_strip : 2u, tn.rows_per_strip1 ? tn.rows_per_strip1 : tn.rows_per_strip ? tn.rows_per_strip : 2u};
    uint32_t *dbg = tn.debug_acc ? (uint32_t *)ctx->d_slow_count + 10 : nullptr;
    uint32_t *queue = (uint32_t *)ctx->d_slow_count + 9, *n_owners = (uint32_t *)ctx->d_slow_count + 15;
    unsigned long long *ltab = use_ltab ? (unsigned long long *)ctx->d_ltab : nullptr;
    uint32_t *lmult = use_ltab ? (uint32_t *)((unsigned long long *)ctx->d_ltab + ltab_slots) : nullptr;
    for (uint64_t p0 = 0; p0 < slots_pairs; p0 += sub_pairs) {
        const uint64_t np = slots_pairs - p0 < sub_pairs ? slots_pairs - p0 : sub_pairs;
        const uint32_t *sl = (const uint32_t *)ctx->d_lists + 2u * p0 * LC, *sh = (const uint32_t *)ctx->d_lists + 2u * slots_pairs * LC + 2u * p0 * 4u;
        const uint32_t *sc = (const uint32_t *)ctx->d_list_counts + 2u * p0;
        uint32_t *mult = (uint32_t *)ctx->d_mult, *owners = mult + 2u * sub_pairs, *gown = owners + 2u * sub_pairs;
        const unsigned n_chunks = (unsigned)((np + ROWS_CHUNK - 1u) / ROWS_CHUNK);
        const unsigned n_chunks1 = (unsigned)((2u * np + ROWS_CHUNK1 - 1u) / ROWS_CHUNK1);  // (every end could be an owner; a chunk past the last owner returns at once)
        VS_HIP(ctx, hipMemsetAsync(rows, 0, sizeof(uint32_t) * 6u * cap, st));
        VS_HIP(ctx, hipMemsetAsync(n_owners, 0, sizeof(uint32_t), st));
        if (use_ltab) VS_HIP(ctx, hipMemsetAsync(ctx->d_ltab, 0, (sizeof(uint64_t) + sizeof(uint32_t)) * ltab_slots, st));
        hipLaunchKernelGGL(k_list_owners, dim3((unsigned)((2u * np + 255u) / 256u)), dim3(256), 0, st, sl, sh, sc, 2u * np, mult, gown, ltab, lmult, ltab_bits);
        if (use_ltab)
            hipLaunchKernelGGL(k_owners_mult, dim3((unsigned)((ltab_slots + 255u) / 256u)), dim3(256), 0, st, (const unsigned long long *)ltab,
                               (const uint32_t *)lmult, ltab_slots, mult);
        {
            const uint64_t per_wg = ((2u * np + ctx->n_cu * 8ull - 1u) / (ctx->n_cu * 8ull) + COLLECT_TPB - 1u) / COLLECT_TPB * COLLECT_TPB;
            hipLaunchKernelGGL(k_owners_collect, dim3((unsigned)((2u * np + per_wg - 1u) / per_wg)), dim3(COLLECT_TPB), 0, st, (const uint32_t *)mult, 2u * np, per_wg,
                               owners, n_owners);
        }
        for (int mode = 0; mode < 2; mode++) {
            uint32_t *row_count = rows + 3u * mode * cap, *row_cursor = row_count + cap, *row_ptr = row_cursor + cap;
            uint32_t *entries = (uint32_t *)ctx->d_row_entries + (mode ? np * LCAP : 0u);
            for (uint32_t key_lo = 0; key_lo < N; key_lo += n_keys) {
                const uint32_t nk = N - key_lo < n_keys ? N - key_lo : n_keys;
                if (mode) hipLaunchKernelGGL(k_rows_count<1>, dim3(n_chunks1), dim3(ROWS_TPB), rl, st, sl, sh, sc, (const uint32_t *)owners, 0ull, (const uint32_t *)n_owners, key_lo, nk, row_count);
                else hipLaunchKernelGGL(k_rows_count<0>, dim3(n_chunks), dim3(ROWS_TPB), rl, st, sl, sh, sc, (const uint32_t *)nullptr, np, (const uint32_t *)nullptr, key_lo, nk, row_count);
            }
            int rc = vs_scan_u32(ctx, row_count, row_ptr, (uint64_t)N + 1u, scan_tmp, nullptr);
            if (rc) return rc;
            for (uint32_t key_lo = 0; key_lo < N; key_lo += n_keys) {
                const uint32_t nk = N - key_lo < n_keys ? N - key_lo : n_keys;
                if (mode) hipLaunchKernelGGL(k_rows_fill<1>, dim3(n_chunks1), dim3(ROWS_TPB), rl, st, sl, sh, sc, (const uint32_t *)owners, 0ull, (const uint32_t *)n_owners, (const uint32_t *)gown, key_lo, nk, (const uint32_t *)row_ptr, row_cursor, entries);
                else hipLaunchKernelGGL(k_rows_fill<0>, dim3(n_chunks), dim3(ROWS_TPB), rl, st, sl, sh, sc, (const uint32_t *)nullptr, np, (const uint32_t *)nullptr, (const uint32_t *)gown, key_lo, nk, (const uint32_t *)row_ptr, row_cursor, entries);
            }
        }
        for (int mode = 0; mode < 2; mode++) {
            const uint32_t *row_ptr = rows + 3u * mode * cap + 2u * cap;
            const uint32_t *entries = (const uint32_t *)ctx->d_row_entries + (mode ? np * LCAP : 0u);
            const uint32_t n_strips = (N + R[mode] - 1u) / R[mode];
            uint32_t grid = (uint32_t)ctx->n_cu * (RSUM_LDS_BYTES <= 40000u ? 4u : 2u) * (1024u / RS_TPB);
            if (grid > n_strips) grid = n_strips;
            uint32_t *m = mode ? d_short_mat : d_node_mat;
            const uint32_t off = (uint32_t)(((uintptr_t)m >> 2) & 15u);
            VS_HIP(ctx, hipMemsetAsync(queue, 0, sizeof(uint32_t), st));
            if (mode)
                hipLaunchKernelGGL(k_rows_sum<1>, dim3(grid), dim3(RS_TPB), RSUM_LDS_BYTES, st, sl, sh, sc, (const uint32_t *)mult, N, row_ptr, entries, R[1], n_strips, fill, m,
                                   off, d_tile_map, T, queue, dbg);
            else
                hipLaunchKernelGGL(k_rows_sum<0>, dim3(grid), dim3(RS_TPB), RSUM_LDS_BYTES, st, sl, sh, sc, (const uint32_t *)mult, N, row_ptr, entries, R[0], n_strips, fill, m,
                                   off, d_tile_map, T, queue, dbg);
        }
    }
    VS_HIP(ctx, hipGetLastError());
    return VS_OK;
}

static int pe_launch(vs_ctx *ctx, const vs_reads *reads, uint32_t *d_node_mat, uint32_t *d_short_mat, uint64_t *d_stats,
                     uint32_t *d_dbg_lists, uint32_t *d_dbg_counts, uint32_t dbg_cap, uint8_t *d_tile_map = nullptr) {
    if (!ctx->has_index) return vs_fail(ctx, VS_E_STATE, "vs_pe_count: build an index first (vs_index_build)");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const VsIndexDev &idx = ctx->idx;
    const uint64_t n_ends = reads->n_ends, n_pairs = n_ends / 2;
    ctx->last_ms[0] = ctx->last_ms[1] = ctx->last_ms[2] = 0;
    ctx->last_launched = 0;
    if (n_ends == 0) return VS_OK;
    const uint32_t maxlen = (uint32_t)reads->max_len;
    const uint32_t wpe = maxlen ? (maxlen + 15u) / 16u : 1u;
    if (idx.n_nodes > 0x01FFFFFEu) return vs_fail(ctx, VS_E_RANGE, "more than 2^25-2 nodes");
    // experiment switches: fixed defaults unless the process runs with VS_EXPERIMENT (see VsTuning)
    if (ctx->experiment_level) vs_tuning_load(ctx->tune, ctx->experiment_level);
    const VsTuning &tn = ctx->tune;
    // probes of the longest end (vs_seed_probes grows with the length): the probe slots a tile reserves per end
    uint32_t pmax = vs_seed_probes(maxlen, idx.w, idx.s, tn.phase0);
    if (!pmax) pmax = 1u;
    uint32_t ept = tn.ept ? tn.ept : STD_EPT;
    if (ept < 2 || ept > TTPB / 2u) ept = STD_EPT;
    while (ept > 2 && (ept * pmax > NI_CAP || lds_bytes(ept, pmax, ept * wpe) > LDS_BUDGET_BYTES)) ept -= 2;
    // (LDS is handed out in 1280-byte pieces: 32 000 B per workgroup lets five share a CU, 40 000 four.  A tile of
    // at least 32 ends that fits one of these is taken over a larger one that wastes the rest.)
    if (!tn.ept) {
        for (size_t fit : {(size_t)(TTPB >= 256 ? 32000 : TTPB == 128 ? 15360 : 7680), (size_t)(TTPB >= 256 ? 40000 : TTPB == 128 ? 17920 : 8960)}) {
            uint32_t e2 = ept;
            while (e2 > STD_EPT / 2u && lds_bytes(e2, pmax, e2 * wpe) > fit) e2 -= 2;
            if (lds_bytes(e2, pmax, e2 * wpe) <= fit) { ept = e2; break; }
        }
    }
    size_t lds = lds_bytes(ept, pmax, ept * wpe);
    if (lds > 160u * 1024u)
        return vs_fail(ctx, VS_E_RANGE, "reads of %u bases with k+1=%u need %zu B of LDS per pair (limit 160 KiB)", maxlen, idx.K, lds);

    // scratch: slow list (one slot per pair), counter, dense state
    if (ctx->slow_cap < n_pairs) {
        if (ctx->d_slow_list) VS_HIP(ctx, hipFree(ctx->d_slow_list));
        ctx->d_slow_list = nullptr;
        VS_HIP(ctx, hipMalloc(&ctx->d_slow_list, sizeof(uint32_t) * n_pairs));
        ctx->slow_cap = n_pairs;
    }
    if (!ctx->d_slow_count) VS_HIP(ctx, hipMalloc(&ctx->d_slow_count, 64));  // [0] pairs for k_pe_mid, [1] queue, [2..3] postings, [4..7] acc, [8] pairs for k_pe_slow, [9] strip queue, [10..14] k_rows_sum debug, [15] owning ends
    if (ctx->slow2_cap < n_pairs) {
        if (ctx->d_slow_list2) VS_HIP(ctx, hipFree(ctx->d_slow_list2));
        ctx->d_slow_list2 = nullptr;
        ctx->slow2_cap = 0;
        VS_HIP(ctx, hipMalloc(&ctx->d_slow_list2, sizeof(uint32_t) * n_pairs));
        ctx->slow2_cap = n_pairs;
    }
    const uint32_t SLOW_GRID = slow_grid_for(idx.n_nodes);
    uint64_t need_dense = sizeof(uint32_t) * (uint64_t)SLOW_WORDS_PER_NODE * (idx.n_nodes ? idx.n_nodes : 1) * SLOW_GRID;
    if (ctx->dense_bytes < need_dense || ctx->dense_nodes != idx.n_nodes) {
        if (ctx->d_dense) VS_HIP(ctx, hipFree(ctx->d_dense));
        ctx->d_dense = nullptr;
        ctx->dense_bytes = 0;
        VS_HIP(ctx, hipMalloc(&ctx->d_dense, need_dense));
        ctx->dense_bytes = need_dense;
        ctx->dense_nodes = idx.n_nodes;
        // cnt = 0, minp/minj = ~0 once; k_pe_slow restores this state after every end it sweeps
        uint64_t N = idx.n_nodes ? idx.n_nodes : 1;
        VS_HIP(ctx, hipMemsetAsync(ctx->d_dense, 0xFF, need_dense, st));
        hipLaunchKernelGGL(k_dense_zero_cnt, dim3((unsigned)((N * SLOW_GRID + TPB - 1) / TPB)), dim3(TPB), 0, st,
                           (uint32_t *)ctx->d_dense, N, (uint64_t)SLOW_GRID);
    }
    VS_HIP(ctx, hipMemsetAsync(ctx->d_slow_count, 0, 64, st));

    // locus order of the pairs (see k_pe_locus); VS_NO_SORT=1 keeps the input order
    const bool use_sort = !tn.no_sort && n_pairs >= 4096 && n_pairs < 0xFFFFFFF0ull;
    if (use_sort) {
        const uint64_t nk = (uint64_t)idx.n_nodes + 2u;
        if (ctx->locus_cap < n_pairs) {
            if (ctx->d_locus_keys) VS_HIP(ctx, hipFree(ctx->d_locus_keys));
            if (ctx->d_perm) VS_HIP(ctx, hipFree(ctx->d_perm));
            ctx->d_locus_keys = ctx->d_perm = nullptr;
            ctx->locus_cap = 0;
            VS_HIP(ctx, hipMalloc(&ctx->d_locus_keys, sizeof(uint32_t) * n_pairs));
            VS_HIP(ctx, hipMalloc(&ctx->d_perm, sizeof(uint32_t) * n_pairs));
            ctx->locus_cap = n_pairs;
        }
        const uint64_t hist_words = nk <= LOCUS_LDS_MAX_PASSES * LOCUS_LDS_KEYS ? nk * LOCUS_WGS : nk;
        if (ctx->hist_cap < hist_words) {
            if (ctx->d_locus_hist) VS_HIP(ctx, hipFree(ctx->d_locus_hist));
            if (ctx->d_scan_tmp) VS_HIP(ctx, hipFree(ctx->d_scan_tmp));
            ctx->d_locus_hist = ctx->d_scan_tmp = nullptr;
            ctx->hist_cap = 0;
            VS_HIP(ctx, hipMalloc(&ctx->d_locus_hist, sizeof(uint32_t) * hist_words));
            VS_HIP(ctx, hipMalloc(&ctx->d_scan_tmp, sizeof(uint64_t) * (hist_words / 2048 + 4)));
            ctx->hist_cap = hist_words;
        }
    }

    // which counter kernels follow decides the layout of the hand-off: pair-major with one cell table while 2*N*N fits its
    // 32-bit keys (packed lists), by row owners above (rows of LCAP words); VS_ACC_ROWS=0 / 1 overrides (0 beyond 46 340
    // nodes: no table, every increment a global atomic).  VS_NO_AGG=1 turns the summing in LDS off
    uint32_t use_table = tn.no_agg ? 0u : 1u;
    if (tn.acc_ablate >= 0) use_table = (uint32_t)tn.acc_ablate;  // 2: decode only, 3: no write-outs (VS_EXPERIMENT=timing only)
    const bool fits32 = 2ull * idx.n_nodes * idx.n_nodes < 0xFFFFFFFFull;
    bool use_rows = tn.acc_rows >= 0 ? tn.acc_rows != 0 : !fits32;
    if (idx.n_nodes == 0 || use_table != 1u) use_rows = false;
    if (!use_rows && !fits32 && use_table == 1u) use_table = 0u;
    // per-end lists handed from k_pe_tiles to k_pe_accumulate
    const uint64_t n_tiles_all = (n_pairs + ept / 2 - 1) / (ept / 2);
    const uint64_t list_ends = n_tiles_all * ept;
    const uint64_t list_words = list_ends * (use_rows ? LC + 4u : LC) + 16u;  // (either layout of the hand-off)
    if (d_node_mat && (ctx->lists_cap < list_ends || ctx->lists_words < list_words)) {
        if (ctx->d_lists) VS_HIP(ctx, hipFree(ctx->d_lists));
        if (ctx->d_list_counts) VS_HIP(ctx, hipFree(ctx->d_list_counts));
        ctx->d_lists = ctx->d_list_counts = nullptr;
        ctx->lists_cap = ctx->lists_words = 0;
        VS_HIP(ctx, hipMalloc(&ctx->d_lists, sizeof(uint32_t) * list_words));
        VS_HIP(ctx, hipMalloc(&ctx->d_list_counts, sizeof(uint32_t) * (list_ends + 2)));
        ctx->lists_cap = list_ends;
        ctx->lists_words = list_words;
    }

    PeParams P;
    P.idx = idx;
    P.rd = reads->dev();
    P.out_lists = (uint32_t *)ctx->d_lists;
    P.out_counts = (uint32_t *)ctx->d_list_counts;
    P.node_mat = d_node_mat;
    P.short_mat = d_short_mat;
    P.stats = (unsigned long long *)d_stats;
    P.ept = ept;
    P.pmax = pmax;
    P.words_cap = ept * wpe;
    P.pool = pool_for(ept, &P.pool_bits);
    P.debug_stop = tn.debug_stop;  // (VS_EXPERIMENT=timing only)
    P.n_pairs = n_pairs;
    P.n_tiles = (n_pairs + ept / 2 - 1) / (ept / 2);
    P.wpe = wpe;
    P.count_postings = tn.debug_postings ? 1u : 0u;
    P.magic_pmax = pmax > 1u ? (uint32_t)(0x100000000ull / pmax) + 1u : 0u;
    P.magic_wpe = wpe > 1u ? (uint32_t)(0x100000000ull / wpe) + 1u : 0u;
    P.perm = use_sort ? (const uint32_t *)ctx->d_perm : nullptr;
    P.slow_list = (uint32_t *)ctx->d_slow_list;
    P.slow_count = (uint32_t *)ctx->d_slow_count;
    P.dbg_lists = d_dbg_lists;
    P.dbg_counts = d_dbg_counts;
    P.dbg_cap = dbg_cap;
    P.accumulate = d_node_mat ? 1u : 0u;
    P.out_rows = use_rows ? 1u : 0u;
    P.out_lists_hi = (uint32_t *)ctx->d_lists + list_ends * LC;
    P.tile_map = d_node_mat ? d_tile_map : nullptr;
    P.tile_T = (idx.n_nodes + 63u) >> 6;
    P.no_xcd_map = tn.no_xcd_map ? 1u : 0u;
    P.phase0 = tn.phase0 ? 1u : 0u;
    // The shortcut spares a single posting its extension when the previous probe already owns the
    // match; it pays on graphs whose seeds are mostly unique.  Where seeds repeat (a compacted de
    // Bruijn graph of many strains: 3.5 postings per distinct seed at configs[2]) nearly every
    // wavefront holds some single posting and all 64 lanes walk through the test for it: 6.0 ms with,
    // 5.8 ms without.  VS_SHORTCUT=0/1 overrides.
    P.shortcut = ctx->n_distinct && ctx->n_seed_pos < 2 * ctx->n_distinct ? 1u : 0u;
    if (tn.shortcut >= 0) P.shortcut = (uint32_t)tn.shortcut;
    // (63-base seeds have MIXED keys: equal keys do not prove equal seeds, so "the bases in between match too" does not
    // follow from two key hits on one diagonal -- the shortcut is for exact keys only, whatever the switch says)
    if (VS_SEED_VERIFIED(idx.w) == 0u) P.shortcut = 0u;

    // straight-line extension when the whole block qualifies (see vs_extend_fast)
    // (reads with bytes outside ACGT qualify through their position lists, see k_inv4 / vs_seed_limits)
    // (and nodes below 2^23 bases: the straight-line kernels carry a node's length above the read offset in one table word -- a
    // longer node takes the generic kernel; r6: the guard was missing for MODE 1)
    const bool fast = (!reads->d_mask || reads->d_inv4) && idx.s <= 32u && maxlen <= 128u + idx.w + 32u &&
                      ctx->max_node_len < (1u << 23) && !tn.no_fast;
    P.mid_fast = fast && VS_SEED_VERIFIED(idx.w) && ctx->max_node_len < (1u << 23) ? 1u : 0u;
    // compile-time-shape instantiations (see k_pe_tiles): 1 = (10, 4), 2 = (8, 3), 3 = (7, 2)
    int std_shape = 0;
    if (fast && ept == STD_EPT && P.pool_bits == STD_POOL_BITS && maxlen <= 159u && idx.K == STD_K && idx.w == STD_W &&
        idx.s == STD_S && P.accumulate && !P.debug_stop && !P.count_postings && !P.dbg_counts &&
        !tn.no_std && !tn.phase0) {
        if (wpe == 10u && pmax == 4u) std_shape = 1;       // 2 x 145..159 bases
        else if (wpe == 8u && pmax == 3u) std_shape = 2;   // 2 x 113..128
        else if (wpe == 7u && pmax == 2u) std_shape = 3;   // 2 x 97..107
        else if (wpe == 7u && pmax == 3u) std_shape = 5;   // 2 x 108..112 (r6, ADVICE r5: these took the generic kernel)
    }
    // longer strides and reads (k = 127 with 2 x 250 bases): the straight-line kernel with more windows
    // Its eight right windows compare 256 bases from where the comparison starts (behind a verified seed, at the first base
    // of a 63-base one); what has to fit is the read's part behind its FIRST probe, and the grid does not start at offset 0
    // (vs_seed_phase): reads of up to 317 bases qualify at k = 127 (r5; 256 with the grid of rounds 1-4)
    uint32_t long_reach = 0;
    for (uint32_t len = idx.K; len <= maxlen; len++) {
        const uint32_t behind = len - vs_seed_phase(len, idx.w, idx.s, tn.phase0) - VS_SEED_VERIFIED(idx.w);
        long_reach = behind > long_reach ? behind : long_reach;
    }
    const bool fast_long = !fast && (!reads->d_mask || reads->d_inv4) && idx.s <= 128u && long_reach <= 256u && maxlen < 512u &&
                           ctx->max_node_len < (1u << 23) && !tn.no_fast;
    if (fast_long && ept == STD2_EPT && P.pool_bits == STD2_POOL_BITS && idx.K == STD2_K && idx.w == STD2_W && idx.s == STD2_S &&
        wpe == 16u && pmax == 2u && P.accumulate && !P.debug_stop && !P.count_postings && !P.dbg_counts && !tn.no_std && !tn.phase0)
        std_shape = 4;
    // (r5) the adaptive step grid (k_pe_tiles, ADAPT) probes twice as many seeds to expand fewer postings: it pays where a
    // seed has many postings -- 8.9 seed positions per distinct seed at configs[4]: 24.8 -> 22.5 ms -- and costs a little
    // where it has few (3.6 at configs[2]: 4.80 -> 4.87 ms).  VS_ADAPT_GRID=0 / 1 overrides.
    bool adapt = std_shape != 0 && ctx->n_distinct && ctx->n_seed_pos >= 6u * ctx->n_distinct;
    if (tn.adapt_grid >= 0) adapt = std_shape != 0 && tn.adapt_grid != 0;
    struct TilesFn { const void *fn; const char *name; };
    const TilesFn tf = std_shape == 1   ? (adapt ? TilesFn{(const void *)k_pe_tiles<1, 10u, 4u, true>, "k_pe_tiles<1, 10u, 4u, true>"}
                                                 : TilesFn{(const void *)k_pe_tiles<1, 10u, 4u>, "k_pe_tiles<1, 10u, 4u>"})
                       : std_shape == 2 ? (adapt ? TilesFn{(const void *)k_pe_tiles<1, 8u, 3u, true>, "k_pe_tiles<1, 8u, 3u, true>"}
                                                 : TilesFn{(const void *)k_pe_tiles<1, 8u, 3u>, "k_pe_tiles<1, 8u, 3u>"})
                       : std_shape == 3 ? (adapt ? TilesFn{(const void *)k_pe_tiles<1, 7u, 2u, true>, "k_pe_tiles<1, 7u, 2u, true>"}
                                                 : TilesFn{(const void *)k_pe_tiles<1, 7u, 2u>, "k_pe_tiles<1, 7u, 2u>"})
                       : std_shape == 5 ? (adapt ? TilesFn{(const void *)k_pe_tiles<1, 7u, 3u, true>, "k_pe_tiles<1, 7u, 3u, true>"}
                                                 : TilesFn{(const void *)k_pe_tiles<1, 7u, 3u>, "k_pe_tiles<1, 7u, 3u>"})
                       : fast           ? TilesFn{(const void *)k_pe_tiles<1, 0u, 0u>, "k_pe_tiles<1, 0u, 0u>"}
                       : std_shape == 4 ? (adapt ? TilesFn{(const void *)k_pe_tiles<2, 16u, 2u, true>, "k_pe_tiles<2, 16u, 2u, true>"}
                                                 : TilesFn{(const void *)k_pe_tiles<2, 16u, 2u>, "k_pe_tiles<2, 16u, 2u>"})
                       : fast_long      ? TilesFn{(const void *)k_pe_tiles<2, 0u, 0u>, "k_pe_tiles<2, 0u, 0u>"}
                                        : TilesFn{(const void *)k_pe_tiles<0, 0u, 0u>, "k_pe_tiles<0, 0u, 0u>"};
    const void *tiles_fn = tf.fn;
    if (std_shape && VS_POOL12 && !adapt) lds = (size_t)tile_layout(ept, pmax, ept * wpe, 768u, std_shape == 4 ? 192u : 64u).total * sizeof(uint32_t);
    if (lds > 64u * 1024u)
        VS_HIP(ctx, hipFuncSetAttribute(tiles_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (tn.debug_occ) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tiles_fn, TTPB, lds);
        fprintf(stderr, "[vs] k_pe_tiles: %zu B of LDS per workgroup, %d workgroups per CU\n", lds, nb);
    }
    uint64_t grid = P.n_tiles;
    // Many more workgroups than fit at once (4 per CU): loci differ a lot in postings per read, and
    // short runs let the dispatcher even that out (runs of ~10 tiles at configs[2]: 8.4 ms, against
    // 10.0 ms with 8 workgroups per CU and 9.0 ms with one tile per workgroup)
    const uint64_t max_grid = (uint64_t)ctx->n_cu * tn.grid_per_cu;
    if (grid > max_grid) grid = max_grid;
    P.tiles_per_wg = (uint32_t)((P.n_tiles + grid - 1) / grid);
    grid = (P.n_tiles + P.tiles_per_wg - 1) / P.tiles_per_wg;
    VS_HIP(ctx, hipEventRecord(ctx->ev[3], st));
    if (use_sort) {
        const uint64_t nk = (uint64_t)idx.n_nodes + 2u;
        const bool lds_sort = nk <= LOCUS_LDS_MAX_PASSES * LOCUS_LDS_KEYS && !tn.locus_global;
        ctx->last_launched |= lds_sort ? VS_RAN_LOCUS_LDS_SORT : VS_RAN_LOCUS_GLOBAL_SORT;
        if (lds_sort) {
            const uint32_t n_wg = LOCUS_WGS;
            const uint32_t chunk = (uint32_t)((n_pairs + n_wg - 1) / n_wg);
            const uint32_t per_pass = (uint32_t)(nk < LOCUS_LDS_KEYS ? nk : LOCUS_LDS_KEYS);
            const size_t lds_keys = sizeof(uint32_t) * per_pass;
            if (lds_keys > 64u * 1024u) {
                VS_HIP(ctx, hipFuncSetAttribute((const void *)k_locus_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_keys));
                VS_HIP(ctx, hipFuncSetAttribute((const void *)k_locus_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_keys));
            }
            for (uint32_t key_lo = 0; key_lo < nk; key_lo += per_pass) {
                const uint32_t n_here = (uint32_t)(nk - key_lo < per_pass ? nk - key_lo : per_pass);
                hipLaunchKernelGGL(k_locus_count, dim3(n_wg), dim3(LOCUS_TPB), lds_keys, st, idx, reads->dev(), n_pairs, chunk, n_wg,
                                   (uint32_t *)ctx->d_locus_keys, (uint32_t *)ctx->d_locus_hist, key_lo, n_here, key_lo == 0u ? 1u : 0u);
            }
            int rc = vs_scan_u32(ctx, (const uint32_t *)ctx->d_locus_hist, (uint32_t *)ctx->d_locus_hist, nk * n_wg,
                                 (uint64_t *)ctx->d_scan_tmp, nullptr);
            if (rc) return rc;
            for (uint32_t key_lo = 0; key_lo < nk; key_lo += per_pass) {
                const uint32_t n_here = (uint32_t)(nk - key_lo < per_pass ? nk - key_lo : per_pass);
                hipLaunchKernelGGL(k_locus_scatter, dim3(n_wg), dim3(LOCUS_TPB), lds_keys, st, key_lo, n_here, n_pairs, chunk, n_wg,
                                   (const uint32_t *)ctx->d_locus_keys, (const uint32_t *)ctx->d_locus_hist, (uint32_t *)ctx->d_perm);
            }
        } else {
            VS_HIP(ctx, hipMemsetAsync(ctx->d_locus_hist, 0, sizeof(uint32_t) * nk, st));
            const unsigned pg = (unsigned)((n_pairs + TPB - 1) / TPB);
            hipLaunchKernelGGL(k_pe_locus, dim3(pg), dim3(TPB), 0, st, idx, reads->dev(), n_pairs, (uint32_t *)ctx->d_locus_keys,
                               (uint32_t *)ctx->d_locus_hist);
            int rc = vs_scan_u32(ctx, (const uint32_t *)ctx->d_locus_hist, (uint32_t *)ctx->d_locus_hist, nk,
                                 (uint64_t *)ctx->d_scan_tmp, nullptr);
            if (rc) return rc;
            hipLaunchKernelGGL(k_pe_permute, dim3(pg), dim3(TPB), 0, st, n_pairs, (const uint32_t *)ctx->d_locus_keys,
                               (uint32_t *)ctx->d_locus_hist, (uint32_t *)ctx->d_perm);
        }
    }
    VS_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    if (!d_node_mat) VS_HIP(ctx, hipEventRecord(ctx->ev[4], st));
    ctx->last_kernel = tf.name;
    {
        void *kargs[] = {(void *)&P};
        VS_HIP(ctx, hipLaunchKernel(tiles_fn, dim3((unsigned)grid), dim3(TTPB), kargs, lds, st));
    }
    if (d_node_mat) {
        // the last tile may be partly empty: its unused rows must read as length 0
        const uint64_t used_ends = 2ull * n_pairs;
        if (list_ends > used_ends)
            VS_HIP(ctx, hipMemsetAsync((uint32_t *)ctx->d_list_counts + used_ends, 0, sizeof(uint32_t) * (list_ends - used_ends), st));
        const uint64_t slots_pairs = list_ends / 2;
        VS_HIP(ctx, hipEventRecord(ctx->ev[4], st));
        if (use_rows) {
            const int rc = pe_count_by_rows(ctx, slots_pairs, d_node_mat, d_short_mat, d_tile_map, P.tile_T);
            if (rc) return rc;
            ctx->last_launched |= VS_RAN_ROW_OWNERS;
        } else {
            // one workgroup fits per CU (the cell table); 32 per CU queued, for the same reason as above
            // (4.9 -> 3.9 ms), each at least one round of ACC_TPB pairs
            uint32_t acc_grid = (uint32_t)ctx->n_cu * tn.acc_grid_per_cu;
            uint32_t per_wg = (uint32_t)((slots_pairs + acc_grid - 1) / acc_grid);
            per_wg = (per_wg + ACC_TPB - 1) / ACC_TPB * ACC_TPB;
            acc_grid = (uint32_t)((slots_pairs + per_wg - 1) / per_wg);
            // the chunks are not bound to workgroups: two workgroups per CU take the next chunk off a
            // counter whenever they are free, so a cell table lives across chunks and is written out on
            // fill only (3.85 -> 3.6 ms against one workgroup per chunk; VS_ACC_QUEUE=0 for that)
            uint32_t *acc_queue = nullptr;
            if (tn.acc_queue) {
                acc_queue = (uint32_t *)ctx->d_slow_count + 1;
                uint32_t wgs = (uint32_t)ctx->n_cu * 2u;
                if (tn.acc_wgs) wgs = tn.acc_wgs;  // VS_ACC_WGS: fewer workgroups than CUs = the kernel on a part of the chip (r5 gate)
                if (acc_grid > wgs) acc_grid = wgs;
            }
            // the table is written out once this many of its slots are taken: probing stays short at a low
            // fill, and cells of loci the run has left do not pile up (VS_ACC_FILL: percent)
            uint32_t fill_limit = ACC_SLOTS / 16u;
            if (tn.acc_fill_pct >= 0) fill_limit = (uint32_t)((uint64_t)ACC_SLOTS * (uint32_t)tn.acc_fill_pct / 100u);
            uint32_t *acc_dbg = tn.debug_acc ? (uint32_t *)ctx->d_slow_count + 4 : nullptr;
            const uint32_t acc_ppw = tn.acc_round ? tn.acc_round / (ACC_TPB / 64u) : 64u;  // VS_ACC_ROUND: pairs per round
            if (d_tile_map && slots_pairs)  // (timed with the counter kernel: it is part of the counting)
                hipLaunchKernelGGL(k_mark_tiles, dim3((unsigned)((slots_pairs + 255u) / 256u)), dim3(256), 0, st, (const uint32_t *)ctx->d_lists,
                                   (const uint32_t *)ctx->d_list_counts, slots_pairs, d_tile_map, P.tile_T, ept);
            VS_HIP(ctx, hipFuncSetAttribute((const void *)k_pe_accumulate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ACC_LDS_BYTES));
            hipLaunchKernelGGL(k_pe_accumulate, dim3(acc_grid), dim3(ACC_TPB), ACC_LDS_BYTES, st, (const uint32_t *)ctx->d_lists,
                               (const uint32_t *)ctx->d_list_counts, slots_pairs, per_wg, idx.n_nodes, use_table, fill_limit, d_node_mat,
                               d_short_mat, acc_queue, acc_dbg, acc_ppw, ept);
        }
    }
    VS_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    // overflow pairs: one wavefront per pair with its state in LDS first, the general kernel for what that cannot hold
    ctx->last_launched |= tn.no_mid ? 0u : VS_RAN_PE_MID;
    if (!tn.no_mid) {
        hipLaunchKernelGGL(k_pe_mid, dim3((unsigned)ctx->n_cu * 8u), dim3(TPB), 0, st, P, (const uint32_t *)ctx->d_slow_list,
                           (const uint32_t *)ctx->d_slow_count, (uint32_t)n_pairs, (uint32_t *)ctx->d_slow_list2, (uint32_t *)ctx->d_slow_count + 8);
        hipLaunchKernelGGL(k_pe_slow, dim3(SLOW_GRID), dim3(TPB), 0, st, P, (uint32_t *)ctx->d_dense, (uint32_t)n_pairs,
                           (const uint32_t *)ctx->d_slow_list2, (const uint32_t *)ctx->d_slow_count + 8);
    } else {
        hipLaunchKernelGGL(k_pe_slow, dim3(SLOW_GRID), dim3(TPB), 0, st, P, (uint32_t *)ctx->d_dense, (uint32_t)n_pairs,
                           (const uint32_t *)ctx->d_slow_list, (const uint32_t *)ctx->d_slow_count);
    }
    VS_HIP(ctx, hipEventRecord(ctx->ev[2], st));
    VS_HIP(ctx, hipGetLastError());
    return VS_OK;
}

extern "C" int vs_pe_count(vs_ctx *ctx, const vs_reads *reads, uint32_t *d_node_mat, uint32_t *d_short_mat, uint64_t *d_stats) {
    if (!ctx || !reads || !d_node_mat || !d_short_mat || !d_stats) return VS_E_ARG;
    return pe_launch(ctx, reads, d_node_mat, d_short_mat, d_stats, nullptr, nullptr, 0);
}

extern "C" int vs_pe_count_tracked(vs_ctx *ctx, const vs_reads *reads, uint32_t *d_node_mat, uint32_t *d_short_mat, uint64_t *d_stats,
                                   uint8_t *d_tile_map) {
    if (!ctx || !reads || !d_node_mat || !d_short_mat || !d_stats || !d_tile_map) return VS_E_ARG;
    return pe_launch(ctx, reads, d_node_mat, d_short_mat, d_stats, nullptr, nullptr, 0, d_tile_map);
}

extern "C" int vs_counts_zero_tracked(vs_ctx *ctx, uint32_t *d_node_mat, uint32_t *d_short_mat, uint32_t n, uint8_t *d_tile_map) {
    if (!ctx || (n && (!d_node_mat || !d_short_mat || !d_tile_map))) return vs_fail(ctx, VS_E_ARG, "vs_counts_zero_tracked: bad argument");
    if (!n) return VS_OK;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t T = (n + 63u) >> 6;
    const uint64_t tiles = 2ull * T * T;
    if (tiles > 0x7FFFFFFFull) return vs_fail(ctx, VS_E_RANGE, "vs_counts_zero_tracked: %u nodes make more tiles than one launch takes", n);
    hipLaunchKernelGGL(k_zero_tiles, dim3((unsigned)((tiles + 63u) / 64u)), dim3(64), 0, ctx->stream, d_node_mat, d_short_mat, n, T, d_tile_map, tiles);
    VS_HIP(ctx, hipGetLastError());
    return VS_OK;
}

extern "C" const char *vs_pe_last_kernel(const vs_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }
extern "C" uint32_t vs_pe_last_launched(const vs_ctx *ctx) { return ctx ? ctx->last_launched : 0u; }

extern "C" int vs_pe_last_timing(vs_ctx *ctx, double ms[5]) {
    if (!ctx || !ms) return VS_E_ARG;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    VS_HIP(ctx, hipEventSynchronize(ctx->ev[2]));
    float a = 0, b = 0, c = 0, d = 0;
    VS_HIP(ctx, hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]));
    VS_HIP(ctx, hipEventElapsedTime(&d, ctx->ev[4], ctx->ev[1]));
    if (d > a) d = 0;  // no accumulate pass in this call
    a -= d;
    VS_HIP(ctx, hipEventElapsedTime(&b, ctx->ev[1], ctx->ev[2]));
    VS_HIP(ctx, hipEventElapsedTime(&c, ctx->ev[3], ctx->ev[0]));
    ctx->last_sort_ms = c;
    uint32_t n_slow = 0;
    VS_HIP(ctx, hipMemcpy(&n_slow, ctx->d_slow_count, sizeof n_slow, hipMemcpyDeviceToHost));
    ms[0] = a; ms[1] = b; ms[2] = (double)n_slow; ms[3] = c; ms[4] = d;
    if (ctx->tune.debug_postings) {
        unsigned long long np = 0;
        VS_HIP(ctx, hipMemcpy(&np, (char *)ctx->d_slow_count + 8, sizeof np, hipMemcpyDeviceToHost));
        fprintf(stderr, "[vs] postings expanded by the last vs_pe_count: %llu\n", np);
    }
    if (ctx->tune.debug_acc) {
        uint32_t d4[4] = {0, 0, 0, 0};
        VS_HIP(ctx, hipMemcpy(d4, (char *)ctx->d_slow_count + 16, sizeof d4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[vs] k_pe_accumulate: %u increments went past the cell table, %u write-outs of %u cells, %u rounds\n", d4[0], d4[1], d4[2], d4[3]);
        if (ctx->last_launched & VS_RAN_ROW_OWNERS) {
            uint32_t r3[3] = {0, 0, 0};
            VS_HIP(ctx, hipMemcpy(r3, (char *)ctx->d_slow_count + 40, sizeof r3, hipMemcpyDeviceToHost));
            fprintf(stderr, "[vs] k_rows_sum (both matrices): %u increments went past the cell table, %u write-outs of %u cells\n", r3[0], r3[1], r3[2]);

            uint32_t e2[2] = {0, 0};  // entries of the last transposition: row_ptr[N] of either mode
            const uint32_t *rows = (const uint32_t *)ctx->d_rows;
            VS_HIP(ctx, hipMemcpy(&e2[0], rows + 2u * ctx->rows_cap + ctx->idx.n_nodes, sizeof(uint32_t), hipMemcpyDeviceToHost));
            VS_HIP(ctx, hipMemcpy(&e2[1], rows + 5u * ctx->rows_cap + ctx->idx.n_nodes, sizeof(uint32_t), hipMemcpyDeviceToHost));
            fprintf(stderr, "[vs] row entries: node_mat %u, short_mat %u\n", e2[0], e2[1]);
        }
    }
    ctx->last_ms[0] = a; ctx->last_ms[1] = b; ctx->last_ms[2] = n_slow;
    return VS_OK;
}

extern "C" int vs_pe_map_ends(vs_ctx *ctx, const vs_reads *reads, uint32_t cap, uint32_t *lists, uint32_t *counts) {
    if (!ctx || !reads || !lists || !counts || !cap) return VS_E_ARG;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t n = reads->n_ends;
    if (!n) return VS_OK;
    uint32_t *d_lists = nullptr, *d_counts = nullptr;
    VS_HIP(ctx, hipMalloc((void **)&d_lists, sizeof(uint32_t) * n * cap));
    hipError_t e1 = hipMalloc((void **)&d_counts, sizeof(uint32_t) * n);
    int rc = VS_OK;
    if (e1 == hipSuccess) e1 = hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * n, ctx->stream);
    if (e1 == hipSuccess) e1 = hipMemsetAsync(d_lists, 0xFF, sizeof(uint32_t) * n * cap, ctx->stream);
    if (e1 == hipSuccess) rc = pe_launch(ctx, reads, nullptr, nullptr, nullptr, d_lists, d_counts, cap);
    if (e1 == hipSuccess && rc == VS_OK) e1 = hipMemcpyAsync(lists, d_lists, sizeof(uint32_t) * n * cap, hipMemcpyDeviceToHost, ctx->stream);
    if (e1 == hipSuccess && rc == VS_OK) e1 = hipMemcpyAsync(counts, d_counts, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e1 == hipSuccess && rc == VS_OK) e1 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_lists);
    if (d_counts) (void)hipFree(d_counts);
    if (e1 != hipSuccess) return vs_fail(ctx, VS_E_HIP, "vs_pe_map_ends: %s", hipGetErrorString(e1));
    return rc;
}
