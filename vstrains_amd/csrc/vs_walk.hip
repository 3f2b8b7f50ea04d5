// K2+K3 for certified graphs: single_end_read_mapping (utils/VStrains_PE_Inference.py:16-48) by FOLLOWING the read
// through the graph.  vs_walk.h states what the index certifies: every (k+1)-mer has one home, the window after a
// window at (node strand, q) sits at (same strand, q + 1) or -- at the strand's end -- at position 0 of the successor
// its next base selects, or nowhere.  So the coincidences of PE_Inference.py:23-31 of a read end are runs of consecutive
// windows, and per end the kernel needs
//   * one exact (k+1)-mer lookup per run (hash of the window, one 16-B slot, the text compared: no fingerprint decides),
//   * per node of the run one 32-B record (length, 32 bases of text behind the overlap, four successors) and one or two
//     64-bit comparisons: v += windows in the node, coords = first forward offset, kindices = first window
//     (reverse strand: the forward offset of a window at q is len - K - q, smallest at the LAST window, :132),
//   * where a run breaks at read base p: every window over p is unresolved.  Any such window holds the wp-mer that ends
//     at p or the one that starts at p (wp = K/2 + 1: one flank of p inside a K-window has that many bases); if neither
//     occurs in any node (presence set, two 8-B probes) none of them can coincide and the scan resumes at p + 1 -- the
//     case of a sequencing error.  Otherwise, and after a missed lookup, windows are skipped only as far as absent
//     wp-mers prove them empty.
// One lane per read end, 256 ends (128 pairs, locus order) per tile and workgroup; the packed reads of a tile sit in
// LDS, a lane's touched nodes in an LDS row of its own (node | accepted << 31).  A node met twice in a row (an error
// inside a node) is merged in registers; a node met again later (a read around a short cycle), more than LC touched
// nodes, or more bytes outside ACGT than inv4 holds send the pair to the general overflow kernel (k_pe_slow).
// Output: the per-end accepted lists k_pe_accumulate reads (same layout as k_pe_tiles writes).
#include <stdio.h>

#include "vs_internal.h"

#define WTPB 256u  // threads = read ends per tile
#define WLC 16u    // list row of an end (= LC of vs_pe.hip: the row k_pe_accumulate reads)
#define WNONE 0xFFFFFFFFu

extern __shared__ __attribute__((aligned(16))) uint32_t vs_wlds[];

// state of the ends of a tile, one array per field (LDS), and the class queues
#define WS_FIELDS 13u
enum { F_J = 0, F_LO, F_HI, F_NS, F_Q, F_CNODE, F_CV, F_CCOORD, F_CKIDX, F_CNLEN, F_NT, F_PM, F_OVER };

__device__ __forceinline__ void ws_load(const uint32_t *st, uint32_t e, VsWalkEnd &x) {
    x.j = st[F_J * WTPB + e]; x.lo = st[F_LO * WTPB + e]; x.hi = st[F_HI * WTPB + e]; x.ns = st[F_NS * WTPB + e]; x.q = st[F_Q * WTPB + e];
    x.cur_node = st[F_CNODE * WTPB + e]; x.cur_v = st[F_CV * WTPB + e]; x.cur_coord = st[F_CCOORD * WTPB + e];
    x.cur_kidx = st[F_CKIDX * WTPB + e]; x.cur_nlen = st[F_CNLEN * WTPB + e]; x.nt = st[F_NT * WTPB + e]; x.pm = st[F_PM * WTPB + e];
    x.over = st[F_OVER * WTPB + e];
}
__device__ __forceinline__ void ws_store(uint32_t *st, uint32_t e, const VsWalkEnd &x) {
    st[F_J * WTPB + e] = x.j; st[F_LO * WTPB + e] = x.lo; st[F_HI * WTPB + e] = x.hi; st[F_NS * WTPB + e] = x.ns; st[F_Q * WTPB + e] = x.q;
    st[F_CNODE * WTPB + e] = x.cur_node; st[F_CV * WTPB + e] = x.cur_v; st[F_CCOORD * WTPB + e] = x.cur_coord;
    st[F_CKIDX * WTPB + e] = x.cur_kidx; st[F_CNLEN * WTPB + e] = x.cur_nlen; st[F_NT * WTPB + e] = x.nt; st[F_PM * WTPB + e] = x.pm;
    st[F_OVER * WTPB + e] = x.over;
}
// Append the ends of the lanes with `pred` to a queue: wavefront ballot, one LDS atomic per wavefront for the block of
// slots, prefix count of the lanes below for the place inside it.  Every lane of the wavefront calls it.
__device__ __forceinline__ void ws_push(uint32_t *queue, uint32_t *counter, bool pred, uint32_t e) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__builtin_ctzll(mask);
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader, 64);
    if (pred) queue[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = e;
}

template <uint32_t NW>
__global__ void __launch_bounds__(WTPB) k_pe_walk(VsWalkParams P) {
    const uint32_t tid = threadIdx.x;
    const uint32_t K = P.idx.K;
    const uint32_t ws = P.wpe | 1u;  // LDS row stride of an end's packed words (odd: lanes hit different banks)
    constexpr uint32_t LS = WLC + 1u;  // ... of its list row
    uint32_t *s_words = vs_wlds;                   // [WTPB * ws + 8]
    uint32_t *s_list = s_words + WTPB * ws + 8u;   // [WTPB * LS] node | accepted << 31 per touched node
    uint32_t *s_st = s_list + WTPB * LS;           // [WS_FIELDS][WTPB] VsWalkEnd, one array per field
    uint32_t *s_meta = s_st + WS_FIELDS * WTPB;    // [WTPB] length | flags << 24
    uint32_t *s_inv = s_meta + WTPB;               // [WTPB] positions of bytes outside ACGT
    uint32_t *s_gend = s_inv + WTPB;               // [WTPB] global end index
    uint32_t *s_qw = s_gend + WTPB;                // [2][WTPB] class W, this pass / next pass
    uint32_t *s_qp = s_qw + 2u * WTPB;             // [2][WTPB] class P
    uint32_t *s_ql = s_qp + 2u * WTPB;             // [WTPB]    class L (also scratch while the words are loaded)
    uint32_t *s_cnt = s_ql + WTPB;                 // [8]: W cur/nxt at [0],[1]; P at [2],[3]; L at [4]
    uint32_t *s_misc = s_cnt + 8u;                 // [4] pair classes of this workgroup
    const uint32_t *text = P.idx.fwd_words;
    const bool has_inv = P.rd.inv4 != nullptr;
    if (tid < 3u) s_misc[tid] = 0u;

    uint32_t wg = blockIdx.x;  // XCD x works through the x-th eighth of the locus order (see k_pe_tiles)
    if ((gridDim.x & 7u) == 0u && !P.no_xcd_map) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint64_t tile_lo = (uint64_t)wg * P.tiles_per_wg;
    const uint64_t tile_hi = tile_lo + P.tiles_per_wg < P.n_tiles ? tile_lo + P.tiles_per_wg : P.n_tiles;

    for (uint64_t tile = tile_lo; tile < tile_hi; tile++) {
        const uint64_t p0 = tile * (WTPB / 2u);
        const uint32_t npair = (uint32_t)((P.n_pairs - p0) < (WTPB / 2u) ? (P.n_pairs - p0) : (WTPB / 2u));
        const uint32_t ne = 2u * npair;
        __syncthreads();  // the previous tile's rows are copied out
        // ---- headers and packed words of the tile
        uint32_t meta = 0u, inv = 0xFFFFFFFFu, gwoff = 0u, gend = 0u;
        if (tid < ne) {
            const uint64_t p = p0 + (tid >> 1);
            const uint32_t pair = P.perm ? P.perm[p] : (uint32_t)p;
            gend = 2u * pair + (tid & 1u);
            gwoff = P.rd.woff[gend];
            meta = P.rd.meta[gend];
            if (has_inv) inv = P.rd.inv4[gend];
        }
        const uint32_t rlen = meta & VS_LEN_MASK;
        s_gend[tid] = gend;
        s_meta[tid] = meta;
        s_inv[tid] = inv;
        s_qw[tid] = gwoff;                                     // (scratch: first word of the end in the block's words)
        s_qp[tid] = tid < ne ? (rlen + 15u) >> 4 : 0u;          // (scratch: its packed words)
        if (tid < 8u) { s_words[WTPB * ws + tid] = 0u; s_cnt[tid] = 0u; }
        __syncthreads();
        // packed reads, coalesced: consecutive lanes fetch consecutive words of a read (row words past the read's own
        // are zeroed: windows may read beyond its end, and what they find there is masked off)
        for (uint32_t i = tid; i < WTPB * ws; i += WTPB) {
            const uint32_t e = vs_fastdiv(i, P.magic_ws), k2 = i - e * ws;
            s_words[i] = k2 < s_qp[e] ? P.rd.words[s_qw[e] + k2] : 0u;
        }
        // pair classification (PE_Inference.py:160-165): the partner's header through a cross-lane read
        uint32_t state = 0u;
        {
            const uint32_t pm = __shfl_xor(meta, 1, 64);
            if (tid < ne) {
                const uint32_t fl = ((meta | pm) >> 24);
                uint32_t cls;
                if (fl & VS_FLAG_N) cls = 0u;
                else if (rlen < K || (pm & VS_LEN_MASK) < K) cls = 1u;
                else cls = 2u;
                if (!(tid & 1u)) atomicAdd(&s_misc[cls], 1u);
                state = cls == 2u ? 1u : 0u;
                if (state && (fl & VS_FLAG_MANY)) state = 3u;  // more bytes outside ACGT than inv4 holds: overflow kernel
            }
        }
        __syncthreads();  // words in place, scratch use of the queues over
        {
            VsWalkEnd e0;
            vsw_end_init(e0, rlen, inv, has_inv && ((meta >> 24) & VS_FLAG_INVALID));
            ws_store(s_st, tid, e0);
            ws_push(s_ql, &s_cnt[4], state == 1u, tid);
        }
        // ---- passes: every end takes the steps of its class until it is done
        uint32_t cur = 0u;
        for (;;) {
            __syncthreads();
            const uint32_t n_w = s_cnt[cur], n_p0 = s_cnt[2u + cur], n_l0 = s_cnt[4];
            if (n_w + n_p0 + n_l0 == 0u) break;
            uint32_t *qw_cur = s_qw + cur * WTPB, *qw_nxt = s_qw + (cur ^ 1u) * WTPB;
            uint32_t *qp_cur = s_qp + cur * WTPB, *qp_nxt = s_qp + (cur ^ 1u) * WTPB;
            // W: follow the node
            for (uint32_t b0 = 0; b0 < n_w; b0 += WTPB) {
                const uint32_t i = b0 + tid;
                uint32_t cls = 0xFFu, e = 0u;
                if (i < n_w) {
                    e = qw_cur[i];
                    VsWalkEnd x;
                    ws_load(s_st, e, x);
                    const uint32_t rl = s_meta[e] & VS_LEN_MASK;
                    cls = vsw_step_walk(P.wk, text, K, s_words, e * ws * 16u, rl, x, s_list + e * LS, WLC);
                    if (cls == VSW_DONE) vsw_flush(x, s_list + e * LS, WLC, rl, K);
                    ws_store(s_st, e, x);
                }
                ws_push(qw_nxt, &s_cnt[cur ^ 1u], cls == VSW_W, e);
                ws_push(qp_cur, &s_cnt[2u + cur], cls == VSW_P, e);
                ws_push(s_ql, &s_cnt[4], cls == VSW_L, e);
            }
            __syncthreads();
            // P: presence probes (also of the runs that broke in this very pass)
            const uint32_t n_p = s_cnt[2u + cur];
            for (uint32_t b0 = 0; b0 < n_p; b0 += WTPB) {
                const uint32_t i = b0 + tid;
                uint32_t cls = 0xFFu, e = 0u;
                if (i < n_p) {
                    e = qp_cur[i];
                    VsWalkEnd x;
                    ws_load(s_st, e, x);
                    cls = vsw_step_probe(P.wk, K, s_words, e * ws * 16u, x);
                    ws_store(s_st, e, x);
                }
                ws_push(qp_nxt, &s_cnt[2u + (cur ^ 1u)], cls == VSW_P, e);
                ws_push(s_ql, &s_cnt[4], cls == VSW_L, e);
            }
            __syncthreads();
            // L: exact lookups
            const uint32_t n_l = s_cnt[4];
            for (uint32_t b0 = 0; b0 < n_l; b0 += WTPB) {
                const uint32_t i = b0 + tid;
                uint32_t cls = 0xFFu, e = 0u;
                if (i < n_l) {
                    e = s_ql[i];
                    VsWalkEnd x;
                    ws_load(s_st, e, x);
                    const uint32_t rl = s_meta[e] & VS_LEN_MASK;
                    cls = vsw_step_lookup<NW>(P.wk, text, K, s_words, e * ws * 16u, rl, s_inv[e], x);
                    if (cls == VSW_DONE) vsw_flush(x, s_list + e * LS, WLC, rl, K);
                    ws_store(s_st, e, x);
                }
                ws_push(qw_nxt, &s_cnt[cur ^ 1u], cls == VSW_W, e);
                ws_push(qp_nxt, &s_cnt[2u + (cur ^ 1u)], cls == VSW_P, e);
            }
            __syncthreads();
            if (tid == 0u) { s_cnt[cur] = 0u; s_cnt[2u + cur] = 0u; s_cnt[4] = 0u; }
            cur ^= 1u;
        }
        // ---- accepted nodes to the front of the row; overflowed pairs to the overflow list
        const uint32_t nt = s_st[F_NT * WTPB + tid];
        if (state == 1u && s_st[F_OVER * WTPB + tid]) state |= 2u;
        uint32_t cnt = 0u;
        if (state == 1u) {
            for (uint32_t i = 0; i < nt && i < WLC; i++) {
                const uint32_t wv = s_list[tid * LS + i];
                if (wv >> 31) s_list[tid * LS + cnt++] = wv & 0x7FFFFFFFu;
            }
        }
        const uint32_t pstate = state | __shfl_xor(state, 1, 64);
        const bool slow = (pstate & 1u) && (pstate & 2u);
        if (slow) cnt = 0u;
        if (slow && !(tid & 1u) && tid < ne) P.slow_list[atomicAdd(P.slow_count, 1u)] = gend >> 1;
        s_meta[tid] = (state & 1u) && tid < ne ? cnt : 0u;  // (the header is not needed any more: the row's length)
        __syncthreads();
        // ---- rows out, coalesced: WLC words per end, tile order (= what k_pe_accumulate reads)
        if (P.accumulate) {
            uint32_t *ol = P.out_lists + tile * (uint64_t)WTPB * WLC;
            for (uint32_t i = tid; i < ne * WLC; i += WTPB) ol[i] = s_list[(i / WLC) * LS + (i % WLC)];
            if (tid < ne) P.out_counts[tile * WTPB + tid] = s_meta[tid];
        }
        if (P.dbg_counts && tid < ne && !slow && state == 1u) {  // (the overflow kernel reports its own pairs)
            P.dbg_counts[gend] = cnt;
            for (uint32_t k2 = 0; k2 < cnt && k2 < P.dbg_cap; k2++) P.dbg_lists[(uint64_t)gend * P.dbg_cap + k2] = s_list[tid * LS + k2];
        }
        if (P.dbg_counts && tid < ne && !slow && state == 0u) P.dbg_counts[gend] = 0u;
    }
    __syncthreads();
    if (tid < 3u && P.stats && s_misc[tid]) atomicAdd(&P.stats[tid], (unsigned long long)s_misc[tid]);
}

size_t vs_walk_lds_bytes(uint32_t wpe) {
    const uint32_t ws = wpe | 1u;
    return sizeof(uint32_t) * ((size_t)WTPB * ws + 8u + (size_t)WTPB * (WLC + 1u) + (13u + 3u + 5u) * WTPB + 8u + 4u);
}

int vs_walk_launch(vs_ctx *ctx, const VsWalkParams &P, uint32_t grid, hipStream_t st) {
    const size_t lds = vs_walk_lds_bytes(P.wpe);
    const void *fn = nullptr;
    switch (P.wk.nw) {
        case 1: fn = (const void *)k_pe_walk<1>; break;
        case 2: fn = (const void *)k_pe_walk<2>; break;
        case 3: fn = (const void *)k_pe_walk<3>; break;
        case 4: fn = (const void *)k_pe_walk<4>; break;
        case 5: fn = (const void *)k_pe_walk<5>; break;
        default: return vs_fail(ctx, VS_E_RANGE, "vs_walk_launch: %u words per (k+1)-mer", P.wk.nw);
    }
    if (lds > 64u * 1024u) VS_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    switch (P.wk.nw) {
        case 1: hipLaunchKernelGGL(k_pe_walk<1>, dim3(grid), dim3(WTPB), lds, st, P); break;
        case 2: hipLaunchKernelGGL(k_pe_walk<2>, dim3(grid), dim3(WTPB), lds, st, P); break;
        case 3: hipLaunchKernelGGL(k_pe_walk<3>, dim3(grid), dim3(WTPB), lds, st, P); break;
        case 4: hipLaunchKernelGGL(k_pe_walk<4>, dim3(grid), dim3(WTPB), lds, st, P); break;
        default: hipLaunchKernelGGL(k_pe_walk<5>, dim3(grid), dim3(WTPB), lds, st, P); break;
    }
    return VS_OK;
}

const char *vs_walk_kernel_name(uint32_t nw) {
    static const char *names[6] = {"", "k_pe_walk<1>", "k_pe_walk<2>", "k_pe_walk<3>", "k_pe_walk<4>", "k_pe_walk<5>"};
    return nw >= 1u && nw <= 5u ? names[nw] : "";
}
