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

template <uint32_t NW>
__global__ void __launch_bounds__(WTPB) k_pe_walk(VsWalkParams P) {
    const uint32_t tid = threadIdx.x;
    const uint32_t K = P.idx.K;
    const uint32_t ws = P.wpe | 1u;  // LDS row stride of an end's packed words (odd: lanes hit different banks)
    constexpr uint32_t LS = WLC + 1u;  // ... of its list row
    uint32_t *s_words = vs_wlds;                   // [WTPB * ws + 8]
    uint32_t *s_list = s_words + WTPB * ws + 8u;   // [WTPB * LS] node | accepted << 31 per visit
    uint32_t *s_cnt = s_list + WTPB * LS;          // [WTPB] accepted nodes of the end
    uint32_t *s_gwoff = s_cnt + WTPB;              // [WTPB] first word of the end in the block's words
    uint32_t *s_nwords = s_gwoff + WTPB;           // [WTPB] its packed words
    uint32_t *s_misc = s_nwords + WTPB;            // [4] pair classes of this workgroup
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
        s_gwoff[tid] = gwoff;
        s_nwords[tid] = tid < ne ? (rlen + 15u) >> 4 : 0u;
        if (tid < 8u) s_words[WTPB * ws + tid] = 0u;
        __syncthreads();
        // packed reads, coalesced: consecutive lanes fetch consecutive words of a read (row words past the read's own
        // are zeroed: windows may read beyond its end, and what they find there is masked off)
        for (uint32_t i = tid; i < WTPB * ws; i += WTPB) {
            const uint32_t e = vs_fastdiv(i, P.magic_ws), k2 = i - e * ws;
            s_words[i] = k2 < s_nwords[e] ? P.rd.words[s_gwoff[e] + k2] : 0u;
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
        __syncthreads();
        uint32_t nt = 0u;  // visits written to the row (may exceed WLC: overflow)
        if (state == 1u) {
            bool over = false;
            const bool dirty = has_inv && ((meta >> 24) & VS_FLAG_INVALID);
            nt = vs_walk_end<NW>(P.wk, text, K, s_words, tid * ws * 16u, rlen, inv, dirty, s_list + tid * LS, WLC, &over);
            if (over) state |= 2u;
        }
        // ---- accepted nodes to the front of the row; overflowed pairs to the overflow list
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
        s_cnt[tid] = (state & 1u) && tid < ne ? cnt : 0u;
        __syncthreads();
        // ---- rows out, coalesced: WLC words per end, tile order (= what k_pe_accumulate reads)
        if (P.accumulate) {
            uint32_t *ol = P.out_lists + tile * (uint64_t)WTPB * WLC;
            for (uint32_t i = tid; i < ne * WLC; i += WTPB) ol[i] = s_list[(i / WLC) * LS + (i % WLC)];
            if (tid < ne) P.out_counts[tile * WTPB + tid] = s_cnt[tid];
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
    return sizeof(uint32_t) * ((size_t)WTPB * ws + 8u + (size_t)WTPB * (WLC + 1u) + 3u * WTPB + 4u);
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

// ---------------------------------------------------------------------------------------------------------------------
// Probe-rate gate for a one-lane-per-WINDOW mapping kernel (VERDICT r3 "next" #2a; timing only, no result is kept).
// Such a kernel would look up every (k+1)-window of every read end in the (k+1)-mer table above: 2 x 95 x 1e7 = 1.9e9
// scattered 16-byte slot loads per launch at configs[2].  This kernel does just that and nothing else:
//   mode 0: one slot load per lane at a slot derived from (end, window) by an integer mix -- the bare rate of scattered
//           dwordx4 loads out of an L2-resident table;
//   mode 1: the real addresses -- the packed reads of a tile of 64 ends staged in LDS, the window at read offset j formed
//           by funnel shifts and hashed as k_pe_walk hashes it, its probe chain followed to the tag match or the empty slot.
// One lane per window: item i of a tile is (end i / P, window i % P), P = rlen_max - K + 1.
// ---------------------------------------------------------------------------------------------------------------------
#define GATE_EPT 64u
template <uint32_t NW, int MODE>
__global__ void __launch_bounds__(256) k_probe_gate(VsWalkDev wk, VsReadsDev rd, const uint32_t *__restrict__ perm, uint32_t K, uint32_t P,
                                                    uint32_t wpe, uint64_t n_ends, uint64_t n_tiles, uint32_t tiles_per_wg,
                                                    unsigned long long *sink) {
    __shared__ uint32_t s_words[GATE_EPT * 20u + 8u];
    __shared__ uint32_t s_gwoff[GATE_EPT], s_len[GATE_EPT];
    const uint32_t tid = threadIdx.x;
    const uint32_t ws = wpe | 1u;
    const uint32_t kmask = (1u << wk.k_bits) - 1u;
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint64_t tile_lo = (uint64_t)wg * tiles_per_wg;
    const uint64_t tile_hi = tile_lo + tiles_per_wg < n_tiles ? tile_lo + tiles_per_wg : n_tiles;
    uint32_t acc = 0u;
    const uint32_t magic_p = (uint32_t)(0x100000000ull / P) + 1u;
    for (uint64_t tile = tile_lo; tile < tile_hi; tile++) {
        const uint64_t e0 = tile * GATE_EPT;
        const uint32_t ne = (uint32_t)((n_ends - e0) < GATE_EPT ? (n_ends - e0) : GATE_EPT);
        __syncthreads();
        if (tid < GATE_EPT) {
            uint32_t gw = 0u, len = 0u;
            if (tid < ne) {
                const uint64_t e = e0 + tid;
                const uint32_t pair = perm ? perm[e >> 1] : (uint32_t)(e >> 1);
                const uint32_t gend = 2u * pair + (uint32_t)(e & 1u);
                gw = rd.woff[gend];
                len = rd.meta[gend] & VS_LEN_MASK;
            }
            s_gwoff[tid] = gw;
            s_len[tid] = len;
        }
        if (tid < 8u) s_words[GATE_EPT * ws + tid] = 0u;
        __syncthreads();
        if (MODE == 1) {
            for (uint32_t i = tid; i < GATE_EPT * ws; i += 256u) {
                const uint32_t e = i / ws, k2 = i - e * ws;
                s_words[i] = k2 < ((s_len[e] + 15u) >> 4) ? rd.words[s_gwoff[e] + k2] : 0u;
            }
            __syncthreads();
        }
        const uint32_t items = ne * P;
        for (uint32_t it = tid; it < items; it += 256u) {
            const uint32_t e = __umulhi(it, magic_p), j = it - e * P;
            if (MODE == 0) {
                const uint64_t h = vs_mix64(((e0 + e) << 8) | j);
                const VsKSlot sl = wk.ktab[(uint32_t)h & kmask];
                acc ^= sl.pos + sl.ns + sl.tag + sl.woff;
            } else {
                if (j + K > s_len[e]) continue;
                uint64_t h = vs_kmer_hash_init(K);
#pragma unroll
                for (uint32_t i = 0; i < NW; i++) {
                    uint64_t w = vsw_win(s_words, e * ws * 16u + j + 32u * i);
                    if (i == NW - 1u) w &= vsw_lowmask(2u * K - 64u * (NW - 1u));
                    h = vs_kmer_hash_step(h, w);
                }
                h = vs_kmer_hash_done(h);
                const uint32_t tag = (uint32_t)(h >> 32);
                uint32_t s = (uint32_t)h & kmask;
                for (uint32_t tries = 0; tries < 16u; tries++) {
                    const VsKSlot sl = wk.ktab[s];
                    if (sl.ns == VS_WALK_EMPTY) break;
                    if (sl.tag == tag) { acc ^= sl.pos + sl.ns + sl.woff; break; }
                    s = (s + 1u) & kmask;
                }
            }
        }
    }
    if (acc == 0x9E3779B9u) atomicAdd(sink, 1ull);  // (keeps the loads alive)
}

extern "C" int vs_exp_probe_gate(vs_ctx *ctx, const vs_reads *reads, int mode, int use_perm, int reps, double *ms_out,
                                 uint64_t *probes_out) {
    if (!ctx || !reads || !ms_out) return vs_fail(ctx, VS_E_ARG, "vs_exp_probe_gate: bad argument");
    if (!ctx->has_index || !ctx->walk_ok) return vs_fail(ctx, VS_E_STATE, "vs_exp_probe_gate: no walk index (VS_EXPERIMENT=1 VS_WALK=1 and a certified graph)");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t K = ctx->idx.K;
    if (reads->max_len < K) return vs_fail(ctx, VS_E_ARG, "vs_exp_probe_gate: reads shorter than k + 1");
    const uint32_t P = (uint32_t)reads->max_len - K + 1u;
    const uint32_t wpe = (uint32_t)((reads->max_len + 15u) / 16u) + 1u;
    if ((wpe | 1u) > 20u) return vs_fail(ctx, VS_E_RANGE, "vs_exp_probe_gate: reads longer than the gate's LDS rows");
    const uint64_t n_ends = reads->n_ends, n_tiles = (n_ends + GATE_EPT - 1u) / GATE_EPT;
    uint32_t grid = (uint32_t)ctx->n_cu * 32u;
    if (grid > n_tiles) grid = (uint32_t)(n_tiles ? n_tiles : 1u);
    grid &= ~7u;
    if (!grid) grid = 8u;
    const uint32_t tpw = (uint32_t)((n_tiles + grid - 1u) / grid);
    unsigned long long *sink = nullptr;
    VS_HIP(ctx, hipMalloc((void **)&sink, 8));
    VS_HIP(ctx, hipMemsetAsync(sink, 0, 8, ctx->stream));
    const uint32_t *perm = use_perm && ctx->d_perm && ctx->locus_cap >= n_ends / 2u ? (const uint32_t *)ctx->d_perm : nullptr;
    hipEvent_t a, b;
    VS_HIP(ctx, hipEventCreate(&a));
    VS_HIP(ctx, hipEventCreate(&b));
    const VsReadsDev rd = reads->dev();
    auto launch = [&]() {
#define GATE_CASE(NWV)                                                                                                        \
    if (mode == 0) hipLaunchKernelGGL((k_probe_gate<NWV, 0>), dim3(grid), dim3(256), 0, ctx->stream, ctx->walk, rd, perm, K, P, wpe, n_ends, n_tiles, tpw, sink); \
    else hipLaunchKernelGGL((k_probe_gate<NWV, 1>), dim3(grid), dim3(256), 0, ctx->stream, ctx->walk, rd, perm, K, P, wpe, n_ends, n_tiles, tpw, sink);
        switch (ctx->walk.nw) {
            case 1: GATE_CASE(1) break;
            case 2: GATE_CASE(2) break;
            case 3: GATE_CASE(3) break;
            case 4: GATE_CASE(4) break;
            default: GATE_CASE(5) break;
        }
#undef GATE_CASE
    };
    launch();  // warm-up
    VS_HIP(ctx, hipEventRecord(a, ctx->stream));
    for (int r = 0; r < (reps > 0 ? reps : 1); r++) launch();
    VS_HIP(ctx, hipEventRecord(b, ctx->stream));
    VS_HIP(ctx, hipEventSynchronize(b));
    VS_HIP(ctx, hipGetLastError());
    float ms = 0.f;
    VS_HIP(ctx, hipEventElapsedTime(&ms, a, b));
    *ms_out = (double)ms / (reps > 0 ? reps : 1);
    if (probes_out) *probes_out = n_ends * (uint64_t)P;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(sink);
    return VS_OK;
}
