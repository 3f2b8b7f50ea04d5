// K1: node packing and the seed table.
//
// The reference keeps a dict of every (k+1)-mer window of every node under both strands
// (utils/VStrains_PE_Inference.py:116-135) and looks up every window of every read (:24-31).
// The device index answers the same question -- "which (node, offset) windows equal this read
// window, on which strand" -- from something much smaller: every maximal exact match of
// length >= K = k+1 between a read and a node strand contains a w-mer (w = min(31, K) made odd)
// that starts at a read position divisible by s = K - w + 1.  So only w-mers ("seeds") of the
// nodes are indexed, under their canonical (min of forward / reverse-complement) value, and
// only every s-th read position probes.  A hit is extended base-exactly against the packed
// node text, which yields the whole run of K-windows on that diagonal at once (vs_pe.hip).
// w odd => no seed equals its own reverse complement, so strand is always well defined.
//
// Build = pack_nodes -> seed_insert (atomicCAS claim + count) -> scan -> seed_fill -> finalize.
#include <vector>

#include "vs_internal.h"

#define TPB 256

// One thread per node word: 16 forward bases and the matching 16 reverse-complement bases.
__global__ void __launch_bounds__(TPB)
k_pack_nodes(const uint8_t *__restrict__ ascii, const uint64_t *__restrict__ aoff,
             const uint32_t *__restrict__ woff, uint32_t n_nodes, uint32_t total_words, uint32_t K,
             uint32_t *__restrict__ fwd, uint32_t *__restrict__ rc, uint32_t *__restrict__ bad_node) {
    uint32_t wi = blockIdx.x * TPB + threadIdx.x;
    if (wi >= total_words) return;
    uint32_t node = vs_upper_idx(woff, n_nodes + 1, wi);
    uint64_t a = aoff[node];
    uint32_t len = (uint32_t)(aoff[node + 1] - a);
    uint32_t b0 = (wi - woff[node]) * 16u;
    uint32_t f = 0, r = 0;
    bool bad = false;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        uint32_t p = b0 + i;
        if (p < len) {
            uint32_t cf = vs_code(ascii[a + p]);
            uint32_t cr = vs_code(ascii[a + (len - 1 - p)]);
            bad |= (cf > 3u);
            f |= (cf & 3u) << (2 * i);
            r |= ((cr ^ 3u) & 3u) << (2 * i);
        }
    }
    fwd[wi] = f;
    rc[wi] = r;
    if (bad && len >= K) atomicMin(bad_node, node);
}

// One thread per seed position: claim / find the slot of its canonical seed, count it, remember
// the slot for the fill pass.
__global__ void __launch_bounds__(TPB)
k_seed_insert(VsIndexDev idx, const uint64_t *__restrict__ seed_off, uint64_t n_pos,
              unsigned long long *__restrict__ keys, uint32_t *__restrict__ cnts,
              uint32_t *__restrict__ pos_slot, uint32_t *__restrict__ n_claimed) {
    uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (g >= n_pos) return;
    uint32_t node = vs_upper_idx(seed_off, idx.n_nodes + 1, g);
    uint32_t p = (uint32_t)(g - seed_off[node]);
    VsNodeMeta m = idx.meta[node];
    uint32_t strand;
    uint64_t key = vs_seed_key(idx.fwd_words, (uint64_t)m.woff * 16u + p, idx.w, &strand);
    uint32_t mask = (1u << idx.table_bits) - 1u;
    uint32_t sl = vs_slot_of(key, idx.table_bits);
    for (;;) {
        unsigned long long old = atomicCAS(&keys[sl], (unsigned long long)VS_EMPTY_KEY, (unsigned long long)key);
        if (old == VS_EMPTY_KEY && n_claimed) atomicAdd(n_claimed, 1u);
        if (old == VS_EMPTY_KEY || old == key) break;
        sl = (sl + 1u) & mask;
    }
    if (cnts) atomicAdd(&cnts[sl], 1u);
    if (pos_slot) pos_slot[g] = sl;
}

__global__ void __launch_bounds__(TPB)
k_seed_fill(VsIndexDev idx, const uint64_t *__restrict__ seed_off, uint64_t n_pos,
            const uint32_t *__restrict__ pos_slot, const uint32_t *__restrict__ offs,
            uint32_t *__restrict__ cursor, uint4 *__restrict__ postings) {
    uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (g >= n_pos) return;
    uint32_t node = vs_upper_idx(seed_off, idx.n_nodes + 1, g);
    uint32_t p = (uint32_t)(g - seed_off[node]);
    VsNodeMeta m = idx.meta[node];
    uint32_t strand;  // 1: the stored key is the reverse complement of the node text
    (void)vs_seed_key(idx.fwd_words, (uint64_t)m.woff * 16u + p, idx.w, &strand);
    uint32_t sl = pos_slot[g];
    uint32_t at = offs[sl] + atomicAdd(&cursor[sl], 1u);
    VsPosting rec;  // (carries the node header: the mapping kernel needs no second load for it)
    rec.node = node; rec.pos = p; rec.strand = strand; rec.len = m.len; rec.woff = m.woff;
    postings[at] = vs_posting_pack(rec);
}

__global__ void __launch_bounds__(TPB)
k_table_finalize(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ cnts,
                 const uint32_t *__restrict__ offs, const uint4 *__restrict__ postings,
                 uint32_t n_slots, VsSlot *__restrict__ table, uint32_t *__restrict__ n_distinct) {
    uint32_t sl = blockIdx.x * TPB + threadIdx.x;
    if (sl >= n_slots) return;
    VsSlot out;
    uint64_t key = keys[sl];
    if (key == VS_EMPTY_KEY) {
        out.key = VS_EMPTY_KEY; out.a = 0; out.b = 0;
    } else {
        uint32_t c = cnts[sl];
        if (c == 1u) {
            const uint4 p = postings[offs[sl]];
            out.key = key; out.a = p.x; out.b = p.y;
        } else {
            out.key = key | VS_MULTI_BIT; out.a = offs[sl]; out.b = c;
        }
        atomicAdd(n_distinct, 1u);
    }
    table[sl] = out;
}

static void seed_geometry(uint32_t K, uint32_t *w, uint32_t *s) {
    uint32_t ww = K < 31u ? K : 31u;
    // Long overlaps (k >= 95) put a 31-mer into every node that shares the overlap it lies in: 15.7 postings per seed
    // at configs[3] (k = 127).  A 63-mer lies in a third fewer nodes and a 2 x 250 end still needs three probes
    // (stride K - 62); its key is a mix of the 126 bits, the comparison covers the seed's own bases (vs_seed_key).
    if (K >= 96u) ww = 63u;
    if ((ww & 1u) == 0u) ww -= 1u;  // K >= 2 here, so ww >= 1
    *w = ww;
    *s = K - ww + 1u;
}

extern "C" int vs_index_build(vs_ctx *ctx, const uint8_t *node_ascii, const uint64_t *node_off,
                              uint32_t n_nodes, uint32_t ksize, uint32_t *bad_node_out, uint8_t *bad_char_out) {
    if (!ctx) return VS_E_ARG;
    if (!node_off || (!node_ascii && n_nodes && node_off[n_nodes])) return vs_fail(ctx, VS_E_ARG, "vs_index_build: NULL input");
    if (ksize < 1 || ksize > 0x00FFFFF0u) return vs_fail(ctx, VS_E_ARG, "vs_index_build: kmer size %u unsupported", ksize);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    vs_ctx_free_index(ctx);

    const uint32_t K = ksize + 1;
    uint32_t w, s;
    seed_geometry(K, &w, &s);

    // host-side prefix arrays (N is small next to the reads)
    std::vector<uint32_t> woff(n_nodes + 1);
    std::vector<uint64_t> seed_off(n_nodes + 1);
    std::vector<VsNodeMeta> meta(n_nodes ? n_nodes : 1);
    uint64_t words = 0, npos = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        uint64_t len = node_off[i + 1] - node_off[i];
        if (len > VS_LEN_MASK) return vs_fail(ctx, VS_E_RANGE, "node %u is %llu bases long (limit %u)", i, (unsigned long long)len, VS_LEN_MASK);
        woff[i] = (uint32_t)words;
        seed_off[i] = npos;
        meta[i].woff = (uint32_t)words;
        meta[i].len = (uint32_t)len;
        words += (len + 15) / 16;
        if (len >= K) npos += len - w + 1;
        if (words > 0xFFFFFFF0ull) return vs_fail(ctx, VS_E_RANGE, "node text too large");
    }
    woff[n_nodes] = (uint32_t)words;
    seed_off[n_nodes] = npos;
    if (npos > 0x3FFFFFF0ull) return vs_fail(ctx, VS_E_RANGE, "too many seed positions (%llu)", (unsigned long long)npos);
    if (2 * (words + VS_PAD_WORDS) >= (1ull << 28)) return vs_fail(ctx, VS_E_RANGE, "node text of %llu packed words (both strands) exceeds 2^32 bases", (unsigned long long)words);
    uint32_t bits = 4;
    while ((1ull << bits) < 2 * npos + 2) bits++;
    const uint64_t n_slots = 1ull << bits;
    const uint64_t total_ascii = node_off[n_nodes];

    // device buffers: permanent
    size_t b_meta = sizeof(VsNodeMeta) * (n_nodes ? n_nodes : 1);
    size_t b_words = sizeof(uint32_t) * (words + VS_PAD_WORDS);
    size_t b_post = sizeof(uint4) * (npos ? npos : 1);
    VS_HIP(ctx, hipMalloc(&ctx->d_meta, b_meta));
    VS_HIP(ctx, hipMalloc(&ctx->d_fwd, 2 * b_words));  // forward text, then the reverse complements
    ctx->d_rc = nullptr;
    VS_HIP(ctx, hipMalloc(&ctx->d_post, b_post));
    if (ctx->experiment_level) vs_tuning_load(ctx->tune, ctx->experiment_level);
    ctx->index_bytes = b_meta + 2 * b_words + b_post;  // (+ the table, sized below)
    // temporaries
    uint8_t *d_ascii = nullptr;
    uint64_t *d_aoff = nullptr, *d_seed_off = nullptr, *d_tmp = nullptr;
    uint32_t *d_woff = nullptr, *d_cnts = nullptr, *d_offs = nullptr, *d_cursor = nullptr, *d_pos_slot = nullptr, *d_flags = nullptr;
    unsigned long long *d_keys = nullptr;
    int rc = VS_OK;
    hipStream_t st = ctx->stream;
#define TRY(call)                                                                                  \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            rc = vs_fail(ctx, e__ == hipErrorOutOfMemory ? VS_E_OOM : VS_E_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
            goto done;                                                                             \
        }                                                                                          \
    } while (0)
    {
        TRY(hipMalloc((void **)&d_ascii, total_ascii + 16));
        TRY(hipMalloc((void **)&d_aoff, sizeof(uint64_t) * (n_nodes + 1)));
        TRY(hipMalloc((void **)&d_seed_off, sizeof(uint64_t) * (n_nodes + 1)));
        TRY(hipMalloc((void **)&d_woff, sizeof(uint32_t) * (n_nodes + 1)));
        TRY(hipMalloc((void **)&d_keys, sizeof(unsigned long long) * n_slots));
        TRY(hipMalloc((void **)&d_pos_slot, sizeof(uint32_t) * (npos ? npos : 1)));
        TRY(hipMalloc((void **)&d_flags, sizeof(uint32_t) * 4));
        if (total_ascii) TRY(hipMemcpyAsync(d_ascii, node_ascii, total_ascii, hipMemcpyHostToDevice, st));
        TRY(hipMemcpyAsync(d_aoff, node_off, sizeof(uint64_t) * (n_nodes + 1), hipMemcpyHostToDevice, st));
        TRY(hipMemcpyAsync(d_seed_off, seed_off.data(), sizeof(uint64_t) * (n_nodes + 1), hipMemcpyHostToDevice, st));
        TRY(hipMemcpyAsync(d_woff, woff.data(), sizeof(uint32_t) * (n_nodes + 1), hipMemcpyHostToDevice, st));
        TRY(hipMemcpyAsync(ctx->d_meta, meta.data(), b_meta, hipMemcpyHostToDevice, st));
        TRY(hipMemsetAsync(ctx->d_fwd, 0, 2 * b_words, st));
        TRY(hipMemsetAsync(d_keys, 0xFF, sizeof(unsigned long long) * n_slots, st));
        uint32_t init_flags[4] = {0xFFFFFFFFu, 0, 0, 0};
        TRY(hipMemcpyAsync(d_flags, init_flags, sizeof init_flags, hipMemcpyHostToDevice, st));

        VsIndexDev d{};
        d.n_nodes = n_nodes; d.K = K; d.w = w; d.s = s; d.table_bits = bits;
        d.meta = (const VsNodeMeta *)ctx->d_meta;
        d.fwd_words = (const uint32_t *)ctx->d_fwd;
        d.rc_delta = (uint32_t)(words + VS_PAD_WORDS);
        d.rc_words = d.fwd_words + d.rc_delta;
        d.table = (const VsSlot *)ctx->d_table;
        d.postings = (const uint4 *)ctx->d_post;

        if (words)
            hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)((words + TPB - 1) / TPB)), dim3(TPB), 0, st, d_ascii, d_aoff,
                               d_woff, n_nodes, (uint32_t)words, K, (uint32_t *)ctx->d_fwd, (uint32_t *)ctx->d_fwd + (words + VS_PAD_WORDS), d_flags);
        uint32_t h_flags[4];
        TRY(hipMemcpyAsync(h_flags, d_flags, sizeof h_flags, hipMemcpyDeviceToHost, st));
        TRY(hipStreamSynchronize(st));
        if (h_flags[0] != 0xFFFFFFFFu) {
            // Name the byte the reference's KeyError names: last offending byte of the first
            // (k+1)-window of that node that holds one (reverse_seq scans the window from its end).
            uint32_t nd = h_flags[0];
            const uint8_t *t = node_ascii + node_off[nd];
            uint64_t len = node_off[nd + 1] - node_off[nd], first = 0;
            auto isbad = [](uint8_t c) { return c != 'A' && c != 'C' && c != 'G' && c != 'T'; };
            while (first < len && !isbad(t[first])) first++;
            uint64_t ws = first + 1 > K ? first + 1 - K : 0;
            uint8_t last = t[first];
            for (uint64_t q = ws; q < ws + K && q < len; q++)
                if (isbad(t[q])) last = t[q];
            if (bad_node_out) *bad_node_out = nd;
            if (bad_char_out) *bad_char_out = last;
            rc = vs_fail(ctx, VS_E_NODE_BASE, "node %u holds byte 0x%02x outside ACGT", nd, last);
            goto done;
        }
        // The table is sized by the DISTINCT seeds, which only an insertion pass can count: a graph whose nodes overlap
        // by k bases holds every seed of an overlap once per node that shares it (3.6 positions per distinct seed at
        // configs[2], 8.9 at configs[4]), and a table sized by positions is 5-12 % full -- every occupied 16-byte slot
        // alone in its 128-byte line, 8 MB where 2 MB do, and the probes of seeds that are in no node (every read
        // error makes some) land anywhere in it.  First pass: claim slots in the scratch table only to count.
        uint64_t n_slots_final = n_slots;
        uint32_t bits_final = bits;
        if (npos) {
            unsigned nb = (unsigned)((npos + TPB - 1) / TPB);
            hipLaunchKernelGGL(k_seed_insert, dim3(nb), dim3(TPB), 0, st, d, d_seed_off, npos, d_keys, (uint32_t *)nullptr,
                               (uint32_t *)nullptr, d_flags + 3);
            TRY(hipMemcpyAsync(h_flags, d_flags, sizeof h_flags, hipMemcpyDeviceToHost, st));
            TRY(hipStreamSynchronize(st));
            const uint64_t want = ((uint64_t)h_flags[3] << ctx->tune.table_shift) + 2u;
            bits_final = 4;
            while ((1ull << bits_final) < want) bits_final++;
            if (bits_final > 30u) bits_final = 30u;
            // (the cap, or a table the device has no room for: fewer slots per seed -- longer probe chains, the same
            // answers -- down to three quarters full; beyond that the build fails and says why)
            const uint64_t distinct = h_flags[3];
            uint32_t gave_up_bits = 0;
            if (bits_final > bits) {
                (void)hipFree(d_keys);
                d_keys = nullptr;
                while (hipMalloc((void **)&d_keys, sizeof(unsigned long long) << bits_final) != hipSuccess) {
                    (void)hipGetLastError();
                    d_keys = nullptr;
                    if (bits_final <= bits + 1u || (3ull << (bits_final - 1u)) / 4u < distinct) break;
                    bits_final--;
                }
                if (!d_keys) {
                    gave_up_bits = bits_final;  // (the smallest table that was asked for and refused)
                    bits_final = bits;
                    TRY(hipMalloc((void **)&d_keys, sizeof(unsigned long long) << bits_final));
                }
            }
            if ((3ull << bits_final) / 4u < distinct) {
                //  two ways to get here: the 2^30-slot cap (a graph beyond what this build supports), or a device
                // without room for the table the graph needs -- the caller must be able to tell them apart
                if (gave_up_bits)
                    rc = vs_fail(ctx, VS_E_OOM, "vs_index_build: no device memory for a seed table of 2^%u slots (%llu bytes of keys) and the %llu distinct seeds do not fit 2^%u",
                                 gave_up_bits, (unsigned long long)(sizeof(unsigned long long) << gave_up_bits), (unsigned long long)distinct, bits_final);
                else
                    rc = vs_fail(ctx, VS_E_RANGE, "vs_index_build: %llu distinct seeds do not fit a seed table of 2^%u slots", (unsigned long long)distinct, bits_final);
                goto done;
            }
            n_slots_final = 1ull << bits_final;
            d.table_bits = bits_final;
            TRY(hipMemsetAsync(d_keys, 0xFF, sizeof(unsigned long long) * n_slots_final, st));
        }
        TRY(hipMalloc((void **)&d_cnts, sizeof(uint32_t) * n_slots_final));
        TRY(hipMalloc((void **)&d_offs, sizeof(uint32_t) * n_slots_final));
        TRY(hipMalloc((void **)&d_cursor, sizeof(uint32_t) * n_slots_final));
        TRY(hipMalloc((void **)&d_tmp, sizeof(uint64_t) * (n_slots_final / 2048 + 4)));
        TRY(hipMemsetAsync(d_cnts, 0, sizeof(uint32_t) * n_slots_final, st));
        TRY(hipMemsetAsync(d_cursor, 0, sizeof(uint32_t) * n_slots_final, st));
        TRY(hipMalloc(&ctx->d_table, sizeof(VsSlot) * n_slots_final));
        d.table = (const VsSlot *)ctx->d_table;
        ctx->index_bytes += sizeof(VsSlot) * n_slots_final;
        if (npos) {
            unsigned nb = (unsigned)((npos + TPB - 1) / TPB);
            hipLaunchKernelGGL(k_seed_insert, dim3(nb), dim3(TPB), 0, st, d, d_seed_off, npos, d_keys, d_cnts, d_pos_slot,
                               (uint32_t *)nullptr);
            rc = vs_scan_u32(ctx, d_cnts, d_offs, n_slots_final, d_tmp, nullptr);
            if (rc) goto done;
            hipLaunchKernelGGL(k_seed_fill, dim3(nb), dim3(TPB), 0, st, d, d_seed_off, npos, d_pos_slot, d_offs, d_cursor,
                               (uint4 *)ctx->d_post);
        }
        hipLaunchKernelGGL(k_table_finalize, dim3((unsigned)((n_slots_final + TPB - 1) / TPB)), dim3(TPB), 0, st, d_keys, d_cnts,
                           d_offs, (const uint4 *)ctx->d_post, (uint32_t)n_slots_final, (VsSlot *)ctx->d_table, d_flags + 1);
        TRY(hipGetLastError());
        TRY(hipMemcpyAsync(h_flags, d_flags, sizeof h_flags, hipMemcpyDeviceToHost, st));
        TRY(hipStreamSynchronize(st));
        ctx->idx = d;
        ctx->n_seed_pos = npos;
        ctx->max_node_len = 0;
        for (uint32_t i = 0; i < n_nodes; i++) ctx->max_node_len = meta[i].len > ctx->max_node_len ? meta[i].len : ctx->max_node_len;
        ctx->n_slots = n_slots_final;
        ctx->n_distinct = h_flags[1];
        ctx->has_index = true;
    }
done:
    (void)hipStreamSynchronize(st);
    {
        void *tmps[] = {d_ascii, d_aoff, d_seed_off, d_woff, d_keys, d_cnts, d_offs, d_cursor, d_pos_slot, d_tmp, d_flags};
        for (void *p : tmps)
            if (p) (void)hipFree(p);
    }
    if (rc != VS_OK) vs_ctx_free_index(ctx);
    return rc;
#undef TRY
}

extern "C" int vs_index_info(const vs_ctx *ctx, uint64_t info[6]) {
    if (!ctx || !info) return VS_E_ARG;
    if (!ctx->has_index) return VS_E_STATE;
    info[0] = ctx->idx.w;
    info[1] = ctx->idx.s;
    info[2] = ctx->n_seed_pos;
    info[3] = ctx->n_slots;
    info[4] = ctx->n_distinct;
    info[5] = ctx->index_bytes;
    return VS_OK;
}
