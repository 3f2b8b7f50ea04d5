// Read blocks: ASCII -> 2-bit packing on the device, unpacking (test aid) and the on-device
// synthetic pair generator used by bench.py.
//
// Layout (see vs_internal.h): ends interleaved (2r = forward, 2r+1 = reverse), every end starts
// on a uint32 word, 16 bases per word, LSB first; meta[e] = length | flags << 24.
#include <vector>

#include "vs_internal.h"

#define TPB 256

// one thread per packed word
__global__ void __launch_bounds__(TPB)
k_pack_reads(const uint8_t *__restrict__ ascii, const uint64_t *__restrict__ aoff,
             const uint32_t *__restrict__ woff, uint64_t n_ends, uint32_t total_words,
             uint32_t *__restrict__ words, uint32_t *__restrict__ mask, uint32_t *__restrict__ meta) {
    uint32_t wi = blockIdx.x * TPB + threadIdx.x;
    if (wi >= total_words) return;
    uint32_t e = vs_upper_idx(woff, (uint32_t)n_ends + 1u, wi);
    uint64_t a = aoff[e];
    uint32_t len = (uint32_t)(aoff[e + 1] - a);
    uint32_t b0 = (wi - woff[e]) * 16u;
    uint32_t v = 0, m = 0, fl = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        uint32_t p = b0 + i;
        if (p < len) {
            uint8_t c = ascii[a + p];
            uint32_t code = vs_code(c);
            if (code > 3u) {
                fl |= (c == 'N') ? VS_FLAG_N : VS_FLAG_INVALID;
                m |= 3u << (2 * i);
            }
            v |= (code & 3u) << (2 * i);
        }
    }
    words[wi] = v;
    if (mask) mask[wi] = m;
    if (fl) atomicOr(&meta[e], fl << 24);
}

__global__ void __launch_bounds__(TPB)
k_count_invalid(const uint32_t *__restrict__ meta, uint64_t n_ends, uint32_t *__restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    bool inv = e < n_ends && ((meta[e] >> 24) & VS_FLAG_INVALID);
    unsigned long long b = __ballot(inv);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (uint32_t)__popcll(b));
}

// One thread per end: the positions of its bytes outside ACGT, read off the mask -- up to four of
// them, one byte each (0xFF = none), for vs_seed_limits in the straight-line mapping kernels; an end
// with more (or with one beyond position 254) is flagged VS_FLAG_MANY and takes the overflow path.
__global__ void __launch_bounds__(TPB)
k_inv4(const uint32_t *__restrict__ woff, const uint32_t *__restrict__ mask, uint64_t n_ends, uint32_t *__restrict__ meta,
       uint32_t *__restrict__ inv4) {
    const uint64_t e = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= n_ends) return;
    const uint32_t m0 = meta[e];
    uint32_t out = 0xFFFFFFFFu;
    if ((m0 >> 24) & VS_FLAG_INVALID) {
        const uint32_t len = m0 & VS_LEN_MASK, nw = (len + 15u) >> 4;
        const uint32_t *mw = mask + woff[e];
        uint32_t count = 0;
        bool many = false;
        for (uint32_t wi = 0; wi < nw && !many; wi++) {
            uint32_t m = mw[wi];
            while (m) {
                const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
                const uint32_t pos = wi * 16u + (bit >> 1);
                m &= ~(3u << (bit & ~1u));
                if (pos >= len) continue;
                if (count < 4u && pos < 255u) out = (out & ~(0xFFu << (8u * count))) | (pos << (8u * count));
                else many = true;
                count++;
            }
        }
        if (many) meta[e] = m0 | (VS_FLAG_MANY << 24);
    }
    inv4[e] = out;
}

__global__ void __launch_bounds__(TPB)
k_unpack_reads(VsReadsDev rd, const uint64_t *__restrict__ out_off, uint8_t *__restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= rd.n_ends) return;
    uint32_t len = rd.meta[e] & VS_LEN_MASK;
    const uint32_t *w = rd.words + rd.woff[e];
    uint8_t *o = out + out_off[e];
    for (uint32_t i = 0; i < len; i++) o[i] = "ACGT"[(w[i >> 4] >> ((i & 15u) * 2u)) & 3u];
}

static int alloc_reads(vs_ctx *ctx, vs_reads *r, bool with_mask) {
    size_t b_woff = sizeof(uint32_t) * (r->n_ends + 1);
    size_t b_meta = sizeof(uint32_t) * (r->n_ends ? r->n_ends : 1);
    size_t b_words = sizeof(uint32_t) * (r->n_words + VS_PAD_WORDS);
    VS_HIP(ctx, hipMalloc(&r->d_woff, b_woff));
    VS_HIP(ctx, hipMalloc(&r->d_meta, b_meta));
    VS_HIP(ctx, hipMalloc(&r->d_words, b_words));
    VS_HIP(ctx, hipMemsetAsync((char *)r->d_words + sizeof(uint32_t) * r->n_words, 0, VS_PAD_WORDS * sizeof(uint32_t), ctx->stream));
    r->bytes = b_woff + b_meta + b_words;
    if (with_mask) {
        VS_HIP(ctx, hipMalloc(&r->d_mask, b_words));
        VS_HIP(ctx, hipMemsetAsync((char *)r->d_mask + sizeof(uint32_t) * r->n_words, 0, VS_PAD_WORDS * sizeof(uint32_t), ctx->stream));
        r->bytes += b_words;
    }
    return VS_OK;
}

extern "C" void vs_reads_free(vs_ctx *ctx, vs_reads *r) {
    if (!r) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    void *ps[] = {r->d_woff, r->d_meta, r->d_words, r->d_mask, r->d_inv4};
    for (void *p : ps) {
        if (!p) continue;
        if (r->cached && ctx) vs_cache_release(ctx, p);
        else (void)hipFree(p);
    }
    delete r;
}

extern "C" int vs_reads_info(const vs_reads *r, uint64_t info[5]) {
    if (!r || !info) return VS_E_ARG;
    info[0] = r->n_ends; info[1] = r->n_words; info[2] = r->max_len; info[3] = r->n_invalid; info[4] = r->bytes;
    return VS_OK;
}

extern "C" int vs_reads_pack(vs_ctx *ctx, const uint8_t *ascii, const uint64_t *off, uint64_t n_ends, vs_reads **out) {
    if (!ctx || !off || !out) return VS_E_ARG;
    *out = nullptr;
    if (n_ends & 1ull) return vs_fail(ctx, VS_E_ARG, "vs_reads_pack: ends come in pairs (got %llu)", (unsigned long long)n_ends);
    if (n_ends > 0xFFFFFFF0ull) return vs_fail(ctx, VS_E_RANGE, "vs_reads_pack: split the input into blocks of < 2^32 ends");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    vs_reads *r = new vs_reads();
    r->n_ends = n_ends;
    std::vector<uint32_t> woff(n_ends + 1), meta(n_ends ? n_ends : 1);
    uint64_t words = 0, maxlen = 0;
    for (uint64_t e = 0; e < n_ends; e++) {
        uint64_t len = off[e + 1] - off[e];
        if (len > VS_LEN_MASK) { delete r; return vs_fail(ctx, VS_E_RANGE, "read end %llu is %llu bytes long", (unsigned long long)e, (unsigned long long)len); }
        woff[e] = (uint32_t)words;
        meta[e] = (uint32_t)len;
        words += (len + 15) / 16;
        if (len > maxlen) maxlen = len;
        if (words > 0xFFFFFFF0ull) { delete r; return vs_fail(ctx, VS_E_RANGE, "vs_reads_pack: block exceeds 2^32 packed words"); }
    }
    woff[n_ends] = (uint32_t)words;
    r->n_words = words;
    r->max_len = maxlen;
    uint8_t *d_ascii = nullptr;
    uint64_t *d_aoff = nullptr;
    uint32_t *d_cnt = nullptr;
    int rc = alloc_reads(ctx, r, false);
    hipStream_t st = ctx->stream;
    uint64_t total = off[n_ends];
    hipError_t e1 = hipSuccess;
    if (rc == VS_OK) {
        do {
            if ((e1 = hipMalloc((void **)&d_ascii, total + 16)) != hipSuccess) break;
            if ((e1 = hipMalloc((void **)&d_aoff, sizeof(uint64_t) * (n_ends + 1))) != hipSuccess) break;
            if ((e1 = hipMalloc((void **)&d_cnt, sizeof(uint32_t))) != hipSuccess) break;
            if (total && (e1 = hipMemcpyAsync(d_ascii, ascii, total, hipMemcpyHostToDevice, st)) != hipSuccess) break;
            if ((e1 = hipMemcpyAsync(d_aoff, off, sizeof(uint64_t) * (n_ends + 1), hipMemcpyHostToDevice, st)) != hipSuccess) break;
            if ((e1 = hipMemcpyAsync(r->d_woff, woff.data(), sizeof(uint32_t) * (n_ends + 1), hipMemcpyHostToDevice, st)) != hipSuccess) break;
            if (n_ends && (e1 = hipMemcpyAsync(r->d_meta, meta.data(), sizeof(uint32_t) * n_ends, hipMemcpyHostToDevice, st)) != hipSuccess) break;
            if ((e1 = hipMemsetAsync(d_cnt, 0, sizeof(uint32_t), st)) != hipSuccess) break;
            unsigned nbw = (unsigned)((words + TPB - 1) / TPB);
            if (words)
                hipLaunchKernelGGL(k_pack_reads, dim3(nbw), dim3(TPB), 0, st, d_ascii, d_aoff, (const uint32_t *)r->d_woff, n_ends,
                                   (uint32_t)words, (uint32_t *)r->d_words, (uint32_t *)nullptr, (uint32_t *)r->d_meta);
            if (n_ends)
                hipLaunchKernelGGL(k_count_invalid, dim3((unsigned)((n_ends + TPB - 1) / TPB)), dim3(TPB), 0, st,
                                   (const uint32_t *)r->d_meta, n_ends, d_cnt);
            uint32_t h_cnt = 0;
            if ((e1 = hipMemcpyAsync(&h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
            if ((e1 = hipStreamSynchronize(st)) != hipSuccess) break;
            r->n_invalid = h_cnt;
            if (h_cnt) {  // rare: some end holds a byte outside ACGTN -> build the validity mask too
                size_t b_words = sizeof(uint32_t) * (words + VS_PAD_WORDS);
                if ((e1 = hipMalloc(&r->d_mask, b_words)) != hipSuccess) break;
                if ((e1 = hipMemsetAsync(r->d_mask, 0, b_words, st)) != hipSuccess) break;
                r->bytes += b_words;
                hipLaunchKernelGGL(k_pack_reads, dim3(nbw), dim3(TPB), 0, st, d_ascii, d_aoff, (const uint32_t *)r->d_woff, n_ends,
                                   (uint32_t)words, (uint32_t *)r->d_words, (uint32_t *)r->d_mask, (uint32_t *)r->d_meta);
                if ((e1 = hipMalloc(&r->d_inv4, sizeof(uint32_t) * n_ends)) != hipSuccess) break;
                r->bytes += sizeof(uint32_t) * n_ends;
                hipLaunchKernelGGL(k_inv4, dim3((unsigned)((n_ends + TPB - 1) / TPB)), dim3(TPB), 0, st, (const uint32_t *)r->d_woff,
                                   (const uint32_t *)r->d_mask, n_ends, (uint32_t *)r->d_meta, (uint32_t *)r->d_inv4);
            }
            if ((e1 = hipGetLastError()) != hipSuccess) break;
            e1 = hipStreamSynchronize(st);
        } while (0);
        if (e1 != hipSuccess) rc = vs_fail(ctx, e1 == hipErrorOutOfMemory ? VS_E_OOM : VS_E_HIP, "vs_reads_pack: %s", hipGetErrorString(e1));
    }
    if (d_ascii) (void)hipFree(d_ascii);
    if (d_aoff) (void)hipFree(d_aoff);
    if (d_cnt) (void)hipFree(d_cnt);
    if (rc != VS_OK) { vs_reads_free(ctx, r); return rc; }
    *out = r;
    return VS_OK;
}

extern "C" int vs_reads_unpack(vs_ctx *ctx, const vs_reads *r, uint8_t *out, uint32_t *lens, uint8_t *flags) {
    if (!ctx || !r || !out) return VS_E_ARG;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> meta(r->n_ends ? r->n_ends : 1);
    if (r->n_ends) VS_HIP(ctx, hipMemcpyAsync(meta.data(), r->d_meta, sizeof(uint32_t) * r->n_ends, hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> ooff(r->n_ends + 1);
    uint64_t tot = 0;
    for (uint64_t e = 0; e < r->n_ends; e++) {
        ooff[e] = tot;
        tot += meta[e] & VS_LEN_MASK;
        if (lens) lens[e] = meta[e] & VS_LEN_MASK;
        if (flags) flags[e] = (uint8_t)(meta[e] >> 24);
    }
    ooff[r->n_ends] = tot;
    if (!tot) return VS_OK;
    uint64_t *d_ooff = nullptr;
    uint8_t *d_out = nullptr;
    VS_HIP(ctx, hipMalloc((void **)&d_ooff, sizeof(uint64_t) * (r->n_ends + 1)));
    hipError_t e1 = hipMalloc((void **)&d_out, tot);
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(d_ooff, ooff.data(), sizeof(uint64_t) * (r->n_ends + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e1 == hipSuccess) {
        hipLaunchKernelGGL(k_unpack_reads, dim3((unsigned)((r->n_ends + TPB - 1) / TPB)), dim3(TPB), 0, ctx->stream, r->dev(), d_ooff, d_out);
        e1 = hipMemcpyAsync(out, d_out, tot, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_ooff);
    if (d_out) (void)hipFree(d_out);
    if (e1 != hipSuccess) return vs_fail(ctx, VS_E_HIP, "vs_reads_unpack: %s", hipGetErrorString(e1));
    return VS_OK;
}

// ---- synthetic pairs ---------------------------------------------------------------------------
// Integer recipe shared with oracle/pe_oracle.c:peo_synth_pairs (the CPU twin).  One thread per
// output word.
struct SynthParams {
    const uint32_t *gwords;  // packed genomes
    const uint64_t *gbase;   // [n_strains] first base (multiple of 16) of each genome in gwords
    const uint64_t *glen;    // [n_strains]
    const uint32_t *cum;     // [n_strains]
    uint32_t n_strains;
    uint64_t seed, first_pair, n_pairs;
    uint32_t read_len, words_per_end, sub_thresh, n_thresh;
};

__device__ __forceinline__ uint32_t gbase_at(const uint32_t *w, uint64_t i) {
    return (w[i >> 4] >> ((uint32_t)(i & 15u) * 2u)) & 3u;
}

__global__ void __launch_bounds__(TPB)
k_synth(SynthParams P, uint32_t *__restrict__ words, uint32_t *__restrict__ meta) {
    uint64_t t = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    uint64_t per_pair = 2ull * P.words_per_end;
    if (t >= P.n_pairs * per_pair) return;
    uint64_t q = t / per_pair;
    uint32_t rem = (uint32_t)(t - q * per_pair);
    uint32_t e = rem / P.words_per_end, wi = rem - e * P.words_per_end;
    uint64_t r = P.first_pair + q;
    uint64_t base = vs_mix64(P.seed * 0xD1342543DE82EF95ull + r);
    uint64_t u0 = vs_mix64(base + 1), u1 = vs_mix64(base + 2), u2 = vs_mix64(base + 3), u3 = vs_mix64(base + 4);
    uint32_t pick = (uint32_t)(u0 >> 32), s = 0;
    while (s + 1 < P.n_strains && pick > P.cum[s]) s++;
    int64_t L = P.read_len, glen = (int64_t)P.glen[s];
    int64_t sum = (int64_t)(u1 & 0xFFFF) + (int64_t)((u1 >> 16) & 0xFFFF) + (int64_t)((u1 >> 32) & 0xFFFF) + (int64_t)((u1 >> 48) & 0xFFFF);
    int64_t flen = 3 * L + (sum - 131070) * (3 * L) / 378372;
    if (flen < L) flen = L;
    if (flen > glen) flen = glen;
    int64_t start = (int64_t)((u2 >> 11) % (uint64_t)(glen - flen + 1));
    uint32_t flip = (uint32_t)(u2 & 1ull);
    // end e reads forward from `start` when (e == flip), else reverse-complemented from the far end
    bool forward = (e == flip);
    uint64_t g0 = P.gbase[s];
    uint32_t n_thr_hit = ((uint32_t)u3 < P.n_thresh) ? 1u : 0u;
    uint32_t n_end = (uint32_t)(u3 >> 62) & 1u;
    uint32_t n_pos = (uint32_t)((u3 >> 32) & 0x3FFFFFFFu) % P.read_len;
    uint32_t v = 0;
    for (uint32_t i = 0; i < 16; i++) {
        int64_t p = (int64_t)wi * 16 + i;
        if (p >= L) break;
        uint32_t b = forward ? gbase_at(P.gwords, g0 + (uint64_t)(start + p))
                             : (gbase_at(P.gwords, g0 + (uint64_t)(start + flen - 1 - p)) ^ 3u);
        if (P.sub_thresh) {
            uint64_t h = vs_mix64(base + 16 + (uint64_t)e * 4096 + (uint64_t)p);
            if ((uint32_t)h < P.sub_thresh) b = (b + 1u + (uint32_t)((h >> 32) % 3ull)) & 3u;
        }
        v |= b << (2 * i);
    }
    uint64_t end_id = 2 * q + e;
    words[end_id * P.words_per_end + wi] = v;
    if (wi == 0) {
        uint32_t fl = (n_thr_hit && n_end == e) ? VS_FLAG_N : 0u;
        (void)n_pos;  // the N's position does not matter once the pair is flagged
        meta[end_id] = P.read_len | (fl << 24);
    }
}

__global__ void __launch_bounds__(TPB) k_iota_woff(uint32_t *woff, uint64_t n, uint32_t stride) {
    uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) woff[i] = (uint32_t)(i * stride);
}

extern "C" int vs_synth_pairs(vs_ctx *ctx, const uint8_t *genomes, const uint64_t *goff, const uint32_t *cum,
                              uint32_t n_strains, uint64_t seed, uint64_t first_pair, uint64_t n_pairs, uint32_t read_len,
                              uint32_t sub_thresh, uint32_t n_thresh, vs_reads **out) {
    if (!ctx || !genomes || !goff || !cum || !out || !n_strains) return VS_E_ARG;
    *out = nullptr;
    if (read_len == 0 || read_len > 4096) return vs_fail(ctx, VS_E_ARG, "vs_synth_pairs: read_len must be 1..4096");
    uint32_t wpe = (read_len + 15) / 16;
    if (2 * n_pairs * wpe > 0xFFFFFFF0ull) return vs_fail(ctx, VS_E_RANGE, "vs_synth_pairs: block exceeds 2^32 packed words");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    // pack the genomes on the host (tiny) into word-aligned 2-bit text
    std::vector<uint64_t> gbase(n_strains), glen(n_strains);
    uint64_t gw = 0;
    for (uint32_t s = 0; s < n_strains; s++) {
        gbase[s] = gw * 16;
        glen[s] = goff[s + 1] - goff[s];
        if (glen[s] < read_len) return vs_fail(ctx, VS_E_ARG, "vs_synth_pairs: genome %u shorter than a read", s);
        gw += (glen[s] + 15) / 16;
    }
    std::vector<uint32_t> gwords(gw + 4, 0u);
    for (uint32_t s = 0; s < n_strains; s++)
        for (uint64_t i = 0; i < glen[s]; i++) {
            uint8_t c = genomes[goff[s] + i];
            uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
            if (code > 3u) return vs_fail(ctx, VS_E_ARG, "vs_synth_pairs: genome %u holds a byte outside ACGT", s);
            uint64_t b = gbase[s] + i;
            gwords[b >> 4] |= code << ((b & 15u) * 2u);
        }
    vs_reads *r = new vs_reads();
    r->n_ends = 2 * n_pairs;
    r->n_words = 2 * n_pairs * wpe;
    r->max_len = read_len;
    int rc = alloc_reads(ctx, r, false);
    uint32_t *d_gw = nullptr, *d_cum = nullptr;
    uint64_t *d_gb = nullptr, *d_gl = nullptr;
    hipStream_t st = ctx->stream;
    hipError_t e1 = hipSuccess;
    if (rc == VS_OK) {
        do {
            if ((e1 = hipMalloc((void **)&d_gw, sizeof(uint32_t) * gwords.size())) != hipSuccess) break;
            if ((e1 = hipMalloc((void **)&d_cum, sizeof(uint32_t) * n_strains)) != hipSuccess) break;
            if ((e1 = hipMalloc((void **)&d_gb, sizeof(uint64_t) * n_strains)) != hipSuccess) break;
            if ((e1 = hipMalloc((void **)&d_gl, sizeof(uint64_t) * n_strains)) != hipSuccess) break;
            if ((e1 = hipMemcpyAsync(d_gw, gwords.data(), sizeof(uint32_t) * gwords.size(), hipMemcpyHostToDevice, st)) != hipSuccess) break;
            if ((e1 = hipMemcpyAsync(d_cum, cum, sizeof(uint32_t) * n_strains, hipMemcpyHostToDevice, st)) != hipSuccess) break;
            if ((e1 = hipMemcpyAsync(d_gb, gbase.data(), sizeof(uint64_t) * n_strains, hipMemcpyHostToDevice, st)) != hipSuccess) break;
            if ((e1 = hipMemcpyAsync(d_gl, glen.data(), sizeof(uint64_t) * n_strains, hipMemcpyHostToDevice, st)) != hipSuccess) break;
            SynthParams P;
            P.gwords = d_gw; P.gbase = d_gb; P.glen = d_gl; P.cum = d_cum; P.n_strains = n_strains;
            P.seed = seed; P.first_pair = first_pair; P.n_pairs = n_pairs; P.read_len = read_len;
            P.words_per_end = wpe; P.sub_thresh = sub_thresh; P.n_thresh = n_thresh;
            uint64_t nthreads = r->n_words;
            if (nthreads)
                hipLaunchKernelGGL(k_synth, dim3((unsigned)((nthreads + TPB - 1) / TPB)), dim3(TPB), 0, st, P,
                                   (uint32_t *)r->d_words, (uint32_t *)r->d_meta);
            hipLaunchKernelGGL(k_iota_woff, dim3((unsigned)((r->n_ends + 1 + TPB - 1) / TPB)), dim3(TPB), 0, st,
                               (uint32_t *)r->d_woff, r->n_ends + 1, wpe);
            if ((e1 = hipGetLastError()) != hipSuccess) break;
            e1 = hipStreamSynchronize(st);
        } while (0);
        if (e1 != hipSuccess) rc = vs_fail(ctx, e1 == hipErrorOutOfMemory ? VS_E_OOM : VS_E_HIP, "vs_synth_pairs: %s", hipGetErrorString(e1));
    }
    void *tmps[] = {d_gw, d_cum, d_gb, d_gl};
    for (void *p : tmps)
        if (p) (void)hipFree(p);
    if (rc != VS_OK) { vs_reads_free(ctx, r); return rc; }
    *out = r;
    return VS_OK;
}
