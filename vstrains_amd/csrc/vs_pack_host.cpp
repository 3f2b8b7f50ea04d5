// Host-side 2-bit packing of one sequence line (FASTQ ingest, vs_fastq.hip): 16 bases per word,
// LSB first, A C G T = 0 1 2 3; a byte outside ACGT packs as 0 and raises a flag (VS_FLAG_N for
// 'N', VS_FLAG_INVALID for any other ASCII byte, 0x80 for a byte >= 0x80, which the ingest
// refuses).  Plain C++ (no HIP in this file) so that the x86 vector headers can be used: SSE2 is
// part of every x86-64; the AVX2 body is picked at run time where the host has it.
//
// Per chunk of 16 / 32 bytes: four byte compares say whether every byte is one of ACGT (almost
// always), the code is ((c >> 1) & 3) ^ (that >> 1), and three shift-or steps squeeze the 2-bit
// fields of a 64-bit lane into 16 bits.  A chunk with any other byte, and the tail, go byte by byte.
#include <emmintrin.h>
#include <immintrin.h>
#include <stdint.h>

#include "vs_pack_host.h"

namespace {

struct CodeLut {
    uint8_t v[256];
    CodeLut() {
        for (int i = 0; i < 256; i++) v[i] = i < 128 ? 5 : 8;
        v['A'] = 0; v['C'] = 1; v['G'] = 2; v['T'] = 3; v['N'] = 4;
    }
};
const CodeLut g_code;

// bytes [0, m) of q, m <= 16
inline uint32_t pack_word_bytes(const uint8_t *q, uint32_t m, uint32_t *flags) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < m; i++) {
        const uint32_t c = g_code.v[q[i]];
        if (c <= 3u) v |= c << (2u * i);
        else if (c == 4u) *flags |= VS_PACK_FLAG_N;
        else if (c == 5u) *flags |= VS_PACK_FLAG_INVALID;
        else *flags |= VS_PACK_FLAG_NON_ASCII;
    }
    return v;
}

inline uint32_t pack_sse2(const uint8_t *q, uint32_t len, uint32_t *out) {
    uint32_t flags = 0;
    const __m128i cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T');
    const __m128i three = _mm_set1_epi8(3), one = _mm_set1_epi8(1);
    uint32_t wi = 0, b = 0;
    for (; b + 16u <= len; b += 16u, wi++) {
        const __m128i x = _mm_loadu_si128((const __m128i *)(q + b));
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(x, cA), _mm_cmpeq_epi8(x, cC)),
                                        _mm_or_si128(_mm_cmpeq_epi8(x, cG), _mm_cmpeq_epi8(x, cT)));
        if (_mm_movemask_epi8(ok) != 0xFFFF) {
            out[wi] = pack_word_bytes(q + b, 16u, &flags);
            continue;
        }
        __m128i t = _mm_and_si128(_mm_srli_epi16(x, 1), three);
        t = _mm_xor_si128(t, _mm_and_si128(_mm_srli_epi16(t, 1), one));
        t = _mm_and_si128(_mm_or_si128(t, _mm_srli_epi16(t, 6)), _mm_set1_epi16(0x000F));
        t = _mm_and_si128(_mm_or_si128(t, _mm_srli_epi32(t, 12)), _mm_set1_epi32(0x000000FF));
        t = _mm_or_si128(t, _mm_srli_epi64(t, 24));
        out[wi] = ((uint32_t)_mm_cvtsi128_si32(t) & 0xFFFFu) | ((uint32_t)_mm_extract_epi16(t, 4) << 16);
    }
    if (b < len) out[wi] = pack_word_bytes(q + b, len - b, &flags);
    return flags;
}

__attribute__((target("avx2"))) uint32_t pack_avx2(const uint8_t *q, uint32_t len, uint32_t *out) {
    uint32_t flags = 0;
    const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
    const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1);
    uint32_t wi = 0, b = 0;
    for (; b + 32u <= len; b += 32u, wi += 2) {
        const __m256i x = _mm256_loadu_si256((const __m256i *)(q + b));
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(x, cA), _mm256_cmpeq_epi8(x, cC)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(x, cG), _mm256_cmpeq_epi8(x, cT)));
        if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) {
            out[wi] = pack_word_bytes(q + b, 16u, &flags);
            out[wi + 1] = pack_word_bytes(q + b + 16u, 16u, &flags);
            continue;
        }
        __m256i t = _mm256_and_si256(_mm256_srli_epi16(x, 1), three);
        t = _mm256_xor_si256(t, _mm256_and_si256(_mm256_srli_epi16(t, 1), one));
        t = _mm256_and_si256(_mm256_or_si256(t, _mm256_srli_epi16(t, 6)), _mm256_set1_epi16(0x000F));
        t = _mm256_and_si256(_mm256_or_si256(t, _mm256_srli_epi32(t, 12)), _mm256_set1_epi32(0x000000FF));
        t = _mm256_or_si256(t, _mm256_srli_epi64(t, 24));
        out[wi] = ((uint32_t)_mm256_extract_epi16(t, 0)) | ((uint32_t)_mm256_extract_epi16(t, 4) << 16);
        out[wi + 1] = ((uint32_t)_mm256_extract_epi16(t, 8)) | ((uint32_t)_mm256_extract_epi16(t, 12) << 16);
    }
    if (b < len) flags |= pack_sse2(q + b, len - b, out + wi);
    return flags;
}

typedef uint32_t (*pack_fn)(const uint8_t *, uint32_t, uint32_t *);
pack_fn pick() {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx2") ? pack_avx2 : pack_sse2;
}
const pack_fn g_pack = pick();

}  // namespace

uint32_t vs_pack_sequence_host(const uint8_t *q, uint32_t len, uint32_t *out) { return g_pack(q, len, out); }

uint32_t vs_pack_sequence_host_plain(const uint8_t *q, uint32_t len, uint32_t *out) {
    uint32_t flags = 0;
    const uint32_t nw = (len + 15u) >> 4;
    for (uint32_t wi = 0; wi < nw; wi++) out[wi] = pack_word_bytes(q + 16u * wi, len - 16u * wi < 16u ? len - 16u * wi : 16u, &flags);
    return flags;
}
