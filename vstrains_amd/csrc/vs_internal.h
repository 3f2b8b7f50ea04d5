// Internal structures and device helpers shared by the HIP translation units.
// gfx950 only: wave = 64 lanes, no other target is considered.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vstrains_hip.h"

#define VS_WAVE 64
#define VS_PAD_WORDS 16  // zero words behind every packed text buffer (window reads may overshoot)

// ---- packed text ---------------------------------------------------------------------------
// 2 bits per base (A=0 C=1 G=2 T=3), 16 bases per uint32 word, base i of a sequence sits in
// word i/16 at bits 2*(i%16) (LSB first).  Every sequence starts on a word boundary and every
// buffer carries VS_PAD_WORDS zero words so that window reads never leave the allocation.

struct VsNodeMeta {
    uint32_t woff;  // first word of the node in fwd_words / rc_words
    uint32_t len;   // bases
};

// One open-address slot of the seed table (16 B, read with one dwordx4 load).
//  key  : canonical w-mer (2w <= 62 bits); bit 62 set = several postings; all ones = empty
//  a, b : single -> a = node, b = pos | (node_strand << 31)
//         multi  -> a = first posting index, b = number of postings
struct VsSlot {
    uint64_t key;
    uint32_t a;
    uint32_t b;
};
#define VS_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define VS_MULTI_BIT (1ull << 62)

// One posting of a seed with several postings (16 B, one dwordx4 load): where the seed lies and
// the header of its node, so that the mapping kernel needs no second round trip for the header.
//   x: node   y: pos [0..23] | strand [31]   z: node length   w: first word of the node text
struct VsPosting {
    uint32_t node, pos, strand, len, woff;
};
#ifdef __HIPCC__
__host__ __device__ inline uint4 vs_posting_pack(const VsPosting &p) {
    uint4 r;
    r.x = p.node;
    r.y = p.pos | (p.strand << 31);
    r.z = p.len;
    r.w = p.woff;
    return r;
}
__host__ __device__ inline VsPosting vs_posting_unpack(const uint4 r) {
    VsPosting p;
    p.node = r.x;
    p.pos = r.y & 0x00FFFFFFu;
    p.strand = r.y >> 31;
    p.len = r.z;
    p.woff = r.w;
    return p;
}
#endif

struct VsIndexDev {
    uint32_t n_nodes;
    uint32_t K;  // split_len = ksize + 1
    uint32_t w;  // seed length (odd, <= 31, <= K)
    uint32_t s;  // probe stride on the read = K - w + 1
    uint32_t table_bits;
    const VsNodeMeta *meta;     // [n_nodes]
    const uint32_t *fwd_words;  // packed node texts
    const uint32_t *rc_words;   // packed reverse complements, same offsets; = fwd_words + rc_delta (one allocation,
    uint32_t rc_delta;          // so that a kernel can address either strand off one uniform base)
    const VsSlot *table;        // [1 << table_bits]
    const uint4 *postings;      // VsPosting records, the postings of one seed contiguous
};

struct VsReadsDev {
    uint64_t n_ends;
    const uint32_t *woff;   // [n_ends + 1] word offsets into words
    const uint32_t *meta;   // [n_ends] length (bits 0..23) | flags << 24
    const uint32_t *words;  // packed bases
    const uint32_t *mask;   // same layout, 0b11 at bytes outside ACGT; NULL when no end has any
    const uint32_t *inv4;   // [n_ends] up to four positions (one byte each, 0xFF = none) of such bytes; NULL with mask
};
#define VS_FLAG_N 1u        // read holds an upper-case 'N'
#define VS_FLAG_INVALID 2u  // read holds some other byte outside ACGT
#define VS_FLAG_MANY 4u     // ... more of them than inv4 holds (or one beyond position 254): overflow path
#define VS_LEN_MASK 0x00FFFFFFu

// ---- host-side objects -----------------------------------------------------------------------
// pinned staging of one FASTQ block in flight (vs_fastq_block): packed words (+ pad), word offsets,
// lengths | flags; two sets alternate, the cores fill one while the other is still being uploaded
struct FqStage {
    uint32_t *words = nullptr, *woff = nullptr, *meta = nullptr;
    size_t words_cap = 0, ends_cap = 0;
    hipEvent_t done = nullptr;  // the uploads out of this set have finished
    bool in_flight = false;
};

// Experiment switches of vs_pe_count (tuning sweeps, parity-test variants, timing ablations).  A production
// context never reads the environment per call: the switches exist only when the process was started with
// VS_EXPERIMENT=1 (parity-safe ones: every one of them is a test variant) or VS_EXPERIMENT=timing (also the
// ones that stop kernels early / skip work and therefore produce WRONG counters); they are then re-read on
// every call so that a test can flip them on a live context.  Defaults = the measured best.
struct VsTuning {
    uint32_t ept = 0;               // VS_EPT (0 = automatic)
    uint32_t grid_per_cu = 128;     // VS_GRID_PER_CU
    uint32_t acc_grid_per_cu = 32;  // VS_ACC_GRID_PER_CU
    int acc_fill_pct = -1;          // VS_ACC_FILL (-1 = 1/16 of the slots)
    uint32_t acc_round = 0;         // VS_ACC_ROUND: pairs per round of the counter kernel (0 = automatic; 64 .. 1024, power of two)
    int shortcut = -1;              // VS_SHORTCUT (-1 = by index statistics)
    int adapt_grid = -1;            // VS_ADAPT_GRID (-1 = by index statistics): the adaptive step grid of the compile-time-shape kernels
    uint32_t table_shift = 3;       // VS_TABLE_SHIFT: seed table of >= (distinct seeds << shift) slots (3: at most an eighth full)
    int acc_rows = -1;              // VS_ACC_ROWS (-1 = by graph size): counters summed by row owners (k_rows_sum) instead of pair-major (k_pe_accumulate)
    int ltab_bits = -1;             // VS_LTAB_BITS: log2 slots of the block's list table (-1 = by block size, 0 = no table: every end stands for itself)
    uint32_t rows_keys = 0, rows_sub = 0;  // VS_ROWS_KEYS / VS_ROWS_SUB: rows per histogram pass, pairs per transposition (0 = the constants; tests shrink them)
    uint32_t rows_per_strip1 = 0;   // VS_ROWS_PER_STRIP1: the same for short_mat alone
    uint32_t rows_per_strip = 0;    // VS_ROWS_PER_STRIP (0 = automatic): matrix rows one workgroup of k_rows_sum owns at a time
    bool no_sort = false, locus_global = false, no_xcd_map = false, no_fast = false, no_std = false, no_agg = false;
    bool acc_queue = true;
    uint32_t acc_wgs = 0;           // VS_ACC_WGS: workgroups of k_pe_accumulate's chunk queue (0 = two per CU)
    bool no_mid = false;            // VS_NO_MID: overflow pairs straight to k_pe_slow
    bool phase0 = false;            // VS_PHASE0: probe grid 0, s, 2s, ... as before round 5 (vs_seed_phase); the generic kernels only (implies VS_NO_STD)
    bool debug_postings = false, debug_occ = false, debug_acc = false;
    // timing only (VS_EXPERIMENT=timing): wrong counters by design
    uint32_t debug_stop = 0;        // VS_DEBUG_STOP=1..5
    int acc_ablate = -1;            // VS_ACC_ABLATE=2|3
};
void vs_tuning_load(VsTuning &t, int level);  // level 0: defaults, 1: parity-safe switches, 2: + timing-only ones

struct vs_ctx {
    int device = 0;
    int experiment_level = 0;  // VS_EXPERIMENT at vs_ctx_create: 0 none, 1 "1", 2 "timing"
    VsTuning tune;
    hipStream_t stream = nullptr;
    std::string err;
    FqStage fq_stage[2];
    unsigned fq_next = 0;
    bool has_index = false;
    VsIndexDev idx{};
    // owned device allocations of the index
    void *d_meta = nullptr, *d_fwd = nullptr, *d_rc = nullptr, *d_table = nullptr, *d_post = nullptr;
    uint64_t n_seed_pos = 0, n_slots = 0, n_distinct = 0, index_bytes = 0;
    uint32_t max_node_len = 0;
    // scratch for vs_pe_count
    void *d_slow_list = nullptr;   // pair indices sent to the slow path
    uint64_t slow_cap = 0;
    void *d_slow_count = nullptr;  // uint32 counters (see pe_launch)
    void *links_spare = nullptr;   // vs_links_reserve: the buffer of the next link table (links_spare_n nodes)
    uint32_t links_spare_n = 0;
    void *d_slow_list2 = nullptr;  // pairs k_pe_mid hands on to k_pe_slow
    uint64_t slow2_cap = 0;
    void *d_dense = nullptr;       // dense per-workgroup state for the slow path
    uint64_t dense_bytes = 0;
    uint32_t dense_nodes = 0xFFFFFFFFu;  // node count the dense layout was initialised for
    // locus order scratch (k_pe_locus / k_pe_permute)
    void *d_locus_keys = nullptr, *d_perm = nullptr, *d_locus_hist = nullptr, *d_scan_tmp = nullptr;
    uint64_t locus_cap = 0, hist_cap = 0;
    double last_sort_ms = 0;
    // per-end accepted lists between k_pe_tiles and k_pe_accumulate
    void *d_lists = nullptr, *d_list_counts = nullptr;
    uint64_t lists_cap = 0, lists_words = 0;
    // row-owner counting (k_list_owners / k_rows_count / k_rows_fill / k_rows_sum): per matrix and row the counts, cursors
    // and offsets (6 x (N + 2) words), the items of every row (one word per listed node), the multiplicity of every end's list
    void *d_rows = nullptr, *d_row_entries = nullptr, *d_mult = nullptr, *d_ltab = nullptr;
    uint64_t rows_cap = 0, row_entries_cap = 0 /* pairs */, ltab_cap = 0 /* slots */;
    // grow-only device scratch slots of the graph-stage entry points (no hipMalloc per call)
    void *scratch[32] = {};
    size_t scratch_cap[32] = {};
    // device buffers of freed read blocks, kept for the next block of about the same size (the
    // FASTQ ingest makes and frees one block per million pairs)
    struct CachedBuf { void *p; size_t cap; bool used; };
    std::vector<CachedBuf> cache;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double last_ms[3] = {0, 0, 0};
    const char *last_kernel = "";  // mapping-kernel instantiation of the last vs_pe_count
    uint32_t last_launched = 0;    // VS_RAN_* bits: which optional kernels that call launched
    int n_cu = 256;
};

struct vs_reads {
    uint64_t n_ends = 0, n_words = 0, max_len = 0, n_invalid = 0, bytes = 0;
    void *d_woff = nullptr, *d_meta = nullptr, *d_words = nullptr, *d_mask = nullptr, *d_inv4 = nullptr;
    bool cached = false;  // buffers came from vs_cache_alloc
    VsReadsDev dev() const {
        VsReadsDev r;
        r.n_ends = n_ends;
        r.woff = (const uint32_t *)d_woff;
        r.meta = (const uint32_t *)d_meta;
        r.words = (const uint32_t *)d_words;
        r.mask = (const uint32_t *)d_mask;
        r.inv4 = (const uint32_t *)d_inv4;
        return r;
    }
};

// The probe grid of a read end (round 5).  A match of K bases and more holds s = K - w + 1 consecutive seed starts, so any
// grid phi, phi + s, phi + 2s, ... of read offsets finds it, and the first grid point inside it credits it (left extension
// below s: vs_extend / k_pe_tiles).  phi = 0 spends a probe on the read's last, partly covered stride: floor((len - w) / s) + 1
// probes.  With r = (len - w) mod s every phi in (r, s) needs one probe less -- floor((len - w + 1) / s), the fewest any
// exact grid of stride s can have (the len - K + 1 windows of a read in runs of s) -- and the grid's seeds then lie
// inside the read instead of flush with its first base.  phi = (r + s) / 2 is the middle of that range (and s - 1, the
// only choice, when r = s - 1).  Measured on the configs[2] stream: 21.9 instead of 27.4 postings per end, four probes
// instead of five (profiles/r5/phase_gate_config2.json).  Ends shorter than K are never probed (PE_Inference.py:160-163).
__host__ __device__ inline uint32_t vs_seed_phase(uint32_t len, uint32_t w, uint32_t s, bool phase0 = false) {
    return (phase0 || len < w) ? 0u : ((len - w) % s + s) / 2u;
}
__host__ __device__ inline uint32_t vs_seed_probes(uint32_t len, uint32_t w, uint32_t s, bool phase0 = false) {
    if (len < w) return 0u;
    if (phase0) return (len - w) / s + 1u;
    return (len - w + 1u) / s;  // (0 for an end shorter than K = w + s - 1: it has no (k+1)-window, PE_Inference.py:23, and is never probed)
}

int vs_fail(vs_ctx *ctx, int code, const char *fmt, ...);
// grow-only cache of device buffers (see vs_ctx::cache); NULL on allocation failure
void *vs_cache_alloc(vs_ctx *ctx, size_t bytes);
void vs_cache_release(vs_ctx *ctx, void *p);
extern "C" void vs_ctx_free_index(vs_ctx *ctx);
#define VS_HIP(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess)                                                             \
            return vs_fail(ctx, e__ == hipErrorOutOfMemory ? VS_E_OOM : VS_E_HIP,          \
                           "%s failed: %s", #call, hipGetErrorString(e__));                \
    } while (0)

// Exclusive scan of n uint32 values on the ctx stream (in -> out, may alias); total (uint64) is
// written to d_total if not NULL.  tmp must hold ceil(n/2048)+1 uint64.
int vs_scan_u32(vs_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, uint64_t *d_tmp,
                uint64_t *d_total);

// ---- device helpers ----------------------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ uint32_t vs_code(uint8_t c) {
    // A C G T -> 0 1 2 3, anything else -> 4
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

// 32 bases (64 bits) starting at base offset `base` of a packed word array.
__device__ __forceinline__ uint64_t vs_win64(const uint32_t *w, uint64_t base) {
    uint64_t i = base >> 4;
    uint32_t sh = (uint32_t)(base & 15u) * 2u;
    uint64_t lo = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
    uint64_t hi = (uint64_t)w[i + 2];
    return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}

// Same with a 32-bit base offset (LDS tiles, node texts below 2^32 bases): no 64-bit address math.
__device__ __forceinline__ uint64_t vs_win64_u32(const uint32_t *w, uint32_t base) {
    const uint32_t i = base >> 4;
    const uint32_t sh = (base & 15u) * 2u;
    const uint32_t w0 = w[i], w1 = w[i + 1], w2 = w[i + 2];
    // funnel shifts: (w1:w0) >> sh and (w2:w1) >> sh, sh in 0..30
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__device__ __forceinline__ uint64_t vs_win(const uint32_t *w, uint32_t base) { return vs_win64_u32(w, base); }
__device__ __forceinline__ uint64_t vs_win(const uint32_t *w, uint64_t base) { return vs_win64(w, base); }

// x / d with a precomputed magic = floor(2^32 / d) + 1 (0 stands for d == 1); exact while
// x * d < 2^32, which holds for the tile-sized operands used here
__device__ __forceinline__ uint32_t vs_fastdiv(uint32_t x, uint32_t magic) { return magic ? __umulhi(x, magic) : x; }

__device__ __forceinline__ uint64_t vs_lowmask(uint32_t bits) {  // bits in 0..64
    return bits >= 64u ? ~0ull : ((1ull << bits) - 1ull);
}

// Reverse complement of a w-mer held LSB-first in the low 2w bits.
__device__ __forceinline__ uint64_t vs_rc(uint64_t x, uint32_t w) {
    // order of the 2-bit codes reversed = all 64 bits reversed (two v_bfrev_b32), then the two bits of
    // every code swapped back; complement = bitwise not
    x = __builtin_bitreverse64(x);
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    return (~x) >> (64u - 2u * w);
}

__device__ __forceinline__ uint64_t vs_mix64(uint64_t z);
// Key of the seed (w-mer, w <= 63) at base offset `base` of a packed text, and which strand it is the smaller on.
// w <= 31: the canonical w-mer itself.  Longer seeds (k >= 95: w = 63) do not fit a 62-bit key: the key is a mix of the
// canonical 2w bits, so two different seeds may share a key -- which is why, for those indexes, the comparison that
// follows a probe starts at the seed's FIRST base instead of behind it (VS_SEED_VERIFIED below): a posting is credited
// on compared text only, the key merely nominates it.
template <typename B>
__device__ __forceinline__ uint64_t vs_seed_key(const uint32_t *words, B base, uint32_t w, uint32_t *strand) {
    if (w <= 31u) {
        const uint64_t f = vs_win(words, base) & vs_lowmask(2u * w);
        const uint64_t r = vs_rc(f, w);
        *strand = r < f ? 1u : 0u;
        return r < f ? r : f;
    }
    const uint32_t sh = 2u * (w - 32u);  // bits of the seed in the second word pair (w = 63: 62)
    const uint64_t lo = vs_win(words, base), hi = vs_win(words, base + (B)32) & vs_lowmask(sh);
    // reverse complement of the 2w bits (hi:lo): the first 32 bases reversed go to the top
    const uint64_t rl = vs_rc(lo, 32u), rh = vs_rc(hi, w - 32u);
    const uint64_t rc_lo = (rl << sh) | rh, rc_hi = sh ? rl >> (64u - sh) : 0ull;
    const bool rev = rc_hi < hi || (rc_hi == hi && rc_lo < lo);
    *strand = rev ? 1u : 0u;
    const uint64_t a = rev ? rc_lo : lo, b = rev ? rc_hi : hi;
    return vs_mix64(a ^ vs_mix64(b + 0x632BE59BD9B4E019ull)) >> 2;  // 62 bits: neither the empty key nor the multi flag
}
// does the seed at `base` hold a byte outside ACGT?  (mask: 0b11 per such byte)
template <typename B>
__device__ __forceinline__ bool vs_seed_dirty(const uint32_t *mask, B base, uint32_t w) {
    if (w <= 32u) return (vs_win(mask, base) & vs_lowmask(2u * w)) != 0ull;
    return vs_win(mask, base) != 0ull || (vs_win(mask, base + (B)32) & vs_lowmask(2u * (w - 32u))) != 0ull;
}
// bases of the seed the comparison may take for granted: all of them for exact keys, none for mixed ones
#define VS_SEED_VERIFIED(w) ((w) <= 31u ? (w) : 0u)

__device__ __forceinline__ uint32_t vs_slot_of(uint64_t key, uint32_t bits) {
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64u - bits));
}

__device__ __forceinline__ uint64_t vs_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// index of the last element <= x in a sorted array a[0..n) with a[0] <= x
template <typename T>
__device__ __forceinline__ uint32_t vs_upper_idx(const T *a, uint32_t n, T x) {
    uint32_t lo = 0, hi = n;  // invariant: a[lo] <= x, a[hi] > x (virtually)
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (a[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

#endif  // __HIPCC__
