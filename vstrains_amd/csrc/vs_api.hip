// Context, errors, device-memory helpers and the generic exclusive scan.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vs_internal.h"

static thread_local std::string g_create_err;

int vs_fail(vs_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

static bool env_on(const char *name) {
    const char *v = getenv(name);
    return v && atoi(v) != 0;
}

void vs_tuning_load(VsTuning &t, int level) {
    t = VsTuning();
    if (level < 1) return;
    if (const char *v = getenv("VS_EPT")) t.ept = (uint32_t)atoi(v) & ~1u;
    if (const char *v = getenv("VS_GRID_PER_CU")) t.grid_per_cu = atoi(v) > 0 ? (uint32_t)atoi(v) : 128u;
    if (const char *v = getenv("VS_ACC_GRID_PER_CU")) t.acc_grid_per_cu = atoi(v) > 0 ? (uint32_t)atoi(v) : 32u;
    if (const char *v = getenv("VS_ACC_FILL")) t.acc_fill_pct = atoi(v);
    if (const char *v = getenv("VS_ACC_ROUND")) { const int r = atoi(v); t.acc_round = (r == 64 || r == 128 || r == 256 || r == 512 || r == 1024) ? (uint32_t)r : 0u; }
    if (const char *v = getenv("VS_SHORTCUT")) t.shortcut = atoi(v) != 0 ? 1 : 0;
    if (const char *v = getenv("VS_ADAPT_GRID")) t.adapt_grid = atoi(v) != 0 ? 1 : 0;
    if (const char *v = getenv("VS_TABLE_SHIFT")) t.table_shift = atoi(v) < 1 ? 1u : atoi(v) > 8 ? 8u : (uint32_t)atoi(v);
    if (const char *v = getenv("VS_ACC_ROWS")) t.acc_rows = atoi(v) != 0 ? 1 : 0;
    if (const char *v = getenv("VS_LTAB_BITS")) t.ltab_bits = atoi(v) >= 0 && atoi(v) <= 31 ? atoi(v) : -1;
    if (const char *v = getenv("VS_ROWS_KEYS")) t.rows_keys = atoi(v) >= 2 && atoi(v) <= 65536 ? (uint32_t)atoi(v) : 0u;
    if (const char *v = getenv("VS_ROWS_SUB")) t.rows_sub = atoi(v) >= 1024 ? (uint32_t)atoi(v) : 0u;
    if (const char *v = getenv("VS_ROWS_PER_STRIP1")) t.rows_per_strip1 = atoi(v) > 0 && atoi(v) <= 64 ? (uint32_t)atoi(v) : 0u;
    if (const char *v = getenv("VS_ROWS_PER_STRIP")) t.rows_per_strip = atoi(v) > 0 && atoi(v) <= 64 ? (uint32_t)atoi(v) : 0u;
    t.no_sort = env_on("VS_NO_SORT");
    t.locus_global = env_on("VS_LOCUS_GLOBAL");
    t.no_xcd_map = env_on("VS_NO_XCD_MAP");
    t.no_fast = env_on("VS_NO_FAST");
    t.no_std = env_on("VS_NO_STD");
    t.phase0 = env_on("VS_PHASE0");
    t.no_agg = env_on("VS_NO_AGG");
    t.no_mid = env_on("VS_NO_MID");
    if (const char *v = getenv("VS_ACC_QUEUE")) t.acc_queue = atoi(v) != 0;
    if (const char *v = getenv("VS_ACC_WGS")) t.acc_wgs = atoi(v) > 0 ? (uint32_t)atoi(v) : 0u;
    t.debug_postings = getenv("VS_DEBUG_POSTINGS") != nullptr;
    t.debug_occ = getenv("VS_DEBUG_OCC") != nullptr;
    t.debug_acc = getenv("VS_DEBUG_ACC") != nullptr;
    if (level < 2) return;
    if (const char *v = getenv("VS_DEBUG_STOP")) t.debug_stop = (uint32_t)atoi(v);
    if (const char *v = getenv("VS_ACC_ABLATE")) t.acc_ablate = atoi(v);
}

void *vs_cache_alloc(vs_ctx *ctx, size_t bytes) {
    if (!bytes) bytes = 16;
    for (auto &b : ctx->cache)
        if (!b.used && b.cap >= bytes && b.cap <= 2 * bytes + (1u << 20)) {
            b.used = true;
            return b.p;
        }
    void *p = nullptr;
    const size_t cap = bytes + bytes / 8;  // (blocks of one file differ a little in size)
    if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
    ctx->cache.push_back({p, cap, true});
    return p;
}

void vs_cache_release(vs_ctx *ctx, void *p) {
    if (!p) return;
    size_t idle = 0;
    for (auto &b : ctx->cache)
        if (b.p == p) b.used = false;
    for (auto &b : ctx->cache)
        if (!b.used) idle++;
    if (idle > 24) {  // do not hoard: drop the idle ones
        std::vector<vs_ctx::CachedBuf> keep;
        for (auto &b : ctx->cache) {
            if (b.used) keep.push_back(b);
            else (void)hipFree(b.p);
        }
        ctx->cache.swap(keep);
    }
}

extern "C" {

int vs_abi_version(void) { return VS_ABI_VERSION; }

int vs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *vs_last_error(const vs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int vs_ctx_create(int device, vs_ctx **out) {
    if (!out) return vs_fail(nullptr, VS_E_ARG, "vs_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return vs_fail(nullptr, VS_E_HIP, "no HIP device available (%s); this library has no CPU path",
                       e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return vs_fail(nullptr, VS_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return vs_fail(nullptr, VS_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    vs_ctx *ctx = new vs_ctx();
    ctx->device = device;
    if (const char *ev = getenv("VS_EXPERIMENT")) ctx->experiment_level = strcmp(ev, "timing") == 0 ? 2 : strcmp(ev, "1") == 0 ? 1 : 0;
    vs_tuning_load(ctx->tune, ctx->experiment_level);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    for (int i = 0; i < 5; i++) {
        e = hipEventCreate(&ctx->ev[i]);
        if (e != hipSuccess) {
            int rc = vs_fail(nullptr, VS_E_HIP, "hipEventCreate: %s", hipGetErrorString(e));
            delete ctx;
            return rc;
        }
    }
    *out = ctx;
    return VS_OK;
}

static void free_index(vs_ctx *ctx) {
    void **ps[] = {&ctx->d_meta, &ctx->d_fwd, &ctx->d_rc, &ctx->d_table, &ctx->d_post};
    for (void **p : ps) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    ctx->has_index = false;
    ctx->index_bytes = 0;
}

void vs_ctx_free_index(vs_ctx *ctx) { free_index(ctx); }

void vs_ctx_destroy(vs_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_index(ctx);
    if (ctx->d_slow_list) (void)hipFree(ctx->d_slow_list);
    if (ctx->d_slow_count) (void)hipFree(ctx->d_slow_count);
    if (ctx->links_spare) (void)hipFree(ctx->links_spare);
    if (ctx->d_slow_list2) (void)hipFree(ctx->d_slow_list2);
    if (ctx->d_dense) (void)hipFree(ctx->d_dense);
    for (void *q : {ctx->d_locus_keys, ctx->d_perm, ctx->d_locus_hist, ctx->d_scan_tmp, ctx->d_lists, ctx->d_list_counts, ctx->d_rows, ctx->d_row_entries, ctx->d_mult, ctx->d_ltab})
        if (q) (void)hipFree(q);
    for (void *q : ctx->scratch)
        if (q) (void)hipFree(q);
    for (auto &b : ctx->cache) (void)hipFree(b.p);
    for (FqStage &st : ctx->fq_stage) {
        if (st.in_flight && st.done) (void)hipEventSynchronize(st.done);
        if (st.words) (void)hipHostFree(st.words);
        if (st.woff) (void)hipHostFree(st.woff);
        if (st.meta) (void)hipHostFree(st.meta);
        if (st.done) (void)hipEventDestroy(st.done);
    }
    for (int i = 0; i < 5; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    delete ctx;
}

int vs_ctx_set_stream(vs_ctx *ctx, void *stream) {
    if (!ctx) return VS_E_ARG;
    ctx->stream = (hipStream_t)stream;
    return VS_OK;
}

int vs_ctx_sync(vs_ctx *ctx) {
    if (!ctx) return VS_E_ARG;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VS_OK;
}

int vs_dev_alloc(vs_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return VS_E_ARG;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    VS_HIP(ctx, hipMalloc(out, bytes ? bytes : 16));
    VS_HIP(ctx, hipMemsetAsync(*out, 0, bytes ? bytes : 16, ctx->stream));
    return VS_OK;
}

int vs_dev_free(vs_ctx *ctx, void *ptr) {
    if (!ctx) return VS_E_ARG;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ptr) VS_HIP(ctx, hipFree(ptr));
    return VS_OK;
}

int vs_dev_zero(vs_ctx *ctx, void *ptr, size_t bytes) {
    if (!ctx || !ptr) return VS_E_ARG;
    VS_HIP(ctx, hipMemsetAsync(ptr, 0, bytes, ctx->stream));
    return VS_OK;
}

int vs_dev_to_host(vs_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
    if (!ctx || !host_dst || !dev_src) return VS_E_ARG;
    VS_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VS_OK;
}

}  // extern "C"

// ---- exclusive scan: 2048 values per 256-thread block, three launches ---------------------------
#define SCAN_TPB 256
#define SCAN_PER_THREAD 8
#define SCAN_PER_BLOCK (SCAN_TPB * SCAN_PER_THREAD)

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v) {
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__global__ void __launch_bounds__(SCAN_TPB) k_scan_block_sums(const uint32_t *in, uint64_t n, uint64_t *sums) {
    __shared__ uint64_t ws[SCAN_TPB / 64];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_PER_BLOCK + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; i++)
        if (base + i < n) s += in[base + i];
    uint64_t incl = wave_incl_scan(s);
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int i = 0; i < SCAN_TPB / 64; i++) t += ws[i];
        sums[blockIdx.x] = t;
    }
}

// single block: exclusive scan of the block sums in place; total goes to sums[nb]
__global__ void __launch_bounds__(SCAN_TPB) k_scan_sums(uint64_t *sums, uint64_t nb, uint64_t *total) {
    __shared__ uint64_t ws[SCAN_TPB / 64];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t start = 0; start < nb; start += SCAN_TPB) {
        uint64_t i = start + threadIdx.x;
        uint64_t v = i < nb ? sums[i] : 0;
        uint64_t incl = wave_incl_scan(v);
        if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t off = carry;
        for (int wv = 0; wv < (int)(threadIdx.x >> 6); wv++) off += ws[wv];
        if (i < nb) sums[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == SCAN_TPB - 1) carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[nb] = carry;
        if (total) *total = carry;
    }
}

__global__ void __launch_bounds__(SCAN_TPB) k_scan_apply(const uint32_t *in, uint32_t *out, uint64_t n, const uint64_t *sums) {
    __shared__ uint64_t ws[SCAN_TPB / 64];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_PER_BLOCK + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; i++) {
        v[i] = base + i < n ? in[base + i] : 0u;
        s += v[i];
    }
    uint64_t incl = wave_incl_scan(s);
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t off = sums[blockIdx.x];
    for (int wv = 0; wv < (int)(threadIdx.x >> 6); wv++) off += ws[wv];
    uint64_t run = off + incl - s;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; i++) {
        if (base + i < n) out[base + i] = (uint32_t)run;
        run += v[i];
    }
}

int vs_scan_u32(vs_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, uint64_t *d_tmp, uint64_t *d_total) {
    if (n == 0) {
        if (d_total) VS_HIP(ctx, hipMemsetAsync(d_total, 0, sizeof(uint64_t), ctx->stream));
        return VS_OK;
    }
    uint64_t nb = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(SCAN_TPB), 0, ctx->stream, in, n, d_tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_TPB), 0, ctx->stream, d_tmp, nb, d_total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_TPB), 0, ctx->stream, in, out, n, d_tmp);
    VS_HIP(ctx, hipGetLastError());
    return VS_OK;
}
