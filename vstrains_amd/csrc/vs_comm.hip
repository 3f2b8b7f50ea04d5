// C1: sum of the PE counters over ranks with RCCL (ncclAllReduce over xGMI), and the widening fold
// of uint32 counters into int64 totals.
//
// The reference is one process (utils/VStrains_PE_Inference.py); its pair loop (:155-188) only ever
// adds to node_mat / short_mat, so ranks may count disjoint read blocks and sum afterwards -- any
// partition gives the same integers.  One process per GPU; the communicator is an ncclComm_t passed
// as void*.  RCCL is not a link-time dependency: a process that already holds an RCCL (torch ships
// one under the same soname) must not get a second copy, so the entry points are looked up at the
// first call -- in the copy the process has loaded, else by dlopen("librccl.so.1").
#include <dlfcn.h>
#include <string.h>

#include "vs_internal.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2: enum values and the 128-byte id)
typedef struct { char internal[128]; } RcclUniqueId;
typedef int (*fn_get_unique_id)(RcclUniqueId *);
typedef int (*fn_comm_init_rank)(void **comm, int nranks, RcclUniqueId id, int rank);
typedef int (*fn_comm_destroy)(void *comm);
typedef int (*fn_all_reduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st);
typedef const char *(*fn_error_string)(int);
enum { RCCL_SUM = 0, RCCL_UINT32 = 3, RCCL_INT64 = 4, RCCL_UINT64 = 5 };

struct Rccl {
    bool tried = false;
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    bool ok() const { return get_unique_id && comm_init_rank && comm_destroy && all_reduce; }
};
Rccl g_rccl;

const Rccl &rccl() {
    Rccl &r = g_rccl;
    if (r.tried) return r;
    r.tried = true;
    // a copy already in the global scope of the process wins; otherwise the soname (an already
    // mapped library of that soname is handed back by dlopen instead of a second one)
    // (RTLD_DEFAULT is a null handle, hence the separate flag)
    const bool in_scope = dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr;
    void *h = RTLD_DEFAULT;
    if (!in_scope) {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return r;
    }
    r.handle = h;
    r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    return r;
}

int rccl_missing(vs_ctx *ctx) {
    const char *why = dlerror();
    return vs_fail(ctx, VS_E_HIP, "RCCL (librccl.so.1) cannot be loaded: %s", why ? why : "entry points not found");
}

int rccl_fail(vs_ctx *ctx, const char *what, int code) {
    const Rccl &r = rccl();
    return vs_fail(ctx, VS_E_HIP, "%s failed: %s (rccl code %d)", what, r.error_string ? r.error_string(code) : "?", code);
}

__global__ void __launch_bounds__(256) k_counts_fold(uint32_t *__restrict__ counts, long long *__restrict__ wide, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = counts[i];
    if (c) {
        wide[i] += (long long)c;
        counts[i] = 0u;
    }
}

// occ[j] = 1 where the j-th stretch of 64 consecutive 32-bit words holds a non-zero word (int64 totals: a stretch of 64
// cells is two of these, OR-ed by the caller's stride).  One wavefront per stretch and turn, a lane per word, the verdict
// by ballot: the buffer is read once, coalesced, at the rate the chip streams.
__global__ void __launch_bounds__(256) k_counts_occupied(const uint32_t *__restrict__ words, uint64_t n_stretches, uint32_t words_per_stretch,
                                                         uint8_t *__restrict__ occ) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4u;
    for (uint64_t j = wave; j < n_stretches; j += n_waves) {
        uint32_t any = 0u;
        for (uint32_t k = lane; k < words_per_stretch; k += 64u) any |= words[j * words_per_stretch + k];
        const unsigned long long hit = __ballot(any != 0u);
        if (lane == 0u) occ[j] = hit ? 1u : 0u;
    }
}

}  // namespace

extern "C" {

int vs_counts_occupied(vs_ctx *ctx, const void *d_cells, uint32_t cell_bytes, uint64_t n_stretches, uint8_t *d_occ) {
    if (!ctx || (n_stretches && (!d_cells || !d_occ)) || (cell_bytes != 4u && cell_bytes != 8u)) return vs_fail(ctx, VS_E_ARG, "vs_counts_occupied: bad argument");
    if (!n_stretches) return VS_OK;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t grid = (n_stretches + 3u) / 4u;
    const uint64_t cap = (uint64_t)ctx->n_cu * 64u;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_counts_occupied, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint32_t *)d_cells, n_stretches, 16u * cell_bytes, d_occ);
    VS_HIP(ctx, hipGetLastError());
    return VS_OK;
}

int vs_counts_fold(vs_ctx *ctx, uint32_t *d_counts, int64_t *d_wide, uint64_t n) {
    if (!ctx || !d_counts || !d_wide) return VS_E_ARG;
    if (!n) return VS_OK;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_counts_fold, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, ctx->stream, d_counts,
                       (long long *)d_wide, n);
    VS_HIP(ctx, hipGetLastError());
    return VS_OK;
}

int vs_comm_unique_id(vs_ctx *ctx, uint8_t id[128]) {
    if (!id) return vs_fail(ctx, VS_E_ARG, "vs_comm_unique_id: id is NULL");
    const Rccl &r = rccl();
    if (!r.ok()) return rccl_missing(ctx);
    RcclUniqueId u;
    int rc = r.get_unique_id(&u);
    if (rc) return rccl_fail(ctx, "ncclGetUniqueId", rc);
    memcpy(id, u.internal, 128);
    return VS_OK;
}

int vs_comm_init_rank(vs_ctx *ctx, int n_ranks, const uint8_t id[128], int rank, void **comm) {
    if (!ctx || !id || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return vs_fail(ctx, VS_E_ARG, "vs_comm_init_rank: bad argument");
    *comm = nullptr;
    const Rccl &r = rccl();
    if (!r.ok()) return rccl_missing(ctx);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    RcclUniqueId u;
    memcpy(u.internal, id, 128);
    int rc = r.comm_init_rank(comm, n_ranks, u, rank);
    if (rc) return rccl_fail(ctx, "ncclCommInitRank", rc);
    return VS_OK;
}

int vs_comm_destroy(vs_ctx *ctx, void *comm) {
    if (!comm) return VS_OK;
    const Rccl &r = rccl();
    if (!r.ok()) return rccl_missing(ctx);
    if (ctx) {
        VS_HIP(ctx, hipSetDevice(ctx->device));
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    int rc = r.comm_destroy(comm);
    if (rc) return rccl_fail(ctx, "ncclCommDestroy", rc);
    return VS_OK;
}

int vs_pe_allreduce(vs_ctx *ctx, void *comm, void *d_node_mat, void *d_short_mat, uint64_t *d_stats, uint32_t n, int wide) {
    if (!ctx || !comm || !d_node_mat || !d_short_mat) return vs_fail(ctx, VS_E_ARG, "vs_pe_allreduce: bad argument");
    const Rccl &r = rccl();
    if (!r.ok()) return rccl_missing(ctx);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)n * n;
    const int dt = wide ? RCCL_INT64 : RCCL_UINT32;
    const size_t width = wide ? 8u : 4u;
    int rc;
    // one call when the two matrices are one [2, N, N] allocation (the layout PeCounter uses)
    if ((char *)d_short_mat == (char *)d_node_mat + cells * width) {
        if (cells && (rc = r.all_reduce(d_node_mat, d_node_mat, 2 * cells, dt, RCCL_SUM, comm, ctx->stream))) return rccl_fail(ctx, "ncclAllReduce", rc);
    } else {
        if (cells && (rc = r.all_reduce(d_node_mat, d_node_mat, cells, dt, RCCL_SUM, comm, ctx->stream))) return rccl_fail(ctx, "ncclAllReduce", rc);
        if (cells && (rc = r.all_reduce(d_short_mat, d_short_mat, cells, dt, RCCL_SUM, comm, ctx->stream))) return rccl_fail(ctx, "ncclAllReduce", rc);
    }
    if (d_stats && (rc = r.all_reduce(d_stats, d_stats, 3, RCCL_UINT64, RCCL_SUM, comm, ctx->stream))) return rccl_fail(ctx, "ncclAllReduce", rc);
    return VS_OK;
}

}  // extern "C"
