// Graph-stage kernels (K5 PE-link table, K6 vertex scan / chain ranking, K7 edge
// flow).  gfx950 only.  Everything here is integer / fp64 index work bound by memory latency
// or HBM bandwidth; no MFMA.
//
// Reference lines (under /root/reference/utils/) each kernel stands in for are cited at the
// kernels.  fp64 arithmetic must reproduce numpy bit for bit: the file is compiled with
// -ffp-contract=off and the sums follow numpy's pairwise_sum order.
#include "vs_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

struct vs_links {
    uint32_t n = 0;
    int64_t *d_p0 = nullptr;  // [n*n] symmetric; diagonal = node[i][i] + short[i][i]  (dense form)
    // sparse form (r6, graphs of 2^15 nodes and more whose counters keep a dirty-tile map): CSR rows of the non-zero cells,
    // columns ascending inside a row -- 0.4 GB instead of 23.7 GB at 54 465 nodes (1.2 % of the cells are non-zero)
    uint32_t *d_row_ptr = nullptr;  // [n + 1]
    uint32_t *d_col = nullptr;      // [nnz]
    int64_t *d_val = nullptr;       // [nnz]
    uint64_t nnz = 0;
    bool sparse() const { return d_row_ptr != nullptr; }
};

// ---------------------------------------------------------------------------------------------
// K5a  P0 = symmetrised counts.  VStrains_IO.py:598-627 (process_pe_info): every line u:v:c of
// pe_info and st_info is added under the key (min(u,v), max(u,v)); so for i != j the key holds
// node[i][j] + node[j][i] + short[i][j] + short[j][i] and for i == j node[i][i] + short[i][i].
// One workgroup handles the tile pair (I,J),(J,I), I <= J: every input element is read once
// with row-contiguous (coalesced) loads, transposed through LDS, and both output tiles are
// written row-contiguously.
// ---------------------------------------------------------------------------------------------
#define SYM_T 32
template <typename TIn>
__global__ void __launch_bounds__(256) k_links_symmetrize(const TIn *__restrict__ node, const TIn *__restrict__ shrt,
                                                         uint32_t n, uint32_t tiles, int64_t *__restrict__ p0) {
    __shared__ int64_t a[SYM_T][SYM_T + 1];  // sum tile (I,J)
    __shared__ int64_t b[SYM_T][SYM_T + 1];  // sum tile (J,I)
    // linear pair index -> (I, J) with I <= J
    uint32_t p = blockIdx.x;
    uint32_t I = 0;
    uint32_t rowlen = tiles;
    while (p >= rowlen) {
        p -= rowlen;
        I++;
        rowlen--;
    }
    uint32_t J = I + p;
    uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;  // 32 x 8
    for (uint32_t r = ty; r < SYM_T; r += 8) {
        uint32_t gi = I * SYM_T + r, gj = J * SYM_T + tx;
        int64_t v = 0;
        if (gi < n && gj < n) v = (int64_t)node[(uint64_t)gi * n + gj] + (int64_t)shrt[(uint64_t)gi * n + gj];
        a[r][tx] = v;
        uint32_t hi = J * SYM_T + r, hj = I * SYM_T + tx;
        int64_t w = 0;
        if (hi < n && hj < n) w = (int64_t)node[(uint64_t)hi * n + hj] + (int64_t)shrt[(uint64_t)hi * n + hj];
        b[r][tx] = w;
    }
    __syncthreads();
    for (uint32_t r = ty; r < SYM_T; r += 8) {
        uint32_t gi = I * SYM_T + r, gj = J * SYM_T + tx;
        if (gi < n && gj < n) p0[(uint64_t)gi * n + gj] = (gi == gj) ? a[r][tx] : a[r][tx] + b[tx][r];
        if (I != J) {
            uint32_t hi = J * SYM_T + r, hj = I * SYM_T + tx;
            if (hi < n && hj < n) p0[(uint64_t)hi * n + hj] = b[r][tx] + a[tx][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K5b  block sums: out[q] = sum_{r in list qa[q]} sum_{c in list qb[q]} P0[r][c].
// Stands in for every pe_info[(min,max)] read of the stages after splits/contractions
// (Decomposition.py:178,273; Extension.py:62) -- see vstrains_amd/graph/ops.py for why this sum
// equals the rewritten dict entry.  One wavefront per query; lanes stride the column list.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_links_block_sums(const int64_t *__restrict__ p0, uint32_t n,
                                                         const uint64_t *__restrict__ list_off,
                                                         const uint32_t *__restrict__ list_idx,
                                                         const uint32_t *__restrict__ qa, const uint32_t *__restrict__ qb,
                                                         uint64_t n_queries, int64_t *__restrict__ out) {
    uint64_t q = (uint64_t)blockIdx.x * (blockDim.x / VS_WAVE) + (threadIdx.x / VS_WAVE);
    if (q >= n_queries) return;
    uint32_t lane = threadIdx.x & (VS_WAVE - 1);
    uint64_t a0 = list_off[qa[q]], a1 = list_off[qa[q] + 1];
    uint64_t b0 = list_off[qb[q]], b1 = list_off[qb[q] + 1];
    int64_t s = 0;
    for (uint64_t i = a0; i < a1; i++) {
        const int64_t *row = p0 + (uint64_t)list_idx[i] * n;
        for (uint64_t j = b0 + lane; j < b1; j += VS_WAVE) s += row[list_idx[j]];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, VS_WAVE);
    if (lane == 0) out[q] = s;
}

// K5c  grouped contraction C = A P0 A^T in two passes (Extension.py:766-799 final_link_info;
// Utilities.py:488-495 row sums of a contracted path).  Pass 1: T[g][c] = sum_{r in G_g} P0[r][c]
// (coalesced along c).  Pass 2: C[g][h] = sum_{c in G_h} T[g][c] (one wavefront per (g,h)).
__global__ void __launch_bounds__(256) k_links_group_rows(const int64_t *__restrict__ p0, uint32_t n,
                                                         const uint64_t *__restrict__ list_off,
                                                         const uint32_t *__restrict__ list_idx,
                                                         int64_t *__restrict__ t) {
    uint32_t g = blockIdx.y;
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    int64_t s = 0;
    for (uint64_t i = list_off[g]; i < list_off[g + 1]; i++) s += p0[(uint64_t)list_idx[i] * n + c];
    t[(uint64_t)g * n + c] = s;
}

__global__ void __launch_bounds__(256) k_links_group_cols(const int64_t *__restrict__ t, uint32_t n, uint32_t n_groups,
                                                         const uint64_t *__restrict__ list_off,
                                                         const uint32_t *__restrict__ list_idx,
                                                         int64_t *__restrict__ out) {
    uint64_t pair = (uint64_t)blockIdx.x * (blockDim.x / VS_WAVE) + (threadIdx.x / VS_WAVE);
    if (pair >= (uint64_t)n_groups * n_groups) return;
    uint32_t g = (uint32_t)(pair / n_groups), h = (uint32_t)(pair % n_groups);
    uint32_t lane = threadIdx.x & (VS_WAVE - 1);
    int64_t s = 0;
    const int64_t *row = t + (uint64_t)g * n;
    for (uint64_t j = list_off[h] + lane; j < list_off[h + 1]; j += VS_WAVE) s += row[list_idx[j]];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, VS_WAVE);
    if (lane == 0) out[pair] = s;
}

// ---------------------------------------------------------------------------------------------
// K5 sparse (r6): the same table as CSR rows, built from the counters' DIRTY 64 x 64 tiles only
// (vs_pe_count_tracked marks a tile wherever a block adds to a cell; a cell outside the marked tiles
// is zero).  process_pe_info (IO.py:598-627) as above: cell {i, j} = node[i][j] + node[j][i] +
// short[i][j] + short[j][i], the diagonal once.  A workgroup takes the tile pair (I, J), I <= J, if any
// of its four source tiles is marked, sums them into S = tile (I, J) of the table (LDS, int64) --
// tile (J, I) is S transposed -- and
//   pass 1 (k_sl_count)  counts the non-zero cells per table row and tile column -> cnt[row][tile col] (bytes)
//   (k_sl_rows)          per row: the prefix of its tile-column counts -> where every (row, tile col) piece starts in
//                        the row, and the row's total; a scan over the totals gives row_ptr
//   pass 2 (k_sl_fill)   sums the same tiles again and writes every row's cells of that tile column, ascending, by a
//                        ballot / prefix count over the 64 columns (one wavefront per row)
// Nothing but the marked tiles is read: 2 x 0.5 GB instead of the 2 x 11.9 GB of counters + 23.7 GB of table.
// ---------------------------------------------------------------------------------------------
#define SL_T 64u
template <typename TIn>
__device__ __forceinline__ bool sl_load_pair(const TIn *__restrict__ node, const TIn *__restrict__ shrt, const uint8_t *__restrict__ map,
                                             uint32_t n, uint32_t T, uint32_t I, uint32_t J, int64_t (*S)[SL_T + 1]) {
    // marked source tiles: node (I,J), node (J,I), short (I,J), short (J,I)
    const uint8_t *smap = map + (uint64_t)T * T;
    const bool a = map[(uint64_t)I * T + J], b = map[(uint64_t)J * T + I], c = smap[(uint64_t)I * T + J], d = smap[(uint64_t)J * T + I];
    if (!(a | b | c | d)) return false;
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;  // 64 x 4
    for (uint32_t r = ty; r < SL_T; r += 4u) S[r][tx] = 0;
    __syncthreads();
    for (uint32_t r = ty; r < SL_T; r += 4u) {
        const uint32_t gi = I * SL_T + r, gj = J * SL_T + tx;
        if (gi < n && gj < n && (a | c)) {
            const uint64_t at = (uint64_t)gi * n + gj;
            S[r][tx] += (a ? (int64_t)node[at] : 0) + (c ? (int64_t)shrt[at] : 0);
        }
    }
    __syncthreads();
    // row r of tile (J, I) adds to column r of S; on the diagonal tile (J == I) that is S + S^T, the diagonal cells once
    for (uint32_t r = ty; r < SL_T; r += 4u) {
        const uint32_t hi = J * SL_T + r, hj = I * SL_T + tx;
        if (hi < n && hj < n && (b | d) && (I != J || r != tx)) {
            const uint64_t at = (uint64_t)hi * n + hj;
            S[tx][r] += (b ? (int64_t)node[at] : 0) + (d ? (int64_t)shrt[at] : 0);
        }
    }
    __syncthreads();
    return true;
}

__device__ __forceinline__ void sl_pair_of(uint32_t p, uint32_t T, uint32_t &I, uint32_t &J) {
    // linear pair index -> (I, J) with I <= J, row I of the upper triangle holds T - I pairs
    const double tt = 2.0 * T + 1.0;
    uint32_t i = (uint32_t)((tt - sqrt(tt * tt - 8.0 * (double)p)) * 0.5);
    while (i > 0 && (uint64_t)i * T - (uint64_t)i * (i - 1u) / 2u > p) i--;
    while ((uint64_t)(i + 1u) * T - (uint64_t)(i + 1u) * i / 2u <= p) i++;
    I = i;
    J = i + (uint32_t)(p - ((uint64_t)i * T - (uint64_t)i * (i - 1u) / 2u));
}

template <typename TIn>
__global__ void __launch_bounds__(256) k_sl_count(const TIn *__restrict__ node, const TIn *__restrict__ shrt, const uint8_t *__restrict__ map,
                                                 uint32_t n, uint32_t T, uint8_t *__restrict__ cnt) {
    __shared__ int64_t S[SL_T][SL_T + 1];
    uint32_t I, J;
    sl_pair_of(blockIdx.x, T, I, J);
    if (!sl_load_pair(node, shrt, map, n, T, I, J, S)) return;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t r = wv; r < SL_T; r += 4u) {
        const unsigned long long row_nz = __ballot(S[r][lane] != 0), col_nz = __ballot(S[lane][r] != 0);
        if (lane == 0) {
            const uint32_t gi = I * SL_T + r, gj = J * SL_T + r;
            if (gi < n && row_nz) cnt[(uint64_t)gi * T + J] = (uint8_t)__popcll(row_nz);           // row r of tile (I, J)  (at most 64: fits the byte)
            if (I != J && gj < n && col_nz) cnt[(uint64_t)gj * T + I] = (uint8_t)__popcll(col_nz);  // row r of tile (J, I) = column r of S
        }
    }
}

// (cnt: cells of a 64-cell piece of a row, one byte; 0 = the piece is empty and has no offset)
__global__ void __launch_bounds__(256) k_sl_rows(const uint8_t *__restrict__ cnt, uint32_t n, uint32_t T, uint32_t *__restrict__ piece_off,
                                                uint32_t *__restrict__ row_total) {
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row >= n) return;
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
        const uint32_t t = t0 + lane;
        const uint32_t c = t < T ? cnt[(uint64_t)row * T + t] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (c) piece_off[(uint64_t)row * T + t] = run + incl - c;
        run += __shfl(incl, 63, 64);
    }
    if (lane == 0) row_total[row] = run;
}

template <typename TIn>
__global__ void __launch_bounds__(256) k_sl_fill(const TIn *__restrict__ node, const TIn *__restrict__ shrt, const uint8_t *__restrict__ map,
                                                uint32_t n, uint32_t T, const uint32_t *__restrict__ piece_off, const uint32_t *__restrict__ row_ptr,
                                                uint32_t *__restrict__ col, int64_t *__restrict__ val) {
    __shared__ int64_t S[SL_T][SL_T + 1];
    uint32_t I, J;
    sl_pair_of(blockIdx.x, T, I, J);
    if (!sl_load_pair(node, shrt, map, n, T, I, J, S)) return;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t r = wv; r < SL_T; r += 4u) {
        {   // row r of tile (I, J): columns J * 64 + lane
            const int64_t v = S[r][lane];
            const unsigned long long nz = __ballot(v != 0);
            const uint32_t gi = I * SL_T + r;
            if (v != 0 && gi < n) {
                const uint32_t at = row_ptr[gi] + piece_off[(uint64_t)gi * T + J] + (uint32_t)__popcll(nz & ((1ull << lane) - 1ull));
                col[at] = J * SL_T + lane;
                val[at] = v;
            }
        }
        if (I != J) {  // row r of tile (J, I): columns I * 64 + lane
            const int64_t v = S[lane][r];
            const unsigned long long nz = __ballot(v != 0);
            const uint32_t gj = J * SL_T + r;
            if (v != 0 && gj < n) {
                const uint32_t at = row_ptr[gj] + piece_off[(uint64_t)gj * T + I] + (uint32_t)__popcll(nz & ((1ull << lane) - 1ull));
                col[at] = I * SL_T + lane;
                val[at] = v;
            }
        }
    }
}

// block sums over the CSR rows: one wavefront per query; for every row of list a the lanes take the columns of list b and
// look each up in the row (binary search over ascending columns)
__global__ void __launch_bounds__(256) k_links_block_sums_csr(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                                                             const int64_t *__restrict__ val, const uint64_t *__restrict__ list_off,
                                                             const uint32_t *__restrict__ list_idx, const uint32_t *__restrict__ qa,
                                                             const uint32_t *__restrict__ qb, uint64_t n_queries, int64_t *__restrict__ out) {
    uint64_t q = (uint64_t)blockIdx.x * (blockDim.x / VS_WAVE) + (threadIdx.x / VS_WAVE);
    if (q >= n_queries) return;
    uint32_t lane = threadIdx.x & (VS_WAVE - 1);
    uint64_t a0 = list_off[qa[q]], a1 = list_off[qa[q] + 1];
    uint64_t b0 = list_off[qb[q]], b1 = list_off[qb[q] + 1];
    int64_t s = 0;
    for (uint64_t i = a0; i < a1; i++) {
        const uint32_t r = list_idx[i];
        const uint32_t lo0 = row_ptr[r], hi0 = row_ptr[r + 1];
        if (lo0 == hi0) continue;
        for (uint64_t j = b0 + lane; j < b1; j += VS_WAVE) {
            const uint32_t c = list_idx[j];
            uint32_t lo = lo0, hi = hi0;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (col[mid] < c) lo = mid + 1u; else hi = mid;
            }
            if (lo < hi0 && col[lo] == c) s += val[lo];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, VS_WAVE);
    if (lane == 0) out[q] = s;
}

// ---------------------------------------------------------------------------------------------
// numpy.sum of a contiguous float64 vector = pairwise_sum over all elements
// (numpy/_core/src/umath/loops_utils.h.src): n < 8 sequential from 0; n <= 128 eight running
// accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) then the tail; larger n splits at
// n/2 rounded down to a multiple of 8.  Elements are dp[nbr[base + i]].
// ---------------------------------------------------------------------------------------------
__device__ double vs_np_pairwise_le128(const double *__restrict__ dp, const uint32_t *__restrict__ nbr, uint64_t base, uint32_t n) {
    if (n < 8) {
        double res = 0.0;
        for (uint32_t i = 0; i < n; i++) res += dp[nbr[base + i]];
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = dp[nbr[base + j]];
    uint32_t i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] += dp[nbr[base + i + j]];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += dp[nbr[base + i]];
    return res;
}

__device__ double vs_np_pairwise(const double *__restrict__ dp, const uint32_t *__restrict__ nbr, uint64_t base, uint32_t n) {
    if (n <= 128) return vs_np_pairwise_le128(dp, nbr, base, n);
    // degree > 128: numpy recurses on (n/2 rounded down to a multiple of 8, rest); explicit stack
    uint64_t sb[34];
    uint32_t sn[34];
    double acc[34];
    int state[34];
    int sp = 0;
    sb[0] = base;
    sn[0] = n;
    state[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        uint32_t cn = sn[sp];
        if (cn <= 128) {
            ret = vs_np_pairwise_le128(dp, nbr, sb[sp], cn);
            sp--;
            continue;
        }
        uint32_t n2 = cn / 2;
        n2 -= n2 % 8;
        if (state[sp] == 0) {
            state[sp] = 1;
            sb[sp + 1] = sb[sp];
            sn[sp + 1] = n2;
            state[sp + 1] = 0;
            sp++;
        } else if (state[sp] == 1) {
            acc[sp] = ret;
            state[sp] = 2;
            sb[sp + 1] = sb[sp] + n2;
            sn[sp + 1] = cn - n2;
            state[sp + 1] = 0;
            sp++;
        } else {
            ret = acc[sp] + ret;
            sp--;
        }
    }
    return ret;
}

// ---------------------------------------------------------------------------------------------
// K6  vertex scan over the stage CSR (adjacency row = out-entries [0, n_out) then in-entries).
//   nontrivial : is_non_trivial, Utilities.py:162-172 (black edges; ids are unique per vertex,
//                so the id-set intersection is the set of vertices that are both a black
//                in-source and a black out-target)
//   fork_kind  : the tests of Decomposition.py:715 / :763 (black vertex, black edges)
//   chain_pred : simple in-edge source (simp_path, Utilities.py:398-402: src.out_degree()==1 and
//                target.in_degree()==1 and src != target; degrees count every edge)
//   chain_next : simple out-edge target
//   out_sum / in_sum : numpy.sum of neighbour dp in adjacency order (Utilities.py:20,23)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_vertex_scan(uint32_t nv, const uint64_t *__restrict__ row_ptr,
                                                    const uint32_t *__restrict__ n_out, const uint32_t *__restrict__ nbr,
                                                    const uint32_t *__restrict__ eidx, const double *__restrict__ dp,
                                                    const uint8_t *__restrict__ vblack, const uint8_t *__restrict__ eblack,
                                                    double *__restrict__ out_sum, double *__restrict__ in_sum,
                                                    uint8_t *__restrict__ nontrivial, uint8_t *__restrict__ fork_kind,
                                                    int32_t *__restrict__ chain_next, int32_t *__restrict__ chain_pred,
                                                    uint32_t *__restrict__ big_list, uint32_t *__restrict__ big_count,
                                                    uint32_t *__restrict__ bad_init) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    if (bad_init && v == 0) *bad_init = 0xFFFFFFFFu;  // (k_edge_flow's "no zero sum" word: set here instead of by a fill of its own)
    uint64_t lo = row_ptr[v], hi = row_ptr[v + 1];
    uint32_t no = n_out[v];
    uint32_t ni = (uint32_t)(hi - lo) - no;
    // sums of more than 128 addends split recursively (numpy's pairwise_sum): rare, and the explicit stack that takes
    // costs every thread scratch memory -- such vertices are listed for k_vertex_sums_big instead
    if (no > 128u || ni > 128u) {
        big_list[atomicAdd(big_count, 1u)] = v;
    } else {
        out_sum[v] = vs_np_pairwise_le128(dp, nbr, lo, no);
        in_sum[v] = vs_np_pairwise_le128(dp, nbr, lo + no, ni);
    }
    uint32_t bin = 0, bout = 0, both = 0;
    for (uint64_t i = lo + no; i < hi; i++)
        if (eblack[eidx[i]]) bin++;
    for (uint64_t k = lo; k < lo + no; k++)
        if (eblack[eidx[k]]) bout++;
    if (bin > 1 && bout > 1) {  // (with one black edge on a side the vertex is no branch whatever the overlap is)
        for (uint64_t i = lo + no; i < hi; i++) {
            if (!eblack[eidx[i]]) continue;
            // count this source once if it is also a black out-target and was not seen earlier
            uint32_t s = nbr[i];
            bool is_target = false;
            for (uint64_t k = lo; k < lo + no; k++)
                if (nbr[k] == s && eblack[eidx[k]]) { is_target = true; break; }
            if (!is_target) continue;
            bool first = true;
            for (uint64_t k = lo + no; k < i; k++)
                if (nbr[k] == s && eblack[eidx[k]]) { first = false; break; }
            if (first) both++;
        }
    }
    uint32_t m = both > 1 ? both : 1;
    nontrivial[v] = (bin > m && bout > m) ? 1 : 0;
    uint8_t fk = 0;
    if (vblack[v]) {
        if (bin == 1 && bout > 1) fk = 1;
        else if (bin > 1 && bout == 1) fk = 2;
    }
    fork_kind[v] = fk;
    int32_t nx = -1, pd = -1;
    if (no == 1) {
        uint32_t t = nbr[lo];
        uint32_t t_in = (uint32_t)(row_ptr[t + 1] - row_ptr[t]) - n_out[t];
        if (t_in == 1 && t != v) nx = (int32_t)t;
    }
    if (ni == 1) {
        uint32_t s = nbr[lo + no];
        if (n_out[s] == 1 && s != v) pd = (int32_t)s;
    }
    chain_next[v] = nx;
    chain_pred[v] = pd;
}

// neighbour sums of the vertices k_vertex_scan listed (more than 128 neighbours on a side)
__global__ void __launch_bounds__(64) k_vertex_sums_big(const uint32_t *__restrict__ big_list, const uint32_t *__restrict__ big_count,
                                                       const uint64_t *__restrict__ row_ptr, const uint32_t *__restrict__ n_out,
                                                       const uint32_t *__restrict__ nbr, const double *__restrict__ dp,
                                                       double *__restrict__ out_sum, double *__restrict__ in_sum) {
    const uint32_t n = *big_count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t v = big_list[i];
        const uint64_t lo = row_ptr[v], hi = row_ptr[v + 1];
        const uint32_t no = n_out[v], ni = (uint32_t)(hi - lo) - no;
        out_sum[v] = vs_np_pairwise(dp, nbr, lo, no);
        in_sum[v] = vs_np_pairwise(dp, nbr, lo + no, ni);
    }
}

// K6b  list ranking of the simple chains by pointer jumping: after ceil(log2(longest chain)) rounds anc[v] = -1 and
// rank[v] = distance of v from the head of its chain, top[v] = that head.  ONE workgroup runs all the rounds (a stage
// graph has at most tens of thousands of vertices and is re-ranked several hundred times per run: a launch per round
// cost more than the rounds themselves), double buffered in global memory, and stops after the first round in which no
// vertex still had an ancestor to jump over -- chains are short, so that is a handful of rounds instead of log2(V).
// Vertices on a ring of simple edges never reach -1 and are reported with rank = -1 (the reference finds no head there
// either and leaves rings alone).  buf: 6 * stride int32 of scratch (anc / rank / top, two copies each).
#define CHAIN_TPB 1024
#define CHAIN_FIRST_ROUNDS 4u  // rounds of the multi-launch ranking that are launched before anyone looks whether more are needed
__global__ void __launch_bounds__(CHAIN_TPB) k_chain_rank(uint32_t nv, const int32_t *__restrict__ pred, int32_t *__restrict__ buf,
                                                          uint32_t stride, int32_t *__restrict__ top_out, int32_t *__restrict__ rank_out) {
    __shared__ int s_live;
    int32_t *anc[2] = {buf, buf + stride}, *rank[2] = {buf + 2 * stride, buf + 3 * stride}, *top[2] = {buf + 4 * stride, buf + 5 * stride};
    for (uint32_t v = threadIdx.x; v < nv; v += CHAIN_TPB) {
        const int32_t p = pred[v];
        anc[0][v] = p;
        rank[0][v] = p >= 0 ? 1 : 0;
        top[0][v] = p >= 0 ? p : (int32_t)v;
    }
    __syncthreads();
    int cur = 0;
    for (uint64_t span = 1; span < nv; span <<= 1) {
        if (threadIdx.x == 0) s_live = 0;
        __syncthreads();
        int live = 0;
        for (uint32_t v = threadIdx.x; v < nv; v += CHAIN_TPB) {
            const int32_t a = anc[cur][v];
            if (a >= 0) {
                rank[cur ^ 1][v] = rank[cur][v] + rank[cur][a];
                top[cur ^ 1][v] = top[cur][a];
                const int32_t aa = anc[cur][a];
                anc[cur ^ 1][v] = aa;
                live |= aa >= 0;
            } else {
                rank[cur ^ 1][v] = rank[cur][v];
                top[cur ^ 1][v] = top[cur][v];
                anc[cur ^ 1][v] = -1;
            }
        }
        if (live) s_live = 1;
        __syncthreads();
        cur ^= 1;
        if (!s_live) break;
        __syncthreads();
    }
    // (a vertex whose ancestor is still set after the last round lies on a ring)
    for (uint32_t v = threadIdx.x; v < nv; v += CHAIN_TPB) {
        top_out[v] = top[cur][v];
        rank_out[v] = anc[cur][v] >= 0 ? -1 : rank[cur][v];
    }
}

// The same ranking for graphs too large for one workgroup to be quick about (tens of thousands of vertices): one launch
// per round over all vertices, up to ceil(log2(V)) of them -- a round first looks at what the round before it reported
// (ctl[1 + r]: some vertex still had an ancestor to jump over) and returns at once when there was nothing left, so a launch
// behind the last productive round costs a few microseconds.  The stage handle enqueues CHAIN_FIRST_ROUNDS of them with the
// rest of its operation and the others only when k_chain_finish_wide reports that the last of those still found work
// (HipStageOps::run).  ctl[0] = rounds that did work (its parity names the buffer that holds the result).
__global__ void __launch_bounds__(256) k_chain_init_wide(uint32_t nv, const int32_t *__restrict__ pred, int32_t *__restrict__ buf, uint32_t stride,
                                                        int32_t *__restrict__ ctl) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < 40u) ctl[v] = v == 1u ? 1 : 0;  // ctl[1] = 1: round 0 always runs
    if (v >= nv) return;
    const int32_t p = pred[v];
    buf[v] = p;
    buf[2 * stride + v] = p >= 0 ? 1 : 0;
    buf[4 * stride + v] = p >= 0 ? p : (int32_t)v;
}
__global__ void __launch_bounds__(256) k_chain_jump_wide(uint32_t nv, uint32_t round, int32_t *__restrict__ buf, uint32_t stride,
                                                        int32_t *__restrict__ ctl) {
    if (!__hip_atomic_load(&ctl[1 + round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cur = round & 1u, nxt = cur ^ 1u;
    const int32_t *anc_in = buf + cur * stride, *rank_in = buf + (2 + cur) * stride, *top_in = buf + (4 + cur) * stride;
    int32_t *anc_out = buf + nxt * stride, *rank_out = buf + (2 + nxt) * stride, *top_out = buf + (4 + nxt) * stride;
    if (v == 0) ctl[0] = (int32_t)round + 1;
    if (v >= nv) return;
    const int32_t a = anc_in[v];
    if (a >= 0) {
        rank_out[v] = rank_in[v] + rank_in[a];
        top_out[v] = top_in[a];
        const int32_t aa = anc_in[a];
        anc_out[v] = aa;
        if (aa >= 0) __hip_atomic_store(&ctl[2 + round], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        rank_out[v] = rank_in[v];
        top_out[v] = top_in[v];
        anc_out[v] = -1;
    }
}
__global__ void __launch_bounds__(256) k_chain_finish_wide(uint32_t nv, const int32_t *__restrict__ buf, uint32_t stride,
                                                          const int32_t *__restrict__ ctl, int32_t *__restrict__ top_out,
                                                          int32_t *__restrict__ rank_out, uint32_t next_round, uint32_t *__restrict__ live_out) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    // (live_out: did the last round launched still find an ancestor to jump over -- the caller then launches the rest)
    if (live_out && v == 0) *live_out = ctl[1 + next_round] ? 1u : 0u;
    if (v >= nv) return;
    const uint32_t cur = (uint32_t)ctl[0] & 1u;
    top_out[v] = buf[(4 + cur) * stride + v];
    rank_out[v] = buf[cur * stride + v] >= 0 ? -1 : buf[(2 + cur) * stride + v];
}

// ---------------------------------------------------------------------------------------------
// K7  edge flow, Utilities.py:14-31: flow(u->v) = numpy.mean([ (dp_v/out_sum_u)*dp_u,
// (dp_u/in_sum_v)*dp_v ]) = ((0.0 + a) + b) / 2.  One thread per out-entry of the CSR.
// A zero sum is reported instead of dividing (the reference raises FloatingPointError under
// numpy.seterr(all="raise"), vstrains:25).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_edge_flow(uint32_t nv, const uint64_t *__restrict__ row_ptr,
                                                  const uint32_t *__restrict__ n_out, const uint32_t *__restrict__ nbr,
                                                  const uint32_t *__restrict__ eidx, const double *__restrict__ dp,
                                                  const double *__restrict__ out_sum, const double *__restrict__ in_sum,
                                                  double *__restrict__ flow, uint32_t *__restrict__ bad) {
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nv) return;
    uint64_t lo = row_ptr[u];
    uint32_t no = n_out[u];
    double du = dp[u], os = out_sum[u];
    for (uint32_t k = 0; k < no; k++) {
        uint32_t v = nbr[lo + k];
        double dv = dp[v], is = in_sum[v];
        if (os == 0.0 || is == 0.0) {
            atomicMin(bad, eidx[lo + k]);
            flow[eidx[lo + k]] = 0.0;
            continue;
        }
        double a = (dv / os) * du;
        double b = (du / is) * dv;
        flow[eidx[lo + k]] = (a + b) / 2.0;
    }
}

// =============================================================================================
// host entry points
// =============================================================================================
namespace {
// A view of one of the context's grow-only scratch slots: the graph stages call these entry
// points hundreds of times per run with similar sizes, so nothing is allocated after warm-up.
struct DevBuf {
    void *p = nullptr;
    template <typename T>
    T *as() { return (T *)p; }
};

int slot_reserve(vs_ctx *ctx, int slot, DevBuf &b, size_t bytes) {
    if (bytes < 16) bytes = 16;
    if (ctx->scratch_cap[slot] < bytes) {
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scratch[slot]) VS_HIP(ctx, hipFree(ctx->scratch[slot]));
        ctx->scratch[slot] = nullptr;
        ctx->scratch_cap[slot] = 0;
        size_t cap = bytes + bytes / 2;
        VS_HIP(ctx, hipMalloc(&ctx->scratch[slot], cap));
        ctx->scratch_cap[slot] = cap;
    }
    b.p = ctx->scratch[slot];
    return VS_OK;
}

struct SlotCounter {
    int next = 0;
};

int dev_upload(vs_ctx *ctx, SlotCounter &sc, DevBuf &b, const void *host, size_t bytes) {
    int rc = slot_reserve(ctx, sc.next++, b, bytes);
    if (rc) return rc;
    if (bytes) VS_HIP(ctx, hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return VS_OK;
}
int dev_alloc(vs_ctx *ctx, SlotCounter &sc, DevBuf &b, size_t bytes) { return slot_reserve(ctx, sc.next++, b, bytes); }
#define VS_TRY(x)            \
    do {                     \
        int rc__ = (x);      \
        if (rc__) return rc__; \
    } while (0)

template <typename TIn>
int links_build(vs_ctx *ctx, const TIn *d_node, const TIn *d_short, uint32_t n, vs_links **out) {
    vs_links *L = new vs_links();
    L->n = n;
    const bool timing = getenv("VS_LINKS_TIMING") != nullptr;  // where the call's host time goes, on stderr
    auto now = [] {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    };
    if (timing) (void)hipStreamSynchronize(ctx->stream);
    const double t0 = now();
    hipError_t e = hipSuccess;
    const bool reserved = ctx->links_spare && ctx->links_spare_n == n && n;
    if (reserved) {  // vs_links_reserve set it aside
        L->d_p0 = (int64_t *)ctx->links_spare;
        ctx->links_spare = nullptr;
        ctx->links_spare_n = 0;
    } else {
        e = hipMalloc((void **)&L->d_p0, (size_t)(n ? (uint64_t)n * n : 1) * sizeof(int64_t));
    }
    const double t1 = now();
    if (e != hipSuccess) {
        delete L;
        return vs_fail(ctx, VS_E_OOM, "vs_links: %s", hipGetErrorString(e));
    }
    if (n) {
        uint32_t tiles = (n + SYM_T - 1) / SYM_T;
        uint64_t pairs = (uint64_t)tiles * (tiles + 1) / 2;
        hipLaunchKernelGGL((k_links_symmetrize<TIn>), dim3((unsigned)pairs), dim3(256), 0, ctx->stream, d_node, d_short, n,
                           tiles, L->d_p0);
        e = hipGetLastError();
        if (e != hipSuccess) {
            (void)hipFree(L->d_p0);
            delete L;
            return vs_fail(ctx, VS_E_HIP, "k_links_symmetrize: %s", hipGetErrorString(e));
        }
    }
    if (timing) {
        (void)hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[vs] link table of %u nodes: %.2f GB %s %.4f s, k_links_symmetrize %.4f s\n", n, (double)n * n * 8 / 1e9,
                reserved ? "taken from vs_links_reserve" : "hipMalloc", t1 - t0, now() - t1);
    }
    *out = L;
    return VS_OK;
}
#define VS_LINKS_SPARSE_MIN 32768u  // nodes from which a table with a dirty-tile map is held as CSR rows (4 GiB of counters)
template <typename TIn>
int links_build_sparse(vs_ctx *ctx, const TIn *d_node, const TIn *d_short, uint32_t n, const uint8_t *d_tile_map, vs_links **out) {
    vs_links *L = new vs_links();
    L->n = n;
    const uint32_t T = (n + SL_T - 1u) / SL_T;
    const uint64_t pairs = (uint64_t)T * (T + 1u) / 2u;
    uint8_t *d_cnt = nullptr;
    uint32_t *d_piece = nullptr, *d_total = nullptr;
    uint64_t *d_scan_tmp = nullptr, *d_sum = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_cnt, (void *)d_piece, (void *)d_total, (void *)d_scan_tmp, (void *)d_sum})
            if (q) (void)hipFree(q);
    };
    auto fail = [&](int code, const char *what, hipError_t e) {
        cleanup();
        vs_links_free(ctx, L);
        return vs_fail(ctx, code, "vs_links (sparse): %s: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&d_cnt, (size_t)n * T)) != hipSuccess) return fail(VS_E_OOM, "piece counts", e);
    if ((e = hipMalloc((void **)&d_piece, (size_t)n * T * sizeof(uint32_t))) != hipSuccess) return fail(VS_E_OOM, "piece offsets", e);
    if ((e = hipMalloc((void **)&d_total, ((size_t)n + 2u) * sizeof(uint32_t))) != hipSuccess) return fail(VS_E_OOM, "row totals", e);
    if ((e = hipMalloc((void **)&L->d_row_ptr, ((size_t)n + 2u) * sizeof(uint32_t))) != hipSuccess) return fail(VS_E_OOM, "row_ptr", e);
    if ((e = hipMalloc((void **)&d_scan_tmp, ((size_t)n / 2048u + 8u) * sizeof(uint64_t))) != hipSuccess) return fail(VS_E_OOM, "scan", e);
    if ((e = hipMalloc((void **)&d_sum, sizeof(uint64_t))) != hipSuccess) return fail(VS_E_OOM, "sum", e);
    if ((e = hipMemsetAsync(d_cnt, 0, (size_t)n * T, ctx->stream)) != hipSuccess) return fail(VS_E_HIP, "memset", e);
    if ((e = hipMemsetAsync(d_total, 0, ((size_t)n + 2u) * sizeof(uint32_t), ctx->stream)) != hipSuccess) return fail(VS_E_HIP, "memset", e);
    hipLaunchKernelGGL((k_sl_count<TIn>), dim3((unsigned)pairs), dim3(256), 0, ctx->stream, d_node, d_short, d_tile_map, n, T, d_cnt);
    hipLaunchKernelGGL(k_sl_rows, dim3((n + 3u) / 4u), dim3(256), 0, ctx->stream, (const uint8_t *)d_cnt, n, T, d_piece, d_total);
    int rc = vs_scan_u32(ctx, d_total, L->d_row_ptr, (uint64_t)n + 1u, d_scan_tmp, d_sum);
    if (rc) { cleanup(); vs_links_free(ctx, L); return rc; }
    uint64_t nnz = 0;
    if ((e = hipMemcpyAsync(&nnz, d_sum, sizeof nnz, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return fail(VS_E_HIP, "copy", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(VS_E_HIP, "count pass", e);
    if (nnz > 0xFFFFFFF0ull) { cleanup(); vs_links_free(ctx, L); return vs_fail(ctx, VS_E_RANGE, "vs_links (sparse): %llu non-zero cells", (unsigned long long)nnz); }
    L->nnz = nnz;
    if ((e = hipMalloc((void **)&L->d_col, (size_t)(nnz + 1u) * sizeof(uint32_t))) != hipSuccess) return fail(VS_E_OOM, "columns", e);
    if ((e = hipMalloc((void **)&L->d_val, (size_t)(nnz + 1u) * sizeof(int64_t))) != hipSuccess) return fail(VS_E_OOM, "values", e);
    hipLaunchKernelGGL((k_sl_fill<TIn>), dim3((unsigned)pairs), dim3(256), 0, ctx->stream, d_node, d_short, d_tile_map, n, T, (const uint32_t *)d_piece,
                       (const uint32_t *)L->d_row_ptr, L->d_col, L->d_val);
    if ((e = hipGetLastError()) != hipSuccess) return fail(VS_E_HIP, "k_sl_fill", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(VS_E_HIP, "fill pass", e);
    cleanup();
    if (getenv("VS_LINKS_TIMING"))
        fprintf(stderr, "[vs] link table of %u nodes held as CSR rows: %llu non-zero cells, %.3f GB (dense: %.2f GB)\n", n, (unsigned long long)nnz,
                (double)nnz * 12 / 1e9, (double)n * n * 8 / 1e9);
    *out = L;
    return VS_OK;
}
}  // namespace

extern "C" {

int vs_links_reserve(vs_ctx *ctx, uint32_t n) {
    if (!ctx) return VS_E_ARG;
    VS_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->links_spare && ctx->links_spare_n == n) return VS_OK;
    if (ctx->links_spare) {
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->links_spare);
        ctx->links_spare = nullptr;
        ctx->links_spare_n = 0;
    }
    if (!n) return VS_OK;
    const hipError_t e = hipMalloc(&ctx->links_spare, (size_t)n * n * sizeof(int64_t));
    if (e != hipSuccess) {
        ctx->links_spare = nullptr;
        return vs_fail(ctx, VS_E_OOM, "vs_links_reserve: %.2f GB: %s", (double)n * n * 8 / 1e9, hipGetErrorString(e));
    }
    ctx->links_spare_n = n;
    return VS_OK;
}

int vs_links_from_counts(vs_ctx *ctx, const uint32_t *d_node_mat, const uint32_t *d_short_mat, uint32_t n, vs_links **out) {
    if (!ctx || !out || (n && (!d_node_mat || !d_short_mat))) return vs_fail(ctx, VS_E_ARG, "vs_links_from_counts: bad argument");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    return links_build<uint32_t>(ctx, d_node_mat, d_short_mat, n, out);
}

int vs_links_from_counts_tracked(vs_ctx *ctx, const uint32_t *d_node_mat, const uint32_t *d_short_mat, uint32_t n, const uint8_t *d_tile_map,
                                 uint32_t sparse_min_nodes, vs_links **out) {
    if (!ctx || !out || (n && (!d_node_mat || !d_short_mat))) return vs_fail(ctx, VS_E_ARG, "vs_links_from_counts_tracked: bad argument");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t min_nodes = sparse_min_nodes ? sparse_min_nodes : VS_LINKS_SPARSE_MIN;
    if (!d_tile_map || n < min_nodes || n < SL_T) return links_build<uint32_t>(ctx, d_node_mat, d_short_mat, n, out);
    if (ctx->links_spare) {  // (a dense buffer set aside earlier is not needed: give it back)
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->links_spare);
        ctx->links_spare = nullptr;
        ctx->links_spare_n = 0;
    }
    return links_build_sparse<uint32_t>(ctx, d_node_mat, d_short_mat, n, d_tile_map, out);
}

int vs_links_from_wide(vs_ctx *ctx, const int64_t *d_node_mat, const int64_t *d_short_mat, uint32_t n, vs_links **out) {
    if (!ctx || !out || (n && (!d_node_mat || !d_short_mat))) return vs_fail(ctx, VS_E_ARG, "vs_links_from_wide: bad argument");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    return links_build<int64_t>(ctx, d_node_mat, d_short_mat, n, out);
}

int vs_links_from_host(vs_ctx *ctx, const int64_t *node_mat, const int64_t *short_mat, uint32_t n, vs_links **out) {
    if (!ctx || !out || (n && (!node_mat || !short_mat))) return vs_fail(ctx, VS_E_ARG, "vs_links_from_host: bad argument");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    SlotCounter sc;
    DevBuf a, b;
    size_t bytes = (size_t)n * n * sizeof(int64_t);
    VS_TRY(dev_upload(ctx, sc, a, node_mat, bytes));
    VS_TRY(dev_upload(ctx, sc, b, short_mat, bytes));
    int rc = links_build<int64_t>(ctx, a.as<int64_t>(), b.as<int64_t>(), n, out);
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

void vs_links_free(vs_ctx *ctx, vs_links *links) {
    if (!links) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (links->d_p0) (void)hipFree(links->d_p0);
    if (links->d_row_ptr) (void)hipFree(links->d_row_ptr);
    if (links->d_col) (void)hipFree(links->d_col);
    if (links->d_val) (void)hipFree(links->d_val);
    delete links;
}

int vs_links_size(const vs_links *links, uint32_t *n) {
    if (!links || !n) return VS_E_ARG;
    *n = links->n;
    return VS_OK;
}

int vs_links_to_host(vs_ctx *ctx, const vs_links *links, int64_t *out) {
    if (!ctx || !links || !out) return vs_fail(ctx, VS_E_ARG, "vs_links_to_host: bad argument");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    if (links->sparse()) {  // (the dense matrix the caller asked for, filled from the rows)
        const uint32_t n = links->n;
        std::vector<uint32_t> rp((size_t)n + 1u), col((size_t)links->nnz);
        std::vector<int64_t> val((size_t)links->nnz);
        VS_HIP(ctx, hipMemcpyAsync(rp.data(), links->d_row_ptr, rp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (links->nnz) {
            VS_HIP(ctx, hipMemcpyAsync(col.data(), links->d_col, col.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            VS_HIP(ctx, hipMemcpyAsync(val.data(), links->d_val, val.size() * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        }
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memset(out, 0, (size_t)n * n * sizeof(int64_t));
        for (uint32_t r = 0; r < n; r++)
            for (uint32_t i = rp[r]; i < rp[r + 1]; i++) out[(size_t)r * n + col[i]] = val[i];
        return VS_OK;
    }
    VS_HIP(ctx, hipMemcpyAsync(out, links->d_p0, (size_t)links->n * links->n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VS_OK;
}

static int check_lists(vs_ctx *ctx, const vs_links *links, const uint64_t *list_off, const uint32_t *list_idx, uint32_t n_lists) {
    for (uint32_t l = 0; l < n_lists; l++)
        if (list_off[l + 1] < list_off[l]) return vs_fail(ctx, VS_E_ARG, "list offsets must not decrease");
    for (uint64_t i = 0; i < list_off[n_lists]; i++)
        if (list_idx[i] >= links->n) return vs_fail(ctx, VS_E_RANGE, "list index %u out of range (n=%u)", list_idx[i], links->n);
    return VS_OK;
}

int vs_links_block_sums(vs_ctx *ctx, const vs_links *links, const uint64_t *list_off, const uint32_t *list_idx,
                        uint32_t n_lists, const uint32_t *qa, const uint32_t *qb, uint64_t n_queries, int64_t *out) {
    if (!ctx || !links || !list_off || (n_queries && (!qa || !qb || !out)))
        return vs_fail(ctx, VS_E_ARG, "vs_links_block_sums: bad argument");
    if (n_queries == 0) return VS_OK;
    if (list_off[n_lists] && !list_idx) return vs_fail(ctx, VS_E_ARG, "vs_links_block_sums: list_idx is NULL");
    VS_TRY(check_lists(ctx, links, list_off, list_idx, n_lists));
    for (uint64_t q = 0; q < n_queries; q++)
        if (qa[q] >= n_lists || qb[q] >= n_lists) return vs_fail(ctx, VS_E_RANGE, "query %llu names a list out of range", (unsigned long long)q);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    SlotCounter sc;
    DevBuf d_off, d_idx, d_qa, d_qb, d_out;
    VS_TRY(dev_upload(ctx, sc, d_off, list_off, (size_t)(n_lists + 1) * sizeof(uint64_t)));
    VS_TRY(dev_upload(ctx, sc, d_idx, list_idx, (size_t)list_off[n_lists] * sizeof(uint32_t)));
    VS_TRY(dev_upload(ctx, sc, d_qa, qa, (size_t)n_queries * sizeof(uint32_t)));
    VS_TRY(dev_upload(ctx, sc, d_qb, qb, (size_t)n_queries * sizeof(uint32_t)));
    VS_TRY(dev_alloc(ctx, sc, d_out, (size_t)n_queries * sizeof(int64_t)));
    const unsigned waves = 256 / VS_WAVE;
    if (links->sparse())
        hipLaunchKernelGGL(k_links_block_sums_csr, dim3((unsigned)((n_queries + waves - 1) / waves)), dim3(256), 0, ctx->stream,
                           (const uint32_t *)links->d_row_ptr, (const uint32_t *)links->d_col, (const int64_t *)links->d_val, d_off.as<uint64_t>(),
                           d_idx.as<uint32_t>(), d_qa.as<uint32_t>(), d_qb.as<uint32_t>(), n_queries, d_out.as<int64_t>());
    else
    hipLaunchKernelGGL(k_links_block_sums, dim3((unsigned)((n_queries + waves - 1) / waves)), dim3(256), 0, ctx->stream,
                       links->d_p0, links->n, d_off.as<uint64_t>(), d_idx.as<uint32_t>(), d_qa.as<uint32_t>(),
                       d_qb.as<uint32_t>(), n_queries, d_out.as<int64_t>());
    VS_HIP(ctx, hipGetLastError());
    VS_HIP(ctx, hipMemcpyAsync(out, d_out.p, (size_t)n_queries * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VS_OK;
}

int vs_links_group_matrix(vs_ctx *ctx, const vs_links *links, const uint64_t *list_off, const uint32_t *list_idx,
                          uint32_t n_groups, int64_t *out) {
    if (!ctx || !links || !list_off || (n_groups && !out)) return vs_fail(ctx, VS_E_ARG, "vs_links_group_matrix: bad argument");
    if (n_groups == 0) return VS_OK;
    if (list_off[n_groups] && !list_idx) return vs_fail(ctx, VS_E_ARG, "vs_links_group_matrix: list_idx is NULL");
    VS_TRY(check_lists(ctx, links, list_off, list_idx, n_groups));
    VS_HIP(ctx, hipSetDevice(ctx->device));
    if (links->sparse()) {  // every (g, h) pair as a block-sum query over the rows
        std::vector<uint32_t> qa((size_t)n_groups * n_groups), qb(qa.size());
        for (uint32_t g = 0; g < n_groups; g++)
            for (uint32_t h = 0; h < n_groups; h++) { qa[(size_t)g * n_groups + h] = g; qb[(size_t)g * n_groups + h] = h; }
        return vs_links_block_sums(ctx, links, list_off, list_idx, n_groups, qa.data(), qb.data(), qa.size(), out);
    }
    uint32_t n = links->n;
    SlotCounter sc;
    DevBuf d_off, d_idx, d_t, d_out;
    VS_TRY(dev_upload(ctx, sc, d_off, list_off, (size_t)(n_groups + 1) * sizeof(uint64_t)));
    VS_TRY(dev_upload(ctx, sc, d_idx, list_idx, (size_t)list_off[n_groups] * sizeof(uint32_t)));
    VS_TRY(dev_alloc(ctx, sc, d_t, (size_t)n_groups * (n ? n : 1) * sizeof(int64_t)));
    VS_TRY(dev_alloc(ctx, sc, d_out, (size_t)n_groups * n_groups * sizeof(int64_t)));
    if (n) {
        hipLaunchKernelGGL(k_links_group_rows, dim3((n + 255) / 256, n_groups), dim3(256), 0, ctx->stream, links->d_p0, n,
                           d_off.as<uint64_t>(), d_idx.as<uint32_t>(), d_t.as<int64_t>());
    }
    uint64_t pairs = (uint64_t)n_groups * n_groups;
    const unsigned waves = 256 / VS_WAVE;
    hipLaunchKernelGGL(k_links_group_cols, dim3((unsigned)((pairs + waves - 1) / waves)), dim3(256), 0, ctx->stream,
                       d_t.as<int64_t>(), n, n_groups, d_off.as<uint64_t>(), d_idx.as<uint32_t>(), d_out.as<int64_t>());
    VS_HIP(ctx, hipGetLastError());
    VS_HIP(ctx, hipMemcpyAsync(out, d_out.p, (size_t)pairs * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VS_OK;
}

int vs_graph_refresh(vs_ctx *ctx, uint32_t n_vertices, uint32_t n_edge_slots, const uint64_t *row_ptr, const uint32_t *n_out,
                     const uint32_t *nbr, const uint32_t *eidx, const double *dp, const uint8_t *vertex_black,
                     const uint8_t *edge_black, double *flow, uint8_t *nontrivial, uint8_t *fork_kind,
                     int32_t *chain_next, int32_t *chain_top, int32_t *chain_rank, uint32_t *zero_sum_edge) {
    if (!ctx || !row_ptr || (n_vertices && (!n_out || !dp || !vertex_black)))
        return vs_fail(ctx, VS_E_ARG, "vs_graph_refresh: bad argument");
    if (zero_sum_edge) *zero_sum_edge = 0xFFFFFFFFu;
    if (n_vertices == 0) return VS_OK;
    uint64_t n_adj = row_ptr[n_vertices];
    if (n_adj && (!nbr || !eidx || !edge_black)) return vs_fail(ctx, VS_E_ARG, "vs_graph_refresh: adjacency arrays missing");
    for (uint32_t v = 0; v < n_vertices; v++) {
        if (row_ptr[v + 1] < row_ptr[v] || n_out[v] > row_ptr[v + 1] - row_ptr[v])
            return vs_fail(ctx, VS_E_ARG, "vs_graph_refresh: malformed row %u", v);
    }
    for (uint64_t i = 0; i < n_adj; i++)
        if (nbr[i] >= n_vertices || eidx[i] >= n_edge_slots) return vs_fail(ctx, VS_E_RANGE, "vs_graph_refresh: adjacency entry %llu out of range", (unsigned long long)i);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    SlotCounter sc;
    DevBuf d_row, d_no, d_nbr, d_eidx, d_dp, d_vb, d_eb, d_os, d_is, d_nt, d_fk, d_nx, d_pd, d_flow, d_bad;
    DevBuf d_chain, d_top, d_rank, d_big;
    VS_TRY(dev_upload(ctx, sc, d_row, row_ptr, (size_t)(n_vertices + 1) * sizeof(uint64_t)));
    VS_TRY(dev_upload(ctx, sc, d_no, n_out, (size_t)n_vertices * sizeof(uint32_t)));
    VS_TRY(dev_upload(ctx, sc, d_nbr, nbr, (size_t)n_adj * sizeof(uint32_t)));
    VS_TRY(dev_upload(ctx, sc, d_eidx, eidx, (size_t)n_adj * sizeof(uint32_t)));
    VS_TRY(dev_upload(ctx, sc, d_dp, dp, (size_t)n_vertices * sizeof(double)));
    VS_TRY(dev_upload(ctx, sc, d_vb, vertex_black, (size_t)n_vertices));
    VS_TRY(dev_upload(ctx, sc, d_eb, edge_black, (size_t)n_edge_slots));
    VS_TRY(dev_alloc(ctx, sc, d_os, (size_t)n_vertices * sizeof(double)));
    VS_TRY(dev_alloc(ctx, sc, d_is, (size_t)n_vertices * sizeof(double)));
    VS_TRY(dev_alloc(ctx, sc, d_nt, n_vertices));
    VS_TRY(dev_alloc(ctx, sc, d_fk, n_vertices));
    VS_TRY(dev_alloc(ctx, sc, d_nx, (size_t)n_vertices * sizeof(int32_t)));
    VS_TRY(dev_alloc(ctx, sc, d_pd, (size_t)n_vertices * sizeof(int32_t)));
    VS_TRY(dev_alloc(ctx, sc, d_flow, (size_t)(n_edge_slots ? n_edge_slots : 1) * sizeof(double)));
    VS_TRY(dev_alloc(ctx, sc, d_bad, sizeof(uint32_t)));
    VS_TRY(dev_alloc(ctx, sc, d_big, (size_t)(n_vertices + 1) * sizeof(uint32_t)));
    VS_HIP(ctx, hipMemsetAsync(d_big.p, 0, sizeof(uint32_t), ctx->stream));
    const uint32_t chain_stride = (n_vertices + 3u) & ~3u;
    VS_TRY(dev_alloc(ctx, sc, d_chain, (size_t)6 * chain_stride * sizeof(int32_t)));
    VS_TRY(dev_alloc(ctx, sc, d_top, (size_t)n_vertices * sizeof(int32_t)));
    VS_TRY(dev_alloc(ctx, sc, d_rank, (size_t)n_vertices * sizeof(int32_t)));
    VS_HIP(ctx, hipMemsetAsync(d_bad.p, 0xFF, sizeof(uint32_t), ctx->stream));
    VS_HIP(ctx, hipMemsetAsync(d_flow.p, 0, (size_t)(n_edge_slots ? n_edge_slots : 1) * sizeof(double), ctx->stream));
    dim3 grid((n_vertices + 255) / 256), block(256);
    hipLaunchKernelGGL(k_vertex_scan, grid, block, 0, ctx->stream, n_vertices, d_row.as<uint64_t>(), d_no.as<uint32_t>(),
                       d_nbr.as<uint32_t>(), d_eidx.as<uint32_t>(), d_dp.as<double>(), d_vb.as<uint8_t>(), d_eb.as<uint8_t>(),
                       d_os.as<double>(), d_is.as<double>(), d_nt.as<uint8_t>(), d_fk.as<uint8_t>(), d_nx.as<int32_t>(),
                       d_pd.as<int32_t>(), d_big.as<uint32_t>() + 1, d_big.as<uint32_t>(), (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_vertex_sums_big, dim3(64), dim3(64), 0, ctx->stream, d_big.as<uint32_t>() + 1, d_big.as<uint32_t>(), d_row.as<uint64_t>(),
                       d_no.as<uint32_t>(), d_nbr.as<uint32_t>(), d_dp.as<double>(), d_os.as<double>(), d_is.as<double>());
    hipLaunchKernelGGL(k_edge_flow, grid, block, 0, ctx->stream, n_vertices, d_row.as<uint64_t>(), d_no.as<uint32_t>(),
                       d_nbr.as<uint32_t>(), d_eidx.as<uint32_t>(), d_dp.as<double>(), d_os.as<double>(), d_is.as<double>(),
                       d_flow.as<double>(), d_bad.as<uint32_t>());
    hipLaunchKernelGGL(k_chain_rank, dim3(1), dim3(CHAIN_TPB), 0, ctx->stream, n_vertices, d_pd.as<int32_t>(), d_chain.as<int32_t>(),
                       chain_stride, d_top.as<int32_t>(), d_rank.as<int32_t>());
    VS_HIP(ctx, hipGetLastError());
    uint32_t bad = 0xFFFFFFFFu;
    if (flow && n_edge_slots) VS_HIP(ctx, hipMemcpyAsync(flow, d_flow.p, (size_t)n_edge_slots * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (nontrivial) VS_HIP(ctx, hipMemcpyAsync(nontrivial, d_nt.p, n_vertices, hipMemcpyDeviceToHost, ctx->stream));
    if (fork_kind) VS_HIP(ctx, hipMemcpyAsync(fork_kind, d_fk.p, n_vertices, hipMemcpyDeviceToHost, ctx->stream));
    if (chain_next) VS_HIP(ctx, hipMemcpyAsync(chain_next, d_nx.p, (size_t)n_vertices * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (chain_top) VS_HIP(ctx, hipMemcpyAsync(chain_top, d_top.p, (size_t)n_vertices * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (chain_rank) VS_HIP(ctx, hipMemcpyAsync(chain_rank, d_rank.p, (size_t)n_vertices * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(ctx, hipMemcpyAsync(&bad, d_bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (zero_sum_edge) *zero_sum_edge = bad;
    return VS_OK;
}

}  // extern "C"

// =============================================================================================
// Native stage handle (vs_stage.h / vs_stage.cpp): its three device operations on this context.
// =============================================================================================
#include "vs_stage.h"

namespace {
// One re-initialised stage graph per call, several hundred calls per run: everything the kernels need travels in ONE
// pinned upload and everything they produce in ONE pinned download; device and staging buffers are kept and grow only.
struct HipStageOps : VsStageOps {
    vs_ctx *ctx;
    const vs_links *links;
    void *h_up = nullptr, *h_down = nullptr, *d_up = nullptr, *d_down = nullptr, *d_tmp = nullptr, *d_ones = nullptr;
    size_t cap_up = 0, cap_down = 0, cap_tmp = 0, cap_ones = 0;
    double t_pack = 0, t_enqueue = 0, t_wait = 0, t_unpack = 0;  // where a call's host time goes (VS_STAGE_OP_TIMING=1 prints the sums)
    uint64_t n_calls = 0, n_second_waits = 0;
    static double now() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    HipStageOps(vs_ctx *c, const vs_links *l) : ctx(c), links(l) {}
    ~HipStageOps() override {
        if (getenv("VS_STAGE_OP_TIMING"))
            fprintf(stderr, "[vs] stage flow/scan operation: %llu calls (%llu with a second wait for long chains), pack %.4f s, enqueue %.4f s, wait %.4f s, unpack %.4f s\n",
                    (unsigned long long)n_calls, (unsigned long long)n_second_waits, t_pack, t_enqueue, t_wait, t_unpack);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if (h_up) (void)hipHostFree(h_up);
        if (h_down) (void)hipHostFree(h_down);
        if (d_up) (void)hipFree(d_up);
        if (d_down) (void)hipFree(d_down);
        if (d_tmp) (void)hipFree(d_tmp);
        if (d_ones) (void)hipFree(d_ones);
    }
    static size_t up8(size_t x) { return (x + 7u) & ~(size_t)7u; }
    int grow(void **host, void **dev, size_t *cap, size_t need) {
        if (*cap >= need) return VS_OK;
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (host && *host) { VS_HIP(ctx, hipHostFree(*host)); *host = nullptr; }
        if (*dev) { VS_HIP(ctx, hipFree(*dev)); *dev = nullptr; }
        *cap = 0;
        const size_t c = need + need / 2 + 4096;
        if (host) VS_HIP(ctx, hipHostMalloc(host, c, hipHostMallocDefault));
        VS_HIP(ctx, hipMalloc(dev, c));
        *cap = c;
        return VS_OK;
    }
    int run(uint32_t nv, uint32_t ne, const uint64_t *row_ptr, const uint32_t *n_out, const uint32_t *nbr, const uint32_t *eidx,
            const double *dp, double *flow, uint8_t *nontrivial, uint8_t *fork_kind, int32_t *chain_next, int32_t *chain_top,
            int32_t *chain_rank, uint32_t *zero_sum_edge) {
        *zero_sum_edge = 0xFFFFFFFFu;
        if (nv == 0) return VS_OK;
        VS_HIP(ctx, hipSetDevice(ctx->device));
        const uint64_t n_adj = row_ptr[nv];
        // upload: row_ptr | dp | n_out | nbr | eidx
        const size_t o_row = 0, o_dp = o_row + up8((size_t)(nv + 1) * 8), o_no = o_dp + (size_t)nv * 8, o_nbr = o_no + up8((size_t)nv * 4),
                     o_eidx = o_nbr + up8((size_t)n_adj * 4), up_bytes = o_eidx + up8((size_t)n_adj * 4);
        // download: flow | chain_next | chain_top | chain_rank | bad | nontrivial | fork_kind  (top / rank are copied in on the device)
        const size_t q_flow = 0, q_next = q_flow + (size_t)ne * 8, q_top = q_next + up8((size_t)nv * 4), q_rank = q_top + up8((size_t)nv * 4),
                     q_bad = q_rank + up8((size_t)nv * 4), q_nt = q_bad + 8, q_fk = q_nt + up8(nv), down_bytes = q_fk + up8(nv);
        // device scratch: out_sum | in_sum | pred | chain ranking (anc / rank / top, two copies each)
        const uint32_t chain_stride = (nv + 3u) & ~3u;
        const size_t t_os = 0, t_is = t_os + (size_t)nv * 8, t_pd = t_is + (size_t)nv * 8, t_pj = t_pd + up8((size_t)nv * 4),
                     t_ctl = t_pj + (size_t)6 * chain_stride * 4, t_big = t_ctl + 64 * 4, tmp_bytes = t_big + ((size_t)nv + 2) * 4;
        VS_TRY(grow(&h_up, &d_up, &cap_up, up_bytes));
        VS_TRY(grow(&h_down, &d_down, &cap_down, down_bytes));
        VS_TRY(grow(nullptr, &d_tmp, &cap_tmp, tmp_bytes));
        const size_t need_ones = (size_t)(nv > ne ? nv : ne) + 1;
        if (cap_ones < need_ones) {
            VS_TRY(grow(nullptr, &d_ones, &cap_ones, need_ones));
            VS_HIP(ctx, hipMemsetAsync(d_ones, 1, cap_ones, ctx->stream));
        }
        char *hu = (char *)h_up, *du = (char *)d_up, *dd = (char *)d_down, *dt = (char *)d_tmp;
        const double t0 = now();
        n_calls++;
        memcpy(hu + o_row, row_ptr, (size_t)(nv + 1) * 8);
        memcpy(hu + o_dp, dp, (size_t)nv * 8);
        memcpy(hu + o_no, n_out, (size_t)nv * 4);
        bool any_big = false;
        for (uint32_t v = 0; v < nv; v++) {
            const uint64_t no = n_out[v], ni = row_ptr[v + 1] - row_ptr[v] - no;
            any_big |= (no > 128u) | (ni > 128u);
        }
        if (n_adj) {
            memcpy(hu + o_nbr, nbr, (size_t)n_adj * 4);
            memcpy(hu + o_eidx, eidx, (size_t)n_adj * 4);
        }
        const double t1 = now();
        VS_HIP(ctx, hipMemcpyAsync(d_up, h_up, up_bytes, hipMemcpyHostToDevice, ctx->stream));
        // rows of more than 128 neighbours on a side are rare: without one, neither their list's counter nor the kernel that
        // sums them is needed (the "no zero sum" word is set by the scan kernel) -- three device operations less per stage graph
        if (any_big) VS_HIP(ctx, hipMemsetAsync(dt + t_big, 0, 4, ctx->stream));
        const uint64_t *d_row = (const uint64_t *)(du + o_row);
        const double *d_dp = (const double *)(du + o_dp);
        const uint32_t *d_no = (const uint32_t *)(du + o_no), *d_nbr = (const uint32_t *)(du + o_nbr), *d_eidx = (const uint32_t *)(du + o_eidx);
        double *d_os = (double *)(dt + t_os), *d_is = (double *)(dt + t_is);
        int32_t *d_pd = (int32_t *)(dt + t_pd);
        dim3 grid((nv + 255) / 256), block(256);
        hipLaunchKernelGGL(k_vertex_scan, grid, block, 0, ctx->stream, nv, d_row, d_no, d_nbr, d_eidx, d_dp, (const uint8_t *)d_ones,
                           (const uint8_t *)d_ones, d_os, d_is, (uint8_t *)(dd + q_nt), (uint8_t *)(dd + q_fk), (int32_t *)(dd + q_next), d_pd,
                           (uint32_t *)(dt + t_big) + 1, (uint32_t *)(dt + t_big), (uint32_t *)(dd + q_bad));
        if (any_big)
            hipLaunchKernelGGL(k_vertex_sums_big, dim3(64), dim3(64), 0, ctx->stream, (const uint32_t *)(dt + t_big) + 1,
                               (const uint32_t *)(dt + t_big), d_row, d_no, d_nbr, d_dp, d_os, d_is);
        hipLaunchKernelGGL(k_edge_flow, grid, block, 0, ctx->stream, nv, d_row, d_no, d_nbr, d_eidx, d_dp, d_os, d_is, (double *)(dd + q_flow),
                           (uint32_t *)(dd + q_bad));
        uint32_t rounds_total = 0, rounds_done = 0;
        int32_t *buf = (int32_t *)(dt + t_pj), *ctl = (int32_t *)(dt + t_ctl);
        if (nv <= 8192u) {
            hipLaunchKernelGGL(k_chain_rank, dim3(1), dim3(CHAIN_TPB), 0, ctx->stream, nv, d_pd, (int32_t *)(dt + t_pj), chain_stride,
                               (int32_t *)(dd + q_top), (int32_t *)(dd + q_rank));
        } else {
            hipLaunchKernelGGL(k_chain_init_wide, grid, block, 0, ctx->stream, nv, d_pd, buf, chain_stride, ctl);
            // ceil(log2(V)) rounds rank any chain, but a stage graph's chains are short (they were contracted the pass before):
            // four rounds (chains of up to 16) are launched, the finish kernel reports whether the fourth still found work, and
            // only then are the others launched (a second wait; the long paths of tests and of uncontracted input graphs)
            for (uint64_t span = 1; span < nv && rounds_total < 36u; span <<= 1) rounds_total++;
            uint32_t round = 0;
            for (; round < rounds_total && round < CHAIN_FIRST_ROUNDS; round++)
                hipLaunchKernelGGL(k_chain_jump_wide, grid, block, 0, ctx->stream, nv, round, buf, chain_stride, ctl);
            rounds_done = round;
            hipLaunchKernelGGL(k_chain_finish_wide, grid, block, 0, ctx->stream, nv, buf, chain_stride, ctl, (int32_t *)(dd + q_top),
                               (int32_t *)(dd + q_rank), round, (uint32_t *)(dd + q_bad) + 1);
        }
        VS_HIP(ctx, hipGetLastError());
        VS_HIP(ctx, hipMemcpyAsync(h_down, d_down, down_bytes, hipMemcpyDeviceToHost, ctx->stream));
        const double t2 = now();
        VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const char *hd = (const char *)h_down;
        if (rounds_done < rounds_total) {
            uint32_t live = 0;
            memcpy(&live, hd + q_bad + 4, 4);
            if (live) {  // a chain of more than 2^CHAIN_FIRST_ROUNDS vertices: the other rounds, ranks and heads again
                n_second_waits++;
                for (uint32_t round = rounds_done; round < rounds_total; round++)
                    hipLaunchKernelGGL(k_chain_jump_wide, grid, block, 0, ctx->stream, nv, round, buf, chain_stride, ctl);
                hipLaunchKernelGGL(k_chain_finish_wide, grid, block, 0, ctx->stream, nv, buf, chain_stride, ctl, (int32_t *)(dd + q_top),
                                   (int32_t *)(dd + q_rank), rounds_total, (uint32_t *)nullptr);
                VS_HIP(ctx, hipGetLastError());
                VS_HIP(ctx, hipMemcpyAsync((char *)h_down + q_top, dd + q_top, (q_bad - q_top), hipMemcpyDeviceToHost, ctx->stream));
                VS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
        }
        const double t3 = now();
        if (ne) memcpy(flow, hd + q_flow, (size_t)ne * 8);
        memcpy(chain_next, hd + q_next, (size_t)nv * 4);
        memcpy(chain_top, hd + q_top, (size_t)nv * 4);
        memcpy(chain_rank, hd + q_rank, (size_t)nv * 4);
        memcpy(zero_sum_edge, hd + q_bad, 4);
        memcpy(nontrivial, hd + q_nt, nv);
        memcpy(fork_kind, hd + q_fk, nv);
        t_pack += t1 - t0;
        t_enqueue += t2 - t1;
        t_wait += t3 - t2;
        t_unpack += now() - t3;
        return VS_OK;
    }
    int refresh(uint32_t nv, uint32_t ne, const uint64_t *row_ptr, const uint32_t *n_out, const uint32_t *nbr, const uint32_t *eidx,
                const double *dp, double *flow, uint8_t *nontrivial, uint8_t *fork_kind, int32_t *chain_next, int32_t *chain_top,
                int32_t *chain_rank, uint32_t *zero_sum_edge, std::string &err) override {
        int rc = run(nv, ne, row_ptr, n_out, nbr, eidx, dp, flow, nontrivial, fork_kind, chain_next, chain_top, chain_rank, zero_sum_edge);
        if (rc) err = ctx->err;
        return rc;
    }
    uint32_t link_rows() const override { return links ? links->n : 0u; }
    int block_sums(const uint64_t *list_off, const uint32_t *list_idx, uint32_t n_lists, const uint32_t *qa, const uint32_t *qb,
                   uint64_t n_queries, int64_t *out, std::string &err) override {
        int rc = vs_links_block_sums(ctx, links, list_off, list_idx, n_lists, qa, qb, n_queries, out);
        if (rc) err = ctx->err;
        return rc;
    }
    int group_matrix(const uint64_t *list_off, const uint32_t *list_idx, uint32_t n_groups, int64_t *out, std::string &err) override {
        int rc = vs_links_group_matrix(ctx, links, list_off, list_idx, n_groups, out);
        if (rc) err = ctx->err;
        return rc;
    }
};
}  // namespace

extern "C" int vs_stage_create(vs_ctx *ctx, const vs_links *links, vs_stage **out) {
    if (!ctx || !links || !out) return vs_fail(ctx, VS_E_ARG, "vs_stage_create: bad argument");
    *out = vs_stage_make(new HipStageOps(ctx, links));
    return VS_OK;
}
