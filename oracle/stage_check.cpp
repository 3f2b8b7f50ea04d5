// CHECKER (test infrastructure, not product code): the three data-parallel operations of the graph stages on the CPU,
// so that the native stage engine (vstrains_amd/csrc/vs_stage.cpp, compiled into THIS test library unchanged) can be
// run against the golden cases on a box without a GPU.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load oracle/_build/libvs_stage_check.so; the product library holds the HIP implementation of the same interface
// (vstrains_amd/csrc/vs_graph.hip) and no CPU one.
//
// Restated here, in the reference's own terms:
//   assign_edge_flow             utils/VStrains_Utilities.py:14-31 (numpy.sum over neighbours in adjacency order, numpy.mean
//                                of the two products; a zero sum raises under numpy.seterr(all="raise"), vstrains:25)
//   is_non_trivial               utils/VStrains_Utilities.py:162-172
//   fork tests                   utils/VStrains_Decomposition.py:715,763
//   simple edges / simple paths  utils/VStrains_Utilities.py:383-418
//   pe_info sums                 utils/VStrains_IO.py:598-627 (the symmetrised table) summed over two index lists
// Pinned by tests/golden/graph/* through the pipeline tests and compared with oracle/graph_ops.py (numpy) in
// tests/test_native_stage_cpu.py.
#include <stdint.h>

#include <string>
#include <vector>

#include "../include/vstrains_hip.h"
#include "../vstrains_amd/csrc/vs_stage.h"

namespace {

// numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src) behind the reduction's identity 0.0
double pairwise(const std::vector<double> &a, size_t lo, size_t n) {
    if (n < 8) {
        double res = 0.;
        for (size_t i = 0; i < n; i++) res += a[lo + i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[lo + j];
        size_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[lo + i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[lo + i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise(a, lo, n2) + pairwise(a, lo + n2, n - n2);
}
double numpy_sum(const std::vector<double> &a) { return 0.0 + pairwise(a, 0, a.size()); }

struct CheckOps : VsStageOps {
    std::vector<int64_t> p0;  // symmetric [n, n] -- or, for graphs whose dense table would not fit a host (54 k nodes: 23.7 GB),
    std::vector<uint64_t> rp;  // its non-zero cells as CSR rows (columns ascending): vs_stage_check_create_sparse
    std::vector<uint32_t> ci;
    std::vector<int64_t> cv;
    uint32_t n = 0;
    int64_t cell(uint32_t i, uint32_t j) const {
        if (rp.empty()) return p0[(uint64_t)i * n + j];
        uint64_t lo = rp[i], hi = rp[i + 1];
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (ci[mid] < j) lo = mid + 1; else hi = mid;
        }
        return lo < rp[i + 1] && ci[lo] == j ? cv[lo] : 0;
    }

    int refresh(uint32_t nv, uint32_t ne, const uint64_t *row_ptr, const uint32_t *n_out, const uint32_t *nbr, const uint32_t *eidx,
                const double *dp, double *flow, uint8_t *nontrivial, uint8_t *fork_kind, int32_t *chain_next, int32_t *chain_top,
                int32_t *chain_rank, uint32_t *zero_sum_edge, std::string &) override {
        uint32_t bad = 0xFFFFFFFFu;
        std::vector<double> out_sum(nv), in_sum(nv);
        for (uint32_t v = 0; v < nv; v++) {
            std::vector<double> o, i;
            for (uint64_t k = row_ptr[v]; k < row_ptr[v] + n_out[v]; k++) o.push_back(dp[nbr[k]]);
            for (uint64_t k = row_ptr[v] + n_out[v]; k < row_ptr[v + 1]; k++) i.push_back(dp[nbr[k]]);
            out_sum[v] = numpy_sum(o);
            in_sum[v] = numpy_sum(i);
        }
        for (uint32_t e = 0; e < ne; e++) flow[e] = 0.0;
        std::vector<uint8_t> has_simple_in(nv, 0);
        for (uint32_t u = 0; u < nv; u++) {
            for (uint64_t k = row_ptr[u]; k < row_ptr[u] + n_out[u]; k++) {
                const uint32_t v = nbr[k], e = eidx[k];
                if (out_sum[u] == 0.0 || in_sum[v] == 0.0) {
                    if (e < bad) bad = e;
                    continue;
                }
                const double a = (dp[v] / out_sum[u]) * dp[u], b = (dp[u] / in_sum[v]) * dp[v];
                flow[e] = ((0.0 + a) + b) / 2.0;  // numpy.mean of the two
            }
            // every vertex and edge of a re-initialised graph is black
            const uint32_t no = n_out[u], ni = (uint32_t)(row_ptr[u + 1] - row_ptr[u]) - no;
            uint32_t both = 0;
            for (uint64_t i = row_ptr[u] + no; i < row_ptr[u + 1]; i++) {
                bool first = true;
                for (uint64_t j = row_ptr[u] + no; j < i; j++)
                    if (nbr[j] == nbr[i]) first = false;
                if (!first) continue;
                for (uint64_t j = row_ptr[u]; j < row_ptr[u] + no; j++)
                    if (nbr[j] == nbr[i]) { both++; break; }
            }
            const uint32_t m = both > 1 ? both : 1;
            nontrivial[u] = (ni > m && no > m) ? 1 : 0;
            fork_kind[u] = (ni == 1 && no > 1) ? 1 : (ni > 1 && no == 1) ? 2 : 0;
            chain_next[u] = -1;
            if (no == 1) {
                const uint32_t t = nbr[row_ptr[u]];
                const uint32_t t_in = (uint32_t)(row_ptr[t + 1] - row_ptr[t]) - n_out[t];
                if (t_in == 1 && t != u) chain_next[u] = (int32_t)t;
            }
        }
        for (uint32_t u = 0; u < nv; u++)
            if (chain_next[u] >= 0) has_simple_in[chain_next[u]] = 1;
        for (uint32_t v = 0; v < nv; v++) { chain_top[v] = (int32_t)v; chain_rank[v] = 0; }
        for (uint32_t v = 0; v < nv; v++) {
            if (chain_next[v] < 0 || has_simple_in[v]) continue;
            int32_t cur = (int32_t)v, d = 0;
            while (chain_next[cur] >= 0) {
                cur = chain_next[cur];
                d++;
                chain_top[cur] = (int32_t)v;
                chain_rank[cur] = d;
            }
        }
        for (uint32_t v = 0; v < nv; v++)  // rings of simple edges have no head
            if (has_simple_in[v] && chain_top[v] == (int32_t)v) chain_rank[v] = -1;
        *zero_sum_edge = bad;
        return VS_OK;
    }
    uint32_t link_rows() const override { return n; }
    int block_sums(const uint64_t *list_off, const uint32_t *list_idx, uint32_t n_lists, const uint32_t *qa, const uint32_t *qb,
                   uint64_t n_queries, int64_t *out, std::string &err) override {
        for (uint64_t q = 0; q < n_queries; q++) {
            if (qa[q] >= n_lists || qb[q] >= n_lists) { err = "query names a list out of range"; return VS_E_RANGE; }
            int64_t s = 0;
            for (uint64_t i = list_off[qa[q]]; i < list_off[qa[q] + 1]; i++)
                for (uint64_t j = list_off[qb[q]]; j < list_off[qb[q] + 1]; j++) {
                    if (list_idx[i] >= n || list_idx[j] >= n) { err = "list index out of range"; return VS_E_RANGE; }
                    s += cell(list_idx[i], list_idx[j]);
                }
            out[q] = s;
        }
        return VS_OK;
    }
    int group_matrix(const uint64_t *list_off, const uint32_t *list_idx, uint32_t n_groups, int64_t *out, std::string &err) override {
        std::vector<int64_t> t((size_t)n_groups * (n ? n : 1), 0);
        for (uint32_t g = 0; g < n_groups; g++)
            for (uint64_t i = list_off[g]; i < list_off[g + 1]; i++) {
                if (list_idx[i] >= n) { err = "list index out of range"; return VS_E_RANGE; }
                if (rp.empty()) {
                    for (uint32_t c = 0; c < n; c++) t[(size_t)g * n + c] += p0[(uint64_t)list_idx[i] * n + c];
                } else {
                    for (uint64_t x = rp[list_idx[i]]; x < rp[list_idx[i] + 1]; x++) t[(size_t)g * n + ci[x]] += cv[x];
                }
            }
        for (uint32_t g = 0; g < n_groups; g++)
            for (uint32_t h = 0; h < n_groups; h++) {
                int64_t s = 0;
                for (uint64_t j = list_off[h]; j < list_off[h + 1]; j++) s += t[(size_t)g * n + list_idx[j]];
                out[(size_t)g * n_groups + h] = s;
            }
        return VS_OK;
    }
};

}  // namespace

// p0: the symmetrised PE-link table of process_pe_info as a dense [n, n] int64 matrix (host)
// the same table as CSR rows of its non-zero cells (row_ptr[n + 1], columns ascending within a row)
extern "C" int vs_stage_check_create_sparse(const uint64_t *row_ptr, const uint32_t *col, const int64_t *val, uint32_t n, vs_stage **out) {
    if (!out || !row_ptr || (row_ptr[n] && (!col || !val))) return VS_E_ARG;
    CheckOps *ops = new CheckOps();
    ops->n = n;
    ops->rp.assign(row_ptr, row_ptr + n + 1);
    ops->ci.assign(col, col + row_ptr[n]);
    ops->cv.assign(val, val + row_ptr[n]);
    *out = vs_stage_make(ops);
    return VS_OK;
}

extern "C" int vs_stage_check_create(const int64_t *p0, uint32_t n, vs_stage **out) {
    if (!out || (n && !p0)) return VS_E_ARG;
    CheckOps *ops = new CheckOps();
    ops->n = n;
    ops->p0.assign(p0, p0 + (size_t)n * n);
    *out = vs_stage_make(ops);
    return VS_OK;
}

// ---- probes of the engine's Python-semantics helpers (vs_stage_core.h), for tests/test_native_stage_cpu.py ---------------
#include <string.h>

#include "../vstrains_amd/csrc/vs_stage_core.h"

extern "C" {
// repr(float) as the engine prints depths into the stage files; returns the length
int vs_check_py_repr(double x, char *buf, int cap) {
    const std::string s = vsg::py_repr(x);
    if ((int)s.size() >= cap) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}
double vs_check_py_round2(double x) { return vsg::py_round2(x); }
double vs_check_np_mean(const double *a, uint64_t n) { return vsg::np_mean(std::vector<double>(a, a + n)); }
double vs_check_np_median(const double *a, uint64_t n) { return vsg::np_median(std::vector<double>(a, a + n)); }
// iteration order of set(values) for small non-negative ints; returns how many distinct values came out
uint32_t vs_check_int_set_order(const uint32_t *values, uint32_t n, uint32_t *out) {
    vsg::PyIntSet s;
    for (uint32_t i = 0; i < n; i++) s.add(values[i]);
    std::vector<uint32_t> o = s.order();
    for (size_t i = 0; i < o.size(); i++) out[i] = o[i];
    return (uint32_t)o.size();
}
}
