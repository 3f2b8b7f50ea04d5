// Native stage graph of the disentanglement / path-extraction half of the hot path (SURVEY.md 8a-8 .. a-22).
//
// The reference keeps this state in a graph_tool.Graph plus Python dicts and re-derives it from a GFA file after every
// pass (utils/VStrains_IO.py:630-642).  Here it lives in one C++ object behind the C ABI (`vs_stage`, include/
// vstrains_hip.h): vertex / edge arrays, colours, depths, overlaps, the edge-index free list, adjacency rows in the
// container's order, the two insertion-ordered maps, the contig records and the PE-link bookkeeping -- and every
// decision of the stages runs on it (vs_stage.cpp).  The data-parallel parts go to the device through VsStageOps:
// edge flows + vertex scan + chain ranking of a re-initialised graph (K6 / K7) and sums over the resident PE-link
// matrix (K5).  The product library implements VsStageOps with the HIP kernels of vs_graph.hip; tests link the same
// engine against a CPU checker of the three operations (oracle/stage_check.cpp) -- there is no CPU implementation in
// the product library.
#pragma once
#include <stdint.h>

#include <string>

struct VsStageOps {
    virtual ~VsStageOps() {}
    // One re-initialised graph (all vertices and edges live).  CSR in adjacency order: row v = [row_ptr[v], row_ptr[v+1]),
    // the first n_out[v] entries out-edges (target, edge), the rest in-edges (source, edge).  Fills flow[n_edges] and the
    // per-vertex scan (see vs_graph_refresh in include/vstrains_hip.h).  Returns 0 or a VS_E_* code with `err` set.
    virtual int refresh(uint32_t n_vertices, uint32_t n_edges, const uint64_t *row_ptr, const uint32_t *n_out, const uint32_t *nbr,
                        const uint32_t *eidx, const double *dp, double *flow, uint8_t *nontrivial, uint8_t *fork_kind,
                        int32_t *chain_next, int32_t *chain_top, int32_t *chain_rank, uint32_t *zero_sum_edge, std::string &err) = 0;
    // rows of the PE-link matrix P0 (process_pe_info, IO.py:598-627)
    virtual uint32_t link_rows() const = 0;
    // out[q] = sum_{r in list qa[q]} sum_{c in list qb[q]} P0[r][c]; list l = list_idx[list_off[l] .. list_off[l+1])
    virtual int block_sums(const uint64_t *list_off, const uint32_t *list_idx, uint32_t n_lists, const uint32_t *qa, const uint32_t *qb,
                           uint64_t n_queries, int64_t *out, std::string &err) = 0;
    // out[g * n_groups + h] = block sum of group g x group h
    virtual int group_matrix(const uint64_t *list_off, const uint32_t *list_idx, uint32_t n_groups, int64_t *out, std::string &err) = 0;
};

struct vs_stage;
// The engine takes ownership of `ops`.
vs_stage *vs_stage_make(VsStageOps *ops);
