/*
 * vstrains_hip.h -- C ABI of libvstrains_hip.so, the MI355X (gfx950) implementation of the
 * VStrains hot path.  extern "C", plain pointers and sizes only; no C++ or torch types cross
 * this line.  Host code (Python via ctypes, or any C caller) binds exactly these symbols.
 *
 * Every entry returns 0 on success and a negative VS_E_* code on failure; the message is
 * available from vs_last_error(ctx) (or vs_last_error(NULL) for vs_ctx_create failures).
 * A context belongs to one device and one host thread at a time.  Nothing here falls back
 * to the CPU: without a usable HIP device the calls fail with VS_E_HIP.
 *
 * "Replaces" cites the reference lines (under /root/reference/) each entry stands in for.
 */
#ifndef VSTRAINS_HIP_H
#define VSTRAINS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_ABI_VERSION 10

enum {
    VS_OK = 0,
    VS_E_ARG = -1,       /* bad argument */
    VS_E_HIP = -2,       /* HIP runtime error (no device, launch failure, ...) */
    VS_E_OOM = -3,       /* device or host allocation failed */
    VS_E_NODE_BASE = -4, /* a node of length >= k+1 holds a byte outside ACGT: the reference
                            dies with KeyError in reverse_seq (PE_Inference.py:12-13,122) */
    VS_E_STATE = -5,     /* call order (e.g. counting before an index exists) */
    VS_E_RANGE = -6,     /* a size exceeds what this build supports */
    VS_E_UTF8 = -7,      /* a FASTQ sequence line holds bytes that are not valid UTF-8: the reference's
                            text-mode readlines() raises UnicodeDecodeError (PE_Inference.py:147-152) */
    VS_E_KEY = -8,       /* a graph stage looked up an id / index that is not there: the reference raises
                            KeyError / IndexError / ValueError at that point (vs_stage_error names which) */
    VS_E_FPE = -9,       /* an edge flow would divide by a zero neighbour sum: FloatingPointError under the
                            reference's numpy.seterr(all="raise") (vstrains:25, Utilities.py:20-30) */
    VS_E_RECURSION = -10 /* (ABI 8) a chain of forked ids is longer than the reference's recursive merge_id
                            (Utilities.py:318-327) can follow under CPython's recursion limit: the reference
                            ends with RecursionError there (runaway trivial splits on circular graphs) */
};

typedef struct vs_ctx vs_ctx;     /* one per device */
typedef struct vs_reads vs_reads; /* a device-resident block of packed read pairs */

/* ---- context ------------------------------------------------------------------------------ */
int vs_abi_version(void);
int vs_device_count(void);
int vs_ctx_create(int device, vs_ctx **out);
void vs_ctx_destroy(vs_ctx *ctx);
const char *vs_last_error(const vs_ctx *ctx);
/* Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream). */
int vs_ctx_set_stream(vs_ctx *ctx, void *stream);
int vs_ctx_sync(vs_ctx *ctx);

/* ---- K1: node index -------------------------------------------------------------------------
 * Replaces utils/VStrains_PE_Inference.py:114-135 (split_len = k+1; the dict of every
 * (k+1)-mer window of every node, forward and reverse-complemented).
 * node_ascii: the N node sequences concatenated (host memory); node_off[N+1]: byte offsets.
 * On VS_E_NODE_BASE, bad_node / bad_char (may be NULL) receive the node index and the byte the
 * reference's KeyError would name.  Rebuilding replaces the previous index. */
int vs_index_build(vs_ctx *ctx, const uint8_t *node_ascii, const uint64_t *node_off,
                   uint32_t n_nodes, uint32_t ksize, uint32_t *bad_node, uint8_t *bad_char);

/* Index facts for reporting: info[0]=seed length w, [1]=probe stride s, [2]=seed positions
 * indexed, [3]=hash slots, [4]=distinct seeds, [5]=bytes of device memory held by the index. */
int vs_index_info(const vs_ctx *ctx, uint64_t info[6]);

/* A numbering of the nodes that runs along the graph's paths (host only; depth-first over k-base overlaps, either strand).
 * order_out[n_nodes]: order_out[r] = the node (position in node_off) that should be handed to vs_index_build as number r.
 * Optional, for speed only: the matrices vs_pe_count fills are indexed by the numbering vs_index_build was given
 * (PE_Inference.py:139-140 indexes them by GFA position), and every result is the same sum under any numbering -- but
 * pairs are processed in the order of the node their forward read starts in, a slice per XCD, and a numbering that
 * scatters the neighbours of a path costs 20-25 % of the step (csrc/vs_order_host.cpp).  The Python host side does this
 * by default and maps the matrices back (vstrains_amd/pe.py). */
int vs_node_order_host(const uint8_t *node_ascii, const uint64_t *node_off, uint32_t n_nodes, uint32_t ksize,
                       uint32_t *order_out);

/* ---- read blocks ----------------------------------------------------------------------------
 * Replaces the FASTQ record slicing of PE_Inference.py:146-159 from the point where the two
 * sequence strings of a pair are known.  Ends are interleaved: end 2r = forward read of pair
 * r, end 2r+1 = its reverse read.  ascii/off are host memory: off[n_ends+1] byte offsets into
 * ascii.  Bytes are taken verbatim: 'N' marks the pair as an N-pair (:160); any other byte
 * outside ACGT makes every window covering it miss, as a dict lookup would (:25-26). */
int vs_reads_pack(vs_ctx *ctx, const uint8_t *ascii, const uint64_t *off, uint64_t n_ends,
                  vs_reads **out);
void vs_reads_free(vs_ctx *ctx, vs_reads *reads);
/* info[0]=ends, [1]=packed words, [2]=max read length, [3]=ends holding bytes outside ACGTN,
 * [4]=device bytes held. */
int vs_reads_info(const vs_reads *reads, uint64_t info[5]);
/* Unpack back to ASCII (ACGT only; testing aid).  out must hold off[n_ends] bytes where off is
 * the cumulative read length; lens[n_ends] and flags[n_ends] (bit0: has N, bit1: has other
 * non-ACGT) may be NULL. */
int vs_reads_unpack(vs_ctx *ctx, const vs_reads *reads, uint8_t *out, uint32_t *lens,
                    uint8_t *flags);

/* ---- FASTQ ingest (host, multi-threaded) -----------------------------------------------------
 * Replaces PE_Inference.py:146-159: both files read in text mode (universal newlines), record r
 * = lines 4r..4r+3, sequence = line 4r+1 minus its last character (the newline, or a real
 * character on a final line without one), n_pairs = min(lines_f // 4, lines_r // 4).  Text mode means
 * the reference sees CHARACTERS: a valid UTF-8 multi-byte sequence inside a sequence line is one character
 * (counted once, every window over it misses); vs_fastq_sequence / _gather / _block deliver one byte per
 * character ('?' for a multi-byte one); invalid UTF-8 there is VS_E_UTF8.
 * vs_fastq_open maps and indexes both files on the host cores (VS_HOST_THREADS overrides the
 * count); a file that starts with the gzip magic is inflated into memory first (zlib; several
 * members in a row are fine, a cut-off stream is VS_E_ARG).  vs_fastq_block turns pairs
 * [first, first+count) into a device read block. */
/* The host packer the ingest uses on every sequence line: `len` bytes -> ceil(len/16) words, 16
 * bases per word, LSB first, A C G T = 0 1 2 3, any other byte packs as 0; *flags: bit 0 an 'N',
 * bit 1 another ASCII byte outside ACGT, bit 7 a byte >= 0x80.  plain != 0 takes the byte-by-byte
 * body instead of the vector one (same result; tests compare the two). */
int vs_pack_sequence(const uint8_t *seq, uint32_t len, uint32_t *words, uint32_t *flags, int plain);
typedef struct vs_fastq vs_fastq;
int vs_fastq_open(vs_ctx *ctx, const char *fwd_path, const char *rve_path, vs_fastq **out);
void vs_fastq_close(vs_fastq *fq);
/* Cooperative open for one process per GPU, so that no rank reads a whole file (PE_Inference.py:146-154 reads both
 * files completely; its total = min(lines_f // 4, lines_r // 4) follows from the ranks' counts):
 *   1. every rank r calls vs_fastq_count_part(path, r, world, out) for both files: out[0] = newlines in its byte
 *      range, out[1] = file size, out[2] = flags (bit 0: the range holds '\r', bit 1: gzip file, bit 2: the file does
 *      not end in a newline); the ranks exchange these (an all-gather of three integers per file);
 *   2. from the totals every rank derives its record range and calls vs_fastq_open_records with everybody's counts:
 *      only the bytes of records [first, last) are indexed; the handle numbers them from 0.
 * Files with '\r' or gzip files are opened whole (vs_fastq_open) by every rank instead. */
int vs_fastq_count_part(const char *path, uint32_t part, uint32_t n_parts, uint64_t out[3]);
int vs_fastq_open_records(vs_ctx *ctx, const char *fwd_path, const char *rve_path, uint32_t n_parts,
                          const uint64_t *counts_f, const uint64_t *counts_r, uint64_t first, uint64_t last,
                          vs_fastq **out);
/* text bytes the handle went through when it was opened (both files) */
uint64_t vs_fastq_bytes_indexed(const vs_fastq *fq);
/* info[0] = pairs, [1] = lines of the forward file, [2] = lines of the reverse file */
int vs_fastq_info(const vs_fastq *fq, uint64_t info[3]);
int vs_fastq_sequence(const vs_fastq *fq, int which, uint64_t record, uint8_t *buf, uint32_t cap,
                      uint32_t *len);
/* off[2*count+1] byte offsets of the interleaved ends (2r forward, 2r+1 reverse); ascii (may be
 * NULL to get the sizes only) receives the bytes.  Host pointers. */
int vs_fastq_gather(const vs_fastq *fq, uint64_t first, uint64_t count, uint64_t *off,
                    uint8_t *ascii);
int vs_fastq_block(vs_ctx *ctx, vs_fastq *fq, uint64_t first, uint64_t count, vs_reads **out);

/* pe_info / st_info text (PE_Inference.py:194-205): "{id_i}:{id_j}:{count}\n" for all i, j in
 * row-major order, zeros included.  ids: the n node names concatenated, id_off[n+1]; mat: HOST
 * n*n int64.  Formatted on all host cores, one write(). */
int vs_write_matrix_text(vs_ctx *ctx, const char *path, const uint8_t *ids, const uint64_t *id_off,
                         uint32_t n, const int64_t *mat);

/* Synthetic pairs generated on the device from a seed (bench workload; the CPU twin is
 * oracle/pe_oracle.c:peo_synth_pairs).  genomes: concatenated ACGT ASCII (host), goff
 * [n_strains+1]; cum[s]: inclusive upper bound of strain s in a uniform u32 draw (last =
 * 0xFFFFFFFF).  sub_thresh / n_thresh: u32 thresholds (rate * 2^32) for per-base
 * substitutions and for pairs that get one 'N'.  Pairs first_pair .. first_pair+n_pairs-1 of
 * the stream are produced, so ranks can take disjoint slices of one stream. */
int vs_synth_pairs(vs_ctx *ctx, const uint8_t *genomes, const uint64_t *goff,
                   const uint32_t *cum, uint32_t n_strains, uint64_t seed, uint64_t first_pair,
                   uint64_t n_pairs, uint32_t read_len, uint32_t sub_thresh, uint32_t n_thresh,
                   vs_reads **out);

/* ---- K2-K4: PE-link counting ----------------------------------------------------------------
 * Replaces PE_Inference.py:155-188 (pair filters, single_end_read_mapping on both ends,
 * short_mat / node_mat updates) for all pairs of `reads`.
 * d_node_mat, d_short_mat: DEVICE pointers to N*N row-major uint32 counters, d_stats: DEVICE
 * pointer to 3 uint64 {n_reads, short_reads, used_reads}.  Counts are ADDED (atomically), so
 * several blocks / ranks can accumulate and the caller all-reduces (RCCL) as it likes.
 * A cell grows by at most 2 per pair (short_mat[i][i] is incremented once per end, :174-184), so
 * uint32 is exact while 2 * (pairs counted into one buffer, over all ranks that will be summed
 * into it) < 2^32; vs_counts_fold moves a buffer into int64 totals before that. */
int vs_pe_count(vs_ctx *ctx, const vs_reads *reads, uint32_t *d_node_mat, uint32_t *d_short_mat,
                uint64_t *d_stats);

/* The same with a record of WHERE the counts went: d_tile_map holds one byte per 64 x 64 tile of node_mat followed by one
 * per tile of short_mat (2 * T * T bytes, T = ceil(N / 64), DEVICE memory, zero before the first call); the tiles the
 * block adds to are set to 1.  A caller that wants empty counters for the next block then calls vs_counts_zero_tracked
 * -- zero every marked tile, clear the map -- instead of clearing 2 * N * N cells: the counters of a 50 k-node graph are
 * 23.7 GB and a block touches a few per cent of them.  (The reference allocates its matrices once, PE_Inference.py:
 * 139-140; per-block zeroing exists only where a caller counts blocks separately, as bench.py's steps do.)  Ranks that
 * sum their counters must OR their maps too before they zero (a maximum over the bytes). */
int vs_pe_count_tracked(vs_ctx *ctx, const vs_reads *reads, uint32_t *d_node_mat, uint32_t *d_short_mat,
                        uint64_t *d_stats, uint8_t *d_tile_map);
int vs_counts_zero_tracked(vs_ctx *ctx, uint32_t *d_node_mat, uint32_t *d_short_mat, uint32_t n, uint8_t *d_tile_map);

/* d_wide[i] += d_counts[i] (read as uint32); d_counts[i] = 0, for i < n.  DEVICE pointers.  The
 * reference's matrices are numpy.zeros(..., dtype=int) = int64 (PE_Inference.py:139-140): a caller
 * that counts more pairs than a uint32 cell can hold folds into int64 totals in between. */
int vs_counts_fold(vs_ctx *ctx, uint32_t *d_counts, int64_t *d_wide, uint64_t n);

/* (ABI 8) d_occ[j] = 1 where the j-th stretch of 64 consecutive cells (cell_bytes = 4: uint32 counters, 8: int64 totals)
 * holds a non-zero cell, else 0, for j < n_stretches.  DEVICE pointers.  No counterpart in the reference (single
 * process): the ranks of a multi-GPU run exchange only the occupied stretches of their counters (the non-zero cells of
 * node_mat / short_mat, PE_Inference.py:174-188, lie in a band), and this is the one pass over the buffer that finds them. */
int vs_counts_occupied(vs_ctx *ctx, const void *d_cells, uint32_t cell_bytes, uint64_t n_stretches, uint8_t *d_occ);

/* ---- C1: sum of the counters over ranks (RCCL over xGMI) ---------------------------------------
 * No counterpart in the reference (single process); read pairs are independent and the counters
 * add (PE_Inference.py:174-188), so ranks count disjoint read blocks and sum.  One process per
 * GPU.  The RCCL library is resolved at the first call (the copy already loaded into the process,
 * else librccl.so.1); without it the calls fail with VS_E_HIP.
 *   vs_comm_unique_id : rank 0 makes the 128-byte id and hands it to the other ranks out of band
 *   vs_comm_init_rank : collective over the n_ranks processes; *comm receives an ncclComm_t
 *   vs_pe_allreduce   : in-place ncclAllReduce(ncclSum) of node_mat[n*n], short_mat[n*n] (uint32,
 *                       or int64 totals when `wide` != 0) and of the 3 uint64 stats, enqueued on the
 *                       ctx stream.  `comm` is an ncclComm_t made by vs_comm_init_rank or by the
 *                       caller's own RCCL calls in this process.  DEVICE pointers. */
int vs_comm_unique_id(vs_ctx *ctx, uint8_t id[128]);
int vs_comm_init_rank(vs_ctx *ctx, int n_ranks, const uint8_t id[128], int rank, void **comm);
int vs_comm_destroy(vs_ctx *ctx, void *comm);
int vs_pe_allreduce(vs_ctx *ctx, void *comm, void *d_node_mat, void *d_short_mat, uint64_t *d_stats,
                    uint32_t n, int wide);

/* Per-end result of single_end_read_mapping (PE_Inference.py:16-48) for testing: for each end
 * e, counts[e] = number of accepted nodes (0 for ends of pairs the filters drop), and up to
 * `cap` of them, ascending, in lists[e*cap ...].  Host pointers. */
int vs_pe_map_ends(vs_ctx *ctx, const vs_reads *reads, uint32_t cap, uint32_t *lists,
                   uint32_t *counts);

/* Timing of the most recent vs_pe_count on this ctx, measured with HIP events on the ctx
 * stream: ms[0] = mapping kernel (k_pe_tiles), ms[1] = overflow (slow-path) kernel, ms[2] =
 * pairs sent to the slow path, ms[3] = locus ordering of the pairs in front of the mapping
 * kernel (k_pe_locus + scan + k_pe_permute), ms[4] = counter kernels (k_pe_accumulate, or the row
 * owners k_list_owners .. k_rows_sum on graphs beyond 46 340 nodes).
 * Synchronises the stream. */
int vs_pe_last_timing(vs_ctx *ctx, double ms[5]);
/* Name of the mapping-kernel instantiation the most recent vs_pe_count launched, as a profiler
 * prints it (e.g. "k_pe_tiles<true, 10u, 5u>"); "" before the first call. */
const char *vs_pe_last_kernel(const vs_ctx *ctx);
/* Which of the optional kernels the most recent vs_pe_count launched (a run's choice by graph size and read shape): */
enum {
    VS_RAN_LOCUS_LDS_SORT = 1,    /* locus order by per-workgroup LDS histograms (k_locus_count / k_locus_scatter) */
    VS_RAN_LOCUS_GLOBAL_SORT = 2, /* ... by global atomics (k_pe_locus / k_pe_permute): graphs beyond 147 k nodes */
    VS_RAN_PE_MID = 8,            /* overflow pairs through the wavefront-per-pair kernel first (k_pe_mid) */
    VS_RAN_ROW_OWNERS = 16        /* counters summed by row owners (k_list_owners / k_rows_count / k_rows_fill / k_rows_sum): graphs beyond 46 340 nodes */
};
uint32_t vs_pe_last_launched(const vs_ctx *ctx);

/* ---- graph stages: K5 PE-link table ---------------------------------------------------------
 * Replaces process_pe_info (utils/VStrains_IO.py:598-627) and every later read or rewrite of the
 * pe_info dict (utils/VStrains_Decomposition.py:141-143,178,273,492-503,608-617,672-684;
 * utils/VStrains_Utilities.py:488-499; utils/VStrains_Extension.py:62,766-799).
 * The table is P0[i][j] = node[i][j] + node[j][i] + short[i][j] + short[j][i] for i != j and
 * P0[i][i] = node[i][i] + short[i][i], int64, resident on the device; it is never rewritten:
 * a lookup for nodes made by splits / contractions is a sum over two lists of original rows
 * (see vstrains_amd/graph/ops.py for the equivalence, tests/test_graph_golden.py for the check
 * against the literal dict). */
typedef struct vs_links vs_links;
/* d_node_mat / d_short_mat: DEVICE pointers to the N*N uint32 counters vs_pe_count filled. */
int vs_links_from_counts(vs_ctx *ctx, const uint32_t *d_node_mat, const uint32_t *d_short_mat,
                         uint32_t n, vs_links **out);
/* ABI 10: the same from counters that keep a dirty-tile map (vs_pe_count_tracked: one byte per 64 x 64 tile of node_mat, then of
 * short_mat; a cell outside the marked tiles is zero).  From `sparse_min_nodes` nodes on (0 = the default, 32 768) the table
 * is held as CSR rows of its non-zero cells, built from the marked tiles only -- 0.4 GB instead of 23.7 GB at 54 465 nodes, and
 * no pass over the counters; below, or without a map, exactly vs_links_from_counts.  Every vs_links_* call takes either form.
 * (process_pe_info, IO.py:598-627: a dict of N (N + 1) / 2 keys in the reference.) */
int vs_links_from_counts_tracked(vs_ctx *ctx, const uint32_t *d_node_mat, const uint32_t *d_short_mat, uint32_t n,
                                 const uint8_t *d_tile_map, uint32_t sparse_min_nodes, vs_links **out);
/* Same from DEVICE int64 totals (vs_counts_fold). */
int vs_links_from_wide(vs_ctx *ctx, const int64_t *d_node_mat, const int64_t *d_short_mat,
                       uint32_t n, vs_links **out);
/* Same from HOST int64 matrices (e.g. parsed back from pe_info / st_info text). */
int vs_links_from_host(vs_ctx *ctx, const int64_t *node_mat, const int64_t *short_mat, uint32_t n,
                       vs_links **out);
/* Optional, ABI 9: set aside the device buffer of the next table of n nodes (n*n int64) now -- typically when the counters
 * are allocated, before any read is counted -- so that vs_links_from_counts / _from_wide / _from_host does not have to ask the
 * driver for it later (a hipMalloc of tens of gigabytes takes 0.3 ms or half a second depending on what the process freed
 * before: 23.7 GB at 54 465 nodes).  The buffer belongs to the context until a table of that size takes it; a second call
 * replaces it, n = 0 gives it back.  The reference has no counterpart: its table is a Python dict. */
int vs_links_reserve(vs_ctx *ctx, uint32_t n);
void vs_links_free(vs_ctx *ctx, vs_links *links);
int vs_links_size(const vs_links *links, uint32_t *n);
int vs_links_to_host(vs_ctx *ctx, const vs_links *links, int64_t *out /* n*n */);
/* A pool of index lists: list l is list_idx[list_off[l] .. list_off[l+1]) (rows of P0, repeats
 * allowed, may be empty).  out[q] = sum over r in list qa[q], c in list qb[q] of P0[r][c].
 * Host pointers. */
int vs_links_block_sums(vs_ctx *ctx, const vs_links *links, const uint64_t *list_off,
                        const uint32_t *list_idx, uint32_t n_lists, const uint32_t *qa,
                        const uint32_t *qb, uint64_t n_queries, int64_t *out);
/* out[g * n_groups + h] = block sum of group g x group h, for all pairs (final_link_info,
 * Extension.py:766-799).  Host pointers. */
int vs_links_group_matrix(vs_ctx *ctx, const vs_links *links, const uint64_t *list_off,
                          const uint32_t *list_idx, uint32_t n_groups, int64_t *out);

/* ---- graph stages: K6 vertex scan + chain ranking, K7 edge flow -------------------------------
 * One call per re-initialised stage graph (store_reinit_graph, VStrains_IO.py:630-642).
 * The graph is a CSR in adjacency order: row v = nbr/eidx[row_ptr[v] .. row_ptr[v+1]), its
 * first n_out[v] entries are out-edges (target, edge index), the rest in-edges (source, edge
 * index).  edge_black is indexed by edge index (n_edge_slots of them).  Host pointers; any
 * output pointer may be NULL.
 *   flow[e]        assign_edge_flow, VStrains_Utilities.py:14-31 (numpy.sum / numpy.mean order)
 *   nontrivial[v]  is_non_trivial, VStrains_Utilities.py:162-172
 *   fork_kind[v]   1: one black in / several black out, 2: several in / one out
 *                  (VStrains_Decomposition.py:715,763)
 *   chain_next[v]  target of v's simple out-edge or -1 (simp_path, VStrains_Utilities.py:398-402)
 *   chain_top[v], chain_rank[v]  head of v's chain of simple edges and v's distance from it
 *                  (pointer jumping; rank -1 on a ring of simple edges)
 *   *zero_sum_edge smallest edge index whose flow would divide by a zero neighbour sum, or
 *                  0xFFFFFFFF (the reference raises FloatingPointError there, vstrains:25) */
int vs_graph_refresh(vs_ctx *ctx, uint32_t n_vertices, uint32_t n_edge_slots,
                     const uint64_t *row_ptr, const uint32_t *n_out, const uint32_t *nbr,
                     const uint32_t *eidx, const double *dp, const uint8_t *vertex_black,
                     const uint8_t *edge_black, double *flow, uint8_t *nontrivial,
                     uint8_t *fork_kind, int32_t *chain_next, int32_t *chain_top,
                     int32_t *chain_rank, uint32_t *zero_sum_edge);

/* ---- graph stages: native stage handle ----------------------------------------------------------
 * The state the reference keeps in a graph_tool.Graph plus Python dicts and re-derives from a GFA file after every
 * pass -- graph, simp_node_dict, simp_edge_dict, contig_dict, the rewritten pe_info, full_link, usages -- as ONE
 * object in this library, with every stage of VStrains_SPAdes.py:140-248 as one call on it.  Flows, vertex scan and
 * chain ranking of each re-initialised graph and all PE-link sums run on the device (K5-K7 above); the decisions run
 * on the host inside the call; stage GFA files are written by worker threads of the handle and are complete when the
 * call that names them returns.  One entry per reference function:
 *   vs_stage_edge_cleaning          edge_cleaning, Decomposition.py:822-905
 *   vs_stage_reinit                 store_reinit_graph, IO.py:630-642
 *   vs_stage_disentangle            iter_graph_disentanglement, Decomposition.py:908-1042 (balance_split :91-530,
 *                                   trivial_split :533-688, simp_path_compactification Utilities.py:383-574,
 *                                   contig_dict_remapping :281-380, contig_dup_removed_s :589-616, trim_contig_dict :147-159)
 *   vs_stage_best_matching          best_matching, Extension.py:10-111
 *   vs_stage_increment_nt_coverage  increment_nt_branch_coverage, Utilities.py:183-208
 *   vs_stage_write_gfa              graph_to_gfa, IO.py:337-372
 *   vs_stage_path_extension         path_extension, Extension.py:484-899 (global_trivial_split Decomposition.py:691-819,
 *                                   contig_extension / final_extension Extension.py:115-418, reduce_graph :429-455)
 *   vs_stage_write_contigs          contig_dict_to_path IO.py:558-595 / contig_dict_to_fasta :518-536 of contig_dict
 * State crosses the boundary as byte blobs (vs_stage_import / vs_stage_export): a sequence of sections, each a
 * uint32 tag (VS_STAGE_*) and a payload of little-endian arrays and newline-joined string lists; the layout is stated
 * where it is written, vstrains_amd/graph/native_stage.py.  `links` must outlive the handle.  Errors: the return
 * code, and vs_stage_error for the message and the name of the exception the reference raises in that situation. */
typedef struct vs_stage vs_stage;
enum {
    VS_STAGE_GRAPH = 1,    /* vertices, adjacency rows, edge slots, free list, the two ordered maps */
    VS_STAGE_CONTIGS = 2,  /* contig_dict */
    VS_STAGE_LINKS = 4,    /* full_link (best_matching's result, consumed by path_extension) */
    VS_STAGE_STRAINS = 8,  /* strain_dict (export only) */
    VS_STAGE_USAGES = 16,  /* usages (export only) */
    VS_STAGE_LOG = 32,     /* the log lines of the calls since the last export of this section (export only) */
    VS_STAGE_SCAN = 64,    /* non-trivial branches, fork kinds, chain ranks of the last re-initialisation */
    VS_STAGE_ASSIGNED = 128 /* edge_cleaning's result: (source id, target id) -> accounted for (export only) */
};
int vs_stage_create(vs_ctx *ctx, const vs_links *links, vs_stage **out);
void vs_stage_destroy(vs_stage *st);
const char *vs_stage_error(const vs_stage *st, const char **kind);
int vs_stage_set_debug(vs_stage *st, int on); /* also collect the DEBUG lines */
/* names of the rows of `links`, '\n'-joined (the nodes of s_graph_L1 in the numbering the table was built in) */
int vs_stage_set_link_names(vs_stage *st, uint32_t n, const uint8_t *joined, uint64_t len);
int vs_stage_import(vs_stage *st, const uint8_t *blob, uint64_t len);
/* the buffer belongs to the handle and is valid until the next call on it */
int vs_stage_export(vs_stage *st, uint32_t what, const uint8_t **blob, uint64_t *len);
/* pe_info[(a, b)] as the dict the reference rewrites through every split / fork / contraction would hold it now */
int vs_stage_link(vs_stage *st, const char *a, const char *b, int64_t *out);
int vs_stage_edge_cleaning(vs_stage *st);
int vs_stage_reinit(vs_stage *st, const char *gfa_path);
/* scan of an imported graph without gray objects (the reference asks get_non_trivial_branches, Utilities.py:175-180,
 * of whatever graph it is handed); flows are left as they are */
int vs_stage_refresh_scan(vs_stage *st);
int vs_stage_disentangle(vs_stage *st, double threshold, const char *temp_dir);
int vs_stage_best_matching(vs_stage *st);
int vs_stage_increment_nt_coverage(vs_stage *st);
int vs_stage_write_gfa(vs_stage *st, const char *path);
int vs_stage_write_contigs(vs_stage *st, const char *paths_file, const char *fasta_file); /* either may be NULL */
int vs_stage_path_extension(vs_stage *st, double threshold, const char *temp_dir);
/* VStrains_SPAdes.py:251-262 on the strain records path_extension left: contig_resolve (Utilities.py:211-224),
 * trim_contig_dict (:147-159) measured on the graph kept by vs_stage_keep_graph -- call it right after the
 * es_graph_L2 re-initialisation --, contig_dup_removed_s (:589-616), tmp/tmp_strain.paths (IO.py:558-595) */
int vs_stage_keep_graph(vs_stage *st);
int vs_stage_finish_strains(vs_stage *st, const char *tmp_paths_file);
/* numpy.median of the vertex depths (the thresholds of VStrains_SPAdes.py:187,237 are 0.05 x this) */
int vs_stage_median_depth(vs_stage *st, double *out);
/* info[0] re-initialisations, [1] of which reused an untouched state, [2] flow/scan launches, [3] link-sum launches,
 * [4] files written, [5] bytes written, [6] vertices, [7] live edges; secs[0] in re-initialisations, [1] of which in
 * the flow/scan operation, [2] in link sums, [3] busy time of the file writers */
int vs_stage_counters(vs_stage *st, uint64_t info[8], double secs[4]);
/* "name=seconds;" per section of the stage calls so far (where a leg's time goes), into buf */
int vs_stage_sections(vs_stage *st, char *buf, uint64_t cap);

/* ---- device memory helpers for C callers without another allocator ----------------------- */
int vs_dev_alloc(vs_ctx *ctx, size_t bytes, void **out); /* zero-filled */
int vs_dev_free(vs_ctx *ctx, void *ptr);
int vs_dev_zero(vs_ctx *ctx, void *ptr, size_t bytes);
int vs_dev_to_host(vs_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* VSTRAINS_HIP_H */
