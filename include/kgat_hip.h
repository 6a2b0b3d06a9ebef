/* kgat_hip.h - C ABI of libkgat_hip.so: KGAT's attentive embedding-propagation path on
 * MI355X (gfx950).  This is the drop-in boundary: every entry point below
 * replaces one DGL 0.4.x kernel family that the reference reaches from models.py, and is
 * exactly what a binding for that call site would bind (ctypes stub: INTEGRATION.md).
 *
 * Conventions
 *  - Plain pointers and sizes only.  Every pointer is a DEVICE pointer (hipMalloc'd or
 *    torch-allocated) unless the parameter name ends in `_host`.  Buffers are borrowed for
 *    the duration of the call; the library allocates nothing persistent and frees nothing
 *    it did not allocate.  Scratch memory is passed in by the caller (`*_workspace_bytes`).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is
 *    enqueued asynchronously on it; no entry point synchronises the host.
 *  - Return value: 0 on success, negative KGAT_E_* on failure; the message is available
 *    from kgat_last_error() (thread-local).  Nothing throws or aborts across the ABI.
 *  - Indices are int32 (E, N < 2^31); byte offsets are formed in 64 bit.  Feature matrices
 *    are row-major contiguous fp32.  Per-edge arrays are in EDGE-ID order (the order edges
 *    were added, reference dataset.py:116) unless the name says `_csr` (destination-major
 *    CSR position order) - outputs are always fully overwritten.
 *  - Edge direction (reference dataset.py:116, add_edges(t, h)): src = tail t, dst = head h.
 */
#ifndef KGAT_HIP_H_
#define KGAT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGAT_ABI_VERSION 10

enum {
  KGAT_OK = 0,
  KGAT_E_BADARG = -1,      /* null pointer, negative size, inconsistent sizes */
  KGAT_E_UNSUPPORTED = -2, /* e.g. feature width the kernels do not cover */
  KGAT_E_WORKSPACE = -3,   /* workspace too small */
  KGAT_E_HIP = -4          /* a HIP runtime call or launch failed */
};

/* flags for kgat_spmm_umule_sum_f32 */
enum {
  KGAT_SPMM_MUL_SELF = 1, /* out[v,:] *= X[row0+v,:]  (the h * h_neighbor of models.py:66) */
  KGAT_SPMM_DEFER_FINISH = 2 /* the second launch is left to kgat_bi_interaction_mul_deferred_f32 (see there) */
};

/* algorithm selectors (AUTO picks the tuned kernel; the others exist for A/B and tests) */
enum {
  KGAT_SPMM_ALGO_AUTO = 0,
  KGAT_SPMM_ALGO_MERGE = 1,  /* edge-balanced tiles over the destination-sorted edge array */
  KGAT_SPMM_ALGO_ROWS = 2,   /* one lane group per destination row (optionally degree ordered) */
  KGAT_SPMM_ALGO_GENERIC = 3, /* any feature width, one wavefront per row */
  KGAT_SPMM_ALGO_MERGE1 = 4   /* first form of the merge kernel (shuffle-fed), kept for A/B */
};
enum {
  KGAT_ATT_ALGO_AUTO = 0,
  KGAT_ATT_ALGO_MFMA = 1,   /* v_mfma_f32_16x16x4_f32; d == k in {16,32,64,128} */
  KGAT_ATT_ALGO_GENERIC = 2, /* VALU, any (d,k) with d*k*4 <= 64 KiB */
  KGAT_ATT_ALGO_MFMA_CHUNK = 3 /* the workgroup-chunk MFMA kernel (W_r in LDS) that AUTO takes at d = 128, beyond
                                * 4,096 relations and for tables of 4 GiB and more - selectable so that tests reach it
                                * on small inputs */
};

/* flags for kgat_att_score_fused_f32 and kgat_att_score_folded_f32 */
enum {
  KGAT_ATT_F32_PRODUCTS = 1, /* both products as v_mfma_f32_16x16x4_f32 (the round-1 form) instead of
                             * the default where a kernel has it (fused: d % 32 == 0; folded: d = 128):
                             * every fp32 operand cut by round-to-nearest into three bf16 pieces that
                             * sum to it exactly (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|), the six piece
                             * products of weight >= 2^-16 accumulated in fp32 by
                             * v_mfma_f32_16x16x32_bf16; the three dropped products are together
                             * < 2^-23 of |a*b| (one fp32 ulp), of either sign; error against fp64
                             * measured no larger than the fp32 form's */
  KGAT_ATT_TILES32 = 2      /* kgat_att_score_fused_f32 at d = k = 64: `tiles` / `rec_g` were built with 32 groups per
                             * tile (kgat_fold_tiles / kgat_att_pack_records, groups_per_tile = 32): the kernel on
                             * v_mfma_f32_32x32x16_f16 with both products on fp16 pieces (W_r 2^shift three, the head
                             * rows - scaled per row to [2^13, 2^14) - and the tanh values 2^14 two: five piece products
                             * each).  Opt-in (KGAT_ATT_TILES32=1): 25 % fewer vector and 55 % fewer matrix instructions per launch at the
                             * same run time as the 16-group kernel (profiles/r05_att32_experiments.txt). */
};

typedef void* kgat_stream_t; /* hipStream_t */

int kgat_version(void);
const char* kgat_last_error(void);
/* sha256[:16] of the sources (csrc/ + this header + compiler flags) the library was built from,
 * as the build recipe passed it in; the loader refuses a library whose hash is not the hash of
 * the sources beside it (a stale build would otherwise be called with other argument lists). */
const char* kgat_build_hash(void);

/* ---------------------------------------------------------------- graph structure (G0)
 * Replaces DGL's COO -> in-CSR conversion that runs on the first kernel call on a graph
 * built by reference dataset.py:112-120.  Stable: within a destination row, edge ids
 * ascend.  indptr[N+1], col[E] (source ids), eid[E] (original edge id per CSR position),
 * row_of[E] (destination id per CSR position; may be NULL).
 * Ids outside [0,N) are a caller error (checked by the host wrapper, not on device). */
size_t kgat_csr_from_coo_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int kgat_csr_from_coo(int64_t n_nodes, int64_t n_edges, const int32_t* src, const int32_t* dst,
                      int32_t* indptr, int32_t* col, int32_t* eid, int32_t* row_of,
                      void* workspace, size_t workspace_bytes, kgat_stream_t stream);

/* Replaces the R full-graph g.filter_edges(lambda e: e.data['type'] == i) sweeps of
 * reference models.py:149-150 by one stable grouping: perm[rel_ptr[r] .. rel_ptr[r+1]) are
 * the edge ids of relation r, ascending.  Types outside [0,R) are placed after rel_ptr[R]
 * (the reference loop never visits them). */
size_t kgat_group_by_relation_workspace_bytes(int64_t n_edges, int n_rel);
int kgat_group_by_relation(int64_t n_edges, int n_rel, const int32_t* etype, int32_t* rel_ptr,
                           int32_t* perm, void* workspace, size_t workspace_bytes,
                           kgat_stream_t stream);

/* Inverse of a permutation: inv[perm[i]] = i  (CSR position of each edge id). */
int kgat_invert_permutation(int64_t n, const int32_t* perm, int32_t* inv, kgat_stream_t stream);

/* Row schedule for the row-parallel kernels: a permutation of the CSR rows by descending
 * in-degree (stable), so that the row groups that share a wavefront carry similar work and
 * the heaviest rows start first.  order[N]. */
size_t kgat_row_order_workspace_bytes(int64_t n_rows);
int kgat_row_order_by_degree(int64_t n_rows, const int32_t* indptr, int32_t* order,
                             void* workspace, size_t workspace_bytes, kgat_stream_t stream);

/* ---------------------------------------------------------------- attention score (A1+A2)
 * Replaces the per-relation g.apply_edges(self._att_score, e_idxs) loop of reference
 * models.py:135-152 by one launch over relation-grouped edges:
 *   logits[e] = sum_j (ent[src e] W_R[r])_j * tanh((ent[dst e] W_R[r])_j + rel[r]_j), r = type(e)
 * ent (N,d), W_R (R,d,k), rel (R,k).  rel_ptr/perm come from kgat_group_by_relation;
 * src_g[i] = src[perm[i]], dst_g[i] = dst[perm[i]] (relation-grouped endpoint arrays, built
 * once per graph with kgat_gather_i32).  Edges whose type is outside [0,R) get logit 0 (DGL
 * zero-initialises the column on the first partial write).  logits[E] in edge-id order; if
 * logits_csr is non-NULL the same value is also written in CSR position order through
 * pos_g[i] = CSR position of edge perm[i] (= kgat_gather_i32(perm, csr_pos)). */
int kgat_att_score_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                       const int32_t* rel_ptr, const int32_t* perm, const int32_t* src_g,
                       const int32_t* dst_g, const float* ent, const float* W_R, const float* rel,
                       float* logits, float* logits_csr, const int32_t* pos_g, int algo,
                       kgat_stream_t stream);

/* Head groups for the split attention path.  Input: a relation-grouped edge list whose
 * relations are internally sorted by destination (group the CSR-ordered edge list by relation:
 * rel_ptr[R+1], dst_g[E]).  Consecutive positions with the same (relation, destination) form
 * one group: gid[E] = group of each position, gptr[R+1] = first group of each relation
 * (gptr[R] = number of groups over the scored relations), g_node[<= E] = destination (head)
 * of each group.  The attention kernels read gid in 16-byte pieces: allocate 16 entries of
 * slack after gid[E-1] holding valid group ids (e.g. 0). */
size_t kgat_head_groups_workspace_bytes(int64_t n_edges);
int kgat_head_groups(int64_t n_edges, int n_rel, const int32_t* rel_ptr, const int32_t* dst_g,
                     int32_t* gid, int32_t* gptr, int32_t* g_node, void* workspace,
                     size_t workspace_bytes, kgat_stream_t stream);

/* Same logits as kgat_att_score_f32, computed in two launches: tanh(ent[h] W_R[r] + rel[r]) once
 * per (head, relation) group into G_tab (n_groups x k floats, caller scratch), then per edge
 * the tail projection and its dot product with the group's row.  The head projection and all
 * tanh work are shared by the edges of a group (3.8 edges per group on the amazon-book-shaped
 * CKG).  Arithmetic per edge is unchanged.  Needs d == k in {16,32,64}
 * (kgat_att_score_split_supported).  logits (edge-id order) or logits_csr may be NULL. */
int kgat_att_score_split_supported(int64_t n_nodes, int d, int k, int n_rel);
int kgat_att_score_split_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                             const int32_t* rel_ptr, const int32_t* perm, const int32_t* src_g,
                             const int32_t* pos_g, const int32_t* gid, const int32_t* gptr,
                             const int32_t* g_node, int64_t n_groups, const float* ent,
                             const float* W_R, const float* rel, float* G_tab, float* logits,
                             float* logits_csr, kgat_stream_t stream);

/* Folded form of the same logits (reference models.py:135-144, restated).  The logit is bilinear
 * in the tail row: sum_j (ent[t] W_r)_j T_j = ent[t] . (W_r T) with T = tanh(ent[h] W_r + rel[r]),
 * so V[g] = W_r T (a d-vector) is computed once per (head, relation) group into V_tab
 * (n_groups x d floats, caller scratch) and an edge costs one d-length dot product.  Same
 * inputs and outputs as kgat_att_score_split_f32; the contraction order differs from the
 * reference's, so the results agree with the other forms to fp32 rounding (~1e-6 relative to
 * sum_j |t_j T_j|), not bit for bit.  Widths: d == k in {16,32,64,128} (MFMA), or d in {8,16,32}
 * with any k <= 32 (one thread per group; BASELINE configs[0] has d = k = 8).
 * flags: 0 or KGAT_ATT_F32_PRODUCTS. */
int kgat_att_score_folded_supported(int64_t n_nodes, int d, int k, int n_rel);
int kgat_att_score_folded_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                              const int32_t* rel_ptr, const int32_t* perm, const int32_t* src_g,
                              const int32_t* pos_g, const int32_t* gid, const int32_t* gptr,
                              const int32_t* g_node, int64_t n_groups, const float* ent,
                              const float* W_R, const float* rel, float* V_tab, float* logits,
                              float* logits_csr, int flags, kgat_stream_t stream);

/* Fused folded form: one launch, no V table.  Work tiles (graph-static, kgat_fold_tiles): at most
 * 16 consecutive head groups of one relation and at most `cap` grouped positions; a 16-group
 * block with more positions appears several times with consecutive position ranges.
 *   tiles    int32[4 * kgat_fold_tiles_max(...)]: (relation, first group, first position, end position)
 *   rel_tptr int32[R+1]: first tile of each relation; rel_tptr[R] = number of tiles
 * A wavefront computes a tile's V rows (as kgat_att_score_folded_f32), keeps them in LDS and
 * takes the dot products of the tile's positions itself.  Same logits as the folded form up to
 * the summation order of the d-length dot product (fp32 rounding).  cap: a positive multiple of
 * 64 (256 is the default of the host code).  Needs d == k in {16,32,64,128}; at d = k = 128 (one
 * 512-thread workgroup per CU holding W_r's three bf16 piece images, 96 KB, and a 16 x 128 V patch
 * per wave in LDS) only the bf16-piece products exist. */
int64_t kgat_fold_tiles_max(int64_t n_edges, int64_t n_groups, int n_rel, int cap);
size_t kgat_fold_tiles_workspace_bytes(int64_t n_groups, int n_rel);
int kgat_fold_tiles(int64_t n_edges, int n_rel, int64_t n_groups, const int32_t* rel_ptr, const int32_t* gid,
                    const int32_t* gptr, int cap, int groups_per_tile /* 16, or 32 for KGAT_ATT_TILES32 */,
                    int32_t* tiles, int32_t* rel_tptr, void* workspace, size_t workspace_bytes, kgat_stream_t stream);
/* Split of the tiles over the n_parts workgroups of the fused kernel (graph-static, like the
 * tiles): part b owns the contiguous tile range [part_tptr[b], part_tptr[b+1]), chosen on the
 * prefix sum of a per-tile cost
 *   cost_tile + cost_chunk * ceil(max(P - 64, 0) / 64) + (the tile opens its relation ? cost_relation : 0)
 * (P = positions of the tile) so that every workgroup gets the same cost, not the same tile
 * count.  The form follows what per-workgroup clock stamps showed on MI355X (least squares over
 * 3,072 workgroup samples, 1 % rms residual): a tile's MFMA phase is a constant (1,459 ticks of
 * workgroup time), its first 64 positions are free (their rows arrive during the MFMA phase),
 * every further chunk of 64 positions costs 267 ticks, and a relation change inside a
 * workgroup's range - W_r reload, two barriers, the prefetch pipeline of eight waves restarting -
 * costs 10,630 ticks, as much as seven tiles.  t_max = number of rows of `tiles`;
 * part_tptr[n_parts + 1]. */
size_t kgat_fold_tile_parts_workspace_bytes(int64_t t_max);
int kgat_fold_tile_parts(int64_t t_max, int n_rel, const int32_t* tiles, const int32_t* rel_tptr, int n_parts,
                         int cost_tile, int cost_chunk, int cost_relation, int32_t* part_tptr, void* workspace,
                         size_t workspace_bytes, kgat_stream_t stream);
int kgat_att_score_fused_supported(int64_t n_nodes, int d, int k, int n_rel);
/* What the fused kernel reads per grouped position, packed into one int32 (graph-static):
 *   rec_g[p] = src_g[p] | (((gid[p] - gptr[relation of p]) & 15) << 28)
 * i.e. the source (tail) node and the slot of the position's head group inside its 16-group block
 * (tiles are cut out of such blocks, whatever the cap).  Needs n_nodes <= 2^28.  Positions past
 * rel_ptr[R] (never scored) get their source node and slot 0. */
int kgat_att_pack_records(int64_t n_edges, int n_rel, const int32_t* rel_ptr, const int32_t* gptr,
                          const int32_t* gid, const int32_t* src_g, int groups_per_tile /* 16: slot << 28; 32: the
                          slot of the 32-group block << 27, node ids below 2^27 */, int32_t* rec_g, kgat_stream_t stream);
/* part_tptr / n_parts: the split above (one workgroup per part); NULL / 0: one workgroup per
 * compute unit, equal tile counts.  flags: 0 or KGAT_ATT_F32_PRODUCTS.
 * Outputs (any non-empty subset): logits_g[E] in grouped order (position p of the relation-grouped
 * list; coalesced stores - the form the propagation path takes: kgat_edge_softmax_f32 then reads
 * them through the CSR-position -> grouped-position map, see there), logits_csr[E] in CSR order
 * (needs pos_g), logits[E] in edge-id order (needs perm).  Never-scored positions get 0. */
int kgat_att_score_fused_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                             const int32_t* rel_ptr, const int32_t* perm, const int32_t* rec_g,
                             const int32_t* pos_g, const int32_t* gptr,
                             const int32_t* g_node, const int32_t* tiles, const int32_t* rel_tptr,
                             const int32_t* part_tptr, int n_parts,
                             const float* ent, const float* W_R, const float* rel, float* logits,
                             float* logits_csr, float* logits_g, int flags, kgat_stream_t stream);
/* Measurement aid: the same launch with every workgroup's start and end time written to part_clocks[2 b], [2 b + 1]
 * (100 MHz ticks of s_memrealtime; part b = workgroup b's tile range).  Same results; needs part_tptr; not with
 * KGAT_ATT_TILES32.  (What it showed, scripts/micro/att_rebalance_probe.py: the cost model of kgat_fold_tile_parts
 * leaves the workgroups' end times 8-9 % apart and the pattern repeats - correlation 0.98 run to run -, so ranges
 * re-cut from measured times take 7-9 % off the launch where the calibration and the use share their surroundings;
 * calibrated on the caller's first launches inside the benchmark step the gain was 1.5 %: not adopted.) */
int kgat_att_score_fused_timed_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                                   const int32_t* rel_ptr, const int32_t* perm, const int32_t* rec_g,
                                   const int32_t* pos_g, const int32_t* gptr,
                                   const int32_t* g_node, const int32_t* tiles, const int32_t* rel_tptr,
                                   const int32_t* part_tptr, int n_parts,
                                   const float* ent, const float* W_R, const float* rel, float* logits,
                                   float* logits_csr, float* logits_g, int flags, long long* part_clocks,
                                   kgat_stream_t stream);

/* ---------------------------------------------------------------- edge softmax (A3)
 * Replaces dgl.nn.pytorch.softmax.edge_softmax (call site reference models.py:153):
 *   a[e] = exp(s[e] - max_{e'->dst e} s[e']) / sum_{e'->dst e} exp(s[e'] - max)
 * over the incoming edges of each destination, all relations together, for the CSR
 * positions [e_begin, e_end) (the whole graph: 0, E; a destination-range shard: its
 * indptr range).  row_of = destination id per CSR position.
 * Input: logits in edge-id order (logits_in_csr_order = 0; read through eid) or in CSR
 * position order (1).  Outputs (either may be NULL, not both): out in edge-id order (written
 * through eid) and out_csr in CSR order.  eid = original edge id per CSR position.
 * One sweep over the positions (rows finished inside a wavefront's 512- or 1,024-position range
 * are normalised on the spot; the positions of the rows the range cuts are stored as
 * exp(s - partial max)) plus one short launch in which every wavefront combines, for the rows
 * its range cuts, the carry entries of the row's whole chain in a fixed order and rescales its
 * own positions of them (it reads the outputs, not the logits).  No atomics, fixed combination
 * order: bitwise reproducible, no bound on a row's length.  Full ranges move as 16-byte loads /
 * stores when row_of, eid, out_csr (and CSR-ordered logits) are 16-byte aligned and e_begin is a
 * multiple of 4; otherwise position by position, same results.  indptr[N+1] is the CSR row pointer (the extent of a cut row - which
 * wavefronts hold its carries - is read from it); every row that has a position in
 * [e_begin, e_end) must lie inside it completely (true for the whole graph and for a
 * destination-range shard).  Workspace: n_edges = e_end - e_begin. */
size_t kgat_edge_softmax_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int kgat_edge_softmax_f32(int64_t n_nodes, int64_t e_begin, int64_t e_end, const int32_t* indptr,
                          const int32_t* row_of, const int32_t* eid, const float* logits,
                          int logits_in_csr_order, float* out, float* out_csr, void* workspace,
                          size_t workspace_bytes, kgat_stream_t stream);
/* The same operator as three streaming passes (integer-ordered atomic row max, row sum in 2^-40
 * fixed point with 64-bit atomic adds, normalise): an independent implementation kept for
 * cross-checks and A/B timing.  At most 2^24 positions per destination. */
size_t kgat_edge_softmax_3pass_workspace_bytes(int64_t n_nodes);
int kgat_edge_softmax_3pass_f32(int64_t n_nodes, int64_t e_begin, int64_t e_end,
                                const int32_t* row_of, const int32_t* eid, const float* logits,
                                int logits_in_csr_order, float* out, float* out_csr, void* workspace,
                                size_t workspace_bytes, kgat_stream_t stream);

/* Backward of the above (DGL 0.4.x EdgeSoftmax.backward), rows [row0, row0 + n_rows):
 *   grad_s[e] = a[e]*g[e] - a[e] * sum_{e'->dst e} a[e']*g[e'];  arrays in edge-id order
 * (eid != NULL) or CSR order (eid == NULL). */
int kgat_edge_softmax_bwd_f32(int64_t n_rows, int64_t row0, const int32_t* indptr,
                              const int32_t* eid, const float* a, const float* grad_a,
                              float* grad_logits, kgat_stream_t stream);

/* ---------------------------------------------------------------- aggregation (S1, S1b)
 * Replaces g.update_all(fn.u_mul_e('h','w','m'), fn.sum('m','h_neighbor')) of reference
 * models.py:63 (DGL binary_reduce(sum, mul, SRC, EDGE) with (E,1) broadcast):
 *   out[v - row0, :] = sum_{p in [indptr[v], indptr[v+1])} w_p * X[col[p], :]
 * for rows v in [row0, row0 + n_rows), whose CSR positions are [e_begin, e_end) =
 * [indptr[row0], indptr[row0 + n_rows]) (passed by value so the call needs no device read).
 * w_p = w[eid[p]] if eid != NULL (w in edge-id order) else w[p] (w in CSR order).
 * Rows with no in-edge are written as 0.  Summation order is fixed, no float atomics:
 * results are bitwise reproducible.  X has D columns, fp32 row-major, 16-byte aligned.
 * row_of (destination per CSR position) is required by the MERGE algorithm; `order` (a row
 * schedule from kgat_row_order_by_degree over the same row range, entries relative to
 * row0) only applies to ROWS.  workspace: kgat_spmm_workspace_bytes(e_end - e_begin, D).
 * self_out (may be NULL; with KGAT_SPMM_MUL_SELF, CSR-ordered weights, MERGE / AUTO): the launch also
 * writes X[v, :] to self_out[(v - row0) * self_stride + 0..D) - the ego block [h0 | ...] of
 * Model.gnn's readout (models.py:159,168) from the register that already holds the row, instead of a
 * separate copy pass; self_stride in floats, a multiple of 4, self_out 16-byte aligned.
 * Backward w.r.t. X (S1b) is this same call on the CSR of the reversed graph. */
size_t kgat_spmm_workspace_bytes(int64_t n_edges, int D);
int kgat_spmm_umule_sum_f32(int64_t n_rows, int64_t row0, int64_t e_begin, int64_t e_end, int D,
                            const int32_t* indptr, const int32_t* col, const int32_t* row_of,
                            const int32_t* eid, const float* X, const float* w, float* out,
                            const int32_t* order, void* workspace, size_t workspace_bytes,
                            unsigned flags, int algo, float* self_out, int64_t self_stride,
                            kgat_stream_t stream);

/* Measurement aid, not an operator of the path (SURVEY 8d asks for the bound that binds; on the cache-resident CKGs
 * that is the rate at which gathered rows cross the cache fabric, not HBM): reads the rows X[col[p], :] of CSR
 * positions p in [0, n_edges) with the aggregation's access pattern - D / 4 lanes x 16 bytes per row, 8 rows in
 * flight per lane group - and does nothing else (no weights, no output).  sink: a scratch of at least
 * ceil(n_edges / 2048) * 4 x 16 bytes that is never written for finite data.  D in {16, 32, 64, 128}. */
int kgat_gather_probe_f32(int64_t n_edges, int D, const int32_t* col, const float* X, float* sink,
                          kgat_stream_t stream);

/* Gradient of the aggregation w.r.t. the edge weight (DGL backward_rhs of the same op):
 *   grad_w[e] = < X[src e, :], grad_out[dst e, :] >,  edge-id order. */
int kgat_sddmm_dot_f32(int64_t n_edges, int D, const int32_t* src, const int32_t* dst,
                       const float* X, const float* grad_out, float* grad_w,
                       kgat_stream_t stream);

/* ---------------------------------------------------------------- bi-interaction (B1 + B2)
 * Forward of the dense part of KGATConv (reference models.py:66) fused with the readout
 * normalisation (models.py:165-167):  Z = LeakyReLU_slope(P @ W2^T), P = h * h_N (n_rows x d_in,
 * produced by the aggregation with KGAT_SPMM_MUL_SELF), W2 = res_fc_2.weight (d_out x d_in).
 * h_out (n_rows x d_out, may be NULL) receives Z (the input of the next layer); norm_out (may
 * be NULL) receives Z / max(||Z_row||_2, 1e-12) with row stride norm_stride floats (a column
 * slice of the concatenated output).  Widths (kgat_bi_interaction_supported): d_in, d_out in
 * {16, 32, 64, 128} (MFMA kernel), or one of them in {4, 8} with the other in {4, 8, 16, 32} (one
 * lane per row). */
int kgat_bi_interaction_supported(int d_in, int d_out);
int kgat_bi_interaction_f32(int64_t n_rows, int d_in, int d_out, const float* P, const float* W2,
                            float negative_slope, float* h_out, float* norm_out,
                            int64_t norm_stride, kgat_stream_t stream);

/* The same with the product formed on the way (round 4): Z = LeakyReLU_slope((H * HN) @ W2^T) - reference
 * models.py:66's th.mul(g.ndata['h'], g.ndata['h_neighbor']) + res_fc_2 + LeakyReLU, and models.py:165's normalize -
 * from the layer input H and the plain aggregation HN = update_all(u_mul_e, sum) (both n_rows x d_in), so that the
 * aggregation needs no epilogue (its KGAT_SPMM_MUL_SELF form pays a dependent load per finished row inside the edge
 * loop: 91 vs 78 us on the benchmark graph).  self_out (may be NULL): the rows of H are also copied to
 * self_out[row * self_stride + 0..d_in) - the ego block [h0 | ...] of Model.gnn's readout (models.py:159,168);
 * 16-byte aligned, self_stride a multiple of 4 floats.  Same bits as KGAT_SPMM_MUL_SELF + kgat_bi_interaction_f32. */
int kgat_bi_interaction_mul_f32(int64_t n_rows, int d_in, int d_out, const float* H, const float* HN, const float* W2,
                                float negative_slope, float* h_out, float* norm_out, int64_t norm_stride,
                                float* self_out, int64_t self_stride, kgat_stream_t stream);

/* The pair update_all(u_mul_e, sum) -> th.mul / res_fc_2 / LeakyReLU / normalize (reference models.py:63-66, :165) with
 * one launch less (round 4).  The aggregation's MERGE algorithm cuts the CSR positions into tiles of
 * kgat_spmm_tile_edges(e_end - e_begin, D) edges; a tile leaves its first and its last row as partial sums in the
 * workspace and a second, dependent launch adds them up and zero-fills the rows without in-edges.  With
 * KGAT_SPMM_DEFER_FINISH (plain operator, CSR-ordered weights, MERGE / AUTO, D in {16, 32, 64, 128}) that launch is
 * skipped: those rows of `out` are NOT written, and kgat_bi_interaction_mul_deferred_f32 - the same operator as
 * kgat_bi_interaction_mul_f32 - forms them on the way from `indptr_rows` (= indptr + row0: the row offsets of the
 * call's rows 0 .. n_rows), the edge range and the untouched workspace, in the second launch's order of additions:
 * bit-identical results.  d_in, d_out in {16, 32, 64, 128}; nothing else may use the workspace between the two calls.
 * (Benchmark graph: the three second launches cost 13 us of a 0.44 ms step; forming their rows costs the dense kernels 6.) */
int kgat_spmm_tile_edges(int64_t n_edges, int D);  /* 0: D outside {16, 32, 64, 128} */
int kgat_bi_interaction_mul_deferred_f32(int64_t n_rows, int d_in, int d_out, const float* H, const float* HN,
                                         const float* W2, float negative_slope, float* h_out, float* norm_out,
                                         int64_t norm_stride, float* self_out, int64_t self_stride,
                                         const int32_t* indptr_rows, int64_t e_begin, int64_t e_end,
                                         const void* spmm_workspace, int tile_edges, kgat_stream_t stream);

/* ---------------------------------------------------------------- one KGATConv forward in one pass (S1 + B1 + B2)
 * Replaces reference models.py:63-66 (update_all(u_mul_e, sum); th.mul; res_fc_2; LeakyReLU) and the
 * F.normalize of models.py:165 for the no-grad forward: kgat_spmm_umule_sum_f32 with KGAT_SPMM_MUL_SELF
 * (same arguments, CSR-ordered weights, MERGE algorithm) whose completed rows P = h * h_N never leave the
 * workgroup: they are collected in LDS and take Z = LeakyReLU_slope(P W2^T) (fp32 MFMA) there; h_out /
 * norm_out / norm_stride as kgat_bi_interaction_f32 (norm_out 16-byte aligned, norm_stride a multiple of
 * 4).  Results are bit-identical to the two-launch sequence.  `scratch` (n_rows x d_in floats, contents
 * undefined afterwards) takes the P rows of tiles that span more rows than the LDS buffer holds.
 * Widths (kgat_spmm_bi_fused_supported): d_in, d_out in {16, 32, 64} with d_out <= d_in.
 * workspace: kgat_spmm_workspace_bytes(e_end - e_begin, d_in). */
int kgat_spmm_bi_fused_supported(int d_in, int d_out);
int kgat_spmm_bi_fused_f32(int64_t n_rows, int64_t row0, int64_t e_begin, int64_t e_end, int d_in, int d_out,
                           const int32_t* indptr, const int32_t* col, const int32_t* row_of,
                           const float* X, const float* w, const float* W2, float negative_slope,
                           float* h_out, float* norm_out, int64_t norm_stride, float* scratch,
                           void* workspace, size_t workspace_bytes, float* self_out, int64_t self_stride,
                           kgat_stream_t stream);

/* F.normalize(x, p=2, dim=1, eps=1e-12) of n_rows contiguous d-wide rows (reference
 * models.py:165) into a destination with row stride out_stride floats (a column slice of the
 * concatenated output, models.py:167).  Used where the rows of a layer come back from the
 * multi-GPU exchange and the fused kernel above could not normalise them. */
int kgat_l2_normalize_rows_f32(int64_t n_rows, int d, const float* x, float* out, int64_t out_stride,
                               kgat_stream_t stream);

/* The concatenated readout of Model.gnn (reference models.py:159-168: [h0 | normalize(h1) | ...]) from
 * n_blocks (<= 8) separately held row-major blocks in one pass: block k is n_rows x widths[k]
 * (widths multiples of 4, <= 128; `blocks` and `widths` are HOST arrays, the pointers in `blocks`
 * device pointers), copied as is or L2-normalised per row (normalize[k] != 0, eps 1e-12) into
 * columns [sum of the earlier widths, +widths[k]) of `out` (row stride out_stride floats). */
int kgat_readout_concat_f32(int64_t n_rows, int n_blocks, const float* const* blocks, const int* widths,
                            const int* normalize, float* out, int64_t out_stride, kgat_stream_t stream);

/* Permute a per-edge array: out[i] = in[index[i]]. */
int kgat_gather_f32(int64_t n, const int32_t* index, const float* in, float* out,
                    kgat_stream_t stream);
int kgat_gather_i32(int64_t n, const int32_t* index, const int32_t* in, int32_t* out,
                    kgat_stream_t stream);

/* ---------------------------------------------------------------- training form of the layer (8f #1)
 * Forward of one KGATConv under autograd (reference models.py:63-70 with mess_drop active):
 *   h_out = dropout_p(LeakyReLU((H * HN) W2^T)),  norm_out = F.normalize(h_out)
 * H * HN is formed while the rows are loaded.  The dropout mask is a counter-based hash of
 * (seed, (row0 + row) * d_out + column): keep <=> hash >= p * 2^32, kept values scaled by 1/(1-p);
 * the backward recomputes it from the same seed.  row0 = global index of the first row when H is a
 * row range of a larger matrix (a destination shard draws the mask of the unsharded layer; 0
 * otherwise).  Widths as kgat_bi_interaction_supported. */
int kgat_bi_interaction_train_f32(int64_t n_rows, int d_in, int d_out, const float* H, const float* HN,
                                  const float* W2, float negative_slope, float drop_p, uint64_t seed, int64_t row0,
                                  float* h_out, float* norm_out, int64_t norm_stride, float* self_out,
                                  int64_t self_stride, kgat_stream_t stream);
/* (self_out / self_stride as in kgat_bi_interaction_mul_f32: the rows of H copied into a column slice of the readout
 * on the way - the ego block of reference models.py:159,168, which the training stack wrote with a separate copy pass.)
 * kgat_add3_rows_f32: out = (a + b) + c over n_rows x d, a being a column slice (rows of a_stride floats) of a wider
 * matrix: the gradient arriving at the embedding table through Model.gnn's three paths (the ego block of the readout's
 * gradient, the aggregated branch, the elementwise branch of layer 0) in one pass instead of two. */
int kgat_add3_rows_f32(int64_t n_rows, int d, const float* a, int64_t a_stride, const float* b, const float* c, float* out,
                       kgat_stream_t stream);
/* Backward head of the same layer: with y = h_out saved by the forward,
 *   grad_z = [grad_a + grad_b + normalize_bwd(grad_norm; y)] * mask/(1-p) * LeakyReLU'(z)
 * (grad_a, grad_b: gradients arriving at h_out from the next layer, either may be NULL;
 * grad_norm: gradient of the normalised copy, row stride grad_norm_stride, may be NULL).
 * The rest of the backward is dense: grad_P = grad_z W2, grad_W2 = grad_z^T (H * HN), then
 * grad_H = grad_P * HN + A^T (grad_P * H) (kgat_mul2_f32 + the SpMM on the reversed CSR). */
int kgat_bi_interaction_bwd_pre_f32(int64_t n_rows, int d_out, const float* h_out, const float* grad_a,
                                    const float* grad_b, const float* grad_norm, int64_t grad_norm_stride,
                                    float negative_slope, float drop_p, uint64_t seed, int64_t row0, float* grad_z,
                                    kgat_stream_t stream);
/* The same two steps in one pass (round 4): grad_P = grad_z W2 formed per 16-row tile on the fp32 MFMA and never
 * written; grad_hn_times_h = grad_P * H (what the reversed-CSR SpMM then aggregates: the gradient through h_N, reference
 * models.py:63,66 under autograd) and grad_h_direct = grad_P * HN (the gradient through the row's own features), both
 * n_rows x d_in.  grad_z n_rows x d_out, W2 = res_fc_2.weight (d_out x d_in).  d_in, d_out in {16, 32, 64, 128}. */
int kgat_bi_interaction_bwd_input_supported(int d_in, int d_out);
int kgat_bi_interaction_bwd_input_f32(int64_t n_rows, int d_in, int d_out, const float* grad_z, const float* W2,
                                      const float* H, const float* HN, float* grad_hn_times_h, float* grad_h_direct,
                                      kgat_stream_t stream);
/* grad_W2 = grad_z^T (H * HN) (d_out x d_in; reference models.py:66's res_fc_2 under autograd) as per-workgroup
 * partial sums: partials[b] (b < n_partials = kgat_bi_interaction_bwd_weight_partials(n_rows), each d_out x d_in row-major)
 * is the product over the 64-row slabs b, b + n_partials, ...; the caller adds the partials up (any fixed order:
 * reproducible).  The product H * HN is formed on the way and never written.  Widths as kgat_bi_interaction_bwd_input. */
int64_t kgat_bi_interaction_bwd_weight_partials(int64_t n_rows);
int kgat_bi_interaction_bwd_weight_f32(int64_t n_rows, int d_in, int d_out, const float* grad_z, const float* H,
                                       const float* HN, float* partials, int64_t n_partials, kgat_stream_t stream);
/* out_s = the sum of set s's n_partials_s partials (each n_elems_s floats, a multiple of 4, back to back) for up to four
 * sets in ONE launch - the weight gradients of a propagation stack's layers, whose partials
 * kgat_bi_interaction_bwd_weight_f32 leaves to the caller.  HOST arrays of n_sets entries.  A fixed order of additions
 * (sixteen strided lane sums, then a shuffle tree): bitwise reproducible; not the order of a sequential sum. */
int kgat_sum_partials_f32(int n_sets, const float* const* partials_host, float* const* out_host,
                          const int64_t* n_partials_host, const int64_t* n_elems_host, kgat_stream_t stream);
/* ab = a * b and ac = a * c elementwise in one pass (n a multiple of 4). */
int kgat_mul2_f32(int64_t n, const float* a, const float* b, const float* c, float* ab, float* ac,
                  kgat_stream_t stream);

/* ---------------------------------------------------------------- TransR KG step
 * Loss and gradients of reference models.py:114-133 (transR; bmm_maybe_select :13-47,
 * _L2_loss_mean :9-11) for one batch of triplets (h[b], r[b], pos_t[b], neg_t[b]):
 *   a_x = ent[x] W_R[r],  u_x = a_x / max(|a_x|, 1e-12),  u_r = rel[r] / max(|rel[r]|, 1e-12)
 *   loss = mean_b softplus(|u_h+u_r-u_p|^2 - |u_h+u_r-u_n|^2)
 *          + reg_lambda * sum_{v in h,r,p,n} mean_b |u_v|^2 / 2
 * loss: 1 float.  grad_ent (n_nodes x d, dense: rows not in the batch are zeroed, as the
 * reference's non-sparse nn.Embedding gradient), grad_W (R x d x k), grad_rel (R x k): all three
 * or none (NULL: loss only).  Fixed summation orders (sorted batch, no atomics): bitwise
 * reproducible.  Needs d, k multiples of 4 and <= 128, 3*batch <= 8192 (the batch is sorted by one workgroup in
 * LDS), n_rel <= 4096, n_nodes < 2^31 (kgat_transr_supported; beyond it the Python layer takes torch operators). */
int kgat_transr_supported(int64_t n_nodes, int d, int k, int n_rel, int64_t batch);
size_t kgat_transr_workspace_bytes(int64_t batch, int d, int k, int n_rel);
int kgat_transr_loss_grad_f32(int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h,
                              const int32_t* r, const int32_t* pos_t, const int32_t* neg_t, const float* ent,
                              const float* W_R, const float* rel, float reg_lambda, float* loss, float* grad_ent,
                              float* grad_W, float* grad_rel, void* workspace, size_t workspace_bytes,
                              kgat_stream_t stream);
/* The same computation in the two halves autograd asks for.  kgat_transr_forward_f32: the loss, with the per-sample
 * rows and the weight-gradient partials left in `workspace`; kgat_transr_backward_f32 (same batch arrays, same
 * workspace, untouched in between): the three gradients, multiplied by grad_scale[0] - a DEVICE scalar, the gradient
 * arriving at the loss (NULL = 1) - inside the final ordered reductions: no host synchronisation and no extra pass
 * over the dense n_nodes x d gradient (round 4 multiplied the three gradients by it in three torch launches, 22 us
 * of the 0.3 ms KG step).  Same bits as kgat_transr_loss_grad_f32 for grad_scale = 1. */
int kgat_transr_forward_f32(int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h, const int32_t* r,
                            const int32_t* pos_t, const int32_t* neg_t, const float* ent, const float* W_R,
                            const float* rel, float reg_lambda, float* loss, void* workspace, size_t workspace_bytes,
                            kgat_stream_t stream);
int kgat_transr_backward_f32(int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h, const int32_t* r,
                             const int32_t* pos_t, const int32_t* neg_t, const float* grad_scale, float* grad_ent,
                             float* grad_W, float* grad_rel, void* workspace, size_t workspace_bytes,
                             kgat_stream_t stream);

/* The KG phase of an epoch as three-launch iterations (round 6; reference kgat.py:116-136: for every sampled batch
 * transR -> backward -> optimizer.step -> zero_grad; 1,641 iterations per epoch on the amazon-book shape).
 *
 * kgat_transr_presort_f32: the sorts an iteration needs depend only on the batch's ids, so the batches of a whole
 * phase - h, r, pos_t, neg_t as n_batches x batch row-major arrays - are sorted by ONE launch (samples by relation
 * with the chunk table of the weight-gradient partials; the 3 x batch entity ids).  Batch b's result is the block of
 * kgat_transr_sorted_bytes(batch, n_rel) bytes at sorted + b * that.
 *
 * kgat_transr_adam_step_f32: one iteration on batch arrays h .. neg_t (batch ids each) and that batch's `sorted`
 * block: the loss (1 float at `loss`) and the step of torch.optim.Adam on the three parameters the loss reaches -
 * ent (n_nodes x d), W_R (n_rel x d x k), rel (n_rel x k), updated IN PLACE with their moments (`exp_avg_host`,
 * `exp_avg_sq_host`: HOST arrays of three device pointers in that order; `steps_host`: the three step counts AFTER
 * this step).  Dense semantics as kgat_adam_step_f32 (every row of the table moves), without a dense gradient and
 * without a scatter launch: the per-sample kernel writes its three gradient rows at their SORTED positions (the presort
 * also emits the inverse permutation and the run lengths), workgroups in the same launch tag the batch's entities in
 * `row_slot`, n_nodes 64-bit words owned by the caller - zero before the first call, never to be cleared - in which a
 * word is valid for the call whose `tag` it carries (pass a tag in [1, 2^50) that differs from every earlier call's on
 * the same row_slot: a counter), and the Adam launch takes g = 0 for an untagged row and the sum of the row's run, in
 * sorted order, for a tagged one.  The bits of kgat_transr_loss_grad_f32 followed by kgat_adam_step_f32, i.e. of the
 * reference's loss.backward(); optimizer.step().  All pointers 16-byte aligned.  Launches: per-sample kernel (+ tags),
 * weight-gradient partials (a chunk's output tiles shared by up to four workgroups) + loss, Adam. */
size_t kgat_transr_sorted_bytes(int64_t batch, int n_rel);
int kgat_transr_presort_f32(int64_t n_nodes, int n_rel, int64_t n_batches, int64_t batch, const int32_t* h,
                            const int32_t* r, const int32_t* pos_t, const int32_t* neg_t, void* sorted,
                            size_t sorted_bytes, kgat_stream_t stream);
size_t kgat_transr_step_workspace_bytes(int64_t batch, int d, int k, int n_rel);
int kgat_transr_adam_step_f32(int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h, const int32_t* r,
                              const int32_t* pos_t, const int32_t* neg_t, const void* sorted, float* ent, float* W_R,
                              float* rel, float* const* exp_avg_host, float* const* exp_avg_sq_host,
                              const int64_t* steps_host, double lr, double beta1, double beta2, double eps,
                              float reg_lambda, float* loss, uint64_t* row_slot, uint64_t tag, void* workspace,
                              size_t workspace_bytes, kgat_stream_t stream);

/* ---------------------------------------------------------------- evaluation
 * recall@K / ndcg@K of reference metric.py:36-68 (calc_recall_ndcg with one_recall_at_k :5-7, one_dcg_at_k :8-22
 * method 1, one_ndcg_at_k :23-34) for a list of test users, without a users x items score matrix:
 *   score[u][i] = <emb[user_ids[u]], emb[item_ids[i]]>  (metric.py:48; v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains)
 *   score[u][train items of u] = 0.0                     (metric.py:50 - set to 0.0, not removed)
 *   rank = the K first of a descending sort              (metric.py:51; equal scores: lower item position first)
 *   recall[u] = hits / |test items of u| (0 for an empty list), ndcg[u] = dcg / dcg of the user's own sorted hit list
 * `emb` (rows of emb_stride floats, the first F used) holds users and items; an item's POSITION i in item_ids is what
 * the train / test lists hold (the reference indexes `score` with the raw item ids of its dicts).  train_ptr /
 * test_ptr [n_users + 1] + train_items / test_items: CSR over the users in user_ids order, every list ASCENDING.
 * kgat_eval_items_kmajor_f32 first lays the item rows out as MFMA fragments (itemT: kgat_eval_items_elems floats).
 * disc[K] = 1 / log2(k + 2) in fp64 (host-computed, so the device divides by the very values numpy uses).
 * recall_out / ndcg_out: n_users doubles; topk_out (optional, n_users x K): the ranked item positions.
 * Needs 1 <= K <= 32, n_items >= K, F <= ~1100 (kgat_eval_supported).  Bitwise reproducible. */
int kgat_eval_supported(int F, int K);
int64_t kgat_eval_items_elems(int64_t n_items, int F);
int kgat_eval_items_kmajor_f32(int64_t n_items, int F, const float* emb, int64_t emb_stride, const int32_t* item_ids,
                               float* itemT, kgat_stream_t stream);
size_t kgat_eval_workspace_bytes(int64_t n_users, int64_t n_items, int F, int K);
int kgat_eval_recall_ndcg_f32(int64_t n_users, const int32_t* user_ids, int64_t n_items, int F, const float* emb,
                              int64_t emb_stride, const float* itemT, const int32_t* train_ptr,
                              const int32_t* train_items, const int32_t* test_ptr, const int32_t* test_items, int K,
                              const double* disc, void* workspace, size_t workspace_bytes, double* recall_out,
                              double* ndcg_out, int32_t* topk_out, kgat_stream_t stream);

/* ---------------------------------------------------------------- optimiser of the training loop (8f #1, #3)
 * One step of torch.optim.Adam (reference kgat.py:85: optim.Adam(model.parameters(), lr); amsgrad off, no weight
 * decay) over up to kgat_adam_max_tensors() parameter tensors in ONE launch - dense semantics: every element moves,
 * also where the gradient is zero.  Per element in fp32, in torch's order of operations:
 *   m <- m + (1 - beta1)(g - m);  v <- v beta2 + (1 - beta2) g g;
 *   p <- p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * `*_host` arrays are HOST arrays of n_tensors entries (device pointers / element counts / the tensor's step count
 * t >= 1 AFTER this step, which may differ between tensors: a parameter that had no gradient in some steps lags).
 * zero_grads != 0 also clears the gradients it has read (optimizer.zero_grad(set_to_none=False) in the same pass). */
int kgat_adam_max_tensors(void);
int kgat_adam_step_f32(int n_tensors, const int64_t* sizes_host, float* const* params_host, float* const* grads_host,
                       float* const* exp_avg_host, float* const* exp_avg_sq_host, const int64_t* steps_host, double lr,
                       double beta1, double beta2, double eps, int zero_grads, kgat_stream_t stream);

/* ---------------------------------------------------------------- BPR loss of the CF phase (8f #1)
 * reference models.py:170-178 (get_loss; _L2_loss_mean :9-11) on the readout `emb` (n_nodes rows of emb_stride
 * floats, the first F used; F and emb_stride multiples of 4):
 *   loss = -mean_b logsigmoid(<s_b,p_b> - <s_b,n_b>) + reg_lambda (mean_b |s_b|^2/2 + mean_b |p_b|^2/2 + mean_b |n_b|^2/2)
 * with s, p, n = rows u[b], p[b], n[b] (ids in [0, n_nodes); an id outside makes the loss NaN, nothing is accessed).
 * kgat_bpr_loss_f32: loss (1 float) and coef[batch] = sigmoid(-(x_b)) for the backward.
 * kgat_bpr_grad_f32: d loss / d emb as a DENSE n_nodes x F matrix (rows outside the batch zero - what torch's
 * index backward produces), times grad_scale[0] (DEVICE scalar = the gradient arriving at the loss; NULL = 1): rows
 * that occur several times are summed after a stable sort of the 3 x batch row ids: in sample order inside a window of
 * 32 sorted positions, a row that spans several windows as its window pieces in window order - a fixed order of
 * additions, no float atomics, bitwise reproducible, and bounded work per wavefront whatever the popularity of a row.
 * Same workspace size for both (it depends on F since ABI version 9: the windows' partial rows). */
size_t kgat_bpr_workspace_bytes(int64_t batch, int F);
int kgat_bpr_loss_f32(int64_t n_nodes, int F, const float* emb, int64_t emb_stride, int64_t batch, const int32_t* u,
                      const int32_t* p, const int32_t* n, float reg_lambda, float* loss, float* coef, void* workspace,
                      size_t workspace_bytes, kgat_stream_t stream);
int kgat_bpr_grad_f32(int64_t n_nodes, int F, const float* emb, int64_t emb_stride, int64_t batch, const int32_t* u,
                      const int32_t* p, const int32_t* n, const float* coef, float reg_lambda, const float* grad_scale,
                      float* grad, void* workspace, size_t workspace_bytes, kgat_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KGAT_HIP_H_ */
