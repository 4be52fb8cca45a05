/*
 * desco_hip.h -- C ABI of libdesco_hip.so, the MI355X (gfx950) native layer of the DeSCo hot path.
 *
 * The reference (fuvty/DeSCo) has no FFI: its hot path is Python calling PyG / torch ops.  This
 * header is the boundary a maintainer would bind (ctypes stub in INTEGRATION.md) to replace those
 * call sites.  Each entry point names the reference call site it replaces (paths relative to the
 * reference root, see SURVEY.md section 8a for the row ids A1..A15 / K1..K24).
 *
 * Conventions
 *   - every function returns 0 on success, otherwise a hipError_t value (device entry points) or a
 *     negative DESCO_E* code (argument errors); desco_last_error() gives a message (thread local).
 *   - device entry points take DEVICE pointers and a hipStream_t passed as void* (NULL = default
 *     stream); they only enqueue work, never allocate, never synchronise (graph-capture safe).
 *   - host entry points (desco_partition_*) take HOST pointers.
 *   - the hidden width H is fixed at 64 (reference default --neigh_hidden_dim/--gossip_hidden_dim,
 *     config.py:250,316); feature rows are fp32.
 *   - "virtual row" CSR: destination row i with relation slot s is virtual row i*S+s.
 */
#ifndef DESCO_HIP_H
#define DESCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DESCO_ABI_VERSION 6
#define DESCO_H 64

#define DESCO_EINVAL (-1)
#define DESCO_ENOMEM (-2)

#define DESCO_ACT_NONE 0
#define DESCO_ACT_RELU 1
#define DESCO_ACT_LEAKY 2 /* slope given separately */

typedef void* desco_stream_t;

int desco_abi_version(void);
/* number of visible HIP devices, or a negative hipError_t when the runtime cannot initialise */
int desco_device_count(void);
const char* desco_last_error(void);

/* ------------------------------------------------------------------------------------------
 * HOST: canonical-partition builder.
 * Replaces NeighborhoodDataset.process (workload.py:243-294: get_neigh_hetero data.py:375-396,
 * k_neigh data.py:329-338, NetworkxToHetero transforms.py:319-412), the per-item ToTconvHetero
 * transform (transforms.py:180-255) and PyG's hetero collate, for a whole dataset at once.
 *
 * Input: G graphs stored as one CSR over global node ids (graph g owns nodes
 * graph_ptr[g]..graph_ptr[g+1]-1, ascending = the reference's nx node order); adjacency must be
 * symmetric, without self loops or duplicates, each row sorted ascending.
 *
 * Output (after desco_partition_sizes / desco_partition_export):
 *   B neighborhoods (one per node whose canonical neighborhood has >= 1 edge), N_c count rows.
 *   Row order: count rows of neighborhood 0, 1, ... (ascending original id inside each), then
 *   the B canonical rows; canonical row of neighborhood b is N_c + b.
 *   neigh_index[B][2] = (graph id, node id inside the graph)   (nx_neighs_index, workload.py:189-195)
 *   indicator[num_nodes]                                       (nx_neighs_indicator)
 *   count_ptr[B+1]: count rows of neighborhood b are count_ptr[b]..count_ptr[b+1]-1
 *   count_orig[N_c]: global node id of every count row
 *   vrowptr[4*(N_c+B)+1], vcol[E]: destination-major CSR with 4 relation slots per row,
 *      slot = 2*(source is the canonical node) + (edge is a "tride" edge, i.e. NOT in a triangle);
 *      vcol holds source ROW ids, ascending inside a slot.
 * quirk_batch > 0 emulates PyG's remove_self_loops on the bipartite edge types for reference
 * batches of quirk_batch consecutive neighborhoods (gnn_model.py:389-390); 0 = off.
 * ------------------------------------------------------------------------------------------ */
typedef struct desco_partition desco_partition;

int desco_partition_build(const int64_t* graph_ptr, int64_t num_graphs, const int64_t* rowptr,
                          const int32_t* col, int depth, int quirk_batch, int num_threads,
                          desco_partition** out);
int desco_partition_sizes(const desco_partition* p, int64_t* num_neigh, int64_t* num_count,
                          int64_t* num_edges, int64_t* num_nodes);
int desco_partition_export(const desco_partition* p, int64_t* neigh_index, uint8_t* indicator,
                           int32_t* count_ptr, int32_t* count_orig, int32_t* vrowptr,
                           int32_t* vcol);
void desco_partition_free(desco_partition* p);

/* Optional re-ordering of a block's count rows (host arrays, same layout in and out): inside every neighborhood the
 * count rows are sorted by their number of count -> count sources (the neighborhood's heavier relation slot first,
 * descending when neigh_key[b] is even, ascending when odd -- pass the canonical node's id inside its graph so that the
 * order is a property of the neighborhood, not of its place in a block or of the shard its graph is in; NULL = the index b), vcol is relabelled and kept ascending inside a slot, count_orig
 * follows the rows; count_ptr, the canonical rows and every per-neighborhood quantity are unchanged.  Row order inside a
 * neighborhood is this library's convention (data.py:375-396 leaves it to CPython set order), and the fused layer kernel
 * needs as many gather steps per 16-row tile as the tile's highest-degree row: -9 % on Syn_1827 shapes. */
int desco_partition_degree_sort(const int32_t* count_ptr, int64_t num_neigh, const int32_t* vrowptr,
                                const int32_t* vcol, const int32_t* count_orig, int32_t* count_orig_out,
                                int32_t* vrowptr_out, int32_t* vcol_out, const int64_t* neigh_key, int num_threads);

/* ------------------------------------------------------------------------------------------
 * DEVICE: the same canonical-partition builder on the GPU (csrc/partition_dev.hip), one wavefront
 * per target node, output streamed into the flat 4-slot CSR in device memory.
 * Inputs (device): graph_ptr[G+1] int64, node_graph[V] (graph id of every node), the CSR
 * rowptr[V+1] / col over global node ids (int32; symmetric, loop-free, rows sorted ascending);
 * n_max = nodes of the largest graph (sizes the per-wave LDS workspace: graphs above ~4400 nodes
 * return DESCO_EINVAL -> use the host builder); num_waves (multiple of 4) = wavefronts to launch.
 *   1. desco_partition_dev_count -> nsize[V] (0 = node skipped), ecnt_count[V], ecnt_canon[V]
 *   2. desco_partition_dev_scan  -> exclusive scans b_index / row_off / eoff_count / eoff_canon [V]
 *      and totals4 = (B, N_c, E_count, E_canon); the caller reads totals4 and allocates the outputs
 *   3. desco_partition_dev_fill  -> neigh_index[B,2], indicator[V], count_ptr[B+1], count_orig[N_c],
 *      vrowptr[4(N_c+B)+1], vcol[E]  -- bit-identical to desco_partition_export (quirk_batch = 0).
 * ------------------------------------------------------------------------------------------ */
int desco_partition_dev_count(const int64_t* graph_ptr, const int32_t* node_graph,
                              const int32_t* rowptr, const int32_t* col, int64_t num_nodes, int depth,
                              int n_max, int num_waves, int32_t* nsize, int32_t* ecnt_count,
                              int32_t* ecnt_canon, desco_stream_t stream);
int desco_partition_dev_scan(const int32_t* nsize, const int32_t* ecnt_count,
                             const int32_t* ecnt_canon, int64_t num_nodes, int64_t* b_index,
                             int64_t* row_off, int64_t* eoff_count, int64_t* eoff_canon,
                             int64_t* totals4, desco_stream_t stream);
int desco_partition_dev_fill(const int64_t* graph_ptr, const int32_t* node_graph,
                             const int32_t* rowptr, const int32_t* col, int64_t num_nodes, int depth,
                             int n_max, int num_waves, const int64_t* b_index, const int64_t* row_off,
                             const int64_t* eoff_count, const int64_t* eoff_canon, int64_t num_neigh,
                             int64_t num_count, int64_t edges_count, int64_t edges_canon,
                             int64_t* neigh_index, uint8_t* indicator, int32_t* count_ptr,
                             int32_t* count_orig, int32_t* vrowptr, int32_t* vcol,
                             desco_stream_t stream);

/* HOST: exact canonical (induced, symmetry-normalised) counts of connected query graphs with
 * 2..6 nodes for every node of every graph: out[v][q] = #{S : max(S) = v, G[S] isomorphic to q}.
 * Replaces the VF2 ground truth (workload.py:327-348 MatchSubgraphWorker divided by
 * data.py:61-88 SymmetricFactor; driver workload.py:551-726).  Query q has q_nodes[q] nodes and the
 * undirected edges q_edges[2*e], q_edges[2*e+1] for e in [q_edge_ptr[q], q_edge_ptr[q+1]).
 * out: int64 [num_nodes][num_queries]. */
int desco_canonical_counts(const int64_t* graph_ptr, int64_t num_graphs, const int64_t* rowptr,
                           const int32_t* col, const int32_t* q_nodes, const int32_t* q_edge_ptr,
                           const int32_t* q_edges, int num_queries, int num_threads,
                           int64_t* out);

/* The same counts on the GPU (csrc/groundtruth_dev.hip): queries of 2..5 nodes, at most 32, pairwise
 * non-isomorphic (the host entry above takes the general case).
 * HOST helper: table[1098] = for k = 2..5 nodes and every adjacency mask on k nodes (bit
 * b(b-1)/2 + a for the pair a < b; the masks of k nodes start at offsets 0, 2, 10, 74) the index of
 * the query of that isomorphism class, or -1; *kmax = the largest query. */
int desco_canonical_class_table(const int32_t* q_nodes, const int32_t* q_edge_ptr,
                                const int32_t* q_edges, int num_queries, int16_t* table, int* kmax);
/* DEVICE: graph_ptr [G+1], rowptr [N+1] (int64), col [num_entries] (global node ids, rows ascending),
 * node_graph [N] (graph of every node), bit_off [G] (first word of graph g's adjacency bitset: rows of
 * ceil(n_g/64) words), bits [num_words] workspace, cls = the table above; all device pointers.
 * out: int64 [N][num_queries], zeroed and filled by the call.  Enqueues on `stream`, does not
 * allocate or synchronise. */
int desco_canonical_counts_dev(const int64_t* graph_ptr, int64_t num_graphs, int64_t num_nodes,
                               const int64_t* rowptr, int64_t num_entries, const int32_t* col,
                               const int32_t* node_graph, const int64_t* bit_off, uint64_t* bits,
                               int64_t num_words, const int16_t* cls, int kmax, int num_queries,
                               int64_t* out, desco_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DEVICE kernels
 * ------------------------------------------------------------------------------------------ */

/* K1/K16  pre_mp = nn.Linear(input_dim, H) (gnn_model.py:131, 231):
 * out[i, 0:n] = feat[i, 0:k] * wt[k][n] + bias   (wt = weight transposed, [k][n] row major) */
int desco_linear_smallk_f32(const float* feat, int64_t ldf, int k, const float* wt,
                            const float* bias, float* out, int64_t ldo, int64_t m, int n,
                            desco_stream_t stream);

/* K2/K3  SAGEConv message+aggregate for all relation slots at once (gnn_model.py:392-394,
 * 402-404: index_select + scatter_add):  out[v, 0:64] = sum_{e in vrow v} x[vcol[e], 0:64].
 * A 16-lane group (float4 per lane) per destination row, eight unconditional source loads in
 * flight per lane (absent sources read a zero row); S in {1,2,4}; vcol must be readable at
 * index 0 even when there is no edge.
 * out is [num_rows*S, 64] contiguous (== [num_rows, S*64]). */
int desco_csr_gather_sum_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                             const int32_t* vcol, int64_t num_rows, int slots, float* out,
                             desco_stream_t stream);

/* K4-K6, K8, K10, K12, K18, K20, K21: every dense projection of the path.
 * C[m, n] = act( [A1 | A2][m, :] * Wt + bias[(m % bias_rows), n] + sum_j S[m, j] * Ws[j, n] )
 *   A1: [M, k1] leading dim lda1;  A2: [M, k2] leading dim lda2 (k2 may be 0);  k1, k2 % 32 == 0
 *   Wt: [(k1+k2), n] row major (= torch weight transposed);  n % 64 == 0
 *   bias: [bias_rows, n] or NULL (bias_rows >= 1);  S: [M, ns] (ns <= 4) or NULL;  Ws: [ns, n]
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain per output). */
int desco_gemm_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2,
                   const float* wt, int n, const float* bias, int bias_rows, const float* s,
                   int ns, const float* ws, int act, float slope, float* c, int64_t ldc,
                   int64_t m, desco_stream_t stream);

/* Counter-based dropout of the training steps (F.dropout after every layer's relu, gnn_model.py:274; nn.Dropout =
 * post_mp.1, gnn_model.py:44-53; --gossip_dropout defaults to 0.01, config.py:316).  No mask is stored: the 32 random
 * bits of element (row, col) are word (row & 3) of
 *   Philox4x32-10(counter = {row >> 2, col | site << 24, lo32(step), hi32(step)}, key = {lo32(seed), hi32(seed)})
 * with (seed, step) = key[0], key[1] read from DEVICE memory by the kernel (desco_rng_next writes them), so forward and
 * backward of one step see the same mask and a captured step draws a fresh mask on every replay.  The element is dropped
 * iff bits < threshold; kept elements are multiplied by scale.  key == NULL: no dropout.
 * Limits: row < 2^34, col < 2^24, site < 256. */
typedef struct desco_dropout {
  const uint64_t* key;  /* device, 2 words */
  uint32_t site;        /* which dropout call of the step (independent streams) */
  uint32_t threshold;   /* round(p * 2^32), p < 1 */
  float scale;          /* 1 / (1 - p)   (0 with threshold 0xffffffff stands for p >= 1) */
} desco_dropout;

/* Up to four INDEPENDENT products of desco_gemm_f32's form in one launch (the count-row and canonical-row halves of a
 * training layer, lightning_model.py:228-254 through gnn_model.py:253-277: same step, disjoint rows, different
 * weights).  Field meaning as the arguments of desco_gemm_f32; descriptors with m == 0 are skipped. */
typedef struct desco_gemm_desc {
  const float* a1; int64_t lda1; int k1;
  const float* a2; int64_t lda2; int k2;
  const float* wt; int n;
  const float* bias; int bias_rows;
  const float* s; int ns; const float* ws;
  int act; float slope;
  float* c; int64_t ldc; int64_t m;
  /* backward use: c = v * act'(gate[row, col]) for gate = the saved activation OUTPUT (gate_act / gate_slope: its
   * activation; NULL: none) -- desco_act_grad_f32 fused into the epilogue; accum != 0: c += v instead of c = v */
  const float* gate; int64_t ldg; int gate_act; float gate_slope; int accum;
  /* drop.key != NULL: the stored value is additionally multiplied by the dropout factor of (row, col) -- forward:
   * dropout(act(..)) (relu / leaky commute with a non-negative factor, so Linear -> Dropout -> LeakyReLU of post_mp is
   * this too); backward with a gate: dC * factor * act'(gate), the factor regenerated, not read */
  desco_dropout drop;
} desco_gemm_desc;
int desco_gemm_f32_multi(int num, const desco_gemm_desc* descs, desco_stream_t stream);

/* Same contract as desco_gemm_f32 but computed on the bf16 matrix pipe with fp32-level accuracy
 * ("bf16x6": each fp32 operand is split into three bf16 terms, the six products of weight >= 2^-16
 * are accumulated in fp32; error ~2^-23 per product).  The weight is passed N-MAJOR and already
 * split: w_planes[3][n][k1+k2] = (hi, mid, lo) bf16 bit patterns of torch's native [out, in] weight,
 * produced once per weight version by desco_split_bf16x3_f32 (planes[3][count] of w[count]). */
int desco_gemm_bf16x6_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2,
                          int k2, const int16_t* w_planes, int n, const float* bias, int bias_rows,
                          const float* s, int ns, const float* ws, int act, float slope, float* c,
                          int64_t ldc, int64_t m, desco_stream_t stream);
int desco_split_bf16x3_f32(const float* w, int64_t count, int16_t* planes, desco_stream_t stream);
/* One descriptor of desco_gemm_f32_multi's form on the bf16x6 pipe: d->wt is ignored, the weight is w_planes[3][n][k1+k2]
 * (n-major, as for desco_gemm_bf16x6_f32); bias / act / gate / drop as documented for desco_gemm_desc, accum must be 0.
 * ws_rows > 1: the scalar tail's matrix is per row class -- v += sum_j s[row, j] * ws[row % ws_rows][j][col] -- which is
 * desco_affine_rows_f32 (the gossip layers' per-query tables, ws_rows = the query count) fused into the epilogue.
 * For the training step's products that stream a million rows through a 64- to 256-wide weight (the gossip step's
 * forward and input-gradient products): fp32-accurate at the HBM rate instead of the fp32 matrix pipe's. */
int desco_gemm_bf16x6_desc_f32(const desco_gemm_desc* d, const int16_t* w_planes, int ws_rows, desco_stream_t stream);
/* planes[3][n][k] of the TRANSPOSE of w [k, n] (row stride ldw >= n): the n-major planes of a weight kept as [in, out], or
 * -- for an input-gradient product dA = dZ W -- of torch's [out, in] weight read as [k = out, n = in]. */
int desco_split_bf16x3_t_f32(const float* w, int k, int n, int64_t ldw, int16_t* planes, desco_stream_t stream);
/* Up to four INDEPENDENT descriptors on the bf16x6 pipe in one launch (planes[i] = the n-major planes of descriptor i;
 * 128 x 64 tiles for every problem): the count-row and canonical-row halves of a training layer's forward and
 * input-gradient products.  As desco_gemm_bf16x6_desc_f32 without the scalar tail (ns must be 0). */
int desco_gemm_bf16x6_multi_f32(int num, const desco_gemm_desc* descs, const int16_t* const* planes,
                                desco_stream_t stream);
/* The same launch with ONE plane per weight (round-to-nearest bf16) and A rounded in the kernel: desco_gemm_bf16_f32's
 * arithmetic (the bf16 training mode, BASELINE config 3), several problems per launch. */
int desco_gemm_bf16_multi_f32(int num, const desco_gemm_desc* descs, const int16_t* const* planes,
                              desco_stream_t stream);
/* planes[num][num_planes][..] of num contiguous matrices w[num][rows][cols]: as they are ([rows][cols]) or, transpose != 0,
 * of their transposes ([cols][rows]) -- a training trunk's stacked weights, all layers in one launch.  num_planes = 3: the
 * truncation split of desco_split_bf16x3_f32; 1: round-to-nearest bf16 (desco_round_bf16_f32). */
int desco_split_bf16x3_batch_f32(const float* w, int64_t num, int rows, int cols, int transpose, int num_planes,
                                 int16_t* planes, desco_stream_t stream);

/* Same contract as desco_gemm_f32 on the fp16 matrix pipe with fp32-level accuracy in THREE products ("f16x3",
 * csrc/gemm_f16x3.hip): operands are scaled by powers of two and split into two fp16 terms (hi = rne(s x),
 * lo = rne(s x - hi): 22 significand bits); hi*hi + hi*lo + lo*hi accumulate in fp32 and the scales are undone in
 * the epilogue (exact).  Measured error <= the f32 MFMA's (tools/micro/f16x3_probe.hip).
 *   w_planes[2][n][k1+k2]: (hi, lo) fp16 bit patterns of scale * W, N-MAJOR, and w_scale[2] = {scale, 1/scale} on
 *     the DEVICE, both produced once per weight version by desco_split_f16x2_f32 (one power of two per matrix:
 *     largest |w| -> [2^14, 2^15));
 *   row_scale[m]: a bound of every row's largest |a| over [A1 | A2] (the kernel derives the power of two that puts it
 *     into [2^14, 2^15)): desco_row_absmax_f32 on the same operands, or left by the kernel that wrote A
 *     (desco_degree_affine_f32 / desco_shmp_layer_f16x3_f32 with row_absmax). */
int desco_gemm_f16x3_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2,
                         const int16_t* w_planes, const float* w_scale, int n, const float* bias,
                         int bias_rows, const float* s, int ns, const float* ws, int act, float slope,
                         float* c, int64_t ldc, int64_t m, const float* row_scale, desco_stream_t stream);
int desco_row_absmax_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2,
                        int64_t m, float* row_scale, desco_stream_t stream);
int desco_split_f16x2_f32(const float* w, int64_t count, int16_t* planes, float* scale, desco_stream_t stream);

/* bf16 training mode (BASELINE config 3): the same GEMM contract with ONE bf16 product per
 * multiply-add -- A is rounded to nearest-even bf16 inside the kernel, the N-MAJOR weight arrives
 * rounded by desco_round_bf16_f32 (out[count] bf16 bit patterns); fp32 accumulation and output. */
int desco_gemm_bf16_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2,
                        const int16_t* w_bf16, int n, const float* bias, int bias_rows,
                        const float* s, int ns, const float* ws, int act, float slope, float* c,
                        int64_t ldc, int64_t m, desco_stream_t stream);
int desco_round_bf16_f32(const float* w, int64_t count, int16_t* out, desco_stream_t stream);

/* Fused SHMP layer (K2-K7 in one launch; csrc/shmp_layer.hip).  For destination rows i in
 * [row0, row0+num_rows), with sm = slots_mfma <= 3, st = slots_table <= 2, S = slots_stored (sm+st <= S <= 4):
 *   out[i] = relu( sum_{s<sm} (sum_{e in vrow(i*S+s)} x[vcol[e]]) * Wt_s + x[i] * Wt_sm + bias
 *                  + sum_{sm<=s<sm+st} sum_{e in vrow(i*S+s)} ytab[vcol[e]-ytab_row0][(s-sm)*64 : +64] )
 * Wt: [(sm+1)*64, 64] row major.  "Table slots" are relations whose sources were already multiplied
 * by their weight block (ytab = x_src * [Wt_sm' | ...], ld ldy): (sum_j x_j) W = sum_j (x_j W).
 * x / out are indexed by the same global row ids as vrowptr; out must not alias x.
 * out2 (optional, may be NULL): a second copy of the produced rows, row i stored at
 * out2[(i - row0)*ldo2 ...] -- the canonical launches fill their column block of the anchor-MLP
 * operand [B, 64*(layers+1)] this way instead of a separate concatenation pass.
 * Replaces SAGEConv.propagate + lin + to_hetero sum + updates + relu
 * (gnn_model.py:262-264, 273, 392-395) without materialising the aggregates. */
int desco_shmp_layer_f32(const float* x, int64_t ldx, const int32_t* vrowptr, const int32_t* vcol,
                         int64_t row0, int64_t num_rows, int slots_stored, int slots_mfma,
                         int slots_table, const float* wt, const float* bias, const float* ytab,
                         int64_t ldy, int64_t ytab_row0, float* out, int64_t ldo, float* out2,
                         int64_t ldo2, desco_stream_t stream);
/* Same layer with the matrix work on the bf16 pipe at fp32 accuracy (the 6-product split of
 * desco_gemm_bf16x6_f32): wt_planes[3][64 n][(slots_mfma+1)*64 k] = desco_split_bf16x3_f32 of the
 * N-MAJOR folded weight (the transpose of desco_shmp_layer_f32's wt).  slots_mfma <= 2. */
int desco_shmp_layer_bf16x6_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                const int32_t* vcol, int64_t row0, int64_t num_rows,
                                int slots_stored, int slots_mfma, int slots_table,
                                const int16_t* wt_planes, const float* bias, const float* ytab,
                                int64_t ldy, int64_t ytab_row0, float* out, int64_t ldo, float* out2,
                                int64_t ldo2, desco_stream_t stream);

/* desco_shmp_layer_bf16x6_f32 with the layer's global_add_pool (gnn_model.py:88-89, 107) fused into
 * the epilogue: besides (or instead of: out may be NULL, e.g. for the last layer, whose count rows
 * feed nothing but the pooling) storing the produced rows, every wave sums them per segment and
 * writes ONE partial row per (wave tile, segment) to pool_part[slot][0:64].  A wave tile is
 * TR = desco_shmp_pool_tile_rows() rows (16; 32 with DESCO_SHMP_ROWS=32 in the environment, the 32-row
 * form of the kernel kept for A/B runs):
 *   pool_bits[t] bit r = 1  <=>  row TR t + r is the LAST row of its segment (segments = contiguous row
 *                                ranges covering [row0, row0+num_rows); row0 % TR == 0),
 *   pool_slot[t]            =    first slot of tile t (tiles use consecutive slots: one per segment
 *                                that has a row in the tile),
 * both [ceil((row0+num_rows)/TR)], indexed by absolute row / TR.  desco_pool_reduce_f32 (tile_rows =
 * the same TR) then adds the partials of every segment in tile order (+ extra[b], like
 * desco_segment_sum_f32) -- together they replace one desco_segment_sum_f32 pass over the rows (a
 * full read of x) by ~(1/TR + 1/segment length) of its traffic, deterministically (no float atomics). */
int desco_shmp_pool_tile_rows(void);
int desco_shmp_layer_pool_bf16x6_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                     const int32_t* vcol, int64_t row0, int64_t num_rows,
                                     int slots_stored, int slots_mfma, int slots_table,
                                     const int16_t* wt_planes, const float* bias, const float* ytab,
                                     int64_t ldy, int64_t ytab_row0, float* out, int64_t ldo,
                                     const uint32_t* pool_bits, const int32_t* pool_slot,
                                     float* pool_part, desco_stream_t stream);
/* The same fused layer (and its pooling variant) in the THREE-product fp16 form (csrc/shmp_layer16.hip, 16-row wave
 * tiles): wt_planes[2][64 n][(slots_mfma+1)*64 k] fp16 (hi, lo) and w_scale[2] = {scale, 1/scale} on the device, both
 * from desco_split_f16x2_f32 of the transposed folded weight; slots_mfma <= 2.  The activation side is scaled inside
 * the kernel: one power of two per (row, K block) -- the largest |value| of the row's 64 gathered sums -> [2^14,
 * 2^15) --, the accumulators follow by exact multiplications and leave the scales in the epilogue. */
int desco_shmp_layer_f16x3_f32(const float* x, int64_t ldx, const int32_t* vrowptr, const int32_t* vcol,
                               int64_t row0, int64_t num_rows, int slots_stored, int slots_mfma,
                               int slots_table, const int16_t* wt_planes, const float* w_scale,
                               const float* bias, const float* ytab, int64_t ldy, int64_t ytab_row0,
                               float* out, int64_t ldo, float* out2, int64_t ldo2, float* row_absmax,
                               const float* xself, int64_t ldxs, desco_stream_t stream);
/* xself (optional): the launch's OWN rows -- the self block's operand -- are read at xself + i * ldxs (i = the global row
 * index that addresses x) instead of from x; out may then be NULL when out2 is given (the rows are stored once, in out2's
 * tensor).  The canonical launches use both: their rows live only in the anchor operand's column blocks.
 * row_absmax (optional, [num_rows]): row_absmax[i - row0] = max(row_absmax[i - row0], max_c |out[i, c]|) is ACCUMULATED
 * over the launches that fill the column blocks of one operand (out2): its per-row bound for desco_gemm_f16x3_f32. */
int desco_shmp_layer_pool_f16x3_f32(const float* x, int64_t ldx, const int32_t* vrowptr, const int32_t* vcol,
                                    int64_t row0, int64_t num_rows, int slots_stored, int slots_mfma,
                                    int slots_table, const int16_t* wt_planes, const float* w_scale,
                                    const float* bias, const float* ytab, int64_t ldy, int64_t ytab_row0,
                                    float* out, int64_t ldo, const uint32_t* pool_bits,
                                    const int32_t* pool_slot, float* pool_part, desco_stream_t stream);

/* desco_shmp_layer_pool_f16x3_f32 for a layer input that exists as a TABLE of its distinct rows only.  The closed-form first
 * layer's output is a function of a row's S slot degrees, so X_1 has a few thousand distinct rows: x = that table (L2-resident),
 * vcol = the column ids with the sources of the MFMA slots replaced by their table rows, and the launch's OWN rows (the self
 * block's operand) are RECOMPUTED from their slot degrees with the first layer's coefficients self_coef [slots_stored + 1][64]:
 *   own row i = relu(self_coef[S] + sum_s degree_s(i) * self_coef[s])          (desco_degree_affine_f32's arithmetic)
 * -- X_1 [N, 64] is never written or read (host side: NeighborhoodBatch.degree_table_index, gnn_model.FIRST_LAYER_TABLE).
 * Results are bit-identical to the launch on the materialised tensor. */
int desco_shmp_layer_pool_table_f16x3_f32(const float* x, int64_t ldx, const int32_t* vrowptr, const int32_t* vcol,
                                          int64_t row0, int64_t num_rows, int slots_stored, int slots_mfma,
                                          int slots_table, const int16_t* wt_planes, const float* w_scale,
                                          const float* bias, const float* ytab, int64_t ldy, int64_t ytab_row0,
                                          float* out, int64_t ldo, const uint32_t* pool_bits,
                                          const int32_t* pool_slot, float* pool_part, const float* self_coef,
                                          desco_stream_t stream);

int desco_pool_reduce_f32(const float* pool_part, const uint32_t* pool_bits, const int32_t* pool_slot,
                          const int32_t* seg_ptr, int64_t num_seg, const float* extra,
                          int64_t ld_extra, float* out, int64_t ldo, int tile_rows,
                          desco_stream_t stream);
/* The same reduce for 1..8 layers' partial arrays in one launch (they share the tile index and the segments): HOST arrays
 * of DEVICE pointers pool_parts[num], extras[num] (or NULL; entries may be NULL), outs[num]; common ld_extra / ldo.
 * Bit-identical to num calls of desco_pool_reduce_f32. */
int desco_pool_reduce_multi_f32(int num, const float* const* pool_parts, const uint32_t* pool_bits,
                                const int32_t* pool_slot, const int32_t* seg_ptr, int64_t num_seg,
                                const float* const* extras, int64_t ld_extra, float* const* outs, int64_t ldo,
                                int tile_rows, desco_stream_t stream);

/* post_mp.0 (gnn_model.py:44-53, first Linear) on the pooled embeddings WITHOUT materialising them: row b of the operand is
 *   [ anch[b, 0:64] + rows(b) * x0 | anch[b, 64 l : 64 (l + 1)] + sum of segment b's partial rows of layer l, l = 1..L ]
 * (what desco_pool_reduce_f32 would have written, same summation order), formed in the product's load phase:
 *   c[b, 0:64] = act( operand[b, :] * W^T + bias ),  W as desco_split_bf16x3_f32 planes [3][64][64 (L + 1)].
 * anch: [m, >= 64 (L + 1)]; parts: HOST array of the L layers' DEVICE partial arrays; seg_ptr / pool_bits / pool_slot: the
 * fused pooling's index for 16-row tiles.  EVERY SEGMENT MUST LIE IN AT MOST THREE TILES (<= 33 rows; the caller checks:
 * further tiles are ignored).  Saves the write and the read of the pooled [m, 64 (L + 1)] tensor. */
int desco_pool_post_bf16x6_f32(const float* anch, int64_t lda, int num_layers, const int16_t* w_planes, int n,
                               const float* bias, int act, float slope, float* c, int64_t ldc, int64_t m,
                               const int32_t* seg_ptr, const uint32_t* pool_bits, const int32_t* pool_slot,
                               const float* const* parts, const float* x0, int tile_rows, desco_stream_t stream);

/* post_mp.3 -> ReLU -> post_mp.5 -> ReLU -> post_mp.7 (gnn_model.py:44-53: Linear(64, 64), Linear(64, 256), Linear(256, 64))
 * in one launch:  out[i, 0:64] = W3 relu(W2 relu(W1 x[i, 0:64] + b1) + b2) + b3.  A row is read once and written once;
 * the [m, 64] and [m, 256] intermediates stay in registers (f16x3 arithmetic, fp32-accurate: the products are formed
 * transposed so that a layer's accumulators ARE the next layer's operand registers; per-row power-of-two scales).
 * wK_planes / wK_scale = desco_split_f16x2_f32 of the [out, in] weights ([2][64][64], [2][256][64], [2][64][256]);
 * biases [64], [256], [64] or NULL.  x == out is allowed (a wave reads its 32 rows before it writes them). */
int desco_post_mp_tail_f16x3_f32(const float* x, int64_t ldx, int64_t m, const int16_t* w1_planes,
                                 const float* w1_scale, const float* b1, const int16_t* w2_planes,
                                 const float* w2_scale, const float* b2, const int16_t* w3_planes,
                                 const float* w3_scale, const float* b3, float* out, int64_t ldo,
                                 desco_stream_t stream);

/* Row-wise Linear with 64 inputs on the fused layer's streaming machinery (bf16x6 arithmetic,
 * fp32-accurate): out[i, 0:64*num_blocks] = act(x[i, 0:64] * W^T + bias[0:64*num_blocks]);
 * w_planes[num_blocks][3][64 n][64 k] = desco_split_bf16x3_f32 of every 64-row block of the
 * [out, in] weight.  Memory-shaped projections (canonical table, post_mp.5, count_model target
 * half): x is read once per two output blocks, rows stream at HBM rate.  x == out is not allowed. */
int desco_linear64_bf16x6_f32(const float* x, int64_t ldx, const int16_t* w_planes, int num_blocks,
                              const float* bias, int act, float slope, float* out, int64_t ldo,
                              int64_t num_rows, desco_stream_t stream);

/* First SHMP layer (and first pooling block) when every node of a type has the same input row --
 * the default pipeline's all-zero node features (workload.py:431-440, transforms.py:380-384) make
 * pre_mp's output its bias.  Then agg_s[i] = deg_s(i) * x0_src(s) and, for rows [row0, row0+num_rows):
 *   out[i, 0:64] = act( coef[slots] + sum_{s<slots} (vrowptr[i*slots+s+1] - vrowptr[i*slots+s]) * coef[s] )
 *                  + extra[i - row0]                       (coef: [slots+1, 64]; extra optional)
 * mathematically identical to desco_shmp_layer_f32 on the constant input; no gather, no GEMM. */
int desco_degree_affine_f32(const int32_t* vrowptr, int64_t row0, int64_t num_rows, int slots,
                            const float* coef, int act, float slope, const float* extra,
                            int64_t ld_extra, float* out, int64_t ldo, float* row_absmax,
                            desco_stream_t stream);
/* The same rows for a launch that starts at row 0 (the count rows), with their global_add_pool (gnn_model.py:88-89, 107)
 * fused in as in desco_shmp_layer_pool_*: besides storing the rows (out may be NULL) every 16-row tile leaves one partial
 * row per segment that has a row in it at pool_part[pool_slot[t] + k] (pool_bits / pool_slot: the index of
 * desco_shmp_layer_pool_bf16x6_f32 for 16-row tiles); desco_pool_reduce(_multi)_f32 finishes the sums.  Saves the one
 * read of the produced rows that the segment sum of this layer cost. */
int desco_degree_affine_pool_f32(const int32_t* vrowptr, int64_t num_rows, int slots, const float* coef, int act,
                                 float slope, float* out, int64_t ldo, const uint32_t* pool_bits,
                                 const int32_t* pool_slot, float* pool_part, desco_stream_t stream);
/* row_absmax (optional, [num_rows]): row_absmax[i - row0] = max_c |out[i, c]| is WRITTEN -- the start of the per-row
 * bound desco_gemm_f16x3_f32 takes when these rows are the first column block of its operand. */

/* Indices of the backward pass, built on the device (replaces the per-batch host transposes):
 *  desco_vcsr_transpose_sym: the transposed index of a SYMMETRIC virtual-row CSR (every edge
 *    dst<-src has its mirror src<-dst; true for canonical-partition blocks, query graphs and the
 *    gossip CSR): t_rowptr[num_rows+1] (= vrowptr[slots*j]) and t_col[E] = the virtual rows
 *    (k*slots + mirror slot) that read row j, ascending.  num_count: rows >= num_count are
 *    canonical rows (only used for slots == 4; pass num_rows otherwise).
 *  desco_segment_ids: seg_id[r] = b for seg_ptr[b] <= r < seg_ptr[b+1]. */
int desco_vcsr_transpose_sym(const int32_t* vrowptr, const int32_t* vcol, int64_t num_rows, int slots,
                             int64_t num_count, int32_t* t_rowptr, int32_t* t_col,
                             desco_stream_t stream);
int desco_segment_ids(const int32_t* seg_ptr, int64_t num_seg, int32_t* seg_id, desco_stream_t stream);

/* K9  global_add_pool (gnn_model.py:107) over contiguous row segments, plus one optional extra row
 * per segment (the anchored canonical embedding, gnn_model.py:69-73, 88-89):
 * out[b, 0:ncols] = sum_{r in [seg_ptr[b], seg_ptr[b+1])} x[r, 0:ncols] + extra[b, 0:ncols]
 * Also used for the node->graph aggregation of counts (workload.py:136-148, 303-324). */
int desco_segment_sum_f32(const float* x, int64_t ldx, int ncols, const int32_t* seg_ptr,
                          int64_t num_seg, const float* extra, int64_t ld_extra, float* out,
                          int64_t ldo, desco_stream_t stream);

/* The same for num_layers feature blocks in one launch (training trunk: X_1 .. X_L live in one buffer):
 * out[b, 64 l .. 64 l + 63] = sum_{r in segment b} x[l * layer_stride + r * ldx + 0..63] + extra[b, 64 l ..] */
int desco_segment_sum_layers_f32(const float* x, int64_t ldx, int64_t layer_stride, int num_layers,
                                 const int32_t* seg_ptr, int64_t num_seg, const float* extra,
                                 int64_t ld_extra, float* out, int64_t ldo, desco_stream_t stream);

/* K12/K13  count head (lightning_model.py:176-193, 210-221) in separable form:
 * logit[b,q] = sum_c w2[c] * leaky(T[b,c] + Qh[q,c]) + b2;  out = exp2 ? 2^logit - 1 : logit
 * T: [B, hid] (target half of count_model.0), Qh: [Q, hid] (query half + bias);
 * hid % 64 == 0, hid <= 256, Q <= 32 */
int desco_count_head_f32(const float* t, int64_t ldt, const float* qh, int64_t ldq, int hid,
                         const float* w2, float b2, const float* b2_dev, float slope,
                         int exp2_minus_1, float* out, int64_t ldo, int64_t num_b, int num_q,
                         desco_stream_t stream);   /* b2_dev != NULL: the bias is read from the device */

/* K14  GossipDataset.apply_neighborhood_count (workload.py:107-112): dst[rows[b], :] = src[b, :] */
int desco_scatter_rows_f32(const float* src, int64_t lds, const int32_t* rows, int64_t num_src,
                           int ncols, float* dst, int64_t ldd, desco_stream_t stream);

/* K15-K20, layer 0 of the gossip GNN for ALL queries at once, in closed form:
 * per node i and query q:
 *   lo/hi = neighbours j<i / j>i (== edge_weight of gnn_model.py:248),
 *   a0 = g0[q]*deg_lo + (1-g0[q])*deg_hi,  b0 = g0[q]*sum_lo x[j,q] + (1-g0[q])*sum_hi x[j,q]
 *   h1[i,q,:] = relu(a0*p[q,:] + b0*r[:] + x[i,q]*t[:] + z[q,:])
 *   scal[i,q,0] = g1[q]*deg_lo + (1-g1[q])*deg_hi,  scal[i,q,1] = x[i,q]
 * rowptr/col: symmetric CSR of the batch, col ascending per row. */
int desco_gossip_layer0_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                            int64_t num_nodes, int num_q, const float* g0, const float* g1,
                            const float* p, const float* r, const float* t, const float* z,
                            float* h1, float* scal, desco_stream_t stream);

/* K18/K19 for layers >= 1, aggregate-then-transform form:
 * out[i,q,:] = sum_{j~i} (j<i ? g[q] : 1-g[q]) * h[j,q,:]      (h, out: [num_nodes, num_q, 64])
 * g == NULL selects the signed form sum_{j<i} h[j] - sum_{j>i} h[j] (= d out / d g, training). */
int desco_gossip_gather_f32(const float* h, const int32_t* rowptr, const int32_t* col,
                            int64_t num_nodes, int num_q, const float* g, float* out,
                            desco_stream_t stream);

/* Fused gossip stage (csrc/gossip_fused.hip), two launches for all queries:
 *  (1) desco_gossip_scalars_f32: scal4[i*Q+q] = (a0, b0, a1, x[i,q]) with
 *      a_l = g_l[q]*deg_lo(i) + (1-g_l[q])*deg_hi(i),  b0 = g0[q]*sum_{j<i} x[j,q] + (1-g0[q])*sum_{j>i} x[j,q]
 *  (2) desco_gossip_fused_f32: per tile of 128 nodes and one query, entirely on chip,
 *      h1 = relu(a0*p_q + b0*r + x*t + z_q);  hh = sum_j (j<i ? g1 : 1-g1)*h1_j;
 *      h2 = relu([hh|h1] w1 + a1*u + d1);  y1 = leaky_0.1([h1|h2] wp + x*tp + zp_q);
 *      y2 = relu(y1 w3 + b3);  out[i,q] = x + b7 + sum_c relu(y2 w5 + b5)[c]*w7[c]
 *      The four weight matrices are passed N-MAJOR (torch's [out, in] layout of the folded
 *      matrices: w1, wp [64,128]; w3 [64,64]; w5 [256,64]) and pre-split into bf16 planes by
 *      desco_split_bf16x3_f32: w*_planes[3][out][in].  All GEMMs run as the fp32-accurate
 *      6-product bf16 split of desco_gemm_bf16x6_f32.  p, z, r, t must be 8-byte aligned.
 * Replaces BaseGNN.forward (gossip) for every query: gnn_model.py:58-103, 230-260, 303-350 and the
 * loop of lightning_model.py:613-628; equals layer0 + gather + 4 GEMMs + rowdot of the unfused path.
 * desco_gossip_fused_f32 (the six-product bf16 form; since round 4 the cross-check of desco_gossip_fused_f16x3_f32)
 * draws its work items from a process-wide ring of 64 ticket slots taken in launch order: launches must be issued
 * from ONE stream at a time and captured launches must not be replayed concurrently with other launches of it (the
 * fp16 entry point takes a caller-owned queue instead). */
int desco_gossip_scalars_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                             int64_t num_nodes, int num_q, const float* g0, const float* g1,
                             float* scal4, desco_stream_t stream);
int desco_gossip_fused_f32(const float* scal4, const int32_t* rowptr, const int32_t* col,
                           int64_t num_nodes, int num_q, const float* g1, const float* p,
                           const float* z, const float* zp, const float* r, const float* t,
                           const float* u, const float* tp, const float* d1,
                           const int16_t* w1_planes, const int16_t* wp_planes,
                           const int16_t* w3_planes, const float* b3, const int16_t* w5_planes,
                           const float* b5, const float* w7, float b7, float* out,
                           const uint8_t* tile_perm, desco_stream_t stream);
/* The same fused gossip pass in the THREE-product fp16 form (csrc/gossip_f16.hip; the product path since round 4):
 * per-node power-of-two scales carry the range; a wave owns 16 nodes x all features, all nine weight blocks stay
 * resident in LDS and a wave carries its nodes through the whole network for 5 queries per work unit with no barrier
 * in the loop.  tile_perm (optional, desco_gossip_tile_order's output): the 16 nodes of a wave are the degree-sorted
 * ranks 16 i .. 16 i + 15 of their 128-node tile (speed only; p, z, r, t must be 16-byte aligned).
 *   wstream [9][2][4096] fp16 + winv[4]: desco_gossip_f16_stream of the four desco_split_f16x2_f32 plane sets
 *     (w1, wp [2][64][128]; w3 [2][64][64]; w5 [2][256][64]) and the 1/scale of each matrix, in that order;
 *   queue: two zeroed 64-bit words owned by the caller (work-item tickets of this launch; the kernel leaves them
 *     zero).  Launches that may run CONCURRENTLY (other streams, graph replays) need queues of their own; launches
 *     on one stream can share one.
 * Work per (node, query) row: 73 728 MFMA flop (x3 products); bytes the algorithm has to move per row: its 16-byte
 * record + 4-byte result + 16 bytes per NEIGHBOUR record (deg x 16: 12 of them used) + the column ids once per node --
 * rows * 20 + 4 * (edges * (1 + 4 Q) + nodes), which is what bench.py prices `achieved` bytes with (round 4's
 * "16-B record + 4-B result" left the neighbour records out; they are 2/3 of the bytes on COX2 shapes).
 * Round 5: packed fp32 selection on, weight fragments through a register ring two pair steps ahead
 * (csrc/gossip_f16.hip); the library is only linked if no packed fp32 instruction in it has OP_SEL on src1 / src2
 * (tools/check_isa.py, profiles/r5_a_gossip_f16_hazard.md). */
int desco_gossip_f16_stream(const int16_t* w1_planes, const int16_t* wp_planes, const int16_t* w3_planes,
                            const int16_t* w5_planes, int16_t* wstream, desco_stream_t stream);
int desco_gossip_fused_f16x3_f32(const float* scal4, const int32_t* rowptr, const int32_t* col,
                                 int64_t num_nodes, int num_q, const float* g1, const float* p,
                                 const float* z, const float* zp, const float* r, const float* t,
                                 const float* u, const float* tp, const float* d1, const int16_t* wstream,
                                 const float* winv, const float* b3, const float* b5, const float* w7, float b7,
                                 float* out, const uint8_t* tile_perm, uint64_t* queue, desco_stream_t stream);
/* tile_perm (optional, 4-byte aligned, [ceil(num_nodes/128)*128] bytes from desco_gossip_tile_order): the order in
 * which the 8 waves of a block walk the rows of a 128-node tile in the neighbour-sum phase -- rows sorted by degree,
 * paired, pairs dealt to the waves in snake order (a half wave per row, the two halves of a wave in lock step, a block
 * barrier at the end: with rows in node order the phase takes 1.3-1.8x its balanced time).  NULL = node order.
 * The result does not depend on it bit for bit (every row's sum keeps its own CSR order). */
int desco_gossip_tile_order(const int32_t* rowptr, int64_t num_nodes, uint8_t* tile_perm, desco_stream_t stream);

/* K21 tail: out[r] = add[r] + sum_c y[r,c]*w[c] + b   (post_mp.7 with output_dim 1, then
 * pred = neigh_pred + gossip_pred, lightning_model.py:622-625) */
int desco_rowdot_add_f32(const float* y, int64_t ldy, int ncols, const float* w, float b,
                         const float* add, float* out, int64_t num_rows, desco_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward pass (training, lightning_model.py:228-254 train_forward + autograd in the reference).
 * Input gradients reuse the forward entry points (dA = dZ * W is desco_gemm_f32 with the
 * un-transposed weight; the transpose of a CSR gather is desco_csr_gather_sum_f32 over the
 * transposed index).  All reductions are deterministic (workspace partials, fixed order).
 * ------------------------------------------------------------------------------------------ */

/* Building blocks of the fused training trunk (desco_amd/autograd.py::ShmpTrunk: the whole
 * BaseGNNCore.forward SAGE loop + anchor + pooling as ONE autograd node, its backward running the chain rule
 * on preallocated buffers -- lightning_model.py:228-254 via loss.backward()):
 *  desco_csr_gather_sum_add_f32: out[j, 0:64] = extra[j, 0:64] + sum_{e in [rowptr[j], rowptr[j+1])} x[col[e], 0:64]
 *    (one slot; out rows ldo apart, extra rows ld_extra apart, extra may be out): the transposed gather of the
 *    backward pass accumulating onto the gradient a row already has from its other consumers;
 *  desco_add_rows_f32: dst[i, 0:ncols] += src[i, 0:ncols]  (ncols % 4 == 0, 16-byte aligned rows). */
int desco_csr_gather_sum_add_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                                 int64_t num_rows, const float* extra, int64_t ld_extra, float* out,
                                 int64_t ldo, desco_stream_t stream);
int desco_add_rows_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t num_rows, int ncols,
                       desco_stream_t stream);
/* The input-row gradient of one SHMP layer of the training trunk in one launch:
 *   out[i, 0:64] = mask_i * ( seed_i + d[i, self_off .. +63] + sum_{v in [t_rowptr[i], t_rowptr[i+1])} dv[t_col[v]] )
 * d = dZ Wt^T [num_rows, ldd] (relation-slot blocks, then the self block at column self_off_count for rows
 * < num_count, self_off_canon for the others), dv = d viewed as rows of 64 floats (t_col indexes them: the
 * transposed index over (row, slot) virtual rows with ldd / 64 blocks per row); seed_i = dpool[seg_id[i]] (the
 * pooling's broadcast gradient, rows ld_pool apart) for i < num_count, dcanon[i - num_count] (NULL: 0) otherwise;
 * mask = (relu_src[i] > 0) * mask_scale elementwise, or 1 when relu_src is NULL (mask_scale: 1, or the factor
 * 1 / (1 - p) of an F.dropout behind the relu, gnn_model.py:273-274: relu_src = dropout(relu(z)) is positive exactly
 * where the element was kept).  Replaces a seed build, two adds, the transposed gather and the activation gradient
 * (five launches) of the per-op wiring. */
int desco_shmp_bwd_dx_f32(const float* d, int64_t ldd, const int32_t* t_rowptr, const int32_t* t_col,
                          int64_t num_rows, int64_t num_count, int self_off_count, int self_off_canon,
                          const float* dpool, int64_t ld_pool, const int32_t* seg_id, const float* dcanon,
                          int64_t ld_canon, const float* relu_src, float mask_scale, float* out,
                          desco_stream_t stream);

/* bytes of workspace desco_gemm_tn_f32 needs for this shape (and the number of M slabs it uses) */
size_t desco_gemm_tn_workspace(int64_t m, int k, int n, int* splits_out);

/* weight gradient: out[k, n] (+)= A[m, k]^T * B[m, n];  k % 64 == 0, n % 64 == 0 */
int desco_gemm_tn_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int k,
                      int n, float* out, int64_t ldo, int accumulate, float* workspace,
                      desco_stream_t stream);

/* weight AND bias gradient of one Linear c = act([a1 | a2] wt + bias) in two launches (partials over M
 * slabs + one fixed-order reduce): dwt[k1+k2, n] = [a1 | a2]^T dz, dbias[n] = sum_m dz[m, n] (dbias may be
 * NULL; a2 may be NULL with k2 = 0).  k1 % 64 == k2 % 64 == n % 64 == 0.  Replaces two desco_gemm_tn_f32
 * and one desco_colsum_f32 call (six launches) of the training step.  m == 0 gives zero gradients (a1 / dz may then be
 * NULL).  workspace:
 * desco_linear_bwd_w_workspace(m, k1 + k2, n) bytes. */
size_t desco_linear_bwd_w_workspace(int64_t m, int k, int n);
int desco_linear_bwd_w_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2,
                           const float* dz, int64_t lddz, int64_t m, int n, float* dwt, int64_t lddw,
                           float* dbias, float* workspace, desco_stream_t stream);

/* Up to 16 independent problems of desco_linear_bwd_w_f32's form (contiguous dwt: lddw = n) in one partial launch and
 * one reduce launch: the (layer, row type) weight gradients of a training step's trunk, which nothing but the optimizer
 * waits for.  m == 0 gives zero gradients (a1 / dz may then be NULL).  workspace: desco_linear_bwd_w_multi_workspace(num, descs) bytes. */
typedef struct desco_bwd_w_desc {
  const float* a1; int64_t lda1; int k1;
  const float* a2; int64_t lda2; int k2;
  const float* dz; int64_t lddz; int64_t m; int n;
  float* dwt; float* dbias;
} desco_bwd_w_desc;
size_t desco_linear_bwd_w_multi_workspace(int num, const desco_bwd_w_desc* descs);
int desco_linear_bwd_w_multi_f32(int num, const desco_bwd_w_desc* descs, float* workspace, desco_stream_t stream);

/* bias gradient: out[n] (+)= sum_m x[m, n];  workspace: 512 * n floats */
int desco_colsum_f32(const float* x, int64_t ldx, int64_t m, int n, float* out, int accumulate,
                     float* workspace, desco_stream_t stream);

/* torch.optim.Adam's step (both models' configure_optimizers, reference lightning_model.py:160-173, 570-583:
 * Adam(lr, weight_decay), default betas / eps, L2 decay, no amsgrad) for a LIST of tensors, one launch per 128 tensors:
 *   g' = g + wd p;  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;
 *   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)              (bias corrections and 1 - beta in double, as torch forms them)
 * params / grads / sizes: HOST arrays of `num` device pointers / element counts; grads[i] == NULL skips tensor i (value,
 * moments and age untouched, as torch skips parameters without gradient).  m, v: device, flat, tensor i at offset
 * sum(sizes[:i]); steps [num] float (per-tensor step count t, advanced by the launch), arrivals [num] uint32 (zero
 * before the first call; scratch), lr [1] float: all device memory, so a captured launch replays without host work. */
int desco_adam_step_f32(int num, float* const* params, const float* const* grads, const int64_t* sizes, float* m,
                        float* v, float* steps, uint32_t* arrivals, const float* lr, double beta1, double beta2,
                        double eps, double weight_decay, desco_stream_t stream);

/* The SHMP trunk of a SMALL single-type batch -- the query graphs of the neighborhood model, which the reference embeds
 * again on every training batch (lightning_model.py:204-207; BaseGNNCore.forward gnn_model.py:253-277 + global_add_pool
 * :88-89, 107) -- in ONE workgroup per direction: rows in LDS, layers separated by barriers.  num_rows <=
 * desco_shmp_trunk_small_max_rows() (144), two relation slots per row (vrowptr [2 n + 1]), exact fp32 FMA chains.
 *   fwd: X_0 = x0; X_{l+1} = relu([agg_0(X_l) | agg_1(X_l) | X_l] wt[l] + bias[l]) -> xall[l] ([L][n][64]);
 *        pooled[b, 64 l : 64 (l + 1)] = sum of X_l over rows seg_ptr[b] .. seg_ptr[b + 1]
 *   bwd: from dpooled [num_seg][ldp]: dwt [L][192][64], dbias [L][64], dx0 [n][64]; wt_t = wt with its last two axes
 *        swapped ([L][64][192]); t_rowptr / t_col: the transposed index over virtual rows 3 k + s
 *        (desco_vcsr_transpose_sym's, re-based to 3 blocks per row), seg_id [n]: the row's segment. */
int desco_shmp_trunk_small_max_rows(void);
int desco_shmp_trunk_small_fwd_f32(const float* x0, const int32_t* vrowptr, const int32_t* vcol, int num_rows,
                                   int num_layers, const float* wt, const float* bias, const int32_t* seg_ptr,
                                   int num_seg, float* xall, float* pooled, int64_t ldp, desco_stream_t stream);
int desco_shmp_trunk_small_bwd_f32(const float* x0, const float* xall, const int32_t* vrowptr, const int32_t* vcol,
                                   const int32_t* t_rowptr, const int32_t* t_col, const int32_t* seg_id, int num_rows,
                                   int num_layers, const float* wt_t, const float* dpooled, int64_t ldp, float* dwt,
                                   float* dbias, float* dx0, desco_stream_t stream);

/* round 6: the same trunk with ONE WORKGROUP PER GRAPH (segment), for batches whose graphs have at most
 * desco_shmp_trunk_graphs_max_rows() (8) rows each -- the 29 query graphs (3..5 nodes; lightning_model.py:204-207): the
 * graphs are independent of each other, so 29 workgroups stream the layers' weights side by side instead of one
 * workgroup walking 135 rows behind one weight stream.  Arguments as the _small_ entry points; rows of a segment beyond
 * the limit are ignored (the caller checks its segment sizes); the backward takes wt itself ([L][192][64], no transposed
 * copy), seg_ptr instead of seg_id, and a workspace of num_layers * num_rows * 64 floats; no limit on num_rows.
 * drop (NULL: none): F.dropout behind every layer's relu (gnn_model.py:274) -- layer l's rows are multiplied by the
 * factor of (row, col) of site drop->site + 2 l; the backward then takes mask_scale = drop->scale (1 without). */
int desco_shmp_trunk_graphs_max_rows(void);
int desco_shmp_trunk_graphs_fwd_f32(const float* x0, const int32_t* vrowptr, const int32_t* vcol, int64_t num_rows,
                                    int num_layers, const float* wt, const float* bias, const int32_t* seg_ptr,
                                    int num_seg, const desco_dropout* drop, float* xall, float* pooled, int64_t ldp,
                                    desco_stream_t stream);
int desco_shmp_trunk_graphs_bwd_f32(const float* x0, const float* xall, const int32_t* vrowptr, const int32_t* vcol,
                                    const int32_t* t_rowptr, const int32_t* t_col, const int32_t* seg_ptr, int num_seg,
                                    int64_t num_rows, int num_layers, const float* wt, const float* dpooled, int64_t ldp,
                                    float mask_scale, float* dwt, float* dbias, float* dx0, float* workspace,
                                    desco_stream_t stream);

/* Backward of desco_linear_smallk_f32 with n = 64 (pre_mp, gnn_model.py:131; feat carries no gradient):
 * dwb[k][0:64] = sum_m feat[m, k] dout[m, :] for k < K, dwb[K][0:64] = sum_m dout[m, :] (the bias gradient), in one
 * pass over dout and one reduce.  workspace: 512 * (k + 1) * 64 floats. */
int desco_linear_smallk_bwd_f32(const float* feat, int64_t ldf, int k, const float* dout, int64_t ldd, int64_t m,
                                float* dwb, float* workspace, desco_stream_t stream);

/* Backward of out[r] = add[r] + y[r, :] . w + b with y = relu(z) (post_mp.7 on post_mp.6's output, gnn_model.py:40-53):
 *   dz[r, c] = dout[r] * w[c] * (y[r, c] > 0),  dwb[c] = sum_r dout[r] * y[r, c] (c < n),  dwb[n] = sum_r dout[r]
 * in one pass over y (n % 4 == 0, n <= 1024, 256 % (n / 4) == 0).  workspace: 1024 * (n + 1) floats. */
int desco_rowdot_bwd_f32(const float* y, int64_t ldy, int n, const float* w, const float* dout, int64_t num_rows,
                         float* dz, int64_t lddz, float* dwb, float* workspace, desco_stream_t stream);

/* dz = dc * act'(c) for c = act(z) (contiguous, count elements) */
int desco_act_grad_f32(const float* dc, const float* c, int act, float slope, float* dz,
                       int64_t count, desco_stream_t stream);

/* The count head straight from the target embeddings (inference): T[b, :] = Wt emb[b, 0:64] (count_model.0's target half,
 * lightning_model.py:127-131) is formed 32 features at a time on the matrix pipe (f16x3, wt_planes / wt_scale =
 * desco_split_f16x2_f32 of the [256, 64] weight) and consumed from registers, so the [m, 256] tensor T that
 * desco_linear64_bf16x6_f32 would write and desco_count_head_f32 read never exists.  Same result as that pair up to
 * rounding.  hid must be 256 and num_q 29 (the standard query set, data.py:37); other shapes use the two-launch form. */
int desco_count_head_emb_f16x3_f32(const float* emb, int64_t lde, int64_t m, const int16_t* wt_planes,
                                   const float* wt_scale, const float* qh, int64_t ldq, int hid, const float* w2,
                                   float b2, const float* b2_dev, float slope, int exp2_minus_1, float* out,
                                   int64_t ldo, int num_q, desco_stream_t stream);

/* backward of desco_count_head_f32 (logit mode): given dl[b,q] = dLoss/dlogit,
 * dt[b,c], dqh[q,c] (contiguous [num_q, hid]), dw2[c];  (db2 = sum dl is left to the caller)
 * workspace: desco_count_head_bwd_workspace(num_b, num_q, hid) bytes (ABI 1 documented a constant of
 * 256 * (num_q+1) * hid floats; the slab count grew to 1024 -- ask, do not assume) */
size_t desco_count_head_bwd_workspace(int64_t num_b, int num_q, int hid);
int desco_count_head_bwd_f32(const float* t, int64_t ldt, const float* qh, int64_t ldq, int hid,
                             const float* w2, float slope, const float* dl, int64_t lddl,
                             int64_t num_b, int num_q, float* dt, int64_t lddt, float* dqh,
                             float* dw2, float* workspace, desco_stream_t stream);

/* Training-path helpers of the gossip stage (the fused inference kernel has the same terms folded
 * into its epilogues): a rank-ks affine term with per-query coefficient vectors,
 *   out[r,:] = act( base[r,:] + sum_{k<ks} c[r,k] * v[r % qv][k][:] ),   rows of 64, ks <= 8,
 * its weight gradient dv[qv][k][:] = sum_{r = i*qv+q} c[r,k] * dz[r,:]
 * (workspace: 64 * qv * ks * 64 floats), and a row-wise dot product out[r] = a[r,:] . b[r,:]. */
int desco_affine_rows_f32(const float* base, const float* c, int ks, const float* v, int qv,
                          int act, float slope, float* out, int64_t num_rows,
                          desco_stream_t stream);
int desco_affine_rows_bwd_f32(const float* c, int ks, const float* dz, int qv, int64_t num_rows,
                              float* dv, float* workspace, desco_stream_t stream);
int desco_rowdot2_f32(const float* a, const float* b, int ncols, float* out, int64_t num_rows,
                      desco_stream_t stream);

/* ---- round 6: dropout of the training steps (desco_dropout above) -------------------------------------------------- */

/* key_out[0..1] = state[0..1] = (seed, step); state[1] += 1.  One launch per training forward pass: the step's kernels
 * (forward and backward) read key_out, the next step gets the next counter -- also when the step is a replayed hipGraph. */
int desco_rng_next(uint64_t* state, uint64_t* key_out, desco_stream_t stream);
/* out[r, c] = the factor (0 or d->scale) of element (r, c): what the fused epilogues multiply by.  For tests (the oracle
 * takes the mask as an input) and for callers that want the mask itself. */
int desco_dropout_mask_f32(const desco_dropout* d, int64_t num_rows, int num_cols, float* out, int64_t ldo,
                           desco_stream_t stream);
/* desco_affine_rows_f32 followed by dropout: out = factor(r, c) * act(base + sum_k c[r,k] v[r % qv][k][:]) */
int desco_affine_rows_dropout_f32(const float* base, const float* c, int ks, const float* v, int qv, int act,
                                  float slope, const desco_dropout* d, float* out, int64_t num_rows,
                                  desco_stream_t stream);
/* desco_act_grad_f32 through a dropout: dz[r, c] = dc[r, c] * factor(r, c) * act'(c[r, c]) for c = factor * act(z)
 * (contiguous [num_rows, num_cols]) */
int desco_act_grad_dropout_f32(const float* dc, const float* c, int act, float slope, const desco_dropout* d,
                               float* dz, int64_t num_rows, int num_cols, desco_stream_t stream);

/* ---- round 5: the glue of the training steps (lightning_model.py:228-254, 285-289, 585-608, 630-635) ----------------- */

/* Strided 2-D copies / transposes, any number per call (24 per launch): dst[r][c] = src[r][c], or with transpose != 0
 * dst[c][r] = src[r][c] (dst is then cols x rows); accumulate != 0 adds into dst.  The moves between nn.Linear's
 * [out, in] layout and the K-major operands of the GEMM entry points, the canonical rows into the anchor operand, the
 * halves of a split weight -- what torch's .t().contiguous() / cat / slice assignment launched in rounds 2-4. */
typedef struct desco_copy2d_desc {
  const float* src; int64_t lds;
  float* dst; int64_t ldd;
  int32_t rows, cols, transpose, accumulate;
} desco_copy2d_desc;
int desco_copy2d_multi_f32(int num, const desco_copy2d_desc* descs, desco_stream_t stream);

/* The weight folding of the fused SHMP layer (gnn_model.py:253-277: per edge type SAGEConv.lin, per node type the update
 * Linear over cat(aggregate, x)) for all layers of one row type, read from the parameters where torch keeps them:
 *   wt[l][s 64 + k][n] = sum_j U_l[n][j] W_{l,s}[j][k]  (s < slots),   wt[l][slots 64 + k][n] = U_l[n][64 + k],
 *   fb[l][n] = sum_j U_l[n][j] (sum_u b_{l,u}[j]) + c_l[n].
 * table: device array, per layer [U, c, W_0 .. W_{slots-1}, b_0 .. b_{num_bias-1}] as int64 device addresses (U
 * [64, 128], W [64, 64], c / b [64], contiguous fp32; slots of one relation carry the same W address).
 * bwd: from dwt / dfb the gradient of every parameter, written at grad_offsets[same index] (in floats) of `grads`
 * (tied W: the sum over its slots, written once).  Exact fp32 FMA chains in a fixed order. */
int desco_fold_shmp_fwd_f32(const int64_t* table, int num_layers, int slots, int num_bias, float* wt, float* fb,
                            desco_stream_t stream);
int desco_fold_shmp_bwd_f32(const int64_t* table, const int64_t* grad_offsets, int num_layers, int slots, int num_bias,
                            const float* dwt, const float* dfb, float* grads, desco_stream_t stream);

/* The two training losses and their gradients in one pass (+ a one-block fold of the partial sums, fixed order):
 *   mode 0  loss = mean_i smooth_l1(pred[i] - log2(y[i] + 1))     (lightning_model.py:285-289, 246-254; beta = 1)
 *   mode 1  loss = sum_i log2(|pred[i] - y[i]| + 1)                (lightning_model.py:630-635)
 * dpred[i] = d loss / d pred[i].  workspace: 1024 floats. */
int desco_loss_f32(const float* pred, const float* y, int64_t count, int mode, float* loss, float* dpred,
                   float* workspace, desco_stream_t stream);

/* out[i] = a[i] * mul[0] + add[0] + addv[i], every term but a optional (NULL): a saved gradient times the upstream
 * gradient of its scalar loss; x + correction + post_mp.7's bias.  p[i] = value. */
int desco_affine_scalar_f32(const float* a, const float* mul, const float* add, const float* addv, float* out,
                            int64_t count, desco_stream_t stream);
int desco_fill_f32(float* p, float value, int64_t count, desco_stream_t stream);

/* The operands of the gossip training trunk as functions of the raw parameters (gnn_model.py:58-103, 230-260, 303-350;
 * algebra DESIGN.md 4.2), for num_q <= 64 queries with embeddings E [num_q, 64]; E, w_pre = pre_mp.weight[:, 0] and
 * b_pre = pre_mp.bias carry no gradient (the layer-0 input is detached, gnn_model.py:236-240):
 *   V0 [Q,6,64] = [p, g0 p, r, g0 r, t, z], V1 [Q,3,64] = [u, g1 u, db1], Vp [Q,2,64] = [tp, zp], wt1 = [(D1a C1)^T; D1b^T],
 *   wtp = [P0c^T; P0d^T], w3t = P3^T, w5t = P5^T, gates g0 / g1 = lin_gate(E), g1c = 1 - g1      (formulas: train_native.hip)
 * All matrices in nn.Linear's [out, in] layout: C0 [64,128], D0 [64,192], C1 [64,64], D1 [64,128], G0 [64,64], g2 [64]
 * (lin_gate.2.weight[0]), gb2 [1], P0 [64,256], P3 [64,64], P5 [256,64].  a, h0, h1 [Q,64]: saved for the backward.
 * bwd: the gradient of every parameter from the gradients of the outputs (dg1: the trunk's own gate gradient, may be
 * NULL); scratch: 4 * num_q * 64 floats.  One workgroup each; exact fp32 FMA chains in a fixed order. */
typedef struct desco_gossip_fold_params {
  const float *E, *w_pre, *b_pre;
  const float *C0, *cb0, *D0, *db0, *C1, *cb1, *D1, *db1;
  const float *G0[2], *gb0[2], *g2[2], *gb2[2];
  const float *P0, *p0, *P3, *P5;
  int32_t num_q;
} desco_gossip_fold_params;
typedef struct desco_gossip_fold_out {
  float *V0, *V1, *Vp, *wt1, *wtp, *w3t, *w5t, *g0, *g1, *g1c, *a, *h0, *h1;
} desco_gossip_fold_out;
typedef struct desco_gossip_fold_grads {
  const float *dV0, *dV1, *dVp, *dwt1, *dwtp, *dw3t, *dw5t, *dg1;
  float *dC0, *dcb0, *dD0, *ddb0, *dC1, *dcb1, *dD1, *ddb1;
  float *dG0[2], *dgb0[2], *dg2[2], *dgb2[2];
  float *dP0, *dp0, *dP3, *dP5;
  float* scratch;
} desco_gossip_fold_grads;
int desco_gossip_fold_fwd_f32(const desco_gossip_fold_params* p, const desco_gossip_fold_out* o, desco_stream_t stream);
int desco_gossip_fold_bwd_f32(const desco_gossip_fold_params* p, const desco_gossip_fold_out* o,
                              const desco_gossip_fold_grads* d, desco_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DESCO_HIP_H */
