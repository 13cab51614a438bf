/*
 * evi_hip.h — C-ABI of the MI355X (gfx950) evidence-retrieval hot path.
 *
 * Drop-in boundary for the retriever evaluation path of Martin1007Wang/EVI-RAG.  The reference
 * is pure Python (no FFI of its own), so every entry point below cites the reference function
 * (file:line under the reference checkout) whose arithmetic it replaces.  INTEGRATION.md shows
 * the ctypes binding a reference maintainer would add at each call site.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / C++ types cross this boundary.
 *   - Every pointer is a DEVICE pointer (HBM) unless its name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued on that one stream; nothing here synchronises the device, allocates device
 *     memory, or copies to the host, so every call can be captured into a hipGraph.
 *   - Scratch memory comes from the caller: ask evi_*_workspace_bytes(), allocate once, pass it.
 *   - Return value: 0 (EVI_OK) or a negative errno-style code; evi_last_error() holds the text
 *     of the last failure on the calling thread.  The Python shim maps EVI_ERR_INVALID to
 *     ValueError and the others to RuntimeError (the reference's own conventions, e.g.
 *     src/models/components/retriever.py:203,423,431,618; src/utils/graph_utils.py:58-99).
 *   - Thread-compatible: calls on different streams may run concurrently from different threads;
 *     a workspace must not be shared by two in-flight calls.
 *   - Integer results (row ids, edge ids, levels, masks, hit counts) are bit-exact w.r.t. the
 *     CPU oracle under oracle/; float results are within 1e-3 absolute (tests state tighter
 *     per-kernel bounds).  Ranking order everywhere is (score descending, index ascending),
 *     the order of the reference's only defined sort: argsort(descending=True, stable=True)
 *     at src/data/components/g_agent_builder.py:651.
 */
#ifndef EVI_HIP_H_
#define EVI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVI_OK               0
#define EVI_ERR_INVALID    (-22) /* EINVAL: bad shape / range / null pointer        */
#define EVI_ERR_NOMEM      (-12) /* ENOMEM: workspace too small                     */
#define EVI_ERR_HIP         (-5) /* EIO:    a HIP runtime call or launch failed     */
#define EVI_ERR_UNSUPPORTED (-95)/* EOPNOTSUPP: shape outside the built kernels     */

#define EVI_ABI_VERSION 1
#define EVI_TOPK_MAX_K 2048     /* largest k any top-k entry point accepts */

/* ---- library ---------------------------------------------------------------------------- */

/* ABI version of the loaded library (EVI_ABI_VERSION at build time). */
int evi_version(void);

/* Copies the calling thread's last error text (NUL-terminated, truncated) into buf; returns the
 * number of bytes the full message needs (excluding the NUL). */
size_t evi_last_error(char* buf, size_t buf_bytes);

/* Optional per-kernel timing for the bench's roofline leg.  While enabled, each call times its
 * dominant kernels with hipEvents on the call's stream (class 0: the cosine scan kernel,
 * class 1: top-k selection / re-scoring kernels, class 2: scorer GEMMs, class 3: edge-feature kernel).  The scan and
 * selection kernels are launched with hipExtLaunchKernelGGL start / stop events, which carry the dispatch's own begin
 * and end times and put nothing between the kernels of the stream; the scorer classes bracket groups of kernels with
 * recorded events.  evi_timing_read synchronises those events, writes the summed
 * milliseconds and launch counts per class into HOST arrays of n_classes entries, and clears the
 * log.  Calls made while timing is enabled must not be captured into a hipGraph. */
int evi_timing_enable(int on);
int evi_timing_read(double* ms_host, int32_t* launches_host, int n_classes);

/* ---- C1: row normalisation -------------------------------------------------------------- */

/* inv_norm[i] = 1 / max(||x[i,:]||_2, eps).  x: [n, D] f32 row-major.
 * Replaces the denominator of _normalize_embeddings,
 * scripts/build_retrieval_pipeline.py:833-837 (clamp is on the norm, not its square). */
int evi_row_inv_norm(const float* x, int64_t n, int D, float eps, float* inv_norm, void* stream);

/* out[i,:] = x[i,:] / max(||x[i,:]||_2, eps) (true division, as the reference writes it).
 * In-place (out == x) is allowed.  Zero rows stay zero rows.
 * Replaces _normalize_embeddings, scripts/build_retrieval_pipeline.py:833-837. */
int evi_row_normalize(const float* x, int64_t n, int D, float eps, float* out, void* stream);

/* ---- C3 generalised: dense query x index cosine top-k ------------------------------------ */

/* Bytes of scratch evi_cosine_topk needs for this problem (an upper bound that is safe for any
 * data, including an index sorted by score).  A smaller workspace is accepted down to
 * evi_cosine_topk_min_workspace_bytes(); it only shortens the row segments between threshold
 * updates. */
size_t evi_cosine_topk_workspace_bytes(int Q, int64_t N, int D, int k);
size_t evi_cosine_topk_min_workspace_bytes(int Q, int64_t N, int D, int k);

/* For every query row q[i,:] return the k rows of idx with the largest
 *     score = (sum_d q[i,d] * idx[r,d]) * (row_scale ? row_scale[r] : 1)
 * ordered by (score desc, row id asc).  With q and idx L2-normalised by evi_row_normalize (or
 * idx raw and row_scale = evi_row_inv_norm(idx)) this is the cosine the reference computes at
 * scripts/build_retrieval_pipeline.py:868-873 (index_select + mv + argmax), generalised from a
 * per-group arg-max to a global top-k.
 *
 *   q          [Q, D] f32          Q >= 1
 *   idx        [N, D] f32          N >= 0; D % 16 == 0 and 16 <= D <= 1280
 *   row_scale  [N] f32 or NULL
 *   k          1..EVI_TOPK_MAX_K
 *   row_id_base  added to every returned row id (the shard's first global row, for a row-sharded
 *                index; 0 otherwise)
 *   out_score  [Q, k] f32   slots beyond min(k, N) are filled with -inf
 *   out_index  [Q, k] i64   slots beyond min(k, N) are filled with -1
 *
 * Arithmetic: every dot product is one f32 FMA chain in a fixed order of d (MFMA f32 16x16x4),
 * so a row's score does not depend on N, on the shard it sits in, or on the workspace size. */
int evi_cosine_topk(const float* q, int Q, const float* idx, int64_t N, int D,
                    const float* row_scale, int k, int64_t row_id_base,
                    float* out_score, int64_t* out_index,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Same contract with the index stored as IEEE f16 ([N, D] halves, D % 32 == 0): half the HBM bytes
 * per pass.  Queries stay f32: each is split q = hi + lo into two f16 values and multiplied on
 * v_mfma_f32_16x16x32_f16 with f32 accumulation, so scores equal the f32 dot product with the
 * (f16-rounded) stored rows to ~1e-6.  For the 100 M-row index of BASELINE config 4. */
int evi_cosine_topk_f16(const float* q, int Q, const void* idx_f16, int64_t N, int D,
                        const float* row_scale, int k, int64_t row_id_base,
                        float* out_score, int64_t* out_index,
                        void* workspace, size_t workspace_bytes, void* stream);

/* norms[i] = ||x_i||_2 (f32 sum of squares, one wave per row).  torch.norm(features, p=2, dim=-1) of
 * FeatureMonitor.update, src/metrics/feature_monitor.py:42-46. */
int evi_row_norms(const float* x, int64_t n, int D, float* norms, void* stream);

/* fp8 storage (BASELINE config 5): rows quantised to OCP e4m3 with one f32 scale per row,
 *   scale = max|x| / 448 (1 for an all-zero row), out = e4m3(x / scale), round-to-nearest-even. */
int evi_quantize_rows_fp8(const float* x, int64_t n, int D, uint8_t* out_fp8, float* out_scale, void* stream);

/* Same contract as evi_cosine_topk over an e4m3 index ([N, D] bytes, D % 64 == 0) with its per-row
 * scale in row_scale (required): score = (sum_d q[i,d] * e4m3(idx[r,d])) * row_scale[r].  A quarter
 * of the f32 HBM bytes.  The bytes are widened to f16 in registers (exactly) and multiplied with the
 * hi + lo split f32 queries on f16 MFMA, so the only error is the index quantisation itself: results
 * are exact w.r.t. the dequantised rows and are compared with the f32 index by overlap@k. */
int evi_cosine_topk_fp8(const float* q, int Q, const void* idx_fp8, int64_t N, int D,
                        const float* row_scale, int k, int64_t row_id_base,
                        float* out_score, int64_t* out_index,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same index through the NATIVE fp8 matrix instruction (v_mfma_f32_16x16x32_fp8_fp8): the e4m3 bytes are the B
 * operand as they are (no widening work on the stream), the f32 query is written as two e4m3 pieces with a
 * power-of-two scale each (8 significant bits per element) and accumulated separately.  Scores differ from
 * evi_cosine_topk_fp8's (which are exact for the dequantised rows) by <= 2^-8 |q|_inf |row|_1 in the worst case,
 * ~20x below the index's own e4m3 rounding; ranking differences are reported as overlap@k by the callers. */
int evi_cosine_topk_fp8_mfma(const float* q, int Q, const void* idx_fp8, int64_t N, int D,
                             const float* row_scale, int k, int64_t row_id_base, float* out_score,
                             int64_t* out_index, void* workspace, size_t workspace_bytes, void* stream);

/* The same result as evi_cosine_topk (ids and scores bit-identical) for MANY queries (Q in the hundreds; f32
 * index, rows of unit norm after row_scale, D % 16 == 0, k <= 1365): the scores of all queries against a
 * slab of rows are formed as one split-bf16 MFMA GEMM and used only to SELECT k + reserve candidates per
 * query; the candidates are re-scored with the scan's own f32 MFMA sequence and the top-k is taken on
 * those.  *status (device int32) is written 0 when the result is proven exact; non-zero (1: the k-th and the
 * last kept approximate scores are closer than twice the GEMM's error bound — heavy ties; 2: a candidate
 * list overflowed — adversarial row order) means the outputs must be discarded and evi_cosine_topk run
 * instead.  One pass over the index instead of ceil(Q / 32).
 * products = 3: split-bf16 selection scores (error 2.5e-4 |q|, reserve max(256, k/2), k <= 1365);
 * products = 1: plain bf16 selection scores, three times fewer MFMAs (error 4.2e-3 |q|, reserve max(1024, k),
 *               k <= 1024) — the proof fails earlier on large or clustered indexes; try 1, then 3, then the scan.
 * shadow_bf16 (nullable, products = 1 only): a bf16 copy of the f32 index made once by evi_index_shadow_bf16
 * (+ 50 % memory).  The selection GEMM then streams half the bytes and converts nothing; the arithmetic - and
 * so the result and the proof - is that of products = 1. */
size_t evi_cosine_topk_gemm_workspace_bytes(int Q, int64_t N, int D, int k);
int evi_index_shadow_bf16(const float* idx, int64_t N, int D, void* out_bf16, void* stream);
int evi_cosine_topk_gemm(const float* q, int Q, const float* idx, int64_t N, int D, const float* row_scale,
                         int k, int64_t row_id_base, int products, const void* shadow_bf16, float* out_score,
                         int64_t* out_index, int32_t* status, void* workspace, size_t workspace_bytes, void* stream);
/* The same over an f16-stored index (D % 32 == 0): bit-identical to evi_cosine_topk_f16.  An f16 value is exactly
 * hi + lo in bf16, so the selection GEMM loses nothing on the index side. */
int evi_cosine_topk_gemm_f16(const float* q, int Q, const void* idx_f16, int64_t N, int D, const float* row_scale,
                             int k, int64_t row_id_base, int products, float* out_score, int64_t* out_index,
                             int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* Two-stage exact scan: the result of evi_cosine_topk (ids and scores bit-identical) at half the HBM bytes per
 * batch, for an f32 index of rows with norm <= 1 (evi_row_normalize output; no row_scale) that is kept together
 * with an f16 copy of itself - the shadow, rn_f16(x) element by element, made once by evi_index_shadow_f16
 * (+ 50 % memory; D % 32 == 0).  Stage 1 streams the shadow (evi_cosine_topk_f16 arithmetic) and keeps
 * k + max(256, k/2) rows per query; their scores are within 7e-4 |q| of the f32 scan's (f16 rounding of a row
 * moves a dot product by at most 2^-11 |q| |x|, plus the two accumulation chains), so when the k-th and the last
 * kept shadow scores are more than twice that apart the true top-k is among the kept rows.  Stage 2 re-scores
 * them from the f32 rows with the scan's own v_mfma_f32_16x16x4_f32 chain and takes the top-k.
 * *status (device int32, zeroed by the CALLER; bits are OR-ed in, so a pipeline of batches can share one flag and
 * read it once): 0 = proven exact; bit 0 = the gap test failed for some query (heavy ties, NaN scores).
 * device_fallback != 0 (what the host mirrors use): the f32 scan of the same batch is enqueued behind stage 2 into
 * the same outputs, every kernel of it gated on this call's own proof flag - while the proof held they return at
 * once (a few empty launches), when it failed they overwrite the unproven result on the device, so the outputs are
 * ALWAYS those of evi_cosine_topk, with no read-back and no agreement between ranks; status may then be NULL and is
 * only a record that a fallback ran.  device_fallback == 0: a non-zero status means discard the outputs and run
 * evi_cosine_topk (status required).  The shadow must be the one evi_index_shadow_f16 made from idx: the proof is
 * about that pair.  k <= 1365.  Never synchronises. */
int evi_index_shadow_f16(const float* idx, int64_t N, int D, void* out_f16, void* stream);
size_t evi_cosine_topk_two_stage_workspace_bytes(int Q, int64_t N, int D, int k);
int evi_cosine_topk_two_stage(const float* q, int Q, const float* idx, const void* shadow_f16, int64_t N, int D,
                              int k, int64_t row_id_base, float* out_score, int64_t* out_index, int32_t* status,
                              int device_fallback, void* workspace, size_t workspace_bytes, void* stream);

/* Merge P per-shard top-k lists (the all-gathered outputs of evi_cosine_topk on P ranks) into
 * the global top-k, same (score desc, id asc) order; ids < 0 are padding and never win.
 *   scores [P, Q, k] f32, ids [P, Q, k] i64  ->  out_score [Q, k], out_index [Q, k].
 * Requires P * k <= 8192.  Replaces the gather at
 * src/callbacks/retriever_topk_edge_writer.py:450-462 (all_gather_object of per-sample lists). */
int evi_topk_merge(const float* scores, const int64_t* ids, int P, int Q, int k,
                   float* out_score, int64_t* out_index, void* stream);

/* The same merge on the PACKED exchange layout, so that one all-gather moves scores and ids
 * together: each rank's record is [Q*k f32 scores | pad to 8 B | Q*k i64 ids]
 * (evi_topk_packed_bytes(Q, k) bytes; evi_cosine_topk can write straight into it), and `packed`
 * holds the P records back to back in ascending shard order. */
size_t evi_topk_packed_bytes(int Q, int k);
int evi_topk_merge_packed(const void* packed, int P, int Q, int k, float* out_score, int64_t* out_index,
                          void* stream);

/* ---- T1-T3 / G9: per-graph (segmented) top-k over edge scores ---------------------------- */

/* For each graph g (edges edge_ptr[g] .. edge_ptr[g+1]) write the min(k, E_g) edges with the
 * largest score, ordered (score desc, edge position asc), as LOCAL positions within the graph.
 *   scores    [E] f32
 *   edge_ptr  [B+1] i64, non-decreasing, edge_ptr[0] = 0, edge_ptr[B] = E
 *   out_index [B, k] i32 (local edge position, -1 padding); out_score [B, k] f32 (-inf padding,
 *   may be NULL); out_count [B] i32 = min(k, E_g) (may be NULL).
 * Replaces torch.topk(scores, k, sorted=True) at src/metrics/reachability.py:146-147,
 * src/metrics/retriever_metrics.py:141-145, src/callbacks/retriever_topk_edge_writer.py:299-302
 * and the stable argsort at src/data/components/g_agent_builder.py:640-652. */
int evi_segment_topk(const float* scores, const int64_t* edge_ptr, int B, int k,
                     int32_t* out_index, float* out_score, int32_t* out_count, void* stream);

/* ---- T1-T5: fused per-graph ranking metrics ------------------------------------------------- */

/* One pass per graph over its edge scores: exact top-k_max (score desc, position asc), then for
 * every k of the window (k_values_host: num_k <= 16 strictly ascending ints on the HOST,
 * k_max = the last one <= EVI_TOPK_MAX_K):
 *   edge_recall  [B, nk] f32   hits among the first min(k, E_g) ranked edges / max(positives, 1)
 *                              (EdgeRecallAtK, src/metrics/retriever_metrics.py:132-157);
 *                              recall_valid [B] = graph has edges
 *   reach        [B, nk] u8    some answer node shares an undirected component with some seed node
 *                              after the first min(k, E_g) ranked edges (AnswerReachability,
 *                              src/metrics/reachability.py:129-179, 330-381); reach_valid [B] = graph
 *                              has edges, nodes, and an in-range seed and answer
 *   answer_hit   [B, nk] u8,  answer_recall [B, nk] f32   an / the fraction of answer ENTITY ids seen
 *                              as head or tail within the first k ranked edges
 *                              (_oracle_metrics_for_sample, src/models/reasoner_module.py:17-68;
 *                              compute_answer_hit, src/utils/metrics.py:206-238); answer_valid [B] =
 *                              1 if the graph has answers, 2 if it has more than 2048 (unsupported);
 *                              skipped when node_global_ids or answer_ids is NULL
 *   score_margin [B] f32       min positive score - max negative score; margin_valid [B] = both
 *                              classes present (ScoreMargin, retriever_metrics.py:376-391)
 *   topk_index [B, k_max] i32 (-1 padding), topk_score [B, k_max] f32 (-inf padding), topk_count [B]
 *                              the ranked list itself (RetrieverTopKEdgeWriter._select_topk_edges,
 *                              src/callbacks/retriever_topk_edge_writer.py:294-320); each may be NULL
 * target [E] u8 = labels > 0.5 (NULL skips recall and margin); q_idx / a_idx are the batch-global
 * q_local_indices / a_local_indices with their [B+1] pointers; uf_workspace: [2 * num_nodes] i32. */
int evi_retriever_metrics(
    const float* scores, const uint8_t* target, const int64_t* edge_index, int64_t E,
    const int64_t* edge_ptr, const int64_t* node_ptr, int B, const int64_t* q_idx, const int64_t* q_ptr,
    const int64_t* a_idx, const int64_t* a_ptr, const int64_t* node_global_ids, const int64_t* answer_ids,
    const int64_t* answer_ptr, const int32_t* k_values_host, int num_k, float* edge_recall,
    uint8_t* recall_valid, uint8_t* reach, uint8_t* reach_valid, uint8_t* answer_hit, float* answer_recall,
    uint8_t* answer_valid, float* score_margin, uint8_t* margin_valid, int32_t* topk_index,
    float* topk_score, int32_t* topk_count, int32_t* uf_workspace, void* stream);

/* Adds one batch's per-graph outputs of evi_retriever_metrics into the epoch's f64 state vector
 * acc [4 * num_k + 6] (fixed summation order, nothing read back):
 *   [0, nk) sum edge recall@k, [nk] graphs with edges; [nk+1, 2nk+1) reachability hits@k, [2nk+1] graphs
 *   with seeds and answers; [2nk+2, 3nk+2) answer hit@k, [3nk+2, 4nk+2) answer recall@k, [4nk+2] graphs
 *   with answer ids; [4nk+3] sum of score margins, [4nk+4] graphs with both classes; [4nk+5] graphs
 *   whose answer list exceeded the kernel's capacity.  answer_* may be null (no answer ids in the batch).
 * The torchmetrics `add_state(..., dist_reduce_fx="sum")` bookkeeping of EdgeRecallAtK /
 * AnswerReachability / ScoreMargin (src/metrics/retriever_metrics.py:95-99,141-166; reachability.py:27-33)
 * without their per-batch `.item()` reads. */
int evi_metric_accumulate(const float* edge_recall, const uint8_t* recall_valid, const uint8_t* reach,
                          const uint8_t* reach_valid, const uint8_t* answer_hit, const float* answer_recall,
                          const uint8_t* answer_valid, const float* score_margin, const uint8_t* margin_valid,
                          int B, int num_k, double* acc, void* stream);

/* per graph: out[g, 0..3] = {positives, negatives, sum sigmoid(score) over positives, over negatives}
 * (f64, deterministic).  Building block of BridgeProbQuality / BridgePositiveCoverage,
 * src/metrics/retriever_metrics.py:270-327, 400-476 (applied to the bridge-edge sub-lists).
 *   scores [E] f32, target [E] u8, edge_ptr [B+1] i64, out [B, 4] f64. */
int evi_graph_class_stats(const float* scores, const uint8_t* target, const int64_t* edge_ptr, int B,
                          double* out, void* stream);

/* ---- D1: embedding feed ---------------------------------------------------------------------- */

/* out[i, :] = table[ids[i], :] from an HBM-resident table.  An id outside [0, num_rows) zero-fills its
 * row and ORs 1 into *status, which the CALLER zeroes (so several gathers can share one flag that is
 * read once).  Replaces the CPU index_select + pinned-buffer H2D copy of
 * GlobalEmbeddingStore.get_entity_embeddings / get_relation_embeddings,
 * src/data/components/embedding_store.py:101-150. */
int evi_gather_rows(const float* table, int64_t num_rows, int D, const int64_t* ids, int64_t n,
                    float* out, int32_t* status, void* stream);

/* Batch collation from a split held in HBM as flat arrays (every sample's items concatenated, one
 * pointer array src_ptr[num_samples + 1] per field family).  evi_segment_offsets: out_ptr[b + 1] =
 * items of samples ids[0..b] (status bit 1 = an id outside [0, num_samples)).  evi_gather_segments
 * copies sample ids[b]'s items (row_words words of word_bytes = 4 or 8 each) to
 * out + out_ptr[b] * row_words, adding add[b] to 8-byte words when add != null — the per-sample
 * increment PyG's collate applies to index fields (num_nodes for edge_index, q/a_local_indices and
 * pair_*_node_locals, num_edges for pair_edge_local_ids: GRetrievalData.__inc__,
 * src/data/g_retrieval_dataset.py:29-37).  Replaces LMDB get + unpickle + torch_geometric Collater
 * (src/data/components/loader.py:22-99) for a resident split. */
int evi_segment_offsets(const int64_t* src_ptr, int64_t num_samples, const int64_t* ids, int B,
                        int64_t* out_ptr, int32_t* status, void* stream);
int evi_gather_segments(const void* src, int word_bytes, int64_t row_words, const int64_t* src_ptr,
                        int64_t num_samples, const int64_t* ids, int B, const int64_t* out_ptr,
                        const int64_t* add, void* out, void* stream);

/* ---- G11: edge -> graph assignment and the Q/A "near" mask --------------------------------- */

/* edge_batch[e] = bucketize(edge_index[0, e], node_ptr[1:], right=True) (node_ptr[g] <= v <
 * node_ptr[g+1]); edge_ptr = exclusive prefix of the per-graph edge counts.  `status` (one int32
 * on the device) receives a bit mask the host shim turns into the reference's ValueErrors:
 * 1 = graph id out of range, 2 = head and tail in different graphs ("edge_index crosses graph
 * boundaries"), 4 = edge list not grouped by graph ("edge_batch is not non-decreasing").
 *   edge_index [2, E] i64; node_ptr [B+1] i64; edge_batch [E] i64; edge_ptr [B+1] i64.
 * Replaces compute_edge_batch, src/utils/graph_utils.py:50-104. */
int evi_edge_batch(const int64_t* edge_index, int64_t E, const int64_t* node_ptr, int B,
                   int64_t* edge_batch, int64_t* edge_ptr, int32_t* status, void* stream);

/* out_mask[e] = head in (Q u A) or tail in (Q u A); status bit 1 = a q/a index outside
 * [0, num_nodes).  node_mask_ws: [num_nodes] bytes of scratch.
 * Replaces compute_qa_edge_mask, src/utils/graph_utils.py:107-153. */
int evi_qa_edge_mask(const int64_t* edge_index, int64_t E, int64_t num_nodes, const int64_t* q_idx,
                     int64_t nq, const int64_t* a_idx, int64_t na, uint8_t* node_mask_ws,
                     uint8_t* out_mask, int32_t* status, void* stream);

/* ---- G1: adjacency as CSR ------------------------------------------------------------------- */

/* In-edge and out-edge CSR of every graph of the batch (one workgroup per graph).  Row v of the
 * in-CSR lists (source node, edge id) of the edges u -> v; the out-CSR lists (target, edge id) of
 * v -> w.  Together they are the undirected adjacency of _build_undirected_adjacency
 * (scripts/build_retrieval_pipeline.py:570-586); self loops appear once in each half.  Node and
 * edge ids are batch-global int32; the order inside a row is unspecified.
 *   *_ptr [N+1] i32, *_nbr [E] i32, *_eid [E] i32; workspace: evi_graph_csr_workspace_bytes(N). */
size_t evi_graph_csr_workspace_bytes(int64_t N);
int evi_graph_csr(const int64_t* edge_index, int64_t E, const int64_t* node_ptr, const int64_t* edge_ptr,
                  int B, int64_t N, int32_t* in_ptr, int32_t* in_nbr, int32_t* in_eid, int32_t* out_ptr,
                  int32_t* out_nbr, int32_t* out_eid, void* workspace, size_t workspace_bytes, void* stream);

/* ---- G2/G3: BFS levels and shortest-path labelling --------------------------------------------- */

/* num_jobs independent multi-source BFS runs, one workgroup each.  Job j explores graph
 * job_graph[j] from the batch-global source nodes src_idx[src_ptr[j] .. src_ptr[j+1]) (sources
 * outside the graph are ignored) and writes levels (unreachable = -1) for the graph's nodes, in
 * local node order, at dist_out + dist_off[j].  mode 0: undirected adjacency (both CSR halves),
 * 1: along edges, 2: against edges.
 * Replaces _bfs_dist over _build_undirected_adjacency / _build_directed_adjacency,
 * scripts/build_retrieval_pipeline.py:570-631. */
int evi_bfs_levels(const int32_t* job_graph, const int64_t* src_ptr, const int64_t* src_idx,
                   const int64_t* dist_off, int num_jobs, const int64_t* node_ptr, const int32_t* in_ptr,
                   const int32_t* in_nbr, const int32_t* out_ptr, const int32_t* out_nbr, int mode,
                   int32_t* dist_out, void* stream);

/* The same levels with the batch's edge list given as well (edge_index [2, E] i64 batch-global, edge_ptr [B+1]): a graph of at
 * most 12 288 edges and 32 767 nodes (every WebQSP / CWQ graph) is searched EDGE-parallel — every thread of the workgroup keeps
 * <= 12 edges in registers as packed 16-bit (u, v) pairs, the levels sit in LDS, and a level is one round of independent LDS
 * reads, a few plain stores and one barrier: no queue, no row gathers, no CSR.  Larger graphs take evi_bfs_levels' path inside
 * the same launch (that is what the CSR arguments are for). */
int evi_bfs_levels_edges(const int32_t* job_graph, const int64_t* src_ptr, const int64_t* src_idx,
                         const int64_t* dist_off, int num_jobs, const int64_t* node_ptr, const int64_t* edge_ptr,
                         const int64_t* edge_index, int64_t E, const int32_t* in_ptr, const int32_t* in_nbr,
                         const int32_t* out_ptr, const int32_t* out_nbr, int mode, int32_t* dist_out, void* stream);

/* Shortest-path DAG edges of (seed, answer) pairs.  Pair slot p belongs to graph pair_graph[p]; its
 * seed / answer distances are the BFS jobs pair_seed_job[p] / pair_answer_job[p] of evi_bfs_levels
 * (answers explored with mode 2 when directed) and pair_answer_node[p] is the batch-global answer.
 *   pass 0: pair_len[p] = dist_seed[answer] (-1: no path, the reference emits no pair),
 *           pair_edge_count[p], and edge_mask[e] = 1 (caller zeroes it) for every edge u->v with
 *           dist_s[u] + 1 + dist_a[v] == pair_len[p], in either orientation unless `directed`;
 *   pass 1: the pair's edge ids (batch-global, ascending) at pair_edge_ids[pair_edge_off[p] ...].
 * Replaces _shortest_path_union_mask_by_pair(_directed) and _select_shortest_edges_*,
 * scripts/build_retrieval_pipeline.py:650-815. */
int evi_shortest_path_pairs(int pass, const int32_t* pair_graph, const int32_t* pair_seed_job,
                            const int32_t* pair_answer_job, const int64_t* pair_answer_node, int num_pairs,
                            const int64_t* dist_off, const int32_t* dist, const int64_t* edge_index, int64_t E,
                            const int64_t* node_ptr, const int64_t* edge_ptr, int directed, int32_t* pair_len,
                            int32_t* pair_edge_count, uint8_t* edge_mask, const int64_t* pair_edge_off,
                            int64_t* pair_edge_ids, void* stream);

/* The reference's deterministic single shortest path between a source set and a target set over the
 * undirected graph (G4).  Job j works on graph job_graph[j] with batch-global sources
 * src_idx[src_ptr[j]..) and targets tgt_idx[tgt_ptr[j]..) (out-of-range ids are ignored) and uses
 * dist_ws[dist_off[j] .. + 2 * N_g) as scratch.  out_len[j] = number of edges (-1: no source, no
 * target or no path); out_nodes[j, 0..len] graph-local node ids from the source side;
 * out_edges[j, 0..len) graph-local edge ids.  A path longer than path_cap is reported through
 * out_len only (its first path_cap hops are written).  The target is the nearest reachable one
 * (ties: smallest id); the path is the lexicographically smallest node sequence among all shortest
 * paths, each hop on the smallest edge id joining its two nodes — what the reference's FIFO BFS over
 * (neighbour, edge id)-sorted adjacency returns.  CSR (with edge ids) from evi_graph_csr.
 * Replaces _shortest_path_single, scripts/build_retrieval_pipeline.py:453-530. */
int evi_shortest_path_single(const int32_t* job_graph, const int64_t* src_ptr, const int64_t* src_idx,
                             const int64_t* tgt_ptr, const int64_t* tgt_idx, const int64_t* dist_off, int num_jobs,
                             const int64_t* node_ptr, const int64_t* edge_ptr, const int32_t* in_ptr,
                             const int32_t* in_nbr, const int32_t* in_eid, const int32_t* out_ptr,
                             const int32_t* out_nbr, const int32_t* out_eid, int32_t* dist_ws, int path_cap,
                             int32_t* out_len, int64_t* out_nodes, int64_t* out_edges, void* stream);

/* ---- G5 / f2: segmented de-duplication and re-indexing ------------------------------------------- */

/* keys [T, W] i64 row-major (W = 1, 2 or 3 words), segments seg_ptr [S+1].  out_first[p] = the
 * segment-local position of the first entry of p's segment with the same key and drop == 0
 * (-1 for entries with drop[p] != 0; drop may be null).  The dict / set bookkeeping of build_graph
 * (node_index, edge_key_to_indices: scripts/build_retrieval_pipeline.py:1465-1497) and of
 * GAgentBuilder._build_and_add_sample (triple_to_agg: src/data/components/g_agent_builder.py:338-354). */
size_t evi_first_occurrence_workspace_bytes(int64_t T, int S);
int evi_first_occurrence(const int64_t* keys, int W, int64_t T, const int64_t* seg_ptr, int S,
                         const uint8_t* drop, int32_t* out_first, void* workspace, size_t workspace_bytes,
                         void* stream);

/* From out_first: out_rank[p] = index of p's key in first-seen order (the reference's
 * local_index(), :1470-1476; list(triple_to_agg.keys()), g_agent_builder.py:356), out_count[s] = number
 * of distinct keys, out_uniq_pos[seg_ptr[s] + r] = segment-local position of the r-th distinct key
 * (may be null).  With limit (may be null), only positions < limit[s] found new keys; later entries
 * are look-ups that take the rank of the entry they match, or -1. */
int evi_first_seen_rank(const int32_t* first, int64_t T, const int64_t* seg_ptr, int S, const int64_t* limit,
                        int32_t* out_rank, int32_t* out_count, int32_t* out_uniq_pos, void* stream);

/* Stable ascending sort position of each of the first seg_len[s] (null: all) keys of every segment,
 * and the sorted keys (out_sorted may be null).  Replaces torch.sort(torch.cat([heads, tails]).unique())
 * and node_map, src/data/components/g_agent_builder.py:368-371. */
int evi_segment_sort_rank(const int64_t* keys, int64_t T, const int64_t* seg_ptr, const int32_t* seg_len, int S,
                          int32_t* out_rank, int64_t* out_sorted, void* stream);

/* out[group[i]] = max(out[group[i]], values[i]) for group[i] >= 0; the caller pre-fills out (-inf).
 * Replaces the score / label max aggregation at src/data/components/g_agent_builder.py:353-354. */
int evi_group_max_f32(const float* values, const int32_t* group, int64_t T, float* out, void* stream);

/* ---- S7: the loss the eval step logs ------------------------------------------------------------------ */

/* RetrieverLoss.forward over a batch whose edges are grouped by graph (edge_ptr [B+1]):
 *   InfoNCE  mean over graphs with >= 1 positive and >= 1 negative of
 *            logsumexp_e(s_e) - logsumexp_{e positive}(s_e),  s_e = logit_e / temperature (+ log w_e),
 *            0 when the batch has no positive or no negative edge at all;
 *   BCE      (bce_weight > 0) mean over graphs of sum_e w_e * bce_with_logits(logit_e, target_e) / sum_e w_e;
 *   w_e      = edge_weight_near / edge_weight_bridge by edge_is_near when it is non-null, else 1;
 *   target_e > 0.5 marks a positive.
 * out_scalars [15] f64: 0 infonce, 1 bce, 2 infonce_weight * infonce + bce_weight * bce, 3 positive edges,
 * 4 negative edges, 5 InfoNCE graphs, 6 graphs without positives, 7 graphs without negatives,
 * 8 BCE graphs, 9 BCE edges, 10 mean sigmoid(logit | positive), 11 mean sigmoid(logit | negative),
 * 12 their difference, 13-14 internal.  grad_logits (nullable) receives d out[2] / d logits.
 * Sums are formed in f64 in a fixed order (deterministic).
 * Replaces RetrieverLoss, src/losses/retriever_loss.py:72-325 (called by
 * RetrieverModule._compute_loss_output, src/models/retriever_module.py:251-275). */
size_t evi_retriever_loss_workspace_bytes(int B);
int evi_retriever_loss(const float* logits, const float* targets, const int64_t* edge_ptr, int B,
                       const uint8_t* edge_is_near, float infonce_temperature, float infonce_weight,
                       float bce_weight, float edge_weight_near, float edge_weight_bridge,
                       double* out_scalars, float* grad_logits, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- G8/G9/G10: seed expansion and score post-processing ----------------------------------------- */

/* logit of p = (softmax of the score over the head's out-edges + softmax over the tail's in-edges)/2,
 * p clamped to [1e-6, 1 - 1e-6].  Node ids in edge_index only need to be < N (batch-global or local).
 * Replaces GAgentBuilder._node_softmax_logit, src/data/components/g_agent_builder.py:595-626. */
size_t evi_node_softmax_logit_workspace_bytes(int64_t N);
int evi_node_softmax_logit(const float* edge_scores, const int64_t* edge_index, int64_t E, int64_t N,
                           float* out_logit, void* workspace, size_t workspace_bytes, void* stream);

/* Undirected one-hop seed expansion: for every seed node keep its
 * min(deg, min(max_edges, max(min_edges, ceil(float(deg) * ratio)))) best incident edges (the seed
 * as head OR tail; a self loop counts twice) by (score desc, head-incidence before tail-incidence,
 * edge id asc); out_mask[e] = 1 for the union (the sorted unique ids are its non-zeros).
 * start_max_edges < 0 = no cap; any degree and any k are supported (radix select of the k-th key).
 * status: 1 = a seed outside [0, N).  CSR from evi_graph_csr.
 * Replaces GAgentBuilder._select_start_edges, src/data/components/g_agent_builder.py:655-724. */
int evi_select_start_edges(const float* edge_scores, int64_t E, const int64_t* seed_nodes, int64_t num_seeds,
                           const int32_t* in_ptr, const int32_t* in_eid, const int32_t* out_ptr,
                           const int32_t* out_eid, int64_t N, float start_keep_ratio, int start_min_edges,
                           int start_max_edges, uint8_t* out_mask, int32_t* status, void* stream);

/* Incident edge count and positive incident count of each seed (bincount(heads) + bincount(tails));
 * -1 for seeds outside [0, N).  Replaces scripts/seed_onehop_stats.py:96-117. */
int evi_seed_onehop_stats(const int64_t* seed_nodes, int64_t num_seeds, const uint8_t* positive,
                          const int32_t* in_ptr, const int32_t* in_eid, const int32_t* out_ptr,
                          const int32_t* out_eid, int64_t N, int32_t* out_degree, int32_t* out_positive_degree,
                          void* stream);

/* ---- E2/E3: the text-encoding tail ------------------------------------------------------------------ */

/* out[b, d] = sum_l hidden[b, l, d] * mask[b, l] / max(sum_l mask[b, l], eps)  (eps = 1e-6), in the
 * pooling dtype (f16 when pool_fp16: hidden is rounded to f16, the sum is rounded once to f16 and
 * the clamp and division happen in f16), returned as f32.  hidden_dtype: 0 f32, 1 f16, 2 bf16;
 * attention_mask [b, L] i64.  Replaces the pooling of TextEncoder.encode,
 * scripts/text_encode_utils.py:60-65. */
int evi_masked_mean_pool(const void* hidden, int hidden_dtype, const int64_t* attention_mask, int b, int L,
                         int D, int pool_fp16, float eps, float* out, void* stream);

/* table[ids[i], :] = src[i, :] for 0 <= ids[i] <= max_embedding_id (other rows skipped); when an id
 * repeats, the last row wins, as in the reference's sequential loop.
 * Replaces _write_chunk, scripts/text_encode_utils.py:137-146. */
size_t evi_scatter_rows_workspace_bytes(int64_t max_embedding_id);
int evi_scatter_rows(const float* src, const int64_t* ids, int64_t n, int D, float* table,
                     int64_t max_embedding_id, void* workspace, size_t workspace_bytes, void* stream);

/* ---- G6/G7: DDE structure features ----------------------------------------------------------- */

/* node_struct[v, c*S + j], S = 1 + rounds + rev_rounds (topic-major, as
 * stack([topic, f1.., r1..], -1).reshape(N, -1) at src/models/components/retriever.py:546-553):
 * j = 0 the topic one-hot, j = 1..rounds mean propagation along edges (PyG MessagePassing
 * aggr="mean", source_to_target), then rev_rounds rounds along reversed edges restarted from the
 * one-hot.  A node without in-edges gets 0.  Sums are formed in f64 and rounded once.
 * Replaces PEConv/DDE, src/models/components/graph.py:13-74. */
int evi_dde_node_struct(const float* topic_one_hot, int topic_stride, int num_topics, int64_t N,
                        const int32_t* in_ptr, const int32_t* in_nbr, const int32_t* out_ptr,
                        const int32_t* out_nbr, int rounds, int rev_rounds, float* node_struct, void* stream);

/* The same features with the graph boundaries given (node_ptr [B+1], batch-global): from 112 graphs per batch on, one
 * workgroup per graph builds the graph's [N_g, C*S] block in LDS (all rounds, workgroup barriers in between) and writes it out
 * once, coalesced — the node-parallel form gathers 8 bytes out of 40-byte rows that no longer fit the L2s at that size
 * (3.7x the algorithmic bytes by the memory-side counters).  Smaller batches take evi_dde_node_struct's path.  Same results
 * bit for bit (f64 row sums, rounded once). */
int evi_dde_node_struct_graphs(const float* topic_one_hot, int topic_stride, int num_topics, int64_t N,
                               const int64_t* node_ptr, int B, const int32_t* in_ptr, const int32_t* in_nbr,
                               const int32_t* out_ptr, const int32_t* out_nbr, int rounds, int rev_rounds,
                               float* node_struct, void* stream);


/* ---- S1-S6: the edge scorer ------------------------------------------------------------------- */

/* C[M,N] = act(A[M,K] * W[N,K]^T + bias), f32 in / f32 accumulate on MFMA (exact f32 FMA chains).
 * act: 0 none, 1 tanh, 2 sigmoid.  K, lda, ldw multiples of 4.  The building block of every
 * Linear in the reference Retriever (EmbeddingProjector, src/models/components/projections.py:9-40;
 * q_gate/q_bias/state_net, src/models/components/retriever.py:157-182). */
int evi_gemm_nt_f32(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw,
                    const float* bias, int act, float* C, int64_t ldc, void* stream);

/* Same contract as evi_gemm_nt_f32 at ~5x the speed: every f32 operand is split into two bf16
 * values (x = hi + lo) and the product formed as hi*hi + hi*lo + lo*hi on bf16 MFMA with f32
 * accumulation (gfx950 has no TF32; an f32-input MFMA runs at 1/16 of the bf16 rate).  Per-product
 * relative error <= ~2^-17, i.e. results agree with the exact f32 path to ~1e-5 of sum|a*w| — the
 * precision class of the TF32 matmuls the reference runs with on CUDA
 * (torch.set_float32_matmul_precision("high"), configs/extras/default.yaml:11).
 * workspace: evi_gemm_nt_bf16x3_workspace_bytes(N, K) for the split weight planes. */
size_t evi_gemm_nt_bf16x3_workspace_bytes(int N, int K);
int evi_gemm_nt_bf16x3(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw,
                       const float* bias, int act, float* C, int64_t ldc, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Parameters of src.models.components.retriever.Retriever as device pointers, in state_dict
 * order (every tensor f32, row-major, exactly the checkpoint's shapes). */
typedef struct EviRetrieverWeights {
    int emb_dim;             /* D */
    int hidden_dim;          /* H */
    int num_topics;          /* 2 */
    int dde_rounds;          /* parity_meta[2] */
    int dde_reverse_rounds;  /* parity_meta[3] */
    const float* entity_w;       /* entity_proj.network.0.weight   [D, D] */
    const float* entity_b;       /* entity_proj.network.0.bias     [D]    */
    const float* relation_w;     /* relation_proj.network.0.weight [D, D] */
    const float* relation_b;
    const float* query_w;        /* query_proj.network.0.weight    [D, D] */
    const float* query_b;
    const float* non_text_emb;   /* non_text_entity_emb.weight     [1, D] */
    const float* q_gate_w;       /* q_gate.0.weight [D, D] */
    const float* q_gate_b;
    const float* q_bias_w;       /* q_bias.0.weight [D, D] */
    const float* q_bias_b;
    const float* struct_w;       /* struct_proj.0.weight [D, 4*(1+rounds+rev)] */
    const float* struct_b;
    const float* struct_ln_w;    /* struct_proj.1.weight [D] */
    const float* struct_ln_b;
    const float* struct_gate_w;  /* struct_gate_net.0.weight [1, D] */
    const float* struct_gate_b;  /* [1] */
    const float* state0_w;       /* state_net.0.weight [H, 3D+1] */
    const float* state0_b;
    const float* state_ln_w;     /* state_net.1.weight [H] */
    const float* state_ln_b;
    const float* state4_w;       /* state_net.4.weight [H, H] */
    const float* state4_b;
    const float* score_w;        /* score_head.weight [1, H] */
    const float* score_b;        /* [1] */
    const void* prepared;        /* NULL, or the buffer evi_retriever_prepare filled FROM THESE WEIGHTS: everything the forward
                                  * derives from the weights alone (column blocks of state_net.0, the folded head, bf16 hi / lo
                                  * planes of the nine GEMM weights).  The caller re-prepares whenever a weight changes. */
} EviRetrieverWeights;

/* The flat PyG batch the reference's loader produces (src/data/components/loader.py:43-99), as
 * device pointers.  edge_batch / edge_ptr come from evi_edge_batch. */
typedef struct EviRetrieverBatch {
    int64_t num_nodes, num_edges;
    int num_graphs;
    const int64_t* edge_index;          /* [2, E], batch-global node ids */
    const int64_t* node_ptr;            /* [B+1] */
    const int64_t* edge_ptr;            /* [B+1] */
    const int64_t* edge_batch;          /* [E] */
    const float* question_emb;          /* [B, D] */
    const float* node_embeddings;       /* [N, D] */
    const int64_t* node_embedding_ids;  /* [N], 0 = non-text entity */
    const float* edge_embeddings;       /* [E, D] (= relation_table[edge_attr]); NULL allowed with relation_rows (below) */
    const int64_t* edge_attr;           /* [E] relation ids */
    int64_t num_relations;  /* if 0 < num_relations <= E: every edge_attr < num_relations and equal ids
                               carry equal edge_embeddings rows, so each relation is projected once;
                               0 = project per edge, as the reference does */
    const float* topic_one_hot;         /* [N, topic_stride] */
    int topic_stride;
    const float* edge_bias;             /* [E] or NULL: added to both directional logits before they are combined
                                           (the hide-and-seek penalty, src/models/components/retriever.py:247-256) */
    float dropout_p;                    /* training: nn.Dropout(p) between state_net's GELU and state_net.4
                                         * (src/models/components/retriever.py:179); 0 = off (evaluation) */
    uint64_t dropout_seed;              /* the mask is a counter-based hash of (seed, direction, edge, column): the backward
                                         * regenerates it from the same seed.  Not torch's Philox stream: masks are equal in
                                         * distribution (keep probability 1 - round(p 2^16) / 2^16, kept values scaled by its
                                         * reciprocal), not bit for bit */
    int matmul_precision;               /* 0 (default): every large product runs as three bf16 MFMA products of the split f32
                                         * operands (f32-grade results; what evaluation and the parity tests use).  1: ONE bf16
                                         * product with f32 accumulation and f32 results, forward and backward — at least the
                                         * arithmetic of Lightning's `precision: bf16-mixed` (bf16 autocast rounds the results
                                         * to bf16 as well), which configs/trainer/default.yaml:13-14 recommends for training
                                         * on GPUs that have it; the few-row products (question side) stay exact f32.
                                         * 2 ("f16x2", evaluation only — the backward refuses it): TWO f16 MFMA products,
                                         * activations split hi + lo in f16 (22 significant bits), each weight rounded once
                                         * to f16 (2^-12 relative: four times finer than the TF32 rounding of both operands
                                         * the reference's CUDA run applies, configs/extras/default.yaml:11); two thirds of
                                         * the matrix work of 0; operands must stay inside f16's range (|x| < 65 504) */
    const float* relation_rows;         /* [num_relations, D] or NULL: the relation table itself (row r = the embedding every edge
                                         * with edge_attr == r carries).  With it and 0 < num_relations <= E the relation rows are
                                         * taken from here — edge_embeddings may then be NULL: nobody has to gather [E, D] rows only
                                         * for this call to find one per relation again */
} EviRetrieverBatch;

/* RetrieverOutput (src/models/components/retriever.py:80-99); any pointer but logits may be NULL. */
typedef struct EviRetrieverOutput {
    float* logits;         /* [E] */
    float* logits_fwd;     /* [E] */
    float* logits_bwd;     /* [E] */
    float* edge_features;  /* [E, H]  (RetrieverOutput.edge_embeddings / extract_edge_tokens); NULL = logits
                            * only: score_head is folded into state_net.4 and the [2E, H] features are never formed */
    float* node_struct;    /* [N, 2*(1+rounds+rev)] */
    int32_t* status;       /* device int32, caller-zeroed, sticky, may be NULL: bit 0 = an edge_attr outside
                            * [0, batch.num_relations) was seen on the relation-dedupe path (the edge is scored with a
                            * clamped relation row; the reference raises IndexError at the embedding gather,
                            * src/data/components/embedding_store.py:139-150) */
    void* saved;           /* training only, may be NULL: evi_retriever_saved_bytes() of device memory the forward fills
                            * with its per-edge intermediates (state_net.0's operand and product rows); handing the same
                            * buffer to evi_retriever_backward saves it the per-edge forward recomputation */
    size_t saved_bytes;
} EviRetrieverOutput;

/* Retriever._forward_impl (src/models/components/retriever.py:195-289).  Evaluation: batch.dropout_p = 0 and
 * batch.edge_bias = NULL (dropout is the identity, the hide-and-seek bias is off: apply_in_eval: false,
 * configs/model/retriever_module.yaml:25).  Training: batch.dropout_p / dropout_seed switch on nn.Dropout of state_net,
 * batch.edge_bias carries the hide-and-seek penalty, output.saved keeps the per-edge intermediates for
 * evi_retriever_backward.  direction_mode: 0 bidirectional, 1 forward, 2 backward.
 * Dense contractions run on the split-bf16 GEMM (evi_gemm_nt_bf16x3) unless the environment
 * variable EVI_SCORER_GEMM=f32 selects the exact f32-MFMA GEMM; batch.matmul_precision = 1 opts a training run into
 * single-product bf16 (see the field).
 * Forward-only calls (no output.saved) multiply state_net.0's relation-context block once per distinct (relation, graph)
 * pair of the batch instead of once per edge — same values, bit for bit (tests/test_retriever_gpu.py); it needs
 * batch.num_relations <= E and num_graphs * num_relations <= 2^22, else — or with EVI_SCORER_PAIRS=0 — one row per edge. */
/* C [M, N] (+)= A^T B for A [K, M], B [K, N] f32 row-major (row strides lda, ldb) and K long: the weight-gradient product of
 * a Linear layer over K rows (autograd's `grad_out.t() @ input` behind every nn.Linear of src/models/components/retriever.py).
 * Split-bf16 arithmetic like evi_gemm_nt_bf16x3, split-K with an ordered reduction (deterministic, no float atomics). */
size_t evi_gemm_tn_bf16x3_workspace_bytes(int M, int N);
int evi_gemm_tn_bf16x3(const float* A, int64_t lda, int M, const float* B, int64_t ldb, int N, int64_t K, float* C,
                       int accumulate, void* workspace, size_t workspace_bytes, void* stream);

size_t evi_retriever_prepare_bytes(int D, int H, int dde_rounds, int dde_reverse_rounds);
int evi_retriever_prepare(const EviRetrieverWeights* weights, void* prepared, size_t prepared_bytes, void* stream);
size_t evi_retriever_forward_workspace_bytes(int64_t N, int64_t E, int B, int D, int H, int dde_rounds,
                                             int dde_reverse_rounds, int64_t num_relations);
int evi_retriever_forward(const EviRetrieverWeights* weights, const EviRetrieverBatch* batch,
                          int direction_mode, const EviRetrieverOutput* out, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Backward of evi_retriever_forward: the gradient of a scalar loss with respect to every parameter,
 * given dL/dlogits [E] — the autograd of Retriever._forward_impl (src/models/components/retriever.py:195-289, 403-507) in the
 * arithmetic of the forward above (eval-mode graph: no dropout, no hide-and-seek bias).  `grads` has the weights struct's
 * layout with every pointer a caller-owned OUTPUT buffer of the parameter's shape (overwritten; `prepared` ignored).
 * Without `saved` the forward is recomputed inside; with it the kept rows are replayed (see EviRetrieverOutput.saved and
 * evi_retriever_saved_bytes_full).  When batch.num_relations is given (relation rows
 * de-duplicated), rel_perm [E] lists the edge ids grouped by relation id (stable order) and rel_ptr [R+1] the group bounds.
 * Reductions run in a fixed order (f64 segment sums along the CSR / the relation grouping; no float atomics).  Gradients
 * with respect to the batch's embeddings are not produced (the reference's tables are frozen inputs). */
size_t evi_retriever_backward_workspace_bytes(int64_t N, int64_t E, int B, int D, int H, int dde_rounds,
                                              int dde_reverse_rounds, int64_t num_relations);
size_t evi_retriever_saved_bytes(int64_t E, int D, int H, int direction_mode);
/* The same plus room for the forward's NODE-level results (projected nodes / questions / relations, structure features, both CSR
 * halves, node_repr Wc^T): a `saved` buffer of at least this size makes evi_retriever_forward keep them too and lets
 * evi_retriever_backward skip the node-level forward as well (three large GEMMs, CSR, DDE).  A buffer of only
 * evi_retriever_saved_bytes() keeps the per-edge rows alone. */
size_t evi_retriever_saved_bytes_full(int64_t N, int64_t E, int B, int D, int H, int dde_rounds, int dde_reverse_rounds,
                                      int64_t num_relations, int direction_mode);
int evi_retriever_backward(const EviRetrieverWeights* weights, const EviRetrieverBatch* batch, int direction_mode,
                           const float* dlogits, const EviRetrieverWeights* grads, const int64_t* rel_perm,
                           const int64_t* rel_ptr, void* workspace, size_t workspace_bytes, const void* saved,
                           size_t saved_bytes, void* stream);

/* ---- training path: optimiser step over flat buffers (csrc/optim.hip) ---------------------------------------------------
 * evi_grad_norm: norm_out[0] = |scale| * ||g||_2 (f64 accumulation in a fixed order) — the total norm Lightning's
 * gradient_clip_val feeds to torch.nn.utils.clip_grad_norm_ (configs/trainer/default.yaml:20).
 * evi_adamw_step: torch.optim.AdamW's update (src/utils/optimization.py:20-35; configs/model/retriever_module.yaml:37-40) on
 * g * grad_scale * min(1, max_norm / (grad_norm[0] + 1e-6)); grad_norm NULL = no clipping.  step counts from 1. */
size_t evi_grad_norm_workspace_bytes(int64_t n);
int evi_grad_norm(const float* g, int64_t n, float scale, float* norm_out, void* workspace, size_t workspace_bytes, void* stream);
int evi_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t step, float grad_scale, const float* grad_norm, float max_norm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVI_HIP_H_ */
