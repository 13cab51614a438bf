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

/* Optional per-kernel timing for the bench's roofline leg.  While enabled, each call brackets its
 * dominant kernels with hipEvents on the call's stream (class 0: the cosine scan kernel,
 * class 1: top-k selection kernels).  evi_timing_read synchronises those events, writes the summed
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

/* Merge P per-shard top-k lists (the all-gathered outputs of evi_cosine_topk on P ranks) into
 * the global top-k, same (score desc, id asc) order; ids < 0 are padding and never win.
 *   scores [P, Q, k] f32, ids [P, Q, k] i64  ->  out_score [Q, k], out_index [Q, k].
 * Requires P * k <= 8192.  Replaces the gather at
 * src/callbacks/retriever_topk_edge_writer.py:450-462 (all_gather_object of per-sample lists). */
int evi_topk_merge(const float* scores, const int64_t* ids, int P, int Q, int k,
                   float* out_score, int64_t* out_index, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* EVI_HIP_H_ */
