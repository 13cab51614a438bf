// Embedding feed (D1) and per-graph class statistics (T5).
//
//   evi_gather_rows        index_select of embedding rows, on the device, from HBM-resident tables.
//                          Replaces GlobalEmbeddingStore.get_entity_embeddings/get_relation_embeddings
//                          (CPU index_select into a pinned buffer + H2D copy),
//                          src/data/components/embedding_store.py:101-150, called at
//                          src/data/components/loader.py:60-66,171-185.
//   evi_graph_class_stats  per graph: positive / negative counts and the sums of sigmoid(score) per
//                          class.  Building block of BridgeProbQuality / BridgePositiveCoverage,
//                          src/metrics/retriever_metrics.py:270-327, 400-476.
//
//   evi_segment_offsets / evi_gather_segments
//                          batch collation from a split that is resident in HBM as flat arrays
//                          (all samples concatenated, one pointer array per field): a batch is B
//                          segment copies per field, with the per-sample increment PyG's collate adds
//                          to index fields (GRetrievalData.__inc__, src/data/g_retrieval_dataset.py:29-37;
//                          Collater at src/data/components/loader.py:36-44).  Replaces LMDB get +
//                          unpickle + Collater in 16 worker processes.
//
// Both are HBM-bound: the gather moves n*D*4 bytes in and out (one wave per row, 16-byte lanes);
// the stats read E*(4+1) bytes.  Sums are formed in f64 through a fixed LDS tree: deterministic.
#include "common.hpp"

namespace evi {

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ table, int64_t T, int D,
                                                     const int64_t* __restrict__ ids, int64_t n,
                                                     float* __restrict__ out, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += nwaves) {
        const int64_t id = ids[i];
        if (id < 0 || id >= T) {  // torch.index_select raises; report and write zeros
            if (lane == 0) atomicOr(status, 1);
            for (int d = lane; d < D; d += 64) out[i * D + d] = 0.f;
            continue;
        }
        const float* src = table + id * (int64_t)D;
        float* dst = out + i * (int64_t)D;
        if ((D & 3) == 0) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(dst);
            for (int c = lane; c < (D >> 2); c += 64) d4[c] = s4[c];
        } else {
            for (int d = lane; d < D; d += 64) dst[d] = src[d];
        }
    }
}

// out[g] = {positives, negatives, sum sigmoid(score | positive), sum sigmoid(score | negative)}
__global__ __launch_bounds__(256) void k_graph_class_stats(const float* __restrict__ scores,
                                                           const uint8_t* __restrict__ target,
                                                           const int64_t* __restrict__ edge_ptr,
                                                           double* __restrict__ out) {
    __shared__ double red[4][256];
    const int g = blockIdx.x, tid = threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t e = edge_ptr[g] + tid; e < edge_ptr[g + 1]; e += 256) {
        const float p = 1.0f / (1.0f + expf(-scores[e]));  // torch.sigmoid in f32
        if (target[e]) {
            v[0] += 1.0;
            v[2] += (double)p;
        } else {
            v[1] += 1.0;
            v[3] += (double)p;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[c][tid] = v[c];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[c][tid] += red[c][tid + off];
        __syncthreads();
    }
    if (tid < 4) out[(int64_t)g * 4 + tid] = red[tid][0];
}

// out_ptr[0] = 0, out_ptr[b+1] = sum_{i<=b} items of sample ids[b]; one workgroup, B is a batch size.
__global__ __launch_bounds__(1024) void k_segment_offsets(const int64_t* __restrict__ src_ptr, const int64_t* __restrict__ ids,
                                                          int B, int64_t num_samples, int64_t* __restrict__ out_ptr,
                                                          int32_t* __restrict__ status) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    // thread t owns the contiguous slice [t*per, (t+1)*per)
    const int per = (B + 1023) / 1024;
    int64_t local = 0;
    for (int i = tid * per; i < (tid + 1) * per && i < B; ++i) {
        const int64_t s = ids[i];
        if (s < 0 || s >= num_samples) {
            atomicOr(status, 1);
            continue;
        }
        local += src_ptr[s + 1] - src_ptr[s];
    }
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t add = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    int64_t run = part[tid] - local;  // exclusive prefix of this thread's slice
    if (tid == 0) out_ptr[0] = 0;
    for (int i = tid * per; i < (tid + 1) * per && i < B; ++i) {
        const int64_t s = ids[i];
        if (s >= 0 && s < num_samples) run += src_ptr[s + 1] - src_ptr[s];
        out_ptr[i + 1] = run;
    }
}

// grid = (chunks, B): sample ids[b]'s items [src_ptr[s], src_ptr[s+1]) x row_words 8-byte (WORD 8) or
// 4-byte (WORD 4) words are copied to out + out_ptr[b] * row_words; 8-byte words get add[b] added
// (the collate increment of index fields).  Coalesced: consecutive lanes move consecutive words.
template <typename Word>
__global__ __launch_bounds__(256) void k_gather_segments(const Word* __restrict__ src, int64_t row_words,
                                                         const int64_t* __restrict__ src_ptr, const int64_t* __restrict__ ids,
                                                         int64_t num_samples, const int64_t* __restrict__ out_ptr,
                                                         const int64_t* __restrict__ add, Word* __restrict__ out) {
    const int b = blockIdx.y;
    const int64_t s = ids[b];
    if (s < 0 || s >= num_samples) return;
    const int64_t n = (src_ptr[s + 1] - src_ptr[s]) * row_words;
    const Word* from = src + src_ptr[s] * row_words;
    Word* to = out + out_ptr[b] * row_words;
    const Word inc = add ? (Word)add[b] : (Word)0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) to[i] = from[i] + inc;
}

}  // namespace evi

using namespace evi;

extern "C" int evi_segment_offsets(const int64_t* src_ptr, int64_t num_samples, const int64_t* ids, int B,
                                   int64_t* out_ptr, int32_t* status, void* stream) {
    EVI_REQUIRE(B >= 0 && num_samples >= 0, "evi_segment_offsets: need B >= 0 and num_samples >= 0");
    EVI_REQUIRE(out_ptr && status && (B == 0 || (src_ptr && ids)), "evi_segment_offsets: null pointer");
    hipLaunchKernelGGL(k_segment_offsets, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), src_ptr, ids, B,
                       num_samples, out_ptr, status);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_gather_segments(const void* src, int word_bytes, int64_t row_words, const int64_t* src_ptr,
                                   int64_t num_samples, const int64_t* ids, int B, const int64_t* out_ptr,
                                   const int64_t* add, void* out, void* stream) {
    EVI_REQUIRE(B >= 0 && row_words >= 1 && num_samples >= 0, "evi_gather_segments: bad sizes");
    EVI_REQUIRE(word_bytes == 4 || word_bytes == 8, "evi_gather_segments: word_bytes must be 4 or 8, got %d", word_bytes);
    EVI_REQUIRE(word_bytes == 8 || !add, "evi_gather_segments: increments apply to 8-byte (int64) fields only");
    if (B == 0) return EVI_OK;
    EVI_REQUIRE(src_ptr && ids && out_ptr, "evi_gather_segments: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(32, B);
    if (word_bytes == 8)
        hipLaunchKernelGGL(k_gather_segments<int64_t>, grid, dim3(256), 0, st, static_cast<const int64_t*>(src), row_words,
                           src_ptr, ids, num_samples, out_ptr, add, static_cast<int64_t*>(out));
    else
        hipLaunchKernelGGL(k_gather_segments<int32_t>, grid, dim3(256), 0, st, static_cast<const int32_t*>(src), row_words,
                           src_ptr, ids, num_samples, out_ptr, static_cast<const int64_t*>(nullptr), static_cast<int32_t*>(out));
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}


extern "C" int evi_gather_rows(const float* table, int64_t num_rows, int D, const int64_t* ids, int64_t n, float* out,
                               int32_t* status, void* stream) {
    EVI_REQUIRE(num_rows >= 0 && D >= 0 && n >= 0, "evi_gather_rows: bad sizes");
    EVI_REQUIRE(status, "evi_gather_rows: null status");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n == 0 || D == 0) return EVI_OK;
    EVI_REQUIRE(ids && out && (table || num_rows == 0), "evi_gather_rows: null pointer");
    int64_t blocks = (n + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)blocks), dim3(256), 0, st, table, num_rows, D, ids, n, out, status);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_graph_class_stats(const float* scores, const uint8_t* target, const int64_t* edge_ptr, int B,
                                     double* out, void* stream) {
    EVI_REQUIRE(B >= 0, "evi_graph_class_stats: B must be >= 0");
    if (B == 0) return EVI_OK;
    EVI_REQUIRE(edge_ptr && out, "evi_graph_class_stats: null pointer");
    hipLaunchKernelGGL(k_graph_class_stats, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), scores, target,
                       edge_ptr, out);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
