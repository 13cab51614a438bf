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

}  // namespace evi

using namespace evi;

extern "C" int evi_gather_rows(const float* table, int64_t num_rows, int D, const int64_t* ids, int64_t n, float* out,
                               int32_t* status, void* stream) {
    EVI_REQUIRE(num_rows >= 0 && D >= 0 && n >= 0, "evi_gather_rows: bad sizes");
    EVI_REQUIRE(status, "evi_gather_rows: null status");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    EVI_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
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
