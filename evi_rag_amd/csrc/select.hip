// Exact top-k selection kernels that are not tied to the cosine scan:
//   evi_topk_merge    — merge of per-shard top-k lists after the RCCL all-gather
//                       (reference gather: src/callbacks/retriever_topk_edge_writer.py:450-462)
//   evi_segment_topk  — per-graph top-k over edge scores
//                       (reference: src/metrics/reachability.py:146-147,
//                        src/metrics/retriever_metrics.py:141-145,
//                        src/callbacks/retriever_topk_edge_writer.py:299-302,
//                        src/data/components/g_agent_builder.py:640-652)
// Both are one workgroup per list; order is (score desc, position asc). HBM-bound on the list
// read (8 B/entry for merge, 4 B/entry for segment top-k); the sort runs in LDS.
#include "common.hpp"

namespace evi {

// scores/ids [P, Q, k]; block = query.  Slot order (shard asc, rank asc) equals id order among
// equal scores when shards are passed in ascending row-id order, so the slot number is the
// tie-break and ids may be full 64-bit.
// shard p's scores start at scores + p * score_stride (floats), its ids at ids + p * id_stride (i64):
// contiguous [P, Q, k] arrays have strides Q*k; the packed all-gather layout ([scores | ids] per
// rank) has the per-rank record size.
__global__ __launch_bounds__(kSelectThreads) void k_topk_merge(
    const float* __restrict__ scores, const int64_t* __restrict__ ids, int64_t score_stride, int64_t id_stride,
    int P, int Q, int k, float* __restrict__ out_score, int64_t* __restrict__ out_index) {
    __shared__ SelectShared sh;
    const int qi = blockIdx.x;
    const int64_t cnt = (int64_t)P * k;
    auto slot_of = [&](int64_t i, int& p) -> int64_t {
        p = (int)(i / k);
        return (int64_t)qi * k + (i % k);
    };
    auto load = [&](int64_t i) -> uint64_t {
        int p;
        const int64_t s = slot_of(i, p);
        if (ids[p * id_stride + s] < 0) return 0ull;
        return make_key(scores[p * score_stride + s], (uint32_t)i);
    };
    const int m = block_topk(sh, load, cnt, k);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint64_t key = i < m ? sh.keys[i] : 0ull;
        if (key != 0ull) {
            int p;
            const int64_t s = slot_of((int64_t)key_index(key), p);
            out_score[(int64_t)qi * k + i] = scores[p * score_stride + s];
            out_index[(int64_t)qi * k + i] = ids[p * id_stride + s];
        } else {
            out_score[(int64_t)qi * k + i] = -INFINITY;
            out_index[(int64_t)qi * k + i] = -1;
        }
    }
}

__global__ __launch_bounds__(kSelectThreads) void k_segment_topk(
    const float* __restrict__ scores, const int64_t* __restrict__ edge_ptr, int k,
    int32_t* __restrict__ out_index, float* __restrict__ out_score, int32_t* __restrict__ out_count) {
    __shared__ SelectShared sh;
    const int gid = blockIdx.x;
    const int64_t begin = edge_ptr[gid], end = edge_ptr[gid + 1];
    const int64_t cnt = end > begin ? end - begin : 0;
    const float* s = scores + begin;
    auto load = [&](int64_t i) -> uint64_t { return make_key(s[i], (uint32_t)i); };
    const int m = block_topk(sh, load, cnt, k);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        if (i < m) {
            const uint64_t key = sh.keys[i];
            out_index[(int64_t)gid * k + i] = (int32_t)key_index(key);
            if (out_score) out_score[(int64_t)gid * k + i] = s[key_index(key)];
        } else {
            out_index[(int64_t)gid * k + i] = -1;
            if (out_score) out_score[(int64_t)gid * k + i] = -INFINITY;
        }
    }
    if (out_count && threadIdx.x == 0) out_count[gid] = m;
}

}  // namespace evi

using namespace evi;

extern "C" int evi_topk_merge(const float* scores, const int64_t* ids, int P, int Q, int k,
                              float* out_score, int64_t* out_index, void* stream) {
    EVI_REQUIRE(P >= 1 && Q >= 0, "evi_topk_merge: need P >= 1 and Q >= 0, got P=%d Q=%d", P, Q);
    EVI_REQUIRE(k >= 1 && k <= EVI_TOPK_MAX_K, "evi_topk_merge: k must be in [1, %d], got %d",
                EVI_TOPK_MAX_K, k);
    EVI_REQUIRE((int64_t)P * k <= kSortCap, "evi_topk_merge: P*k = %lld exceeds %d",
                (long long)P * k, kSortCap);
    if (Q == 0) return EVI_OK;
    EVI_REQUIRE(scores && ids && out_score && out_index, "evi_topk_merge: null pointer");
    hipLaunchKernelGGL(k_topk_merge, dim3(Q), dim3(kSelectThreads), 0, reinterpret_cast<hipStream_t>(stream), scores,
                       ids, (int64_t)Q * k, (int64_t)Q * k, P, Q, k, out_score, out_index);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" size_t evi_topk_packed_bytes(int Q, int k) {
    if (Q <= 0 || k <= 0) return 0;
    return align_up((size_t)Q * k * sizeof(float), 8) + (size_t)Q * k * sizeof(int64_t);
}

extern "C" int evi_topk_merge_packed(const void* packed, int P, int Q, int k, float* out_score, int64_t* out_index,
                                     void* stream) {
    EVI_REQUIRE(P >= 1 && Q >= 0, "evi_topk_merge_packed: need P >= 1 and Q >= 0, got P=%d Q=%d", P, Q);
    EVI_REQUIRE(k >= 1 && k <= EVI_TOPK_MAX_K, "evi_topk_merge_packed: k must be in [1, %d], got %d", EVI_TOPK_MAX_K, k);
    EVI_REQUIRE((int64_t)P * k <= kSortCap, "evi_topk_merge_packed: P*k = %lld exceeds %d", (long long)P * k, kSortCap);
    if (Q == 0) return EVI_OK;
    EVI_REQUIRE(packed && out_score && out_index, "evi_topk_merge_packed: null pointer");
    const size_t rec = evi_topk_packed_bytes(Q, k);
    const size_t ids_off = align_up((size_t)Q * k * sizeof(float), 8);
    const char* base = static_cast<const char*>(packed);
    hipLaunchKernelGGL(k_topk_merge, dim3(Q), dim3(kSelectThreads), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float*>(base), reinterpret_cast<const int64_t*>(base + ids_off),
                       (int64_t)(rec / sizeof(float)), (int64_t)(rec / sizeof(int64_t)), P, Q, k, out_score, out_index);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_segment_topk(const float* scores, const int64_t* edge_ptr, int B, int k,
                                int32_t* out_index, float* out_score, int32_t* out_count, void* stream) {
    EVI_REQUIRE(B >= 0, "evi_segment_topk: B must be >= 0, got %d", B);
    EVI_REQUIRE(k >= 1 && k <= EVI_TOPK_MAX_K, "evi_segment_topk: k must be in [1, %d], got %d",
                EVI_TOPK_MAX_K, k);
    if (B == 0) return EVI_OK;
    EVI_REQUIRE(edge_ptr && out_index, "evi_segment_topk: null pointer");
    hipLaunchKernelGGL(k_segment_topk, dim3(B), dim3(kSelectThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), scores, edge_ptr, k, out_index, out_score,
                       out_count);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
