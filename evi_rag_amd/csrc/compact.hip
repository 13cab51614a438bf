// Segmented de-duplication and re-indexing of integer keys (G5 build_graph, f2 g_agent builder).
//
//   evi_first_occurrence   for every entry of a segment, the position of the first entry with the
//                          same W-word key: the dict / set bookkeeping of
//                          build_graph (scripts/build_retrieval_pipeline.py:1465-1497: node_index,
//                          edge_key_to_indices) and of GAgentBuilder._build_and_add_sample
//                          (src/data/components/g_agent_builder.py:338-354: triple_to_agg)
//   evi_first_seen_rank    ids in first-seen order + the list of first occurrences
//                          (local_index(), :1470-1476; list(triple_to_agg.keys()), :356)
//   evi_segment_sort_rank  ascending rank of a segment's keys (torch.sort(unique), :368; node_map :371)
//   evi_group_max_f32      per-group maximum (score / label aggregation, :353-354)
//
// One workgroup per segment (a sample's triples, a graph's environment edges).  Integer work on
// lists of 10^2..10^5 entries that live in L2: bound by atomic / gather latency, not by HBM
// (bytes per entry: W*8 key reads per probe + 4 table + 4 output).
#include "common.hpp"

namespace evi {

constexpr int kCompactThreads = 1024;

__device__ inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// Open-addressing table of segment-local positions, 2 len + 1 slots per segment at table[2 seg_ptr[s] + s].
// A slot is claimed by CAS and then only ever lowered to a smaller position WITH THE SAME KEY, so every
// entry of a key class stops at the same slot whatever the interleaving, and the slot ends at the minimum.
template <int W>
__global__ __launch_bounds__(kCompactThreads) void k_first_occurrence(
    const int64_t* __restrict__ keys, const int64_t* __restrict__ seg_ptr, const uint8_t* __restrict__ drop,
    int32_t* __restrict__ table_all, int32_t* __restrict__ out_first) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = seg_ptr[s];
    const int len = (int)(seg_ptr[s + 1] - p0);
    if (len <= 0) return;
    const uint32_t cap = 2u * (uint32_t)len + 1u;
    int32_t* table = table_all + 2 * p0 + s;
    for (uint32_t i = tid; i < cap; i += kCompactThreads) table[i] = -1;
    __syncthreads();
    const int64_t* k = keys + p0 * W;
    for (int p = tid; p < len; p += kCompactThreads) {
        if (drop && drop[p0 + p]) {
            out_first[p0 + p] = -1;
            continue;
        }
        int64_t mine[W];
        uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            mine[w] = k[(int64_t)p * W + w];
            h = mix64(h ^ (uint64_t)mine[w]);
        }
        uint32_t slot = (uint32_t)(h % cap);
        for (uint32_t probes = 0; probes < cap; ++probes) {  // the table is never full: this always breaks
            int32_t cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur < 0) {
                cur = atomicCAS(&table[slot], -1, p);
                if (cur < 0) break;  // claimed an empty slot
            }
            bool same = true;
#pragma unroll
            for (int w = 0; w < W; ++w) same = same && (k[(int64_t)cur * W + w] == mine[w]);
            if (same) {
                atomicMin(&table[slot], p);
                break;
            }
            slot = slot + 1 == cap ? 0 : slot + 1;
        }
        out_first[p0 + p] = (int32_t)slot;  // resolved below, once every insert has landed
    }
    __threadfence_block();
    __syncthreads();
    for (int p = tid; p < len; p += kCompactThreads) {
        const int32_t slot = out_first[p0 + p];
        if (slot >= 0) out_first[p0 + p] = table[slot];
    }
}

// Exclusive prefix count of `flag` over the workgroup (1024 threads = 16 waves); total in `total`.
__device__ inline int block_count_before(bool flag, int* wave_tot, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();  // wave_tot may still be read from the previous call
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kCompactThreads / 64; ++w) {
        const int c = wave_tot[w];
        if (w < wave) base += c;
        tot += c;
    }
    total = tot;
    return base + before;
}

// first[] from k_first_occurrence.  An entry is a "first" iff first[p] == p and p < limit (entries at
// or beyond limit are look-ups: they take the rank of the entry they match, or -1).
__global__ __launch_bounds__(kCompactThreads) void k_first_seen_rank(
    const int32_t* __restrict__ first, const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ limit,
    int32_t* __restrict__ out_rank, int32_t* __restrict__ out_count, int32_t* __restrict__ out_uniq_pos) {
    __shared__ int wave_tot[kCompactThreads / 64];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = seg_ptr[s];
    const int len = (int)(seg_ptr[s + 1] - p0);
    const int lim = limit ? (int)(limit[s] < len ? limit[s] : len) : len;
    int base = 0;
    for (int c0 = 0; c0 < len; c0 += kCompactThreads) {
        const int p = c0 + tid;
        const bool flag = p < lim && first[p0 + p] == p;
        int total;
        const int before = block_count_before(flag, wave_tot, total);
        if (flag) {
            out_rank[p0 + p] = base + before;
            if (out_uniq_pos) out_uniq_pos[p0 + base + before] = p;
        }
        base += total;
    }
    if (tid == 0) out_count[s] = base;
    __threadfence_block();
    __syncthreads();
    for (int p = tid; p < len; p += kCompactThreads) {
        const int f = first[p0 + p];
        if (p < lim && f == p) continue;
        out_rank[p0 + p] = (f >= 0 && f < lim) ? out_rank[p0 + f] : -1;
    }
}

// rank[i] = #{ j : key[j] < key[i], or key[j] == key[i] and j < i } over the first seg_len[s] entries of
// the segment (a stable ascending sort position), by all-pairs counting against LDS tiles: exact for
// any length, n^2 / 1024 compares per thread — the lists here are a few thousand node ids.
__global__ __launch_bounds__(kCompactThreads) void k_segment_sort_rank(
    const int64_t* __restrict__ keys, const int64_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_len,
    int32_t* __restrict__ out_rank, int64_t* __restrict__ out_sorted) {
    __shared__ int64_t tile[kCompactThreads];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = seg_ptr[s];
    int len = (int)(seg_ptr[s + 1] - p0);
    if (seg_len && seg_len[s] < len) len = seg_len[s];
    for (int i0 = 0; i0 < len; i0 += kCompactThreads) {
        const int i = i0 + tid;
        const int64_t mine = i < len ? keys[p0 + i] : 0;
        int rank = 0;
        for (int j0 = 0; j0 < len; j0 += kCompactThreads) {
            __syncthreads();
            tile[tid] = j0 + tid < len ? keys[p0 + j0 + tid] : 0;
            __syncthreads();
            const int m = len - j0 < kCompactThreads ? len - j0 : kCompactThreads;
            for (int j = 0; j < m; ++j) {
                const int64_t other = tile[j];
                rank += (other < mine || (other == mine && j0 + j < i)) ? 1 : 0;
            }
        }
        if (i < len) {
            out_rank[p0 + i] = rank;
            if (out_sorted) out_sorted[p0 + rank] = mine;
        }
    }
}

// out[group[i]] = max(out[group[i]], values[i]); out is pre-filled by the caller (-inf).  The maximum is
// order-free, so the atomics cannot change a result.  NaN values are ignored.
__global__ void k_group_max_f32(const float* __restrict__ values, const int32_t* __restrict__ group, int64_t T,
                                float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t g = group[i];
        const float v = values[i];
        if (g < 0 || v != v) continue;
        if (v >= 0.f)
            atomicMax(reinterpret_cast<int*>(out + g), __float_as_int(v));
        else
            atomicMin(reinterpret_cast<unsigned int*>(out + g), __float_as_uint(v));
    }
}

}  // namespace evi

using namespace evi;

extern "C" size_t evi_first_occurrence_workspace_bytes(int64_t T, int S) {
    if (T < 0 || S < 0) return 0;
    return (size_t)(2 * T + S) * sizeof(int32_t);
}

extern "C" int evi_first_occurrence(const int64_t* keys, int W, int64_t T, const int64_t* seg_ptr, int S,
                                    const uint8_t* drop, int32_t* out_first, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    EVI_REQUIRE(T >= 0 && S >= 0, "evi_first_occurrence: need T >= 0 and S >= 0, got T=%lld S=%d", (long long)T, S);
    if (W != 1 && W != 2 && W != 3)
        return fail(EVI_ERR_UNSUPPORTED, "evi_first_occurrence: keys of 1, 2 or 3 words are supported, got W=%d", W);
    if (T == 0 || S == 0) return EVI_OK;
    EVI_REQUIRE(keys && seg_ptr && out_first && workspace, "evi_first_occurrence: null pointer");
    if (workspace_bytes < evi_first_occurrence_workspace_bytes(T, S))
        return fail(EVI_ERR_NOMEM, "evi_first_occurrence: workspace %zu B < %zu B", workspace_bytes,
                    evi_first_occurrence_workspace_bytes(T, S));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int32_t* table = static_cast<int32_t*>(workspace);
    if (W == 1)
        hipLaunchKernelGGL(k_first_occurrence<1>, dim3(S), dim3(kCompactThreads), 0, st, keys, seg_ptr, drop, table, out_first);
    else if (W == 2)
        hipLaunchKernelGGL(k_first_occurrence<2>, dim3(S), dim3(kCompactThreads), 0, st, keys, seg_ptr, drop, table, out_first);
    else
        hipLaunchKernelGGL(k_first_occurrence<3>, dim3(S), dim3(kCompactThreads), 0, st, keys, seg_ptr, drop, table, out_first);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_first_seen_rank(const int32_t* first, int64_t T, const int64_t* seg_ptr, int S, const int64_t* limit,
                                   int32_t* out_rank, int32_t* out_count, int32_t* out_uniq_pos, void* stream) {
    EVI_REQUIRE(T >= 0 && S >= 0, "evi_first_seen_rank: need T >= 0 and S >= 0, got T=%lld S=%d", (long long)T, S);
    if (S == 0) return EVI_OK;
    EVI_REQUIRE(seg_ptr && out_count && (T == 0 || (first && out_rank)), "evi_first_seen_rank: null pointer");
    hipLaunchKernelGGL(k_first_seen_rank, dim3(S), dim3(kCompactThreads), 0, reinterpret_cast<hipStream_t>(stream), first,
                       seg_ptr, limit, out_rank, out_count, out_uniq_pos);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_segment_sort_rank(const int64_t* keys, int64_t T, const int64_t* seg_ptr, const int32_t* seg_len,
                                     int S, int32_t* out_rank, int64_t* out_sorted, void* stream) {
    EVI_REQUIRE(T >= 0 && S >= 0, "evi_segment_sort_rank: need T >= 0 and S >= 0, got T=%lld S=%d", (long long)T, S);
    if (T == 0 || S == 0) return EVI_OK;
    EVI_REQUIRE(keys && seg_ptr && out_rank, "evi_segment_sort_rank: null pointer");
    hipLaunchKernelGGL(k_segment_sort_rank, dim3(S), dim3(kCompactThreads), 0, reinterpret_cast<hipStream_t>(stream), keys,
                       seg_ptr, seg_len, out_rank, out_sorted);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_group_max_f32(const float* values, const int32_t* group, int64_t T, float* out, void* stream) {
    EVI_REQUIRE(T >= 0, "evi_group_max_f32: T must be >= 0, got %lld", (long long)T);
    if (T == 0) return EVI_OK;
    EVI_REQUIRE(values && group && out, "evi_group_max_f32: null pointer");
    int64_t blocks = (T + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_group_max_f32, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), values, group,
                       T, out);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
