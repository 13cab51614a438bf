// Seed expansion and score post-processing on the per-graph CSR (G8, G9, G10).
//
//   evi_node_softmax_logit  GAgentBuilder._node_softmax_logit, src/data/components/g_agent_builder.py:595-626
//   evi_select_start_edges  GAgentBuilder._select_start_edges (undirected one-hop seed expansion),
//                           src/data/components/g_agent_builder.py:655-724
//   evi_seed_onehop_stats   scripts/seed_onehop_stats.py:96-117
//
// All three are gathers over edge lists / CSR rows: HBM/L2-latency-bound integer and byte work
// (E * 8..20 bytes per pass).  Float reductions that the reference performs with scatter ops are
// done order-free here: maxima through ordered-integer atomics (exact), sums of exponentials in
// f64 (rounded once to f32), so results do not depend on the order edges arrive in.
#include "common.hpp"

namespace evi {

constexpr float kProbEps = 1e-6f;

// ---- G9 ---------------------------------------------------------------------------------------------
__global__ void k_softmax_init(uint32_t* mx_h, uint32_t* mx_t, double* sm_h, double* sm_t, int64_t N) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    mx_h[v] = mx_t[v] = float_to_ordered(-INFINITY);
    sm_h[v] = sm_t[v] = 0.0;
}
__global__ void k_softmax_max(const float* __restrict__ s, const int64_t* __restrict__ ei, int64_t E,
                              uint32_t* mx_h, uint32_t* mx_t) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint32_t k = float_to_ordered(s[e]);
    atomicMax(&mx_h[ei[e]], k);
    atomicMax(&mx_t[ei[E + e]], k);
}
__global__ void k_softmax_sum(const float* __restrict__ s, const int64_t* __restrict__ ei, int64_t E,
                              const uint32_t* __restrict__ mx_h, const uint32_t* __restrict__ mx_t, double* sm_h,
                              double* sm_t) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t h = ei[e], t = ei[E + e];
    atomicAdd(&sm_h[h], (double)expf(s[e] - ordered_to_float(mx_h[h])));
    atomicAdd(&sm_t[t], (double)expf(s[e] - ordered_to_float(mx_t[t])));
}
__global__ void k_softmax_logit(const float* __restrict__ s, const int64_t* __restrict__ ei, int64_t E,
                                const uint32_t* __restrict__ mx_h, const uint32_t* __restrict__ mx_t,
                                const double* __restrict__ sm_h, const double* __restrict__ sm_t,
                                float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t h = ei[e], t = ei[E + e];
    const float ph = expf(s[e] - ordered_to_float(mx_h[h])) / fmaxf((float)sm_h[h], kProbEps);
    const float pt = expf(s[e] - ordered_to_float(mx_t[t])) / fmaxf((float)sm_t[t], kProbEps);
    float p = (ph + pt) * 0.5f;
    p = fminf(fmaxf(p, kProbEps), 1.0f - kProbEps);
    out[e] = logf(p) - log1pf(-p);
}

// ---- G8 ---------------------------------------------------------------------------------------------
// one launch zeroes the E-byte mask and the status word (two hipMemsetAsync cost two ~4.5 us fill kernels)
__global__ void k_zero_mask(uint8_t* __restrict__ mask, int64_t E, int32_t* __restrict__ status) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i == 0) *status = 0;
    if (i + 16 <= E && (reinterpret_cast<uintptr_t>(mask) & 15) == 0) {
        *reinterpret_cast<uint4*>(mask + i) = make_uint4(0, 0, 0, 0);
    } else {
        for (int64_t j = i; j < E && j < i + 16; ++j) mask[j] = 0;
    }
}

// One workgroup per (seed entry).  Incident entries of seed s: its out-row (s is the head: "heads
// block") then its in-row (s is the tail: "tails block").  The reference ranks them by a stable
// descending score sort of [heads block ; tails block] (each block in ascending edge id), so equal
// scores keep (block, edge id) order: key = (score desc, block asc, edge id asc).
__global__ __launch_bounds__(kSelectThreads) void k_select_start_edges(
    const float* __restrict__ scores, const int64_t* __restrict__ seeds, int64_t num_seeds,
    const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_eid, const int32_t* __restrict__ out_ptr,
    const int32_t* __restrict__ out_eid, int64_t N, float keep_ratio, int min_edges, int max_edges /* <0: none */,
    uint8_t* __restrict__ mask, int32_t* __restrict__ status) {
    __shared__ SelectShared sh;
    const int64_t si = blockIdx.x;
    if (si >= num_seeds) return;
    const int64_t v = seeds[si];
    if (v < 0 || v >= N) {
        if (threadIdx.x == 0) atomicOr(status, 1);
        return;
    }
    const int ob = out_ptr[v], oe = out_ptr[v + 1], ib = in_ptr[v], ie = in_ptr[v + 1];
    const int dout = oe - ob, din = ie - ib;
    const int deg = dout + din;
    // k_s = min(deg, min(max_edges, max(min_edges, ceil(float(deg) * ratio))))   (f32 product, :676)
    long long k = (long long)ceilf((float)deg * keep_ratio);
    if (min_edges > 0 && k < min_edges) k = min_edges;
    if (max_edges >= 0) k = max_edges == 0 ? 0 : (k < max_edges ? k : max_edges);
    if (k > deg) k = deg;
    if (k <= 0) return;
    if (k >= deg) {  // everything incident is kept: no ranking needed
        for (int i = threadIdx.x; i < deg; i += blockDim.x) mask[i < dout ? out_eid[ob + i] : in_eid[ib + i - dout]] = 1;
        return;
    }
    auto load = [&](int64_t i) -> uint64_t {
        const bool tail_block = i >= dout;
        const uint32_t e = (uint32_t)(tail_block ? in_eid[ib + (i - dout)] : out_eid[ob + i]);
        return make_key(scores[e], (tail_block ? 0x80000000u : 0u) | e);
    };
    // only the SET of the k best entries is needed: find the k-th key, keep everything at or above it.  The keys of a row
    // that fits LDS (8 192 entries: every seed but the largest hubs) are gathered ONCE — each costs two dependent loads,
    // edge id then score — and the eight radix passes read them from LDS.
    if (deg <= kSortCap) {
        for (int i = threadIdx.x; i < deg; i += blockDim.x) sh.keys[i] = load(i);
        __syncthreads();
        auto staged = [&](int64_t i) -> uint64_t { return sh.keys[i]; };
        const uint64_t kth = block_kth_largest(sh, staged, deg, k);
        for (int i = threadIdx.x; i < deg; i += blockDim.x) {
            const uint64_t key = sh.keys[i];
            if (key >= kth) mask[key_index(key) & 0x7FFFFFFFu] = 1;
        }
        return;
    }
    const uint64_t kth = block_kth_largest(sh, load, deg, k);
    for (int i = threadIdx.x; i < deg; i += blockDim.x) {
        const uint64_t key = load(i);
        if (key >= kth) mask[key_index(key) & 0x7FFFFFFFu] = 1;
    }
}

// ---- G10 --------------------------------------------------------------------------------------------
__global__ void k_seed_stats(const int64_t* __restrict__ seeds, int64_t num_seeds, const uint8_t* __restrict__ positive,
                             const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_eid,
                             const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_eid, int64_t N,
                             int32_t* __restrict__ deg, int32_t* __restrict__ pos_deg) {
    const int64_t si = blockIdx.x;
    if (si >= num_seeds) return;
    const int64_t v = seeds[si];
    if (v < 0 || v >= N) {
        if (threadIdx.x == 0) deg[si] = pos_deg[si] = -1;
        return;
    }
    const int ob = out_ptr[v], oe = out_ptr[v + 1], ib = in_ptr[v], ie = in_ptr[v + 1];
    int p = 0;
    for (int i = ob + threadIdx.x; i < oe; i += blockDim.x) p += positive[out_eid[i]] ? 1 : 0;
    for (int i = ib + threadIdx.x; i < ie; i += blockDim.x) p += positive[in_eid[i]] ? 1 : 0;
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    if (p) atomicAdd(&total, p);
    __syncthreads();
    if (threadIdx.x == 0) {
        deg[si] = (oe - ob) + (ie - ib);
        pos_deg[si] = total;
    }
}

}  // namespace evi

using namespace evi;

extern "C" size_t evi_node_softmax_logit_workspace_bytes(int64_t N) { return (size_t)(N > 0 ? N : 1) * 24 + 512; }

extern "C" int evi_node_softmax_logit(const float* edge_scores, const int64_t* edge_index, int64_t E, int64_t N,
                                      float* out_logit, void* workspace, size_t workspace_bytes, void* stream) {
    EVI_REQUIRE(E >= 0 && N >= 0, "evi_node_softmax_logit: bad sizes E=%lld N=%lld", (long long)E, (long long)N);
    if (E == 0) return EVI_OK;
    EVI_REQUIRE(edge_scores && edge_index && out_logit && workspace, "evi_node_softmax_logit: null pointer");
    if (workspace_bytes < evi_node_softmax_logit_workspace_bytes(N))
        return fail(EVI_ERR_NOMEM, "evi_node_softmax_logit: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(workspace);
    double* sm_h = reinterpret_cast<double*>(base);
    double* sm_t = sm_h + N;
    uint32_t* mx_h = reinterpret_cast<uint32_t*>(sm_t + N);
    uint32_t* mx_t = mx_h + N;
    const dim3 gn((unsigned)((N + 255) / 256)), ge((unsigned)((E + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_softmax_init, gn, blk, 0, st, mx_h, mx_t, sm_h, sm_t, N);
    hipLaunchKernelGGL(k_softmax_max, ge, blk, 0, st, edge_scores, edge_index, E, mx_h, mx_t);
    hipLaunchKernelGGL(k_softmax_sum, ge, blk, 0, st, edge_scores, edge_index, E, mx_h, mx_t, sm_h, sm_t);
    hipLaunchKernelGGL(k_softmax_logit, ge, blk, 0, st, edge_scores, edge_index, E, mx_h, mx_t, sm_h, sm_t, out_logit);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_select_start_edges(const float* edge_scores, int64_t E, const int64_t* seed_nodes, int64_t num_seeds,
                                      const int32_t* in_ptr, const int32_t* in_eid, const int32_t* out_ptr,
                                      const int32_t* out_eid, int64_t N, float start_keep_ratio, int start_min_edges,
                                      int start_max_edges, uint8_t* out_mask, int32_t* status, void* stream) {
    EVI_REQUIRE(E >= 0 && N >= 0 && num_seeds >= 0, "evi_select_start_edges: bad sizes");
    EVI_REQUIRE(out_mask || E == 0, "evi_select_start_edges: null mask");
    EVI_REQUIRE(status, "evi_select_start_edges: null status");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {
        const int64_t threads = (E + 15) / 16 > 0 ? (E + 15) / 16 : 1;
        hipLaunchKernelGGL(k_zero_mask, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, out_mask, E, status);
        EVI_LAUNCH_CHECK();
    }
    if (E == 0 || num_seeds == 0) return EVI_OK;
    EVI_REQUIRE(edge_scores && seed_nodes && in_ptr && in_eid && out_ptr && out_eid, "evi_select_start_edges: null pointer");
    hipLaunchKernelGGL(k_select_start_edges, dim3((unsigned)num_seeds), dim3(kSelectThreads), 0, st, edge_scores,
                       seed_nodes, num_seeds, in_ptr, in_eid, out_ptr, out_eid, N, start_keep_ratio, start_min_edges,
                       start_max_edges, out_mask, status);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_seed_onehop_stats(const int64_t* seed_nodes, int64_t num_seeds, const uint8_t* positive,
                                     const int32_t* in_ptr, const int32_t* in_eid, const int32_t* out_ptr,
                                     const int32_t* out_eid, int64_t N, int32_t* out_degree, int32_t* out_positive_degree,
                                     void* stream) {
    EVI_REQUIRE(num_seeds >= 0 && N >= 0, "evi_seed_onehop_stats: bad sizes");
    if (num_seeds == 0) return EVI_OK;
    EVI_REQUIRE(seed_nodes && positive && in_ptr && in_eid && out_ptr && out_eid && out_degree && out_positive_degree,
                "evi_seed_onehop_stats: null pointer");
    hipLaunchKernelGGL(k_seed_stats, dim3((unsigned)num_seeds), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       seed_nodes, num_seeds, positive, in_ptr, in_eid, out_ptr, out_eid, N, out_degree,
                       out_positive_degree);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
