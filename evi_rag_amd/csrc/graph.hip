// Per-graph structure kernels on the flat (PyG-style) batch: one workgroup per question graph.
//
//   evi_edge_batch        — compute_edge_batch (src/utils/graph_utils.py:50-104)
//   evi_qa_edge_mask      — compute_qa_edge_mask (src/utils/graph_utils.py:107-153)
//   evi_graph_csr         — in-/out-edge CSR of every graph (the adjacency of
//                           _build_undirected_adjacency, scripts/build_retrieval_pipeline.py:570-586,
//                           kept as two directed halves)
//   evi_dde_node_struct   — PEConv/DDE mean propagation + topic-major stacking
//                           (src/models/components/graph.py:13-74, retriever.py:519-553)
//
// Graphs in a batch are independent and small (N_g ~ 10^3, E_g ~ 10^3..10^5), so a graph never
// leaves its CU: rounds are separated by workgroup barriers, not kernel launches, and the per-node
// state stays in L2.  All kernels are HBM/L2-latency-bound integer and gather work; algorithmic
// bytes per DDE round = E*(4 nbr + C*4 gather) + N*(8 rowptr + C*4 write).
//
// CSR rows are filled through atomic cursors, so the order of a row's entries is not defined.
// Nothing downstream depends on it: BFS levels are order-free, and DDE sums a row in f64 before
// rounding once to f32, which makes the f32 result independent of the summation order.
#include "common.hpp"

namespace evi {

constexpr int kGraphThreads = 1024;
constexpr int kHubDegree = 32;
constexpr int kHubListCap = 4096;

// ---- edge -> graph assignment -----------------------------------------------------------------
// status bits: 1 = head outside [ptr[0], ptr[B]), 2 = head/tail in different graphs,
//              4 = edge list not grouped by graph (edge_batch decreases).
__global__ void k_edge_batch(const int64_t* __restrict__ edge_index, int64_t E,
                             const int64_t* __restrict__ node_ptr, int B, int64_t* __restrict__ edge_batch,
                             int64_t* __restrict__ edge_ptr, int32_t* __restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    auto bucket = [&](int64_t v) -> int {  // bucketize(v, ptr[1:], right=True): #{j >= 1 : ptr[j] <= v}
        int lo = 0, hi = B;                // search in ptr[1..B]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (node_ptr[mid + 1] <= v) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int64_t h = edge_index[e], t = edge_index[E + e];
    const int gh = bucket(h), gt = bucket(t);
    edge_batch[e] = gh;
    int st = 0;
    if (gh < 0 || gh >= B || h < node_ptr[0]) st |= 1;
    if (gh != gt) st |= 2;
    if (e > 0) {
        const int gp = bucket(edge_index[e - 1]);
        if (gp > gh) st |= 4;
    }
    if (st) atomicOr(status, st);
    // edge_ptr without atomics: edge e opens every graph in (batch[e-1], batch[e]]; the last edge
    // closes the rest.  (Equals the reference's scatter_add + cumsum whenever the list is grouped,
    // i.e. whenever status stays 0.)
    if (gh >= 0 && gh < B) {
        const int gprev = e > 0 ? bucket(edge_index[e - 1]) : -1;
        for (int gg = (gprev < -1 ? -1 : gprev) + 1; gg <= gh; ++gg) edge_ptr[gg] = e;
        if (e == E - 1)
            for (int gg = gh + 1; gg <= B; ++gg) edge_ptr[gg] = E;
    }
}

__global__ void k_mark_nodes(const int64_t* __restrict__ idx, int64_t n, int64_t num_nodes,
                             uint8_t* __restrict__ node_mask, int32_t* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t v = idx[i];
    if (v < 0 || v >= num_nodes) {
        atomicOr(status, 1);
        return;
    }
    node_mask[v] = 1;
}

__global__ void k_edge_near_mask(const int64_t* __restrict__ edge_index, int64_t E,
                                 const uint8_t* __restrict__ node_mask, uint8_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    out[e] = (node_mask[edge_index[e]] | node_mask[edge_index[E + e]]) ? 1 : 0;
}

// ---- CSR ------------------------------------------------------------------------------------------
struct GraphShared {
    int scan[kGraphThreads];
    int hubs[kHubListCap];
    int hub_count;
    int carry;
};

// Exclusive scan of cnt[0..n1-n0) (one counter per node of the graph) into ptr[n0..n1) (+ base);
// ptr[n1] = base + total.  All threads call.
__device__ inline void block_exclusive_scan(GraphShared& sh, const int32_t* cnt, int32_t* ptr, int64_t n0,
                                            int64_t n1, int base) {
    const int tid = threadIdx.x;
    if (tid == 0) sh.carry = base;
    __syncthreads();
    for (int64_t v0 = n0; v0 < n1; v0 += kGraphThreads) {
        const int64_t v = v0 + tid;
        const int val = v < n1 ? cnt[v - n0] : 0;
        sh.scan[tid] = val;
        __syncthreads();
        for (int off = 1; off < kGraphThreads; off <<= 1) {
            const int add = tid >= off ? sh.scan[tid - off] : 0;
            __syncthreads();
            sh.scan[tid] += add;
            __syncthreads();
        }
        const int incl = sh.scan[tid];
        const int carry = sh.carry;
        if (v < n1) ptr[v] = carry + incl - val;
        __syncthreads();
        if (tid == kGraphThreads - 1) sh.carry = carry + incl;
        __syncthreads();
    }
    if (tid == 0) ptr[n1] = sh.carry;
    __syncthreads();
}

// Counting-sort CSR of one graph.  cin / cout are the per-node counters, indexed by LOCAL node id: in
// LDS when the graph's 2 * N_g counters fit the launch's dynamic LDS (ds_add_rtn instead of L2 atomics),
// else in the global workspace.
__device__ inline void csr_build(GraphShared& sh, const int64_t* __restrict__ edge_index, int64_t E, int64_t n0,
                                 int64_t n1, int64_t e0, int64_t e1, int32_t* cin, int32_t* cout,
                                 int32_t* __restrict__ in_ptr, int32_t* __restrict__ in_nbr, int32_t* __restrict__ in_eid,
                                 int32_t* __restrict__ out_ptr, int32_t* __restrict__ out_nbr,
                                 int32_t* __restrict__ out_eid) {
    const int tid = threadIdx.x;
    const int ng = (int)(n1 - n0);
    for (int v = tid; v < ng; v += kGraphThreads) cin[v] = cout[v] = 0;
    __syncthreads();
    for (int64_t e = e0 + tid; e < e1; e += kGraphThreads) {
        const int64_t s = edge_index[e], d = edge_index[E + e];
        if (s < n0 || s >= n1 || d < n0 || d >= n1) continue;  // validated upstream; never scatter outside
        atomicAdd(&cin[d - n0], 1);
        atomicAdd(&cout[s - n0], 1);
    }
    __syncthreads();
    block_exclusive_scan(sh, cin, in_ptr, n0, n1, (int)e0);
    block_exclusive_scan(sh, cout, out_ptr, n0, n1, (int)e0);
    for (int v = tid; v < ng; v += kGraphThreads) {
        cin[v] = in_ptr[n0 + v];
        cout[v] = out_ptr[n0 + v];
    }
    __syncthreads();
    for (int64_t e = e0 + tid; e < e1; e += kGraphThreads) {
        const int64_t s = edge_index[e], d = edge_index[E + e];
        if (s < n0 || s >= n1 || d < n0 || d >= n1) continue;
        const int pi = atomicAdd(&cin[d - n0], 1);
        in_nbr[pi] = (int32_t)s;
        in_eid[pi] = (int32_t)e;
        const int po = atomicAdd(&cout[s - n0], 1);
        out_nbr[po] = (int32_t)d;
        out_eid[po] = (int32_t)e;
    }
}

__global__ __launch_bounds__(kGraphThreads) void k_graph_csr(
    const int64_t* __restrict__ edge_index, int64_t E, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ edge_ptr, int32_t* __restrict__ in_ptr, int32_t* __restrict__ in_nbr,
    int32_t* __restrict__ in_eid, int32_t* __restrict__ out_ptr, int32_t* __restrict__ out_nbr,
    int32_t* __restrict__ out_eid, int32_t* __restrict__ cnt_in, int32_t* __restrict__ cnt_out, int lds_nodes) {
    __shared__ GraphShared sh;
    extern __shared__ int32_t lds_cnt[];  // [2 * lds_nodes]
    const int g = blockIdx.x;
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int64_t e0 = edge_ptr[g], e1 = edge_ptr[g + 1];
    if (n1 - n0 <= lds_nodes)
        csr_build(sh, edge_index, E, n0, n1, e0, e1, lds_cnt, lds_cnt + lds_nodes, in_ptr, in_nbr, in_eid, out_ptr,
                  out_nbr, out_eid);
    else
        csr_build(sh, edge_index, E, n0, n1, e0, e1, cnt_in + n0, cnt_out + n0, in_ptr, in_nbr, in_eid, out_ptr, out_nbr,
                  out_eid);
}

// ---- DDE --------------------------------------------------------------------------------------------
// ns[v][c*S + j], S = 1 + R + RR: j = 0 topic, 1..R forward rounds, R+1..R+RR reverse rounds.
// One mean-propagation round from column jin to column jout over the given CSR (rows = receivers).
template <int C>
__device__ inline void dde_round(GraphShared& sh, float* __restrict__ ns, int S, int jin, int jout,
                                 const int32_t* __restrict__ ptr, const int32_t* __restrict__ nbr, int64_t n0,
                                 int64_t n1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sh.hub_count = 0;
    __syncthreads();
    bool overflow_hub = false;
    for (int64_t v = n0 + tid; v < n1; v += kGraphThreads) {
        const int b = ptr[v], e = ptr[v + 1];
        const int deg = e - b;
        if (deg > kHubDegree) {
            const int slot = atomicAdd(&sh.hub_count, 1);
            if (slot < kHubListCap) {
                sh.hubs[slot] = (int)(v - n0);
                continue;
            }
            overflow_hub = true;  // list full: fall through and do it serially (correct, slower)
        }
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        for (int p = b; p < e; ++p) {
            const float* xu = ns + (int64_t)nbr[p] * (C * S) + jin;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (double)xu[c * S];
        }
        const float cnt = deg > 0 ? (float)deg : 1.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) ns[v * (C * S) + c * S + jout] = (float)acc[c] / cnt;
    }
    (void)overflow_hub;
    __syncthreads();
    const int nh = sh.hub_count < kHubListCap ? sh.hub_count : kHubListCap;
    for (int h = wave; h < nh; h += kGraphThreads / 64) {
        const int64_t v = n0 + sh.hubs[h];
        const int b = ptr[v], e = ptr[v + 1];
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        for (int p = b + lane; p < e; p += 64) {
            const float* xu = ns + (int64_t)nbr[p] * (C * S) + jin;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (double)xu[c * S];
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
        if (lane == 0) {
            const float cnt = (float)(e - b);
#pragma unroll
            for (int c = 0; c < C; ++c) ns[v * (C * S) + c * S + jout] = (float)acc[c] / cnt;
        }
    }
    __syncthreads();
}

template <int C>
__global__ __launch_bounds__(kGraphThreads) void k_dde(
    const float* __restrict__ topic, int topic_stride, const int64_t* __restrict__ node_ptr,
    const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_nbr,
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_nbr, int rounds, int rev_rounds,
    float* __restrict__ ns) {
    __shared__ GraphShared sh;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int S = 1 + rounds + rev_rounds;
    for (int64_t v = n0 + tid; v < n1; v += kGraphThreads)
#pragma unroll
        for (int c = 0; c < C; ++c) ns[v * (C * S) + c * S] = topic[v * topic_stride + c];
    __syncthreads();
    // forward rounds: messages flow src -> dst, so a node averages over its in-edges
    for (int j = 1; j <= rounds; ++j) dde_round<C>(sh, ns, S, j - 1, j, in_ptr, in_nbr, n0, n1);
    // reverse rounds on edge_index.flip(0): a node averages over its out-edges' heads
    for (int j = 1; j <= rev_rounds; ++j)
        dde_round<C>(sh, ns, S, j == 1 ? 0 : rounds + j - 1, rounds + j, out_ptr, out_nbr, n0, n1);
}

}  // namespace evi

using namespace evi;

extern "C" int evi_edge_batch(const int64_t* edge_index, int64_t E, const int64_t* node_ptr, int B,
                              int64_t* edge_batch, int64_t* edge_ptr, int32_t* status, void* stream) {
    EVI_REQUIRE(E >= 0 && B >= 1, "evi_edge_batch: need E >= 0 and B >= 1, got E=%lld B=%d", (long long)E, B);
    EVI_REQUIRE(node_ptr && edge_ptr && status, "evi_edge_batch: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    EVI_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    EVI_HIP_CHECK(hipMemsetAsync(edge_ptr, 0, sizeof(int64_t) * (B + 1), st));  // E == 0: all zeros
    if (E > 0) {
        EVI_REQUIRE(edge_index && edge_batch, "evi_edge_batch: null pointer");
        hipLaunchKernelGGL(k_edge_batch, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, edge_index, E,
                           node_ptr, B, edge_batch, edge_ptr, status);
        EVI_LAUNCH_CHECK();
    }
    return EVI_OK;
}

extern "C" int evi_qa_edge_mask(const int64_t* edge_index, int64_t E, int64_t num_nodes,
                                const int64_t* q_idx, int64_t nq, const int64_t* a_idx, int64_t na,
                                uint8_t* node_mask_ws, uint8_t* out_mask, int32_t* status, void* stream) {
    EVI_REQUIRE(E >= 0 && num_nodes > 0, "evi_qa_edge_mask: num_nodes must be positive, got %lld",
                (long long)num_nodes);
    EVI_REQUIRE(node_mask_ws && status, "evi_qa_edge_mask: null workspace");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    EVI_HIP_CHECK(hipMemsetAsync(node_mask_ws, 0, (size_t)num_nodes, st));
    EVI_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    if (nq > 0) {
        hipLaunchKernelGGL(k_mark_nodes, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, q_idx, nq,
                           num_nodes, node_mask_ws, status);
        EVI_LAUNCH_CHECK();
    }
    if (na > 0) {
        hipLaunchKernelGGL(k_mark_nodes, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, a_idx, na,
                           num_nodes, node_mask_ws, status);
        EVI_LAUNCH_CHECK();
    }
    if (E > 0) {
        EVI_REQUIRE(edge_index && out_mask, "evi_qa_edge_mask: null pointer");
        hipLaunchKernelGGL(k_edge_near_mask, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, edge_index, E,
                           node_mask_ws, out_mask);
        EVI_LAUNCH_CHECK();
    }
    return EVI_OK;
}

extern "C" size_t evi_graph_csr_workspace_bytes(int64_t N) { return (size_t)(N > 0 ? N : 1) * 2 * sizeof(int32_t); }

extern "C" int evi_graph_csr(const int64_t* edge_index, int64_t E, const int64_t* node_ptr,
                             const int64_t* edge_ptr, int B, int64_t N, int32_t* in_ptr, int32_t* in_nbr,
                             int32_t* in_eid, int32_t* out_ptr, int32_t* out_nbr, int32_t* out_eid,
                             void* workspace, size_t workspace_bytes, void* stream) {
    EVI_REQUIRE(B >= 1 && N >= 0 && E >= 0, "evi_graph_csr: bad sizes B=%d N=%lld E=%lld", B, (long long)N,
                (long long)E);
    EVI_REQUIRE(E < (int64_t)0x7FFFFFFF && N < (int64_t)0x7FFFFFFF, "evi_graph_csr: batch too large for int32 CSR");
    EVI_REQUIRE(node_ptr && edge_ptr && in_ptr && out_ptr && workspace, "evi_graph_csr: null pointer");
    EVI_REQUIRE(E == 0 || (edge_index && in_nbr && in_eid && out_nbr && out_eid), "evi_graph_csr: null pointer");
    if (workspace_bytes < evi_graph_csr_workspace_bytes(N))
        return fail(EVI_ERR_NOMEM, "evi_graph_csr: workspace %zu B < %zu B", workspace_bytes,
                    evi_graph_csr_workspace_bytes(N));
    int32_t* cnt_in = static_cast<int32_t*>(workspace);
    int32_t* cnt_out = cnt_in + (N > 0 ? N : 1);
    // 48 KiB of dynamic LDS: the counters of graphs up to 6144 nodes stay on chip (larger graphs use the workspace)
    constexpr int kLdsNodes = 6144;
    hipLaunchKernelGGL(k_graph_csr, dim3(B), dim3(kGraphThreads), 2 * kLdsNodes * sizeof(int32_t),
                       reinterpret_cast<hipStream_t>(stream), edge_index, E, node_ptr, edge_ptr, in_ptr, in_nbr, in_eid,
                       out_ptr, out_nbr, out_eid, cnt_in, cnt_out, kLdsNodes);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_dde_node_struct(const float* topic_one_hot, int topic_stride, int num_topics,
                                   const int64_t* node_ptr, int B, const int32_t* in_ptr, const int32_t* in_nbr,
                                   const int32_t* out_ptr, const int32_t* out_nbr, int rounds, int rev_rounds,
                                   float* node_struct, void* stream) {
    EVI_REQUIRE(B >= 1, "evi_dde_node_struct: B must be >= 1, got %d", B);
    EVI_REQUIRE(rounds >= 0 && rounds <= 4 && rev_rounds >= 0 && rev_rounds <= 4,
                "DDE supports at most 4 rounds per direction; got num_rounds=%d, num_reverse_rounds=%d.", rounds,
                rev_rounds);
    if (num_topics != 2)
        return fail(EVI_ERR_INVALID, "num_topics must be 2 (seed vs non-seed), got %d", num_topics);
    EVI_REQUIRE(topic_stride >= num_topics, "evi_dde_node_struct: topic_one_hot feature dim %d < num_topics=%d",
                topic_stride, num_topics);
    EVI_REQUIRE(topic_one_hot && node_ptr && in_ptr && out_ptr && node_struct, "evi_dde_node_struct: null pointer");
    hipLaunchKernelGGL(k_dde<2>, dim3(B), dim3(kGraphThreads), 0, reinterpret_cast<hipStream_t>(stream),
                       topic_one_hot, topic_stride, node_ptr, in_ptr, in_nbr, out_ptr, out_nbr, rounds,
                       rev_rounds, node_struct);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
