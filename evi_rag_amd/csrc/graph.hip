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
// Graphs in a batch are independent and small (N_g ~ 10^3, E_g ~ 10^3..10^5).  All kernels are HBM/L2-latency-bound
// integer and gather work; algorithmic bytes per DDE round = E*(4 nbr + C*4 gather) + N*(8 rowptr + C*4 write)
// (SURVEY.md §8d).  The reference's batch is 32 graphs, so "one workgroup per graph" leaves 7/8 of the chip idle:
//   * the CSR is a counting sort whose counters live in LDS (global atomics run at the memory side, ~20 G/s scattered);
//     a graph's edge list is cut into P parts (P = 8 at 32 graphs, 1 from 512 graphs on), every part counts and fills from
//     its own LDS counters, and a per-graph scan in between turns the P x N_g partial counts into row pointers and
//     per-part cursors — B x P workgroups, no global atomics, no per-level barriers of 1024 threads;
//   * DDE is node-parallel over the WHOLE batch: one thread per (node, chain) — the forward chain walks in-rows, the
//     reverse chain out-rows, both are independent given the previous round — and one launch per round pair instead of
//     workgroup barriers, so a round is one row gather deep whatever the batch size.  Long rows (power-law hubs) are
//     summed by the whole wave.
//
// CSR rows are filled through atomic cursors, so the order of a row's entries is not defined.
// Nothing downstream depends on it: BFS levels are order-free, and DDE sums a row in f64 before
// rounding once to f32, which makes the f32 result independent of the summation order.
#include "common.hpp"


#include <stdlib.h>

namespace evi {

constexpr int kGraphThreads = 1024;
constexpr int kHubDegree = 32;
constexpr int kHubListCap = 4096;
constexpr int kCsrMaxParts = 8;  // edge-list parts per graph of the small-batch CSR build

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
// ptr[n1] = base + total.  All threads call.  Wave-level shuffles scan 64 counters at a time, the 16 wave totals go
// through LDS: two barriers per 1024 nodes instead of the twenty of a log-step LDS scan.
__device__ inline int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(v, off, 64);
        if (lane >= off) v += up;
    }
    return v;
}

template <class Sh, class Count>
__device__ inline void block_exclusive_scan_fn(Sh& sh, Count cnt, int32_t* ptr, int64_t n0, int64_t n1, int base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kGraphThreads / 64;
    int carry = base;
    for (int64_t v0 = n0; v0 < n1; v0 += kGraphThreads) {
        const int64_t v = v0 + tid;
        const int val = v < n1 ? cnt((int)(v - n0)) : 0;
        const int incl = wave_inclusive_scan(val);
        if (lane == 63) sh.scan[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int t = sh.scan[w];
            if (w < wave) before += t;
            total += t;
        }
        if (v < n1) ptr[v] = carry + before + incl - val;
        carry += total;
        __syncthreads();  // sh.scan is rewritten by the next chunk
    }
    if (tid == 0) ptr[n1] = carry;
    __syncthreads();
}

template <class Sh>
__device__ inline void block_exclusive_scan(Sh& sh, const int32_t* cnt, int32_t* ptr, int64_t n0,
                                            int64_t n1, int base) {
    block_exclusive_scan_fn(sh, [&](int i) { return cnt[i]; }, ptr, n0, n1, base);
}

// Counting-sort CSR of one graph.  cin / cout are the per-node counters, indexed by LOCAL node id: in
// LDS when the graph's 2 * N_g counters fit the launch's dynamic LDS (ds_add_rtn instead of L2 atomics),
// else in the global workspace.
template <class Sh>
__device__ inline void csr_build(Sh& sh, const int64_t* __restrict__ edge_index, int64_t E, int64_t n0,
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

// ---- CSR with the output staged in LDS: one workgroup per (graph, side) ---------------------------------------
// The counting sort above scatters 4-byte entries over a graph's (nbr, eid) arrays.  With the arrays in global memory every
// scattered store dirties a line that leaves the L2 partly filled: the memory-side write counter shows 6.5x (512 graphs per
// batch, the working set of the concurrent workgroups exceeds the L2s) to 7.6x (32 graphs cut into 8 parts that run on
// different XCDs, i.e. different L2s each holding partial lines of the same rows) the bytes of the arrays
// (profiles/r03_pmc_graph.json: WRITE_SIZE 45 MB per batch of 32 graphs against 6 MB of CSR).  Here one side of one graph's
// CSR — in-rows (keyed by target) or out-rows (keyed by source) — is built in LDS: the row counters AND the side's (nbr, eid)
// arrays (up to 14 336 edges = 112 KiB), scattered into with ds atomics, then written to global memory once, coalesced.
// One launch instead of three; each of the two sides reads the graph's edge list twice (the second pass hits the L2: both
// sides of a graph are placed on the same XCD).  Graphs larger than the LDS budget take the global-memory build.
constexpr int kCsrSideNodes = 8192;    // row counters: 32 KiB
constexpr int kCsrSideEdges = 14336;   // staged (nbr, eid): 2 x 56 KiB
struct CsrSideShared {
    int scan[kGraphThreads / 64];
};

__global__ __launch_bounds__(kGraphThreads) void k_graph_csr_side(
    const int64_t* __restrict__ edge_index, int64_t E, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ edge_ptr, int B, int32_t* __restrict__ in_ptr, int32_t* __restrict__ in_nbr,
    int32_t* __restrict__ in_eid, int32_t* __restrict__ out_ptr, int32_t* __restrict__ out_nbr,
    int32_t* __restrict__ out_eid, int32_t* __restrict__ cnt_in, int32_t* __restrict__ cnt_out) {
    __shared__ CsrSideShared sh;
    extern __shared__ int32_t lds[];  // cnt [kCsrSideNodes] | nbr [kCsrSideEdges] | eid [kCsrSideEdges]
    // workgroup i runs on XCD i % 8: both sides of a graph get the same residue, so the second reader of the graph's edge
    // list finds it in that XCD's L2
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int side = q & 1;
    const int g = (q >> 1) * 8 + xcd;
    if (g >= B) return;
    const int tid = threadIdx.x;
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int64_t e0 = edge_ptr[g], e1 = edge_ptr[g + 1];
    const int ng = (int)(n1 - n0);
    const int64_t ne = e1 - e0;
    if (ng > kCsrSideNodes || ne > kCsrSideEdges) {  // too large for LDS: side 0 builds both halves in global memory
        if (side == 0)
            csr_build(sh, edge_index, E, n0, n1, e0, e1, cnt_in + n0, cnt_out + n0, in_ptr, in_nbr, in_eid, out_ptr, out_nbr, out_eid);
        return;
    }
    int32_t* cnt = lds;
    int32_t* s_nbr = lds + kCsrSideNodes;
    int32_t* s_eid = s_nbr + kCsrSideEdges;
    const int64_t* key = edge_index + (side == 0 ? E : 0);  // in-rows are keyed by the target, out-rows by the source
    const int64_t* oth = edge_index + (side == 0 ? 0 : E);
    int32_t* ptr = side == 0 ? in_ptr : out_ptr;
    int32_t* nbr = side == 0 ? in_nbr : out_nbr;
    int32_t* eid = side == 0 ? in_eid : out_eid;
    for (int v = tid; v < ng; v += kGraphThreads) cnt[v] = 0;
    __syncthreads();
    // pass 1: row sizes.  Four edges per trip, loads first (an index past the end re-reads the last edge and is dropped)
    for (int64_t e = e0 + tid; e < e1; e += 4 * kGraphThreads) {
        int64_t k[4], o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t ee = e + (int64_t)i * kGraphThreads < e1 ? e + (int64_t)i * kGraphThreads : e1 - 1;
            k[i] = key[ee];
            o[i] = oth[ee];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (e + (int64_t)i * kGraphThreads >= e1) continue;
            if (k[i] < n0 || k[i] >= n1 || o[i] < n0 || o[i] >= n1) continue;  // validated upstream; never scatter outside
            atomicAdd(&cnt[k[i] - n0], 1);
        }
    }
    __syncthreads();
    block_exclusive_scan_fn(sh, [&](int i) { return cnt[i]; }, ptr, n0, n1, (int)e0);
    for (int v = tid; v < ng; v += kGraphThreads) cnt[v] = ptr[n0 + v] - (int)e0;  // cursors, relative to the graph's first slot
    __syncthreads();
    const int total = ptr[n1] - (int)e0;
    // pass 2: scatter into LDS
    for (int64_t e = e0 + tid; e < e1; e += 4 * kGraphThreads) {
        int64_t k[4], o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t ee = e + (int64_t)i * kGraphThreads < e1 ? e + (int64_t)i * kGraphThreads : e1 - 1;
            k[i] = key[ee];
            o[i] = oth[ee];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t ee = e + (int64_t)i * kGraphThreads;
            if (ee >= e1) continue;
            if (k[i] < n0 || k[i] >= n1 || o[i] < n0 || o[i] >= n1) continue;
            const int pos = atomicAdd(&cnt[k[i] - n0], 1);
            s_nbr[pos] = (int32_t)o[i];
            s_eid[pos] = (int32_t)ee;
        }
    }
    __syncthreads();
    // one coalesced pass out
    for (int i = tid; i < total; i += kGraphThreads) {
        nbr[e0 + i] = s_nbr[i];
        eid[e0 + i] = s_eid[i];
    }
}

// ---- CSR in P parts per graph (small batches) ------------------------------------------------------------
// part counts / cursors: [2][P][N] int32 in the workspace (in-counters first, out-counters second), part-major: the count
// and fill kernels of part p stream their own [N_g] slice, the scan kernel reads P coalesced streams.
__device__ inline void part_range(int64_t e0, int64_t e1, int P, int p, int64_t& b, int64_t& e) {
    const int64_t len = e1 - e0;
    b = e0 + len * p / P;
    e = e0 + len * (p + 1) / P;
}

// A: count the part's edges per node in LDS, publish the partial counts.
__global__ __launch_bounds__(kGraphThreads) void k_csr_part_count(
    const int64_t* __restrict__ edge_index, int64_t E, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ edge_ptr, int64_t N, int P, int32_t* __restrict__ part, int lds_nodes) {
    extern __shared__ int32_t lds_cnt[];  // [2 * lds_nodes]
    const int g = blockIdx.x / P, p = blockIdx.x % P, tid = threadIdx.x;
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int ng = (int)(n1 - n0);
    if (ng > lds_nodes) return;  // oversize graph: k_csr_part_scan builds it whole (global counters)
    int32_t* cin = lds_cnt;
    int32_t* cout = lds_cnt + lds_nodes;
    for (int v = tid; v < ng; v += kGraphThreads) cin[v] = cout[v] = 0;
    __syncthreads();
    int64_t b, e;
    part_range(edge_ptr[g], edge_ptr[g + 1], P, p, b, e);
    for (int64_t i = b + tid; i < e; i += kGraphThreads) {
        const int64_t s = edge_index[i], d = edge_index[E + i];
        if (s < n0 || s >= n1 || d < n0 || d >= n1) continue;
        atomicAdd(&cin[d - n0], 1);
        atomicAdd(&cout[s - n0], 1);
    }
    __syncthreads();
    int32_t* pin = part + (int64_t)p * N + n0;
    int32_t* pout = part + ((int64_t)P + p) * N + n0;
    for (int v = tid; v < ng; v += kGraphThreads) {
        pin[v] = cin[v];
        pout[v] = cout[v];
    }
}

// B: one workgroup per graph: row pointers from the summed partial counts; the partial counts become each part's first
// write position of the row.  Graphs too large for the parts' LDS counters are built here in one piece (legacy path).
__global__ __launch_bounds__(kGraphThreads) void k_csr_part_scan(
    const int64_t* __restrict__ edge_index, int64_t E, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ edge_ptr, int64_t N, int P, int32_t* __restrict__ part, int32_t* __restrict__ in_ptr,
    int32_t* __restrict__ in_nbr, int32_t* __restrict__ in_eid, int32_t* __restrict__ out_ptr,
    int32_t* __restrict__ out_nbr, int32_t* __restrict__ out_eid, int lds_nodes) {
    __shared__ GraphShared sh;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int64_t e0 = edge_ptr[g], e1 = edge_ptr[g + 1];
    const int ng = (int)(n1 - n0);
    if (ng > lds_nodes) {  // the graph's slices of parts 0 of `part` serve as its global counters
        csr_build(sh, edge_index, E, n0, n1, e0, e1, part + n0, part + (int64_t)P * N + n0, in_ptr, in_nbr, in_eid, out_ptr,
                  out_nbr, out_eid);
        return;
    }
    // Both halves in ONE pass: a node's in- and out-totals ride in the two halves of a 64-bit word through the same
    // wave / workgroup scan (each sum stays below 2^31, so no carry crosses), and the part cursors are written from the
    // registers that still hold the part counts — per 1024 nodes: 2 P independent loads, one scan, two barriers.
    int32_t* pci = part + n0;                       // + q * N + v
    int32_t* pco = part + (int64_t)P * N + n0;
    unsigned long long* wsum = reinterpret_cast<unsigned long long*>(sh.scan);
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long carry = (unsigned long long)(uint32_t)e0 | ((unsigned long long)(uint32_t)e0 << 32);
    for (int v0 = 0; v0 < ng; v0 += kGraphThreads) {
        const int v = v0 + tid;
        const bool live = v < ng;
        int ci[kCsrMaxParts], co[kCsrMaxParts];
#pragma unroll
        for (int q = 0; q < kCsrMaxParts; ++q) {
            ci[q] = (live && q < P) ? pci[(int64_t)q * N + v] : 0;
            co[q] = (live && q < P) ? pco[(int64_t)q * N + v] : 0;
        }
        unsigned int ti = 0, to = 0;
#pragma unroll
        for (int q = 0; q < kCsrMaxParts; ++q) {
            ti += (unsigned)ci[q];
            to += (unsigned)co[q];
        }
        const unsigned long long val = (unsigned long long)ti | ((unsigned long long)to << 32);
        unsigned long long incl = val;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned long long before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kGraphThreads / 64; ++w) {
            const unsigned long long t = wsum[w];
            if (w < wave) before += t;
            total += t;
        }
        const unsigned long long mine = carry + before + incl - val;
        if (live) {
            int run_i = (int)(uint32_t)mine, run_o = (int)(uint32_t)(mine >> 32);
            in_ptr[n0 + v] = run_i;
            out_ptr[n0 + v] = run_o;
#pragma unroll
            for (int q = 0; q < kCsrMaxParts; ++q)
                if (q < P) {
                    pci[(int64_t)q * N + v] = run_i;
                    pco[(int64_t)q * N + v] = run_o;
                    run_i += ci[q];
                    run_o += co[q];
                }
        }
        carry += total;
        __syncthreads();  // wsum is rewritten by the next chunk
    }
    if (tid == 0) {
        in_ptr[n1] = (int)(uint32_t)carry;
        out_ptr[n1] = (int)(uint32_t)(carry >> 32);
    }
}

// C: every part scatters its edges from its own cursors (LDS).
__global__ __launch_bounds__(kGraphThreads) void k_csr_part_fill(
    const int64_t* __restrict__ edge_index, int64_t E, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ edge_ptr, int64_t N, int P, const int32_t* __restrict__ part,
    int32_t* __restrict__ in_nbr, int32_t* __restrict__ in_eid, int32_t* __restrict__ out_nbr,
    int32_t* __restrict__ out_eid, int lds_nodes) {
    extern __shared__ int32_t lds_cnt[];
    const int g = blockIdx.x / P, p = blockIdx.x % P, tid = threadIdx.x;
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int ng = (int)(n1 - n0);
    if (ng > lds_nodes) return;
    int32_t* cin = lds_cnt;
    int32_t* cout = lds_cnt + lds_nodes;
    const int32_t* pin = part + (int64_t)p * N + n0;
    const int32_t* pout = part + ((int64_t)P + p) * N + n0;
    for (int v = tid; v < ng; v += kGraphThreads) {
        cin[v] = pin[v];
        cout[v] = pout[v];
    }
    __syncthreads();
    int64_t b, e;
    part_range(edge_ptr[g], edge_ptr[g + 1], P, p, b, e);
    for (int64_t i = b + tid; i < e; i += kGraphThreads) {
        const int64_t s = edge_index[i], d = edge_index[E + i];
        if (s < n0 || s >= n1 || d < n0 || d >= n1) continue;
        const int pi = atomicAdd(&cin[d - n0], 1);
        in_nbr[pi] = (int32_t)s;
        in_eid[pi] = (int32_t)i;
        const int po = atomicAdd(&cout[s - n0], 1);
        out_nbr[po] = (int32_t)d;
        out_eid[po] = (int32_t)i;
    }
}

// ---- DDE --------------------------------------------------------------------------------------------
// ns[v][c*S + j], S = 1 + R + RR: j = 0 topic, 1..R forward rounds, R+1..R+RR reverse rounds.
// One launch = one mean-propagation round of BOTH chains over the whole batch: thread t < Npad works on node t of the
// forward chain (receivers' rows = in-rows), thread Npad + t on node t of the reverse chain (out-rows).  A round reads
// column jin of the neighbours (jin == 0: the topic one-hot itself, so no initial copy pass is needed) and writes column
// jout of its own node; the first launch also copies the topic columns into ns[:, c*S].
struct DdeChain {
    const int32_t* ptr;
    const int32_t* nbr;
    int jin, jout;  // jout < 0: chain idle in this launch
};

template <int C>
__global__ __launch_bounds__(256) void k_dde_round(const float* __restrict__ topic, int topic_stride, float* __restrict__ ns,
                                                   int S, int64_t N, int64_t Npad, DdeChain fwd, DdeChain rev,
                                                   int write_topic) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int chain = t >= Npad ? 1 : 0;  // wave-uniform: Npad is a multiple of 64
    const int64_t v = t - (chain ? Npad : 0);
    const DdeChain ch = chain ? rev : fwd;
    const bool live = v < N;
    if (live && write_topic && chain == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) ns[v * (C * S) + c * S] = topic[v * topic_stride + c];
    }
    if (ch.jout < 0) return;  // uniform per wave
    const bool from_topic = ch.jin == 0;
    auto prev = [&](int64_t u, int c) -> float {
        return from_topic ? topic[u * topic_stride + c] : ns[u * (C * S) + c * S + ch.jin];
    };
    int b = 0, e = 0;
    if (live) {
        b = ch.ptr[v];
        e = ch.ptr[v + 1];
    }
    const int deg = e - b;
    const bool hub = deg > kHubDegree;
    if (live && !hub) {
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        // four neighbours per trip, every load of a trip issued before the first use (an index past the row's end is
        // clamped and its value dropped): a row of <= 4 entries costs two dependent memory latencies, not two per entry
        for (int p = b; p < e; p += 4) {
            int64_t u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = ch.nbr[p + i < e ? p + i : e - 1];
            float x[4][C];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < C; ++c) x[i][c] = prev(u[i], c);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (p + i < e) {
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] += (double)x[i][c];
                }
        }
        const float cnt = deg > 0 ? (float)deg : 1.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) ns[v * (C * S) + c * S + ch.jout] = (float)acc[c] / cnt;
    }
    // long rows (power-law hubs): the wave sums them together, one after the other
    unsigned long long hubs = __ballot(hub);
    while (hubs) {
        const int l = __ffsll((long long)hubs) - 1;
        hubs &= hubs - 1;
        const int hb = __shfl(b, l, 64), he = __shfl(e, l, 64);
        const int64_t hv = (t - lane + l) - (chain ? Npad : 0);
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        for (int p = hb + lane; p < he; p += 256) {  // four entries per lane per trip, loads first
            int64_t u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = ch.nbr[p + 64 * i < he ? p + 64 * i : he - 1];
            float x[4][C];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < C; ++c) x[i][c] = prev(u[i], c);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (p + 64 * i < he) {
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] += (double)x[i][c];
                }
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
        if (lane == 0) {
            const float cnt = (float)(he - hb);
#pragma unroll
            for (int c = 0; c < C; ++c) ns[hv * (C * S) + c * S + ch.jout] = (float)acc[c] / cnt;
        }
    }
}

// ---- DDE, one workgroup per graph: the whole feature block of the graph lives in LDS ---------------------------------
// The node-parallel kernel above gathers 8 bytes of a neighbour's previous column out of its 40-byte ns row: at 512 graphs
// per batch the ns array (61 MB) is far beyond the L2s, every gather pulls a whole line from memory, and the memory-side
// counters show 3.7x the algorithmic bytes at 5.6 TB/s — HBM-bound on over-fetch (profiles/r03_pmc_graph_before_lds_kernels.json).
// Here one workgroup owns a graph: the propagated state, the row pointers and the neighbour lists (16-bit local ids) of the side
// being walked live in LDS, rounds are separated by workgroup barriers, every round's column goes straight to the graph's ns
// rows.  Memory traffic: each side's rows once, the topic rows in, the ns columns out.  A graph that does not fit the LDS runs
// the row-walking rounds on its global ns rows (dde_graph_rounds).
// (An edge-parallel variant — the edge list in LDS, acc[dst] += x[src] with f64 LDS atomic adds, exact and order-free — was built
// and measured in round 3: 100 us per CWQ-shaped graph against 84 us for this kernel and 30 us per BATCH of 32 for the
// node-parallel one: 80 000 ds_add_f64 per graph run at ~2.5 cycles each.  Removed.)
constexpr int kDdeGraphLdsFloats = 36864;  // 144 KiB

template <int C, class F>
__device__ inline void dde_graph_rounds(F* __restrict__ st, int W, int S, int ng, int64_t n0, const int32_t* __restrict__ in_ptr,
                                        const int32_t* __restrict__ in_nbr, const int32_t* __restrict__ out_ptr,
                                        const int32_t* __restrict__ out_nbr, int rounds, int rev_rounds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int ngpad = (ng + 63) / 64 * 64;
    for (int step = 0; step < rounds + rev_rounds; ++step) {
        const bool rev = step >= rounds;
        const int j = rev ? step - rounds + 1 : step + 1;
        const int jin = rev ? (j == 1 ? 0 : rounds + j - 1) : j - 1;
        const int jout = rev ? rounds + j : j;
        const int32_t* __restrict__ ptr = rev ? out_ptr : in_ptr;
        const int32_t* __restrict__ nbr = rev ? out_nbr : in_nbr;
        for (int v = tid; v < ngpad; v += kGraphThreads) {  // whole waves: the hub rows below are summed by all 64 lanes
            const bool live = v < ng;
            int b = 0, e = 0;
            if (live) {
                b = ptr[n0 + v];
                e = ptr[n0 + v + 1];
            }
            const int deg = e - b;
            const bool hub = deg > kHubDegree;
            if (live && !hub) {
                double acc[C];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = 0.0;
                for (int p = b; p < e; p += 4) {  // four neighbours per trip, loads first (see k_dde_round)
                    int u[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[i] = nbr[p + i < e ? p + i : e - 1] - (int)n0;
                    float x[4][C];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < C; ++c) x[i][c] = st[u[i] * W + c * S + jin];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (p + i < e) {
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[c] += (double)x[i][c];
                        }
                }
                const float cnt = deg > 0 ? (float)deg : 1.0f;
#pragma unroll
                for (int c = 0; c < C; ++c) st[v * W + c * S + jout] = (float)acc[c] / cnt;
            }
            unsigned long long hubs = __ballot(hub);
            while (hubs) {
                const int l = __ffsll((long long)hubs) - 1;
                hubs &= hubs - 1;
                const int hb = __shfl(b, l, 64), he = __shfl(e, l, 64);
                const int hv = v - lane + l;
                double acc[C];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = 0.0;
                for (int p = hb + lane; p < he; p += 256) {
                    int u[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[i] = nbr[p + 64 * i < he ? p + 64 * i : he - 1] - (int)n0;
                    float x[4][C];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < C; ++c) x[i][c] = st[u[i] * W + c * S + jin];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (p + 64 * i < he) {
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[c] += (double)x[i][c];
                        }
                }
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
                if (lane == 0) {
                    const float cnt = (float)(he - hb);
#pragma unroll
                    for (int c = 0; c < C; ++c) st[hv * W + c * S + jout] = (float)acc[c] / cnt;
                }
            }
        }
        __syncthreads();  // column jout is complete before a later round reads it
    }
}

// One mean-propagation round on LDS-resident rows: xn[v] = mean over row(v) of xc[u]; also ns[v][c * S + jout] (global).
// ptr: [ng + 1] row offsets relative to the graph's first slot; nbr: local ids as 16-bit.
template <int C>
__device__ inline void dde_lds_round(const float* __restrict__ xc, float* __restrict__ xn, const int32_t* __restrict__ ptr,
                                     const uint16_t* __restrict__ nbr, int ng, float* __restrict__ nsg, int W, int S, int jout) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int ngpad = (ng + 63) / 64 * 64;
    for (int v = tid; v < ngpad; v += kGraphThreads) {  // whole waves: hub rows are summed by all 64 lanes
        const bool live = v < ng;
        int b = 0, e = 0;
        if (live) {
            b = ptr[v];
            e = ptr[v + 1];
        }
        const int deg = e - b;
        const bool hub = deg > kHubDegree;
        if (live && !hub) {
            double acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0.0;
            for (int p = b; p < e; p += 4) {  // four neighbours per trip, reads first
                int u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = nbr[p + i < e ? p + i : e - 1];
                float x[4][C];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < C; ++c) x[i][c] = xc[u[i] * C + c];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (p + i < e) {
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] += (double)x[i][c];
                    }
            }
            const float cnt = deg > 0 ? (float)deg : 1.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float val = (float)acc[c] / cnt;
                xn[v * C + c] = val;
                nsg[v * W + c * S + jout] = val;
            }
        }
        unsigned long long hubs = __ballot(hub);
        while (hubs) {
            const int l = __ffsll((long long)hubs) - 1;
            hubs &= hubs - 1;
            const int hb = __shfl(b, l, 64), he = __shfl(e, l, 64);
            const int hv = v - lane + l;
            double acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0.0;
            for (int p = hb + lane; p < he; p += 256) {
                int u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = nbr[p + 64 * i < he ? p + 64 * i : he - 1];
                float x[4][C];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < C; ++c) x[i][c] = xc[u[i] * C + c];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (p + 64 * i < he) {
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] += (double)x[i][c];
                    }
            }
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
            if (lane == 0) {
                const float cnt = (float)(he - hb);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float val = (float)acc[c] / cnt;
                    xn[hv * C + c] = val;
                    nsg[hv * W + c * S + jout] = val;
                }
            }
        }
    }
}

// LDS (ints): x0 [ng C] | x1 [ng C] | ptr [ng + 1] | nbr16 [ceil(ne / 2)] — the rows of ONE side at a time (the forward chain
// walks in-rows, the reverse chain out-rows; the side is reloaded between the chains): a CWQ graph needs 12 + 12 + 12 + 20 KB.
// Everything a round touches is then on chip: the state, the row pointers AND the neighbour lists — the first version of this
// kernel kept the whole [N_g, 10] output block in LDS instead and read the neighbour lists from global memory: every hub row
// (summed by one wave, one hub after the other) paid a fresh memory latency, 84 us per CWQ-shaped graph.
template <int C>
__global__ __launch_bounds__(kGraphThreads) void k_dde_graph(const float* __restrict__ topic, int topic_stride, float* __restrict__ ns,
                                                             int S, const int64_t* __restrict__ node_ptr,
                                                             const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_nbr,
                                                             const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_nbr,
                                                             int rounds, int rev_rounds, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds_ns[];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int ng = (int)(n1 - n0);
    const int W = C * S;
    float* __restrict__ nsg = ns + n0 * W;
    const int e0 = in_ptr[n0];
    const int ne = in_ptr[n1] - e0;  // valid entries of either half
    for (int v = tid; v < ng; v += kGraphThreads)
#pragma unroll
        for (int c = 0; c < C; ++c) nsg[v * W + c * S] = topic[(n0 + v) * topic_stride + c];  // column 0: the one-hot itself
    if (ng > 65535 || 2 * (int64_t)ng * C + (ng + 1) + (ne + 1) / 2 > lds_floats) {
        // does not fit: the same rounds on the graph's global ns rows, rows from global memory
        __syncthreads();
        dde_graph_rounds<C>(nsg, W, S, ng, n0, in_ptr, in_nbr, out_ptr, out_nbr, rounds, rev_rounds);
        return;
    }
    float* x0 = lds_ns;
    float* x1 = x0 + ng * C;
    int32_t* l_ptr = reinterpret_cast<int32_t*>(x1 + ng * C);
    uint16_t* l_nbr = reinterpret_cast<uint16_t*>(l_ptr + ng + 1);
    for (int chain = 0; chain < 2; ++chain) {
        const int nr = chain ? rev_rounds : rounds;
        if (nr == 0) continue;
        const int32_t* __restrict__ gptr = chain ? out_ptr : in_ptr;
        const int32_t* __restrict__ gnbr = chain ? out_nbr : in_nbr;
        __syncthreads();  // the previous chain is done with the LDS
        for (int v = tid; v < ng; v += kGraphThreads)
#pragma unroll
            for (int c = 0; c < C; ++c) x0[v * C + c] = topic[(n0 + v) * topic_stride + c];  // a chain starts from the one-hot
        for (int i0 = tid; i0 <= ng; i0 += 8 * kGraphThreads) {  // eight per trip, loads first
            int t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = gptr[n0 + (i0 + u * kGraphThreads <= ng ? i0 + u * kGraphThreads : ng)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * kGraphThreads <= ng) l_ptr[i0 + u * kGraphThreads] = t[u] - e0;
        }
        for (int i0 = tid; i0 < ne; i0 += 8 * kGraphThreads) {
            int t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = gnbr[e0 + (i0 + u * kGraphThreads < ne ? i0 + u * kGraphThreads : ne - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * kGraphThreads < ne) l_nbr[i0 + u * kGraphThreads] = (uint16_t)(t[u] - (int)n0);
        }
        __syncthreads();
        float* xc = x0;
        float* xn = x1;
        for (int j = 1; j <= nr; ++j) {
            dde_lds_round<C>(xc, xn, l_ptr, l_nbr, ng, nsg, W, S, chain ? rounds + j : j);
            __syncthreads();
            float* t = xc;
            xc = xn;
            xn = t;
        }
    }
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

constexpr int kCsrLdsNodes = 6144;  // 48 KiB of dynamic LDS: the counters of graphs up to 6144 nodes stay on chip

// parts per graph: enough workgroups to cover the chip twice at small batches, one from 512 graphs on
static int csr_parts(int B) {
    int P = 1;
    while (P < kCsrMaxParts && (int64_t)B * P * 2 <= 512) P <<= 1;
    if (const char* e = getenv("EVI_CSR_PARTS")) {  // tuning / tests: force the number of parts (1, 2, 4, 8)
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) P = v;
    }
    return P;
}

extern "C" size_t evi_graph_csr_workspace_bytes(int64_t N) {
    return (size_t)(N > 0 ? N : 1) * 2 * kCsrMaxParts * sizeof(int32_t);
}

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
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int P = csr_parts(B);
    const size_t lds = 2 * kCsrLdsNodes * sizeof(int32_t);
    // default: the LDS-staged build, one workgroup per (graph, side) in ONE launch; EVI_CSR_BUILD=parts selects the older
    // global-scatter kernels (A/B runs, and the path graphs beyond the LDS budget take inside the new kernel anyway)
    static const bool staged = [] {
        const char* e = getenv("EVI_CSR_BUILD");
        return !(e && e[0] == 'p');
    }();
    if (staged && !getenv("EVI_CSR_PARTS")) {
        static thread_local bool attr = false;
        const size_t side_lds = (size_t)(kCsrSideNodes + 2 * kCsrSideEdges) * sizeof(int32_t);
        if (!attr) {
            EVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_graph_csr_side),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)side_lds));
            attr = true;
        }
        int32_t* cnt_in = static_cast<int32_t*>(workspace);
        int32_t* cnt_out = cnt_in + (N > 0 ? N : 1);
        const unsigned grid = (unsigned)((B + 7) / 8 * 16);
        hipLaunchKernelGGL(k_graph_csr_side, dim3(grid), dim3(kGraphThreads), side_lds, st, edge_index, E, node_ptr, edge_ptr, B,
                           in_ptr, in_nbr, in_eid, out_ptr, out_nbr, out_eid, cnt_in, cnt_out);
        EVI_LAUNCH_CHECK();
        return EVI_OK;
    }
    if (P == 1) {
        int32_t* cnt_in = static_cast<int32_t*>(workspace);
        int32_t* cnt_out = cnt_in + (N > 0 ? N : 1);
        hipLaunchKernelGGL(k_graph_csr, dim3(B), dim3(kGraphThreads), lds, st, edge_index, E, node_ptr, edge_ptr, in_ptr, in_nbr,
                           in_eid, out_ptr, out_nbr, out_eid, cnt_in, cnt_out, kCsrLdsNodes);
    } else {
        int32_t* part = static_cast<int32_t*>(workspace);  // [2][N][P]
        const int64_t Nn = N > 0 ? N : 1;
        hipLaunchKernelGGL(k_csr_part_count, dim3(B * P), dim3(kGraphThreads), lds, st, edge_index, E, node_ptr, edge_ptr, Nn, P,
                           part, kCsrLdsNodes);
        hipLaunchKernelGGL(k_csr_part_scan, dim3(B), dim3(kGraphThreads), 0, st, edge_index, E, node_ptr, edge_ptr, Nn, P, part,
                           in_ptr, in_nbr, in_eid, out_ptr, out_nbr, out_eid, kCsrLdsNodes);
        hipLaunchKernelGGL(k_csr_part_fill, dim3(B * P), dim3(kGraphThreads), lds, st, edge_index, E, node_ptr, edge_ptr, Nn, P,
                           (const int32_t*)part, in_nbr, in_eid, out_nbr, out_eid, kCsrLdsNodes);
    }
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

// graphs per batch from which DDE runs one workgroup per graph (k_dde_graph) instead of node-parallel over the batch
// (k_dde_round); EVI_DDE_MODE=graph | nodes forces either
static int dde_graph_min_batch() {  // read per call (a getenv is nothing next to a launch): tests switch it in-process
    const char* e = getenv("EVI_DDE_MODE");
    if (e && e[0] == 'g') return 1;
    if (e && e[0] == 'n') return 0x7FFFFFFF;
    // measured crossover on CWQ-shaped graphs (2 + 2 rounds): node-parallel 63 / 82 / 114 / 146 / 266 us at 96 / 128 / 192 / 256 / 512
    // graphs per batch, one workgroup per graph 76 / 67 / 75 / 74 / 137 us
    return 112;
}

extern "C" int evi_dde_node_struct(const float* topic_one_hot, int topic_stride, int num_topics, int64_t N,
                                   const int32_t* in_ptr, const int32_t* in_nbr, const int32_t* out_ptr,
                                   const int32_t* out_nbr, int rounds, int rev_rounds, float* node_struct, void* stream);

extern "C" int evi_dde_node_struct_graphs(const float* topic_one_hot, int topic_stride, int num_topics, int64_t N,
                                          const int64_t* node_ptr, int B, const int32_t* in_ptr, const int32_t* in_nbr,
                                          const int32_t* out_ptr, const int32_t* out_nbr, int rounds, int rev_rounds,
                                          float* node_struct, void* stream) {
    if (node_ptr == nullptr || B < dde_graph_min_batch() || num_topics != 2 || N == 0 || rounds < 0 || rounds > 4 || rev_rounds < 0 ||
        rev_rounds > 4)
        return evi_dde_node_struct(topic_one_hot, topic_stride, num_topics, N, in_ptr, in_nbr, out_ptr, out_nbr, rounds, rev_rounds,
                                   node_struct, stream);  // (also the one place the argument checks live)
    EVI_REQUIRE(topic_stride >= num_topics, "evi_dde_node_struct: topic_one_hot feature dim %d < num_topics=%d", topic_stride, num_topics);
    EVI_REQUIRE(topic_one_hot && in_ptr && out_ptr && node_struct, "evi_dde_node_struct: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static thread_local bool attr = false;
    if (!attr) {
        EVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dde_graph<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kDdeGraphLdsFloats * (int)sizeof(float)));
        attr = true;
    }
    const int S = 1 + rounds + rev_rounds;
    // 144 KiB per workgroup, one graph per CU.  (80 KiB — two CWQ-shaped graphs per CU — was measured: 234 against 136 us per
    // batch of 512, because every graph a little above the average size then falls to the global-memory rounds.)
    const int lds_floats = kDdeGraphLdsFloats;
    hipLaunchKernelGGL(k_dde_graph<2>, dim3((unsigned)B), dim3(kGraphThreads), (size_t)lds_floats * sizeof(float), st, topic_one_hot,
                       topic_stride, node_struct, S, node_ptr, in_ptr, in_nbr, out_ptr, out_nbr, rounds, rev_rounds, lds_floats);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_dde_node_struct(const float* topic_one_hot, int topic_stride, int num_topics, int64_t N,
                                   const int32_t* in_ptr, const int32_t* in_nbr, const int32_t* out_ptr,
                                   const int32_t* out_nbr, int rounds, int rev_rounds, float* node_struct, void* stream) {
    EVI_REQUIRE(N >= 0, "evi_dde_node_struct: N must be >= 0, got %lld", (long long)N);
    EVI_REQUIRE(rounds >= 0 && rounds <= 4 && rev_rounds >= 0 && rev_rounds <= 4,
                "DDE supports at most 4 rounds per direction; got num_rounds=%d, num_reverse_rounds=%d.", rounds,
                rev_rounds);
    if (num_topics != 2)
        return fail(EVI_ERR_INVALID, "num_topics must be 2 (seed vs non-seed), got %d", num_topics);
    EVI_REQUIRE(topic_stride >= num_topics, "evi_dde_node_struct: topic_one_hot feature dim %d < num_topics=%d",
                topic_stride, num_topics);
    if (N == 0) return EVI_OK;
    EVI_REQUIRE(topic_one_hot && in_ptr && out_ptr && node_struct, "evi_dde_node_struct: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int S = 1 + rounds + rev_rounds;
    const int64_t Npad = (N + 63) / 64 * 64;
    const unsigned grid = (unsigned)((2 * Npad + 255) / 256);
    const int steps = rounds > rev_rounds ? rounds : rev_rounds;
    for (int j = 1; j <= (steps > 0 ? steps : 1); ++j) {
        // forward round j: a node averages column j-1 over its in-edges' tails (messages flow src -> dst);
        // reverse round j (edge_index.flip(0)): over its out-edges' heads, restarted from the one-hot
        DdeChain fwd{in_ptr, in_nbr, j - 1, j <= rounds ? j : -1};
        DdeChain rev{out_ptr, out_nbr, j == 1 ? 0 : rounds + j - 1, j <= rev_rounds ? rounds + j : -1};
        hipLaunchKernelGGL(k_dde_round<2>, dim3(grid), dim3(256), 0, st, topic_one_hot, topic_stride, node_struct, S, N, Npad,
                           fwd, rev, j == 1 ? 1 : 0);
    }
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
