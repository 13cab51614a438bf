// BFS levels and shortest-path edge labelling on the per-graph CSR (G2, G3).
//
//   evi_bfs_levels          _bfs_dist, scripts/build_retrieval_pipeline.py:610-631 (multi-source BFS
//                           levels, unreachable = -1), over the undirected adjacency (:570-586) or
//                           one directed half (:589-603)
//   evi_shortest_path_single _shortest_path_single, :453-530 (deterministic single path, G4)
//   evi_shortest_path_pairs _shortest_path_union_mask_by_pair / ..._directed + the edge predicates
//                           _select_shortest_edges_undirected/_directed, :650-815: for every
//                           (seed, answer) pair with a path, the edges u->v with
//                           dist_s[u] + 1 + dist_a[v] == dist_s[a] (either orientation when undirected)
//
// One workgroup per BFS job (a graph and a source set): level-synchronous, levels are separated by workgroup barriers,
// never by launches (a level of a 3 000-node graph is two dependent row gathers deep: spreading one BFS over several
// CUs would pay a cross-CU barrier of ~5 us per level for ~2 us of work).  Graphs of up to 4 096 nodes — every WebQSP / CWQ
// graph — keep their levels AND two frontier queues in LDS: a level touches only the frontier's rows
// (sum_frontier deg * 4 B of neighbours, LDS probes), never the whole dist array; a node is claimed for the next frontier
// by an LDS compare-and-swap, so it is queued once.  Larger graphs fall back to scanning the dist slice per level
// (N_g * 4 B per level).  Integer work, L2-latency-bound.
#include "common.hpp"

#include <stdlib.h>

namespace evi {

constexpr int kBfsThreads = 1024;

constexpr int kBfsHubDegree = 128;  // frontier nodes with more neighbours are expanded by a whole wave
constexpr int kBfsHubCap = 1024;

struct BfsShared {
    int changed;
    int hub_count;
    int hubs[kBfsHubCap];
};

// Frontier-queue BFS for graphs whose levels and queues fit LDS.  dist[0..ng) = -1 except the sources (0), which are
// already in queue[0][0..qcount[0]).  Every thread of the workgroup calls; returns after the first empty frontier.
constexpr int kBfsThreadDegree = 16;   // frontier nodes up to this degree are expanded by the thread that dequeues them
constexpr int kBfsWaveDegree = 512;    // up to this by one wave (lanes stride the rows), above by the whole workgroup
constexpr int kBfsBigCap = 64;

struct BfsQueueShared {
    int qtail;      // entries in the (append-only) queue: a node is queued once in the whole search, so ONE array of N_g holds
    int hub_count;  // every frontier back to back — the current one is [begin, end), the next grows behind it
    int big_count;
    int pad;
    int4 hubs[kBfsHubCap];  // (out begin, out end, in begin, in end) of the node's CSR rows: read once, when it is classified
    int4 big[kBfsBigCap];
};

// CACHED: the graph's CSR rows were copied into LDS by the caller (l_* arrays: row pointers relative to the graph's first
// slot, neighbours as LOCAL node ids) — a level then costs LDS reads only; a frontier of a 3 000-node graph is two dependent
// row gathers deep, ~1.5 us each from the L2, and a search is ~9 such levels.  Otherwise the rows are read from global memory.
template <bool CACHED>
__device__ inline void bfs_block_queue(int32_t* __restrict__ dist, int32_t* __restrict__ q, int64_t n0,
                                       const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_nbr,
                                       const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_nbr, int mode,
                                       BfsQueueShared* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = CACHED ? 0 : (int)n0;          // neighbour entry -> local node id
    const int64_t row0 = CACHED ? 0 : n0;          // local node id -> row pointer index
    int begin = 0;
    for (int level = 0;; ++level) {
        __syncthreads();  // the queue holds the whole frontier
        const int end = sh->qtail;
        const int n = end - begin;
        if (n == 0) break;  // uniform
        if (tid == 0) {
            sh->hub_count = 0;
            sh->big_count = 0;
        }
        __syncthreads();  // every thread has read `end` before the first append of this level
        auto visit = [&](int w) {
            if (dist[w] < 0 && atomicCAS(&dist[w], -1, level + 1) == -1) q[atomicAdd(&sh->qtail, 1)] = w;
        };
        // A row of a frontier node, walked by `width` cooperating threads (this one is number `me`): four entries per
        // trip, the loads of a trip issued before the first LDS compare-and-swap (past-the-end indices are clamped and
        // their values dropped) — a walk that probes after every load runs at one memory latency per neighbour.
        auto walk = [&](const int32_t* __restrict__ nbr, int b, int e, int me, int width) {
            for (int p0 = b + me; p0 < e; p0 += 4 * width) {
                int w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = nbr[p0 + u * width < e ? p0 + u * width : e - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (p0 + u * width < e) visit(w[u] - sub);
            }
        };
        for (int i = begin + tid; i < end; i += kBfsThreads) {
            const int v = q[i];
            const int64_t gv = row0 + v;
            int ob = 0, oe = 0, ib = 0, ie = 0;
            if (mode != 2) {
                ob = out_ptr[gv];
                oe = out_ptr[gv + 1];
            }
            if (mode != 1) {
                ib = in_ptr[gv];
                ie = in_ptr[gv + 1];
            }
            const int deg = (oe - ob) + (ie - ib);
            if (deg > kBfsWaveDegree) {  // power-law hubs: a 2 000-entry row is 30 trips for a wave, 2 for the workgroup
                const int slot = atomicAdd(&sh->big_count, 1);
                if (slot < kBfsBigCap) {
                    sh->big[slot] = make_int4(ob, oe, ib, ie);
                    continue;
                }
            }
            if (deg > kBfsThreadDegree) {
                const int slot = atomicAdd(&sh->hub_count, 1);
                if (slot < kBfsHubCap) {
                    sh->hubs[slot] = make_int4(ob, oe, ib, ie);
                    continue;
                }  // list full: expand it here (correct, slower)
            }
            walk(out_nbr, ob, oe, 0, 1);
            walk(in_nbr, ib, ie, 0, 1);
        }
        __syncthreads();
        const int nh = sh->hub_count < kBfsHubCap ? sh->hub_count : kBfsHubCap;
        for (int h = wave; h < nh; h += kBfsThreads / 64) {  // the row bounds ride in the list: no second trip to the pointers
            const int4 r = sh->hubs[h];
            walk(out_nbr, r.x, r.y, lane, 64);  // an unused direction has an empty range
            walk(in_nbr, r.z, r.w, lane, 64);
        }
        const int nb = sh->big_count < kBfsBigCap ? sh->big_count : kBfsBigCap;
        for (int h = 0; h < nb; ++h) {
            const int4 r = sh->big[h];
            walk(out_nbr, r.x, r.y, tid, kBfsThreads);
            walk(in_nbr, r.z, r.w, tid, kBfsThreads);
        }
        begin = end;
    }
}

// Level-synchronous expansion of the sources already marked 0 in dist[0..ng) (everything else -1), by scanning dist.
// mode 0: undirected (out- and in-rows), 1: follow edges (out-rows), 2: against edges (in-rows).
// A frontier node is expanded by the thread that finds it, unless its rows are long (power-law hubs):
// those are queued in LDS and expanded by one wave each, lanes striding the row.
// Every thread of the workgroup calls; returns after the first empty frontier.
__device__ inline void bfs_block(int32_t* __restrict__ dist, int ng, int64_t n0, const int32_t* __restrict__ in_ptr,
                                 const int32_t* __restrict__ in_nbr, const int32_t* __restrict__ out_ptr,
                                 const int32_t* __restrict__ out_nbr, int mode, BfsShared* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int level = 0;; ++level) {
        if (tid == 0) {
            sh->changed = 0;
            sh->hub_count = 0;
        }
        __syncthreads();
        bool any = false;
        for (int v = tid; v < ng; v += kBfsThreads) {
            if (dist[v] != level) continue;
            const int64_t gv = n0 + v;
            const int deg = (mode != 2 ? out_ptr[gv + 1] - out_ptr[gv] : 0) + (mode != 1 ? in_ptr[gv + 1] - in_ptr[gv] : 0);
            if (deg > kBfsHubDegree) {
                const int slot = atomicAdd(&sh->hub_count, 1);
                if (slot < kBfsHubCap) {
                    sh->hubs[slot] = v;
                    continue;
                }  // queue full: expand it here (correct, slower)
            }
            if (mode != 2)
                for (int p = out_ptr[gv]; p < out_ptr[gv + 1]; ++p) {
                    const int w = out_nbr[p] - (int)n0;
                    if (dist[w] < 0) {
                        dist[w] = level + 1;
                        any = true;
                    }
                }
            if (mode != 1)
                for (int p = in_ptr[gv]; p < in_ptr[gv + 1]; ++p) {
                    const int w = in_nbr[p] - (int)n0;
                    if (dist[w] < 0) {
                        dist[w] = level + 1;
                        any = true;
                    }
                }
        }
        __syncthreads();
        const int nh = sh->hub_count < kBfsHubCap ? sh->hub_count : kBfsHubCap;
        for (int h = wave; h < nh; h += kBfsThreads / 64) {
            const int64_t gv = n0 + sh->hubs[h];
            if (mode != 2)
                for (int p = out_ptr[gv] + lane; p < out_ptr[gv + 1]; p += 64) {
                    const int w = out_nbr[p] - (int)n0;
                    if (dist[w] < 0) {
                        dist[w] = level + 1;
                        any = true;
                    }
                }
            if (mode != 1)
                for (int p = in_ptr[gv] + lane; p < in_ptr[gv + 1]; p += 64) {
                    const int w = in_nbr[p] - (int)n0;
                    if (dist[w] < 0) {
                        dist[w] = level + 1;
                        any = true;
                    }
                }
        }
        if (any) sh->changed = 1;
        __syncthreads();
        if (!sh->changed) break;  // every wave reaches this: the frontier is empty
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBfsThreads) void k_bfs_levels(
    const int32_t* __restrict__ job_graph, const int64_t* __restrict__ src_ptr, const int64_t* __restrict__ src_idx,
    const int64_t* __restrict__ dist_off, const int64_t* __restrict__ node_ptr, const int32_t* __restrict__ in_ptr,
    const int32_t* __restrict__ in_nbr, const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_nbr,
    int mode, int32_t* __restrict__ dist_out, int lds_nodes, int cache_rows) {
    __shared__ BfsShared sh;
    __shared__ BfsQueueShared shq;
    extern __shared__ int32_t lds_dist[];  // [lds_nodes] ints: levels | queue | (cached CSR rows), as far as the graph fits
    const int j = blockIdx.x, tid = threadIdx.x;
    const int g = job_graph[j];
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int ng = (int)(n1 - n0);
    int32_t* out = dist_out + dist_off[j];  // local node id -> level
    int32_t* dist = ng <= lds_nodes ? lds_dist : out;
    const bool queued = 2 * (int64_t)ng <= lds_nodes;  // dist | queue
    // the graph's rows are one contiguous slice of each CSR half: [e0, e1) = [ptr[n0], ptr[n1])
    const int e0 = out_ptr[n0], ne = out_ptr[n1] - e0;
    const int sides = mode == 0 ? 2 : 1;
    const bool cached = cache_rows && queued && ne > 0 && 2 * (int64_t)ng + (int64_t)sides * ((int64_t)ng + 1 + ne) <= lds_nodes;
    if (tid == 0) shq.qtail = 0;
    for (int v = tid; v < ng; v += kBfsThreads) dist[v] = -1;
    int32_t* l_out_ptr = lds_dist + 2 * ng;
    int32_t* l_out_nbr = l_out_ptr + (mode != 2 ? ng + 1 : 0);
    int32_t* l_in_ptr = l_out_nbr + (mode != 2 ? ne : 0);
    int32_t* l_in_nbr = l_in_ptr + (mode != 1 ? ng + 1 : 0);
    if (cached) {  // coalesced copies: row pointers relative to e0, neighbours as local ids
        // eight entries per thread per trip, every load of a trip issued before the first LDS store (an index past the end
        // re-reads the last entry): the copy of a CWQ graph's two halves is ~26 000 ints = 26 per thread — one entry per trip
        // would be 26 dependent memory latencies in front of the search
        auto copy = [&](const int32_t* __restrict__ src, int32_t* dst, int n, int sub) {
            for (int i0 = tid; i0 < n; i0 += 8 * kBfsThreads) {
                int v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[i0 + u * kBfsThreads < n ? i0 + u * kBfsThreads : n - 1];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u * kBfsThreads < n) dst[i0 + u * kBfsThreads] = v[u] - sub;
            }
        };
        if (mode != 2) {
            copy(out_ptr + n0, l_out_ptr, ng + 1, e0);
            copy(out_nbr + e0, l_out_nbr, ne, (int)n0);
        }
        if (mode != 1) {
            const int ie0 = in_ptr[n0];  // = e0: both halves of a graph start at its first edge slot
            copy(in_ptr + n0, l_in_ptr, ng + 1, ie0);
            copy(in_nbr + ie0, l_in_nbr, ne, (int)n0);
        }
    }
    __syncthreads();
    for (int64_t i = src_ptr[j] + tid; i < src_ptr[j + 1]; i += kBfsThreads) {
        const int64_t s = src_idx[i];
        if (s < n0 || s >= n1) continue;  // out-of-range sources are ignored (:619)
        if (!queued)
            dist[s - n0] = 0;
        else if (atomicCAS(&dist[s - n0], -1, 0) == -1)  // a source listed twice is queued once
            lds_dist[ng + atomicAdd(&shq.qtail, 1)] = (int32_t)(s - n0);
    }
    // queue modes: the levels ARE in LDS (2 ng <= lds_nodes) — hand the LDS array itself over, not the LDS-or-global select,
    // so that the compare-and-swaps on the levels compile to ds_cmpst instead of flat atomics
    if (cached)
        bfs_block_queue<true>(lds_dist, lds_dist + ng, n0, l_in_ptr, l_in_nbr, l_out_ptr, l_out_nbr, mode, &shq);
    else if (queued)
        bfs_block_queue<false>(lds_dist, lds_dist + ng, n0, in_ptr, in_nbr, out_ptr, out_nbr, mode, &shq);
    else
        bfs_block(dist, ng, n0, in_ptr, in_nbr, out_ptr, out_nbr, mode, &sh);
    if (dist != out) {
        __syncthreads();
        for (int v = tid; v < ng; v += kBfsThreads) out[v] = dist[v];
    }
}

// ---- edge-parallel BFS: levels in LDS, the edge list in REGISTERS ---------------------------------------------------
// Measured in round 3: with the graph's CSR rows cached in LDS the frontier search above still took ~4.4 us per level (9 levels,
// 40 us per CWQ-shaped graph) — not memory but the frontier machinery: queue appends, compare-and-swaps, three tiers of hub
// expansion, four barriers per level, every LDS access waiting for the one before.  These graphs are SMALL (10^3-10^4 edges):
// each of the 1 024 threads keeps <= 12 edges as packed 16-bit (u, v) pairs in registers for the whole search, the levels
// (16-bit, N_g <= 32 767) sit in LDS, and a level is: read level[u], level[v] of all the thread's edges (24 independent LDS
// reads, issued back to back), then `level[u] == L and level[v] < 0 -> level[v] = L + 1` (both orientations when undirected) as
// plain stores — every writer of a node writes the same value — and ONE barrier.  No CSR, no queue, no hubs, 6 KB of LDS.
// Graphs with more than 12 288 edges or 32 767 nodes take the CSR-based search above, inside the same kernel.
constexpr int kBfsEdgeRegs = 12;

__global__ __launch_bounds__(kBfsThreads) void k_bfs_levels_edges(
    const int32_t* __restrict__ job_graph, const int64_t* __restrict__ src_ptr, const int64_t* __restrict__ src_idx,
    const int64_t* __restrict__ dist_off, const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
    const int64_t* __restrict__ edge_index, int64_t E, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_nbr,
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_nbr, int mode, int32_t* __restrict__ dist_out,
    int lds_nodes) {
    __shared__ BfsShared sh;
    __shared__ BfsQueueShared shq;
    __shared__ int s_changed[3];
    extern __shared__ int32_t lds_dist[];
    const int j = blockIdx.x, tid = threadIdx.x;
    const int g = job_graph[j];
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int64_t e0 = edge_ptr[g], e1 = edge_ptr[g + 1];
    const int ng = (int)(n1 - n0);
    const int64_t ne64 = e1 - e0;
    int32_t* out = dist_out + dist_off[j];  // local node id -> level
    if (ng <= 32767 && ne64 <= (int64_t)kBfsEdgeRegs * kBfsThreads && (ng + 1) / 2 <= lds_nodes) {
        const int ne = (int)ne64;
        constexpr unsigned kInf = 0x7FFFu;  // "no level yet" (levels stay below it: N_g <= 32 767)
        uint16_t* d16 = reinterpret_cast<uint16_t*>(lds_dist);
        for (int v = tid; v < ng; v += kBfsThreads) d16[v] = (uint16_t)kInf;
        if (tid < 3) s_changed[tid] = 0;
        // this thread's edges, all loads issued before the first use; an edge past the end, or with an endpoint outside the graph
        // (validated upstream), becomes the self loop (0, 0), which no level can cross
        uint32_t pr[kBfsEdgeRegs];
        {
            int64_t a[kBfsEdgeRegs], b[kBfsEdgeRegs];
#pragma unroll
            for (int r = 0; r < kBfsEdgeRegs; ++r) {
                const int idx = tid + r * kBfsThreads;
                const int64_t ee = e0 + (idx < ne ? idx : (ne > 0 ? ne - 1 : 0));
                a[r] = ne > 0 ? edge_index[ee] : 0;
                b[r] = ne > 0 ? edge_index[E + ee] : 0;
            }
#pragma unroll
            for (int r = 0; r < kBfsEdgeRegs; ++r) {
                const bool ok = tid + r * kBfsThreads < ne && a[r] >= n0 && a[r] < n1 && b[r] >= n0 && b[r] < n1;
                pr[r] = ok ? ((uint32_t)(a[r] - n0) | ((uint32_t)(b[r] - n0) << 16)) : 0u;
            }
        }
        unsigned live = 0;  // bit r: edge r can still discover a node (self loops and padding never can)
#pragma unroll
        for (int r = 0; r < kBfsEdgeRegs; ++r)
            if ((pr[r] & 0xFFFFu) != (pr[r] >> 16)) live |= 1u << r;
        __syncthreads();
        for (int64_t i = src_ptr[j] + tid; i < src_ptr[j + 1]; i += kBfsThreads) {
            const int64_t s0 = src_idx[i];
            if (s0 >= n0 && s0 < n1) d16[s0 - n0] = 0;  // out-of-range sources are ignored (:619)
        }
        __syncthreads();
        // Where the time goes (wall_clock64 stamps of thread 0 on a CWQ-shaped graph, round 3): 4.0 us to load the edges and
        // initialise, then 1.7-2.7 us per level — ~350 instructions per wave and level with four waves per SIMD, i.e. instruction
        // issue on ONE CU, not LDS or memory; plus ~9 us of launch and drain around the kernel.  Two things trim the work per
        // level without changing that picture (34.3 against 34.7 us per batch of 32): an edge whose endpoints both carry a level is
        // dead, and a register slot that is dead in all 64 lanes is skipped by a scalar branch; the undirected test is
        // min == L and max == INF on the pair of levels — one comparison chain and one store instead of two.
        for (int level = 0;; ++level) {
            // three rotating flags: level L raises flag[L % 3]; flag[(L + 1) % 3] is cleared meanwhile — nobody reads it before
            // the end of level L + 1, and the flag read at the end of level L - 1 is a different one
            if (tid == 0) s_changed[(level + 1) % 3] = 0;
            bool any = false;
#pragma unroll
            for (int r = 0; r < kBfsEdgeRegs; ++r) {
                const bool alive = (live >> r) & 1u;
                if (__ballot(alive) == 0ull) continue;  // wave-uniform: nobody's slot r can discover anything any more
                const unsigned ua = pr[r] & 0xFFFFu, va = pr[r] >> 16;
                unsigned du = kInf, dv = kInf;
                if (alive) {
                    du = d16[ua];
                    dv = d16[va];
                }
                if (!alive) continue;
                const unsigned lo = du < dv ? du : dv, hi = du < dv ? dv : du;
                if (hi != kInf) {  // both settled: dead from now on
                    live &= ~(1u << r);
                    continue;
                }
                if (lo != (unsigned)level) continue;
                // exactly one endpoint is on the frontier, the other has no level yet: does the orientation allow the step?
                const bool from_u = du == (unsigned)level;
                if ((mode == 1 && !from_u) || (mode == 2 && from_u)) continue;
                d16[from_u ? va : ua] = (uint16_t)(level + 1);
                any = true;
            }
            if (any) s_changed[level % 3] = 1;
            __syncthreads();
            if (!s_changed[level % 3]) break;  // uniform: the frontier did not grow
        }
        for (int v = tid; v < ng; v += kBfsThreads) {
            const unsigned d = d16[v];
            out[v] = d == kInf ? -1 : (int32_t)d;
        }
        return;
    }
    // ---- beyond the LDS budget: the CSR-based search (rows from the L2), as k_bfs_levels
    int32_t* dist = ng <= lds_nodes ? lds_dist : out;
    const bool queued = 2 * (int64_t)ng <= lds_nodes;
    if (tid == 0) shq.qtail = 0;
    for (int v = tid; v < ng; v += kBfsThreads) dist[v] = -1;
    __syncthreads();
    for (int64_t i = src_ptr[j] + tid; i < src_ptr[j + 1]; i += kBfsThreads) {
        const int64_t s0 = src_idx[i];
        if (s0 < n0 || s0 >= n1) continue;
        if (!queued)
            dist[s0 - n0] = 0;
        else if (atomicCAS(&dist[s0 - n0], -1, 0) == -1)
            lds_dist[ng + atomicAdd(&shq.qtail, 1)] = (int32_t)(s0 - n0);
    }
    if (queued)
        bfs_block_queue<false>(lds_dist, lds_dist + ng, n0, in_ptr, in_nbr, out_ptr, out_nbr, mode, &shq);
    else
        bfs_block(dist, ng, n0, in_ptr, in_nbr, out_ptr, out_nbr, mode, &sh);
    if (dist != out) {
        __syncthreads();
        for (int v = tid; v < ng; v += kBfsThreads) out[v] = dist[v];
    }
}

// One workgroup per job (graph, source set, target set): the reference's deterministic single shortest
// path (_shortest_path_single, :453-530).  Its FIFO BFS over (neighbour id, edge id)-sorted adjacency
// visits each level in the lexicographic order of the tree paths, so the path it returns is the
// lexicographically smallest node sequence among ALL shortest source->target paths, towards the
// nearest target (ties: smallest id), and each hop uses the smallest edge id between its two nodes.
// That characterisation needs no queue: BFS from the sources (ds), pick the target, BFS from it (dt),
// then walk from the smallest source with dt == D, each hop taking the smallest neighbour with
// dt == D - i (min over the CSR row of (neighbour, edge id)).
__global__ __launch_bounds__(kBfsThreads) void k_shortest_path_single(
    const int32_t* __restrict__ job_graph, const int64_t* __restrict__ src_ptr, const int64_t* __restrict__ src_idx,
    const int64_t* __restrict__ tgt_ptr, const int64_t* __restrict__ tgt_idx, const int64_t* __restrict__ dist_off,
    const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr, const int32_t* __restrict__ in_ptr,
    const int32_t* __restrict__ in_nbr, const int32_t* __restrict__ in_eid, const int32_t* __restrict__ out_ptr,
    const int32_t* __restrict__ out_nbr, const int32_t* __restrict__ out_eid, int32_t* __restrict__ dist_ws, int path_cap,
    int32_t* __restrict__ out_len, int64_t* __restrict__ out_nodes, int64_t* __restrict__ out_edges) {
    __shared__ BfsShared sh;
    __shared__ unsigned long long best;
    const int j = blockIdx.x, tid = threadIdx.x;
    const int g = job_graph[j];
    const int64_t n0 = node_ptr[g], n1 = node_ptr[g + 1], e0 = edge_ptr[g];
    const int ng = (int)(n1 - n0);
    int32_t* ds = dist_ws + dist_off[j];
    int32_t* dt = ds + ng;
    int64_t* nodes = out_nodes + (int64_t)j * (path_cap + 1);
    int64_t* edges = out_edges + (int64_t)j * path_cap;
    constexpr unsigned long long kNone = ~0ull;
    for (int v = tid; v < ng; v += kBfsThreads) ds[v] = dt[v] = -1;
    if (tid == 0) best = kNone;
    __syncthreads();
    for (int64_t i = src_ptr[j] + tid; i < src_ptr[j + 1]; i += kBfsThreads) {
        const int64_t s = src_idx[i];
        if (s >= n0 && s < n1) ds[s - n0] = 0;
    }
    bfs_block(ds, ng, n0, in_ptr, in_nbr, out_ptr, out_nbr, 0, &sh);
    // nearest reachable target, smallest id on ties (:501-508)
    for (int64_t i = tgt_ptr[j] + tid; i < tgt_ptr[j + 1]; i += kBfsThreads) {
        const int64_t t = tgt_idx[i];
        if (t < n0 || t >= n1) continue;
        const int d = ds[t - n0];
        if (d >= 0) atomicMin(&best, ((unsigned long long)d << 32) | (unsigned long long)(t - n0));
    }
    __syncthreads();
    const unsigned long long tb = best;
    __syncthreads();
    if (tb == kNone) {  // no sources, no targets, or no path (uniform across the workgroup)
        if (tid == 0) out_len[j] = -1;
        return;
    }
    const int D = (int)(tb >> 32), target = (int)(tb & 0xffffffffu);
    if (tid == 0) {
        dt[target] = 0;
        best = kNone;
    }
    bfs_block(dt, ng, n0, in_ptr, in_nbr, out_ptr, out_nbr, 0, &sh);
    for (int v = tid; v < ng; v += kBfsThreads)
        if (ds[v] == 0 && dt[v] == D) atomicMin(&best, (unsigned long long)v);
    __syncthreads();
    int cur = (int)best;
    __syncthreads();
    if (tid == 0) nodes[0] = cur;
    for (int i = 1; i <= D; ++i) {
        if (tid == 0) best = kNone;
        __syncthreads();
        const int64_t gv = n0 + cur;
        for (int p = out_ptr[gv] + tid; p < out_ptr[gv + 1]; p += kBfsThreads) {
            const int w = out_nbr[p] - (int)n0;
            if (dt[w] == D - i && ds[w] == i)
                atomicMin(&best, ((unsigned long long)w << 32) | (unsigned long long)(out_eid[p] - e0));
        }
        for (int p = in_ptr[gv] + tid; p < in_ptr[gv + 1]; p += kBfsThreads) {
            const int w = in_nbr[p] - (int)n0;
            if (dt[w] == D - i && ds[w] == i)
                atomicMin(&best, ((unsigned long long)w << 32) | (unsigned long long)(in_eid[p] - e0));
        }
        __syncthreads();
        const unsigned long long nb = best;
        __syncthreads();
        cur = (int)(nb >> 32);
        if (tid == 0 && i <= path_cap) {
            nodes[i] = cur;
            edges[i - 1] = (int64_t)(nb & 0xffffffffu);
        }
    }
    if (tid == 0) out_len[j] = D;
}

// One workgroup per dense pair slot p = (graph g, i-th seed job of g, j-th answer job of g).
//   pair_seed_job[p], pair_answer_job[p]: BFS job ids whose dist slices are dist_from / dist_to
//   pair_answer_node[p]: batch-global answer node
// PASS 0: pair_len[p] = dist_s[a] (or -1), pair_edge_count[p], mask |= edges.  PASS 1: write the
// pair's edge ids (batch-global, ascending) at pair_edge_ids[pair_edge_off[p] ...].
template <int PASS>
__global__ __launch_bounds__(kBfsThreads) void k_shortest_path_pairs(
    const int32_t* __restrict__ pair_graph, const int32_t* __restrict__ pair_seed_job,
    const int32_t* __restrict__ pair_answer_job, const int64_t* __restrict__ pair_answer_node,
    const int64_t* __restrict__ dist_off, const int32_t* __restrict__ dist, const int64_t* __restrict__ edge_index,
    int64_t E, const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr, int directed,
    int32_t* __restrict__ pair_len, int32_t* __restrict__ pair_edge_count, uint8_t* __restrict__ mask,
    const int64_t* __restrict__ pair_edge_off, int64_t* __restrict__ pair_edge_ids) {
    __shared__ int scan[kBfsThreads];
    __shared__ int base;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int g = pair_graph[p];
    const int64_t n0 = node_ptr[g];
    const int64_t e0 = edge_ptr[g], e1 = edge_ptr[g + 1];
    const int32_t* ds = dist + dist_off[pair_seed_job[p]];
    const int32_t* da = dist + dist_off[pair_answer_job[p]];
    const int dsa = ds[pair_answer_node[p] - n0];
    if (PASS == 0 && tid == 0) pair_len[p] = dsa;
    if (dsa < 0) {
        if (PASS == 0 && tid == 0) pair_edge_count[p] = 0;
        return;
    }
    if (tid == 0) base = 0;
    __syncthreads();
    for (int64_t c0 = e0; c0 < e1; c0 += kBfsThreads) {
        const int64_t e = c0 + tid;
        bool keep = false;
        if (e < e1) {
            const int u = (int)(edge_index[e] - n0), v = (int)(edge_index[E + e] - n0);
            const int su = ds[u], av = da[v];
            keep = su >= 0 && av >= 0 && su + 1 + av == dsa;
            if (!directed && !keep) {
                const int sv = ds[v], au = da[u];
                keep = sv >= 0 && au >= 0 && sv + 1 + au == dsa;
            }
        }
        if (PASS == 0) {
            if (keep) mask[e] = 1;
            // count
            scan[tid] = keep ? 1 : 0;
            __syncthreads();
            for (int off = kBfsThreads >> 1; off > 0; off >>= 1) {
                if (tid < off) scan[tid] += scan[tid + off];
                __syncthreads();
            }
            if (tid == 0) base += scan[0];
            __syncthreads();
        } else {
            // ordered compaction: inclusive scan of the keep flags
            scan[tid] = keep ? 1 : 0;
            __syncthreads();
            for (int off = 1; off < kBfsThreads; off <<= 1) {
                const int add = tid >= off ? scan[tid - off] : 0;
                __syncthreads();
                scan[tid] += add;
                __syncthreads();
            }
            if (keep) pair_edge_ids[pair_edge_off[p] + base + scan[tid] - 1] = e;
            __syncthreads();
            if (tid == kBfsThreads - 1) base += scan[tid];
            __syncthreads();
        }
    }
    if (PASS == 0 && tid == 0) pair_edge_count[p] = base;
}

}  // namespace evi

using namespace evi;

extern "C" int evi_bfs_levels(const int32_t* job_graph, const int64_t* src_ptr, const int64_t* src_idx,
                              const int64_t* dist_off, int num_jobs, const int64_t* node_ptr, const int32_t* in_ptr,
                              const int32_t* in_nbr, const int32_t* out_ptr, const int32_t* out_nbr, int mode,
                              int32_t* dist_out, void* stream) {
    EVI_REQUIRE(num_jobs >= 0, "evi_bfs_levels: num_jobs must be >= 0");
    EVI_REQUIRE(mode >= 0 && mode <= 2, "evi_bfs_levels: mode must be 0 (undirected), 1 (forward) or 2 (backward)");
    if (num_jobs == 0) return EVI_OK;
    EVI_REQUIRE(job_graph && src_ptr && dist_off && node_ptr && in_ptr && out_ptr && dist_out, "evi_bfs_levels: null pointer");
    // Two launch shapes.  LATENCY (a search per CU or fewer: num_jobs <= 256, the reference's batch of 32 graphs with its ~150
    // seed / answer jobs): 136 KiB of dynamic LDS (+ 21 KiB static) hold the levels, the queue AND the CSR rows of a CWQ-sized
    // graph (3 000 nodes, 10 000 edges, both halves: 32 002 ints) for the whole search, one workgroup per CU.  THROUGHPUT (more
    // jobs than CUs): 48 KiB, three workgroups per CU hide each other's row gathers, rows read from the L2 — caching them would
    // cost two thirds of the resident searches (measured at 512 jobs: 70 us uncached x 3 per CU against 87 us cached x 1).
    // EVI_BFS_CACHE=0 / 1 forces either.
    constexpr int kLdsBig = 34816, kLdsSmall = 12288;
    bool big = num_jobs <= 256;
    if (const char* e = getenv("EVI_BFS_CACHE")) big = e[0] == '1';
    const int lds_nodes = big ? kLdsBig : kLdsSmall;
    static thread_local bool attr = false;
    if (!attr) {
        EVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bfs_levels), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kLdsBig * (int)sizeof(int32_t)));
        attr = true;
    }
    hipLaunchKernelGGL(k_bfs_levels, dim3(num_jobs), dim3(kBfsThreads), lds_nodes * sizeof(int32_t),
                       reinterpret_cast<hipStream_t>(stream), job_graph, src_ptr, src_idx, dist_off, node_ptr, in_ptr, in_nbr,
                       out_ptr, out_nbr, mode, dist_out, lds_nodes, big ? 1 : 0);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_bfs_levels_edges(const int32_t* job_graph, const int64_t* src_ptr, const int64_t* src_idx,
                                    const int64_t* dist_off, int num_jobs, const int64_t* node_ptr, const int64_t* edge_ptr,
                                    const int64_t* edge_index, int64_t E, const int32_t* in_ptr, const int32_t* in_nbr,
                                    const int32_t* out_ptr, const int32_t* out_nbr, int mode, int32_t* dist_out, void* stream) {
    EVI_REQUIRE(num_jobs >= 0, "evi_bfs_levels_edges: num_jobs must be >= 0");
    EVI_REQUIRE(mode >= 0 && mode <= 2, "evi_bfs_levels_edges: mode must be 0 (undirected), 1 (forward) or 2 (backward)");
    if (num_jobs == 0) return EVI_OK;
    EVI_REQUIRE(job_graph && src_ptr && dist_off && node_ptr && edge_ptr && in_ptr && out_ptr && dist_out, "evi_bfs_levels_edges: null pointer");
    EVI_REQUIRE(E == 0 || edge_index, "evi_bfs_levels_edges: null edge_index");
    if (const char* e = getenv("EVI_BFS_EDGES"))  // A/B runs: 0 = the CSR-based search for every graph
        if (e[0] == '0')
            return evi_bfs_levels(job_graph, src_ptr, src_idx, dist_off, num_jobs, node_ptr, in_ptr, in_nbr, out_ptr, out_nbr, mode, dist_out, stream);
    constexpr int kLds = 12288;  // 48 KiB of dynamic LDS for the graphs that take the CSR-based path; the edge-parallel search
                                 // needs 2 bytes per node of it (its edges live in registers)
    hipLaunchKernelGGL(k_bfs_levels_edges, dim3(num_jobs), dim3(kBfsThreads), kLds * sizeof(int32_t), reinterpret_cast<hipStream_t>(stream),
                       job_graph, src_ptr, src_idx, dist_off, node_ptr, edge_ptr, edge_index, E, in_ptr, in_nbr, out_ptr, out_nbr, mode,
                       dist_out, kLds);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_shortest_path_single(const int32_t* job_graph, const int64_t* src_ptr, const int64_t* src_idx,
                                        const int64_t* tgt_ptr, const int64_t* tgt_idx, const int64_t* dist_off,
                                        int num_jobs, const int64_t* node_ptr, const int64_t* edge_ptr,
                                        const int32_t* in_ptr, const int32_t* in_nbr, const int32_t* in_eid,
                                        const int32_t* out_ptr, const int32_t* out_nbr, const int32_t* out_eid,
                                        int32_t* dist_ws, int path_cap, int32_t* out_len, int64_t* out_nodes,
                                        int64_t* out_edges, void* stream) {
    EVI_REQUIRE(num_jobs >= 0 && path_cap >= 1, "evi_shortest_path_single: need num_jobs >= 0 and path_cap >= 1");
    if (num_jobs == 0) return EVI_OK;
    EVI_REQUIRE(job_graph && src_ptr && tgt_ptr && dist_off && node_ptr && edge_ptr && in_ptr && out_ptr && dist_ws &&
                    out_len && out_nodes && out_edges,
                "evi_shortest_path_single: null pointer");
    hipLaunchKernelGGL(k_shortest_path_single, dim3(num_jobs), dim3(kBfsThreads), 0, reinterpret_cast<hipStream_t>(stream),
                       job_graph, src_ptr, src_idx, tgt_ptr, tgt_idx, dist_off, node_ptr, edge_ptr, in_ptr, in_nbr, in_eid,
                       out_ptr, out_nbr, out_eid, dist_ws, path_cap, out_len, out_nodes, out_edges);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_shortest_path_pairs(int pass, const int32_t* pair_graph, const int32_t* pair_seed_job,
                                       const int32_t* pair_answer_job, const int64_t* pair_answer_node, int num_pairs,
                                       const int64_t* dist_off, const int32_t* dist, const int64_t* edge_index,
                                       int64_t E, const int64_t* node_ptr, const int64_t* edge_ptr, int directed,
                                       int32_t* pair_len, int32_t* pair_edge_count, uint8_t* edge_mask,
                                       const int64_t* pair_edge_off, int64_t* pair_edge_ids, void* stream) {
    EVI_REQUIRE(pass == 0 || pass == 1, "evi_shortest_path_pairs: pass must be 0 (count + mask) or 1 (fill)");
    EVI_REQUIRE(num_pairs >= 0 && E >= 0, "evi_shortest_path_pairs: bad sizes");
    if (num_pairs == 0) return EVI_OK;
    EVI_REQUIRE(pair_graph && pair_seed_job && pair_answer_job && pair_answer_node && dist_off && dist && node_ptr &&
                    edge_ptr && (E == 0 || edge_index),
                "evi_shortest_path_pairs: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (pass == 0) {
        EVI_REQUIRE(pair_len && pair_edge_count && (E == 0 || edge_mask), "evi_shortest_path_pairs: null output");
        hipLaunchKernelGGL(k_shortest_path_pairs<0>, dim3(num_pairs), dim3(kBfsThreads), 0, st, pair_graph, pair_seed_job,
                           pair_answer_job, pair_answer_node, dist_off, dist, edge_index, E, node_ptr, edge_ptr, directed,
                           pair_len, pair_edge_count, edge_mask, pair_edge_off, pair_edge_ids);
    } else {
        EVI_REQUIRE(pair_edge_off && pair_edge_ids, "evi_shortest_path_pairs: null output");
        hipLaunchKernelGGL(k_shortest_path_pairs<1>, dim3(num_pairs), dim3(kBfsThreads), 0, st, pair_graph, pair_seed_job,
                           pair_answer_job, pair_answer_node, dist_off, dist, edge_index, E, node_ptr, edge_ptr, directed,
                           pair_len, pair_edge_count, edge_mask, pair_edge_off, pair_edge_ids);
    }
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
