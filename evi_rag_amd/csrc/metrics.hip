// Per-graph ranking metrics of the retriever evaluation, fused: one workgroup per question graph
// takes the graph's edge scores once, selects the exact top-k_max (score desc, position asc) in
// LDS and derives every metric of the k window from that one ranked list.
//
//   edge/recall@k          EdgeRecallAtK._update_graph_recall, src/metrics/retriever_metrics.py:132-157
//   answer/reachability@k  AnswerReachability._compute_reachability_at_k (undirected union-find with
//                          checkpoints), src/metrics/reachability.py:330-381; validity rules :129-179
//   answer_hit@k, answer_recall@k   _oracle_metrics_for_sample, src/models/reasoner_module.py:17-68;
//                          compute_answer_hit / compute_answer_recall, src/utils/metrics.py:167-238
//   edge/score_margin      ScoreMargin._update_graph_margin, src/metrics/retriever_metrics.py:376-391
//   top-k lists            RetrieverTopKEdgeWriter._select_topk_edges,
//                          src/callbacks/retriever_topk_edge_writer.py:294-320
//
// Per-graph values are written to [B, nk] arrays; the host mirror adds them up in graph order (the
// reference's own accumulation order), so nothing here uses float atomics.  HBM-bound: the scores,
// labels and endpoints of each graph are read once (E * (4 + 1) bytes + 16 B per ranked edge).
#include "common.hpp"

namespace evi {

constexpr int kMaxKValues = 16;
constexpr int kUfLdsNodes = 8192;  // union-find parents + flags kept in LDS up to this many nodes (64 KiB)
constexpr int kMaxAnswers = 2048;

struct KWindow {
    int nk;
    int k[kMaxKValues];  // ascending, positive
};

struct MetricsArgs {
    const float* scores;          // [E]
    const uint8_t* target;        // [E] 0/1 (labels > 0.5), may be NULL
    const int64_t* edge_index;    // [2, E] batch-global node ids
    int64_t E;
    const int64_t* edge_ptr;      // [B+1]
    const int64_t* node_ptr;      // [B+1]
    const int64_t* q_idx;         // seeds, batch-global (q_local_indices after PyG collate)
    const int64_t* q_ptr;         // [B+1]
    const int64_t* a_idx;         // answers, batch-global
    const int64_t* a_ptr;         // [B+1]
    const int64_t* node_global_ids;  // [N] entity ids, may be NULL
    const int64_t* answer_ids;       // entity ids of the answers, may be NULL
    const int64_t* answer_ptr;       // [B+1]
    KWindow kw;
    int k_max;
    float* edge_recall;      // [B, nk]
    uint8_t* recall_valid;   // [B]  graph has >= 1 edge
    uint8_t* reach;          // [B, nk]
    uint8_t* reach_valid;    // [B]
    uint8_t* answer_hit;     // [B, nk]
    float* answer_recall;    // [B, nk]
    uint8_t* answer_valid;   // [B]
    float* score_margin;     // [B]
    uint8_t* margin_valid;   // [B]
    int32_t* topk_index;     // [B, k_max] local edge position, -1 padding (may be NULL)
    float* topk_score;       // [B, k_max] (may be NULL)
    int32_t* topk_count;     // [B] (may be NULL)
    int32_t* uf_ws;          // [2 N] union-find parents + flags for graphs too large for LDS
};

struct MetricsShared {
    SelectShared sel;
    int parent[kUfLdsNodes];
    int flag[kUfLdsNodes];
    int eu[EVI_TOPK_MAX_K], ev[EVI_TOPK_MAX_K];
    int first_rank[kMaxAnswers];
    int misc[8];  // 0: positives, 1: reach rank
    unsigned int s_min_pos, s_max_neg;
};

__device__ inline int uf_find(int* parent, int x) {
    while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
    }
    return x;
}

__global__ __launch_bounds__(kSelectThreads) void k_retriever_metrics(MetricsArgs a) {
    extern __shared__ char smem_raw[];
    MetricsShared& sh = *reinterpret_cast<MetricsShared*>(smem_raw);
    const int g = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int nk = a.kw.nk;
    const int64_t e0 = a.edge_ptr[g], e1 = a.edge_ptr[g + 1];
    const int64_t cnt = e1 > e0 ? e1 - e0 : 0;
    const int64_t n0 = a.node_ptr[g], n1 = a.node_ptr[g + 1];
    const int num_nodes = (int)(n1 > n0 ? n1 - n0 : 0);
    const float* s = a.scores + e0;

    if (tid < 8) sh.misc[tid] = 0;
    if (tid == 1) sh.misc[1] = 0x7FFFFFFF;
    __syncthreads();

    // ---- ranked list ---------------------------------------------------------------------------
    auto load = [&](int64_t i) -> uint64_t { return make_key(s[i], (uint32_t)i); };
    const int m = block_topk(sh.sel, load, cnt, a.k_max);  // sh.sel.keys[0..m) sorted, barrier done
    for (int i = tid; i < a.k_max; i += nt) {
        const bool ok = i < m;
        const uint32_t pos = ok ? key_index(sh.sel.keys[i]) : 0u;
        if (a.topk_index) a.topk_index[(int64_t)g * a.k_max + i] = ok ? (int32_t)pos : -1;
        if (a.topk_score) a.topk_score[(int64_t)g * a.k_max + i] = ok ? s[pos] : -INFINITY;
    }
    if (a.topk_count && tid == 0) a.topk_count[g] = m;

    // ---- edge recall@k + score margin -----------------------------------------------------------
    if (a.target) {
        int local_pos = 0;
        float min_pos = INFINITY, max_neg = -INFINITY;
        for (int64_t i = tid; i < cnt; i += nt) {
            const bool p = a.target[e0 + i] != 0;
            local_pos += p ? 1 : 0;
            const float v = s[i];
            if (p) min_pos = fminf(min_pos, v); else max_neg = fmaxf(max_neg, v);
        }
        // block reductions through LDS atomics (ints) and ordered-uint atomics (floats)
        if (local_pos) atomicAdd(&sh.misc[0], local_pos);
        if (tid == 0) {
            sh.s_min_pos = float_to_ordered(INFINITY);
            sh.s_max_neg = float_to_ordered(-INFINITY);
        }
        __syncthreads();
        atomicMin(&sh.s_min_pos, float_to_ordered(min_pos));
        atomicMax(&sh.s_max_neg, float_to_ordered(max_neg));
        __syncthreads();
        const int positives = sh.misc[0];
        if (tid == 0) {
            const bool has_pos = positives > 0, has_neg = positives < cnt;
            a.margin_valid[g] = (has_pos && has_neg) ? 1 : 0;
            a.score_margin[g] = (has_pos && has_neg) ? ordered_to_float(sh.s_min_pos) - ordered_to_float(sh.s_max_neg) : 0.f;
            a.recall_valid[g] = cnt > 0 ? 1 : 0;
        }
        // hits among the first min(k, m) ranked edges: one wave per k value
        const int lane = tid & 63, wave = tid >> 6;
        for (int ki = wave; ki < nk; ki += nt / 64) {
            const int k_eff = a.kw.k[ki] < m ? a.kw.k[ki] : m;
            int hits = 0;
            for (int i = lane; i < k_eff; i += 64) hits += a.target[e0 + key_index(sh.sel.keys[i])] ? 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) hits += __shfl_xor(hits, off, 64);
            if (lane == 0) {
                const float denom = positives > 0 ? (float)positives : 1.0f;
                a.edge_recall[(int64_t)g * nk + ki] = cnt > 0 ? (float)hits / denom : 0.f;
            }
        }
    }

    else if (tid == 0) {
        a.recall_valid[g] = 0;
        a.margin_valid[g] = 0;
    }

    // ---- answer reachability@k --------------------------------------------------------------------
    {
        const int64_t q0 = a.q_ptr[g], q1 = a.q_ptr[g + 1], a0 = a.a_ptr[g], a1 = a.a_ptr[g + 1];
        const bool in_lds = num_nodes <= kUfLdsNodes;
        int* parent = in_lds ? sh.parent : (a.uf_ws + 2 * n0);
        int* flag = in_lds ? sh.flag : (a.uf_ws + 2 * n0 + num_nodes);  // bit 0: seed, bit 1: answer (valid at roots)
        for (int v = tid; v < num_nodes; v += nt) {
            parent[v] = v;
            flag[v] = 0;
        }
        // endpoints of the ranked edges, as local node ids, staged in LDS by all threads
        for (int i = tid; i < m; i += nt) {
            const int64_t e = e0 + key_index(sh.sel.keys[i]);
            sh.eu[i] = (int)(a.edge_index[e] - n0);
            sh.ev[i] = (int)(a.edge_index[a.E + e] - n0);
        }
        if (tid == 0) sh.misc[2] = sh.misc[3] = 0;
        __syncthreads();
        for (int64_t i = q0 + tid; i < q1; i += nt) {
            const int64_t v = a.q_idx[i];
            if (v >= n0 && v < n1) {
                atomicOr(&flag[v - n0], 1);
                sh.misc[2] = 1;
            }
        }
        for (int64_t i = a0 + tid; i < a1; i += nt) {
            const int64_t v = a.a_idx[i];
            if (v >= n0 && v < n1) {
                atomicOr(&flag[v - n0], 2);
                sh.misc[3] = 1;
            }
        }
        __syncthreads();
        // valid iff edges, nodes, and at least one in-range seed and answer (reachability.py:129-179)
        const bool valid = cnt > 0 && num_nodes > 0 && sh.misc[2] && sh.misc[3];
        if (valid) {  // a seed that is itself an answer is reachable with zero edges
            for (int v = tid; v < num_nodes; v += nt)
                if (flag[v] == 3) sh.misc[1] = 0;
        }
        __syncthreads();
        if (tid == 0) a.reach_valid[g] = valid ? 1 : 0;
        if (valid && sh.misc[1] != 0) {  // block-uniform
            // Smallest number of ranked edges after which a seed's component holds an answer.  Connectivity only: union order
            // does not change the answer — so the ranked edges are united 64 at a time by 64 lanes (lock-free hooking of the
            // larger root under the smaller: atomicCAS on the parent words), the static seed / answer bits of every node are OR-ed
            // into bits 2-3 of its root, and the first batch after which some root holds both is found in m / 64 rounds instead of
            // m dependent LDS round trips by one thread (0.12 ms of this kernel at m = 500).  Only that batch is then replayed
            // edge by edge, on the state rebuilt from the batches before it, for the exact rank.
            constexpr int kBatch = 64;
            auto unite_batch = [&](int b0) {
                const int i = b0 + tid;
                if (tid < kBatch && i < m) {
                    const int u = sh.eu[i], v = sh.ev[i];
                    if (u >= 0 && v >= 0 && u < num_nodes && v < num_nodes) {
                        int ru = uf_find(parent, u), rv = uf_find(parent, v);
                        while (ru != rv) {
                            const int hi = ru > rv ? ru : rv, lo = ru > rv ? rv : ru;
                            if (atomicCAS(&parent[hi], hi, lo) == hi) break;  // hooked; else another lane moved it: look again
                            ru = uf_find(parent, u);
                            rv = uf_find(parent, v);
                        }
                    }
                }
            };
            auto gather_root_bits = [&]() {  // bits 2-3 of a root = OR of the static bits (0-1) of its component's nodes
                for (int v = tid; v < num_nodes; v += nt) {
                    const int f = flag[v] & 3;
                    if (f) atomicOr(&flag[uf_find(parent, v)], f << 2);
                }
            };
            int found_batch = -1;
            for (int b0 = 0; b0 < m; b0 += kBatch) {
                unite_batch(b0);
                __syncthreads();
                gather_root_bits();
                __syncthreads();
                for (int v = tid; v < num_nodes; v += nt)
                    if ((flag[v] >> 2) == 3) sh.misc[4] = 1;
                __syncthreads();
                if (sh.misc[4]) {
                    found_batch = b0;
                    break;
                }
                for (int v = tid; v < num_nodes; v += nt) flag[v] &= 3;
                __syncthreads();
            }
            if (found_batch >= 0) {  // block-uniform: rebuild the state before that batch, then walk the batch in rank order
                for (int v = tid; v < num_nodes; v += nt) {
                    parent[v] = v;
                    flag[v] &= 3;
                }
                __syncthreads();
                for (int b0 = 0; b0 < found_batch; b0 += kBatch) {
                    unite_batch(b0);
                    __syncthreads();
                }
                gather_root_bits();
                __syncthreads();
                if (tid == 0) {
                    int reach_rank = 0x7FFFFFFF;
                    const int end = found_batch + kBatch < m ? found_batch + kBatch : m;
                    for (int i = found_batch; i < end; ++i) {
                        const int u = sh.eu[i], v = sh.ev[i];
                        if (u < 0 || v < 0 || u >= num_nodes || v >= num_nodes) continue;
                        const int ru = uf_find(parent, u), rv = uf_find(parent, v);
                        if (ru == rv) continue;
                        parent[rv] = ru;
                        flag[ru] |= flag[rv] & 12;
                        if ((flag[ru] >> 2) == 3) {
                            reach_rank = i + 1;
                            break;
                        }
                    }
                    sh.misc[1] = reach_rank;
                }
            } else if (tid == 0) {
                sh.misc[1] = 0x7FFFFFFF;
            }
        }
        __syncthreads();
        if (tid < nk) {
            const int k_eff = a.kw.k[tid] < m ? a.kw.k[tid] : m;
            a.reach[(int64_t)g * nk + tid] = (valid && sh.misc[1] <= k_eff && k_eff > 0) ? 1 : 0;
        }
    }

    // ---- answer hit@k / answer recall@k over entity ids ---------------------------------------------
    if (a.node_global_ids && a.answer_ids) {
        const int64_t b0 = a.answer_ptr[g], b1 = a.answer_ptr[g + 1];
        const bool too_many = (b1 - b0) > kMaxAnswers;  // reported as answer_valid = 2; the host raises
        const int na = (int)(too_many ? kMaxAnswers : (b1 - b0));
        for (int j = tid; j < na; j += nt) {
            // duplicates of an earlier answer id never count (the reference works on the id set)
            bool dup = false;
            for (int j2 = 0; j2 < j; ++j2) dup |= a.answer_ids[b0 + j2] == a.answer_ids[b0 + j];
            sh.first_rank[j] = dup ? -1 : 0x7FFFFFFF;
        }
        __syncthreads();
        for (int i = tid; i < m; i += nt) {
            const int64_t e = e0 + key_index(sh.sel.keys[i]);
            const int64_t hg = a.node_global_ids[a.edge_index[e]], tg = a.node_global_ids[a.edge_index[a.E + e]];
            for (int j = 0; j < na; ++j) {
                if (sh.first_rank[j] < 0) continue;
                const int64_t aid = a.answer_ids[b0 + j];
                if (hg == aid || tg == aid) atomicMin(&sh.first_rank[j], i + 1);
            }
        }
        __syncthreads();
        if (tid < nk) {
            int uniq = 0, found = 0;
            for (int j = 0; j < na; ++j) {
                if (sh.first_rank[j] < 0) continue;
                ++uniq;
                found += sh.first_rank[j] <= a.kw.k[tid] ? 1 : 0;
            }
            a.answer_hit[(int64_t)g * nk + tid] = found > 0 ? 1 : 0;
            a.answer_recall[(int64_t)g * nk + tid] = uniq > 0 ? (float)((double)found / (double)uniq) : 0.f;
            if (tid == 0) a.answer_valid[g] = too_many ? 2 : (uniq > 0 ? 1 : 0);
        }
    }
}

// Adds one batch's per-graph results (the outputs of k_retriever_metrics) into the epoch's f64 states:
//   acc[0 .. nk)        sum of edge recall@k over graphs with edges       acc[nk]      their count
//   acc[nk+1 .. 2nk+1)  reachability hits@k over graphs with seeds+answers  acc[2nk+1]   their count
//   acc[2nk+2 .. 3nk+2) answer hit@k, acc[3nk+2 .. 4nk+2) answer recall@k   acc[4nk+2]   graphs with answer ids
//   acc[4nk+3]          sum of score margins                              acc[4nk+4]   graphs with both classes
//   acc[4nk+5]          graphs whose answer list exceeded the kernel's capacity (answer_valid == 2)
// One thread per state walks the graphs in order: a fixed summation order, no atomics.
__global__ __launch_bounds__(64) void k_metric_accumulate(
    const float* __restrict__ edge_recall, const uint8_t* __restrict__ recall_valid, const uint8_t* __restrict__ reach,
    const uint8_t* __restrict__ reach_valid, const uint8_t* __restrict__ answer_hit, const float* __restrict__ answer_recall,
    const uint8_t* __restrict__ answer_valid, const float* __restrict__ score_margin, const uint8_t* __restrict__ margin_valid,
    int B, int nk, double* __restrict__ acc) {
    // one WAVE per state: lane l adds the graphs l, l + 64, ... in ascending order, then the 64 partial sums meet in a fixed
    // butterfly — a fixed summation order without float atomics, and all loads of a state in flight at once (one thread walking
    // the graphs paid a dependent load per graph: 42 us for 32 graphs)
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    double sum = 0.0;
    auto term = [&](int g) -> double {
        if (s < nk) return recall_valid[g] ? (double)edge_recall[(int64_t)g * nk + s] : 0.0;
        if (s == nk) return recall_valid[g] ? 1.0 : 0.0;
        if (s < 2 * nk + 1) return reach_valid[g] ? (double)reach[(int64_t)g * nk + (s - nk - 1)] : 0.0;
        if (s == 2 * nk + 1) return reach_valid[g] ? 1.0 : 0.0;
        if (s < 3 * nk + 2) return (answer_valid && answer_valid[g] == 1) ? (double)answer_hit[(int64_t)g * nk + (s - 2 * nk - 2)] : 0.0;
        if (s < 4 * nk + 2) return (answer_valid && answer_valid[g] == 1) ? (double)answer_recall[(int64_t)g * nk + (s - 3 * nk - 2)] : 0.0;
        if (s == 4 * nk + 2) return (answer_valid && answer_valid[g] == 1) ? 1.0 : 0.0;
        if (s == 4 * nk + 3) return margin_valid[g] ? (double)score_margin[g] : 0.0;
        if (s == 4 * nk + 4) return margin_valid[g] ? 1.0 : 0.0;
        return (answer_valid && answer_valid[g] == 2) ? 1.0 : 0.0;
    };
    for (int g = lane; g < B; g += 64) sum += term(g);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) acc[s] += sum;
}

}  // namespace evi

using namespace evi;

extern "C" int evi_retriever_metrics(
    const float* scores, const uint8_t* target, const int64_t* edge_index, int64_t E, const int64_t* edge_ptr,
    const int64_t* node_ptr, int B, const int64_t* q_idx, const int64_t* q_ptr, const int64_t* a_idx,
    const int64_t* a_ptr, const int64_t* node_global_ids, const int64_t* answer_ids, const int64_t* answer_ptr,
    const int32_t* k_values_host, int num_k, float* edge_recall, uint8_t* recall_valid, uint8_t* reach,
    uint8_t* reach_valid, uint8_t* answer_hit, float* answer_recall, uint8_t* answer_valid, float* score_margin,
    uint8_t* margin_valid, int32_t* topk_index, float* topk_score, int32_t* topk_count, int32_t* uf_workspace,
    void* stream) {
    EVI_REQUIRE(B >= 0 && E >= 0, "evi_retriever_metrics: bad sizes B=%d E=%lld", B, (long long)E);
    EVI_REQUIRE(num_k >= 1 && num_k <= kMaxKValues, "evi_retriever_metrics: need 1..%d k values, got %d", kMaxKValues,
                num_k);
    EVI_REQUIRE(k_values_host, "evi_retriever_metrics: null k_values");
    if (B == 0) return EVI_OK;
    MetricsArgs a;
    a.kw.nk = num_k;
    for (int i = 0; i < num_k; ++i) {
        a.kw.k[i] = k_values_host[i];
        EVI_REQUIRE(a.kw.k[i] >= 1 && (i == 0 || a.kw.k[i] > a.kw.k[i - 1]),
                    "evi_retriever_metrics: k values must be positive and strictly ascending");
    }
    a.k_max = a.kw.k[num_k - 1];
    EVI_REQUIRE(a.k_max <= EVI_TOPK_MAX_K, "evi_retriever_metrics: max k %d exceeds %d", a.k_max, EVI_TOPK_MAX_K);
    EVI_REQUIRE(edge_ptr && node_ptr && q_ptr && a_ptr && edge_recall && recall_valid && reach && reach_valid &&
                    score_margin && margin_valid && uf_workspace,
                "evi_retriever_metrics: null pointer");
    EVI_REQUIRE(E == 0 || (scores && edge_index), "evi_retriever_metrics: null scores/edge_index");
    EVI_REQUIRE(!(node_global_ids && answer_ids) || (answer_ptr && answer_hit && answer_recall && answer_valid),
                "evi_retriever_metrics: answer-hit outputs missing");
    a.scores = scores; a.target = target; a.edge_index = edge_index; a.E = E; a.edge_ptr = edge_ptr;
    a.node_ptr = node_ptr; a.q_idx = q_idx; a.q_ptr = q_ptr; a.a_idx = a_idx; a.a_ptr = a_ptr;
    a.node_global_ids = node_global_ids; a.answer_ids = answer_ids; a.answer_ptr = answer_ptr;
    a.edge_recall = edge_recall; a.recall_valid = recall_valid; a.reach = reach; a.reach_valid = reach_valid;
    a.answer_hit = answer_hit; a.answer_recall = answer_recall; a.answer_valid = answer_valid;
    a.score_margin = score_margin; a.margin_valid = margin_valid; a.topk_index = topk_index;
    a.topk_score = topk_score; a.topk_count = topk_count; a.uf_ws = uf_workspace;
    static thread_local bool attr = false;
    if (!attr) {
        EVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_retriever_metrics),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MetricsShared)));
        attr = true;
    }
    hipLaunchKernelGGL(k_retriever_metrics, dim3(B), dim3(kSelectThreads), sizeof(MetricsShared),
                       reinterpret_cast<hipStream_t>(stream), a);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_metric_accumulate(const float* edge_recall, const uint8_t* recall_valid, const uint8_t* reach,
                                     const uint8_t* reach_valid, const uint8_t* answer_hit, const float* answer_recall,
                                     const uint8_t* answer_valid, const float* score_margin, const uint8_t* margin_valid,
                                     int B, int num_k, double* acc, void* stream) {
    EVI_REQUIRE(B >= 0 && num_k >= 1 && num_k <= kMaxKValues, "evi_metric_accumulate: bad sizes B=%d num_k=%d", B, num_k);
    if (B == 0) return EVI_OK;
    EVI_REQUIRE(edge_recall && recall_valid && reach && reach_valid && score_margin && margin_valid && acc,
                "evi_metric_accumulate: null pointer");
    EVI_REQUIRE(!answer_valid || (answer_hit && answer_recall), "evi_metric_accumulate: answer arrays must come together");
    const int total = 4 * num_k + 6;
    hipLaunchKernelGGL(k_metric_accumulate, dim3((unsigned)total), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       edge_recall, recall_valid, reach, reach_valid, answer_hit, answer_recall, answer_valid, score_margin,
                       margin_valid, B, num_k, acc);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
