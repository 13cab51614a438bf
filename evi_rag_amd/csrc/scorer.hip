// Retriever edge scorer (S1-S6): the forward of
//   src/models/components/retriever.py:195-289 (Retriever._forward_impl) on a flat PyG-style batch — evaluation, and training
//   (dropout inside state_net, the hide-and-seek bias, per-edge intermediates kept for the backward of scorer_bwd.hpp).
//
// Pipeline (all on the caller's stream, intermediates in the caller's workspace):
//   1. projections  tanh(x W^T + b): nodes [N,D], questions [B,D] (then q_gate / q_bias on the B
//      projected questions instead of per edge — :464 gathers first, which is the same arithmetic
//      per row), relations either per unique relation id (when the caller states num_relations <= E;
//      a row's projection depends on nothing but the row) or per edge.          -> gemm.hip (MFMA f32)
//   2. structure    CSR + DDE mean propagation -> node_struct [N, 2*(1+R+RR)]    -> graph.hip
//   3. edge features, one wave per edge: struct MLP (K = 20, LN, exact GELU) and nav gate per
//      direction, the DistMult product p = h*r_ctx*t, r_ctx, -||h + r_ctx - t||      -> k_edge_features
//   4. state_net.0, FACTORED.  Its input is [p*nav | s | h + r_ctx - t | -dist] per direction
//      (:471-481), so with W1 = [Wa | Wb | Wc | wd] by column block
//          W1 x_dir = nav_dir (Wa p) + Wb s_dir + (+-)(Wc h - Wc t) + Wc r_ctx + wd (-dist_dir) + b1:
//      p, r_ctx and the node terms do not depend on the direction and the node term only on the
//      node, so the work is  Wc node_repr [N rows, once per batch],  Wa p and Wc r_ctx [E rows],
//      Wb s [2E rows]  —  8 D H flops per edge instead of 12 D H for the concatenated form, and the
//      [2E, 3D+1] input is never materialised                                      -> gemm*.hip
//      Forward-only calls go one step further: r_ctx depends on the edge only through its (relation, graph) pair, so
//      Wc r_ctx is multiplied once per distinct PAIR (k_pair_* below; count kept on the device) instead of once per edge.
//      then one wave per edge sums the terms, LayerNorm + GELU                    -> k_state_combine
//   5. score_head after state_net.4 is one linear map of the normalised row, so the head is FOLDED:
//      logit_dir = (W2^T w) . y_dir + (w . b2 + b), and the 2-way softmax combine happens in k_state_combine.
//      Edge features (EviRetrieverOutput.edge_features), when asked for, are w_f f_f + w_b f_b = W2 (w_f y_f + w_b y_b) + b2:
//      ONE state_net.4 GEMM over E combined rows, written straight into the output (not 2E rows + a combine pass).
// Bound: MFMA (steps 1, 4, 5: 2*(4DH + 2H^2) flops per edge); steps 2-3 are gathers.
#include "common.hpp"

#include <stdlib.h>

namespace evi {

// edges per chunk of the per-edge pipeline (EVI_EDGE_CHUNK overrides: the tests that need several chunks on small batches).
// 262 144: a WebQSP / CWQ batch of 32-64 graphs is ONE chunk — 27 KB of forward and 36 KB of backward scratch per edge, 16 GB at
// the limit, on a 288 GB part; against 65 536 the bench batch (131 k edges) runs half the per-chunk launches (weight-gradient
// products and their reductions above all): forward 3.24 -> 3.21 ms, optimiser step 12.89 -> 12.63 ms, same results.
constexpr int kEdgeChunkDefault = 262144;
static int edge_chunk() {  // read per call (the multi-chunk tests switch it inside one process)
    const char* e = getenv("EVI_EDGE_CHUNK");
    const int n = e ? atoi(e) : 0;
    return n >= 64 ? n : kEdgeChunkDefault;
}
constexpr float kLnEps = 1e-5f;  // torch.nn.LayerNorm default

// Sum over the 64 lanes of a wave, returned wave-uniform.  DPP adds inside the 16-lane rows (quad swaps, half-row and row
// mirrors), row broadcasts across rows, one readlane of lane 63: seven VALU instructions and a scalar result — the
// __shfl_xor butterfly compiles to six DEPENDENT ds_bpermute round trips through the LDS pipe (~100 cycles each), and the
// per-edge kernels reduce 8-16 times per edge.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ inline float wsum(float v) {
    v += dpp_mov<0xB1, 0xF>(v);   // quad_perm [1, 0, 3, 2]
    v += dpp_mov<0x4E, 0xF>(v);   // quad_perm [2, 3, 0, 1]
    v += dpp_mov<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_mov<0x140, 0xF>(v);  // row_mirror: every lane holds its row's sum
    v += dpp_mov<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_mov<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the wave's sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// GELU with erf (torch.nn.GELU default).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, about two
// f32 ulps of the result): 13 instructions instead of libm erff's ~40 — k_edge_features and
// k_state_combine evaluate 48 GELUs per lane per edge and were VALU-bound on them.
__device__ inline float erf_as(float x) {
    const float ax = fabsf(x);
    // v_rcp_f32 (1 ulp) on a denominator in [1, inf): the correctly rounded __frcp_rn is a ten-instruction division sequence,
    // a fifth of these kernels' vector instructions; the approximation's own error is twice the reciprocal's
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// node_repr[v,:] = non_text[:] where node_embedding_ids[v] == 0   (retriever.py:497-507)
__global__ void k_overwrite_non_text(float* __restrict__ node_repr, const int64_t* __restrict__ emb_ids,
                                     const float* __restrict__ non_text, int64_t N, int D) {
    const int64_t v = blockIdx.x;
    if (v >= N || emb_ids[v] != 0) return;
    for (int d = threadIdx.x; d < D; d += blockDim.x) node_repr[v * D + d] = non_text[d];
}

// ---- relation de-duplication ---------------------------------------------------------------------
__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// status |= 1 when some edge_attr lies outside [0, R): what k_first_edge_of_relation reports on the gather path, for callers that hand
// the relation table over (EviRetrieverBatch.relation_rows)
__global__ void k_check_relation_range(const int64_t* __restrict__ edge_attr, int64_t E, int64_t R, int32_t* __restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t r = edge_attr[e];
    if (r < 0 || r >= R) atomicOr(status, 1);
}
__global__ void k_first_edge_of_relation(const int64_t* __restrict__ edge_attr, int64_t E, int64_t R,
                                         int32_t* __restrict__ first, int32_t* __restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t r = edge_attr[e];
    if (r < 0 || r >= R) {
        atomicOr(status, 1);
        return;
    }
    atomicMin(&first[r], (int32_t)e);
}
// rows[r,:] = edge_embeddings[first[r],:] (zeros for relation ids no edge of the batch uses)
__global__ void k_gather_relation_rows(const float* __restrict__ edge_emb, const int32_t* __restrict__ first,
                                       int D, float* __restrict__ rows) {
    const int64_t r = blockIdx.x;
    const int32_t e = first[r];
    for (int d = threadIdx.x; d < D; d += blockDim.x)
        rows[r * D + d] = e == 0x7FFFFFFF ? 0.f : edge_emb[(int64_t)e * D + d];
}

// struct_proj.0.weight [D, F] -> transposed [F, D] so lanes read consecutive d
__global__ void k_transpose(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * cols) dst[(i % cols) * rows + (i / cols)] = src[i];
}

// ---- (relation, graph) pairs --------------------------------------------------------------------------
// r_ctx = rel_repr[relation] * gate_q[graph] + bias_q[graph] (retriever.py:453-462) depends on the edge only through its
// (relation, graph) pair, and state_net.0's Wc block is applied to it as it stands (RC = r_ctx Wc^T): a graph of 4 096 edges has
// a few hundred to ~2 600 distinct pairs, so the forward multiplies one row per PAIR instead of one per edge and the combine
// kernel looks its edge's row up.  The table [B, R] maps a pair to its row (slot), the slots are dealt graph by graph in
// relation order (deterministic); their number stays on the device — the GEMM gets it as GemmBatch::m_dev.
__global__ void k_pair_mark(const int64_t* __restrict__ edge_attr, const int64_t* __restrict__ edge_batch, int64_t E, int64_t R,
                            int32_t* __restrict__ table) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t r = edge_attr[e];
    r = r < 0 ? 0 : (r >= R ? R - 1 : r);  // out-of-range ids are flagged by k_first_edge_of_relation and scored clamped
    table[edge_batch[e] * R + r] = 1;
}
// one workgroup per graph: the set entries of its table row -> 1 + their rank in the row; count[g] = how many
__global__ __launch_bounds__(1024) void k_pair_rank(int32_t* __restrict__ table, int64_t R, int32_t* __restrict__ count) {
    __shared__ int s_sum[1024];
    const int t = threadIdx.x;
    int32_t* row = table + (int64_t)blockIdx.x * R;
    const int64_t per = (R + 1023) / 1024;
    const int64_t lo = t * per, hi = lo + per < R ? lo + per : R;
    int mine = 0;
    for (int64_t i = lo; i < hi; ++i) mine += row[i];
    s_sum[t] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? s_sum[t - off] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    int next = s_sum[t] - mine;
    for (int64_t i = lo; i < hi; ++i)
        if (row[i]) row[i] = ++next;
    if (t == 1023) count[blockIdx.x] = s_sum[1023];
}
// one workgroup per graph: rank in the row -> slot in the batch (graphs in order); pair_key[slot] = g * R + r; count[B] = total
__global__ __launch_bounds__(256) void k_pair_slots(int32_t* __restrict__ table, int64_t R, int B, int32_t* __restrict__ count,
                                                    int32_t* __restrict__ pair_key) {
    __shared__ int s_part[256];
    const int g = blockIdx.x, t = threadIdx.x;
    int before = 0;
    for (int i = t; i < g; i += 256) before += count[i];
    s_part[t] = before;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) s_part[t] += s_part[t + off];
        __syncthreads();
    }
    const int base = s_part[0];
    int32_t* row = table + (int64_t)g * R;
    for (int64_t r = t; r < R; r += 256) {
        const int v = row[r];
        if (v > 0) {
            const int slot = base + v - 1;
            row[r] = slot;
            pair_key[slot] = (int32_t)((int64_t)g * R + r);
        }
    }
    if (g == B - 1 && t == 0) count[B] = base + count[g];
}

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ inline f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ inline void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
// the relation context of one (relation row, graph) at features d .. d + 3 — ONE definition for the per-edge kernel and the per-pair one
__device__ inline f4 rel_ctx4(const float* rp, const float* gp, const float* bp, int d) { return ld4(rp + d) * ld4(gp + d) + ld4(bp + d); }

// one wave per pair: RCX[p, :] = rel_repr[r] * gate_q[g] + bias_q[g]
__global__ __launch_bounds__(256) void k_pair_rows(const int32_t* __restrict__ pair_key, const int32_t* __restrict__ total, int64_t R,
                                                   const float* __restrict__ rel_repr, const float* __restrict__ gate_q,
                                                   const float* __restrict__ bias_q, int D, float* __restrict__ RCX) {
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= *total) return;
    const int lane = threadIdx.x & 63;
    const int64_t key = pair_key[p];
    const int64_t g = key / R, r = key % R;
    const float *rp = rel_repr + r * D, *gp = gate_q + g * D, *bp = bias_q + g * D;
    for (int d = 4 * lane; d < D; d += 256) st4(RCX + p * D + d, rel_ctx4(rp, gp, bp, d));
}

// ---- edge features ----------------------------------------------------------------------------------
// One wave per edge; lane owns features d = lane + 64 i (i < DPL).  Writes two rows of X.
struct EdgeFeatArgs {
    const int64_t* edge_index;  // [2, E]
    int64_t E;
    const int64_t* edge_batch;  // [E]
    const int64_t* edge_attr;   // [E]
    int rel_by_edge;            // 1: rel_repr row = edge; 0: rel_repr row = edge_attr
    int64_t R;                  // rows of rel_repr when rel_by_edge == 0
    const float* node_repr;     // [N, D]
    const float* rel_repr;
    const float* gate_q;        // [B, D]
    const float* bias_q;        // [B, D]
    const float* node_struct;   // [N, F/2]
    int F;                      // 2 * C * S
    const float* struct_wt;     // [F, D] (transposed struct_proj.0.weight)
    const float* struct_b;
    const float* struct_ln_w;
    const float* struct_ln_b;
    const float* struct_gate_w;  // [D]
    const float* struct_gate_b;  // [1]
    int D;
    int64_t e_begin, e_count;  // chunk
    int dir_fwd, dir_bwd;
    float* P;    // [e_count, D]  h * r_ctx * t
    float* RCX;  // [e_count, D]  r_ctx; null when the caller multiplies r_ctx per (relation, graph) pair instead
    float* XS;   // [(dir_fwd + dir_bwd) * e_count, D]  struct context per direction
    float* aux;  // [(dir_fwd + dir_bwd) * e_count, 2]  (nav gate, -||translation error||)
};

// Lane -> feature map of the per-edge kernels: lane owns the float4 chunks d = 4 * lane + 256 * i, i < C4 = ceil(D / 256)
// (D % 4 == 0), so every row access is a 16-byte load / store per lane — 1 KiB per wave instruction, four times fewer
// memory and LDS instructions than one float per lane (cdna_hip_programming.md guideline 13).
__device__ inline float hsum4(f4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
template <int C4>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(C4 <= 3 ? 8 : 4, 8))) void k_edge_features(EdgeFeatArgs a) {
    extern __shared__ float lds_wt[];  // [F][D] + b, ln_w, ln_b, gate_w [4][D]
    const int D = a.D, F = a.F;
    for (int i = threadIdx.x; i < F * D; i += blockDim.x) lds_wt[i] = a.struct_wt[i];
    float* l_b = lds_wt + F * D;
    float* l_lw = l_b + D;
    float* l_lb = l_lw + D;
    float* l_gw = l_lb + D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        l_b[i] = a.struct_b[i];
        l_lw[i] = a.struct_ln_w[i];
        l_lb[i] = a.struct_ln_b[i];
        l_gw[i] = a.struct_gate_w[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const float gate_b = a.struct_gate_b[0];
    const int half = F >> 1;
    const float inv_d = 1.0f / (float)D;
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};

    for (int64_t le = (int64_t)blockIdx.x * waves + wave; le < a.e_count; le += (int64_t)gridDim.x * waves) {
        const int64_t e = a.e_begin + le;
        // one edge per wave: its ids are wave-uniform — say so (readfirstlane), and the row bases and the raw
        // structure features below become scalar loads instead of per-lane address arithmetic and shuffles
        const int64_t hv = __builtin_amdgcn_readfirstlane((int)a.edge_index[e]);
        const int64_t tv = __builtin_amdgcn_readfirstlane((int)a.edge_index[a.E + e]);
        const int64_t g = __builtin_amdgcn_readfirstlane((int)a.edge_batch[e]);
        // a relation id outside [0, R) was flagged by k_first_edge_of_relation (status word -> IndexError on the host, like
        // the reference's index_select); here it is clamped so that the read below stays inside rel_repr
        int64_t rrow = a.rel_by_edge ? e : (int64_t)__builtin_amdgcn_readfirstlane((int)a.edge_attr[e]);
        if (!a.rel_by_edge) rrow = rrow < 0 ? 0 : (rrow >= a.R ? a.R - 1 : rrow);
        const float* hp = a.node_repr + hv * D;
        const float* tp = a.node_repr + tv * D;
        const float* rp = a.rel_repr + rrow * D;
        const float* gp = a.gate_q + g * D;
        const float* bp = a.bias_q + g * D;
        f4 h[C4], t[C4], rc[C4];
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            if (d < D) {
                h[i] = ld4(hp + d);
                t[i] = ld4(tp + d);
                rc[i] = rel_ctx4(rp, gp, bp, d);
            } else {
                h[i] = t[i] = rc[i] = z4;
            }
        }
        // the DistMult product, r_ctx and the translation errors first: h, t and r_ctx (36 registers at C4 = 3) are dead before the
        // struct MLP starts, which is what lets six waves per SIMD fit
        float negdist[2];
        {
            float dsq_f = 0.f, dsq_b = 0.f;
            float* pp = a.P + le * D;
            float* rx = a.RCX + le * D;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                if (d < D) {
                    const f4 ef = h[i] + rc[i] - t[i], eb = t[i] + rc[i] - h[i];
                    dsq_f += hsum4(ef * ef);
                    dsq_b += hsum4(eb * eb);
                    st4(pp + d, h[i] * rc[i] * t[i]);
                    if (a.RCX) st4(rx + d, rc[i]);
                }
            }
            negdist[0] = -sqrtf(wsum(dsq_f));
            negdist[1] = -sqrtf(wsum(dsq_b));
        }
        // struct_proj.0 for BOTH directions in one pass over the weights: struct_raw = cat(ns[a], ns[b]) with
        // (a, b) = (head, tail) forward and (tail, head) backward, so every weight row is read from LDS once and
        // feeds two accumulators; the raw features are wave-uniform scalars.
        const float* nsh = a.node_struct + hv * half;
        const float* nst = a.node_struct + tv * half;
        f4 s2[2][C4];
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            s2[0][i] = s2[1][i] = d < D ? ld4(l_b + d) : z4;
        }
        for (int j = 0; j < half; ++j) {
            const float xh = nsh[j], xt = nst[j];
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                if (d < D) {
                    const f4 w1 = ld4(lds_wt + j * D + d), w2 = ld4(lds_wt + (half + j) * D + d);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        s2[0][i][c] = fmaf(w2[c], xt, fmaf(w1[c], xh, s2[0][i][c]));
                        s2[1][i][c] = fmaf(w2[c], xh, fmaf(w1[c], xt, s2[1][i][c]));
                    }
                }
            }
        }

        int out_row = 0;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            if ((dir == 0 && !a.dir_fwd) || (dir == 1 && !a.dir_bwd)) continue;
            f4 sv[C4];
#pragma unroll
            for (int i = 0; i < C4; ++i) sv[i] = s2[dir][i];
            // LayerNorm over D, exact GELU
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) sum += (4 * lane + 256 * i < D) ? hsum4(sv[i]) : 0.f;
            const float mean = wsum(sum) * inv_d;
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i)
                if (4 * lane + 256 * i < D) {
                    const f4 c = sv[i] - mean;
                    var += hsum4(c * c);
                }
            const float rstd = 1.0f / sqrtf(wsum(var) * inv_d + kLnEps);
            float gacc = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                if (d < D) {
                    const f4 lw = ld4(l_lw + d), lb = ld4(l_lb + d), gw = ld4(l_gw + d);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        sv[i][c] = gelu_erf((sv[i][c] - mean) * rstd * lw[c] + lb[c]);
                        gacc = fmaf(gw[c], sv[i][c], gacc);
                    }
                }
            }
            const float nav = sigmoidf_(wsum(gacc) + gate_b);
            const int64_t row = (int64_t)out_row * a.e_count + le;
            float* xs = a.XS + row * D;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                if (d < D) st4(xs + d, sv[i]);
            }
            if (lane == 0) {
                a.aux[row * 2] = nav;
                a.aux[row * 2 + 1] = negdist[dir];
            }
            ++out_row;
        }
    }
}

// state_net.0.weight [H, 3D+1] -> contiguous column blocks Wa, Wb, Wc [H, D] and the last column wd [H]
__global__ void k_slice_state0(const float* __restrict__ w1, int H, int D, float* __restrict__ wa, float* __restrict__ wb,
                               float* __restrict__ wc, float* __restrict__ wd) {
    const int r = blockIdx.x;
    const float* src = w1 + (int64_t)r * (3 * D + 1);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        wa[(int64_t)r * D + d] = src[d];
        wb[(int64_t)r * D + d] = src[D + d];
        wc[(int64_t)r * D + d] = src[2 * D + d];
    }
    if (threadIdx.x == 0) wd[r] = src[3 * D];
}

// Folded head for the logits-only path: v[j] = sum_i score_w[i] * W2[i, j], v[H] = score_w . b2 + score_b.
// Two steps so that the H x H read is spread over the chip and still summed in a fixed order:
// partial[s][j] over row slice s (blockIdx.y), then the slices are added in slice order.
constexpr int kFoldSlices = 32;
__global__ void k_fold_head_partial(const float* __restrict__ w2, const float* __restrict__ score_w, int H,
                                    float* __restrict__ partial) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    const int per = (H + kFoldSlices - 1) / kFoldSlices;
    const int i0 = blockIdx.y * per, i1 = (i0 + per < H) ? i0 + per : H;
    float acc = 0.f;
    for (int i = i0; i < i1; ++i) acc = fmaf(score_w[i], w2[(int64_t)i * H + j], acc);
    partial[(int64_t)blockIdx.y * H + j] = acc;
}
__global__ void k_fold_head_final(const float* __restrict__ partial, const float* __restrict__ b2,
                                  const float* __restrict__ score_w, const float* __restrict__ score_b, int H,
                                  float* __restrict__ v) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < H) {
        float acc = 0.f;
        for (int s = 0; s < kFoldSlices; ++s) acc += partial[(int64_t)s * H + j];
        v[j] = acc;
    }
    // v[H] = score_w . b2 + score_b by the first workgroup: per-thread strided partial sums, then a fixed-order tree in LDS (one
    // thread walking all H products took 48 us — twice per training step, where the weights change every step)
    if (blockIdx.x == 0) {
        __shared__ float s_dot[256];
        float acc = 0.f;
        for (int i = threadIdx.x; i < H; i += blockDim.x) acc = fmaf(score_w[i], b2[i], acc);
        s_dot[threadIdx.x] = acc;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) s_dot[threadIdx.x] += s_dot[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) v[H] = s_dot[0] + score_b[0];
    }
}

// One wave per edge: h1_dir = nav_dir * PA[e] + SB[dir, e] + sign_dir * (HcN[head] - HcN[tail]) + RC[e]
// + wd * (-dist_dir)   (SB already holds state_net.0.bias), then LayerNorm + exact GELU = y_dir.
// The head is folded: logit_dir = v . y_dir + v[H] with v = W2^T w (score_head after state_net.4 is one linear map of
// y_dir), and the 2-way softmax combine (retriever.py:369-381) happens right here.  When the caller wants the edge features
// (RetrieverOutput.edge_embeddings = w_f f_f + w_b f_b with f_dir = W2 y_dir + b2 and w_f + w_b = 1), linearity gives
// features = W2 (w_f y_f + w_b y_b) + b2: the combined row goes to h1c [e_count, H] and state_net.4 runs on E rows, not 2E.
struct CombineArgs {
    const int64_t* edge_index;
    int64_t E, e_begin, e_count;
    const float* PA;   // [e_count, H]
    const float* RC;   // [e_count, H]; with pair_table: [pairs, H], row = pair_table[graph * R + relation]
    const int32_t* pair_table;  // or null
    const int64_t* edge_attr;
    const int64_t* edge_batch;
    int64_t R;
    const float* SB;   // [dirs * e_count, H]
    const float* HcN;  // [N, H]
    const float* aux;  // [dirs * e_count, 2]
    const float* wd;   // [H]
    const float* ln_w;
    const float* ln_b;
    int H, dir_fwd, dir_bwd;
    float* h1c;      // [e_count, H] combined normalised rows, or null (logits only)
    const float* v;  // [H + 1] folded head
    const float* edge_bias;  // [E] or null
    float* logits;
    float* logits_fwd;
    float* logits_bwd;
    uint32_t drop_thr;   // training dropout on the GELU output: a column is dropped when its 16-bit hash < drop_thr (0 = off)
    float drop_scale;    // 65536 / (65536 - drop_thr)
    uint64_t drop_seed;
};

// The dropout multipliers of four consecutive columns d .. d + 3 of row `row` (= direction * E + edge): one splitmix64 word
// per (row, d / 4) gives four 16-bit uniforms.  Forward and backward call it with the same arguments.
__device__ inline f4 dropout_mul4(uint64_t seed, int64_t row, int H, int d, uint32_t thr, float scale) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(row * (int64_t)(H >> 2) + (d >> 2) + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    f4 m;
#pragma unroll
    for (int c = 0; c < 4; ++c) m[c] = (uint32_t)((z >> (16 * c)) & 0xFFFFu) >= thr ? scale : 0.f;
    return m;
}

// FEAT: the combined normalised row is wanted (a.h1c); without it the two directions' rows are not kept.  Registers: 4 waves per SIMD at
// six waves per SIMD at C4 <= 3 (80 VGPRs), four at C4 = 4 (104), two at C4 = 5
template <int C4, bool FEAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C4 <= 3 ? 6 : (C4 == 4 ? 4 : 2), 8))) void k_state_combine(CombineArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t le = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (le >= a.e_count) return;
    const int H = a.H;
    const int64_t e = a.e_begin + le;
    const float* pa = a.PA + le * H;
    const float* rc = a.RC + le * H;
    if (a.pair_table) {
        int64_t r = a.edge_attr[e];
        r = r < 0 ? 0 : (r >= a.R ? a.R - 1 : r);
        rc = a.RC + (int64_t)a.pair_table[a.edge_batch[e] * a.R + r] * H;
    }
    const float* hh = a.HcN + a.edge_index[e] * H;
    const float* ht = a.HcN + a.edge_index[a.E + e] * H;
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    // the per-edge rows are loaded up front; the LayerNorm weights (the same 3 KB for every edge: L1 / L2 hits) where they are used —
    // holding them (and wd) too cost 36 registers per lane and a wave or two per SIMD
    f4 base[C4], diff[C4], pav[C4];
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int d = 4 * lane + 256 * i;
        if (d < H) {
            pav[i] = ld4(pa + d);
            base[i] = ld4(rc + d);
            diff[i] = ld4(hh + d) - ld4(ht + d);
        } else {
            pav[i] = base[i] = diff[i] = z4;
        }
    }
    float lg[2] = {0.f, 0.f};
    f4 y[FEAT ? 2 : 1][C4];  // the normalised rows of both directions (kept for the combined row when it is wanted)
    int out_row = 0;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        constexpr int kY = FEAT ? 1 : 0;  // y[dir * kY]: one shared row when nothing is kept
#pragma unroll
        for (int i = 0; i < C4; ++i) y[dir * kY][i] = z4;
        if ((dir == 0 && !a.dir_fwd) || (dir == 1 && !a.dir_bwd)) continue;
        const int64_t row = (int64_t)out_row * a.e_count + le;
        const float nav = a.aux[row * 2], negdist = a.aux[row * 2 + 1];
        const float* sb = a.SB + row * H;
        f4 v[C4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            if (d < H) {
                f4 x = nav * pav[i] + ld4(sb + d);
                x += dir == 0 ? diff[i] : -diff[i];
                x += base[i];
                x += ld4(a.wd + d) * negdist;
                v[i] = x;
                sum += hsum4(x);
            } else {
                v[i] = z4;
            }
        }
        const float mean = wsum(sum) / (float)H;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < C4; ++i)
            if (4 * lane + 256 * i < H) {
                const f4 c = v[i] - mean;
                var += hsum4(c * c);
            }
        const float rstd = 1.0f / sqrtf(wsum(var) / (float)H + kLnEps);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            if (d < H) {
                const f4 lnw = ld4(a.ln_w + d), lnb = ld4(a.ln_b + d);
#pragma unroll
                for (int c = 0; c < 4; ++c) y[dir * kY][i][c] = gelu_erf((v[i][c] - mean) * rstd * lnw[c] + lnb[c]);
                if (a.drop_thr) y[dir * kY][i] = y[dir * kY][i] * dropout_mul4(a.drop_seed, (int64_t)dir * a.E + e, H, d, a.drop_thr, a.drop_scale);
                dot += hsum4(ld4(a.v + d) * y[dir * kY][i]);
            }
        }
        lg[dir] = wsum(dot) + a.v[H];
        ++out_row;
    }
    const float eb = a.edge_bias ? a.edge_bias[e] : 0.f;
    const float lf = lg[0] + eb, lb = lg[1] + eb;
    float wf = 1.f, wb = 0.f, out = lf;
    if (a.dir_fwd && a.dir_bwd) {
        const float m = fmaxf(lf, lb);
        const float ef = expf(lf - m), ebk = expf(lb - m);
        wf = ef / (ef + ebk);
        wb = ebk / (ef + ebk);
        out = wf * lf + wb * lb;
    } else if (a.dir_bwd) {
        wf = 0.f;
        wb = 1.f;
        out = lb;
    }
    if (lane == 0) {
        a.logits[e] = out;
        if (a.logits_fwd && a.dir_fwd) a.logits_fwd[e] = lf;
        if (a.logits_bwd && a.dir_bwd) a.logits_bwd[e] = lb;
    }
    if constexpr (FEAT) {  // edge features requested: state_net.4 is linear, so it runs ONCE on the combined normalised row
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            if (d < H) st4(a.h1c + le * H + d, wf * y[0][i] + wb * y[1][i]);
        }
    }
}

}  // namespace evi
#include "scorer_bwd.hpp"
namespace evi {

static bool use_f32_gemm() {
    const char* v = getenv("EVI_SCORER_GEMM");
    return v && v[0] == 'f';
}

// EviRetrieverBatch.matmul_precision of the call in flight on this thread: 1 = the large GEMMs multiply ONE bf16 product
// (f32 accumulation and results) instead of three.  Set for the duration of retriever_run by SingleProductScope; read by the
// two places every large product of the scorer goes through (scorer_gemm, tn_gemm).
static thread_local int t_gemm_single = 0;
struct SingleProductScope {
    int before;
    explicit SingleProductScope(int v) : before(t_gemm_single) { t_gemm_single = v; }
    ~SingleProductScope() { t_gemm_single = before; }
};

// act(A W^T + b) on the split-bf16 GEMM (default) or the exact f32 GEMM (EVI_SCORER_GEMM=f32).  wplanes: the weight's
// bf16 hi / lo planes when the caller keeps them prepared (evi_retriever_prepare), else they are made here in `wsplit`.
static int scorer_gemm(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw,
                       const float* bias, int act, float* C, int64_t ldc, void* wsplit, hipStream_t st,
                       const void* wplanes = nullptr) {
    // a handful of rows (question-side projections, the non-text embedding): one wave per output column, exact f32 (gemm_skinny.hip)
    if (gemm_skinny_fits(M, K, lda, ldw)) return launch_gemm_skinny(A, M, K, lda, W, N, ldw, bias, act, C, ldc, st);
    if (use_f32_gemm()) return launch_gemm_nt(A, M, K, lda, W, N, ldw, bias, act, C, ldc, st);
    // matmul_precision 2 (f16x2, forward only): the prepared planes are bf16 hi / lo, so the weight is rounded to its f16 plane
    // here, per call, into `wsplit` (a [N, K] pass: 0.6 M elements next to a 50 M-element product)
    if (wplanes && t_gemm_single != 2) return launch_gemm_nt_bf16x3_wplanes(A, M, K, lda, wplanes, N, bias, act, C, ldc, st, t_gemm_single);
    return launch_gemm_nt_bf16x3(A, M, K, lda, W, N, ldw, bias, act, C, ldc, wsplit, st, t_gemm_single);
}

// Everything the forward derives from the WEIGHTS alone, kept across calls by callers whose weights do not change between
// forwards (evaluation): the column blocks of state_net.0, the transposed struct_proj weight, the folded head, and the
// bf16 hi / lo planes of the nine GEMM weights.  ~25 small launches per forward otherwise.
struct PrepLayout {
    size_t wa, wb, wc, wd, wt, vhead, fold, p_entity, p_query, p_gate, p_bias, p_rel, p_wa, p_wb, p_wc, p_s4, total;
};
static PrepLayout prep_layout(int D, int H, int F) {
    PrepLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t f = sizeof(float);
    L.wa = take((size_t)H * D * f);
    L.wb = take((size_t)H * D * f);
    L.wc = take((size_t)H * D * f);
    L.wd = take((size_t)H * f);
    L.wt = take((size_t)F * D * f);
    L.vhead = take((size_t)(H + 1) * f);
    L.fold = take((size_t)32 * H * f);
    L.p_entity = take(gemm_bf16x3_workspace_bytes(D, D));
    L.p_query = take(gemm_bf16x3_workspace_bytes(D, D));
    L.p_gate = take(gemm_bf16x3_workspace_bytes(D, D));
    L.p_bias = take(gemm_bf16x3_workspace_bytes(D, D));
    L.p_rel = take(gemm_bf16x3_workspace_bytes(D, D));
    L.p_wa = take(gemm_bf16x3_workspace_bytes(H, D));
    L.p_wb = take(gemm_bf16x3_workspace_bytes(H, D));
    L.p_wc = take(gemm_bf16x3_workspace_bytes(H, D));
    L.p_s4 = take(gemm_bf16x3_workspace_bytes(H, H));
    L.total = off;
    return L;
}

// float4 chunks per lane: ceil(d / 256), d <= 1280
static int dpl_for(int d) {
    const int need = (d + 255) / 256;
    return need < 1 ? 1 : (need > 5 ? 0 : need);
}

#define EVI_DPL_DISPATCH(dpl, CALL)              \
    switch (dpl) {                               \
        case 1: { constexpr int DPL = 1; CALL; } break;   \
        case 2: { constexpr int DPL = 2; CALL; } break;   \
        case 3: { constexpr int DPL = 3; CALL; } break;   \
        case 4: { constexpr int DPL = 4; CALL; } break;   \
        default: { constexpr int DPL = 5; CALL; } break;  \
    }

struct FwdLayout {
    size_t node_repr, non_text, q_proj, gate_q, bias_q, rel_repr, rel_rows, rel_first, status, ns, in_ptr, in_nbr,
        in_eid, out_ptr, out_nbr, out_eid, csr_ws, wa, wb, wc, wd, vhead, fold, wt, wsplit, hcn, P, RCX, XS, aux, PA, RC, SB,
        h1n, pair_table, pair_count, pair_key, pair_rcx, pair_rc, total;
    int64_t ec, pair_max;  // pair_max: bound on the (relation, graph) pairs of a batch, 0 = the per-edge form
    int dedupe;
};

// (relation, graph) pair rows instead of per-edge r_ctx rows: on unless EVI_SCORER_PAIRS=0; needs the relation table form
// (R <= E) and a [B, R] slot table of at most 4 M entries
static bool scorer_pairs() {  // read per call (the parity test switches it inside one process)
    const char* e = getenv("EVI_SCORER_PAIRS");
    return !(e && e[0] == '0');
}
constexpr int64_t kPairTableMax = (int64_t)1 << 22;

static FwdLayout fwd_layout(int64_t N, int64_t E, int B, int D, int H, int F, int64_t R, int dirs) {
    FwdLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    L.dedupe = (R > 0 && R <= E) ? 1 : 0;
    L.ec = E < edge_chunk() ? (E > 0 ? E : 1) : edge_chunk();
    const size_t f = sizeof(float);
    const size_t n1 = (size_t)(N > 0 ? N : 1), e1 = (size_t)(E > 0 ? E : 1);
    L.node_repr = take(n1 * D * f);
    L.non_text = take((size_t)D * f);
    L.q_proj = take((size_t)B * D * f);
    L.gate_q = take((size_t)B * D * f);
    L.bias_q = take((size_t)B * D * f);
    L.rel_repr = take((size_t)(L.dedupe ? R : e1) * D * f);
    L.rel_rows = take(L.dedupe ? (size_t)R * D * f : 256);
    L.rel_first = take(L.dedupe ? (size_t)R * 4 : 256);
    L.status = take(256);
    L.ns = take(n1 * (F / 2) * f);
    L.in_ptr = take((size_t)(N + 1) * 4);
    L.in_nbr = take(e1 * 4);
    L.in_eid = take(e1 * 4);
    L.out_ptr = take((size_t)(N + 1) * 4);
    L.out_nbr = take(e1 * 4);
    L.out_eid = take(e1 * 4);
    L.csr_ws = take(evi_graph_csr_workspace_bytes(N));
    L.wa = take((size_t)H * D * f);
    L.wb = take((size_t)H * D * f);
    L.wc = take((size_t)H * D * f);
    L.wd = take((size_t)H * f);
    L.vhead = take((size_t)(H + 1) * f);
    L.fold = take((size_t)32 * H * f);  // kFoldSlices partial rows of the folded head
    L.wt = take((size_t)F * D * f);
    L.wsplit = take(gemm_bf16x3_workspace_bytes(H > D ? H : D, H > D ? H : D));
    L.hcn = take(n1 * H * f);
    L.P = take((size_t)L.ec * D * f);
    L.RCX = take((size_t)L.ec * D * f);
    L.XS = take((size_t)dirs * L.ec * D * f);
    L.aux = take((size_t)dirs * L.ec * 2 * f);
    L.PA = take((size_t)L.ec * H * f);
    L.RC = take((size_t)L.ec * H * f);
    L.SB = take((size_t)dirs * L.ec * H * f);
    L.h1n = take((size_t)L.ec * H * f);  // combined normalised rows (input of state_net.4 when features are wanted)
    const int64_t table = (int64_t)B * R;
    L.pair_max = (L.dedupe && scorer_pairs() && !use_f32_gemm() && table <= kPairTableMax && E >= 1024) ? (E < table ? E : table) : 0;
    L.pair_table = take(L.pair_max ? (size_t)table * 4 : 256);
    L.pair_count = take(L.pair_max ? (size_t)(B + 1) * 4 : 256);
    L.pair_key = take(L.pair_max ? (size_t)L.pair_max * 4 : 256);
    L.pair_rcx = take(L.pair_max ? (size_t)L.pair_max * D * f : 256);
    L.pair_rc = take(L.pair_max ? (size_t)L.pair_max * H * f : 256);
    L.total = off;
    return L;
}

}  // namespace evi

using namespace evi;


extern "C" size_t evi_retriever_prepare_bytes(int D, int H, int dde_rounds, int dde_reverse_rounds) {
    if (D < 1 || H < 1 || dde_rounds < 0 || dde_reverse_rounds < 0) return 0;
    return prep_layout(D, H, 2 * 2 * (1 + dde_rounds + dde_reverse_rounds)).total;
}

extern "C" int evi_retriever_prepare(const EviRetrieverWeights* w, void* prepared, size_t prepared_bytes, void* stream) {
    EVI_REQUIRE(w && prepared, "evi_retriever_prepare: null pointer");
    const int D = w->emb_dim, H = w->hidden_dim;
    EVI_REQUIRE(D >= 1 && H >= 1 && D % 4 == 0 && H % 4 == 0, "evi_retriever_prepare: D and H must be positive multiples of 4");
    const int F = 2 * 2 * (1 + w->dde_rounds + w->dde_reverse_rounds);
    const PrepLayout PL = prep_layout(D, H, F);
    if (prepared_bytes < PL.total)
        return fail(EVI_ERR_NOMEM, "evi_retriever_prepare: buffer %zu B < %zu B", prepared_bytes, PL.total);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(prepared);
    auto PF = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    hipLaunchKernelGGL(k_slice_state0, dim3(H), dim3(256), 0, st, w->state0_w, H, D, PF(PL.wa), PF(PL.wb), PF(PL.wc), PF(PL.wd));
    hipLaunchKernelGGL(k_transpose, dim3((F * D + 255) / 256), dim3(256), 0, st, w->struct_w, D, F, PF(PL.wt));
    hipLaunchKernelGGL(k_fold_head_partial, dim3((H + 255) / 256, kFoldSlices), dim3(256), 0, st, w->state4_w, w->score_w, H,
                       PF(PL.fold));
    hipLaunchKernelGGL(k_fold_head_final, dim3((H + 255) / 256), dim3(256), 0, st, PF(PL.fold), w->state4_b, w->score_w,
                       w->score_b, H, PF(PL.vhead));
    EVI_LAUNCH_CHECK();
    int rc;
    if ((rc = split_weight_bf16x3(w->entity_w, D, D, D, base + PL.p_entity, st))) return rc;
    if ((rc = split_weight_bf16x3(w->query_w, D, D, D, base + PL.p_query, st))) return rc;
    if ((rc = split_weight_bf16x3(w->q_gate_w, D, D, D, base + PL.p_gate, st))) return rc;
    if ((rc = split_weight_bf16x3(w->q_bias_w, D, D, D, base + PL.p_bias, st))) return rc;
    if ((rc = split_weight_bf16x3(w->relation_w, D, D, D, base + PL.p_rel, st))) return rc;
    if ((rc = split_weight_bf16x3(PF(PL.wa), H, D, D, base + PL.p_wa, st))) return rc;
    if ((rc = split_weight_bf16x3(PF(PL.wb), H, D, D, base + PL.p_wb, st))) return rc;
    if ((rc = split_weight_bf16x3(PF(PL.wc), H, D, D, base + PL.p_wc, st))) return rc;
    if ((rc = split_weight_bf16x3(w->state4_w, H, H, H, base + PL.p_s4, st))) return rc;
    return EVI_OK;
}

extern "C" size_t evi_retriever_forward_workspace_bytes(int64_t N, int64_t E, int B, int D, int H, int dde_rounds,
                                                        int dde_reverse_rounds, int64_t num_relations) {
    if (N < 0 || E < 0 || B < 1 || D < 1 || H < 1) return 0;
    const int F = 2 * 2 * (1 + dde_rounds + dde_reverse_rounds);
    return fwd_layout(N, E, B, D, H, F, num_relations, 2).total;
}

// The node-level results of a training forward, kept behind the per-edge rows of `saved` when the caller's buffer has room
// (evi_retriever_saved_bytes_full): projected nodes / questions / relations, the structure features, both CSR halves and
// node_repr Wc^T.  The backward then starts from them instead of recomputing the node-level forward (three large GEMMs, the CSR
// build, DDE, the relation de-duplication: ~0.6 ms of a 13.6 ms step at the bench shape).
struct NodeKeep {
    size_t node_repr, non_text, q_proj, gate_q, bias_q, rel_repr, rel_rows, ns, in_ptr, in_nbr, in_eid, out_ptr, out_nbr, out_eid, hcn, total;
};
static NodeKeep node_keep_layout(int64_t N, int64_t E, int B, int D, int H, int F, int64_t R, int dedupe) {
    NodeKeep K;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t f = sizeof(float);
    const size_t n1 = (size_t)(N > 0 ? N : 1), e1 = (size_t)(E > 0 ? E : 1);
    K.node_repr = take(n1 * D * f);
    K.non_text = take((size_t)D * f);
    K.q_proj = take((size_t)B * D * f);
    K.gate_q = take((size_t)B * D * f);
    K.bias_q = take((size_t)B * D * f);
    K.rel_repr = take((size_t)(dedupe ? R : e1) * D * f);
    K.rel_rows = take(dedupe ? (size_t)R * D * f : 256);
    K.ns = take(n1 * (F / 2) * f);
    K.in_ptr = take((size_t)(N + 1) * 4);
    K.in_nbr = take(e1 * 4);
    K.in_eid = take(e1 * 4);
    K.out_ptr = take((size_t)(N + 1) * 4);
    K.out_nbr = take(e1 * 4);
    K.out_eid = take(e1 * 4);
    K.hcn = take(n1 * H * f);
    K.total = off;
    return K;
}

// Backward context of retriever_run (null: forward only).  Gradients use the weights struct's layout (float* written).
struct BwdCtx {
    const float* dlogits;          // [E]
    const EviRetrieverWeights* g;  // gradient buffers, same shapes as the weights
    const int64_t* rel_perm;       // [E] edge ids grouped by relation id (stable), or null when relations are not de-duplicated
    const int64_t* rel_ptr;        // [R + 1] segment bounds into rel_perm
    char* ws;                      // backward workspace (evi_retriever_backward_workspace_bytes)
    const char* saved;             // the forward's per-edge intermediates (EviRetrieverOutput.saved), or null: recompute them
    size_t saved_bytes;
};

struct BwdLayout {
    size_t DZ, DPA, DRC, daux, dP, dRCX, dXS, DU, SX, partC, partE, DDF, DH, DT, DRR, DGQ, DBQ, dNR, dHcN, tmpN, dRRu, dGQ, dBQ, dQP,
        dnt, WaT, WbT, WcT, WgT, WbqT, WeT, gWa, gWb, gWc, gwd, ysum, ssum, At, Bt, tnpart, wsplit2, colpart, total;
    int64_t kmax;
    int gridC, gridE, wavesE;
};
constexpr int kTnSlice = 8192;
constexpr int kTnMaxSlices = 64;

static BwdLayout bwd_layout(int64_t N, int64_t E, int B, int D, int H, int F, int64_t R, int64_t ec) {
    BwdLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t f = sizeof(float);
    const int mx = D > H ? D : H;
    const size_t e1 = (size_t)(E > 0 ? E : 1), n1 = (size_t)(N > 0 ? N : 1);
    L.gridC = (int)((ec + 3) / 4 < 1024 ? (ec + 3) / 4 : 1024);
    if (L.gridC < 1) L.gridC = 1;
    L.wavesE = 6;  // 384-thread workgroups: ~150 registers per lane and (F + 4) D floats of LDS -> two workgroups, 12 waves per CU
    L.gridE = (int)((ec + L.wavesE - 1) / L.wavesE < 1024 ? (ec + L.wavesE - 1) / L.wavesE : 1024);
    if (L.gridE < 1) L.gridE = 1;
    L.DZ = take((size_t)2 * ec * H * f);
    L.DPA = take((size_t)ec * H * f);
    L.DRC = take((size_t)ec * H * f);
    L.daux = take((size_t)2 * ec * 2 * f);
    L.dP = take((size_t)ec * D * f);
    L.dRCX = take((size_t)ec * D * f);
    L.dXS = take((size_t)2 * ec * D * f);
    L.DU = take((size_t)2 * ec * D * f);
    L.SX = take((size_t)2 * ec * F * f);
    L.partC = take(((size_t)L.gridC * 5 * H + L.gridC) * f);
    L.partE = take(((size_t)L.gridE * L.wavesE * 4 * D + (size_t)L.gridE * L.wavesE) * f);
    L.DDF = take(e1 * H * f);
    L.DH = take(e1 * D * f);
    L.DT = take(e1 * D * f);
    L.DRR = take(e1 * D * f);
    L.DGQ = take(e1 * D * f);
    L.DBQ = take(e1 * D * f);
    L.dNR = take(n1 * D * f);
    L.dHcN = take(n1 * H * f);
    L.tmpN = take(n1 * D * f);
    L.dRRu = take((size_t)(R > 0 ? R : 1) * D * f);
    L.dGQ = take((size_t)B * D * f);
    L.dBQ = take((size_t)B * D * f);
    L.dQP = take((size_t)B * D * f * 2);
    L.dnt = take((size_t)D * f * 2);
    L.WaT = take((size_t)D * H * f);
    L.WbT = take((size_t)D * H * f);
    L.WcT = take((size_t)D * H * f);
    L.WgT = take((size_t)D * D * f);
    L.WbqT = take((size_t)D * D * f);
    L.WeT = take((size_t)D * D * f);
    L.gWa = take((size_t)H * D * f);
    L.gWb = take((size_t)H * D * f);
    L.gWc = take((size_t)H * D * f);
    L.gwd = take((size_t)H * f);
    L.ysum = take((size_t)H * f);
    L.ssum = take(256);
    int64_t kmax = 2 * ec;
    if ((int64_t)N > kmax) kmax = N;
    if (E > kmax && R <= 0) kmax = E;
    if (R > kmax) kmax = R;
    if (B > kmax) kmax = B;
    kmax = (kmax + 31) / 32 * 32;
    L.kmax = kmax;
    L.At = take((size_t)mx * kmax * f);
    L.Bt = take((size_t)mx * kmax * f);
    L.tnpart = take((size_t)kTnMaxSlices * mx * (mx > F ? mx : F) * f);  // slice partials [S][M][N]: M, N <= max(D, H, F)
    L.wsplit2 = take(gemm_bf16x3_workspace_bytes(mx, (int)kmax));  // planes of a whole transposed operand (split-K batch)
    L.colpart = take(((size_t)(kmax + kColsumRows - 1) / kColsumRows + 16) * 5 * mx * f);  // also the scratch row of the five / three column vectors
    L.total = off;
    return L;
}

// C [M, N] (+)= A^T B with A [K, M], B [K, N] row-major and K large: split-K launches of the TN kernel (gemm_tn.hip) and an
// ordered reduction of the slice results (deterministic).  The exact-f32 mode of the tests transposes both operands and goes
// through the f32 NT GEMM slice by slice.
static int tn_gemm(const float* A, int M, const float* Bm, int N, int64_t K, float* C, int accumulate, const BwdLayout& L,
                   char* ws, hipStream_t st) {
    if (K <= 0) {
        if (!accumulate) hipLaunchKernelGGL(k_zero_f32, dim3(64), dim3(256), 0, st, C, (int64_t)M * N);
        return EVI_OK;
    }
    if (K <= 64) {  // a handful of rows: exact f32, one launch (k_gemm_tn_small)
        hipLaunchKernelGGL(k_gemm_tn_small, dim3((unsigned)((N + 255) / 256), (unsigned)M), dim3(256), 0, st, A, M, Bm, N, (int)K, C, accumulate ? 1 : 0);
        EVI_LAUNCH_CHECK();
        return EVI_OK;
    }
    const int64_t Kp = (K + 31) / 32 * 32;
    float* part = reinterpret_cast<float*>(ws + L.tnpart);
    if (!use_f32_gemm()) {
        // one launch of the TN kernel (gemm_tn.hip: the operands are read as they lie, k-major); the K-slices ride in
        // gridDim.y — the output is only (M / 256) x (N / 256) tiles, slices fill the chip
        int64_t Ks;
        int S;
        gemm_tn_plan(M, N, K, kTnMaxSlices, &Ks, &S);
        int rc = launch_gemm_tn_bf16x3(A, M, M, Bm, N, N, K, Ks, (int)S, part, st, t_gemm_single == 1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, st, part, (int)S,
                           (int64_t)M * N, C, accumulate ? 1 : 0);
        EVI_LAUNCH_CHECK();
        return EVI_OK;
    }
    float* At = reinterpret_cast<float*>(ws + L.At);
    float* Bt = reinterpret_cast<float*>(ws + L.Bt);
    const dim3 ga((unsigned)((Kp + 31) / 32), (unsigned)((M + 31) / 32)), gb((unsigned)((Kp + 31) / 32), (unsigned)((N + 31) / 32));
    hipLaunchKernelGGL(k_transpose_pad, ga, dim3(256), 0, st, A, K, M, (int64_t)M, At, Kp);
    hipLaunchKernelGGL(k_transpose_pad, gb, dim3(256), 0, st, Bm, K, N, (int64_t)N, Bt, Kp);
    EVI_LAUNCH_CHECK();
    // exact-f32 mode (tests): slice by slice through the f32 GEMM
    int64_t S = (Kp + kTnSlice - 1) / kTnSlice;
    if (S > kTnMaxSlices) S = kTnMaxSlices;
    int64_t Ks = ((Kp + S - 1) / S + 31) / 32 * 32;
    if (Ks > kTnSlice + 32) Ks = kTnSlice;
    int used = 0;
    int first_round = 1;
    for (int64_t k0 = 0; k0 < Kp; k0 += Ks) {
        const int64_t kl = Kp - k0 < Ks ? Kp - k0 : Ks;
        int rc = scorer_gemm(At + k0, M, (int)kl, Kp, Bt + k0, N, Kp, nullptr, 0, part + (int64_t)used * M * N, N, ws + L.wsplit2, st);
        if (rc) return rc;
        if (++used == kTnMaxSlices || k0 + Ks >= Kp) {
            hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, st, part, used,
                               (int64_t)M * N, C, (accumulate || !first_round) ? 1 : 0);
            EVI_LAUNCH_CHECK();
            used = 0;
            first_round = 0;
        }
    }
    return EVI_OK;
}

// out[cols] (+)= column sums of X [rows, cols]
static int colsum_into(const float* X, int64_t rows, int cols, float* out, int accumulate, const BwdLayout& L, char* ws,
                       hipStream_t st) {
    if (rows <= 0) return EVI_OK;
    float* part = reinterpret_cast<float*>(ws + L.colpart);
    const int nb = (int)((rows + kColsumRows - 1) / kColsumRows);
    if (nb == 1) {  // one row block: summed straight into `out`
        hipLaunchKernelGGL(k_colsum_partial, dim3(1, (unsigned)((cols + 255) / 256)), dim3(256), 0, st, X, rows, cols, part, out, accumulate);
        EVI_LAUNCH_CHECK();
        return EVI_OK;
    }
    hipLaunchKernelGGL(k_colsum_partial, dim3(nb, (unsigned)((cols + 255) / 256)), dim3(256), 0, st, X, rows, cols, part, static_cast<float*>(nullptr), 0);
    hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, part, nb, (int64_t)cols, out, accumulate);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

// dst[q][0..W) += column sums of X [rows, Q * W], columns grouped into Q vectors of W (two ordered stages, two launches)
static int colsum_into_multi(const float* X, int64_t rows, int Q, int W, float* const* dst, const BwdLayout& L, char* ws, hipStream_t st) {
    if (rows <= 0) return EVI_OK;
    float* part = reinterpret_cast<float*>(ws + L.colpart);
    const int cols = Q * W;
    const int nb = (int)((rows + kColsumRows - 1) / kColsumRows);
    ReduceDst d{};
    for (int q = 0; q < Q; ++q) d.p[q] = dst[q];
    hipLaunchKernelGGL(k_colsum_partial, dim3(nb, (unsigned)((cols + 255) / 256)), dim3(256), 0, st, X, rows, cols, part, static_cast<float*>(nullptr), 0);
    hipLaunchKernelGGL(k_reduce_partials_multi, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, part, nb, Q, W, d);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

// out[s][:] = sum of the rows of X in segment s (k_segment_rowsum); long segments are cut into slices summed in order afterwards
static int segment_rowsum_into(const float* X, int D, const int64_t* ptr, const int64_t* perm, int64_t S, int64_t rows_total,
                               float* out, const BwdLayout& L, char* ws, hipStream_t st) {
    if (S <= 0) return EVI_OK;
    int64_t Z = (rows_total / S + 255) / 256;
    if (Z < 1) Z = 1;
    if (Z > 32) Z = 32;
    const size_t cap = (size_t)kTnMaxSlices * (D > 1 ? D : 1) * (D > 1 ? D : 1);  // floats in the TN partial buffer (>= mx * mx per slice)
    while (Z > 1 && (size_t)Z * S * D > cap) --Z;
    float* dst = Z == 1 ? out : reinterpret_cast<float*>(ws + L.tnpart);
    hipLaunchKernelGGL(k_segment_rowsum, dim3((unsigned)S, (unsigned)((D + 255) / 256), (unsigned)Z), dim3(256), 0, st, X, D, ptr, perm, dst, S);
    if (Z > 1)
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((S * D + 255) / 256)), dim3(256), 0, st, dst, (int)Z, S * D, out, 0);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

// out2[s][:] = sum over segment s of X[row] * T[clamp(idx[row])], and (out != null) out[s][:] = sum of X[row], from one pass
static int segment_rowsum_mul_into(const float* X, int D, const int64_t* ptr, const int64_t* perm, int64_t S, int64_t rows_total,
                                   float* out, float* out2, const float* T, const int64_t* idx, int64_t idx_hi, const BwdLayout& L,
                                   char* ws, hipStream_t st) {
    if (S <= 0) return EVI_OK;
    int64_t Z = (rows_total / S + 255) / 256;
    if (Z < 1) Z = 1;
    if (Z > 32) Z = 32;
    const size_t cap = (size_t)kTnMaxSlices * (D > 1 ? D : 1) * (D > 1 ? D : 1) / 2;  // two partial tables share the TN partial buffer
    while (Z > 1 && (size_t)Z * S * D > cap) --Z;
    float* part = reinterpret_cast<float*>(ws + L.tnpart);
    float* dst = Z == 1 ? out : part;
    float* dst2 = Z == 1 ? out2 : part + (size_t)Z * S * D;
    const dim3 grid((unsigned)S, (unsigned)((D + 255) / 256), (unsigned)Z);
    if (out) hipLaunchKernelGGL(k_segment_rowsum_mul<true>, grid, dim3(256), 0, st, X, D, ptr, perm, dst, dst2, S, T, idx, idx_hi);
    else hipLaunchKernelGGL(k_segment_rowsum_mul<false>, grid, dim3(256), 0, st, X, D, ptr, perm, dst, dst2, S, T, idx, idx_hi);
    if (Z > 1) {
        if (out) hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((S * D + 255) / 256)), dim3(256), 0, st, dst, (int)Z, S * D, out, 0);
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((S * D + 255) / 256)), dim3(256), 0, st, dst2, (int)Z, S * D, out2, 0);
    }
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

__global__ void k_add_inplace(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
// state_net.0.weight gradient [H, 3D+1] from its column blocks
__global__ void k_merge_state0(const float* __restrict__ ga, const float* __restrict__ gb, const float* __restrict__ gc,
                               const float* __restrict__ gd, int H, int D, float* __restrict__ out) {
    const int r = blockIdx.x;
    float* dst = out + (int64_t)r * (3 * D + 1);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        dst[d] = ga[(int64_t)r * D + d];
        dst[D + d] = gb[(int64_t)r * D + d];
        dst[2 * D + d] = gc[(int64_t)r * D + d];
    }
    if (threadIdx.x == 0) dst[3 * D] = gd[r];
}

// Per-edge intermediates a training forward keeps for the backward (EviRetrieverOutput.saved): for the chunk starting at edge
// e0, at float offset e0 * saved_floats_per_edge: P [ec, D], RCX [ec, D], XS [dirs ec, D], PA [ec, H], RC [ec, H],
// SB [dirs ec, H], aux [dirs ec, 2] — the chunk workspace's own layout, kept for all chunks.
static inline size_t saved_floats_per_edge(int D, int H, int dirs) { return (size_t)(2 + dirs) * D + (size_t)(2 + dirs) * H + 2 * (size_t)dirs; }

static int retriever_run(const EviRetrieverWeights* w, const EviRetrieverBatch* b, int direction_mode,
                         const EviRetrieverOutput* out, void* workspace, size_t workspace_bytes, void* stream,
                         const BwdCtx* bw) {
    EVI_REQUIRE(w && b && out, "evi_retriever_forward: null struct pointer");
    const int D = w->emb_dim, H = w->hidden_dim;
    const int64_t N = b->num_nodes, E = b->num_edges;
    const int B = b->num_graphs;
    EVI_REQUIRE(D >= 1 && H >= 1, "evi_retriever_forward: bad dims emb_dim=%d hidden_dim=%d", D, H);
    if (D % 4 != 0 || H % 4 != 0 || D > 1280 || H > 1280)
        return fail(EVI_ERR_UNSUPPORTED, "evi_retriever_forward: emb_dim/hidden_dim must be multiples of 4 and <= 1280, got %d/%d", D, H);
    if (w->num_topics != 2) return fail(EVI_ERR_INVALID, "num_topics must be 2 (seed vs non-seed), got %d", w->num_topics);
    EVI_REQUIRE(w->dde_rounds >= 0 && w->dde_rounds <= 4 && w->dde_reverse_rounds >= 0 && w->dde_reverse_rounds <= 4,
                "DDE supports at most 4 rounds per direction; got num_rounds=%d, num_reverse_rounds=%d.",
                w->dde_rounds, w->dde_reverse_rounds);
    EVI_REQUIRE(direction_mode >= 0 && direction_mode <= 2,
                "direction_mode must be one of {'bidirectional', 'forward', 'backward'}");
    EVI_REQUIRE(B >= 1, "num_graphs must be positive, got %d", B);
    EVI_REQUIRE(N >= 1, "total_nodes must be positive, got %lld", (long long)N);
    EVI_REQUIRE(E >= 0, "evi_retriever_forward: negative edge count");
    if (E == 0) return EVI_OK;  // reference returns empty outputs (:206-207)
    EVI_REQUIRE(b->edge_index, "Batch missing edge_index required for scoring.");
    EVI_REQUIRE(b->question_emb && b->node_embedding_ids && b->edge_attr,
                "Batch must provide question_emb, node_embedding_ids, and edge_attr.");
    EVI_REQUIRE(b->node_embeddings && (b->edge_embeddings || (b->relation_rows && b->num_relations > 0 && b->num_relations <= E)),
                "Batch must provide node_embeddings and edge_embeddings.");
    EVI_REQUIRE(b->topic_one_hot, "topic_one_hot is required for DDE-based structure features.");
    EVI_REQUIRE(b->topic_stride >= 2, "topic_one_hot feature dim %d < num_topics=2", b->topic_stride);
    EVI_REQUIRE(b->node_ptr && b->edge_ptr && b->edge_batch, "evi_retriever_forward: node_ptr/edge_ptr/edge_batch required");
    EVI_REQUIRE(out->logits, "evi_retriever_forward: output logits pointer is null");
    const int dir_fwd = direction_mode != 2, dir_bwd = direction_mode != 1;
    const int dirs = dir_fwd + dir_bwd;
    EVI_REQUIRE(b->dropout_p >= 0.f && b->dropout_p < 1.f, "dropout probability has to be between 0 and 1, but got %g", (double)b->dropout_p);
    uint32_t drop_thr = (uint32_t)lrintf(b->dropout_p * 65536.0f);
    if (drop_thr > 65535u) drop_thr = 65535u;
    EVI_REQUIRE(b->matmul_precision >= 0 && b->matmul_precision <= 2,
                "matmul_precision must be 0 (split-bf16, three products), 1 (one bf16 product) or 2 (f16x2: two f16 products, forward only), got %d",
                b->matmul_precision);
    EVI_REQUIRE(!(b->matmul_precision == 2 && bw), "matmul_precision 2 (f16x2) is an evaluation-time option: the backward runs with 0 or 1");
    const SingleProductScope precision_scope(b->matmul_precision);
    const int S = 1 + w->dde_rounds + w->dde_reverse_rounds;
    const int F = 2 * 2 * S;
    const FwdLayout L = fwd_layout(N, E, B, D, H, F, b->num_relations, 2);
    EVI_REQUIRE(workspace, "evi_retriever_forward: null workspace");
    if (workspace_bytes < L.total)
        return fail(EVI_ERR_NOMEM, "evi_retriever_forward: workspace %zu B < %zu B", workspace_bytes, L.total);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(workspace);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    auto I32 = [&](size_t off) { return reinterpret_cast<int32_t*>(base + off); };
    // node-level results: in the workspace, or — training, when `saved` has room behind its per-edge rows — in `saved`, where the
    // backward of the same step finds them (nodes_replayed) instead of recomputing
    const size_t edge_saved_bytes = align_up((size_t)E * saved_floats_per_edge(D, H, dirs) * sizeof(float), 256);
    const NodeKeep NK = node_keep_layout(N, E, B, D, H, F, b->num_relations, L.dedupe);
    char* nkeep = nullptr;
    {
        char* sv = bw ? const_cast<char*>(bw->saved) : static_cast<char*>(out->saved);
        const size_t have = bw ? bw->saved_bytes : out->saved_bytes;
        if (sv && have >= edge_saved_bytes + NK.total) nkeep = sv + edge_saved_bytes;
    }
    const bool nodes_replayed = bw && nkeep;
    auto NF = [&](size_t ws_off, size_t keep_off) { return reinterpret_cast<float*>(nkeep ? nkeep + keep_off : base + ws_off); };
    auto NI = [&](size_t ws_off, size_t keep_off) { return reinterpret_cast<int32_t*>(nkeep ? nkeep + keep_off : base + ws_off); };
    float* node_repr = NF(L.node_repr, NK.node_repr);
    float* non_text = NF(L.non_text, NK.non_text);
    float* q_proj = NF(L.q_proj, NK.q_proj);
    float* gate_q = NF(L.gate_q, NK.gate_q);
    float* bias_q = NF(L.bias_q, NK.bias_q);
    float* rel_repr = NF(L.rel_repr, NK.rel_repr);
    // the relation table handed over (relation_rows): its rows ARE the de-duplicated relation rows, nothing is gathered or kept
    const bool table_given = L.dedupe && b->relation_rows;
    float* rel_rows = table_given ? const_cast<float*>(b->relation_rows) : NF(L.rel_rows, NK.rel_rows);
    float* ns = nkeep ? NF(0, NK.ns) : (out->node_struct ? out->node_struct : F32(L.ns));
    int32_t *in_ptr = NI(L.in_ptr, NK.in_ptr), *in_nbr = NI(L.in_nbr, NK.in_nbr), *in_eid = NI(L.in_eid, NK.in_eid);
    int32_t *out_ptr = NI(L.out_ptr, NK.out_ptr), *out_nbr = NI(L.out_nbr, NK.out_nbr), *out_eid = NI(L.out_eid, NK.out_eid);
    void* wsplit = base + L.wsplit;
    int rc;
    // weight-derived pieces: from the caller's prepared buffer when it keeps one, else made below in the workspace
    const char* prep = static_cast<const char*>(w->prepared);
    const PrepLayout PL = prep_layout(D, H, F);
    auto planes = [&](size_t off) -> const void* { return prep ? prep + off : nullptr; };

    if (!nodes_replayed) {  // (the backward of a step whose forward kept its node-level results skips 1 and 2)
    // 1. projections
    if ((rc = scorer_gemm(b->node_embeddings, N, D, D, w->entity_w, D, D, w->entity_b, 1, node_repr, D, wsplit, st, planes(PL.p_entity)))) return rc;
    if ((rc = scorer_gemm(w->non_text_emb, 1, D, D, w->entity_w, D, D, w->entity_b, 1, non_text, D, wsplit, st, planes(PL.p_entity)))) return rc;
    hipLaunchKernelGGL(k_overwrite_non_text, dim3((unsigned)N), dim3(256), 0, st, node_repr, b->node_embedding_ids,
                       non_text, N, D);
    EVI_LAUNCH_CHECK();
    if ((rc = scorer_gemm(b->question_emb, B, D, D, w->query_w, D, D, w->query_b, 1, q_proj, D, wsplit, st, planes(PL.p_query)))) return rc;
    if ((rc = scorer_gemm(q_proj, B, D, D, w->q_gate_w, D, D, w->q_gate_b, 2, gate_q, D, wsplit, st, planes(PL.p_gate)))) return rc;
    if ((rc = scorer_gemm(q_proj, B, D, D, w->q_bias_w, D, D, w->q_bias_b, 1, bias_q, D, wsplit, st, planes(PL.p_bias)))) return rc;
    if (L.dedupe) {
        const int64_t R = b->num_relations;
        int32_t* first = I32(L.rel_first);
        // out-of-range relation ids: OR-ed into the caller's sticky status word when it gave one (read once per epoch by the
        // host mirror: Retriever.check_deferred), else into a scratch word nobody reads
        int32_t* status = out->status ? out->status : I32(L.status);
        if (!out->status) EVI_HIP_CHECK(hipMemsetAsync(status, 0, 4, st));
        if (table_given) {
            hipLaunchKernelGGL(k_check_relation_range, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, b->edge_attr, E, R, status);
        } else {
            hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, first, R, 0x7FFFFFFF);
            hipLaunchKernelGGL(k_first_edge_of_relation, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, b->edge_attr,
                               E, R, first, status);
            hipLaunchKernelGGL(k_gather_relation_rows, dim3((unsigned)R), dim3(256), 0, st, b->edge_embeddings, first, D,
                               rel_rows);
        }
        EVI_LAUNCH_CHECK();
        if ((rc = scorer_gemm(rel_rows, R, D, D, w->relation_w, D, D, w->relation_b, 1, rel_repr, D, wsplit, st, planes(PL.p_rel)))) return rc;
    } else {
        if ((rc = scorer_gemm(b->edge_embeddings, E, D, D, w->relation_w, D, D, w->relation_b, 1, rel_repr, D, wsplit, st, planes(PL.p_rel)))) return rc;
    }

    // 2. structure features
    if ((rc = evi_graph_csr(b->edge_index, E, b->node_ptr, b->edge_ptr, B, N, in_ptr, in_nbr, in_eid, out_ptr, out_nbr, out_eid,
                            base + L.csr_ws, evi_graph_csr_workspace_bytes(N), stream)))
        return rc;
    if ((rc = evi_dde_node_struct_graphs(b->topic_one_hot, b->topic_stride, 2, N, b->node_ptr, B, in_ptr, in_nbr, out_ptr, out_nbr,
                                         w->dde_rounds, w->dde_reverse_rounds, ns, stream)))
        return rc;
    if (nkeep && out->node_struct)  // the caller wants the structure features as well: a copy of the kept rows
        EVI_HIP_CHECK(hipMemcpyAsync(out->node_struct, ns, (size_t)N * (F / 2) * sizeof(float), hipMemcpyDeviceToDevice, st));
    }

    // 3-5. factored state_net.0 (see the header), per edge chunk
    auto PF = [&](size_t off) { return reinterpret_cast<float*>(const_cast<char*>(prep) + off); };
    float *wa = prep ? PF(PL.wa) : F32(L.wa), *wb = prep ? PF(PL.wb) : F32(L.wb), *wc = prep ? PF(PL.wc) : F32(L.wc);
    float *wd = prep ? PF(PL.wd) : F32(L.wd), *vhead = prep ? PF(PL.vhead) : F32(L.vhead), *hcn = NF(L.hcn, NK.hcn);
    float* wt = prep ? PF(PL.wt) : F32(L.wt);
    if (!prep) {
        hipLaunchKernelGGL(k_slice_state0, dim3(H), dim3(256), 0, st, w->state0_w, H, D, wa, wb, wc, wd);
        hipLaunchKernelGGL(k_transpose, dim3((F * D + 255) / 256), dim3(256), 0, st, w->struct_w, D, F, wt);
        EVI_LAUNCH_CHECK();
    }
    if (!prep) {  // the folded head v = W2^T w, v[H] = w . b2 + b
        float* partial = F32(L.fold);
        hipLaunchKernelGGL(k_fold_head_partial, dim3((H + 255) / 256, kFoldSlices), dim3(256), 0, st, w->state4_w,
                           w->score_w, H, partial);
        hipLaunchKernelGGL(k_fold_head_final, dim3((H + 255) / 256), dim3(256), 0, st, partial, w->state4_b, w->score_w,
                           w->score_b, H, vhead);
        EVI_LAUNCH_CHECK();
    }
    if (!nodes_replayed && (rc = scorer_gemm(node_repr, N, D, D, wc, H, D, nullptr, 0, hcn, H, wsplit, st, planes(PL.p_wc)))) return rc;
    const int dpl_d = dpl_for(D), dpl_h = dpl_for(H);
    const size_t feat_lds = (size_t)(F + 4) * D * sizeof(float);
    // ---- backward: set-up ----------------------------------------------------------------------------------------
    BwdLayout BL{};
    char* bws = nullptr;
    auto G = [&](const float* p) { return const_cast<float*>(p); };  // gradient buffers live in a weights-shaped struct
    auto BF = [&](size_t off) { return reinterpret_cast<float*>(bws + off); };
    ZeroTable ztab{};  // buffers to clear: collected, then cleared by ONE launch (flush_zero) before anything accumulates into them
    auto zero = [&](float* p, int64_t n) {
        if (n <= 0) return;
        if (ztab.count == kZeroMax) {
            hipLaunchKernelGGL(k_zero_multi, dim3(64, (unsigned)ztab.count), dim3(256), 0, st, ztab);
            ztab.count = 0;
        }
        ztab.p[ztab.count] = p;
        ztab.n[ztab.count] = n;
        ++ztab.count;
    };
    auto flush_zero = [&]() {
        if (ztab.count > 0) hipLaunchKernelGGL(k_zero_multi, dim3(64, (unsigned)ztab.count), dim3(256), 0, st, ztab);
        ztab.count = 0;
    };
    auto transpose = [&](const float* src, int R_, int C_, float* dst) {  // dst [C_, R_] = src[R_, C_]^T
        hipLaunchKernelGGL(k_transpose_pad, dim3((unsigned)((R_ + 31) / 32), (unsigned)((C_ + 31) / 32)), dim3(256), 0, st, src,
                           (int64_t)R_, C_, (int64_t)C_, dst, (int64_t)R_);
    };
    if (bw) {
        BL = bwd_layout(N, E, B, D, H, F, L.dedupe ? b->num_relations : 0, L.ec);
        bws = bw->ws;
        const EviRetrieverWeights* g = bw->g;
        zero(G(g->entity_w), (int64_t)D * D); zero(G(g->entity_b), D); zero(G(g->relation_w), (int64_t)D * D); zero(G(g->relation_b), D);
        zero(G(g->query_w), (int64_t)D * D); zero(G(g->query_b), D); zero(G(g->non_text_emb), D);
        zero(G(g->q_gate_w), (int64_t)D * D); zero(G(g->q_gate_b), D); zero(G(g->q_bias_w), (int64_t)D * D); zero(G(g->q_bias_b), D);
        zero(G(g->struct_w), (int64_t)D * F); zero(G(g->struct_b), D); zero(G(g->struct_ln_w), D); zero(G(g->struct_ln_b), D);
        zero(G(g->struct_gate_w), D); zero(G(g->struct_gate_b), 1);
        zero(G(g->state0_w), (int64_t)H * (3 * D + 1)); zero(G(g->state0_b), H); zero(G(g->state_ln_w), H); zero(G(g->state_ln_b), H);
        zero(G(g->state4_w), (int64_t)H * H); zero(G(g->state4_b), H); zero(G(g->score_w), H); zero(G(g->score_b), 1);
        zero(BF(BL.gWa), (int64_t)H * D); zero(BF(BL.gWb), (int64_t)H * D); zero(BF(BL.gWc), (int64_t)H * D); zero(BF(BL.gwd), H);
        zero(BF(BL.ysum), H); zero(BF(BL.ssum), 1);
        flush_zero();
        transpose(wa, H, D, BF(BL.WaT));
        transpose(wb, H, D, BF(BL.WbT));
        transpose(wc, H, D, BF(BL.WcT));
        transpose(w->q_gate_w, D, D, BF(BL.WgT));
        transpose(w->q_bias_w, D, D, BF(BL.WbqT));
        transpose(w->entity_w, D, D, BF(BL.WeT));
        EVI_LAUNCH_CHECK();
    }
    // training: the forward writes its per-edge intermediates into the caller's `saved` buffer and the backward replays them
    // instead of recomputing the per-edge forward (edge features, three GEMMs, the combine)
    float* saved = bw ? reinterpret_cast<float*>(const_cast<char*>(bw->saved)) : static_cast<float*>(out->saved);
    const bool replay = bw && bw->saved;
    if (saved) {
        const size_t need = (size_t)E * saved_floats_per_edge(D, H, dirs) * sizeof(float);
        const size_t have = bw ? bw->saved_bytes : out->saved_bytes;
        if (have < need) return fail(EVI_ERR_NOMEM, "evi_retriever: saved buffer %zu B < %zu B", have, need);
    }
    // r_ctx Wc^T once per (relation, graph) pair (forward-only calls; the backward and the saved-activation forward keep per-edge rows)
    const bool pairs = L.pair_max > 0 && !bw && !saved;
    if (pairs) {
        const int64_t R = b->num_relations;
        int32_t *table = I32(L.pair_table), *count = I32(L.pair_count), *key = I32(L.pair_key);
        EVI_HIP_CHECK(hipMemsetAsync(table, 0, (size_t)B * R * 4, st));
        hipLaunchKernelGGL(k_pair_mark, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, b->edge_attr, b->edge_batch, E, R, table);
        hipLaunchKernelGGL(k_pair_rank, dim3((unsigned)B), dim3(1024), 0, st, table, R, count);
        hipLaunchKernelGGL(k_pair_slots, dim3((unsigned)B), dim3(256), 0, st, table, R, B, count, key);
        hipLaunchKernelGGL(k_pair_rows, dim3((unsigned)((L.pair_max + 3) / 4)), dim3(256), 0, st, key, count + B, R, rel_repr, gate_q, bias_q,
                           D, F32(L.pair_rcx));
        EVI_LAUNCH_CHECK();
        // always the tiled split-bf16 kernel, also for a handful of pairs: a row's result does not depend on where it sits in the
        // product, so the pair rows equal the per-edge rows bit for bit (the exact-f32 skinny kernel would not)
        if (prep && t_gemm_single != 2) {
            if ((rc = launch_gemm_nt_bf16x3_wplanes(F32(L.pair_rcx), L.pair_max, D, D, planes(PL.p_wc), H, nullptr, 0, F32(L.pair_rc), H, st,
                                                    t_gemm_single, count + B)))
                return rc;
        } else if ((rc = launch_gemm_nt_bf16x3(F32(L.pair_rcx), L.pair_max, D, D, wc, H, D, nullptr, 0, F32(L.pair_rc), H, wsplit, st,
                                               t_gemm_single, count + B))) {
            return rc;
        }
    }
    // (r03: alternating the chunks between the caller's stream and a side stream, so that one chunk's HBM-bound per-edge kernels
    // run under the other's GEMMs, was built and measured: 3.471 -> 3.427 ms per forward, nothing once the pair rows above were
    // in.  The GEMM workgroup owns its CU — 2 x 250 of the 512 VGPRs of every SIMD, 128 KB of LDS — so nothing co-resides with it
    // and two streams only interleave whole kernels.  Removed.)
    for (int64_t e0 = 0; e0 < E; e0 += L.ec) {
        const int64_t ec = (E - e0) < L.ec ? (E - e0) : L.ec;
        EdgeFeatArgs a;
        a.edge_index = b->edge_index;
        a.E = E;
        a.edge_batch = b->edge_batch;
        a.edge_attr = b->edge_attr;
        a.rel_by_edge = L.dedupe ? 0 : 1;
        a.R = b->num_relations;
        a.node_repr = node_repr;
        a.rel_repr = rel_repr;
        a.gate_q = gate_q;
        a.bias_q = bias_q;
        a.node_struct = ns;
        a.F = F;
        a.struct_wt = wt;
        a.struct_b = w->struct_b;
        a.struct_ln_w = w->struct_ln_w;
        a.struct_ln_b = w->struct_ln_b;
        a.struct_gate_w = w->struct_gate_w;
        a.struct_gate_b = w->struct_gate_b;
        a.D = D;
        a.e_begin = e0;
        a.e_count = ec;
        a.dir_fwd = dir_fwd;
        a.dir_bwd = dir_bwd;
        const int64_t M = (int64_t)dirs * ec;
        float *pP = F32(L.P), *pRCX = F32(L.RCX), *pXS = F32(L.XS), *pAux = F32(L.aux), *pPA = F32(L.PA), *pRC = F32(L.RC), *pSB = F32(L.SB);
        if (saved) {
            float* q = saved + (size_t)e0 * saved_floats_per_edge(D, H, dirs);
            pP = q, q += ec * D;
            pRCX = q, q += ec * D;
            pXS = q, q += M * D;
            pPA = q, q += ec * H;
            pRC = q, q += ec * H;
            pSB = q, q += M * H;
            pAux = q;
        }
        a.P = pP;
        a.RCX = pairs ? nullptr : pRCX;
        a.XS = pXS;
        a.aux = pAux;
        static const int ef_threads = [] {
            const char* e = getenv("EVI_EF_THREADS");
            const int n = e ? atoi(e) : 0;
            return (n >= 64 && n <= 1024 && n % 64 == 0) ? n : 1024;
        }();
        const int ef_waves = ef_threads / 64;
        int64_t blocks = (ec + ef_waves - 1) / ef_waves;  // one edge per wave
        if (blocks > 512) blocks = 512;   // 2 blocks per CU fit in LDS
        if (!replay) {
        const int tok = timing_begin(kTimeEdge, st);
        EVI_DPL_DISPATCH(dpl_d, {
            static thread_local bool attr = false;
            if (!attr) {
                EVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_edge_features<DPL>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr = true;
            }
            hipLaunchKernelGGL(k_edge_features<DPL>, dim3((unsigned)blocks), dim3(ef_threads), feat_lds, st, a);
        });
        timing_end(tok, st);
        EVI_LAUNCH_CHECK();
        // (r03: writing P / RCX / XS as bf16 hi / lo planes and multiplying them on a pre-split LDS-DMA GEMM was wired in here,
        // bit-identical, and measured with rocprofv3 on this batch: 288.8 us per launch against 285.0 us for this kernel, the edge
        // kernel 344 against 330 us — no gain inside the pipeline, so that kernel and its entry points were removed)
        if ((rc = scorer_gemm(pP, ec, D, D, wa, H, D, nullptr, 0, pPA, H, wsplit, st, planes(PL.p_wa)))) return rc;
        if (!pairs && (rc = scorer_gemm(pRCX, ec, D, D, wc, H, D, nullptr, 0, pRC, H, wsplit, st, planes(PL.p_wc)))) return rc;
        if ((rc = scorer_gemm(pXS, M, D, D, wb, H, D, w->state0_b, 0, pSB, H, wsplit, st, planes(PL.p_wb)))) return rc;
        }
        CombineArgs c;
        c.edge_index = b->edge_index;
        c.E = E;
        c.e_begin = e0;
        c.e_count = ec;
        c.PA = pPA;
        c.RC = pairs ? reinterpret_cast<float*>(base + L.pair_rc) : pRC;
        c.pair_table = pairs ? I32(L.pair_table) : nullptr;
        c.edge_attr = b->edge_attr;
        c.edge_batch = b->edge_batch;
        c.R = b->num_relations;
        c.SB = pSB;
        c.HcN = hcn;
        c.aux = pAux;
        c.wd = wd;
        c.ln_w = w->state_ln_w;
        c.ln_b = w->state_ln_b;
        c.H = H;
        c.dir_fwd = dir_fwd;
        c.dir_bwd = dir_bwd;
        c.h1c = out->edge_features ? F32(L.h1n) : nullptr;
        c.v = vhead;
        c.edge_bias = b->edge_bias;
        c.drop_thr = drop_thr;
        c.drop_scale = drop_thr ? 65536.0f / (float)(65536u - drop_thr) : 1.0f;
        c.drop_seed = b->dropout_seed;
        c.logits = out->logits;
        c.logits_fwd = out->logits_fwd;
        c.logits_bwd = out->logits_bwd;
        const dim3 cgrid((unsigned)((ec + 3) / 4));
        if (!replay) {
            if (c.h1c) {
                EVI_DPL_DISPATCH(dpl_h, hipLaunchKernelGGL((k_state_combine<DPL, true>), cgrid, dim3(256), 0, st, c));
            } else {
                EVI_DPL_DISPATCH(dpl_h, hipLaunchKernelGGL((k_state_combine<DPL, false>), cgrid, dim3(256), 0, st, c));
            }
        }
        EVI_LAUNCH_CHECK();
        if (out->edge_features && !replay)  // state_net.4 on the combined rows, straight into the caller's [E, H] output
            if ((rc = scorer_gemm(F32(L.h1n), ec, H, H, w->state4_w, H, H, w->state4_b, 0, out->edge_features + e0 * H, H, wsplit,
                                  st, planes(PL.p_s4))))
                return rc;
        if (!bw) continue;
        // ---- backward of this chunk (scorer_bwd.hpp) ----------------------------------------------------------------
        const EviRetrieverWeights* g = bw->g;
        CombineBwdArgs cb;
        cb.f = c;
        cb.dlogits = bw->dlogits;
        cb.DZ = BF(BL.DZ);
        cb.DPA = BF(BL.DPA);
        cb.DRC = BF(BL.DRC);
        cb.DDF = BF(BL.DDF);
        cb.daux = BF(BL.daux);
        cb.part = BF(BL.partC);
        EVI_DPL_DISPATCH(dpl_h, hipLaunchKernelGGL((k_combine_bwd<DPL>), dim3(BL.gridC), dim3(256), 0, st, cb));
        EVI_LAUNCH_CHECK();
        {   // column partials: [gridC][5][H] then [gridC] — reduce each of the five vectors and S (accumulating over chunks)
            float* dst5[5] = {G(g->state_ln_w), G(g->state_ln_b), BF(BL.ysum), BF(BL.gwd), G(g->state0_b)};
            // the partial rows interleave the five vectors: one [gridC][5H] table, reduced straight into the five gradients
            if ((rc = colsum_into_multi(BF(BL.partC), BL.gridC, 5, H, dst5, BL, bws, st))) return rc;
            if ((rc = colsum_into(BF(BL.partC) + (int64_t)BL.gridC * 5 * H, BL.gridC, 1, BF(BL.ssum), 1, BL, bws, st))) return rc;
        }
        // d(state_net.0 inputs): dP = dPA Wa, dRCX = dRC Wc, dXS = dz Wb   (NT GEMMs against the transposed blocks)
        if ((rc = scorer_gemm(BF(BL.DPA), ec, H, H, BF(BL.WaT), D, H, nullptr, 0, BF(BL.dP), D, wsplit, st))) return rc;
        if ((rc = scorer_gemm(BF(BL.DRC), ec, H, H, BF(BL.WcT), D, H, nullptr, 0, BF(BL.dRCX), D, wsplit, st))) return rc;
        if ((rc = scorer_gemm(BF(BL.DZ), M, H, H, BF(BL.WbT), D, H, nullptr, 0, BF(BL.dXS), D, wsplit, st))) return rc;
        // weight blocks of state_net.0: dWa += dPA^T P, dWc += dRC^T RCX, dWb += dz^T XS
        if ((rc = tn_gemm(BF(BL.DPA), H, pP, D, ec, BF(BL.gWa), 1, BL, bws, st))) return rc;
        if ((rc = tn_gemm(BF(BL.DRC), H, pRCX, D, ec, BF(BL.gWc), 1, BL, bws, st))) return rc;
        if ((rc = tn_gemm(BF(BL.DZ), H, pXS, D, M, BF(BL.gWb), 1, BL, bws, st))) return rc;
        EdgeBwdArgs eb;
        eb.f = a;
        eb.dP = BF(BL.dP);
        eb.dRCX = BF(BL.dRCX);
        eb.dXS = BF(BL.dXS);
        eb.daux = BF(BL.daux);
        eb.DH = BF(BL.DH);
        eb.DT = BF(BL.DT);
        // with a relation table the three uses of d r_ctx are formed from ONE stored array (DBQ) by k_segment_rowsum_mul
        eb.DRR = L.dedupe ? nullptr : BF(BL.DRR);
        eb.DGQ = L.dedupe ? nullptr : BF(BL.DGQ);
        eb.DBQ = BF(BL.DBQ);
        eb.DU = BF(BL.DU);
        eb.SX = BF(BL.SX);
        eb.part = BF(BL.partE);
        {
            const size_t lds = (size_t)(F + 4) * D * sizeof(float);
            EVI_DPL_DISPATCH(dpl_d, {
                static thread_local bool attr = false;
                if (!attr) {
                    EVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_edge_struct_bwd<DPL>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    attr = true;
                }
                hipLaunchKernelGGL(k_edge_struct_bwd<DPL>, dim3(BL.gridE), dim3(BL.wavesE * 64), lds, st, eb);
                const unsigned gt = (unsigned)((ec + 3) / 4 < 8192 ? (ec + 3) / 4 : 8192);
                hipLaunchKernelGGL(k_edge_translate_bwd<DPL>, dim3(gt > 0 ? gt : 1), dim3(256), 0, st, eb);
            });
            EVI_LAUNCH_CHECK();
            // the per-wave partial table [gridE * waves][4 D] -> one row (two ordered stages), then into the four gradients
            const int64_t prow = (int64_t)BL.gridE * BL.wavesE;
            float* dst4[4] = {G(g->struct_ln_w), G(g->struct_ln_b), G(g->struct_gate_w), G(g->struct_b)};
            if ((rc = colsum_into_multi(BF(BL.partE), prow, 4, D, dst4, BL, bws, st))) return rc;
            if ((rc = colsum_into(BF(BL.partE) + prow * 4 * D, prow, 1, G(g->struct_gate_b), 1, BL, bws, st))) return rc;
        }
        // struct_proj.0: weight [D, F] += dU^T SX (its bias gradient, the column sums of dU, came with the kernel's partials)
        if ((rc = tn_gemm(BF(BL.DU), D, BF(BL.SX), F, M, G(g->struct_w), 1, BL, bws, st))) return rc;
    }
    if (!bw) return EVI_OK;
    // ---- backward: once per batch ----------------------------------------------------------------------------------
    {
        const EviRetrieverWeights* g = bw->g;
        auto blocks_of = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
        // nodes: gather the per-edge gradients through the CSR, add the Wc path (HcN = node_repr Wc^T)
        hipLaunchKernelGGL(k_node_gather_grad, dim3((unsigned)((N * ((D + 255) / 256 + (H + 255) / 256) + 3) / 4)), dim3(256), 0, st, in_ptr, in_eid, out_ptr,
                           out_eid, BF(BL.DH), BF(BL.DT), D, BF(BL.DDF), H, BF(BL.dNR), BF(BL.dHcN), (int64_t)N);
        EVI_LAUNCH_CHECK();
        if ((rc = scorer_gemm(BF(BL.dHcN), N, H, H, BF(BL.WcT), D, H, nullptr, 0, BF(BL.tmpN), D, wsplit, st))) return rc;
        // dNR += the Wc path; the id-0 rows set aside for the non-text embedding (into tmpN, in place); tanh backward: one pass
        hipLaunchKernelGGL(k_node_grad_finish, blocks_of((N * D + 3) / 4), dim3(256), 0, st, BF(BL.dNR), BF(BL.tmpN), node_repr,
                           b->node_embedding_ids, N, D, BF(BL.tmpN));
        if ((rc = tn_gemm(BF(BL.dHcN), H, node_repr, D, N, BF(BL.gWc), 1, BL, bws, st))) return rc;
        hipLaunchKernelGGL(k_merge_state0, dim3(H), dim3(256), 0, st, BF(BL.gWa), BF(BL.gWb), BF(BL.gWc), BF(BL.gwd), H, D,
                           G(g->state0_w));
        // the non-text embedding: its projection replaced every node row with embedding id 0
        float* dnt = BF(BL.dnt);
        if ((rc = colsum_into(BF(BL.tmpN), N, D, dnt, 0, BL, bws, st))) return rc;
        hipLaunchKernelGGL(k_act_bwd, blocks_of(D), dim3(256), 0, st, dnt, non_text, (int64_t)1, D, 1, (const int64_t*)nullptr);
        EVI_LAUNCH_CHECK();
        if ((rc = scorer_gemm(dnt, 1, D, D, BF(BL.WeT), D, D, nullptr, 0, G(g->non_text_emb), D, wsplit, st))) return rc;
        // entity_proj: dpre = dNR (1 - NR^2), rows with id 0 dropped — done by k_node_grad_finish above
        if ((rc = tn_gemm(BF(BL.dNR), D, b->node_embeddings, D, N, G(g->entity_w), 0, BL, bws, st))) return rc;
        if ((rc = tn_gemm(dnt, D, w->non_text_emb, D, 1, G(g->entity_w), 1, BL, bws, st))) return rc;
        if ((rc = colsum_into(BF(BL.dNR), N, D, G(g->entity_b), 0, BL, bws, st))) return rc;
        hipLaunchKernelGGL(k_add_inplace, blocks_of(D), dim3(256), 0, st, G(g->entity_b), dnt, (int64_t)D);
        // relation_proj
        if (L.dedupe) {
            const int64_t R = b->num_relations;
            // d rel_repr[r] = sum over the edges of relation r of d r_ctx[e] * gate_q[graph(e)]
            if ((rc = segment_rowsum_mul_into(BF(BL.DBQ), D, bw->rel_ptr, bw->rel_perm, R, E, nullptr, BF(BL.dRRu), gate_q, b->edge_batch,
                                              (int64_t)B - 1, BL, bws, st)))
                return rc;
            hipLaunchKernelGGL(k_act_bwd, blocks_of(R * D), dim3(256), 0, st, BF(BL.dRRu), rel_repr, R, D, 1, (const int64_t*)nullptr);
            EVI_LAUNCH_CHECK();
            if ((rc = tn_gemm(BF(BL.dRRu), D, rel_rows, D, R, G(g->relation_w), 0, BL, bws, st))) return rc;
            if ((rc = colsum_into(BF(BL.dRRu), R, D, G(g->relation_b), 0, BL, bws, st))) return rc;
        } else {
            hipLaunchKernelGGL(k_act_bwd, blocks_of(E * D), dim3(256), 0, st, BF(BL.DRR), rel_repr, E, D, 1, (const int64_t*)nullptr);
            EVI_LAUNCH_CHECK();
            if ((rc = tn_gemm(BF(BL.DRR), D, b->edge_embeddings, D, E, G(g->relation_w), 0, BL, bws, st))) return rc;
            if ((rc = colsum_into(BF(BL.DRR), E, D, G(g->relation_b), 0, BL, bws, st))) return rc;
        }
        // question side: gate (sigmoid) and bias (tanh) of the projected question, then query_proj (tanh)
        if (L.dedupe) {  // d bias_q[g] = sum of d r_ctx over the graph, d gate_q[g] = sum of d r_ctx * rel_repr[relation]: one pass
            if ((rc = segment_rowsum_mul_into(BF(BL.DBQ), D, b->edge_ptr, nullptr, B, E, BF(BL.dBQ), BF(BL.dGQ), rel_repr, b->edge_attr,
                                              b->num_relations - 1, BL, bws, st)))
                return rc;
        } else {
            if ((rc = segment_rowsum_into(BF(BL.DGQ), D, b->edge_ptr, nullptr, B, E, BF(BL.dGQ), BL, bws, st))) return rc;
            if ((rc = segment_rowsum_into(BF(BL.DBQ), D, b->edge_ptr, nullptr, B, E, BF(BL.dBQ), BL, bws, st))) return rc;
        }
        hipLaunchKernelGGL(k_act_bwd, blocks_of((int64_t)B * D), dim3(256), 0, st, BF(BL.dGQ), gate_q, (int64_t)B, D, 2, (const int64_t*)nullptr);
        hipLaunchKernelGGL(k_act_bwd, blocks_of((int64_t)B * D), dim3(256), 0, st, BF(BL.dBQ), bias_q, (int64_t)B, D, 1, (const int64_t*)nullptr);
        EVI_LAUNCH_CHECK();
        if ((rc = tn_gemm(BF(BL.dGQ), D, q_proj, D, B, G(g->q_gate_w), 0, BL, bws, st))) return rc;
        if ((rc = tn_gemm(BF(BL.dBQ), D, q_proj, D, B, G(g->q_bias_w), 0, BL, bws, st))) return rc;
        if ((rc = colsum_into(BF(BL.dGQ), B, D, G(g->q_gate_b), 0, BL, bws, st))) return rc;
        if ((rc = colsum_into(BF(BL.dBQ), B, D, G(g->q_bias_b), 0, BL, bws, st))) return rc;
        float* dQP = BF(BL.dQP);
        float* dQP2 = dQP + (int64_t)B * D;
        if ((rc = scorer_gemm(BF(BL.dGQ), B, D, D, BF(BL.WgT), D, D, nullptr, 0, dQP, D, wsplit, st))) return rc;
        if ((rc = scorer_gemm(BF(BL.dBQ), B, D, D, BF(BL.WbqT), D, D, nullptr, 0, dQP2, D, wsplit, st))) return rc;
        hipLaunchKernelGGL(k_add_inplace, blocks_of((int64_t)B * D), dim3(256), 0, st, dQP, dQP2, (int64_t)B * D);
        hipLaunchKernelGGL(k_act_bwd, blocks_of((int64_t)B * D), dim3(256), 0, st, dQP, q_proj, (int64_t)B, D, 1, (const int64_t*)nullptr);
        EVI_LAUNCH_CHECK();
        if ((rc = tn_gemm(dQP, D, b->question_emb, D, B, G(g->query_w), 0, BL, bws, st))) return rc;
        if ((rc = colsum_into(dQP, B, D, G(g->query_b), 0, BL, bws, st))) return rc;
        // the head
        hipLaunchKernelGGL(k_head_grads, dim3(H), dim3(256), 0, st, w->state4_w, w->state4_b, w->score_w, BF(BL.ysum), BF(BL.ssum), H,
                           G(g->state4_w), G(g->state4_b), G(g->score_w), G(g->score_b));
        EVI_LAUNCH_CHECK();
    }
    return EVI_OK;
}

extern "C" int evi_retriever_forward(const EviRetrieverWeights* w, const EviRetrieverBatch* b, int direction_mode,
                                     const EviRetrieverOutput* out, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    return retriever_run(w, b, direction_mode, out, workspace, workspace_bytes, stream, nullptr);
}

extern "C" size_t evi_retriever_saved_bytes(int64_t E, int D, int H, int direction_mode) {
    if (E < 0 || D < 1 || H < 1 || direction_mode < 0 || direction_mode > 2) return 0;
    return (size_t)(E > 0 ? E : 1) * saved_floats_per_edge(D, H, direction_mode == 0 ? 2 : 1) * sizeof(float);
}

extern "C" size_t evi_retriever_saved_bytes_full(int64_t N, int64_t E, int B, int D, int H, int dde_rounds, int dde_reverse_rounds,
                                                 int64_t num_relations, int direction_mode) {
    const size_t edges = evi_retriever_saved_bytes(E, D, H, direction_mode);
    if (edges == 0 || N < 0 || B < 1 || dde_rounds < 0 || dde_reverse_rounds < 0) return 0;
    const int F = 2 * 2 * (1 + dde_rounds + dde_reverse_rounds);
    const int dedupe = (num_relations > 0 && num_relations <= E) ? 1 : 0;
    // (the per-edge part is sized for at least one edge; retriever_run places the node section behind E rows)
    return align_up(edges, 256) + node_keep_layout(N, E, B, D, H, F, num_relations, dedupe).total;
}

extern "C" size_t evi_retriever_backward_workspace_bytes(int64_t N, int64_t E, int B, int D, int H, int dde_rounds,
                                                         int dde_reverse_rounds, int64_t num_relations) {
    if (N < 0 || E < 0 || B < 1 || D < 1 || H < 1) return 0;
    const int F = 2 * 2 * (1 + dde_rounds + dde_reverse_rounds);
    const FwdLayout L = fwd_layout(N, E, B, D, H, F, num_relations, 2);
    return align_up(L.total, 256) + bwd_layout(N, E, B, D, H, F, L.dedupe ? num_relations : 0, L.ec).total + align_up((size_t)(E > 0 ? E : 1) * 3 * 4, 256);
}

extern "C" int evi_retriever_backward(const EviRetrieverWeights* w, const EviRetrieverBatch* b, int direction_mode,
                                      const float* dlogits, const EviRetrieverWeights* grads, const int64_t* rel_perm,
                                      const int64_t* rel_ptr, void* workspace, size_t workspace_bytes, const void* saved,
                                      size_t saved_bytes, void* stream) {
    EVI_REQUIRE(w && b && grads && dlogits, "evi_retriever_backward: null pointer");
    EVI_REQUIRE(!w->prepared, "evi_retriever_backward: pass the weights without a prepared buffer (training changes them every step)");
    const int64_t N = b->num_nodes, E = b->num_edges;
    const int D = w->emb_dim, H = w->hidden_dim;
    EVI_REQUIRE(b->num_graphs >= 1 && D >= 1 && H >= 1, "evi_retriever_backward: bad sizes");
    if (E == 0) return EVI_OK;
    const int F = 2 * 2 * (1 + w->dde_rounds + w->dde_reverse_rounds);
    const FwdLayout L = fwd_layout(N, E, b->num_graphs, D, H, F, b->num_relations, 2);
    EVI_REQUIRE(!L.dedupe || (rel_perm && rel_ptr), "evi_retriever_backward: rel_perm / rel_ptr are required when num_relations is given");
    const size_t need = evi_retriever_backward_workspace_bytes(N, E, b->num_graphs, D, H, w->dde_rounds, w->dde_reverse_rounds, b->num_relations);
    EVI_REQUIRE(workspace, "evi_retriever_backward: null workspace");
    if (workspace_bytes < need) return fail(EVI_ERR_NOMEM, "evi_retriever_backward: workspace %zu B < %zu B", workspace_bytes, need);
    char* base = static_cast<char*>(workspace);
    const size_t fwd_bytes = align_up(L.total, 256);
    const BwdLayout BL = bwd_layout(N, E, b->num_graphs, D, H, F, L.dedupe ? b->num_relations : 0, L.ec);
    float* scratch_logits = reinterpret_cast<float*>(base + fwd_bytes + BL.total);  // the recomputed logits (3 x [E])
    EviRetrieverOutput o{};
    o.logits = scratch_logits;
    o.logits_fwd = scratch_logits + E;
    o.logits_bwd = scratch_logits + 2 * E;
    BwdCtx ctx{dlogits, grads, rel_perm, rel_ptr, base + fwd_bytes, static_cast<const char*>(saved), saved_bytes};
    return retriever_run(w, b, direction_mode, &o, base, fwd_bytes, stream, &ctx);
}
