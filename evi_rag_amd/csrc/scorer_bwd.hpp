// Backward of the edge scorer: gradients of a scalar loss with respect to every Retriever parameter, given
// dL/dlogits — the autograd of src/models/components/retriever.py:195-289, 403-507, with the training-mode dropout of state_net
// (the mask is regenerated from the forward's seed: dropout_mul4) and the hide-and-seek bias (an additive constant: no
// gradient of its own).  Included by scorer.hip (it needs the forward's device helpers and argument structs); everything is
// inside namespace evi.
//
// Structure.  The per-edge forward (operand rows P / RCX / XS, product rows PA / RC / SB, aux) is REPLAYED from the buffer the
// training forward kept (EviRetrieverOutput.saved) — or recomputed chunk by chunk when none was kept — then per chunk:
//   k_combine_bwd        logits -> pre-LayerNorm state rows:  dz_dir [2 Ec, H], and what the factored state_net.0 needs
//                        (dPA = nav_f dz_f + nav_b dz_b, dRC = dz_f + dz_b, dDiff = dz_f - dz_b, dnav, d(-dist)), plus the
//                        column sums behind d state_net.1.{weight,bias}, d wd, d state_net.0.bias and the folded head
//   three NT GEMMs       dP = dPA Wa, dRCX = dRC Wc, dXS = dz Wb        (weights transposed once: [D, H] row-major)
//   k_edge_translate_bwd -> per-edge dh, dt, d rel_repr row, d gate_q / d bias_q contributions
//   k_edge_struct_bwd    -> d struct pre-activation rows (dU), the struct MLP's input rows, its LayerNorm / gate column partials
//   TN products          dWa += dPA^T P, dWc += dRC^T RCX, dWb += dz^T XS, dWs += dU^T struct   (tn_gemm: the TN split-bf16
//                        kernel of gemm_tn.hip — operands read as they lie, split-K + an ordered reduction)
// and once per batch: node / relation / graph segment sums (f64, CSR- or sort-ordered: no float atomics), the tanh / sigmoid
// backward of the projections and their TN products.  All reductions run in a fixed order for a given batch shape.
//
// state_net.4 and score_head need no GEMM: with the folded head, d score_w = W2 ysum + b2 S, d score_b = S,
// d W2 = w (x) ysum, d b2 = w S, where ysum = sum_{e,dir} dlg_dir y_dir and S = sum dlg_dir.
#pragma once

namespace evi {

__device__ inline float gelu_erf_grad(float x) {
    // d/dx [0.5 x (1 + erf(x / sqrt2))] = 0.5 (1 + erf(x / sqrt2)) + x phi(x)
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
    return fmaf(x, 0.3989422804014327f * __expf(-0.5f * x * x), cdf);
}
// GELU and its derivative together: one reciprocal and ONE exponential (erf's exp(-(x / sqrt2)^2) is phi's exp(-x^2 / 2)).
// y is computed by gelu_erf's own expression, so it equals the forward's value bit for bit.
__device__ inline void gelu_erf_both(float x, float& y, float& dy) {
    const float xs = x * 0.70710678118654752440f;
    const float ax = fabsf(xs);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float ex = __expf(-ax * ax);
    const float er = copysignf(1.0f - p * t * ex, xs);
    y = 0.5f * x * (1.0f + er);
    dy = fmaf(x, 0.3989422804014327f * ex, 0.5f * (1.0f + er));
}

// ---- generic pieces ---------------------------------------------------------------------------------
__global__ void k_zero_f32(float* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

// dst[c][r] = src[r][c] for r < R, c < C; dst rows are Rp long, columns R..Rp-1 zero.  32 x 32 tiles through LDS.
__global__ __launch_bounds__(256) void k_transpose_pad(const float* __restrict__ src, int64_t R, int C, int64_t ld_src,
                                                       float* __restrict__ dst, int64_t Rp) {
    __shared__ float tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int64_t r = r0 + i;
        const int c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[r * ld_src + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const int64_t r = r0 + tx;
        if (c < C && r < Rp) dst[(int64_t)c * Rp + r] = tile[tx][i];
    }
}

// Up to kZeroMax buffers zeroed by ONE launch (the backward clears ~36 gradient buffers per call: 36 launches of ~4 us each
// were pure launch overhead).  blockIdx.y = buffer, blockIdx.x strides over it.
constexpr int kZeroMax = 40;
struct ZeroTable {
    float* p[kZeroMax];
    int64_t n[kZeroMax];
    int count;
};
__global__ void k_zero_multi(ZeroTable t) {
    const int b = blockIdx.y;
    if (b >= t.count) return;
    float* __restrict__ p = t.p[b];
    const int64_t n = t.n[b];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

// dst[q][i] += sum_s part[s * (Q * W) + q * W + i], s ascending: the second stage of a column sum whose columns are Q vectors of
// W entries bound for Q different gradient buffers (one launch instead of a reduce + Q adds)
struct ReduceDst {
    float* p[8];
};
__global__ void k_reduce_partials_multi(const float* __restrict__ part, int S, int Q, int W, ReduceDst dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Q * W) return;
    const int q = (int)(i / W), c = (int)(i % W);
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += part[(int64_t)s * Q * W + i];
    dst.p[q][c] += acc;  // (the partial sum first, then ONE add into the gradient: what reduce-into-tmp + k_add_inplace did)
}

// out[i] (+)= sum_s part[s * len + i], s ascending
__global__ void k_reduce_partials(const float* __restrict__ part, int S, int64_t len, float* __restrict__ out, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    float acc = accumulate ? out[i] : 0.f;
    for (int s = 0; s < S; ++s) acc += part[(int64_t)s * len + i];
    out[i] = acc;
}

// out[z][s][:] = sum over the z-th of gridDim.z equal slices of the rows p in [ptr[s], ptr[s+1]) of X[perm ? perm[p] : p][:]
// (f64 accumulation in a fixed order; the caller adds the slices in ascending z).  Block = (segment, 256 columns, slice): a lane
// owns four consecutive columns, the four waves take every fourth row (four rows in flight per wave) and their sums are added
// in wave order.
__global__ __launch_bounds__(256) void k_segment_rowsum(const float* __restrict__ X, int D, const int64_t* __restrict__ ptr,
                                                        const int64_t* __restrict__ perm, float* __restrict__ out, int64_t S) {
    __shared__ double red[4][256];
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int d = blockIdx.y * 256 + 4 * lane;
    const int64_t b0 = ptr[s], e0 = ptr[s + 1];
    const int64_t per = (e0 - b0 + gridDim.z - 1) / gridDim.z;
    const int64_t b = b0 + per * blockIdx.z;
    const int64_t e = b + per < e0 ? b + per : e0;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (d < D) {
        int64_t p = b + grp;
        for (; p + 12 < e; p += 16) {
            f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4(X + (perm ? perm[p + 4 * u] : p + 4 * u) * D + d);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += (double)v[u][c];
        }
        for (; p < e; p += 4) {
            const f4 v = ld4(X + (perm ? perm[p] : p) * D + d);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += (double)v[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[grp][4 * lane + c] = acc[c];
    __syncthreads();
    const int dc = blockIdx.y * 256 + threadIdx.x;
    if (dc < D)
        out[((int64_t)blockIdx.z * S + s) * D + dc] =
            (float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

// The same sums with a per-row multiplier row: out2[z][s][:] = sum of X[row][:] * T[clamp(idx[row], 0, idx_hi)][:] (the f32 product,
// then f64 accumulation — what summing a stored product array gives), and with PLAIN also out[z][s][:] = sum of X[row][:] from the
// same loads.  The relation-context gradient d r_ctx [E, D] is written ONCE by k_edge_translate_bwd and its three uses
// (d bias_q = sum over the graph, d gate_q = sum over the graph of d r_ctx * rel_repr[relation], d rel_repr = sum over the
// relation of d r_ctx * gate_q[graph]) are formed here, instead of three [E, D] product arrays written and read back.
template <bool PLAIN>
__global__ __launch_bounds__(256) void k_segment_rowsum_mul(const float* __restrict__ X, int D, const int64_t* __restrict__ ptr,
                                                            const int64_t* __restrict__ perm, float* __restrict__ out,
                                                            float* __restrict__ out2, int64_t S, const float* __restrict__ T,
                                                            const int64_t* __restrict__ idx, int64_t idx_hi) {
    __shared__ double red[4][256];
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int d = blockIdx.y * 256 + 4 * lane;
    const int64_t b0 = ptr[s], e0 = ptr[s + 1];
    const int64_t per = (e0 - b0 + gridDim.z - 1) / gridDim.z;
    const int64_t b = b0 + per * blockIdx.z;
    const int64_t e = b + per < e0 ? b + per : e0;
    double acc[4] = {0.0, 0.0, 0.0, 0.0}, acc2[4] = {0.0, 0.0, 0.0, 0.0};
    if (d < D) {
        int64_t p = b + grp;
        for (; p + 12 < e; p += 16) {
            f4 v[4], m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t row = perm ? perm[p + 4 * u] : p + 4 * u;
                int64_t t = idx[row];
                t = t < 0 ? 0 : (t > idx_hi ? idx_hi : t);
                v[u] = ld4(X + row * D + d);
                m[u] = ld4(T + t * D + d);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f4 pr = v[u] * m[u];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (PLAIN) acc[c] += (double)v[u][c];
                    acc2[c] += (double)pr[c];
                }
            }
        }
        for (; p < e; p += 4) {
            const int64_t row = perm ? perm[p] : p;
            int64_t t = idx[row];
            t = t < 0 ? 0 : (t > idx_hi ? idx_hi : t);
            const f4 v = ld4(X + row * D + d);
            const f4 pr = v * ld4(T + t * D + d);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (PLAIN) acc[c] += (double)v[c];
                acc2[c] += (double)pr[c];
            }
        }
    }
    const int dc = blockIdx.y * 256 + threadIdx.x;
    const int64_t o = ((int64_t)blockIdx.z * S + s) * D + dc;
    if (PLAIN) {
#pragma unroll
        for (int c = 0; c < 4; ++c) red[grp][4 * lane + c] = acc[c];
        __syncthreads();
        if (dc < D) out[o] = (float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[grp][4 * lane + c] = acc2[c];
    __syncthreads();
    if (dc < D) out2[o] = (float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

// sum over the CSR row [eb, ee) of +-rows of a [*, W] matrix into f64 accumulators (edge order = CSR order), eight rows in flight
__device__ inline void csr_rows_accumulate(const int32_t* __restrict__ eid, int eb, int ee, const float* __restrict__ M, int W, int d,
                                           double sign, double (&acc)[4]) {
    int p = eb;
    for (; p + 8 <= ee; p += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld4(M + (int64_t)eid[p + u] * W + d);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += sign * (double)v[u][c];
    }
    for (; p < ee; ++p) {
        const f4 v = ld4(M + (int64_t)eid[p] * W + d);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += sign * (double)v[c];
    }
}

// node gradients through the CSR: dNR[v] = sum_{out edges} DH[e] + sum_{in edges} DT[e]   (v is head / tail of e),
// dHcN[v] = sum_{out} DDF[e] - sum_{in} DDF[e].  One WAVE per (node, 256-column block of dNR or dHcN) — a hub's column
// blocks run on different waves; a lane owns four consecutive columns; f64 sums in CSR row order.
__global__ __launch_bounds__(256) void k_node_gather_grad(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_eid,
                                                          const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_eid,
                                                          const float* __restrict__ DH, const float* __restrict__ DT, int D,
                                                          const float* __restrict__ DDF, int H, float* __restrict__ dNR,
                                                          float* __restrict__ dHcN, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int ud = (D + 255) / 256, uh = (H + 255) / 256;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t v = task / (ud + uh);
    const int u = (int)(task - v * (ud + uh));
    if (v >= N) return;
    const int ob = out_ptr[v], oe = out_ptr[v + 1], ib = in_ptr[v], ie = in_ptr[v + 1];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (u < ud) {
        const int d = 4 * lane + 256 * u;
        if (d >= D) return;
        csr_rows_accumulate(out_eid, ob, oe, DH, D, d, 1.0, acc);
        csr_rows_accumulate(in_eid, ib, ie, DT, D, d, 1.0, acc);
        const f4 r = {(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
        st4(dNR + v * D + d, r);
    } else {
        const int d = 4 * lane + 256 * (u - ud);
        if (d >= H) return;
        csr_rows_accumulate(out_eid, ob, oe, DDF, H, d, 1.0, acc);
        csr_rows_accumulate(in_eid, ib, ie, DDF, H, d, -1.0, acc);
        const f4 r = {(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
        st4(dHcN + v * H + d, r);
    }
}

// dpre = dY * f'(pre) written in place of dY, from the activation's OUTPUT y: tanh: 1 - y^2, sigmoid: y (1 - y).
// skip_ids != null: rows whose id is 0 get dpre = 0 (their output was overwritten by the non-text embedding).
__global__ void k_act_bwd(float* __restrict__ dY, const float* __restrict__ Y, int64_t rows, int cols, int act,
                          const int64_t* __restrict__ skip_ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols;
    const float y = Y[i];
    float g = dY[i] * (act == 1 ? 1.0f - y * y : y * (1.0f - y));
    if (skip_ids && skip_ids[r] == 0) g = 0.f;
    dY[i] = g;
}

// C [M, N] (+)= A^T B for a handful of rows (K <= 64: the question-side weight gradients sum over the B questions of a batch, the
// non-text embedding's over one row): exact f32 FMAs in row order, one thread per output element, coalesced over n — the split-K
// TN kernel and its reduction launch cost 35 us for the same 19 MFLOP.
__global__ void k_gemm_tn_small(const float* __restrict__ A, int M, const float* __restrict__ B, int N, int K, float* __restrict__ C,
                                int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[(int64_t)k * M + m], B[(int64_t)k * N + n], acc);
    float* c = C + (int64_t)m * N + n;
    *c = accumulate ? *c + acc : acc;
}

// The node side's three elementwise passes over [N, D] in one (float4 per thread): g = dNR + add (the Wc path joins the gathered
// gradients); kept[v] = g where ids[v] == 0 else 0 (the rows the non-text embedding replaced: its gradient is their column sum);
// dNR[v] = 0 where ids[v] == 0, else g * (1 - y^2) (tanh backward through entity_proj, y = node_repr).
// `kept` may be `add` itself (a thread reads its four elements before it writes them).
__global__ void k_node_grad_finish(float* __restrict__ dNR, const float* add, const float* __restrict__ Y,
                                   const int64_t* __restrict__ ids, int64_t N, int D, float* kept) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= N * D) return;
    const bool zero_id = ids[i / D] == 0;  // D % 4 == 0: the four elements share a row
    const f4 g = ld4(dNR + i) + ld4(add + i), y = ld4(Y + i), z4 = {0.f, 0.f, 0.f, 0.f};
    st4(kept + i, zero_id ? g : z4);
    f4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = g[c] * (1.0f - y[c] * y[c]);
    st4(dNR + i, zero_id ? z4 : r);
}

// column sums of X [rows, cols] into out[cols] (+)=, two ordered stages: partial per block of kColsumRows rows, then the blocks
constexpr int kColsumRows = 256;
// block = (kColsumRows rows, 256 columns): a thread sums its column over the rows eight at a time (eight loads in flight,
// added in a fixed order)
// `direct` (one row block only, gridDim.x == 1): the sum goes straight to out[c] (+= when accumulate) — no partial table, no
// second launch (the question-side bias gradients sum B = 32 rows)
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ X, int64_t rows, int cols, float* __restrict__ part,
                                                        float* __restrict__ direct = nullptr, int accumulate = 0) {
    const int64_t r0 = (int64_t)blockIdx.x * kColsumRows;
    const int64_t r1 = r0 + kColsumRows < rows ? r0 + kColsumRows : rows;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= cols) return;
    float acc = 0.f;
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = X[(r + u) * cols + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < r1; ++r) acc += X[r * cols + c];
    if (direct) direct[c] = accumulate ? direct[c] + acc : acc;  // the order of k_reduce_partials: out + partial
    else part[(int64_t)blockIdx.x * cols + c] = acc;
}

// ---- the head: state_net.4 and score_head from ysum [H] and S ---------------------------------------------
__global__ void k_head_grads(const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ score_w,
                             const float* __restrict__ ysum, const float* __restrict__ ssum, int H, float* __restrict__ dW2,
                             float* __restrict__ db2, float* __restrict__ dscore_w, float* __restrict__ dscore_b) {
    const int i = blockIdx.x;  // output row of W2
    const float S = ssum[0];
    const float wi = score_w[i];
    float dot = 0.f;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        dW2[(int64_t)i * H + j] = wi * ysum[j];
        dot = fmaf(W2[(int64_t)i * H + j], ysum[j], dot);
    }
    __shared__ float red[256];
    red[threadIdx.x] = dot;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dscore_w[i] = red[0] + b2[i] * S;
        db2[i] = wi * S;
        if (i == 0) dscore_b[0] = S;
    }
}

// ---- logits -> state rows -----------------------------------------------------------------------------
struct CombineBwdArgs {
    CombineArgs f;          // the forward's arguments for this chunk (h1c unused)
    const float* dlogits;   // [E]
    float* DZ;              // [dirs * e_count, H]
    float* DPA;             // [e_count, H]
    float* DRC;             // [e_count, H]
    float* DDF;             // [E, H] (global rows): dz_f - dz_b
    float* daux;            // [dirs * e_count, 2]: (dnav, d(-dist))
    float* part;            // [gridDim.x][5][H] column partials: d ln_w, d ln_b, ysum, d wd, d state0_b; then [gridDim.x] S partials
};

template <int C4>
__global__ __launch_bounds__(256) void k_combine_bwd(CombineBwdArgs b) {
    const CombineArgs& a = b.f;
    __shared__ float red[4][1280 + 4];
    __shared__ float cst[4][1280];  // wd, ln_w, ln_b, v: read from LDS where they are used (registers are the scarce resource here)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = a.H;
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    f4 c_lnw[C4], c_lnb[C4], c_ys[C4], c_wd[C4], c_b1[C4];
#pragma unroll
    for (int i = 0; i < C4; ++i) c_lnw[i] = c_lnb[i] = c_ys[i] = c_wd[i] = c_b1[i] = z4;
    float c_S = 0.f;
    for (int d = threadIdx.x; d < H; d += blockDim.x) {
        cst[0][d] = a.wd[d];
        cst[1][d] = a.ln_w[d];
        cst[2][d] = a.ln_b[d];
        cst[3][d] = a.v[d];
    }
    __syncthreads();
    const float* wdv_ = cst[0];
    const float* lnw_ = cst[1];
    const float* lnb_ = cst[2];
    const float* vh_ = cst[3];
    const float invH = 1.0f / (float)H;
    for (int64_t le = (int64_t)blockIdx.x * 4 + wave; le < a.e_count; le += (int64_t)gridDim.x * 4) {
        const int64_t e = a.e_begin + le;
        const float* pa = a.PA + le * H;
        const float* rc = a.RC + le * H;
        const float* hh = a.HcN + a.edge_index[e] * H;
        const float* ht = a.HcN + a.edge_index[a.E + e] * H;
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
        f4 zh[2][C4], yv[2][C4], gv[2][C4];  // normalised pre-activation, GELU of the affine LayerNorm output and GELU' per direction
        float rstd[2] = {0.f, 0.f}, lg[2] = {0.f, 0.f}, navv[2] = {0.f, 0.f}, ndv[2] = {0.f, 0.f};
        int out_row = 0;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
#pragma unroll
            for (int i = 0; i < C4; ++i) zh[dir][i] = yv[dir][i] = gv[dir][i] = z4;
            if ((dir == 0 && !a.dir_fwd) || (dir == 1 && !a.dir_bwd)) continue;
            const int64_t row = (int64_t)out_row * a.e_count + le;
            const float nav = a.aux[row * 2], negdist = a.aux[row * 2 + 1];
            navv[dir] = nav;
            ndv[dir] = negdist;
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
                    x += ld4(wdv_ + d) * negdist;
                    v[i] = x;
                    sum += hsum4(x);
                } else {
                    v[i] = z4;
                }
            }
            const float mean = wsum(sum) * invH;
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i)
                if (4 * lane + 256 * i < H) {
                    const f4 c = v[i] - mean;
                    var += hsum4(c * c);
                }
            rstd[dir] = 1.0f / sqrtf(wsum(var) * invH + kLnEps);
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                if (d < H) {
                    zh[dir][i] = (v[i] - mean) * rstd[dir];
                    const f4 av = zh[dir][i] * ld4(lnw_ + d) + ld4(lnb_ + d);
                    const f4 vv = ld4(vh_ + d);
                    const f4 dm = a.drop_thr ? dropout_mul4(a.drop_seed, (int64_t)dir * a.E + e, H, d, a.drop_thr, a.drop_scale) : f4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float y, dy;
                        gelu_erf_both(av[c], y, dy);
                        yv[dir][i][c] = dm[c] * y;   // what state_net.4 sees (dropout applied)
                        gv[dir][i][c] = dm[c] * dy;
                        dot = fmaf(vv[c], yv[dir][i][c], dot);
                    }
                }
            }
            lg[dir] = wsum(dot) + a.v[H];
            ++out_row;
        }
        const float eb = a.edge_bias ? a.edge_bias[e] : 0.f;
        const float lf = lg[0] + eb, lb = lg[1] + eb;
        float wf = 1.f, wb = 0.f, logit = lf;
        if (a.dir_fwd && a.dir_bwd) {
            const float m = fmaxf(lf, lb);
            const float ef = expf(lf - m), ebk = expf(lb - m);
            wf = ef / (ef + ebk);
            wb = ebk / (ef + ebk);
            logit = wf * lf + wb * lb;
        } else if (a.dir_bwd) {
            wf = 0.f;
            wb = 1.f;
            logit = lb;
        }
        const float gl = b.dlogits[e];
        // logit = sum_i w_i l_i with w = softmax(l):  d logit / d l_i = w_i (1 + l_i - logit)
        float dlg[2];
        dlg[0] = a.dir_fwd ? gl * wf * ((a.dir_fwd && a.dir_bwd) ? 1.0f + lf - logit : 1.0f) : 0.f;
        dlg[1] = a.dir_bwd ? gl * wb * ((a.dir_fwd && a.dir_bwd) ? 1.0f + lb - logit : 1.0f) : 0.f;
        c_S += dlg[0] + dlg[1];
        f4 dzv[2][C4];
        out_row = 0;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
#pragma unroll
            for (int i = 0; i < C4; ++i) dzv[dir][i] = z4;
            if ((dir == 0 && !a.dir_fwd) || (dir == 1 && !a.dir_bwd)) continue;
            const int64_t row = (int64_t)out_row * a.e_count + le;
            f4 g[C4];
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                g[i] = z4;
                if (d < H) {
                    const f4 vv = ld4(vh_ + d), lw = ld4(lnw_ + d);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float y = yv[dir][i][c];
                        const float da = dlg[dir] * vv[c] * gv[dir][i][c];
                        c_ys[i][c] = fmaf(dlg[dir], y, c_ys[i][c]);
                        c_lnw[i][c] = fmaf(da, zh[dir][i][c], c_lnw[i][c]);
                        c_lnb[i][c] += da;
                        g[i][c] = da * lw[c];
                    }
                    m1 += hsum4(g[i]);
                    m2 += hsum4(g[i] * zh[dir][i]);
                }
            }
            m1 = wsum(m1) * invH;
            m2 = wsum(m2) * invH;
            float dnav = 0.f, dnd = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                if (d < H) {
                    const f4 dz = rstd[dir] * (g[i] - m1 - zh[dir][i] * m2);
                    dzv[dir][i] = dz;
                    st4(b.DZ + row * H + d, dz);
                    dnav += hsum4(dz * pav[i]);
                    dnd += hsum4(dz * ld4(wdv_ + d));
                    c_wd[i] += dz * ndv[dir];
                    c_b1[i] += dz;
                }
            }
            dnav = wsum(dnav);
            dnd = wsum(dnd);
            if (lane == 0) {
                b.daux[row * 2] = dnav;
                b.daux[row * 2 + 1] = dnd;
            }
            ++out_row;
        }
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            if (d < H) {
                st4(b.DPA + le * H + d, navv[0] * dzv[0][i] + navv[1] * dzv[1][i]);
                st4(b.DRC + le * H + d, dzv[0][i] + dzv[1][i]);
                st4(b.DDF + e * H + d, dzv[0][i] - dzv[1][i]);
            }
        }
    }
    // the four waves' column sums, added in wave order, one vector at a time through a [4][H] LDS buffer; one partial row set
    // per workgroup
    float* prow = b.part + (int64_t)blockIdx.x * 5 * H;
#pragma unroll
    for (int which = 0; which < 5; ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            if (d < H) st4(&red[wave][d], which == 0 ? c_lnw[i] : which == 1 ? c_lnb[i] : which == 2 ? c_ys[i] : which == 3 ? c_wd[i] : c_b1[i]);
        }
        if (which == 0 && lane == 0) red[wave][1280] = c_S;  // wave-uniform: every lane holds the same sum
        __syncthreads();
        for (int d = threadIdx.x; d < H; d += blockDim.x) prow[which * H + d] = ((red[0][d] + red[1][d]) + red[2][d]) + red[3][d];
        if (which == 0 && threadIdx.x == 0)
            b.part[(int64_t)gridDim.x * 5 * H + blockIdx.x] = ((red[0][1280] + red[1][1280]) + red[2][1280]) + red[3][1280];
    }
}

// ---- state-row inputs -> per-edge gradients ------------------------------------------------------------------
struct EdgeBwdArgs {
    EdgeFeatArgs f;       // the forward's arguments for this chunk (P / RCX / XS / aux are read, not written)
    const float* dP;      // [e_count, D]
    const float* dRCX;    // [e_count, D]
    const float* dXS;     // [dirs * e_count, D]
    const float* daux;    // [dirs * e_count, 2]
    float* DH;            // [E, D] (global rows)
    float* DT;            // [E, D]
    float* DRR;           // [E, D]  d rel_repr row of the edge  (drc * gate_q); null with a relation table (see k_segment_rowsum_mul)
    float* DGQ;           // [E, D]  drc * rel_repr row; null with a relation table
    float* DBQ;           // [E, D]  drc
    float* DU;            // [dirs * e_count, D]  d (struct_proj.0 output)
    float* SX;            // [dirs * e_count, F]  struct_proj.0 input rows
    float* part;          // [gridDim.x * waves][4][D]: d struct_ln_w, d struct_ln_b, d struct_gate_w, d struct_proj.0.bias; then [gridDim.x * waves] d struct_gate_b
};

// Two kernels (one would need ~400 registers per lane: it spilled).
// (1) the struct MLP side: per (edge, direction) recompute struct_proj.0 -> LayerNorm -> GELU -> gate from the 2 x half
//     node_struct values, then dU (gradient of the struct_proj.0 output row), the MLP's input row SX and the column partials.
template <int C4>
__global__ __launch_bounds__(512) void k_edge_struct_bwd(EdgeBwdArgs b) {
    const EdgeFeatArgs& a = b.f;
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const float gate_b = a.struct_gate_b[0];
    const int half = F >> 1;
    const float inv_d = 1.0f / (float)D;
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    f4 c_lw[C4], c_lb[C4], c_gw[C4], c_du[C4];  // column sums: d struct_ln_w, d struct_ln_b, d struct_gate_w, d struct_proj.0.bias (= sum of dU rows)
#pragma unroll
    for (int i = 0; i < C4; ++i) c_lw[i] = c_lb[i] = c_gw[i] = c_du[i] = z4;
    float c_gb = 0.f;
    const int dirs = (a.dir_fwd ? 1 : 0) + (a.dir_bwd ? 1 : 0);

    for (int64_t le = (int64_t)blockIdx.x * waves + wave; le < a.e_count; le += (int64_t)gridDim.x * waves) {
        const int64_t e = a.e_begin + le;
        const int64_t hv = __builtin_amdgcn_readfirstlane((int)a.edge_index[e]);
        const int64_t tv = __builtin_amdgcn_readfirstlane((int)a.edge_index[a.E + e]);
        const float* nsh = a.node_struct + hv * half;
        const float* nst = a.node_struct + tv * half;
        // struct_proj.0 of BOTH directions in one pass over the weights, as the forward does it (every LDS weight row feeds two
        // accumulators): the per-direction form read the 20 rows twice, and with 12 waves per CU the LDS pipe was as busy as the VALU
        f4 s2d[2][C4];
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            s2d[0][i] = s2d[1][i] = d < D ? ld4(l_b + d) : z4;
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
                        s2d[0][i][c] = fmaf(w2[c], xt, fmaf(w1[c], xh, s2d[0][i][c]));
                        s2d[1][i][c] = fmaf(w2[c], xh, fmaf(w1[c], xt, s2d[1][i][c]));
                    }
                }
            }
        }
#pragma unroll
        for (int out_row = 0; out_row < 2; ++out_row) {
            if (out_row >= dirs) continue;
            const int dir = a.dir_fwd ? out_row : 1;
            const int64_t row = (int64_t)out_row * a.e_count + le;
            const float* na = dir == 0 ? nsh : nst;  // first half of the MLP input
            const float* nb = dir == 0 ? nst : nsh;
            f4 s2[C4];
#pragma unroll
            for (int i = 0; i < C4; ++i) s2[i] = s2d[dir][i];
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) sum += (4 * lane + 256 * i < D) ? hsum4(s2[i]) : 0.f;
            const float mean = wsum(sum) * inv_d;
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i)
                if (4 * lane + 256 * i < D) {
                    const f4 c = s2[i] - mean;
                    var += hsum4(c * c);
                }
            const float rstd = 1.0f / sqrtf(wsum(var) * inv_d + kLnEps);
            f4 uh[C4], av[C4], sv[C4];
            float gacc = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                uh[i] = av[i] = sv[i] = z4;
                if (d < D) {
                    const f4 lw = ld4(l_lw + d), lb = ld4(l_lb + d), gw = ld4(l_gw + d);
                    uh[i] = (s2[i] - mean) * rstd;
                    av[i] = uh[i] * lw + lb;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float y, dy;
                        gelu_erf_both(av[i][c], y, dy);
                        sv[i][c] = y;
                        av[i][c] = dy;  // from here on av holds GELU'
                        gacc = fmaf(gw[c], sv[i][c], gacc);
                    }
                }
            }
            const float nav = sigmoidf_(wsum(gacc) + gate_b);
            const float dpre = b.daux[row * 2] * nav * (1.0f - nav);
            c_gb += dpre;  // same value on every lane: counted once below
            f4 gg[C4];
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                gg[i] = z4;
                if (d < D) {
                    const f4 lw = ld4(l_lw + d), gw = ld4(l_gw + d);
                    const f4 ds = ld4(b.dXS + row * D + d) + dpre * gw;
                    c_gw[i] += dpre * sv[i];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float da = ds[c] * av[i][c];
                        c_lw[i][c] = fmaf(da, uh[i][c], c_lw[i][c]);
                        c_lb[i][c] += da;
                        gg[i][c] = da * lw[c];
                    }
                    m1 += hsum4(gg[i]);
                    m2 += hsum4(gg[i] * uh[i]);
                }
            }
            m1 = wsum(m1) * inv_d;
            m2 = wsum(m2) * inv_d;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int d = 4 * lane + 256 * i;
                if (d < D) {
                    const f4 du = rstd * (gg[i] - m1 - uh[i] * m2);
                    st4(b.DU + row * D + d, du);
                    c_du[i] += du;
                }
            }
            // the struct MLP's input row (wave-uniform values): [ns[a] | ns[b]] with (a, b) = (head, tail) / (tail, head)
            if (lane < F) b.SX[row * F + lane] = lane < half ? na[lane] : nb[lane - half];
        }
    }
    // column partials: one row set per WAVE ([gridDim.x * waves][4][D] then [gridDim.x * waves] scalars); the host reduces the
    // table in two ordered stages
    const int64_t prow_id = (int64_t)blockIdx.x * waves + wave;
    float* prow = b.part + prow_id * 4 * D;
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int d = 4 * lane + 256 * i;
        if (d < D) {
            st4(prow + 0 * D + d, c_lw[i]);
            st4(prow + 1 * D + d, c_lb[i]);
            st4(prow + 2 * D + d, c_gw[i]);
            st4(prow + 3 * D + d, c_du[i]);
        }
    }
    if (lane == 0) b.part[(int64_t)gridDim.x * waves * 4 * D + prow_id] = c_gb;
}

// (2) the product / translation side: p = h * rc * t (nav applied downstream), err = +-(h - t) + rc, -dist = -||err||:
//     per edge dh, dt, and d rc in its three uses (d rel_repr row, d gate_q, d bias_q contributions).  One wave per edge.
template <int C4>
__global__ __launch_bounds__(256) void k_edge_translate_bwd(EdgeBwdArgs b) {
    const EdgeFeatArgs& a = b.f;
    const int D = a.D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int dirs = (a.dir_fwd ? 1 : 0) + (a.dir_bwd ? 1 : 0);
    for (int64_t le = (int64_t)blockIdx.x * waves + wave; le < a.e_count; le += (int64_t)gridDim.x * waves) {
        const int64_t e = a.e_begin + le;
        const int64_t hv = __builtin_amdgcn_readfirstlane((int)a.edge_index[e]);
        const int64_t tv = __builtin_amdgcn_readfirstlane((int)a.edge_index[a.E + e]);
        const int64_t g = __builtin_amdgcn_readfirstlane((int)a.edge_batch[e]);
        int64_t rrow = a.rel_by_edge ? e : (int64_t)__builtin_amdgcn_readfirstlane((int)a.edge_attr[e]);
        if (!a.rel_by_edge) rrow = rrow < 0 ? 0 : (rrow >= a.R ? a.R - 1 : rrow);
        const float* hp = a.node_repr + hv * D;
        const float* tp = a.node_repr + tv * D;
        const float* rp = a.rel_repr + rrow * D;
        const float* gp = a.gate_q + g * D;
        const float* bp = a.bias_q + g * D;
        f4 h[C4], t[C4], rr[C4], gq[C4], rc[C4], dh[C4], dt[C4], drc[C4];
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            if (d < D) {
                h[i] = ld4(hp + d);
                t[i] = ld4(tp + d);
                rr[i] = ld4(rp + d);
                gq[i] = ld4(gp + d);
                rc[i] = rr[i] * gq[i] + ld4(bp + d);
                const f4 dp = ld4(b.dP + le * D + d);
                dh[i] = dp * rc[i] * t[i];
                dt[i] = dp * h[i] * rc[i];
                drc[i] = dp * h[i] * t[i] + ld4(b.dRCX + le * D + d);
            } else {
                h[i] = t[i] = rr[i] = gq[i] = rc[i] = dh[i] = dt[i] = drc[i] = z4;
            }
        }
        // err_f = h + rc - t, err_b = t + rc - h = 2 rc - err_f... kept as two explicit passes: same arithmetic as the forward
#pragma unroll
        for (int out_row = 0; out_row < 2; ++out_row) {
            if (out_row >= dirs) continue;
            const int dir = a.dir_fwd ? out_row : 1;
            const int64_t row = (int64_t)out_row * a.e_count + le;
            float dsq = 0.f;
            f4 err[C4];
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                err[i] = dir == 0 ? h[i] + rc[i] - t[i] : t[i] + rc[i] - h[i];
                if (4 * lane + 256 * i < D) dsq += hsum4(err[i] * err[i]);
            }
            const float nrm = sqrtf(wsum(dsq));
            const float dnd = b.daux[row * 2 + 1];
            // -dist = -||err||:  d err = dnd * (-err / ||err||)   (0 at err == 0, like torch.norm's subgradient)
            const float kerr = nrm > 0.f ? -dnd / nrm : 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const f4 de = kerr * err[i];
                if (dir == 0) {
                    dh[i] += de;
                    dt[i] -= de;
                } else {
                    dt[i] += de;
                    dh[i] -= de;
                }
                drc[i] += de;
            }
        }
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int d = 4 * lane + 256 * i;
            if (d < D) {
                st4(b.DH + e * D + d, dh[i]);
                st4(b.DT + e * D + d, dt[i]);
                if (b.DRR) st4(b.DRR + e * D + d, drc[i] * gq[i]);  // null: formed from DBQ by k_segment_rowsum_mul
                if (b.DGQ) st4(b.DGQ + e * D + d, drc[i] * rr[i]);
                st4(b.DBQ + e * D + d, drc[i]);
            }
        }
    }
}

}  // namespace evi
