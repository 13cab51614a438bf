// Dense query x index cosine top-k for gfx950 (MI355X).
//
// Replaces (generalised from per-group arg-max to a global top-k) the arithmetic of
//   _normalize_embeddings + index_select + torch.mv + argmax
//   reference: scripts/build_retrieval_pipeline.py:833-837, 856-874.
//
// Data path
//   The index [N, D] f32 lives in HBM and is streamed exactly once per batch of <= 32 queries.
//   Each wave owns 16 index rows at a time and loads them straight into MFMA B-fragments
//   (lane (n = l&15, g = l>>4) reads 16 B of row n at column 16 j + 4 g): no LDS round trip for
//   the stream, which no other wave reuses.  The <= 32 queries are pre-fragmented once into the
//   A-operand order and parked in LDS for the whole launch (D * 128 B: 96 KiB at D = 768).
//   v_mfma_f32_16x16x4_f32 computes the [32 queries x 16 rows] block as an exact f32 FMA chain
//   in a fixed order of d, so a row's score never depends on where the row sits.
//
// Selection (exact)
//   Rows are scanned in segments of geometrically growing length.  The first segment writes its
//   scores densely; every later segment appends only rows whose score reaches the query's current
//   k-th best (tau), through a per-query atomic cursor.  After each segment one workgroup per
//   query selects the exact top-k of its candidate list (LDS bitonic sort, with an 8-bit radix
//   select in front when the list exceeds 8192 keys) and raises tau.  The candidate buffer holds
//   k + segment_rows entries, so it cannot overflow for any data (e.g. an index sorted by score).
//
// Bound: HBM.  Algorithmic bytes per batch = N*D*4 + N*4 (row_scale, optional) + Q*D*4 + Q*k*12.
#include "common.hpp"

#include <hip/hip_ext.h>
#include <hip/hip_fp8.h>

#include <stdlib.h>

namespace evi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kQueryBlock = 32;               // queries per pass over the index
constexpr int64_t kFirstSegment = 65536;      // dense segment: every score is written, no filter yet
constexpr int kCntStride = 64;                // one append cursor per 256 B: no false sharing between queries
constexpr int64_t kSegmentGrowth = 16;
constexpr int64_t kMaxSegmentDefault = 1 << 24;  // recommended workspace: 8 B x 32 x (min(N, 2^24) + k)

// Device-side gate: a scan enqueued as the FALLBACK of a proof-carrying fast path (two-stage scan) takes a pointer to that
// path's per-call status word; while the word is 0 (proof held) every kernel of the fallback returns at once, so the
// fallback costs a few empty launches and no read-back, and a failed proof is repaired on the device before anyone can
// consume the unproven result.  gate == nullptr: an ordinary, unconditional scan.
__device__ inline bool gate_closed(const int32_t* gate) {
    return gate != nullptr && *gate == 0;  // written by an earlier kernel of the same stream
}

// every query-fragment kernel also resets the batch's selection state (thresholds and append cursors): one launch
// fewer per batch than a separate reset kernel
__device__ inline void reset_query_state(float* tau, int32_t* cnt) {
    if (blockIdx.x == 0 && threadIdx.x < kQueryBlock) {
        tau[threadIdx.x] = -INFINITY;
        cnt[threadIdx.x * kCntStride] = 0;
    }
}

// q [Q, D] -> qfrag[((j*4 + g) * (NQB*16) + i) * 4 + t] = q[i][16 j + 4 g + t], zero for i >= Q.
__global__ void k_query_fragments(const float* __restrict__ q, int Q, int D, int nq_pad,
                                  float* __restrict__ qfrag, float* __restrict__ tau,
                                  int32_t* __restrict__ cnt, const int32_t* __restrict__ gate) {
    if (gate_closed(gate)) return;
    reset_query_state(tau, cnt);
    const int total = (D / 16) * 4 * nq_pad;  // float4 slots
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gridDim.x * blockDim.x) {
        const int i = s % nq_pad;
        const int jg = s / nq_pad;  // j*4 + g
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < Q) v = *reinterpret_cast<const float4*>(q + (int64_t)i * D + jg * 4);
        reinterpret_cast<float4*>(qfrag)[s] = v;
    }
}

// f16-index variant: per K-step J of 32 and lane group g, the A fragment of query i is 8 halves
// q[i][32 J + 8 g .. + 7], stored twice: hi = f16(q) and lo = f16((q - hi) * 2048) (the scale keeps lo
// out of the f16 subnormal range; the lo accumulator is rescaled by 2^-11 at the end).
//   qfrag16[((J*4 + g)*2 + part) * nq_pad + i]  (16 bytes each), zero for i >= Q.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kLoScale = 2048.0f;
__global__ void k_query_fragments_f16(const float* __restrict__ q, int Q, int D, int nq_pad,
                                      f16x8* __restrict__ qfrag, float* __restrict__ tau,
                                      int32_t* __restrict__ cnt, const int32_t* __restrict__ gate) {
    if (gate_closed(gate)) return;
    reset_query_state(tau, cnt);
    const int total = (D / 32) * 4 * nq_pad;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gridDim.x * blockDim.x) {
        const int i = s % nq_pad;
        const int jg = s / nq_pad;  // J*4 + g
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = i < Q ? q[(int64_t)i * D + jg * 8 + e] : 0.f;
            const _Float16 h = (_Float16)v;
            hi[e] = h;
            lo[e] = (_Float16)((v - (float)h) * kLoScale);
        }
        qfrag[(jg * 2 + 0) * nq_pad + i] = hi;
        qfrag[(jg * 2 + 1) * nq_pad + i] = lo;
    }
}

// fp8-index variant: a 64-byte piece J of a row holds 64 e4m3 values; lane group g owns bytes
// 16 g .. 16 g + 15 and feeds them to two f16 MFMA K-steps (m = 0: bytes 0-7, m = 1: bytes 8-15).
//   qfrag8[(((J*2 + m)*4 + g)*2 + part) * nq_pad + i] = 8 halves of q[i][64 J + 16 g + 8 m .. + 7]
__global__ void k_query_fragments_fp8(const float* __restrict__ q, int Q, int D, int nq_pad,
                                      f16x8* __restrict__ qfrag, float* __restrict__ tau,
                                      int32_t* __restrict__ cnt, const int32_t* __restrict__ gate) {
    if (gate_closed(gate)) return;
    reset_query_state(tau, cnt);
    const int total = (D / 64) * 8 * nq_pad;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gridDim.x * blockDim.x) {
        const int i = s % nq_pad;
        const int jmg = s / nq_pad;            // (J*2 + m)*4 + g
        const int g = jmg & 3, m = (jmg >> 2) & 1, J = jmg >> 3;
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = i < Q ? q[(int64_t)i * D + 64 * J + 16 * g + 8 * m + e] : 0.f;
            const _Float16 h = (_Float16)v;
            hi[e] = h;
            lo[e] = (_Float16)((v - (float)h) * kLoScale);
        }
        qfrag[(jmg * 2 + 0) * nq_pad + i] = hi;
        qfrag[(jmg * 2 + 1) * nq_pad + i] = lo;
    }
}

// NATIVE fp8 MFMA variant (F16 == 3): the e4m3 index bytes feed v_mfma_f32_16x16x32_fp8_fp8 as they are, so the query
// has to be fp8 too.  It is written as TWO e4m3 pieces with a power-of-two scale each,
//     q ~ s1 * p1 + s2 * p2,   p1 = e4m3(q / s1),  p2 = e4m3((q - s1 p1) / s2),
// (8 significant bits: |q - s1 p1 - s2 p2| <= 2^-8 |q|_inf per element, ~20x below the index's own e4m3 rounding), one
// accumulator per piece, recombined in the epilogue.  Same MFMA count as the widening variant (2 per k-step and query
// block) and no conversion work on the stream.  One workgroup per query (max |q| needs the whole row).
//   qfrag8n[((((J*2 + m)*4 + g)*2 + piece) * nq_pad + i] = 8 e4m3 bytes of piece(q[i])[64 J + 16 g + 8 m .. + 7]
//   qscale [piece][32] f32 behind the fragments (byte offset D * 2 * nq_pad)
__device__ inline float e4m3_to_float(uint32_t b) {
    const uint32_t e = (b >> 3) & 0xF, m = b & 7;
    const float mag = e == 0 ? (float)m * 0.001953125f /* 2^-9 */ : __uint_as_float(((e + 120u) << 23) | (m << 20));
    return (b & 0x80) ? -mag : mag;
}
__device__ inline float pow2_scale_for(float amax) {  // smallest power of two s with amax / s <= 448 (1 for amax == 0)
    if (!(amax > 0.f)) return 1.0f;
    int e;
    frexpf(amax / 448.0f, &e);  // amax / 448 = f * 2^e, f in [0.5, 1)
    return ldexpf(1.0f, e);
}
__global__ __launch_bounds__(256) void k_query_fragments_fp8n(const float* __restrict__ q, int Q, int D, int nq_pad,
                                                              uint64_t* __restrict__ qfrag, float* __restrict__ tau,
                                                              int32_t* __restrict__ cnt, const int32_t* __restrict__ gate) {
    if (gate_closed(gate)) return;
    reset_query_state(tau, cnt);
    __shared__ float red[256];
    const int i = blockIdx.x, tid = threadIdx.x;  // one workgroup per query slot (grid = nq_pad)
    float* qscale = reinterpret_cast<float*>(reinterpret_cast<char*>(qfrag) + (size_t)D * 2 * nq_pad);
    auto block_max = [&](float v) -> float {
        red[tid] = v;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
            __syncthreads();
        }
        const float m = red[0];
        __syncthreads();
        return m;
    };
    const bool live = i < Q;
    float a1 = 0.f;
    for (int d = tid; d < D; d += 256) a1 = fmaxf(a1, live ? fabsf(q[(int64_t)i * D + d]) : 0.f);
    const float s1 = pow2_scale_for(block_max(a1));
    float a2 = 0.f;
    for (int d = tid; d < D; d += 256) {
        const float v = live ? q[(int64_t)i * D + d] : 0.f;
        const uint32_t b1 = (uint32_t)__hip_cvt_float_to_fp8(v / s1, __HIP_SATFINITE, __HIP_E4M3);
        a2 = fmaxf(a2, fabsf(v - s1 * e4m3_to_float(b1)));
    }
    const float s2 = pow2_scale_for(block_max(a2));
    if (tid == 0) {
        qscale[i] = s1;
        qscale[kQueryBlock + i] = s2;
    }
    // 8 bytes per (J, m, g): thread -> one such group per trip
    const int groups = (D / 64) * 8;
    for (int jmg = tid; jmg < groups; jmg += 256) {
        const int g = jmg & 3, m = (jmg >> 2) & 1, J = jmg >> 3;
        uint64_t w1 = 0, w2 = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = live ? q[(int64_t)i * D + 64 * J + 16 * g + 8 * m + e] : 0.f;
            const uint32_t b1 = (uint32_t)__hip_cvt_float_to_fp8(v / s1, __HIP_SATFINITE, __HIP_E4M3) & 0xFF;
            const float r = v - s1 * e4m3_to_float(b1);
            const uint32_t b2 = (uint32_t)__hip_cvt_float_to_fp8(r / s2, __HIP_SATFINITE, __HIP_E4M3) & 0xFF;
            w1 |= (uint64_t)b1 << (8 * e);
            w2 |= (uint64_t)b2 << (8 * e);
        }
        qfrag[((int64_t)jmg * 2 + 0) * nq_pad + i] = w1;
        qfrag[((int64_t)jmg * 2 + 1) * nq_pad + i] = w2;
    }
}

// two e4m3 bytes of w (selected by `sel`) -> two f16 whose value is the fp8 value / 256:
// f16 bits = sign << 15 | (low 7 bits) << 7 (exact for every finite e4m3 code, subnormals included)
__device__ inline uint32_t fp8x2_to_f16x2_scaled(uint32_t w, uint32_t sel) {
    const uint32_t p = __builtin_amdgcn_perm(0u, w, sel);  // bytes -> [b_hi, 0, b_lo, 0]
    return (p & 0x80008000u) | ((p >> 1) & 0x3F803F80u);
}


// One pass over rows [seg_begin, seg_end) of the shard.
//   NQB: query blocks of 16 (1 or 2).  U: float4 loads in flight per lane per prefetch group
//   (U divides D/16).  THREADS: workgroup size (one workgroup per CU: LDS holds the queries).
//   NT: stream the index with non-temporal loads.
// Appends are staged per workgroup in LDS (capq entries per query) and flushed once at the end
// with one global atomic per query, so the per-query cursors see ~256 atomics per launch instead
// of one per candidate; entries that do not fit go straight to the global list.
//   F16: the index is stored as f16 (half the HBM bytes); queries are split hi + lo in f16 and
//   multiplied on v_mfma_f32_16x16x32_f16 with f32 accumulation.
//   (F16 == 2: the index is OCP e4m3 with a per-row scale; bytes are widened to f16 in registers.)
template <int NQB, int U, int THREADS, int NT, int F16>
__global__ __launch_bounds__(THREADS) void k_cosine_score(
    const float* __restrict__ qfrag, const void* __restrict__ idx, int64_t seg_begin,
    int64_t seg_end, int D, int Q, const float* __restrict__ row_scale,
    const float* __restrict__ tau, float* __restrict__ cand_score, int32_t* __restrict__ cand_id,
    int32_t* __restrict__ cand_cnt, int64_t cap, int dense, int capq, const int32_t* __restrict__ gate) {
    if (gate_closed(gate)) return;
    extern __shared__ float4 lds_q[];  // [(D/16)*4][NQB*16] float4, then the append staging area
    constexpr int NQ = NQB * 16;
    constexpr int WAVES = THREADS / 64;
    const int tid = threadIdx.x;
    const int chunks = F16 >= 2 ? D / 64 : (F16 ? D / 32 : D / 16);  // 64-byte pieces of a row
    const int qslots = F16 == 3 ? (D / 8) * NQ  // native fp8: two 1-byte pieces per element, D * NQ * 2 bytes
                                : (D / 16) * 4 * NQ;  // same LDS footprint for the other layouts: D * NQ * 4 bytes
    int* lds_cnt = reinterpret_cast<int*>(lds_q + qslots);          // [32]
    float* lds_sc = reinterpret_cast<float*>(lds_cnt + kQueryBlock);  // [32][capq]
    int* lds_id = reinterpret_cast<int*>(lds_sc + kQueryBlock * capq);  // [32][capq]
    {
        const float4* src = reinterpret_cast<const float4*>(qfrag);
        for (int s = tid; s < qslots; s += THREADS) lds_q[s] = src[s];
        if (tid < kQueryBlock) lds_cnt[tid] = 0;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    float tq[NQB][4];
#pragma unroll
    for (int b = 0; b < NQB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = b * 16 + 4 * g + r;
            tq[b][r] = (dense || qi >= Q) ? -INFINITY : tau[qi];
        }

    float qs1[NQB][4], qs2[NQB][4];  // native fp8: the two piece scales of this lane's queries
    if (F16 == 3) {
        const float* qscale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(qfrag) + (size_t)D * 2 * NQ);
#pragma unroll
        for (int b = 0; b < NQB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = b * 16 + 4 * g + r;
                qs1[b][r] = qscale[qi];
                qs2[b][r] = qscale[kQueryBlock + qi];
            }
    }
    const int64_t seg_rows = seg_end - seg_begin;
    const int64_t tiles = (seg_rows + 15) / 16;
    const int groups = chunks / U;
    const float4* lq = lds_q + g * NQ + n;  // + (j*4)*NQ + b*16

    auto ldx = [](const f32x4* p) -> f32x4 {
        if (NT) return __builtin_nontemporal_load(p);
        return *p;
    };

    // The wave's work is ONE stream of load groups (U float4 per lane each) that runs across tile
    // boundaries: while group i is multiplied, group i+1 — possibly the first group of the wave's
    // next tile — is already in flight, in a second register set (ping-pong, no copies).  Every
    // iteration therefore has the same number of loads outstanding, so the compiler's counted
    // vmcnt waits release exactly the group that is needed and never drain the prefetch.
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    const int64_t row_bytes = (int64_t)D * (F16 >= 2 ? 1 : (F16 ? 2 : 4));
    auto tile_ptr = [&](int64_t t) -> const f32x4* {
        const int64_t r = seg_begin + t * 16 + n;
        const int64_t rc = r < seg_end ? r : seg_end - 1;
        return reinterpret_cast<const f32x4*>(static_cast<const char*>(idx) + rc * row_bytes) + g;
    };
    auto load_group = [&](f32x4 (&x)[U], const f32x4* xp, int gi) {
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ldx(xp + (gi * U + u) * 4);
    };
    f32x4 acc[NQB], acc_lo[NQB];
    auto mul_group = [&](const f32x4 (&x)[U], int gi) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = gi * U + u;
            if (F16 == 3) {  // native fp8 MFMA: index bytes as they are, two e4m3 query pieces, one accumulator each
                typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));
                const u32x4n w = __builtin_bit_cast(u32x4n, x[u]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const long xb = (long)(((uint64_t)w[2 * m + 1] << 32) | (uint64_t)w[2 * m]);
                    const long* lq8 = reinterpret_cast<const long*>(lds_q) + ((((j * 2 + m) * 4 + g) * 2) * NQ) + n;
#pragma unroll
                    for (int b = 0; b < NQB; ++b) {
                        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lq8[b * 16], xb, acc[b], 0, 0, 0);
                        acc_lo[b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lq8[NQ + b * 16], xb, acc_lo[b], 0, 0, 0);
                    }
                }
                continue;
            }
            if (F16 == 2) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 w = __builtin_bit_cast(u32x4, x[u]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    u32x4 h;
                    h[0] = fp8x2_to_f16x2_scaled(w[2 * m], 0x010C000Cu);
                    h[1] = fp8x2_to_f16x2_scaled(w[2 * m], 0x030C020Cu);
                    h[2] = fp8x2_to_f16x2_scaled(w[2 * m + 1], 0x010C000Cu);
                    h[3] = fp8x2_to_f16x2_scaled(w[2 * m + 1], 0x030C020Cu);
                    const f16x8 xb = __builtin_bit_cast(f16x8, h);
                    const f16x8* lq16 = reinterpret_cast<const f16x8*>(lds_q) + ((((j * 2 + m) * 4 + g) * 2) * NQ) + n;
#pragma unroll
                    for (int b = 0; b < NQB; ++b) {
                        const f16x8 ah = lq16[b * 16], al = lq16[NQ + b * 16];
                        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xb, acc[b], 0, 0, 0);
                        acc_lo[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xb, acc_lo[b], 0, 0, 0);
                    }
                }
                continue;
            }
            if (F16) {
                const f16x8 xb = __builtin_bit_cast(f16x8, x[u]);
                const f16x8* lq16 = reinterpret_cast<const f16x8*>(lds_q) + ((j * 4 + g) * 2) * NQ + n;
#pragma unroll
                for (int b = 0; b < NQB; ++b) {
                    const f16x8 ah = lq16[b * 16], al = lq16[NQ + b * 16];
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xb, acc[b], 0, 0, 0);
                    acc_lo[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xb, acc_lo[b], 0, 0, 0);
                }
                continue;
            }
            float4 a[NQB];
#pragma unroll
            for (int b = 0; b < NQB; ++b) a[b] = lq[(j * 4) * NQ + b * 16];
#pragma unroll
            for (int b = 0; b < NQB; ++b) {
                const float ab[4] = {a[b].x, a[b].y, a[b].z, a[b].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[e], x[u][e], acc[b], 0, 0, 0);
            }
        }
    };

    int64_t t = (int64_t)blockIdx.x * WAVES + wave;
    const f32x4* xp = tile_ptr(t < tiles ? t : 0);
    f32x4 xa[U], xb[U];
    if (t < tiles) load_group(xa, xp, 0);
    for (; t < tiles; t += stride) {
        const int64_t tn = t + stride;
        const f32x4* xpn = tile_ptr(tn < tiles ? tn : t);  // past the end: a harmless re-read
        const int64_t row = seg_begin + t * 16 + n;
#pragma unroll
        for (int b = 0; b < NQB; ++b) acc[b] = acc_lo[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((groups & 1) == 0) {
            for (int gi = 0; gi < groups; gi += 2) {
                load_group(xb, xp, gi + 1);
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs it overlaps
                mul_group(xa, gi);
                __builtin_amdgcn_sched_barrier(0);
                if (gi + 2 < groups) load_group(xa, xp, gi + 2); else load_group(xa, xpn, 0);
                __builtin_amdgcn_sched_barrier(0);
                mul_group(xb, gi + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {  // odd group count: same stream, with a register copy per group
            for (int gi = 0; gi < groups; ++gi) {
                if (gi + 1 < groups) load_group(xb, xp, gi + 1); else load_group(xb, xpn, 0);
                mul_group(xa, gi);
#pragma unroll
                for (int u = 0; u < U; ++u) xa[u] = xb[u];
            }
        }
        xp = xpn;

        if (row < seg_end) {
            const float scale = row_scale ? row_scale[row] : 1.0f;
#pragma unroll
            for (int b = 0; b < NQB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = b * 16 + 4 * g + r;
                    if (qi >= Q) continue;
                    float dot = F16 == 3 ? fmaf(acc_lo[b][r], qs2[b][r], acc[b][r] * qs1[b][r])
                                : (F16 ? fmaf(acc_lo[b][r], 1.0f / kLoScale, acc[b][r]) : acc[b][r]);
                    if (F16 == 2) dot *= 256.0f;  // the widened bytes carry value / 256
                    const float s = row_scale ? dot * scale : dot;
                    if (dense) {
                        const int64_t pos = row - seg_begin;
                        cand_score[qi * cap + pos] = s;  // the id of a dense entry is its position (seg_begin == 0): not stored
                    } else if (s >= tq[b][r] || s != s) {
                        const int p = atomicAdd(&lds_cnt[qi], 1);
                        if (p < capq) {
                            lds_sc[qi * capq + p] = s;
                            lds_id[qi * capq + p] = (int32_t)row;
                        } else {
                            const int32_t pos = atomicAdd(&cand_cnt[qi * kCntStride], 1);
                            cand_score[qi * cap + pos] = s;
                            cand_id[qi * cap + pos] = (int32_t)row;
                        }
                    }
                }
        }
    }

    if (!dense && capq > 0) {
        __syncthreads();
        for (int qi = wave; qi < Q; qi += WAVES) {
            int cnt = lds_cnt[qi];
            cnt = cnt < capq ? cnt : capq;
            if (cnt <= 0) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&cand_cnt[qi * kCntStride], cnt);
            base = __shfl(base, 0, 64);
            for (int e = lane; e < cnt; e += 64) {
                cand_score[qi * cap + base + e] = lds_sc[qi * capq + e];
                cand_id[qi * cap + base + e] = lds_id[qi * capq + e];
            }
        }
    }
}

// One workgroup per query: exact top-k of its candidate list, written back sorted to the head of
// the list; raises tau to the k-th score.  The last call of a batch also writes the outputs.
__global__ __launch_bounds__(kSelectThreads) void k_candidates_select(
    float* __restrict__ cand_score, int32_t* __restrict__ cand_id, int32_t* __restrict__ cand_cnt,
    float* __restrict__ tau, int64_t cap, int k, int64_t dense_count, int final_pass,
    int64_t row_id_base, float* __restrict__ out_score, int64_t* __restrict__ out_index,
    const int32_t* __restrict__ gate) {
    if (gate_closed(gate)) return;
    __shared__ SelectShared sh;
    const int qi = blockIdx.x;
    const float* cs = cand_score + qi * cap;
    const int32_t* ci = cand_id + qi * cap;
    const int64_t cnt = dense_count >= 0 ? dense_count : (int64_t)cand_cnt[qi * kCntStride];
    // A workgroup pulls its list through ONE CU's memory pipeline (~30 GB/s from HBM), several passes over it: the
    // 65 536 (score, id) pairs of a dense first segment are 1.5 - 2 MB of reads, most of that selection's 75 us.  In the
    // dense segment entry i IS row i of the shard (the segment starts at row 0), so its ids are never read.
    const bool dense = dense_count >= 0;
    auto load = [&](int64_t i) -> uint64_t { return make_key(cs[i], dense ? (uint32_t)i : (uint32_t)ci[i]); };
    const int m = block_topk(sh, load, cnt, k);
    // block_topk ends on a barrier: every read of the old list is done before it is overwritten.
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const uint64_t key = sh.keys[i];
        cand_score[qi * cap + i] = key_score(key);
        cand_id[qi * cap + i] = (int32_t)key_index(key);
    }
    if (threadIdx.x == 0) {
        cand_cnt[qi * kCntStride] = m;
        if (m == k) tau[qi] = key_score(sh.keys[k - 1]);
    }
    if (final_pass) {
        for (int i = threadIdx.x; i < k; i += blockDim.x) {
            if (i < m) {
                const uint64_t key = sh.keys[i];
                out_score[(int64_t)qi * k + i] = key_score(key);
                out_index[(int64_t)qi * k + i] = row_id_base + (int64_t)key_index(key);
            } else {
                out_score[(int64_t)qi * k + i] = -INFINITY;
                out_index[(int64_t)qi * k + i] = -1;
            }
        }
    }
}

struct WsLayout {
    size_t qfrag_off, tau_off, cnt_off, score_off, id_off, total;
    int64_t cap;
};

static WsLayout ws_layout(int D, int k, int64_t seg_max) {
    WsLayout w;
    size_t off = 0;
    w.qfrag_off = off;
    off = align_up(off + (size_t)D * kQueryBlock * sizeof(float), 256);
    w.tau_off = off;
    off = align_up(off + kQueryBlock * sizeof(float), 256);
    w.cnt_off = off;
    off = align_up(off + (size_t)kQueryBlock * kCntStride * sizeof(int32_t), 256);
    w.cap = seg_max + k;
    w.score_off = off;
    off = align_up(off + (size_t)kQueryBlock * w.cap * sizeof(float), 256);
    w.id_off = off;
    off = align_up(off + (size_t)kQueryBlock * w.cap * sizeof(int32_t), 256);
    w.total = off;
    return w;
}

static int64_t clamp_seg(int64_t N, int64_t want) {
    int64_t s = N < want ? N : want;
    return s < 1 ? 1 : s;
}

struct ScanVariant {
    int threads, nt, u_cap;
};

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Tunables (defaults are the measured best on MI355X; the env overrides exist for
// tools/scan_variants.py, which A/Bs them in one process).
static ScanVariant scan_variant() {
    ScanVariant v;
    v.threads = env_int("EVI_SCAN_THREADS", 1024);
    v.nt = env_int("EVI_SCAN_NT", 0);
    v.u_cap = env_int("EVI_SCAN_U", 4);
    return v;
}

struct ScoreArgs {
    int grid;
    size_t lds;
    hipStream_t st;
    const float* qfrag;
    const void* idx;
    int64_t b, e;
    int D, Q;
    const float* row_scale;
    const float* tau;
    float* cs;
    int32_t* ci;
    int32_t* cc;
    int64_t cap;
    int dense, capq;
    const int32_t* gate;
    hipEvent_t ev_start, ev_stop;  // non-null: stamped with the dispatch's begin / end (bench roofline leg)
};

template <int NQB, int U, int THREADS, int NT, int F16 = 0>
static int launch_score(const ScoreArgs& a) {
    static thread_local bool attr_set = false;
    if (!attr_set) {
        EVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cosine_score<NQB, U, THREADS, NT, F16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (a.ev_start)
        hipExtLaunchKernelGGL((k_cosine_score<NQB, U, THREADS, NT, F16>), dim3(a.grid), dim3(THREADS), (uint32_t)a.lds, a.st,
                              a.ev_start, a.ev_stop, 0, a.qfrag, a.idx, a.b, a.e, a.D, a.Q, a.row_scale, a.tau, a.cs, a.ci,
                              a.cc, a.cap, a.dense, a.capq, a.gate);
    else
        hipLaunchKernelGGL((k_cosine_score<NQB, U, THREADS, NT, F16>), dim3(a.grid), dim3(THREADS), a.lds, a.st,
                           a.qfrag, a.idx, a.b, a.e, a.D, a.Q, a.row_scale, a.tau, a.cs, a.ci, a.cc, a.cap,
                           a.dense, a.capq, a.gate);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

template <int NQB, int THREADS, int NT>
static int launch_score_u(int U, const ScoreArgs& a) {
    switch (U) {
        case 8: return launch_score<NQB, 8, THREADS, NT>(a);
        case 4: return launch_score<NQB, 4, THREADS, NT>(a);
        case 2: return launch_score<NQB, 2, THREADS, NT>(a);
        default: return launch_score<NQB, 1, THREADS, NT>(a);
    }
}

// f16 / fp8 index: 1024-thread workgroups, plain loads (the tuned f32 configuration)
template <int NQB, int KIND>
static int launch_score_lowp(int U, const ScoreArgs& a) {
    switch (U) {
        case 8: return launch_score<NQB, 8, 1024, 0, KIND>(a);
        case 4: return launch_score<NQB, 4, 1024, 0, KIND>(a);
        case 2: return launch_score<NQB, 2, 1024, 0, KIND>(a);
        default: return launch_score<NQB, 1, 1024, 0, KIND>(a);
    }
}

template <int NQB>
static int launch_score_v(const ScanVariant& v, int U, const ScoreArgs& a) {
    if (v.threads == 1024) return v.nt ? launch_score_u<NQB, 1024, 1>(U, a) : launch_score_u<NQB, 1024, 0>(U, a);
    return v.nt ? launch_score_u<NQB, 512, 1>(U, a) : launch_score_u<NQB, 512, 0>(U, a);
}

// Segment schedule knobs (tuning only; results do not depend on them):
//   EVI_SCAN_FIRST   rows of the dense first segment, [4096, 65536] (default 65536)
//   EVI_SCAN_GROWTH  next segment = growth x rows scanned so far, [2, 256] (default 16)
struct SegmentSchedule {
    int64_t first, growth;
};
static SegmentSchedule segment_schedule() {
    static SegmentSchedule s{0, 0};
    if (s.first == 0) {
        int64_t first = kFirstSegment, growth = kSegmentGrowth;
        if (const char* e = getenv("EVI_SCAN_FIRST")) first = atoll(e);
        if (const char* e = getenv("EVI_SCAN_GROWTH")) growth = atoll(e);
        first = first < 4096 ? 4096 : (first > kFirstSegment ? kFirstSegment : first);
        first = (first + 15) / 16 * 16;
        growth = growth < 2 ? 2 : (growth > 256 ? 256 : growth);
        s = SegmentSchedule{first, growth};
    }
    return s;
}

static int device_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace evi

using namespace evi;

extern "C" size_t evi_cosine_topk_workspace_bytes(int Q, int64_t N, int D, int k) {
    (void)Q;
    if (D <= 0 || k <= 0 || N < 0) return 0;
    return ws_layout(D, k, clamp_seg(N, kMaxSegmentDefault)).total;
}

extern "C" size_t evi_cosine_topk_min_workspace_bytes(int Q, int64_t N, int D, int k) {
    (void)Q;
    if (D <= 0 || k <= 0 || N < 0) return 0;
    return ws_layout(D, k, clamp_seg(N, kFirstSegment)).total;
}

static int cosine_topk_impl(const float* q, int Q, const void* idx, int f16, int64_t N, int D,
                            const float* row_scale, int k, int64_t row_id_base, float* out_score,
                            int64_t* out_index, void* workspace, size_t workspace_bytes, void* stream,
                            const int32_t* gate = nullptr) {
    EVI_REQUIRE(Q >= 1, "evi_cosine_topk: Q must be >= 1, got %d", Q);
    EVI_REQUIRE(N >= 0, "evi_cosine_topk: N must be >= 0, got %lld", (long long)N);
    EVI_REQUIRE(N < (int64_t)0x7FFFFFFF, "evi_cosine_topk: a shard holds at most 2^31-1 rows, got %lld",
                (long long)N);
    EVI_REQUIRE(k >= 1 && k <= EVI_TOPK_MAX_K, "evi_cosine_topk: k must be in [1, %d], got %d",
                EVI_TOPK_MAX_K, k);
    EVI_REQUIRE(q && out_score && out_index, "evi_cosine_topk: null q/out pointer");
    EVI_REQUIRE(N == 0 || idx, "evi_cosine_topk: null idx with N > 0");
    if (D < 16 || D % 16 != 0 || D > 1280)
        return fail(EVI_ERR_UNSUPPORTED,
                    "evi_cosine_topk: D must be a multiple of 16 in [16, 1280], got %d", D);
    if (f16 == 1 && D % 32 != 0)
        return fail(EVI_ERR_UNSUPPORTED, "evi_cosine_topk_f16: D must be a multiple of 32, got %d", D);
    if (f16 >= 2 && D % 64 != 0)
        return fail(EVI_ERR_UNSUPPORTED, "evi_cosine_topk_fp8: D must be a multiple of 64, got %d", D);
    EVI_REQUIRE(f16 < 2 || row_scale || N == 0, "evi_cosine_topk_fp8: row_scale (the per-row dequantisation scale) is required");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    const size_t min_ws = evi_cosine_topk_min_workspace_bytes(Q, N, D, k);
    EVI_REQUIRE(workspace != nullptr, "evi_cosine_topk: null workspace");
    if (workspace_bytes < min_ws)
        return fail(EVI_ERR_NOMEM, "evi_cosine_topk: workspace %zu B < minimum %zu B", workspace_bytes,
                    min_ws);
    // Largest segment the workspace can hold (halve until it fits).
    int64_t seg_max = clamp_seg(N, kMaxSegmentDefault);
    while (seg_max > kFirstSegment && ws_layout(D, k, seg_max).total > workspace_bytes) seg_max >>= 1;
    if (seg_max < clamp_seg(N, kFirstSegment)) seg_max = clamp_seg(N, kFirstSegment);
    const WsLayout w = ws_layout(D, k, seg_max);
    char* base = static_cast<char*>(workspace);
    float* qfrag = reinterpret_cast<float*>(base + w.qfrag_off);
    float* tau = reinterpret_cast<float*>(base + w.tau_off);
    int32_t* cnt = reinterpret_cast<int32_t*>(base + w.cnt_off);
    float* cs = reinterpret_cast<float*>(base + w.score_off);
    int32_t* ci = reinterpret_cast<int32_t*>(base + w.id_off);

    const int chunks = f16 >= 2 ? D / 64 : (f16 ? D / 32 : D / 16);
    ScanVariant variant = scan_variant();
    if (f16) variant.threads = 1024;
    // loads in flight per lane per group: the largest U <= cap that leaves an EVEN number of groups
    // (copy-free ping-pong), else the largest U that divides the row.
    int U = 1;
    for (int cand = 8; cand >= 1; cand >>= 1)
        if (cand <= variant.u_cap && chunks % (2 * cand) == 0) { U = cand; break; }
    if (U == 1 && chunks % 2 != 0)
        for (int cand = 8; cand >= 1; cand >>= 1)
            if (cand <= variant.u_cap && chunks % cand == 0) { U = cand; break; }
    const int cus = device_cu_count();
    const int waves_per_block = variant.threads == 1024 ? 16 : 8;

    // 32 queries per pass when their fragments (+ the append counters) fit the 160 KiB of LDS, else 16: at D = 1280 the
    // f32 / f16 / widened-fp8 fragments of 32 queries are exactly 160 KiB and leave no room for the counters
    const size_t frag_bytes_32 = f16 == 3 ? (size_t)D * 2 * kQueryBlock : (size_t)D * kQueryBlock * sizeof(float);
    const int qblock = frag_bytes_32 + kQueryBlock * sizeof(int) > 160 * 1024 ? 16 : kQueryBlock;
    for (int q0 = 0; q0 < Q; q0 += qblock) {
        const int qn = (Q - q0) < qblock ? (Q - q0) : qblock;
        const int nqb = qn <= 16 ? 1 : 2;
        const int nq_pad = nqb * 16;
        float* o_score = out_score + (int64_t)q0 * k;
        int64_t* o_index = out_index + (int64_t)q0 * k;

        if (f16 == 3)
            hipLaunchKernelGGL(k_query_fragments_fp8n, dim3(nq_pad), dim3(256), 0, st, q + (int64_t)q0 * D, qn, D, nq_pad,
                               reinterpret_cast<uint64_t*>(qfrag), tau, cnt, gate);
        else if (f16 == 2)
            hipLaunchKernelGGL(k_query_fragments_fp8, dim3(32), dim3(256), 0, st, q + (int64_t)q0 * D, qn, D, nq_pad,
                               reinterpret_cast<f16x8*>(qfrag), tau, cnt, gate);
        else if (f16)
            hipLaunchKernelGGL(k_query_fragments_f16, dim3(32), dim3(256), 0, st, q + (int64_t)q0 * D, qn, D, nq_pad,
                               reinterpret_cast<f16x8*>(qfrag), tau, cnt, gate);
        else
            hipLaunchKernelGGL(k_query_fragments, dim3(32), dim3(256), 0, st, q + (int64_t)q0 * D, qn, D,
                               nq_pad, qfrag, tau, cnt, gate);
        EVI_LAUNCH_CHECK();

        if (N == 0) {
            hipLaunchKernelGGL(k_candidates_select, dim3(qn), dim3(kSelectThreads), 0, st, cs, ci, cnt,
                               tau, w.cap, k, (int64_t)0, 1, row_id_base, o_score, o_index, gate);
            EVI_LAUNCH_CHECK();
            continue;
        }
        const size_t lds_q_bytes = f16 == 3 ? (size_t)D * 2 * nq_pad : (size_t)(D / 16) * 4 * nq_pad * sizeof(float4);
        // staging entries per query: whatever LDS is left, at most 64
        const int64_t lds_left = (int64_t)160 * 1024 - (int64_t)lds_q_bytes - (int64_t)(kQueryBlock * sizeof(int));
        int capq = lds_left > 0 ? (int)(lds_left / (kQueryBlock * 8)) : 0;
        capq = capq > 64 ? 64 : capq;
        const size_t lds = lds_q_bytes + kQueryBlock * sizeof(int) + (size_t)kQueryBlock * capq * 8;
        int64_t begin = 0;
        const SegmentSchedule sched = segment_schedule();
        int64_t seg = clamp_seg(N, sched.first);
        bool first = true;
        while (begin < N) {
            int64_t end = begin + seg;
            if (end > N) end = N;
            const int64_t tiles = (end - begin + 15) / 16;
            int64_t want = (tiles + waves_per_block - 1) / waves_per_block;
            const int grid = (int)(want < cus ? want : cus);
            ScoreArgs sa{grid, lds, st, qfrag, idx, begin, end, D, qn, row_scale, tau,
                         cs, ci, cnt, w.cap, first ? 1 : 0, capq, gate, nullptr, nullptr};
            if (!gate) timing_kernel_events(kTimeCosineScore, &sa.ev_start, &sa.ev_stop);  // gated launches stay out of the accounts
            const int rc = f16 == 3 ? (nqb == 1 ? launch_score_lowp<1, 3>(U, sa) : launch_score_lowp<2, 3>(U, sa))
                           : f16 == 2 ? (nqb == 1 ? launch_score_lowp<1, 2>(U, sa) : launch_score_lowp<2, 2>(U, sa))
                           : f16  ? (nqb == 1 ? launch_score_lowp<1, 1>(U, sa) : launch_score_lowp<2, 1>(U, sa))
                                  : (nqb == 1 ? launch_score_v<1>(variant, U, sa) : launch_score_v<2>(variant, U, sa));
            if (rc != EVI_OK) return rc;
            const int final_pass = end >= N ? 1 : 0;
            hipEvent_t sel_start, sel_stop;
            if (!gate && timing_kernel_events(kTimeSelect, &sel_start, &sel_stop))
                hipExtLaunchKernelGGL(k_candidates_select, dim3(qn), dim3(kSelectThreads), 0, st, sel_start, sel_stop, 0, cs, ci,
                                      cnt, tau, w.cap, k, first ? (end - begin) : (int64_t)-1, final_pass, row_id_base, o_score,
                                      o_index, gate);
            else
                hipLaunchKernelGGL(k_candidates_select, dim3(qn), dim3(kSelectThreads), 0, st, cs, ci, cnt,
                                   tau, w.cap, k, first ? (end - begin) : (int64_t)-1, final_pass,
                                   row_id_base, o_score, o_index, gate);
            EVI_LAUNCH_CHECK();
            begin = end;
            first = false;
            int64_t next = begin * sched.growth;
            seg = next < seg_max ? next : seg_max;
            if (seg < 1) seg = 1;
        }
    }
    return EVI_OK;
}

extern "C" int evi_cosine_topk(const float* q, int Q, const float* idx, int64_t N, int D,
                               const float* row_scale, int k, int64_t row_id_base, float* out_score,
                               int64_t* out_index, void* workspace, size_t workspace_bytes,
                               void* stream) {
    return cosine_topk_impl(q, Q, idx, 0, N, D, row_scale, k, row_id_base, out_score, out_index, workspace,
                            workspace_bytes, stream);
}

extern "C" int evi_cosine_topk_f16(const float* q, int Q, const void* idx_f16, int64_t N, int D,
                                   const float* row_scale, int k, int64_t row_id_base, float* out_score,
                                   int64_t* out_index, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    return cosine_topk_impl(q, Q, idx_f16, 1, N, D, row_scale, k, row_id_base, out_score, out_index, workspace,
                            workspace_bytes, stream);
}

extern "C" int evi_cosine_topk_fp8(const float* q, int Q, const void* idx_fp8, int64_t N, int D,
                                   const float* row_scale, int k, int64_t row_id_base, float* out_score,
                                   int64_t* out_index, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    return cosine_topk_impl(q, Q, idx_fp8, 2, N, D, row_scale, k, row_id_base, out_score, out_index, workspace,
                            workspace_bytes, stream);
}

extern "C" int evi_cosine_topk_fp8_mfma(const float* q, int Q, const void* idx_fp8, int64_t N, int D,
                                        const float* row_scale, int k, int64_t row_id_base, float* out_score,
                                        int64_t* out_index, void* workspace, size_t workspace_bytes, void* stream) {
    return cosine_topk_impl(q, Q, idx_fp8, 3, N, D, row_scale, k, row_id_base, out_score, out_index, workspace,
                            workspace_bytes, stream);
}

// =====================================================================================================
// Many queries (Q in the hundreds): GEMM-shaped, still exact.
//
// The scan above holds 32 queries in LDS and streams the index once per 32 queries: at Q = 512 that is
// 16 passes, bound by the f32 MFMA rate.  Here the scores of ALL queries against a slab of index rows
// are formed as one split-bf16 GEMM (bf16 MFMA, f32 accumulate: |error| <= 2.5e-4 * |q| |x|, see
// kApproxEps) — approximate, so they only SELECT: every query keeps its kk = k + reserve best rows by
// approximate score, and the call succeeds only if the k-th and the kk-th approximate scores are more
// than 2 eps apart, which proves that the true top-k (by the scan's own f32 arithmetic) is inside the
// kk kept rows.  Those kk rows are then re-scored with exactly the scan's instruction sequence
// (v_mfma_f32_16x16x4_f32 over d in the same order), and the final top-k is taken on those scores: ids
// and scores are bit-identical to evi_cosine_topk.  When the gap test fails (heavy ties, adversarial
// data) or a candidate list overflows, *status is set and the caller runs the scan instead.
// =====================================================================================================
namespace evi {

constexpr float kApproxEps = 2.5e-4f;        // |gemm score - scan score| <= kApproxEps * |q| for unit-norm rows:
                                             // bf16 hi/lo split 2^-16 + 2^-18, 2304 f32 accumulations x 2^-24,
                                             // plus the scan's own 768 x 2^-24
constexpr float kApproxEps1 = 4.2e-3f;       // single-product selection: both operands rounded to bf16 (2 x 2^-9 + 2^-18
                                             // relative per product), 768 accumulations, plus the scan's own error
constexpr int kGemmTopkCap = 16384;          // candidate slots per query between two selections
constexpr int64_t kGemmFirstRows = 4096;     // first slab: every row is a candidate (4096 <= cap)
constexpr int64_t kGemmSlabRows = 1 << 21;   // rows per GEMM launch between two selections (bounds the appends)
constexpr int kGemmGrowth = 8;

// candidates kept per query: the coarser the selection scores, the more rows can hide inside 2 eps of the k-th
static int gemm_topk_reserve(int k, int single = 0) {
    if (single) return k + (k > 1024 ? k : 1024);
    int r = k / 2 > 256 ? k / 2 : 256;
    return k + r;
}

__global__ void k_gt_init(float* tau, int32_t* cnt, int Q, int32_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q) {
        tau[i] = -INFINITY;
        cnt[i * kCntStride] = 0;
    }
    if (i == 0) *status = 0;
}

// scores [rows, Q] (row-major) of slab rows [row0, row0 + rows): append (score * row_scale, row) to the list of
// every query whose current threshold it reaches
__global__ __launch_bounds__(256) void k_gt_filter(const float* __restrict__ scores, int64_t rows, int Q, int64_t row0,
                                                   const float* __restrict__ row_scale, const float* __restrict__ tau,
                                                   float* __restrict__ cand_score, int32_t* __restrict__ cand_id,
                                                   int32_t* __restrict__ cand_cnt, int32_t* __restrict__ status) {
    const int64_t total = rows * Q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Q;
        const int qi = (int)(i - r * Q);
        float s = scores[i];
        if (row_scale) s *= row_scale[row0 + r];
        if (s >= tau[qi] || s != s) {
            const int32_t pos = atomicAdd(&cand_cnt[qi * kCntStride], 1);
            if (pos < kGemmTopkCap) {
                cand_score[(int64_t)qi * kGemmTopkCap + pos] = s;
                cand_id[(int64_t)qi * kGemmTopkCap + pos] = (int32_t)(row0 + r);
            } else {
                atomicOr(status, 2);  // list full: the caller falls back to the scan
            }
        }
    }
}

// after an overflowing slab the cursor may exceed the capacity: clamp before the selection reads the list
__global__ void k_gt_clamp(int32_t* cnt, int Q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q && cnt[i * kCntStride] > kGemmTopkCap) cnt[i * kCntStride] = kGemmTopkCap;
}

// gap test: with kk rows kept, the true top-k is among them iff approx[k-1] - approx[kk-1] > 2 eps |q|
__global__ void k_gt_gap(const float* __restrict__ q, int Q, int D, const float* __restrict__ cand_score,
                         const int32_t* __restrict__ cand_cnt, int k, int kk, float eps, int32_t* __restrict__ status) {
    const int qi = blockIdx.x;
    const int lane = threadIdx.x;  // one wave
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) ss = fmaf(q[(int64_t)qi * D + d], q[(int64_t)qi * D + d], ss);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane == 0) {
        const int m = cand_cnt[qi * kCntStride];
        if (m >= kk) {
            const float* s = cand_score + (int64_t)qi * kGemmTopkCap;
            const float gap = s[k - 1] - s[kk - 1];
            if (!(gap > 2.0f * eps * sqrtf(ss))) atomicOr(status, 1);
        }  // m < kk: every row of the index is in the list
    }
}

// Exact re-scoring of candidate rows with the scan's arithmetic: the score of (query, row) is the chain
//   for j = 0 .. D/16 - 1, e = 0 .. 3:  acc = mfma_16x16x4(a = q[16 j + 4 g + e], b = x[16 j + 4 g + e], acc)
// over lane groups g = 0..3 — exactly what k_cosine_score feeds for that pair (an MFMA output element depends
// only on its own row of A and column of B), times row_scale.  One wave per (query, 16 candidates): the query
// sits in row 0 of A, the candidates in the 16 columns of B.
__global__ __launch_bounds__(256) void k_gt_rescore(const float* __restrict__ q, int D, const float* __restrict__ idx,
                                                    const float* __restrict__ row_scale, const int32_t* __restrict__ cand_id,
                                                    const int32_t* __restrict__ cand_cnt, int Q, int kk,
                                                    float* __restrict__ exact /* [Q, kk] */) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const int tiles = (kk + 15) / 16;
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= (int64_t)Q * tiles) return;  // whole waves only: the MFMAs below run with all 64 lanes
    const int qi = (int)(w / tiles), t = (int)(w % tiles);
    const int m = cand_cnt[qi * kCntStride] < kk ? cand_cnt[qi * kCntStride] : kk;
    const int c = t * 16 + n;
    const int32_t row = c < m ? cand_id[(int64_t)qi * kGemmTopkCap + c] : -1;
    const f32x4* xp = reinterpret_cast<const f32x4*>(idx + (int64_t)(row >= 0 ? row : 0) * D) + g;
    const f32x4* qp = reinterpret_cast<const f32x4*>(q + (int64_t)qi * D) + g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // eight 64-byte pieces of the row in flight per lane group: the rows are scattered over the index, so a loop that
    // waits for each piece is bound by the HBM latency (48 round trips at D = 768); the MFMA order is unchanged
    constexpr int PF = 8;
    const int J = D / 16;
    for (int j0 = 0; j0 < J; j0 += PF) {
        f32x4 x[PF], a[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {  // unconditional loads (a piece past the end re-reads the last one): a bounds
            const int j = j0 + u < J ? j0 + u : J - 1;  // branch here would make the compiler wait for each load in turn
            x[u] = xp[j * 4];
            a[u] = n == 0 ? qp[j * 4] : zero;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (j0 + u < J) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], x[u][e], acc, 0, 0, 0);
            }
    }
    // D[row 0][col n] is register 0 of the lanes with g == 0
    if (g == 0 && c < kk) {
        float s = -INFINITY;
        if (row >= 0) s = row_scale ? acc[0] * row_scale[row] : acc[0];
        exact[(int64_t)qi * kk + c] = s;
    }
}

// The same for an f16-stored index: the scan multiplies the f16 rows with the query split hi + lo (lo scaled by
// kLoScale) on v_mfma_f32_16x16x32_f16 into two accumulators, chunk by chunk of 32 dims, and combines them as
// fma(acc_lo, 1 / kLoScale, acc).  Replayed here for (query in row 0, 16 candidates in the columns).
__global__ __launch_bounds__(256) void k_gt_rescore_f16(const float* __restrict__ q, int D, const _Float16* __restrict__ idx,
                                                        const float* __restrict__ row_scale, const int32_t* __restrict__ cand_id,
                                                        const int32_t* __restrict__ cand_cnt, int Q, int kk,
                                                        float* __restrict__ exact) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const int tiles = (kk + 15) / 16;
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= (int64_t)Q * tiles) return;
    const int qi = (int)(w / tiles), t = (int)(w % tiles);
    const int m = cand_cnt[qi * kCntStride] < kk ? cand_cnt[qi * kCntStride] : kk;
    const int c = t * 16 + n;
    const int32_t row = c < m ? cand_id[(int64_t)qi * kGemmTopkCap + c] : -1;
    const f16x8* xp = reinterpret_cast<const f16x8*>(idx + (int64_t)(row >= 0 ? row : 0) * D) + g;
    const float* qp = q + (int64_t)qi * D;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_lo = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < D / 32; ++j) {
        const f16x8 x = xp[j * 4];
        f16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {  // the split of k_query_fragments_f16
            const float v = n == 0 ? qp[32 * j + 8 * g + e] : 0.f;
            const _Float16 h = (_Float16)v;
            ah[e] = h;
            al[e] = (_Float16)((v - (float)h) * kLoScale);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, x, acc, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, x, acc_lo, 0, 0, 0);
    }
    if (g == 0 && c < kk) {
        float s = -INFINITY;
        if (row >= 0) {
            const float dot = fmaf(acc_lo[0], 1.0f / kLoScale, acc[0]);
            s = row_scale ? dot * row_scale[row] : dot;
        }
        exact[(int64_t)qi * kk + c] = s;
    }
}

// final selection on the exact scores: (score desc, row id asc), like the scan
__global__ __launch_bounds__(kSelectThreads) void k_gt_final(const float* __restrict__ exact, const int32_t* __restrict__ cand_id,
                                                             const int32_t* __restrict__ cand_cnt, int kk, int k,
                                                             int64_t row_id_base, float* __restrict__ out_score,
                                                             int64_t* __restrict__ out_index) {
    __shared__ SelectShared sh;
    const int qi = blockIdx.x;
    const int m0 = cand_cnt[qi * kCntStride] < kk ? cand_cnt[qi * kCntStride] : kk;
    const float* es = exact + (int64_t)qi * kk;
    const int32_t* ci = cand_id + (int64_t)qi * kGemmTopkCap;
    auto load = [&](int64_t i) -> uint64_t { return make_key(es[i], (uint32_t)ci[i]); };
    const int m = block_topk(sh, load, m0, k);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        if (i < m) {
            const uint64_t key = sh.keys[i];
            out_score[(int64_t)qi * k + i] = key_score(key);
            out_index[(int64_t)qi * k + i] = row_id_base + (int64_t)key_index(key);
        } else {
            out_score[(int64_t)qi * k + i] = -INFINITY;
            out_index[(int64_t)qi * k + i] = -1;
        }
    }
}

struct GtLayout {
    size_t tau, cnt, cs, ci, exact, scores, wsplit, total;
    int64_t slab;
};

static GtLayout gt_layout(int Q, int64_t N, int D, int k, int single = 1 /* the larger reserve */) {
    GtLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const int kk = gemm_topk_reserve(k, single) <= EVI_TOPK_MAX_K ? gemm_topk_reserve(k, single) : gemm_topk_reserve(k, 0);
    L.slab = N < kGemmSlabRows ? (N > 0 ? N : 1) : kGemmSlabRows;
    const int64_t first = N < kGemmFirstRows ? (N > 0 ? N : 1) : kGemmFirstRows;
    L.tau = take((size_t)Q * sizeof(float));
    L.cnt = take((size_t)Q * kCntStride * sizeof(int32_t));
    L.cs = take((size_t)Q * kGemmTopkCap * sizeof(float));
    L.ci = take((size_t)Q * kGemmTopkCap * sizeof(int32_t));
    L.exact = take((size_t)Q * kk * sizeof(float));
    L.scores = take((size_t)first * Q * sizeof(float));  // only the first slab's scores are materialised
    L.wsplit = take(gemm_bf16x3_workspace_bytes(Q, D));
    L.total = off;
    return L;
}

}  // namespace evi

extern "C" size_t evi_cosine_topk_gemm_workspace_bytes(int Q, int64_t N, int D, int k) {
    if (Q <= 0 || N < 0 || D <= 0 || k <= 0) return 0;
    return gt_layout(Q, N, D, k).total;
}

__global__ void k_shadow_bf16(const float* __restrict__ x, int64_t n, __bf16* __restrict__ out) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
        bf16x4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x4v*>(out + i) = o;
    }
}

extern "C" int evi_index_shadow_bf16(const float* idx, int64_t N, int D, void* out_bf16, void* stream) {
    EVI_REQUIRE(N >= 0 && D >= 1, "evi_index_shadow_bf16: need N >= 0 and D >= 1");
    if (D % 4 != 0) return fail(EVI_ERR_UNSUPPORTED, "evi_index_shadow_bf16: D must be a multiple of 4, got %d", D);
    if (N == 0) return EVI_OK;
    EVI_REQUIRE(idx && out_bf16, "evi_index_shadow_bf16: null pointer");
    hipLaunchKernelGGL(k_shadow_bf16, dim3(4096), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), idx, N * D,
                       static_cast<__bf16*>(out_bf16));
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

static int cosine_topk_gemm_impl(const float* q, int Q, const void* idx, int f16, int products, const void* shadow, int64_t N,
                                 int D, const float* row_scale, int k, int64_t row_id_base, float* out_score,
                                 int64_t* out_index, int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    EVI_REQUIRE(products == 1 || products == 3, "evi_cosine_topk_gemm: products must be 3 (split-bf16) or 1 (plain bf16), got %d",
                products);
    const int single = products == 1;
    EVI_REQUIRE(!shadow || (single && !f16), "evi_cosine_topk_gemm: a bf16 shadow goes with an f32 index and products = 1");
    EVI_REQUIRE(Q >= 1 && N >= 1 && D >= 1, "evi_cosine_topk_gemm: need Q >= 1, N >= 1, D >= 1, got Q=%d N=%lld D=%d", Q,
                (long long)N, D);
    EVI_REQUIRE(k >= 1, "evi_cosine_topk_gemm: k must be >= 1, got %d", k);
    if (D % (f16 ? 32 : 16) != 0)
        return fail(EVI_ERR_UNSUPPORTED, "evi_cosine_topk_gemm: D must be a multiple of %d, got %d", f16 ? 32 : 16, D);
    const int kk = gemm_topk_reserve(k, single);
    if (kk > EVI_TOPK_MAX_K)
        return fail(EVI_ERR_UNSUPPORTED, "evi_cosine_topk_gemm: k + reserve = %d exceeds %d (k <= %d with products = %d)", kk,
                    EVI_TOPK_MAX_K, single ? EVI_TOPK_MAX_K - 1024 : EVI_TOPK_MAX_K * 2 / 3, products);
    EVI_REQUIRE(N < (int64_t)0x7FFFFFFF, "evi_cosine_topk_gemm: N must fit int32 row ids");
    EVI_REQUIRE(q && idx && out_score && out_index && status && workspace, "evi_cosine_topk_gemm: null pointer");
    const GtLayout L = gt_layout(Q, N, D, k);
    if (workspace_bytes < L.total)
        return fail(EVI_ERR_NOMEM, "evi_cosine_topk_gemm: workspace %zu B < %zu B", workspace_bytes, L.total);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(workspace);
    float* tau = reinterpret_cast<float*>(base + L.tau);
    int32_t* cnt = reinterpret_cast<int32_t*>(base + L.cnt);
    float* cs = reinterpret_cast<float*>(base + L.cs);
    int32_t* ci = reinterpret_cast<int32_t*>(base + L.ci);
    float* exact = reinterpret_cast<float*>(base + L.exact);
    float* scores = reinterpret_cast<float*>(base + L.scores);
    // selection operand: the index itself, or its bf16 shadow (same arithmetic as products = 1, no conversion work)
    const int a_kind = shadow ? 2 : f16;
    const char* rows_base = static_cast<const char*>(shadow ? shadow : idx);
    const size_t row_bytes = (size_t)D * (a_kind ? 2 : 4);
    hipLaunchKernelGGL(k_gt_init, dim3((Q + 255) / 256), dim3(256), 0, st, tau, cnt, Q, status);
    EVI_LAUNCH_CHECK();
    int rc = split_weight_bf16x3(q, Q, D, D, base + L.wsplit, st);  // the queries are the "weights": split once
    if (rc != EVI_OK) return rc;
    int64_t begin = 0, seg = N < kGemmFirstRows ? N : kGemmFirstRows;
    while (begin < N) {
        const int64_t rows = (N - begin) < seg ? (N - begin) : seg;
        const void* slab = rows_base + begin * row_bytes;
        if (begin == 0) {
            // first slab: every score passes (tau = -inf), so form the scores and append them with plain stores
            rc = launch_gemm_bf16_presplit(slab, a_kind, single, rows, D, D, base + L.wsplit, Q, scores, Q, st);
            if (rc != EVI_OK) return rc;
            int64_t blocks = (rows * Q + 255) / 256;
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(k_gt_filter, dim3((unsigned)blocks), dim3(256), 0, st, scores, rows, Q, begin, row_scale, tau,
                               cs, ci, cnt, status);
        } else {
            // later slabs: scores ~ idx[begin + r] . q[i] never leave the registers of the GEMM — its epilogue
            // compares them with tau and appends the few survivors
            const GemmFilter flt{tau, row_scale, begin, cs, ci, cnt, status, kGemmTopkCap, kCntStride};
            rc = launch_gemm_bf16x3_filter(slab, a_kind, single, rows, D, D, base + L.wsplit, Q, flt, st);
            if (rc != EVI_OK) return rc;
        }
        hipLaunchKernelGGL(k_gt_clamp, dim3((Q + 255) / 256), dim3(256), 0, st, cnt, Q);
        hipLaunchKernelGGL(k_candidates_select, dim3(Q), dim3(kSelectThreads), 0, st, cs, ci, cnt, tau,
                           (int64_t)kGemmTopkCap, kk, (int64_t)-1, 0, row_id_base, (float*)nullptr, (int64_t*)nullptr,
                           (const int32_t*)nullptr);
        EVI_LAUNCH_CHECK();
        begin += rows;
        int64_t next = begin * kGemmGrowth;
        seg = next < L.slab ? next : L.slab;
    }
    hipLaunchKernelGGL(k_gt_gap, dim3(Q), dim3(64), 0, st, q, Q, D, cs, cnt, k, kk, single ? kApproxEps1 : kApproxEps, status);
    const int tiles = (kk + 15) / 16;
    const int64_t waves = (int64_t)Q * tiles;
    if (f16)
        hipLaunchKernelGGL(k_gt_rescore_f16, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, q, D,
                           static_cast<const _Float16*>(idx), row_scale, ci, cnt, Q, kk, exact);
    else
        hipLaunchKernelGGL(k_gt_rescore, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, q, D, static_cast<const float*>(idx),
                           row_scale, ci, cnt, Q, kk, exact);
    hipLaunchKernelGGL(k_gt_final, dim3(Q), dim3(kSelectThreads), 0, st, exact, ci, cnt, kk, k, row_id_base, out_score,
                       out_index);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_cosine_topk_gemm(const float* q, int Q, const float* idx, int64_t N, int D, const float* row_scale,
                                    int k, int64_t row_id_base, int products, const void* shadow_bf16, float* out_score,
                                    int64_t* out_index, int32_t* status, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    return cosine_topk_gemm_impl(q, Q, idx, 0, products, shadow_bf16, N, D, row_scale, k, row_id_base, out_score, out_index,
                                 status, workspace, workspace_bytes, stream);
}

extern "C" int evi_cosine_topk_gemm_f16(const float* q, int Q, const void* idx_f16, int64_t N, int D, const float* row_scale,
                                        int k, int64_t row_id_base, int products, float* out_score, int64_t* out_index,
                                        int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    return cosine_topk_gemm_impl(q, Q, idx_f16, 1, products, nullptr, N, D, row_scale, k, row_id_base, out_score, out_index,
                                 status, workspace, workspace_bytes, stream);
}

// =====================================================================================================
// Two-stage exact scan (f16 shadow selection + f32 re-scoring): the result of evi_cosine_topk at half
// the HBM bytes per batch.
//
// The f32 index is kept as it is and an f16 copy of it (the shadow, rn(x) element by element, made once
// by evi_index_shadow_f16: + 50 % memory) is what the scan streams.  Stage 1 is evi_cosine_topk_f16
// over the shadow with kk = k + reserve results per query.  For rows of norm <= 1 its scores differ
// from the f32 scan's by at most kShadowEps * |q|:
//     |q . (x - rn_f16(x))| <= 2^-11 |q| |x|                    (Cauchy-Schwarz, f16 has 11 significant bits)
//     f32 chain of the exact scan: 768 roundings x 2^-23         (truncating adders assumed)
//     f16-MFMA chain of the shadow scan + the hi/lo query split: < 5e-5
// so when approx[k-1] - approx[kk-1] > 2 eps |q| every row outside the kk kept ones is beaten, in exact
// score, by k kept rows: the true top-k is inside the list (same argument as the many-query path above).
// Stage 2 re-scores the kk rows from the f32 index with the scan's own v_mfma_f32_16x16x4_f32 chain and
// takes the top-k on those scores: ids and scores bit-identical to evi_cosine_topk.  A failed proof
// ORs 1 into *status (sticky: the caller zeroes it, so a pipeline of batches can share one flag and read it
// once) and the caller runs the f32 scan.
// =====================================================================================================
namespace evi {

constexpr float kShadowEps = 7.0e-4f;

static int two_stage_kk(int k) { return gemm_topk_reserve(k, 0); }

__global__ void k_shadow_f16(const float* __restrict__ x, int64_t n, _Float16* __restrict__ out) {
    // 4 elements per thread per step: one 16-byte load, one 8-byte store
    const int64_t n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4* o4 = reinterpret_cast<f16x4*>(out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = x4[i];
        f16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];  // round to nearest even
        o4[i] = h;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (_Float16)x[i];
}

// k_gt_rescore for [Q, kk] i64 candidate ids (-1 = padding): one wave per (query, 16 candidates)
__global__ __launch_bounds__(256) void k_ts_rescore(const float* __restrict__ q, int D, const float* __restrict__ idx,
                                                    const int64_t* __restrict__ ids, int Q, int kk,
                                                    float* __restrict__ exact /* [Q, kk] */) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const int tiles = (kk + 15) / 16;
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= (int64_t)Q * tiles) return;  // whole waves only: the MFMAs below run with all 64 lanes
    const int qi = (int)(w / tiles), t = (int)(w % tiles);
    const int c = t * 16 + n;
    const int64_t row = c < kk ? ids[(int64_t)qi * kk + c] : -1;
    const f32x4* xp = reinterpret_cast<const f32x4*>(idx + (row >= 0 ? row : 0) * D) + g;
    const f32x4* qp = reinterpret_cast<const f32x4*>(q + (int64_t)qi * D) + g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // eight 64-byte pieces of the row in flight per lane group: the rows are scattered over the index, so a loop that
    // waits for each piece is bound by the HBM latency (48 round trips at D = 768); the MFMA order is unchanged
    constexpr int PF = 8;
    const int J = D / 16;
    for (int j0 = 0; j0 < J; j0 += PF) {
        f32x4 x[PF], a[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {  // unconditional loads (a piece past the end re-reads the last one): a bounds
            const int j = j0 + u < J ? j0 + u : J - 1;  // branch here would make the compiler wait for each load in turn
            x[u] = xp[j * 4];
            a[u] = n == 0 ? qp[j * 4] : zero;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (j0 + u < J) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], x[u][e], acc, 0, 0, 0);
            }
    }
    if (g == 0 && c < kk) exact[(int64_t)qi * kk + c] = row >= 0 ? acc[0] : -INFINITY;
}

// Final top-k on the exact scores, and the proof: approx / ids are the [Q, kk] lists evi_cosine_topk_f16 wrote (sorted,
// (-inf, -1) past the end of a short index); the true top-k is among them when approx[k-1] - approx[kk-1] > 2 eps |q|.
__global__ __launch_bounds__(kSelectThreads) void k_ts_final(const float* __restrict__ exact, const int64_t* __restrict__ ids,
                                                             int kk, int k, int64_t row_id_base, float* __restrict__ out_score,
                                                             int64_t* __restrict__ out_index, const float* __restrict__ q, int D,
                                                             const float* __restrict__ approx, float eps,
                                                             int32_t* __restrict__ status, int32_t* __restrict__ call_flag) {
    __shared__ SelectShared sh;
    const int qi = blockIdx.x;
    const float* es = exact + (int64_t)qi * kk;
    const int64_t* ci = ids + (int64_t)qi * kk;
    if (threadIdx.x < 64) {  // wave 0: |q|^2 in a fixed order, then the gap test
        const int lane = threadIdx.x;
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) ss = fmaf(q[(int64_t)qi * D + d], q[(int64_t)qi * D + d], ss);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        if (lane == 0 && ci[kk - 1] >= 0) {  // otherwise every row of the index is in the list
            const float* s = approx + (int64_t)qi * kk;
            const float gap = s[k - 1] - s[kk - 1];
            // the bound is relative to |q| and assumes the f16 hi / lo split of the query keeps its 22 bits: far from unit
            // norm (elements near the f16 subnormal range, or the 2048 x lo part near overflow) it does not hold, and the
            // proof is refused — the f32 scan then does the batch
            const bool norm_ok = ss >= 0.0625f && ss <= 16.0f;
            if (!(gap > 2.0f * eps * sqrtf(ss)) || !norm_ok) {  // also catches NaN scores
                if (status) atomicOr(status, 1);  // the caller's sticky flag (a pipeline of batches shares it)
                atomicOr(call_flag, 1);           // this call's own flag: opens the gate of the fallback scan
            }
        }
    }
    auto load = [&](int64_t i) -> uint64_t { return ci[i] >= 0 ? make_key(es[i], (uint32_t)ci[i]) : 0ull; };
    const int m = block_topk(sh, load, kk, k);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint64_t key = i < m ? sh.keys[i] : 0ull;
        if (key != 0ull) {
            out_score[(int64_t)qi * k + i] = key_score(key);
            out_index[(int64_t)qi * k + i] = row_id_base + (int64_t)key_index(key);
        } else {
            out_score[(int64_t)qi * k + i] = -INFINITY;
            out_index[(int64_t)qi * k + i] = -1;
        }
    }
}

struct TsLayout {
    size_t flag, approx, ids, exact, scan, scan_bytes, total;
};
static TsLayout ts_layout(int Q, int64_t N, int D, int k) {
    TsLayout L;
    const int kk = two_stage_kk(k);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    L.flag = take(sizeof(int32_t));  // this call's proof flag (gate of the device-side fallback)
    L.approx = take((size_t)Q * kk * sizeof(float));
    L.ids = take((size_t)Q * kk * sizeof(int64_t));
    L.exact = take((size_t)Q * kk * sizeof(float));
    L.scan_bytes = evi_cosine_topk_workspace_bytes(Q, N, D, kk);
    L.scan = take(L.scan_bytes);
    L.total = off;
    return L;
}

}  // namespace evi

extern "C" int evi_index_shadow_f16(const float* idx, int64_t N, int D, void* out_f16, void* stream) {
    EVI_REQUIRE(N >= 0 && D >= 1, "evi_index_shadow_f16: need N >= 0, D >= 1");
    if (N == 0) return EVI_OK;
    EVI_REQUIRE(idx && out_f16, "evi_index_shadow_f16: null pointer");
    hipLaunchKernelGGL(k_shadow_f16, dim3(4096), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), idx, N * D,
                       static_cast<_Float16*>(out_f16));
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" size_t evi_cosine_topk_two_stage_workspace_bytes(int Q, int64_t N, int D, int k) {
    if (Q <= 0 || N < 0 || D <= 0 || k <= 0 || two_stage_kk(k) > EVI_TOPK_MAX_K) return 0;
    return ts_layout(Q, N, D, k).total;
}

extern "C" int evi_cosine_topk_two_stage(const float* q, int Q, const float* idx, const void* shadow_f16, int64_t N, int D,
                                         int k, int64_t row_id_base, float* out_score, int64_t* out_index, int32_t* status,
                                         int device_fallback, void* workspace, size_t workspace_bytes, void* stream) {
    EVI_REQUIRE(Q >= 1 && N >= 1, "evi_cosine_topk_two_stage: need Q >= 1 and N >= 1, got Q=%d N=%lld", Q, (long long)N);
    EVI_REQUIRE(k >= 1, "evi_cosine_topk_two_stage: k must be >= 1, got %d", k);
    if (D < 32 || D % 32 != 0 || D > 1280)
        return fail(EVI_ERR_UNSUPPORTED, "evi_cosine_topk_two_stage: D must be a multiple of 32 in [32, 1280], got %d", D);
    const int kk = two_stage_kk(k);
    if (kk > EVI_TOPK_MAX_K)
        return fail(EVI_ERR_UNSUPPORTED, "evi_cosine_topk_two_stage: k + reserve = %d exceeds %d (k <= %d)", kk, EVI_TOPK_MAX_K,
                    EVI_TOPK_MAX_K * 2 / 3);
    EVI_REQUIRE(q && idx && shadow_f16 && out_score && out_index && workspace, "evi_cosine_topk_two_stage: null pointer");
    EVI_REQUIRE(status || device_fallback,
                "evi_cosine_topk_two_stage: without the device-side fallback the caller must pass a status word and check it");
    const TsLayout L = ts_layout(Q, N, D, k);
    if (workspace_bytes < L.total)
        return fail(EVI_ERR_NOMEM, "evi_cosine_topk_two_stage: workspace %zu B < %zu B", workspace_bytes, L.total);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(workspace);
    float* approx = reinterpret_cast<float*>(base + L.approx);
    int64_t* ids = reinterpret_cast<int64_t*>(base + L.ids);
    float* exact = reinterpret_cast<float*>(base + L.exact);
    int32_t* call_flag = reinterpret_cast<int32_t*>(base + L.flag);
    EVI_HIP_CHECK(hipMemsetAsync(call_flag, 0, sizeof(int32_t), st));
    // stage 1: kk best rows per query by shadow score (local row ids)
    const int rc = cosine_topk_impl(q, Q, shadow_f16, 1, N, D, nullptr, kk, 0, approx, ids, base + L.scan, L.scan_bytes, stream);
    if (rc != EVI_OK) return rc;
    // stage 2: exact scores from the f32 rows, final top-k + proof
    const int tiles = (kk + 15) / 16;
    const int64_t waves = (int64_t)Q * tiles;
    hipEvent_t e0, e1;
    if (timing_kernel_events(kTimeSelect, &e0, &e1))
        hipExtLaunchKernelGGL(k_ts_rescore, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, e0, e1, 0, q, D, idx,
                              (const int64_t*)ids, Q, kk, exact);
    else
        hipLaunchKernelGGL(k_ts_rescore, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, q, D, idx, ids, Q, kk, exact);
    if (timing_kernel_events(kTimeSelect, &e0, &e1))
        hipExtLaunchKernelGGL(k_ts_final, dim3(Q), dim3(kSelectThreads), 0, st, e0, e1, 0, (const float*)exact,
                              (const int64_t*)ids, kk, k, row_id_base, out_score, out_index, q, D, (const float*)approx,
                              kShadowEps, status, call_flag);
    else
        hipLaunchKernelGGL(k_ts_final, dim3(Q), dim3(kSelectThreads), 0, st, exact, ids, kk, k, row_id_base, out_score, out_index,
                           q, D, approx, kShadowEps, status, call_flag);
    EVI_LAUNCH_CHECK();
    if (device_fallback) {
        // The f32 scan of the same batch into the same outputs, gated on this call's flag: while the proof held every
        // kernel returns at once (a few empty launches); when it failed the scan overwrites the unproven result before any
        // later work of the stream can read it.  No read-back, no agreement between ranks needed: each shard's record is
        // exact when it leaves the device.  The stage-1 scan workspace is free again and large enough (kk >= k).
        const int rc2 = cosine_topk_impl(q, Q, idx, 0, N, D, nullptr, k, row_id_base, out_score, out_index, base + L.scan,
                                         L.scan_bytes, stream, call_flag);
        if (rc2 != EVI_OK) return rc2;
    }
    return EVI_OK;
}
