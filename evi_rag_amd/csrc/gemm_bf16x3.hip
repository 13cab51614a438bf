// Split-bf16 ("bf16x3") GEMM for the scorer:  C[M,N] = act(A[M,K] * W[N,K]^T + bias), f32 in / f32 out.
//
// gfx950 has no TF32: an f32-input MFMA runs at 1/16 of the bf16 rate.  Each f32 operand is split
// into two bf16 values, x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and the product is formed
// as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation: three bf16 MFMAs
// instead of sixteen f32-MFMA-equivalents.  The dropped lo*lo term and the rounding of lo are both
// <= 2^-18 relative per product, so a length-K dot product is accurate to ~1e-5 of sum|a w| —
// two decades inside the 1e-3 the scorer is allowed (and comparable to the TF32 the reference runs
// with on CUDA, configs/extras/default.yaml:11).  evi_gemm_nt_f32 stays available as the exact path.
//
// 256 x 256 x 32 block tile, 8 waves (2 x 4), each wave 128 x 64 = eight 32x32 accumulators.
// W is pre-split once per call into bf16 hi / lo planes (it is reused by every block); A is split
// while it is staged global -> VGPR -> LDS.  LDS rows are 64 B (32 bf16); the four 16-byte chunks of
// a row are XOR-swizzled with (row >> 2) & 3 so every ds_read_b128 lane group is conflict-free.
#include "common.hpp"

#include <hip/hip_bf16.h>
#include <stdlib.h>

namespace evi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // a 16-byte chunk (ext vector: selects stay in registers)

constexpr int XM = 256, XN = 256, XK = 32;
constexpr int kXThreads = 512;

__device__ inline float act3(float v, int act) {
    if (act == 1) return tanhf(v);
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}

// 16-byte slot of chunk c (0..3) of row r in a [rows][32 bf16] tile
__device__ inline int slot3(int r, int c) { return r * 4 + (c ^ ((r >> 2) & 3)); }

// W [N, K] f32 (row stride ldw) -> hi / lo bf16 planes [N, Kp], zero padded to Kp
__global__ void k_split_weight(const float* __restrict__ W, int N, int K, int64_t ldw, int Kp,
                               __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Kp) return;
    const int n = (int)(i / Kp), k = (int)(i % Kp);
    const float x = k < K ? W[(int64_t)n * ldw + k] : 0.f;
    const __bf16 h = (__bf16)x;
    hi[i] = h;
    lo[i] = (__bf16)(x - (float)h);
}

// W [N, K] f32 -> ONE f16 plane [N, Kp] (round to nearest), written where the bf16 hi plane would go: the f16x2 form
__global__ void k_round_weight_f16(const float* __restrict__ W, int N, int K, int64_t ldw, int Kp, _Float16* __restrict__ hi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Kp) return;
    const int n = (int)(i / Kp), k = (int)(i % Kp);
    hi[i] = (_Float16)(k < K ? W[(int64_t)n * ldw + k] : 0.f);
}

// the same for very long rows (64-bit element count): the W operand of a split-K batch
__global__ void k_split_weight_long(const float* __restrict__ W, int N, int64_t K, int64_t ldw, int64_t Kp,
                                    __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Kp) return;
    const int64_t n = i / Kp, k = i % Kp;
    const float x = k < K ? W[n * ldw + k] : 0.f;
    const __bf16 h = (__bf16)x;
    hi[i] = h;
    lo[i] = (__bf16)(x - (float)h);
}

// MF = 0: v_mfma_f32_32x32x16_bf16 (eight 32x32 accumulators per wave); MF = 1: v_mfma_f32_16x16x32_bf16
// (thirty-two 16x16 accumulators) — same LDS image, same number of fragment reads and MFMA cycles per
// k-tile; the chip holds a higher clock on the 16x16 shape (MI355X_MICROARCH.md, DVFS item 7).
// AH = 1: A is stored as f16 (lda in halves) — an f16 value is exactly hi + lo in bf16, so nothing is lost.
// AH = 2: A is a bf16 copy of the rows (the "shadow" of an f32 index, = its hi plane): staged without any
//         conversion; only meaningful with P1 (there is no lo plane to multiply).
// P1 = 1: single product hi * hi (plain bf16 GEMM: relative error 2^-8 instead of 2^-16) — only for the candidate
// selection of the many-query top-k, whose results are re-scored exactly; the lo planes are neither loaded,
// staged nor multiplied.
// P1 = 2 ("f16x2", opt-in, f32 A only, MF = 0): TWO products on v_mfma_f32_32x32x16_f16 — A = hi + lo with hi = f16(a),
// lo = f16(a - hi) (22 significant bits), W rounded ONCE to f16 (2^-12 relative per weight; `Whi` then holds f16 bits,
// there is no W lo plane).  Two thirds of the MFMA work of the three-product form; the error is the weights' f16 rounding —
// four times finer than the TF32 rounding of BOTH operands the reference runs with on CUDA
// (configs/extras/default.yaml:11) — and f16's range: |a|, |w| must stay below 65 504.
template <int ACT, int MF, int AH = 0, int P1 = 0>
__global__ __launch_bounds__(kXThreads) void k_gemm_nt_bf16x3(
    const float* __restrict__ A, int64_t M, int K, int64_t lda, const __bf16* __restrict__ Whi,
    const __bf16* __restrict__ Wlo, int N, int Kp, const float* __restrict__ bias, float* __restrict__ C,
    int64_t ldc, GemmFilter flt, GemmBatch bt) {
    // bt.count > 0 (split-K batch, gridDim.y = bt.count): independent products of K-slices of the same operands (split-K of a TN product): slice s multiplies
    // columns [s * K, s * K + K) of A and of the W planes (row stride bt.w_ld) into C + s * bt.c_stride
    if (bt.count > 0) {
        const int sl = blockIdx.y;
        const int64_t k_begin = (int64_t)sl * K;
        A += k_begin;
        Whi += k_begin;
        Wlo += k_begin;
        C += (int64_t)sl * bt.c_stride;
        const int64_t left = bt.k_total - k_begin;
        K = left < K ? (int)(left > 0 ? left : 0) : K;
    }
    const int64_t w_ld = bt.count > 0 ? bt.w_ld : (int64_t)Kp;
    if (bt.count > 0) Kp = (K + XK - 1) / XK * XK;
    constexpr bool kALo = P1 != 1;  // A lo plane staged and multiplied (three-product and f16x2 forms)
    constexpr bool kWLo = P1 == 0;  // W lo plane (three-product form only)
    static_assert(P1 != 2 || (AH == 0 && MF == 0), "f16x2 takes f32 A and the 32x32x16 MFMA");
    __shared__ uint4 sAhi[2][XM * 4], sWhi[2][XN * 4];                                     // 2 stages x 16 KiB each
    __shared__ uint4 sAlo[kALo ? 2 : 1][kALo ? XM * 4 : 1], sWlo[kWLo ? 2 : 1][kWLo ? XN * 4 : 1];  // lo planes where the form has them
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nblocks_n = (N + XN - 1) / XN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2),
    // so give every XCD a contiguous run of logical tiles — the n-tiles that share an A row panel
    // then hit the same L2 instead of fetching the panel once per XCD.  Pure speed; a bijection of the grid.
    unsigned nwg = gridDim.x;
    if (bt.m_dev) {
        // the caller knows only a bound on M: the count is on the device (scorer: distinct (relation, graph) pairs).  The first
        // `live` workgroups cover it — consecutive workgroup ids are dealt round-robin over the XCDs, so the live ones stay spread
        // over all eight — and the rest exit (uniform over the workgroup, before any barrier)
        const int64_t m_now = *bt.m_dev;
        if (m_now < M) M = m_now;
        const unsigned live = (unsigned)((M + XM - 1) / XM) * (unsigned)nblocks_n;
        if (blockIdx.x >= live) return;
        nwg = live;
    }
    // the remap covers the largest multiple of 8 (identity on the few workgroups beyond it)
    const unsigned nwg8 = nwg - nwg % 8;
    const unsigned tile = blockIdx.x < nwg8 ? (blockIdx.x % 8) * (nwg8 / 8) + blockIdx.x / 8 : blockIdx.x;
    const int64_t m0 = (int64_t)(tile / nblocks_n) * XM;
    const int n0 = (int)(tile % nblocks_n) * XN;

    // Staging registers.  NS sets: a tile is loaded NS iterations before it is written to LDS.  One set when all three
    // products are multiplied (the register file is full); two with P1, whose k-tile takes a third of the MFMA time —
    // with one set its loop ran at the latency of a global load per k-tile, not at the speed of its MFMAs.
    constexpr int NS = P1 == 1 ? 2 : 1;
    f32x4 ra[NS][4];
    u32x4 ra16[NS][2];  // AH != 0: two chunks of 8 halves per thread
    u32x4 rwh[NS][2], rwl[NS][2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const u32x4 zero16 = {0u, 0u, 0u, 0u};
    // The staging work of one k-tile is cut into six pieces (four A pieces of 512 float4 chunks, two W
    // pieces of 512 16-byte chunks per plane) so that it can be issued BETWEEN the MFMA groups of the
    // tile being multiplied: a 32x32x16 MFMA holds the SIMD's issue port for 8 of its 32 cycles, the
    // conversions and LDS writes of the next tile run in its shadow instead of in a phase of their own.
    // Loads are branch-free (an out-of-range chunk reads the array's first 16 bytes and is zeroed by a
    // select), so the whole k-loop body is straight-line code and the compiler can count exactly which
    // outstanding loads each staging piece has to wait for.
    auto load_a = [&](int set, int i, int k0) {  // A: 2048 float4 chunks, 8 per row (128 B contiguous)
        const int s = tid + kXThreads * i;
        const int r = s >> 3, c4 = s & 7;
        const int k = k0 + c4 * 4;
        const int64_t am = m0 + r;
        const bool ok = am < M && k < K;
        ra[set][i] = *reinterpret_cast<const f32x4*>(A + (ok ? am * lda + k : 0));  // zeroed where it is consumed (store_a)
    };
    auto load_w = [&](int set, int i, int k0) {  // W planes: 1024 16-byte chunks each, 4 per row
        const int s = tid + kXThreads * i;
        const int r = s >> 2, c = s & 3;
        const int wr = n0 + r;
        const int k = k0 + c * 8;
        const bool ok = wr < N && k < Kp;
        const int64_t off = ok ? (int64_t)wr * w_ld + k : 0;
        rwh[set][i] = *reinterpret_cast<const u32x4*>(Whi + off);
        if (kWLo) rwl[set][i] = *reinterpret_cast<const u32x4*>(Wlo + off);
    };
    // k0 = first k of the tile the registers hold: the range check of the load is repeated here, so the
    // select sits next to the conversion and not behind the load (where it would stall on the load's latency)
    auto store_a = [&](int set, int i, int buf, int k0) {
        const int s = tid + kXThreads * i;
        const int r = s >> 3, c4 = s & 7;
        const bool ok = m0 + r < M && k0 + c4 * 4 < K;
        const f32x4 v = ok ? ra[set][i] : zero4;
        uint2 hb, lb;
        if (P1 == 2) {  // f16 hi / lo
            typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
            f16x4v h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (_Float16)v[e];
                l[e] = (_Float16)(v[e] - (float)h[e]);
            }
            hb = __builtin_bit_cast(uint2, h);
            lb = __builtin_bit_cast(uint2, l);
        } else {
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (__bf16)v[e];
                l[e] = (__bf16)(v[e] - (float)h[e]);
            }
            hb = *reinterpret_cast<uint2*>(&h);
            lb = *reinterpret_cast<uint2*>(&l);
        }
        // 8-byte halves of the 16-byte chunk c = c4 >> 1
        uint2* dh = reinterpret_cast<uint2*>(&sAhi[buf][slot3(r, c4 >> 1)]) + (c4 & 1);
        *dh = hb;
        if (kALo) {
            uint2* dl = reinterpret_cast<uint2*>(&sAlo[kALo ? buf : 0][kALo ? slot3(r, c4 >> 1) : 0]) + (c4 & 1);
            *dl = lb;
        }
    };
    // f16-stored A: 1024 chunks of 8 halves (16 B), 4 per row
    auto load_a16 = [&](int set, int i, int k0) {
        const int s = tid + kXThreads * i;
        const int r = s >> 2, c = s & 3;
        const int k = k0 + c * 8;
        const int64_t am = m0 + r;
        const bool ok = am < M && k < K;
        ra16[set][i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const _Float16*>(A) + (ok ? am * lda + k : 0));
    };
    auto store_a16 = [&](int set, int i, int buf, int k0) {
        const int s = tid + kXThreads * i;
        const int r = s >> 2, c = s & 3;
        const bool ok = m0 + r < M && k0 + c * 8 < K;
        if (AH == 2) {  // already bf16: straight to the hi plane
            *reinterpret_cast<u32x4*>(&sAhi[buf][slot3(r, c)]) = ok ? ra16[set][i] : zero16;
            return;
        }
        typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
        const f16x8v v = __builtin_bit_cast(f16x8v, ok ? ra16[set][i] : zero16);
        bf16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            h[e] = (__bf16)f;
            l[e] = (__bf16)(f - (float)h[e]);
        }
        *reinterpret_cast<bf16x8*>(&sAhi[buf][slot3(r, c)]) = h;
        if (kALo) *reinterpret_cast<bf16x8*>(&sAlo[kALo ? buf : 0][kALo ? slot3(r, c) : 0]) = l;
    };
    auto store_w = [&](int set, int i, int buf, int k0) {
        const int s = tid + kXThreads * i;
        const int r = s >> 2, c = s & 3;
        const bool ok = n0 + r < N && k0 + c * 8 < Kp;
        *reinterpret_cast<u32x4*>(&sWhi[buf][slot3(r, c)]) = ok ? rwh[set][i] : zero16;
        if (kWLo) *reinterpret_cast<u32x4*>(&sWlo[kWLo ? buf : 0][kWLo ? slot3(r, c) : 0]) = ok ? rwl[set][i] : zero16;
    };

    f32x16 acc[MF ? 1 : 4][MF ? 1 : 2];   // MF = 0
    f32x4 acc16[MF ? 8 : 1][MF ? 4 : 1];  // MF = 1
    if (MF) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[MF ? i : 0][MF ? j : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[MF ? 0 : i][MF ? 0 : j][r] = 0.f;
    }

    const int fr = lane & 31, fh = lane >> 5;    // 32x32x16 fragment coordinates
    const int r16 = lane & 15, q16 = lane >> 4;  // 16x16x32 fragment coordinates
    // Two LDS stages: stage `cur` is multiplied while the next tile (already in registers) is split and
    // written to the other stage piece by piece and the tile after it is fetched into the freed
    // registers; one barrier per k-tile.
    auto load_tile = [&](int set, int k0) {
        if (AH) {
#pragma unroll
            for (int i = 0; i < 2; ++i) load_a16(set, i, k0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) load_a(set, i, k0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) load_w(set, i, k0);
    };
    load_tile(0, 0);
    if (AH) {
#pragma unroll
        for (int i = 0; i < 2; ++i) store_a16(0, i, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_a(0, i, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) store_w(0, i, 0, 0);
    __syncthreads();
#pragma unroll
    for (int t = 1; t <= NS; ++t) load_tile(t % NS, t * XK);  // tiles 1 .. NS are in flight when the loop starts
    // NS == 2: two k-tiles per trip so that the staging set and the LDS stage of each are compile-time constants.  A
    // trip may run one tile past the end: that tile was staged as zeros and adds nothing.
    int cur_rt = 0;
    for (int kbase = 0; kbase < K; kbase += XK * NS) {
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int k0 = kbase + u * XK;
        const int cur = NS == 2 ? u : cur_rt;           // LDS stage holding tile k0
        const int set = NS == 2 ? (u ^ 1) : 0;          // staging set holding tile k0 + XK
        // Eight MFMA groups per k-tile: (MF = 0) two 16-wide k-steps x four 32-row blocks, six MFMAs each;
        // (MF = 1) eight 16-row blocks over the whole 32-wide k-tile, twelve MFMAs each.  The A fragments of
        // group g + 1 are read from LDS before group g's MFMAs are issued (ping-pong registers), so an MFMA
        // never waits on the LDS latency of its own operands.
        // A fragments are read AHEAD groups before their MFMAs: one group (six MFMAs = 192 cycles) covers the LDS
        // latency when three products are multiplied; with one product a group is two MFMAs, so read two ahead.
        constexpr int AHEAD = P1 == 1 ? 2 : 1, RING = AHEAD + 1;
        bf16x8 fah[RING], fal[RING], fbh[MF ? 1 : 2][MF ? 4 : 2], fbl[MF ? 1 : 2][MF ? 4 : 2];
        auto read_a = [&](int g) {
            const int c = MF ? q16 : ((g >> 2) << 1) + fh;  // MF = 0: chunk holding k = 16 (g >> 2) + 8 h .. + 7
            const int row = MF ? wm * 128 + g * 16 + r16 : wm * 128 + (g & 3) * 32 + fr;
            fah[g % RING] = *reinterpret_cast<const bf16x8*>(&sAhi[cur][slot3(row, c)]);
            if (kALo) fal[g % RING] = *reinterpret_cast<const bf16x8*>(&sAlo[kALo ? cur : 0][kALo ? slot3(row, c) : 0]);
        };
        auto read_b = [&](int ks) {
            const int c = MF ? q16 : (ks << 1) + fh;
#pragma unroll
            for (int j = 0; j < (MF ? 4 : 2); ++j) {
                const int row = MF ? wn * 64 + j * 16 + r16 : wn * 64 + j * 32 + fr;
                fbh[ks][j] = *reinterpret_cast<const bf16x8*>(&sWhi[cur][slot3(row, c)]);
                if (kWLo) fbl[ks][j] = *reinterpret_cast<const bf16x8*>(&sWlo[kWLo ? cur : 0][kWLo ? slot3(row, c) : 0]);
            }
        };
        read_b(0);
#pragma unroll
        for (int g = 0; g < AHEAD; ++g) read_a(g);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + AHEAD < 8) read_a(g + AHEAD);
            if (!MF && g == 2) read_b(1);
            if (MF) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4& c = acc16[MF ? g : 0][MF ? j : 0];
                    if (!P1) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[g % RING], fbh[0][MF ? j : 0], c, 0, 0, 0);
                    if (!P1) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[g % RING], fbl[0][MF ? j : 0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[g % RING], fbh[0][MF ? j : 0], c, 0, 0, 0);
                }
            } else {
                const int i = g & 3, ks = g >> 2;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16& c = acc[MF ? 0 : i][MF ? 0 : j];
                    if (P1 == 2) {  // f16x2: (lo + hi) * W on the f16 instruction, the small term first
                        typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
                        const f16x8v wv = __builtin_bit_cast(f16x8v, fbh[MF ? 0 : ks][MF ? 0 : j]);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, fal[g % RING]), wv, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, fah[g % RING]), wv, c, 0, 0, 0);
                        continue;
                    }
                    if (!P1) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[g % RING], fbh[MF ? 0 : ks][MF ? 0 : j], c, 0, 0, 0);
                    if (!P1) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[g % RING], fbl[MF ? 0 : ks][MF ? 0 : j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[g % RING], fbh[MF ? 0 : ks][MF ? 0 : j], c, 0, 0, 0);
                }
            }
            // one staging piece in the shadow of this group's MFMAs (pieces 0-3: A, 4-5: W).  Past the last
            // tile the pieces move zeros into a stage nobody reads: cheaper than a branch in this loop.
            if (AH) {
                if (g < 2) {
                    store_a16(set, g, cur ^ 1, k0 + XK);
                    load_a16(set, g, k0 + (1 + NS) * XK);
                } else if (g >= 4 && g < 6) {
                    store_w(set, g - 4, cur ^ 1, k0 + XK);
                    load_w(set, g - 4, k0 + (1 + NS) * XK);
                }
            } else if (g < 4) {
                store_a(set, g, cur ^ 1, k0 + XK);
                load_a(set, g, k0 + (1 + NS) * XK);
            } else if (g < 6) {
                store_w(set, g - 4, cur ^ 1, k0 + XK);
                load_w(set, g - 4, k0 + (1 + NS) * XK);
            }
            // keep each piece (and the reload of its registers) in its own MFMA group: left alone, the
            // scheduler sinks all eight loads to the end of the loop, one barrier before they are needed
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        cur_rt ^= 1;
      }
    }

    if (ACT == 3) {  // threshold filter instead of a store (MF = 0 layout): survivors are rare after the first slab
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + fr;
            if (n >= N) continue;
            const float t = flt.tau[n];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (m >= M) continue;
                    float sc = acc[MF ? 0 : i][MF ? 0 : j][r];
                    if (flt.row_scale) sc *= flt.row_scale[flt.row0 + m];
                    if (sc >= t || sc != sc) {
                        const int32_t pos = atomicAdd(&flt.cand_cnt[n * flt.cnt_stride], 1);
                        if (pos < flt.cap) {
                            flt.cand_score[(int64_t)n * flt.cap + pos] = sc;
                            flt.cand_id[(int64_t)n * flt.cap + pos] = (int32_t)(flt.row0 + m);
                        } else {
                            atomicOr(flt.status, 2);
                        }
                    }
                }
        }
        return;
    }
    if (MF) {  // 16x16 accumulators: col = lane & 15, row = 4 (lane >> 4) + r
        const bool interior = m0 + XM <= M && n0 + XN <= N;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + r16;
            if (!interior && n >= N) continue;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + wm * 128 + i * 16 + q16 * 4 + r;
                    if (interior || m < M) C[m * ldc + n] = act3(acc16[MF ? i : 0][MF ? j : 0][r] + bv, ACT);
                }
        }
        return;
    }
    if (m0 + XM <= M && n0 + XN <= N) {  // interior tile: no bounds checks, the 128 stores of a lane issue back to back
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + fr;
            const float bv = bias ? bias[n] : 0.f;
            float* cp = C + (m0 + wm * 128 + 4 * fh) * ldc + n;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[(int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc] = act3(acc[i][j][r] + bv, ACT);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m < M) C[m * ldc + n] = act3(acc[i][j][r] + bv, ACT);
            }
        }
    }
}

size_t gemm_bf16x3_workspace_bytes(int N, int K) {
    const int Kp = (K + XK - 1) / XK * XK;
    return align_up((size_t)N * Kp * 2, 256) * 2;
}

// wsplit: gemm_bf16x3_workspace_bytes(N, K) bytes of scratch for the split weight planes.
int split_weight_bf16x3(const float* W, int N, int K, int64_t ldw, void* wsplit, hipStream_t st) {
    const int Kp = (K + XK - 1) / XK * XK;
    __bf16* hi = static_cast<__bf16*>(wsplit);
    __bf16* lo = reinterpret_cast<__bf16*>(static_cast<char*>(wsplit) + align_up((size_t)N * Kp * 2, 256));
    const int64_t total = (int64_t)N * Kp;
    hipLaunchKernelGGL(k_split_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, ldw, Kp, hi, lo);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

int launch_gemm_bf16_presplit(const void* A, int a_f16, int single, int64_t M, int K, int64_t lda, const void* wsplit, int N,
                              float* C, int64_t ldc, hipStream_t st) {
    if (M == 0 || N == 0) return EVI_OK;
    const int Kp = (K + XK - 1) / XK * XK;
    const __bf16* hi = static_cast<const __bf16*>(wsplit);
    const __bf16* lo = reinterpret_cast<const __bf16*>(static_cast<const char*>(wsplit) + align_up((size_t)N * Kp * 2, 256));
    const dim3 grid((unsigned)(((M + XM - 1) / XM) * ((N + XN - 1) / XN)));
    const GemmFilter flt{};
    const int tok = timing_begin(kTimeGemm, st);
#define EVI_LAUNCH_PS(AHV, P1V)                                                                                          \
    hipLaunchKernelGGL((k_gemm_nt_bf16x3<0, 0, AHV, P1V>), grid, dim3(kXThreads), 0, st, static_cast<const float*>(A), M, K, lda, \
                       hi, lo, N, Kp, static_cast<const float*>(nullptr), C, ldc, flt, GemmBatch{})
    if (a_f16 == 2) EVI_LAUNCH_PS(2, 1);
    else if (a_f16 && single) EVI_LAUNCH_PS(1, 1);
    else if (a_f16) EVI_LAUNCH_PS(1, 0);
    else if (single) EVI_LAUNCH_PS(0, 1);
    else EVI_LAUNCH_PS(0, 0);
#undef EVI_LAUNCH_PS
    timing_end(tok, st);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

int launch_gemm_bf16x3_filter(const void* A, int a_f16, int single, int64_t M, int K, int64_t lda, const void* wsplit, int N,
                              const GemmFilter& flt, hipStream_t st) {
    if (M == 0 || N == 0) return EVI_OK;
    const int Kp = (K + XK - 1) / XK * XK;
    const __bf16* hi = static_cast<const __bf16*>(wsplit);
    const __bf16* lo = reinterpret_cast<const __bf16*>(static_cast<const char*>(wsplit) + align_up((size_t)N * Kp * 2, 256));
    const dim3 grid((unsigned)(((M + XM - 1) / XM) * ((N + XN - 1) / XN)));
    const int tok = timing_begin(kTimeGemm, st);
#define EVI_LAUNCH_FL(AHV, P1V)                                                                                          \
    hipLaunchKernelGGL((k_gemm_nt_bf16x3<3, 0, AHV, P1V>), grid, dim3(kXThreads), 0, st, static_cast<const float*>(A), M, K, lda, \
                       hi, lo, N, Kp, static_cast<const float*>(nullptr), static_cast<float*>(nullptr), (int64_t)0, flt, GemmBatch{})
    if (a_f16 == 2) EVI_LAUNCH_FL(2, 1);
    else if (a_f16 && single) EVI_LAUNCH_FL(1, 1);
    else if (a_f16) EVI_LAUNCH_FL(1, 0);
    else if (single) EVI_LAUNCH_FL(0, 1);
    else EVI_LAUNCH_FL(0, 0);
#undef EVI_LAUNCH_FL
    timing_end(tok, st);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

int launch_gemm_nt_bf16x3(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw,
                          const float* bias, int act, float* C, int64_t ldc, void* wsplit, hipStream_t st, int single,
                          const int32_t* m_dev) {
    if (M == 0 || N == 0) return EVI_OK;
    if (single == 2) {  // f16x2: the weight as one f16 plane in the hi slot of `wsplit`
        const int Kp = (K + XK - 1) / XK * XK;
        const int64_t total = (int64_t)N * Kp;
        hipLaunchKernelGGL(k_round_weight_f16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, ldw, Kp,
                           static_cast<_Float16*>(wsplit));
        EVI_LAUNCH_CHECK();
    } else if (int rc = split_weight_bf16x3(W, N, K, ldw, wsplit, st)) {
        return rc;
    }
    return launch_gemm_nt_bf16x3_wplanes(A, M, K, lda, wsplit, N, bias, act, C, ldc, st, single, m_dev);
}

// the same with W already split (split_weight_bf16x3 wrote `wplanes`): what a caller that keeps its weights prepared uses.
// single == 1: one bf16 product (hi * hi only, f32 accumulation and output) — the arithmetic of a bf16-autocast Linear
// with an f32 result; the scorer's opt-in `bf16-mixed` training precision.
// single == 2: the f16x2 form (see the kernel): `wplanes` must then hold the weight as ONE f16 plane (k_round_weight_f16).
int launch_gemm_nt_bf16x3_wplanes(const float* A, int64_t M, int K, int64_t lda, const void* wplanes, int N,
                                  const float* bias, int act, float* C, int64_t ldc, hipStream_t st, int single,
                                  const int32_t* m_dev) {
    if (M == 0 || N == 0) return EVI_OK;
    const int Kp = (K + XK - 1) / XK * XK;
    GemmBatch plain;
    plain.m_dev = m_dev;
    const __bf16* hi = static_cast<const __bf16*>(wplanes);
    const __bf16* lo = reinterpret_cast<const __bf16*>(static_cast<const char*>(wplanes) + align_up((size_t)N * Kp * 2, 256));
    const GemmFilter flt{};
    const dim3 grid((unsigned)(((M + XM - 1) / XM) * ((N + XN - 1) / XN)));
    const int tok = timing_begin(kTimeGemm, st);
    // EVI_GEMM_MFMA=16 selects the 16x16x32 form (tuning knob; measured 3-10 % slower here: its stores are 64-byte
    // segments and the clock advantage of the shape does not make up for it)
    static const bool mfma16 = [] {
        const char* e = getenv("EVI_GEMM_MFMA");
        return e && e[0] == '1';
    }();
#define EVI_LAUNCH_X3(ACT)                                                                                              \
    if (single == 2)                                                                                                    \
        hipLaunchKernelGGL((k_gemm_nt_bf16x3<ACT, 0, 0, 2>), grid, dim3(kXThreads), 0, st, A, M, K, lda, hi, lo, N, Kp, bias, C, ldc, flt, plain); \
    else if (single)                                                                                                    \
        hipLaunchKernelGGL((k_gemm_nt_bf16x3<ACT, 0, 0, 1>), grid, dim3(kXThreads), 0, st, A, M, K, lda, hi, lo, N, Kp, bias, C, ldc, flt, plain); \
    else if (mfma16)                                                                                                    \
        hipLaunchKernelGGL((k_gemm_nt_bf16x3<ACT, 1>), grid, dim3(kXThreads), 0, st, A, M, K, lda, hi, lo, N, Kp, bias, C, ldc, flt, plain); \
    else                                                                                                                \
        hipLaunchKernelGGL((k_gemm_nt_bf16x3<ACT, 0>), grid, dim3(kXThreads), 0, st, A, M, K, lda, hi, lo, N, Kp, bias, C, ldc, flt, plain);
    switch (act) {
        case 1: EVI_LAUNCH_X3(1) break;
        case 2: EVI_LAUNCH_X3(2) break;
        default: EVI_LAUNCH_X3(0)
    }
#undef EVI_LAUNCH_X3
    timing_end(tok, st);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

// Split-K batch: C_s [M, N] = A[:, s Ks : (s+1) Ks] W[:, same]^T for s < S in ONE launch (gridDim.y = S), A [M, Ktot] f32 with row
// stride lda, W [N, Ktot] f32 split here into planes of row stride Kp_tot (wsplit: gemm_bf16x3_workspace_bytes(N, Ktot)).
// Ks a multiple of 32.  Used by the TN products of the scorer's backward, whose output is only a few tiles.
int launch_gemm_nt_bf16x3_splitk(const float* A, int64_t M, int64_t Ktot, int64_t lda, const float* W, int N, int64_t ldw, int Ks,
                                 int S, float* Cparts, void* wsplit, hipStream_t st) {
    if (M == 0 || N == 0 || S == 0) return EVI_OK;
    const int64_t Kp_tot = (Ktot + XK - 1) / XK * XK;
    __bf16* hi = static_cast<__bf16*>(wsplit);
    __bf16* lo = reinterpret_cast<__bf16*>(static_cast<char*>(wsplit) + align_up((size_t)N * Kp_tot * 2, 256));
    {
        const int64_t total = (int64_t)N * Kp_tot;
        hipLaunchKernelGGL(k_split_weight_long, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, Ktot, ldw, Kp_tot, hi, lo);
        EVI_LAUNCH_CHECK();
    }
    const GemmFilter flt{};
    GemmBatch bt;
    bt.count = S;
    bt.c_stride = M * (int64_t)N;
    bt.k_total = Kp_tot;
    bt.w_ld = Kp_tot;
    const dim3 grid((unsigned)(((M + XM - 1) / XM) * ((N + XN - 1) / XN)), (unsigned)S);
    const int tok = timing_begin(kTimeGemm, st);
    hipLaunchKernelGGL((k_gemm_nt_bf16x3<0, 0>), grid, dim3(kXThreads), 0, st, A, M, Ks, lda, hi, lo, N, Ks, static_cast<const float*>(nullptr),
                       Cparts, (int64_t)N, flt, bt);
    timing_end(tok, st);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

}  // namespace evi

using namespace evi;

extern "C" size_t evi_gemm_nt_bf16x3_workspace_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return gemm_bf16x3_workspace_bytes(N, K);
}

extern "C" int evi_gemm_nt_bf16x3(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw,
                                  const float* bias, int act, float* C, int64_t ldc, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    EVI_REQUIRE(M >= 0 && N >= 0 && K >= 1, "evi_gemm_nt_bf16x3: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    EVI_REQUIRE(K % 4 == 0 && lda % 4 == 0, "evi_gemm_nt_bf16x3: K and lda must be multiples of 4, got K=%d lda=%lld", K,
                (long long)lda);
    EVI_REQUIRE(lda >= K && ldw >= K && ldc >= N, "evi_gemm_nt_bf16x3: leading dimension smaller than the row");
    EVI_REQUIRE(act >= 0 && act <= 2, "evi_gemm_nt_bf16x3: act must be 0 (none), 1 (tanh) or 2 (sigmoid)");
    if (M == 0 || N == 0) return EVI_OK;
    EVI_REQUIRE(A && W && C && workspace, "evi_gemm_nt_bf16x3: null pointer");
    if (workspace_bytes < gemm_bf16x3_workspace_bytes(N, K))
        return fail(EVI_ERR_NOMEM, "evi_gemm_nt_bf16x3: workspace %zu B < %zu B", workspace_bytes,
                    gemm_bf16x3_workspace_bytes(N, K));
    return launch_gemm_nt_bf16x3(A, M, K, lda, W, N, ldw, bias, act, C, ldc, workspace,
                                 reinterpret_cast<hipStream_t>(stream));
}
