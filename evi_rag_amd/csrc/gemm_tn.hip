// Split-bf16 TN GEMM:  C_s [M, N] = A[k-slice s, :M]^T * B[k-slice s, :N]  for A [K, M], B [K, N] f32 ROW-MAJOR and K long —
// the weight-gradient products of the scorer's backward (dW = dY^T X over tens of thousands of edge rows, M, N <= 1280).
//
// Same tile, LDS image and MFMA loop as the NT kernel (gemm_bf16x3.hip: 256 x 256 x 32, 8 waves, hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16, [row][32 bf16] images with the (row >> 2) & 3 chunk swizzle); what differs is the staging.
// Both operands are k-major in memory, the MFMA wants 8 consecutive k per lane: a thread owns one COLUMN of an operand and
// 8 consecutive k — 8 four-byte loads (the 64 lanes of a wave read 256 contiguous bytes of one row: coalesced), split into
// hi / lo bf16 in registers, one 16-byte LDS write per plane (lanes = consecutive image rows: conflict-free through the
// swizzle).  The transposes and the split pass over the operands that the NT kernel needed in front of it are gone.
// Split-K rides in the grid (slice-major, tile-minor): slice s writes its own [M, N] partial (ordered reduction by the caller: no float atomics).
#include "common.hpp"

#include <hip/hip_bf16.h>
#include <stdlib.h>

namespace evi {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TM = 256, TNN = 256, TK = 32;
constexpr int kTThreads = 512;
__device__ inline int slot_tn(int r, int c) { return r * 4 + (c ^ ((r >> 2) & 3)); }
}  // namespace

// P1 = 1: one product (hi * hi) — the `bf16-mixed` training precision; no lo planes, half the LDS.
// V2 = 1 (M, N, lda, ldb even, 8-byte aligned operands): a thread owns a PAIR of adjacent columns of an operand and 8 consecutive
// k — eight 8-byte loads per operand (a wave reads 512 contiguous bytes of a row) instead of sixteen 4-byte ones, two 16-byte
// LDS writes per plane (image rows 2p and 2p + 1).
template <int P1, int V2 = 0>
__global__ __launch_bounds__(kTThreads) void k_gemm_tn_bf16x3(const float* __restrict__ A, int64_t lda, int M,
                                                              const float* __restrict__ B, int64_t ldb, int N, int64_t K,
                                                              int64_t kslice, float* __restrict__ Cparts) {
    __shared__ uint4 sAhi[2][TM * 4], sAlo[P1 ? 1 : 2][P1 ? 1 : TM * 4], sBhi[2][TNN * 4], sBlo[P1 ? 1 : 2][P1 ? 1 : TNN * 4];  // 8 x 16 KiB
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // XCD-aware order (workgroups are dealt round-robin over the 8 XCDs, each with its own L2): every XCD gets a contiguous
    // run of (slice, tile) pairs with the tile index fastest, so the tiles of one K-slice — which re-read the same operand
    // panels — run on one XCD and share its L2 instead of fetching the panels once per XCD.  Bijective when the grid
    // divides by 8 (the launcher rounds the slice count), identity otherwise.
    const unsigned nwg = gridDim.x;
    const unsigned wg = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;
    const int nblocks_n = (N + TNN - 1) / TNN;
    const int tiles = ((M + TM - 1) / TM) * nblocks_n;
    const int slice = (int)(wg / tiles), tile = (int)(wg % tiles);
    const int64_t k_begin = (int64_t)slice * kslice;
    const int64_t k_end = k_begin + kslice < K ? k_begin + kslice : K;
    const int m0 = (tile / nblocks_n) * TM;
    const int n0 = (tile % nblocks_n) * TNN;
    float* C = Cparts + (int64_t)slice * M * N;

    // staging: unit i of a thread = (operand, column, group of 8 k).  i = 0, 1: A; i = 2, 3: B.  The column is the same for a
    // thread's four units (tid & 255) and the k group is wave-uniform (tid >> 8 plus 2 for the odd units): a row's address is
    // a scalar, the lane adds its column.  Loads are unconditional on clamped indices; out-of-range values are zeroed at the store.
    float rv[4][8];
    const int col = V2 ? 2 * (tid & 127) : (tid & 255);  // V2: the first column of the thread's pair
    const int kg0 = __builtin_amdgcn_readfirstlane(V2 ? (tid >> 7) : (tid >> 8));
    // clamped columns: a column past the edge re-reads the last valid one; what it produces lands in rows / columns of the
    // C tile that are never stored, so it needs no zeroing
    const unsigned ca = V2 ? (unsigned)(m0 + col + 1 < M ? m0 + col : M - 2) : (unsigned)(m0 + col < M ? m0 + col : M - 1);
    const unsigned cb = V2 ? (unsigned)(n0 + col + 1 < N ? n0 + col : N - 2) : (unsigned)(n0 + col < N ? n0 + col : N - 1);
    // V2: both units of an operand (its two columns) come from one 8-byte load per k row
    auto load_pair = [&](int opb, int64_t k0) {
        const float* src = opb ? B : A;
        const int64_t ld = opb ? ldb : lda;
        const unsigned c = opb ? cb : ca;
        const int64_t kb = k0 + 8 * kg0;  // scalar
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t k = kb + j < K ? kb + j : K - 1;
            const float2 x = *reinterpret_cast<const float2*>(src + k * ld + c);
            rv[2 * opb][j] = x.x;
            rv[2 * opb + 1][j] = x.y;
        }
    };
    auto load_unit = [&](int i, int64_t k0) {
        const bool opb = i >= 2;
        const float* src = opb ? B : A;
        const int64_t ld = opb ? ldb : lda;
        const unsigned c = opb ? cb : ca;
        const int64_t kb = k0 + 8 * (kg0 + 2 * (i & 1));  // scalar
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t k = kb + j < K ? kb + j : K - 1;
            rv[i][j] = (src + k * ld)[c];  // scalar row base + 32-bit lane offset
        }
    };
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    auto store_unit = [&](int i, int buf, int64_t k0) {
        const bool opb = i >= 2;
        const int kg = V2 ? kg0 : kg0 + 2 * (i & 1);
        const int64_t left = k_end - (k0 + 8 * kg);  // scalar: how many of the unit's 8 k are inside the slice
        f32x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rv[i][j];
        if (left < 8) {  // the slice's last tile only (uniform branch)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = j < left ? v[j] : 0.f;
        }
        const bf16x8 h = __builtin_convertvector(v, bf16x8);
        const int s = slot_tn(V2 ? col + (i & 1) : col, kg);
        *reinterpret_cast<bf16x8*>(opb ? &sBhi[buf][s] : &sAhi[buf][s]) = h;
        if (!P1) {
            const f32x8 hf = __builtin_convertvector(h, f32x8);
            const bf16x8 l = __builtin_convertvector(v - hf, bf16x8);
            *reinterpret_cast<bf16x8*>(opb ? &sBlo[P1 ? 0 : buf][P1 ? 0 : s] : &sAlo[P1 ? 0 : buf][P1 ? 0 : s]) = l;
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;

    auto load_all = [&](int64_t k0) {
        if (V2) {
            load_pair(0, k0);
            load_pair(1, k0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) load_unit(i, k0);
        }
    };
    load_all(k_begin);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_unit(i, 0, k_begin);
    __syncthreads();
    load_all(k_begin + TK);
    int cur = 0;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += TK) {
        // A fragments are read AHEAD groups before their MFMAs (one group of six MFMAs covers the LDS latency; with one
        // product a group is two MFMAs, so read two ahead — as in the NT kernel)
        constexpr int AHEAD = P1 ? 2 : 1, RING = AHEAD + 1;
        bf16x8 fah[RING], fal[RING], fbh[2][2], fbl[2][2];
        auto read_a = [&](int g) {
            const int c = ((g >> 2) << 1) + fh;
            const int row = wm * 128 + (g & 3) * 32 + fr;
            fah[g % RING] = *reinterpret_cast<const bf16x8*>(&sAhi[cur][slot_tn(row, c)]);
            if (!P1) fal[g % RING] = *reinterpret_cast<const bf16x8*>(&sAlo[P1 ? 0 : cur][P1 ? 0 : slot_tn(row, c)]);
        };
        auto read_b = [&](int ks) {
            const int c = (ks << 1) + fh;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + fr;
                fbh[ks][j] = *reinterpret_cast<const bf16x8*>(&sBhi[cur][slot_tn(row, c)]);
                if (!P1) fbl[ks][j] = *reinterpret_cast<const bf16x8*>(&sBlo[P1 ? 0 : cur][P1 ? 0 : slot_tn(row, c)]);
            }
        };
        read_b(0);
#pragma unroll
        for (int g = 0; g < AHEAD; ++g) read_a(g);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + AHEAD < 8) read_a(g + AHEAD);
            if (g == 2) read_b(1);
            const int i = g & 3, ks = g >> 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16& c = acc[i][j];
                if (!P1) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[g % RING], fbh[ks][j], c, 0, 0, 0);
                if (!P1) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[g % RING], fbl[ks][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[g % RING], fbh[ks][j], c, 0, 0, 0);
            }
            // one staging unit in the shadow of this group's MFMAs: split + write the next tile's unit, refetch its registers
            if (g < 4) {
                store_unit(g, cur ^ 1, k0 + TK);
                if (!V2) load_unit(g, k0 + 2 * TK);
                else if (g & 1) load_pair(g >> 1, k0 + 2 * TK);  // both units of the operand have been stored: refetch the pair
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        cur ^= 1;
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m < M) C[(int64_t)m * N + n] = acc[i][j][r];
            }
    }
}

// Cparts [S][M][N]: slice s = rows [s * Ks, min(K, (s + 1) Ks)) of A [K, M] (row stride lda) and B [K, N] (row stride ldb).
int launch_gemm_tn_bf16x3(const float* A, int64_t lda, int M, const float* B, int64_t ldb, int N, int64_t K, int64_t Ks, int S,
                          float* Cparts, hipStream_t st, int single) {
    if (M <= 0 || N <= 0 || S <= 0) return EVI_OK;
    const dim3 grid((unsigned)(((M + TM - 1) / TM) * ((N + TNN - 1) / TNN) * S));
    const int tok = timing_begin(kTimeGemm, st);
    // 8-byte staging loads when every column pair is aligned and inside the operands (EVI_TN_V2=0: the 4-byte form, for A/B runs)
    static const bool v2_ok = [] { const char* e = getenv("EVI_TN_V2"); return !(e && e[0] == '0'); }();
    const bool v2 = v2_ok && M >= 2 && N >= 2 && M % 2 == 0 && N % 2 == 0 && lda % 2 == 0 && ldb % 2 == 0 &&
                    (reinterpret_cast<uintptr_t>(A) & 7) == 0 && (reinterpret_cast<uintptr_t>(B) & 7) == 0;
    if (single && v2) hipLaunchKernelGGL((k_gemm_tn_bf16x3<1, 1>), grid, dim3(kTThreads), 0, st, A, lda, M, B, ldb, N, K, Ks, Cparts);
    else if (single) hipLaunchKernelGGL((k_gemm_tn_bf16x3<1, 0>), grid, dim3(kTThreads), 0, st, A, lda, M, B, ldb, N, K, Ks, Cparts);
    else if (v2) hipLaunchKernelGGL((k_gemm_tn_bf16x3<0, 1>), grid, dim3(kTThreads), 0, st, A, lda, M, B, ldb, N, K, Ks, Cparts);
    else hipLaunchKernelGGL((k_gemm_tn_bf16x3<0, 0>), grid, dim3(kTThreads), 0, st, A, lda, M, B, ldb, N, K, Ks, Cparts);
    timing_end(tok, st);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

// out[i] (+)= sum_s part[s * len + i], s ascending
__global__ void k_tn_reduce(const float* __restrict__ part, int S, int64_t len, float* __restrict__ out, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    float acc = accumulate ? out[i] : 0.f;
    for (int s = 0; s < S; ++s) acc += part[(int64_t)s * len + i];
    out[i] = acc;
}

// slice plan of a TN product: S slices of Ks rows (Ks a multiple of 32).  One workgroup fits per CU (128 KiB of LDS), so the
// slice count is chosen to fill the 256 CUs once — tiles * S as close below 256 as it gets — which also keeps the partial
// sums few; max_slices bounds the partial buffer.
void gemm_tn_plan(int M, int N, int64_t K, int max_slices, int64_t* Ks_out, int* S_out) {
    const int64_t Kp = (K + 31) / 32 * 32;
    const int tiles = ((M + TM - 1) / TM) * ((N + TNN - 1) / TNN);
    int64_t S = 256 / tiles;
    if (S < 1) S = 1;
    if (const char* e = getenv("EVI_TN_SLICES")) S = atoi(e) > 0 ? atoi(e) : S;
    if (S > max_slices) S = max_slices;
    int64_t Ks = ((Kp + S - 1) / S + 31) / 32 * 32;
    if (Ks < 256) Ks = 256;
    if ((Kp + Ks - 1) / Ks < S) S = (Kp + Ks - 1) / Ks;
    *Ks_out = Ks;
    *S_out = (int)S;
}

}  // namespace evi

using namespace evi;

extern "C" size_t evi_gemm_tn_bf16x3_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)64 * M * N * sizeof(float);
}

extern "C" int evi_gemm_tn_bf16x3(const float* A, int64_t lda, int M, const float* B, int64_t ldb, int N, int64_t K, float* C,
                                  int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    EVI_REQUIRE(M >= 0 && N >= 0 && K >= 0, "evi_gemm_tn_bf16x3: bad shape M=%d N=%d K=%lld", M, N, (long long)K);
    if (M == 0 || N == 0) return EVI_OK;
    EVI_REQUIRE(C && workspace, "evi_gemm_tn_bf16x3: null pointer");
    EVI_REQUIRE(lda >= M && ldb >= N, "evi_gemm_tn_bf16x3: leading dimension smaller than the row");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t len = (int64_t)M * N;
    if (K == 0) {
        if (!accumulate) EVI_HIP_CHECK(hipMemsetAsync(C, 0, (size_t)len * sizeof(float), st));
        return EVI_OK;
    }
    EVI_REQUIRE(A && B, "evi_gemm_tn_bf16x3: null operand");
    if (workspace_bytes < evi_gemm_tn_bf16x3_workspace_bytes(M, N))
        return fail(EVI_ERR_NOMEM, "evi_gemm_tn_bf16x3: workspace %zu B < %zu B", workspace_bytes, evi_gemm_tn_bf16x3_workspace_bytes(M, N));
    int64_t Ks;
    int S;
    gemm_tn_plan(M, N, K, 64, &Ks, &S);
    float* part = static_cast<float*>(workspace);
    int rc = launch_gemm_tn_bf16x3(A, lda, M, B, ldb, N, K, Ks, S, part, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tn_reduce, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, part, S, len, C, accumulate ? 1 : 0);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
