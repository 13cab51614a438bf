// Split-bf16 GEMM on PRE-SPLIT operands:  C[M,N] = act(A W^T + bias) with A and W both given as bf16 hi / lo planes
// ([rows, Kp] row-major, Kp a multiple of 32, zero padded), f32 out.  Same arithmetic as gemm_bf16x3.hip
// (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, f32 accumulate, the same MFMA order per output element — so
// the same bits), different data path:
//
//   gemm_bf16x3.hip takes f32 A and splits it while staging: global -> VGPR -> cvt/sub/cvt -> ds_write, in the shadow
//   of the MFMAs.  That costs ~80 VALU and 12 LDS-write instructions per thread per k-tile plus the staging registers,
//   and its SQ counters showed the matrix pipe busy only 67 % of the cycles.  The scorer's big GEMMs read operands that
//   OUR OWN kernels wrote one launch earlier (k_edge_features, k_state_combine), so those kernels write the hi / lo
//   planes directly (same bytes as the f32 row) and this kernel moves them global -> LDS by LDS-DMA
//   (global_load_lds_dwordx4: no VGPR, no VALU, no ds_write): 8 DMA instructions per thread per k-tile, one issued
//   behind each of the 8 MFMA groups.  The LDS image is lane-linear per DMA instruction (hardware rule), so the
//   XOR swizzle that keeps ds_read_b128 conflict-free is applied to the SOURCE address (cdna_hip_programming.md
//   rule 21): LDS slot s of a plane holds chunk (s & 3) ^ ((s >> 4) & 3) of row s >> 2.
//
// 256 x 256 x 32 tile, 8 waves (2 x 4), eight 32x32 accumulators per wave, two LDS stages of 64 KiB in ONE __shared__
// array (a second LDS object makes hipcc drain the DMA queue before every fragment read).  Tile t + 1 is in flight
// while tile t is multiplied (3 072 MFMA cycles per SIMD: far longer than an L2 round trip); the __syncthreads at the
// end of a k-tile waits for the DMA (vmcnt(0)) and flips the stages.
#include "common.hpp"

#include <hip/hip_bf16.h>
#include <stdlib.h>

namespace evi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PM = 256, PN = 256, PK = 32;
constexpr int kPsThreads = 512;

__device__ inline float act_ps(float v, int act) {
    if (act == 1) return tanhf(v);
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}

__device__ inline int slot_ps(int r, int c) { return r * 4 + (c ^ ((r >> 2) & 3)); }

template <int AUX = 0>
__device__ inline void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

template <int ACT, int DBG = 0>
__global__ __launch_bounds__(kPsThreads) void k_gemm_ps_bf16x3(
    const __bf16* __restrict__ Ahi, const __bf16* __restrict__ Alo, int64_t M, int Kp, const __bf16* __restrict__ Whi,
    const __bf16* __restrict__ Wlo, int N, const float* __restrict__ bias, float* __restrict__ C, int64_t ldc) {
    __shared__ uint4 smem[2][4][PM * 4];  // [stage][plane: A hi, A lo, W hi, W lo][256 rows x 4 slots of 16 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nblocks_n = (N + PN - 1) / PN;
    const unsigned nwg = gridDim.x;
    const unsigned tile = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;  // XCD-aware order
    const int64_t m0 = (int64_t)(tile / nblocks_n) * PM;
    const int n0 = (int)(tile % nblocks_n) * PN;

    // DMA piece p (0..7) of a k-tile: plane p >> 1, slots (p & 1) * 512 + tid.  The wave's 64 lanes fill 64 consecutive
    // slots (16 rows); lane -> (row, chunk) through the inverse swizzle.  Rows past the end re-read the last row: their
    // products land in output rows / columns that are never stored.
    const int s_lo = tid, s_hi = 512 + tid;  // the two slot indices this thread sources, one per half
    auto src_off = [&](int s, int64_t row0, int64_t rows) -> int64_t {
        const int r = s >> 2, c = (s & 3) ^ ((r >> 2) & 3);
        int64_t row = row0 + r;
        row = row < rows ? row : rows - 1;
        return row * Kp + c * 8;
    };
    const int64_t a_off[2] = {src_off(s_lo, m0, M), src_off(s_hi, m0, M)};
    const int64_t w_off[2] = {src_off(s_lo, n0, N), src_off(s_hi, n0, N)};
    auto stage_piece = [&](int stage, int p, int k0) {
        const int plane = p >> 1, half = p & 1;
        const __bf16* base = plane == 0 ? Ahi : (plane == 1 ? Alo : (plane == 2 ? Whi : Wlo));
        const int64_t off = (plane < 2 ? a_off[half] : w_off[half]) + k0;
        glds16<0>(base + off, &smem[stage][plane][half * 512 + wave * 64]);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int p = 0; p < 8; ++p) stage_piece(0, p, 0);
    __syncthreads();  // waits for the DMA (vmcnt(0)) and publishes stage 0

    int cur = 0;
    for (int k0 = 0; k0 < Kp; k0 += PK) {
        // past the last tile the DMA re-stages the tile being multiplied into the idle stage (nobody reads it): cheaper
        // than eight scalar branches in a loop that should stay straight-line
        const int kn = k0 + PK < Kp ? k0 + PK : k0;
        bf16x8 fah[2], fal[2], fbh[2][2], fbl[2][2];
        auto read_a = [&](int g) {
            const int c = ((g >> 2) << 1) + fh;
            const int row = wm * 128 + (g & 3) * 32 + fr;
            fah[g & 1] = *reinterpret_cast<const bf16x8*>(&smem[cur][0][slot_ps(row, c)]);
            fal[g & 1] = *reinterpret_cast<const bf16x8*>(&smem[cur][1][slot_ps(row, c)]);
        };
        auto read_b = [&](int ks) {
            const int c = (ks << 1) + fh;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + fr;
                fbh[ks][j] = *reinterpret_cast<const bf16x8*>(&smem[cur][2][slot_ps(row, c)]);
                fbl[ks][j] = *reinterpret_cast<const bf16x8*>(&smem[cur][3][slot_ps(row, c)]);
            }
        };
        read_b(0);
        read_a(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) read_a(g + 1);
            if (g == 2) read_b(1);
            const int i = g & 3, ks = g >> 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16& c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[g & 1], fbh[ks][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[g & 1], fbl[ks][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[g & 1], fbh[ks][j], c, 0, 0, 0);
            }
            if (DBG != 2) stage_piece(cur ^ 1, g, kn);  // one DMA instruction behind each MFMA group
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // every wave's DMA has landed (vmcnt(0)) and every read of `cur` is done
        cur ^= 1;
    }

    if (DBG == 1 || DBG == 2) {  // timing experiments: keep the accumulators alive, store one value per lane
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 123.456f) C[tid] = t;
        return;
    }
    if (m0 + PM <= M && n0 + PN <= N) {  // interior tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + fr;
            const float bv = bias ? bias[n] : 0.f;
            float* cp = C + (m0 + wm * 128 + 4 * fh) * ldc + n;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[(int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc] = act_ps(acc[i][j][r] + bv, ACT);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m < M) C[m * ldc + n] = act_ps(acc[i][j][r] + bv, ACT);
            }
    }
}

// x [rows, K] f32 (row stride ld) -> hi / lo bf16 planes [rows, Kp], zero padded
__global__ void k_split_rows(const float* __restrict__ x, int64_t rows, int K, int64_t ld, int Kp, __bf16* __restrict__ hi,
                             __bf16* __restrict__ lo) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * Kp) return;
    const int64_t r = i / Kp;
    const int k = (int)(i - r * Kp);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = k + e < K ? x[r * ld + k + e] : 0.f;
        h[e] = (__bf16)v;
        l[e] = (__bf16)(v - (float)h[e]);
    }
    *reinterpret_cast<bf16x4*>(hi + i) = h;
    *reinterpret_cast<bf16x4*>(lo + i) = l;
}

int split_rows_bf16x3(const float* x, int64_t rows, int K, int64_t ld, int Kp, void* hi, void* lo, hipStream_t st) {
    if (rows == 0) return EVI_OK;
    const int64_t quads = rows * Kp / 4;
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, x, rows, K, ld, Kp,
                       static_cast<__bf16*>(hi), static_cast<__bf16*>(lo));
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

int launch_gemm_ps_bf16x3(const void* Ahi, const void* Alo, int64_t M, int Kp, const void* Whi, const void* Wlo, int N,
                          const float* bias, int act, float* C, int64_t ldc, hipStream_t st) {
    if (M == 0 || N == 0) return EVI_OK;
    const dim3 grid((unsigned)(((M + PM - 1) / PM) * ((N + PN - 1) / PN)));
    const int tok = timing_begin(kTimeGemm, st);
#define EVI_LAUNCH_PSX(ACT)                                                                                          \
    hipLaunchKernelGGL((k_gemm_ps_bf16x3<ACT>), grid, dim3(kPsThreads), 0, st, static_cast<const __bf16*>(Ahi),        \
                       static_cast<const __bf16*>(Alo), M, Kp, static_cast<const __bf16*>(Whi),                         \
                       static_cast<const __bf16*>(Wlo), N, bias, C, ldc)
    // EVI_GEMM_PS_DEBUG = 1 / 2: ablation builds for timing only (wrong results): no epilogue stores / no stores and no DMA
    // in the k-loop.  Measured at M = 131 072, K = N = 768 (profiles/r02_gemm_ablation.txt): 0.444 ms as shipped,
    // 0.395 without the f32 stores, 0.324 without stores and DMA — the LDS-fed MFMA loop alone runs at 0.57 of the bf16
    // peak, the L2 -> LDS traffic costs 18 %, the f32 output stores 11 %.  Non-temporal A loads (-7 %) and non-temporal
    // C stores (+-0) were tried and dropped.
    static const int dbg = [] { const char* e = getenv("EVI_GEMM_PS_DEBUG"); return e ? atoi(e) : 0; }();
    if (dbg == 1) { hipLaunchKernelGGL((k_gemm_ps_bf16x3<0, 1>), grid, dim3(kPsThreads), 0, st, static_cast<const __bf16*>(Ahi), static_cast<const __bf16*>(Alo), M, Kp, static_cast<const __bf16*>(Whi), static_cast<const __bf16*>(Wlo), N, bias, C, ldc); timing_end(tok, st); return EVI_OK; }
    if (dbg == 2) { hipLaunchKernelGGL((k_gemm_ps_bf16x3<0, 2>), grid, dim3(kPsThreads), 0, st, static_cast<const __bf16*>(Ahi), static_cast<const __bf16*>(Alo), M, Kp, static_cast<const __bf16*>(Whi), static_cast<const __bf16*>(Wlo), N, bias, C, ldc); timing_end(tok, st); return EVI_OK; }
    switch (act) {
        case 1: EVI_LAUNCH_PSX(1); break;
        case 2: EVI_LAUNCH_PSX(2); break;
        default: EVI_LAUNCH_PSX(0);
    }
#undef EVI_LAUNCH_PSX
    timing_end(tok, st);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

}  // namespace evi

using namespace evi;

extern "C" int evi_split_rows_bf16(const float* x, int64_t rows, int K, int64_t ld, int Kp, void* hi, void* lo, void* stream) {
    EVI_REQUIRE(rows >= 0 && K >= 1 && ld >= K, "evi_split_rows_bf16: bad shape rows=%lld K=%d ld=%lld", (long long)rows, K,
                (long long)ld);
    EVI_REQUIRE(Kp >= K && Kp % 32 == 0, "evi_split_rows_bf16: Kp must be a multiple of 32 and >= K, got Kp=%d K=%d", Kp, K);
    if (rows == 0) return EVI_OK;
    EVI_REQUIRE(x && hi && lo, "evi_split_rows_bf16: null pointer");
    return split_rows_bf16x3(x, rows, K, ld, Kp, hi, lo, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int evi_gemm_nt_bf16x3_presplit(const void* Ahi, const void* Alo, int64_t M, int Kp, const void* Whi,
                                           const void* Wlo, int N, const float* bias, int act, float* C, int64_t ldc,
                                           void* stream) {
    EVI_REQUIRE(M >= 0 && N >= 0 && Kp >= 32 && Kp % 32 == 0, "evi_gemm_nt_bf16x3_presplit: bad shape M=%lld N=%d Kp=%d",
                (long long)M, N, Kp);
    EVI_REQUIRE(ldc >= N, "evi_gemm_nt_bf16x3_presplit: ldc smaller than the row");
    EVI_REQUIRE(act >= 0 && act <= 2, "evi_gemm_nt_bf16x3_presplit: act must be 0 (none), 1 (tanh) or 2 (sigmoid)");
    if (M == 0 || N == 0) return EVI_OK;
    EVI_REQUIRE(Ahi && Alo && Whi && Wlo && C, "evi_gemm_nt_bf16x3_presplit: null pointer");
    return launch_gemm_ps_bf16x3(Ahi, Alo, M, Kp, Whi, Wlo, N, bias, act, C, ldc, reinterpret_cast<hipStream_t>(stream));
}
