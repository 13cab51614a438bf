// f32 GEMM for the scorer:  C[M,N] = act(A[M,K] * W[N,K]^T + bias)   (both operands K-contiguous)
//
// Serves every dense contraction of the reference Retriever (src/models/components/retriever.py):
//   EmbeddingProjector Linear+Tanh (:436-441, projections.py:9-40), q_gate / q_bias (:157-158,464),
//   state_net.0 / state_net.4 / score inputs (:175-182,482-483).
//
// MFMA-bound (f32-input v_mfma_f32_32x32x2_f32: exact f32 FMA chains, 157 TF peak; gfx950 has no
// TF32).  128x128x32 block tile, 4 waves as 2x2, each wave 64x64 = four 32x32 accumulators.
// Tiles are staged global -> VGPR -> LDS (one tile of register prefetch) with the 16-byte chunks
// of each row pair XOR-swizzled so that every ds_read_b128 lane group hits 16 distinct slots.
// One b128 fragment read feeds 4 MFMAs: MFMA t of a slab consumes k = kk + 4*(lane>>5) + t for
// both operands, i.e. a fixed permutation of the k order inside each 8-wide slab.
#include "common.hpp"

namespace evi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kGemmThreads = 256;

enum Act { kActNone = 0, kActTanh = 1, kActSigmoid = 2 };

__device__ inline float apply_act(float v, int act) {
    if (act == kActTanh) return tanhf(v);
    if (act == kActSigmoid) return 1.0f / (1.0f + expf(-v));
    return v;
}

// LDS image of a [128 rows][32 floats] tile: 16-byte chunk c (0..7) of row r lives at float4 slot
//   (r >> 1) * 16 + ((((r & 1) << 3) | c) ^ ((r >> 1) & 15)).
__device__ inline int lds_slot(int r, int c) { return ((r >> 1) << 4) + ((((r & 1) << 3) | c) ^ ((r >> 1) & 15)); }

template <int ACT>
__global__ __launch_bounds__(kGemmThreads) void k_gemm_nt(
    const float* __restrict__ A, int64_t M, int K, int64_t lda, const float* __restrict__ W, int N,
    int64_t ldw, const float* __restrict__ bias, float* __restrict__ C, int64_t ldc) {
    __shared__ f32x4 sA[BM * BK / 4];
    __shared__ f32x4 sW[BN * BK / 4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves
    const int nblocks_n = (N + BN - 1) / BN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2),
    // so give every XCD a contiguous run of logical tiles — the n-tiles that share an A row panel
    // then hit the same L2 instead of fetching the panel once per XCD.  Pure speed; bijective
    // whenever the grid divides by 8, identity otherwise.
    const unsigned nwg = gridDim.x;
    const unsigned tile = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;
    const int64_t m0 = (int64_t)(tile / nblocks_n) * BM;
    const int n0 = (int)(tile % nblocks_n) * BN;

    // staging map: thread handles float4 chunks s = tid + 256 * i (i = 0..3): row = s >> 3, c = s & 7
    f32x4 ra[4], rw[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = tid + kGemmThreads * i;
            const int r = s >> 3, c = s & 7;
            const int k = k0 + c * 4;
            const int64_t am = m0 + r;
            const int wn_row = n0 + r;
            ra[i] = (am < M && k < K) ? *reinterpret_cast<const f32x4*>(A + am * lda + k) : zero4;
            rw[i] = (wn_row < N && k < K) ? *reinterpret_cast<const f32x4*>(W + (int64_t)wn_row * ldw + k) : zero4;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = tid + kGemmThreads * i;
            const int r = s >> 3, c = s & 7;
            sA[lds_slot(r, c)] = ra[i];
            sW[lds_slot(r, c)] = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fh = lane >> 5;
    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();  // previous tile fully consumed
        store_tiles();
        __syncthreads();
        if (k0 + BK < K) load_tiles(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            const int c = (kk >> 2) + fh;  // chunk holding k = kk + 4*fh .. +3
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[lds_slot(wm * 64 + i * 32 + fi, c)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sW[lds_slot(wn * 64 + j * 32 + fi, c)];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: lane holds column n = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fi;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m < M) C[m * ldc + n] = apply_act(acc[i][j][r] + bv, ACT);
            }
        }
    }
}

int launch_gemm_nt(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw,
                   const float* bias, int act, float* C, int64_t ldc, hipStream_t st) {
    if (M == 0 || N == 0) return EVI_OK;
    const dim3 grid((unsigned)(((M + BM - 1) / BM) * ((N + BN - 1) / BN)));
    const int tok = timing_begin(kTimeGemm, st);
    switch (act) {
        case kActTanh:
            hipLaunchKernelGGL(k_gemm_nt<kActTanh>, grid, dim3(kGemmThreads), 0, st, A, M, K, lda, W, N, ldw, bias, C, ldc);
            break;
        case kActSigmoid:
            hipLaunchKernelGGL(k_gemm_nt<kActSigmoid>, grid, dim3(kGemmThreads), 0, st, A, M, K, lda, W, N, ldw, bias, C, ldc);
            break;
        default:
            hipLaunchKernelGGL(k_gemm_nt<kActNone>, grid, dim3(kGemmThreads), 0, st, A, M, K, lda, W, N, ldw, bias, C, ldc);
    }
    timing_end(tok, st);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

}  // namespace evi

using namespace evi;

extern "C" int evi_gemm_nt_f32(const float* A, int64_t M, int K, int64_t lda, const float* W, int N,
                               int64_t ldw, const float* bias, int act, float* C, int64_t ldc, void* stream) {
    EVI_REQUIRE(M >= 0 && N >= 0 && K >= 1, "evi_gemm_nt_f32: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    EVI_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0,
                "evi_gemm_nt_f32: K, lda and ldw must be multiples of 4 (16-byte rows), got K=%d lda=%lld ldw=%lld",
                K, (long long)lda, (long long)ldw);
    EVI_REQUIRE(lda >= K && ldw >= K && ldc >= N, "evi_gemm_nt_f32: leading dimension smaller than the row");
    EVI_REQUIRE(act >= 0 && act <= 2, "evi_gemm_nt_f32: act must be 0 (none), 1 (tanh) or 2 (sigmoid)");
    if (M == 0 || N == 0) return EVI_OK;
    EVI_REQUIRE(A && W && C, "evi_gemm_nt_f32: null pointer");
    EVI_REQUIRE(((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (int64_t)0x7FFFFFFF, "evi_gemm_nt_f32: too many tiles");
    return launch_gemm_nt(A, M, K, lda, W, N, ldw, bias, act, C, ldc, reinterpret_cast<hipStream_t>(stream));
}
