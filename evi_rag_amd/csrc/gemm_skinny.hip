// Skinny NT GEMM:  C[M, N] = act(A[M, K] W[N, K]^T + bias)  for M <= 32 rows (the question-side projections of the scorer:
// query_proj / q_gate / q_bias on B = 32 question rows, the one-row non-text embedding, their backward counterparts).
//
// On the 256 x 256 MFMA tile these are ONE workgroup column walking all of K: ~55-60 us each of pure latency, five of them
// per forward (6 % of it).  Here every output column is one wave: a lane owns k = 4 lane + 256 i (16-byte loads of the W row:
// the wave reads 1 KiB contiguous), multiplies it against the same k of all M rows of A (L1 / L2 hits: A is <= 160 KiB) with
// exact f32 FMAs, and the 64 partial sums per row are added by a butterfly.  N waves fill the chip; ~6 us per call.
#include "common.hpp"

namespace evi {

namespace {
typedef float f4s __attribute__((ext_vector_type(4)));
constexpr int kSkinnyRows = 32;

__device__ inline float act_s(float v, int act) {
    if (act == 1) return tanhf(v);
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}
}  // namespace

__global__ __launch_bounds__(256) void k_gemm_skinny(const float* __restrict__ A, int M, int K, int64_t lda,
                                                     const float* __restrict__ W, int N, int64_t ldw, const float* __restrict__ bias,
                                                     int act, float* __restrict__ C, int64_t ldc) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[kSkinnyRows];
#pragma unroll
    for (int m = 0; m < kSkinnyRows; ++m) acc[m] = 0.f;
    const float* wrow = W + (int64_t)n * ldw;
    for (int k = 4 * lane; k < K; k += 256) {
        const f4s w = *reinterpret_cast<const f4s*>(wrow + k);
#pragma unroll
        for (int m = 0; m < kSkinnyRows; ++m) {
            if (m < M) {  // M is uniform: no divergence
                const f4s a = *reinterpret_cast<const f4s*>(A + (int64_t)m * lda + k);
                acc[m] = fmaf(a[3], w[3], fmaf(a[2], w[2], fmaf(a[1], w[1], fmaf(a[0], w[0], acc[m]))));
            }
        }
    }
    float mine = 0.f;  // lane m keeps row m's sum
#pragma unroll
    for (int m = 0; m < kSkinnyRows; ++m) {
        float v = acc[m];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == m) mine = v;
    }
    if (lane < M) C[(int64_t)lane * ldc + n] = act_s(mine + (bias ? bias[n] : 0.f), act);
}

// M <= 32, K % 4 == 0, lda % 4 == 0, ldw % 4 == 0 (16-byte row loads)
bool gemm_skinny_fits(int64_t M, int K, int64_t lda, int64_t ldw) {
    return M >= 1 && M <= kSkinnyRows && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0;
}

int launch_gemm_skinny(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw, const float* bias, int act,
                       float* C, int64_t ldc, hipStream_t st) {
    if (M == 0 || N == 0) return EVI_OK;
    hipLaunchKernelGGL(k_gemm_skinny, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, A, (int)M, K, lda, W, N, ldw, bias, act, C, ldc);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

}  // namespace evi
