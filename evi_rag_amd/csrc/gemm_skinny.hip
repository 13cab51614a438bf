// Skinny NT GEMM:  C[M, N] = act(A[M, K] W[N, K]^T + bias)  for M <= 32 rows (the question-side projections of the scorer:
// query_proj / q_gate / q_bias on B = 32 question rows, the one-row non-text embedding, their backward counterparts).
//
// On the 256 x 256 MFMA tile these are ONE workgroup column walking all of K: ~55-60 us each of pure latency, five of them
// per forward (6 % of it).  Here every output column is one wave: a lane owns k = 4 lane + 256 i (16-byte loads of the W row:
// the wave reads 1 KiB contiguous), multiplies it against the same k of all M rows of A (L1 / L2 hits: A is <= 160 KiB) with
// exact f32 FMAs, and the 64 partial sums per row are added by a reduce-scatter butterfly.  N waves fill the chip.
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

// R = 32 or 4 accumulator rows (M <= R).  Rows past M re-read row 0 and are never stored: the R loads of a k step carry no
// branch between them, so they are all in flight together (a uniform `if (m < M)` around each load made every load wait for the
// one before it: 96 dependent L2 round trips, 25 us per call).
template <int R>
__global__ __launch_bounds__(256) void k_gemm_skinny(const float* __restrict__ A, int M, int K, int64_t lda,
                                                     const float* __restrict__ W, int N, int64_t ldw, const float* __restrict__ bias,
                                                     int act, float* __restrict__ C, int64_t ldc) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[R];
#pragma unroll
    for (int m = 0; m < R; ++m) acc[m] = 0.f;
    const float* wrow = W + (int64_t)n * ldw;
    for (int k = 4 * lane; k < K; k += 256) {
        const f4s w = *reinterpret_cast<const f4s*>(wrow + k);
        f4s a[R];
#pragma unroll
        for (int m = 0; m < R; ++m) a[m] = *reinterpret_cast<const f4s*>(A + (int64_t)(m < M ? m : 0) * lda + k);
#pragma unroll
        for (int m = 0; m < R; ++m) acc[m] = fmaf(a[m][3], w[3], fmaf(a[m][2], w[2], fmaf(a[m][1], w[1], fmaf(a[m][0], w[0], acc[m]))));
    }
    if (R == 32) {
        // 32 row sums over 64 lanes as a reduce-scatter butterfly: at offset 32, 16, 8, 4, 2 a lane keeps one half of its rows
        // and hands the other half to its partner (16 + 8 + 4 + 2 + 1 exchanges instead of 32 x 6 for a butterfly per row); the
        // lane pair (2 r, 2 r + 1) ends up with row r, one last exchange joins the pair
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int off = 32 >> s, cnt = 16 >> s;
            // bitwise select of the two VALUES (a `?:` on array elements is turned into a lane-dependent register index:
            // a 32-way compare-and-select chain per element)
            const unsigned up = (lane & off) ? 0xffffffffu : 0u;
#pragma unroll
            for (int j = 0; j < cnt; ++j) {
                const unsigned lo = __builtin_bit_cast(unsigned, acc[j]), hi = __builtin_bit_cast(unsigned, acc[(j + cnt) % R]);
                const float keep = __builtin_bit_cast(float, (hi & up) | (lo & ~up));
                const float send = __builtin_bit_cast(float, (lo & up) | (hi & ~up));
                acc[j] = keep + __shfl_xor(send, off, 64);
            }
        }
        const float mine = acc[0] + __shfl_xor(acc[0], 1, 64);
        const int row = lane >> 1;
        if ((lane & 1) == 0 && row < M) C[(int64_t)row * ldc + n] = act_s(mine + (bias ? bias[n] : 0.f), act);
    } else {
        float mine = 0.f;  // lane m keeps row m's sum
#pragma unroll
        for (int m = 0; m < R; ++m) {
            float v = acc[m];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == m) mine = v;
        }
        if (lane < M) C[(int64_t)lane * ldc + n] = act_s(mine + (bias ? bias[n] : 0.f), act);
    }
}

// M <= 32, K % 4 == 0, lda % 4 == 0, ldw % 4 == 0 (16-byte row loads)
bool gemm_skinny_fits(int64_t M, int K, int64_t lda, int64_t ldw) {
    return M >= 1 && M <= kSkinnyRows && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0;
}

int launch_gemm_skinny(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw, const float* bias, int act,
                       float* C, int64_t ldc, hipStream_t st) {
    if (M == 0 || N == 0) return EVI_OK;
    if (M <= 4) hipLaunchKernelGGL(k_gemm_skinny<4>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, A, (int)M, K, lda, W, N, ldw, bias, act, C, ldc);
    else hipLaunchKernelGGL(k_gemm_skinny<32>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, A, (int)M, K, lda, W, N, ldw, bias, act, C, ldc);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

}  // namespace evi
