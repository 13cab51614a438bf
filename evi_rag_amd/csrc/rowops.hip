// Row-wise kernels: L2 normalisation (C1).  HBM-bound: one read (+ one write) of the table.
//   reference: scripts/build_retrieval_pipeline.py:833-837 (_normalize_embeddings)
// One wave per row, 16-byte loads when D % 4 == 0, f32 accumulation, butterfly reduce.
#include "common.hpp"

#include <hip/hip_fp8.h>

namespace evi {

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ inline float row_sumsq(const float* __restrict__ row, int D, int lane) {
    float acc = 0.f;
    if ((D & 3) == 0) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        for (int c = lane; c < (D >> 2); c += 64) {
            const float4 v = r4[c];
            acc = fmaf(v.x, v.x, acc);
            acc = fmaf(v.y, v.y, acc);
            acc = fmaf(v.z, v.z, acc);
            acc = fmaf(v.w, v.w, acc);
        }
    } else {
        for (int c = lane; c < D; c += 64) acc = fmaf(row[c], row[c], acc);
    }
    return wave_sum(acc);
}

// MODE 0: inv_norm only.  MODE 1: write normalised rows (true division).  MODE 2: the norm itself (no clamp).
template <int MODE>
__global__ __launch_bounds__(256) void k_row_norm(const float* __restrict__ x, int64_t n, int D,
                                                  float eps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n; r += nwaves) {
        const float* row = x + r * (int64_t)D;
        const float ss = row_sumsq(row, D, lane);
        const float denom = fmaxf(sqrtf(ss), eps);
        if (MODE == 0) {
            if (lane == 0) out[r] = 1.0f / denom;
        } else if (MODE == 2) {
            if (lane == 0) out[r] = sqrtf(ss);
        } else {
            float* o = out + r * (int64_t)D;
            if ((D & 3) == 0) {
                const float4* r4 = reinterpret_cast<const float4*>(row);
                float4* o4 = reinterpret_cast<float4*>(o);
                for (int c = lane; c < (D >> 2); c += 64) {
                    float4 v = r4[c];
                    v.x /= denom;
                    v.y /= denom;
                    v.z /= denom;
                    v.w /= denom;
                    o4[c] = v;
                }
            } else {
                for (int c = lane; c < D; c += 64) o[c] = row[c] / denom;
            }
        }
    }
}

// Per-row symmetric quantisation to OCP e4m3: scale = max|x| / 448 (1 for an all-zero row),
// out = e4m3(x / scale) with round-to-nearest-even and saturation; x ~ out * scale.
__global__ __launch_bounds__(256) void k_quantize_rows_fp8(const float* __restrict__ x, int64_t n, int D,
                                                           uint8_t* __restrict__ out, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += nwaves) {
        const float* row = x + r * (int64_t)D;
        float mx = 0.f;
        for (int d = lane; d < D; d += 64) mx = fmaxf(mx, fabsf(row[d]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const float sc = mx > 0.f ? mx / 448.0f : 1.0f;
        if (lane == 0) scale[r] = sc;
        uint8_t* o = out + r * (int64_t)D;
        for (int d = lane; d < D; d += 64)
            o[d] = (uint8_t)__hip_cvt_float_to_fp8(row[d] / sc, __HIP_SATFINITE, __HIP_E4M3);
    }
}

static int norm_grid(int64_t n) {
    int64_t blocks = (n + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace evi

using namespace evi;

extern "C" int evi_row_inv_norm(const float* x, int64_t n, int D, float eps, float* inv_norm,
                                void* stream) {
    EVI_REQUIRE(n >= 0 && D >= 1, "evi_row_inv_norm: need n >= 0 and D >= 1, got n=%lld D=%d",
                (long long)n, D);
    if (n == 0) return EVI_OK;
    EVI_REQUIRE(x && inv_norm, "evi_row_inv_norm: null pointer");
    hipLaunchKernelGGL(k_row_norm<0>, dim3(norm_grid(n)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, n, D, eps, inv_norm);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_row_normalize(const float* x, int64_t n, int D, float eps, float* out, void* stream) {
    EVI_REQUIRE(n >= 0 && D >= 0, "evi_row_normalize: need n >= 0 and D >= 0, got n=%lld D=%d",
                (long long)n, D);
    if (n == 0 || D == 0) return EVI_OK;  // reference: empty tensors pass through (:834-835)
    EVI_REQUIRE(x && out, "evi_row_normalize: null pointer");
    hipLaunchKernelGGL(k_row_norm<1>, dim3(norm_grid(n)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, n, D, eps, out);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_row_norms(const float* x, int64_t n, int D, float* norms, void* stream) {
    EVI_REQUIRE(n >= 0 && D >= 1, "evi_row_norms: need n >= 0 and D >= 1, got n=%lld D=%d", (long long)n, D);
    if (n == 0) return EVI_OK;
    EVI_REQUIRE(x && norms, "evi_row_norms: null pointer");
    hipLaunchKernelGGL(k_row_norm<2>, dim3(norm_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n, D, 0.f,
                       norms);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_quantize_rows_fp8(const float* x, int64_t n, int D, uint8_t* out_fp8, float* out_scale,
                                     void* stream) {
    EVI_REQUIRE(n >= 0 && D >= 1, "evi_quantize_rows_fp8: need n >= 0 and D >= 1, got n=%lld D=%d", (long long)n, D);
    if (n == 0) return EVI_OK;
    EVI_REQUIRE(x && out_fp8 && out_scale, "evi_quantize_rows_fp8: null pointer");
    hipLaunchKernelGGL(k_quantize_rows_fp8, dim3(norm_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n,
                       D, out_fp8, out_scale);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
