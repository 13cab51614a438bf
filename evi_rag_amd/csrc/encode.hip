// Text-encoding tail (E2/E3): masked mean pooling of the encoder's last hidden state and the
// id-addressed scatter into the embedding table.  The transformer forward stays in PyTorch-ROCm.
//   reference: TextEncoder.encode, scripts/text_encode_utils.py:60-65;
//              encode_to_memmap/_write_chunk, scripts/text_encode_utils.py:70-146.
// HBM-bound: pooling reads b*L*D*s bytes once (lanes walk d, so every load is a coalesced row
// segment); the scatter moves n*D*4 bytes.
#include "common.hpp"

#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

namespace evi {

template <typename T>
__device__ inline float to_f32(T v);
template <>
__device__ inline float to_f32<float>(float v) { return v; }
template <>
__device__ inline float to_f32<__half>(__half v) { return __half2float(v); }
template <>
__device__ inline float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

__device__ inline float round_f16(float v) { return __half2float(__float2half_rn(v)); }

// out[b, d] = sum_l hid[b, l, d] * mask[b, l] / max(sum_l mask[b, l], eps), in the pooling dtype.
template <typename T>
__global__ void k_masked_mean_pool(const T* __restrict__ hid, const int64_t* __restrict__ mask, int b, int L, int D,
                                   int pool_fp16, float eps, float* __restrict__ out) {
    const int row = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= b || d >= D) return;
    const T* h = hid + (int64_t)row * L * D + d;
    const int64_t* m = mask + (int64_t)row * L;
    float acc = 0.f, cnt = 0.f;
    for (int l = 0; l < L; ++l) {
        const float mv = m[l] != 0 ? 1.f : 0.f;  // attention masks are 0/1
        float hv = to_f32<T>(h[(int64_t)l * D]);
        if (pool_fp16) hv = round_f16(hv);  // hidden.to(float16)
        acc = fmaf(hv, mv, acc);
        cnt += mv;
    }
    float res;
    if (pool_fp16) {
        // torch sums f16 with an f32 accumulator and rounds once; clamp and divide in f16
        const float s16 = round_f16(acc);
        const float den = fmaxf(round_f16(cnt), round_f16(eps));
        res = round_f16(s16 / den);
    } else {
        res = acc / fmaxf(cnt, eps);
    }
    out[(int64_t)row * D + d] = res;
}

// last[id] = max row index writing to id  (later rows win, as the reference's sequential loop)
__global__ void k_scatter_last_writer(const int64_t* __restrict__ ids, int64_t n, int64_t max_id,
                                      int32_t* __restrict__ last) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t id = ids[i];
    if (id < 0 || id > max_id) return;  // out-of-range ids are skipped (:144-145)
    atomicMax(&last[id], (int32_t)i);
}

__global__ void k_scatter_rows(const float* __restrict__ src, const int64_t* __restrict__ ids, int64_t n, int D,
                               int64_t max_id, const int32_t* __restrict__ last, float* __restrict__ table) {
    const int64_t i = blockIdx.x;
    const int64_t id = ids[i];
    if (id < 0 || id > max_id || last[id] != (int32_t)i) return;
    for (int d = threadIdx.x; d < D; d += blockDim.x) table[id * D + d] = src[i * D + d];
}

__global__ void k_fill_i32_neg(int32_t* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = -1;
}

}  // namespace evi

using namespace evi;

extern "C" int evi_masked_mean_pool(const void* hidden, int hidden_dtype, const int64_t* attention_mask, int b, int L,
                                    int D, int pool_fp16, float eps, float* out, void* stream) {
    EVI_REQUIRE(b >= 0 && L >= 0 && D >= 0, "evi_masked_mean_pool: bad shape b=%d L=%d D=%d", b, L, D);
    EVI_REQUIRE(hidden_dtype >= 0 && hidden_dtype <= 2, "evi_masked_mean_pool: hidden_dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    if (b == 0 || D == 0) return EVI_OK;
    EVI_REQUIRE(out && (L == 0 || (hidden && attention_mask)), "evi_masked_mean_pool: null pointer");
    EVI_REQUIRE(b <= 65535, "evi_masked_mean_pool: at most 65535 texts per call, got %d", b);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((D + 255) / 256, b), block(256);
    if (hidden_dtype == 0)
        hipLaunchKernelGGL(k_masked_mean_pool<float>, grid, block, 0, st, static_cast<const float*>(hidden),
                           attention_mask, b, L, D, pool_fp16, eps, out);
    else if (hidden_dtype == 1)
        hipLaunchKernelGGL(k_masked_mean_pool<__half>, grid, block, 0, st, static_cast<const __half*>(hidden),
                           attention_mask, b, L, D, pool_fp16, eps, out);
    else
        hipLaunchKernelGGL(k_masked_mean_pool<__hip_bfloat16>, grid, block, 0, st,
                           static_cast<const __hip_bfloat16*>(hidden), attention_mask, b, L, D, pool_fp16, eps, out);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" size_t evi_scatter_rows_workspace_bytes(int64_t max_embedding_id) {
    return (size_t)(max_embedding_id >= 0 ? max_embedding_id + 1 : 1) * sizeof(int32_t);
}

extern "C" int evi_scatter_rows(const float* src, const int64_t* ids, int64_t n, int D, float* table,
                                int64_t max_embedding_id, void* workspace, size_t workspace_bytes, void* stream) {
    EVI_REQUIRE(n >= 0 && D >= 0, "evi_scatter_rows: bad shape n=%lld D=%d", (long long)n, D);
    if (max_embedding_id < 0 || n == 0 || D == 0) return EVI_OK;
    EVI_REQUIRE(src && ids && table && workspace, "evi_scatter_rows: null pointer");
    EVI_REQUIRE(n < (int64_t)0x7FFFFFFF, "evi_scatter_rows: at most 2^31-1 rows per call");
    if (workspace_bytes < evi_scatter_rows_workspace_bytes(max_embedding_id))
        return fail(EVI_ERR_NOMEM, "evi_scatter_rows: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int32_t* last = static_cast<int32_t*>(workspace);
    const int64_t slots = max_embedding_id + 1;
    hipLaunchKernelGGL(k_fill_i32_neg, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, last, slots);
    hipLaunchKernelGGL(k_scatter_last_writer, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, n,
                       max_embedding_id, last);
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)n), dim3(256), 0, st, src, ids, n, D, max_embedding_id, last, table);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
