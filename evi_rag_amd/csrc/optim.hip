// Optimiser step of the retriever's training path over FLAT parameter / gradient buffers: the mirror keeps
// all 25 parameters (9.4 M floats at D = H = 1024) in one allocation, so a step is one gradient-norm pass and one fused AdamW
// pass — HBM-bound, 7 floats of traffic per parameter — instead of 25 x (clip, weight decay, two moment updates, update).
//   reference: torch.optim.AdamW built by setup_optimizer (src/utils/optimization.py:20-35; type adamw, lr 1e-3,
//   weight_decay 1e-4: configs/model/retriever_module.yaml:37-40) and Lightning's gradient_clip_val: 1.0
//   (configs/trainer/default.yaml:20 -> torch.nn.utils.clip_grad_norm_, L2 over all parameters).
#include "common.hpp"

namespace evi {

constexpr int kNormChunk = 8192;  // elements per workgroup of the norm's first stage

// stage 1: f64 sum of squares of a chunk (fixed order: a thread's strided elements, then the LDS tree)
__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
    __shared__ double red[256];
    const int64_t b = (int64_t)blockIdx.x * kNormChunk;
    const int64_t e = b + kNormChunk < n ? b + kNormChunk : n;
    double acc = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) {
        const double x = (double)g[i];
        acc += x * x;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// stage 2: the chunks in ascending order; norm_out = |scale| * sqrt(sum)
__global__ __launch_bounds__(256) void k_sumsq_final(const double* __restrict__ part, int64_t nparts, float scale, float* __restrict__ norm_out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) norm_out[0] = (float)(fabs((double)scale) * sqrt(red[0]));
}

struct AdamWArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay;
    float step_size;      // lr / (1 - beta1^t)
    float inv_sqrt_bc2;   // 1 / sqrt(1 - beta2^t)
    float grad_scale;     // e.g. 1 / world_size after a SUM all-reduce
    const float* grad_norm;  // device scalar (norm of grad_scale * g) or null: no clipping
    float max_norm;
};

// torch.optim.AdamW's single-tensor update, in its order of operations (torch/optim/adamw.py -> _single_tensor_adam with
// decoupled weight decay): p *= 1 - lr wd; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g g; p -= step_size m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void k_adamw(AdamWArgs a) {
    float clip = 1.0f;
    if (a.grad_norm) {
        const float c = a.max_norm / (a.grad_norm[0] + 1e-6f);  // clip_grad_norm_: clamped to 1
        clip = c < 1.0f ? c : 1.0f;
    }
    const float gs = a.grad_scale * clip;
    const float decay = 1.0f - a.lr * a.weight_decay;
    const int64_t n4 = a.n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        const float4 g4 = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        float* pp = &p.x;
        const float* gp = &g4.x;
        float* mp = &m.x;
        float* vp = &v.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float g = gp[c] * gs;
            pp[c] *= decay;
            mp[c] = mp[c] + (g - mp[c]) * (1.0f - a.beta1);
            vp[c] = vp[c] * a.beta2 + (1.0f - a.beta2) * g * g;
            const float denom = sqrtf(vp[c]) * a.inv_sqrt_bc2 + a.eps;
            pp[c] -= a.step_size * (mp[c] / denom);
        }
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
    }
    // tail (n not a multiple of 4)
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const float g = a.g[i] * gs;
        float p = a.p[i] * decay;
        const float m = a.m[i] + (g - a.m[i]) * (1.0f - a.beta1);
        const float v = a.v[i] * a.beta2 + (1.0f - a.beta2) * g * g;
        p -= a.step_size * (m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps));
        a.p[i] = p;
        a.m[i] = m;
        a.v[i] = v;
    }
}

}  // namespace evi

using namespace evi;

extern "C" size_t evi_grad_norm_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    return (size_t)((n + kNormChunk - 1) / kNormChunk + 1) * sizeof(double);
}

extern "C" int evi_grad_norm(const float* g, int64_t n, float scale, float* norm_out, void* workspace, size_t workspace_bytes,
                             void* stream) {
    EVI_REQUIRE(n >= 0 && norm_out, "evi_grad_norm: bad arguments");
    EVI_REQUIRE(n == 0 || g, "evi_grad_norm: null gradient");
    EVI_REQUIRE(workspace && workspace_bytes >= evi_grad_norm_workspace_bytes(n), "evi_grad_norm: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nparts = (n + kNormChunk - 1) / kNormChunk;
    double* part = static_cast<double*>(workspace);
    if (nparts > 0) hipLaunchKernelGGL(k_sumsq_partial, dim3((unsigned)nparts), dim3(256), 0, st, g, n, part);
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, st, part, nparts, scale, norm_out);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}

extern "C" int evi_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int64_t step, float grad_scale, const float* grad_norm, float max_norm,
                              void* stream) {
    EVI_REQUIRE(n >= 0 && step >= 1, "evi_adamw_step: n must be >= 0 and step >= 1 (got %lld, %lld)", (long long)n, (long long)step);
    if (n == 0) return EVI_OK;
    EVI_REQUIRE(p && g && m && v, "evi_adamw_step: null buffer");
    EVI_REQUIRE(lr >= 0.f && eps >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && weight_decay >= 0.f,
                "evi_adamw_step: invalid hyper-parameter");
    EVI_REQUIRE((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) % 16 == 0,
                "evi_adamw_step: buffers must be 16-byte aligned");
    AdamWArgs a;
    a.p = p, a.g = g, a.m = m, a.v = v, a.n = n;
    a.lr = lr, a.beta1 = beta1, a.beta2 = beta2, a.eps = eps, a.weight_decay = weight_decay;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.grad_scale = grad_scale;
    a.grad_norm = grad_norm;
    a.max_norm = max_norm;
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adamw, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    EVI_LAUNCH_CHECK();
    return EVI_OK;
}
