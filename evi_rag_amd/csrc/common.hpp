// Shared host/device helpers for the evi_hip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string>

#include "../../include/evi_hip.h"

namespace evi {

// ---- error plumbing -------------------------------------------------------------------------
std::string& last_error_ref();
int fail(int code, const char* fmt, ...);

#define EVI_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::evi::fail(EVI_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define EVI_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            return ::evi::fail(EVI_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                               hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

#define EVI_LAUNCH_CHECK()                                                               \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess)                                                            \
            return ::evi::fail(EVI_ERR_HIP, "kernel launch failed: %s (%s:%d)",          \
                               hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- optional per-kernel timing (bench.py's roofline leg) -----------------------------------
// When enabled, launch sites bracket their dominant kernel with hipEvents on the call's stream;
// evi_timing_read() synchronises those events and returns the summed durations per class.
enum TimingClass { kTimeCosineScore = 0, kTimeSelect = 1, kTimeGemm = 2, kTimeEdge = 3, kTimeClasses = 8 };
bool timing_enabled();
// Records a start event on `st`; returns a token (or -1 when timing is off).
int timing_begin(int cls, hipStream_t st);
void timing_end(int token, hipStream_t st);
// For a span that is ONE kernel: a (start, stop) event pair to hand to hipExtLaunchKernelGGL, which stamps them with
// the dispatch's own begin / end times.  Unlike timing_begin/end this puts no event packets between the kernels of the
// stream (each recorded event costs ~5 us of gap, ~60 us per scan step), so a timed run runs like an untimed one.
// Returns false (and null events) when timing is off.
bool timing_kernel_events(int cls, hipEvent_t* start, hipEvent_t* stop);

// ---- shared launchers ------------------------------------------------------------------------
// C[M,N] = act(A[M,K] W[N,K]^T + bias); act: 0 none, 1 tanh, 2 sigmoid (gemm.hip).
int launch_gemm_nt(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw,
                   const float* bias, int act, float* C, int64_t ldc, hipStream_t st);

// Threshold-filter epilogue of the split-bf16 GEMM (many-query top-k): instead of being stored, the element
// (row m, column n) is appended to column n's candidate list when value * row_scale[row0 + m] >= tau[n].
struct GemmFilter {
    const float* tau;        // [N]
    const float* row_scale;  // [>= row0 + M] or null
    int64_t row0;            // global row of A's first row
    float* cand_score;       // [N, cap]
    int32_t* cand_id;        // [N, cap]
    int32_t* cand_cnt;       // [N * cnt_stride]
    int32_t* status;         // |= 2 when a list is full
    int cap, cnt_stride;
};
// W [N, K] -> bf16 hi / lo planes in wsplit (once); then C = A W^T with the planes, or the filter epilogue.
// gridDim.y-batched K-slices of one product (split-K); count == 0: a plain GEMM
struct GemmBatch {
    int count = 0;
    int64_t c_stride = 0;  // floats between the slices' outputs
    int64_t k_total = 0;   // padded K of the whole operands
    int64_t w_ld = 0;      // row stride of the W planes (elements)
    const int32_t* m_dev = nullptr;  // device-side row count: rows >= *m_dev are not computed (their tiles exit); M is the bound
};
bool gemm_skinny_fits(int64_t M, int K, int64_t lda, int64_t ldw);
int launch_gemm_skinny(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw, const float* bias, int act,
                       float* C, int64_t ldc, hipStream_t st);
void gemm_tn_plan(int M, int N, int64_t K, int max_slices, int64_t* Ks_out, int* S_out);
int launch_gemm_tn_bf16x3(const float* A, int64_t lda, int M, const float* B, int64_t ldb, int N, int64_t K, int64_t Ks, int S,
                          float* Cparts, hipStream_t st, int single = 0);
int launch_gemm_nt_bf16x3_splitk(const float* A, int64_t M, int64_t Ktot, int64_t lda, const float* W, int N, int64_t ldw, int Ks,
                                 int S, float* Cparts, void* wsplit, hipStream_t st);
int split_weight_bf16x3(const float* W, int N, int K, int64_t ldw, void* wsplit, hipStream_t st);
// a_f16: 0 = A is f32, 1 = A stored as f16, 2 = A is a bf16 copy (implies single);
// single: one product (hi * hi, plain bf16 accuracy) instead of three
int launch_gemm_bf16x3_filter(const void* A, int a_f16, int single, int64_t M, int K, int64_t lda, const void* wsplit, int N,
                              const GemmFilter& flt, hipStream_t st);
// C = A W^T with W already split (plain store epilogue)
int launch_gemm_bf16_presplit(const void* A, int a_f16, int single, int64_t M, int K, int64_t lda, const void* wsplit, int N,
                              float* C, int64_t ldc, hipStream_t st);
// Split-bf16 variant (gemm_bf16x3.hip); wsplit: gemm_bf16x3_workspace_bytes(N, K) bytes.
size_t gemm_bf16x3_workspace_bytes(int N, int K);
int launch_gemm_nt_bf16x3(const float* A, int64_t M, int K, int64_t lda, const float* W, int N, int64_t ldw,
                          const float* bias, int act, float* C, int64_t ldc, void* wsplit, hipStream_t st, int single = 0,
                          const int32_t* m_dev = nullptr);
int launch_gemm_nt_bf16x3_wplanes(const float* A, int64_t M, int K, int64_t lda, const void* wplanes, int N,
                                  const float* bias, int act, float* C, int64_t ldc, hipStream_t st, int single = 0,
                                  const int32_t* m_dev = nullptr);

// ---- ordered keys ---------------------------------------------------------------------------
// A 64-bit key whose unsigned order is the ranking order used everywhere in this library:
// larger score first, then smaller index first.  -0.0 ranks with +0.0; NaN ranks above +inf
// (torch.topk's convention).  key 0 is reserved for padding: it is smaller than any real key
// (the smallest real high word is that of -inf = 0x007FFFFF).
__host__ __device__ inline uint32_t float_to_ordered(float f) {
    if (f != f) return 0xFFFFFFFFu;
    f += 0.0f;  // -0.0 -> +0.0
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ordered_to_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}
__host__ __device__ inline uint64_t make_key(float score, uint32_t index) {
    return ((uint64_t)float_to_ordered(score) << 32) | (uint64_t)(0xFFFFFFFFu - index);
}
__host__ __device__ inline float key_score(uint64_t key) { return ordered_to_float((uint32_t)(key >> 32)); }
__host__ __device__ inline uint32_t key_index(uint64_t key) { return 0xFFFFFFFFu - (uint32_t)key; }

// ---- block-wide exact top-k -----------------------------------------------------------------
constexpr int kSelectThreads = 1024;
constexpr int kSortCap = 8192;  // keys a block can sort in LDS (64 KiB)
// Keys per thread per trip of the radix-select loops.  The loads of a trip are UNCONDITIONAL (an index past the end is
// clamped to the last entry and the key ignored afterwards): behind a per-key bounds branch the compiler waits for
// every load before it issues the next, and the loop runs at one memory latency per key (~75 us for the three passes
// over the 65 536 candidates of a dense first segment).
constexpr int kRadixUnroll = 8;
__device__ inline int64_t radix_clamp(int64_t i, int64_t cnt) { return i < cnt ? i : cnt - 1; }

struct SelectShared {
    uint64_t keys[kSortCap];
    uint32_t hist[256];
    uint32_t scalar[4];  // [0] gathered count, [1] chosen digit, [2] keys above digit, [3] spare
};

__device__ inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Sorts sh.keys[0..n_pow2) descending (n_pow2 a power of two <= kSortCap). All threads call.
__device__ inline void block_bitonic_sort_desc(SelectShared& sh, int n_pow2) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int size = 2; size <= n_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (n_pow2 >> 1); t += nt) {
                int i = 2 * t - (t & (stride - 1));
                int j = i + stride;
                uint64_t a = sh.keys[i], b = sh.keys[j];
                bool desc = ((i & size) == 0);
                bool swap = desc ? (a < b) : (a > b);
                if (swap) {
                    sh.keys[i] = b;
                    sh.keys[j] = a;
                }
            }
        }
    }
    __syncthreads();
}

// Picks, on wave 0, the digit d (255..0) whose bucket holds the need-th largest matching key:
// scalar[1] = d, scalar[2] = number of matching keys in buckets above d.  Call with all threads;
// a barrier must precede (hist complete) and follow (scalars visible).
__device__ inline void radix_pick_digit(SelectShared& sh, uint32_t need) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        // lane l owns digits 255-4l .. 252-4l (descending), 4 per lane
        uint32_t c[4];
        uint32_t local = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j] = sh.hist[255 - (4 * tid + j)];
            local += c[j];
        }
        uint32_t incl = local;  // inclusive scan over lanes = keys in this lane's digits and above
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (tid >= off) incl += up;
        }
        uint32_t before = incl - local;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (before < need && need <= before + c[j]) {
                sh.scalar[1] = 255u - (uint32_t)(4 * tid + j);
                sh.scalar[2] = before;
            }
            before += c[j];
        }
    }
}

// Sorts 1024 keys, one per thread of a 1024-thread block, descending; returns this thread's key of
// the sorted sequence (position tid).  Strides below 64 exchange through wave shuffles, strides
// >= 64 through `xchg` (2 x 1024 keys of LDS, double-buffered: one barrier per LDS stage).
__device__ inline uint64_t block_sort1024_desc(uint64_t key, uint64_t* xchg) {
    const int tid = threadIdx.x;
    int buf = 0;
    for (int size = 2; size <= 1024; size <<= 1) {
        const bool desc = (tid & size) == 0;
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            uint64_t other;
            if (stride < 64) {
                const uint32_t lo = __shfl_xor((uint32_t)key, stride, 64);
                const uint32_t hi = __shfl_xor((uint32_t)(key >> 32), stride, 64);
                other = ((uint64_t)hi << 32) | lo;
            } else {
                uint64_t* b = xchg + buf * 1024;
                b[tid] = key;
                __syncthreads();
                other = b[tid ^ stride];
                buf ^= 1;
            }
            const bool lower = (tid & stride) == 0;  // this thread holds the lower position of the pair
            const bool want_max = lower == desc;
            const bool take = want_max ? (other > key) : (other < key);
            if (take) key = other;
        }
    }
    return key;
}

// Exact top-k of `cnt` distinct keys produced by load(i), i in [0, cnt).  On return
// sh.keys[0..m) holds the m = min(cnt, k) largest keys in descending order (visible to all
// threads).  k <= EVI_TOPK_MAX_K <= kSortCap.  Keys equal to 0 are padding and sort last.
// Requires blockDim.x == kSelectThreads (1024).
//
// An MSB-first 8-bit radix select first narrows the list to the k-th key's bucket plus everything
// above it, until the survivors fit the sorter: 1024 keys (one per thread, shuffle/LDS hybrid
// bitonic, ~10 barriers) when k <= 1024, else 8192 keys (LDS bitonic).
template <class Load>
__device__ inline int block_topk(SelectShared& sh, Load load, int64_t cnt, int k) {
    const int tid = threadIdx.x, nt = blockDim.x;
    constexpr int kFast = 1024;
    if (cnt <= 0) return 0;
    int got;
    if (cnt <= kFast) {
        got = (int)cnt;
        const uint64_t key = tid < cnt ? load(tid) : 0ull;
        __syncthreads();  // callers may still be reading sh.keys from a previous use
        const uint64_t sorted = block_sort1024_desc(key, sh.keys);
        __syncthreads();
        sh.keys[tid] = sorted;
        __syncthreads();
        return got < k ? got : k;
    }
    const uint32_t limit = k <= kFast ? kFast : kSortCap;
    if (cnt <= (int64_t)limit) {  // only reachable with limit == kSortCap
        const int p2 = next_pow2((int)cnt);
        for (int i = tid; i < p2; i += nt) sh.keys[i] = (i < cnt) ? load(i) : 0ull;
        block_bitonic_sort_desc(sh, p2);
        return (int)(cnt < k ? cnt : k);
    }
    // Radix select.  Invariant: exactly `above` keys are > every key matching (prefix, mask),
    // and the k-th largest key matches (prefix, mask).
    uint64_t prefix = 0, mask = 0;
    uint32_t above = 0;
    got = 0;
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int i = tid; i < 256; i += nt) sh.hist[i] = 0;
        __syncthreads();
        // kRadixUnroll keys per thread per trip, all loads issued before the first atomic: the loop is bound by the
        // latency of the global loads (64 trips for the 65 536 keys of a dense first segment), not by their bytes
        for (int64_t i0 = tid; i0 < cnt; i0 += (int64_t)kRadixUnroll * nt) {
            uint64_t key[kRadixUnroll];
#pragma unroll
            for (int u = 0; u < kRadixUnroll; ++u) key[u] = load(radix_clamp(i0 + (int64_t)u * nt, cnt));
#pragma unroll
            for (int u = 0; u < kRadixUnroll; ++u)
                if (i0 + (int64_t)u * nt < cnt && (key[u] & mask) == prefix) atomicAdd(&sh.hist[(key[u] >> shift) & 0xFF], 1u);
        }
        __syncthreads();
        radix_pick_digit(sh, (uint32_t)k - above);
        __syncthreads();
        const uint32_t d = sh.scalar[1];
        const uint32_t in_bucket = sh.hist[d];
        above += sh.scalar[2];
        prefix |= (uint64_t)d << shift;
        mask |= 0xFFull << shift;
        __syncthreads();
        // Survivors = keys above the bucket + keys in the bucket. If they fit, sort them.
        if (above + in_bucket <= limit || pass == 0) {
            if (tid == 0) sh.scalar[0] = 0;
            __syncthreads();
            for (int64_t i0 = tid; i0 < cnt; i0 += (int64_t)kRadixUnroll * nt) {
                uint64_t key[kRadixUnroll];
#pragma unroll
                for (int u = 0; u < kRadixUnroll; ++u) key[u] = load(radix_clamp(i0 + (int64_t)u * nt, cnt));
#pragma unroll
                for (int u = 0; u < kRadixUnroll; ++u)
                    if (i0 + (int64_t)u * nt < cnt && (key[u] & mask) >= prefix) {
                        const uint32_t pos = atomicAdd(&sh.scalar[0], 1u);
                        if (pos < (uint32_t)kSortCap) sh.keys[pos] = key[u];
                    }
            }
            __syncthreads();
            got = (int)sh.scalar[0];
            if (got > kSortCap) got = kSortCap;  // unreachable for distinct keys
            break;
        }
    }
    if (got <= kFast) {
        const uint64_t key = tid < got ? sh.keys[tid] : 0ull;
        __syncthreads();
        const uint64_t sorted = block_sort1024_desc(key, sh.keys + kFast);  // exchange area beyond the survivors
        __syncthreads();
        sh.keys[tid] = sorted;
        __syncthreads();
    } else {
        const int p2 = next_pow2(got);
        for (int i = got + tid; i < p2; i += nt) sh.keys[i] = 0ull;
        block_bitonic_sort_desc(sh, p2);
    }
    return got < k ? got : k;
}

// The k-th largest of `cnt` distinct keys (1 <= k <= cnt), by 8 MSB-first radix passes; no limit on
// k or cnt.  All threads call; the result is uniform across the block.
template <class Load>
__device__ inline uint64_t block_kth_largest(SelectShared& sh, Load load, int64_t cnt, int64_t k) {
    const int tid = threadIdx.x, nt = blockDim.x;
    uint64_t prefix = 0, mask = 0;
    int64_t need = k;  // rank of the wanted key inside the matching set
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int i = tid; i < 256; i += nt) sh.hist[i] = 0;
        __syncthreads();
        for (int64_t i0 = tid; i0 < cnt; i0 += (int64_t)kRadixUnroll * nt) {
            uint64_t key[kRadixUnroll];
#pragma unroll
            for (int u = 0; u < kRadixUnroll; ++u) key[u] = load(radix_clamp(i0 + (int64_t)u * nt, cnt));
#pragma unroll
            for (int u = 0; u < kRadixUnroll; ++u)
                if (i0 + (int64_t)u * nt < cnt && (key[u] & mask) == prefix) atomicAdd(&sh.hist[(key[u] >> shift) & 0xFF], 1u);
        }
        __syncthreads();
        radix_pick_digit(sh, (uint32_t)need);  // one wave scans the 256 buckets (need <= cnt < 2^32)
        __syncthreads();
        need -= sh.scalar[2];
        prefix |= (uint64_t)sh.scalar[1] << shift;
        mask |= 0xFFull << shift;
        __syncthreads();
    }
    return prefix;
}

}  // namespace evi
