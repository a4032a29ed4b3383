// Flat-buffer AdamW step + gradient sum of squares (for clipping).  "next" row f2 of SURVEY.md §8: replaces the
// DeepSpeed ZeRO-1 bf16 optimizer step of the reference's training scripts (configs/ds_zero1_config_llama.json,
// scripts/train/run_contrastive.sh:33-40) with ONE launch over the whole parameter space.
//
// HBM-bound: per element it reads grad (s B) + master, m, v (12 B) and writes master, m, v (12 B) + param (s B):
// 28 B/element for bf16 parameters.  16-byte accesses, grid-stride, no reuse -> non-temporal where it pays.
#include "common.hpp"

namespace {

constexpr int kOptThreads = 256;

template <typename T>
__global__ __launch_bounds__(kOptThreads) void adamw_kernel(T* __restrict__ param, float* __restrict__ master,
                                                             const T* __restrict__ grad, float* __restrict__ m,
                                                             float* __restrict__ v, int64_t n, float lr, float beta1,
                                                             float beta2, float eps, float wd, float bc1, float bc2,
                                                             const float* __restrict__ grad_scale) {
    const float gs = grad_scale ? grad_scale[0] : 1.0f;
    const float step = lr / bc1;
    const float rsbc2 = 1.0f / sqrtf(bc2);
    const float decay = 1.0f - lr * wd;
    // ONE group of 4 elements per thread (16-byte f32 accesses, 8-byte bf16 accesses), block b owns the contiguous
    // groups [256 b, 256 b + 256), streaming loads / stores: 5.4 ms = 6.4 TB/s for 1.236 G parameters in
    // tools/exp/exp_adamw.hip (round 1; git history), vs 6.1 ms for a 2048-block grid-stride loop and 14.9 ms (!) for 8 elements per thread
    // (two 16-byte f32 accesses per lane at a 32-byte lane stride touch every 128-byte line twice).
    const int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x;
    if (i >= (n >> 2)) return;
    float g[4], w[4];
    if constexpr (sizeof(T) == 2) {
        const unsigned long long gb = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(grad + 4 * i));
        const unsigned lo = (unsigned)gb, hi = (unsigned)(gb >> 32);
        g[0] = __uint_as_float(lo << 16); g[1] = __uint_as_float(lo & 0xffff0000u);
        g[2] = __uint_as_float(hi << 16); g[3] = __uint_as_float(hi & 0xffff0000u);
    } else {
        const uint4_t gb = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(grad + 4 * i));
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = __uint_as_float(gb[k]);
    }
    const uint4_t mm = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(m + 4 * i));
    const uint4_t vv = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(v + 4 * i));
    uint4_t ww = {0u, 0u, 0u, 0u};
    if (master) ww = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(master + 4 * i));
    else if constexpr (sizeof(T) == 4) ww = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(param + 4 * i));
    float ma[4], va[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w[k] = __uint_as_float(ww[k]);
        ma[k] = __uint_as_float(mm[k]);
        va[k] = __uint_as_float(vv[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gk = g[k] * gs;
        w[k] *= decay;
        ma[k] = beta1 * ma[k] + (1.0f - beta1) * gk;
        va[k] = beta2 * va[k] + (1.0f - beta2) * gk * gk;
        const float denom = sqrtf(va[k]) * rsbc2 + eps;
        w[k] -= step * (ma[k] / denom);
    }
    uint4_t om, ov, ow;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        om[k] = __float_as_uint(ma[k]);
        ov[k] = __float_as_uint(va[k]);
        ow[k] = __float_as_uint(w[k]);
    }
    __builtin_nontemporal_store(om, reinterpret_cast<uint4_t*>(m + 4 * i));
    __builtin_nontemporal_store(ov, reinterpret_cast<uint4_t*>(v + 4 * i));
    if (master) __builtin_nontemporal_store(ow, reinterpret_cast<uint4_t*>(master + 4 * i));
    if constexpr (sizeof(T) == 2) {
        const unsigned lo = (unsigned)f32_to_bf16(w[0]) | ((unsigned)f32_to_bf16(w[1]) << 16);
        const unsigned hi = (unsigned)f32_to_bf16(w[2]) | ((unsigned)f32_to_bf16(w[3]) << 16);
        __builtin_nontemporal_store((unsigned long long)lo | ((unsigned long long)hi << 32),
                                    reinterpret_cast<unsigned long long*>(param + 4 * i));
    } else {
        __builtin_nontemporal_store(ow, reinterpret_cast<uint4_t*>(param + 4 * i));
    }
}

// partial[b] = sum over this block's grid-stride share of grad^2 (f32 accumulation, fixed order).
template <typename T>
__global__ __launch_bounds__(kOptThreads) void sumsq_kernel(const T* __restrict__ x, int64_t n,
                                                             float* __restrict__ partial) {
    __shared__ float s_red[kOptThreads / 64];
    constexpr int V = Elem<T>::kVec;
    float acc = 0.f;
    const int64_t nv = n / V;
    // block b owns a contiguous chunk of vectors (rounded up to whole 256-vector rows)
    const int64_t per = ((nv + gridDim.x - 1) / gridDim.x + kOptThreads - 1) / kOptThreads * kOptThreads;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < nv ? lo + per : nv;
    // four independent 16-byte loads in flight per thread (a single load per loop trip left 16 KB in flight per CU with this
    // grid: 4.1-5.5 TB/s), four accumulators, fixed order
    float acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int64_t i = lo + threadIdx.x;
    for (; i + 3 * kOptThreads < hi; i += 4 * kOptThreads) {
        Vec16<T> a, b, c, d;
        a.load_nt(x + i * V);
        b.load_nt(x + (i + kOptThreads) * V);
        c.load_nt(x + (i + 2 * kOptThreads) * V);
        d.load_nt(x + (i + 3 * kOptThreads) * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            acc = fmaf(a.v[k], a.v[k], acc);
            acc1 = fmaf(b.v[k], b.v[k], acc1);
            acc2 = fmaf(c.v[k], c.v[k], acc2);
            acc3 = fmaf(d.v[k], d.v[k], acc3);
        }
    }
    for (; i < hi; i += kOptThreads) {
        Vec16<T> a;
        a.load_nt(x + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) acc = fmaf(a.v[k], a.v[k], acc);
    }
    acc = (acc + acc1) + (acc2 + acc3);
    if (blockIdx.x == 0)
        for (int64_t i = nv * V + threadIdx.x; i < n; i += kOptThreads) {
            const float a = Elem<T>::ld(x + i);
            acc = fmaf(a, a, acc);
        }
    acc = block_sum<kOptThreads / 64>(acc, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

}  // namespace

extern "C" int rpo_adamw_step(void* param, float* master, const void* grad, float* exp_avg, float* exp_avg_sq,
                              int64_t n, int dtype, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float bias_corr1, float bias_corr2, const float* grad_scale,
                              rpo_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0) return RPO_ERR_INVALID_ARG;
    if (dtype == RPO_DT_BF16 && !master) return RPO_ERR_INVALID_ARG;
    if (n % 4 != 0) return RPO_ERR_UNSUPPORTED;   // flat buffers are padded to 16-byte multiples by the caller
    if (!rpo_aligned16(param) || !rpo_aligned16(grad) || !rpo_aligned16(exp_avg) || !rpo_aligned16(exp_avg_sq) ||
        (master && !rpo_aligned16(master)))
        return RPO_ERR_UNSUPPORTED;
    const int64_t blocks = rpo_cdiv(n / 4, kOptThreads);
    if (blocks >= INT32_MAX) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(adamw_kernel<bf16_t>, dim3((unsigned)blocks), dim3(kOptThreads), 0, st, (bf16_t*)param,
                           master, (const bf16_t*)grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                           bias_corr1, bias_corr2, grad_scale);
    else if (dtype == RPO_DT_F32)
        RPO_LAUNCH(adamw_kernel<float>, dim3((unsigned)blocks), dim3(kOptThreads), 0, st, (float*)param,
                           master, (const float*)grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                           bias_corr1, bias_corr2, grad_scale);
    else
        return RPO_ERR_INVALID_ARG;
    return rpo_launch_status();
}

extern "C" int rpo_sumsq_partial(const void* x, int64_t n, int dtype, float* partial_out, int nblocks,
                                 rpo_stream_t stream) {
    if (!x || !partial_out || n <= 0 || nblocks <= 0) return RPO_ERR_INVALID_ARG;
    if (!rpo_aligned16(x)) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(sumsq_kernel<bf16_t>, dim3((unsigned)nblocks), dim3(kOptThreads), 0, st, (const bf16_t*)x, n,
                           partial_out);
    else if (dtype == RPO_DT_F32)
        RPO_LAUNCH(sumsq_kernel<float>, dim3((unsigned)nblocks), dim3(kOptThreads), 0, st, (const float*)x, n,
                           partial_out);
    else
        return RPO_ERR_INVALID_ARG;
    return rpo_launch_status();
}
