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
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kOptThreads) {
        float g[4], w[4];
        if constexpr (sizeof(T) == 2) {
            const uint2 gb = *reinterpret_cast<const uint2*>(grad + 4 * i);
            g[0] = __uint_as_float(gb.x << 16); g[1] = __uint_as_float(gb.x & 0xffff0000u);
            g[2] = __uint_as_float(gb.y << 16); g[3] = __uint_as_float(gb.y & 0xffff0000u);
        } else {
            const float4 gb = *reinterpret_cast<const float4*>(grad + 4 * i);
            g[0] = gb.x; g[1] = gb.y; g[2] = gb.z; g[3] = gb.w;
        }
        float4 mm = *reinterpret_cast<const float4*>(m + 4 * i);
        float4 vv = *reinterpret_cast<const float4*>(v + 4 * i);
        float4 ww;
        if (master) ww = *reinterpret_cast<const float4*>(master + 4 * i);
        else {
            if constexpr (sizeof(T) == 4) ww = *reinterpret_cast<const float4*>(param + 4 * i);
            else ww = make_float4(0, 0, 0, 0);   // bf16 parameters always come with an f32 master copy
        }
        w[0] = ww.x; w[1] = ww.y; w[2] = ww.z; w[3] = ww.w;
        float ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = g[k] * gs;
            w[k] *= decay;
            ma[k] = beta1 * ma[k] + (1.0f - beta1) * gk;
            va[k] = beta2 * va[k] + (1.0f - beta2) * gk * gk;
            const float denom = sqrtf(va[k]) * rsbc2 + eps;
            w[k] -= step * (ma[k] / denom);
        }
        *reinterpret_cast<float4*>(m + 4 * i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
        *reinterpret_cast<float4*>(v + 4 * i) = make_float4(va[0], va[1], va[2], va[3]);
        if (master) *reinterpret_cast<float4*>(master + 4 * i) = make_float4(w[0], w[1], w[2], w[3]);
        if constexpr (sizeof(T) == 2) {
            uint2 o;
            o.x = (unsigned)f32_to_bf16(w[0]) | ((unsigned)f32_to_bf16(w[1]) << 16);
            o.y = (unsigned)f32_to_bf16(w[2]) | ((unsigned)f32_to_bf16(w[3]) << 16);
            *reinterpret_cast<uint2*>(param + 4 * i) = o;
        } else {
            *reinterpret_cast<float4*>(param + 4 * i) = make_float4(w[0], w[1], w[2], w[3]);
        }
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
    for (int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kOptThreads) {
        Vec16<T> a;
        a.load(x + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) acc = fmaf(a.v[k], a.v[k], acc);
    }
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
    int64_t blocks = rpo_cdiv(n / 4, kOptThreads);
    if (blocks > 256 * 8) blocks = 256 * 8;
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
